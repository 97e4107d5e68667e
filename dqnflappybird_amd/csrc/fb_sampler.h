// MT19937 on one wave and CPython's random.sample on top of it -- device code shared by the replay kernels
// (fb_replay.hip) and the two launches that can carry the sampler as a rider: the env step (fb_env.hip, fb_vec_step) and the
// conv3 backward (fb_qnet.hip, fb_train_steps).
// Included inside each translation unit's anonymous namespace.
#pragma once

constexpr int FB_SAMPLE_MAXB = 256;          // indices per draw (4 register slots x 64 lanes)
constexpr int FB_SAMPLE_POOL = 1100, FB_SAMPLE_TAB = 1024;   // pool[]: the shrinking population (n <= setsize) / the selection in order; tab[]: the `selected` set
constexpr int FB_SAMPLE_LDS_WORDS = 624 + FB_SAMPLE_POOL + FB_SAMPLE_TAB;      // mt[624] + pool[] + tab[] (pool points at the last two)

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}

// Regenerate the 624-word block with 64 lanes.  Chunks go in increasing order; inside a chunk every
// lane reads before any lane writes, which preserves the sequential algorithm's dependencies
// (new values are needed at distance 227 behind, old values at distance 1 ahead).
__device__ __forceinline__ void mt_regen(uint32_t *mt, int lane) {
    for (int base = 0; base < 624; base += 64) {
        const int i = base + lane;
        uint32_t v = 0;
        if (i < 624) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[i == 623 ? 0 : i + 1] & 0x7fffffffu);
            v = mt[i + 397 < 624 ? i + 397 : i + 397 - 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        __builtin_amdgcn_wave_barrier();
        if (i < 624) mt[i] = v;
        __builtin_amdgcn_wave_barrier();
    }
}

// result i lives in register slot i >> 6 of lane i & 63; written with compile-time slot indices (a run-time index into the
// register array would put it in scratch memory -- and a kernel that touches scratch pays for it at every launch)
__device__ __forceinline__ void set_sel(long long (&sel)[4], int i, int lane, long long v) {
#pragma unroll
    for (int q = 0; q < 4; q++) if (q == (i >> 6) && lane == (i & 63)) sel[q] = v;
}

// random.sample(range(n), k) on one wave.  Set path (n > setsize, the normal case): the wave tempers up to
// 64 state words at once, ballots the ones below n and walks only those, in stream order, testing each
// against the already selected values with one wave-wide compare -- no per-word LDS round trip.  Words are
// consumed exactly like CPython consumes them (every getrandbits call, also rejected / duplicate ones, up
// to and including the word that completed the sample).  Pool path (n <= setsize): the modulus shrinks
// with every draw, so it stays word-serial.  Lane (i & 63) keeps result i in register slot i >> 6.
__device__ __forceinline__ void sample_cpython_body(const FbSampleCtx &P, int k, long long setsize,
                                                    long long *__restrict__ out, uint32_t *mt, int *pool) {
    const int lane = threadIdx.x;
    if (P.gate && lane == 0) fb_flag_store(&P.gate->c_entry, P.gate_val);      // (split schedule: whatever the caller's stream held before this launch is done)
    // the whole block is fetched beside the cursor (one round trip; a window that depends on the cursor would be two);
    // it is written back only when this call regenerated it -- otherwise the cursor alone
    for (int i = lane; i < 624; i += 64) mt[i] = P.mt->mt[i];
    uint32_t idx = P.mt->idx;
    bool regenerated = false;
    const long long n = P.n;
    if (k > n || k > FB_SAMPLE_MAXB) {
        if (lane == 0) *P.error = 2;                    // "Sample larger than population"
        for (int i = lane; i < k; i += 64) out[i] = 0;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    long long sel[4] = {-1, -1, -1, -1};
    if (n > setsize) {
        const int shift = __builtin_clz((uint32_t)n);   // 32 - n.bit_length()
        int i = 0;
        if (k <= 64) {
            // Optimistic path: if the next <= 64 words hold k candidates below n and those are pairwise distinct (all but
            // ~k^2 / 2n of the calls), they ARE the sample, in stream order -- decided with a ballot, a rank and one round
            // of wave shuffles instead of k dependent iterations.  Otherwise nothing has been consumed and the loop runs.
            if (idx >= 624) { mt_regen(mt, lane); idx = 0; regenerated = true; }
            const int avail = 624 - (int)idx, take = avail < 64 ? avail : 64;
            const uint32_t w = lane < take ? mt_temper(mt[idx + lane]) >> shift : 0xFFFFFFFFu;
            const bool valid = lane < take && w < (uint32_t)n;
            const unsigned long long mask = __ballot(valid);
            if (__popcll(mask) >= k) {
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                if (valid && rank < k) pool[rank] = (int)w;
                __builtin_amdgcn_wave_barrier();        // one wave: LDS executes its writes before its reads; this pins the order
                const uint32_t c = lane < k ? (uint32_t)pool[lane] : (0x80000000u | (uint32_t)lane);   // distinct sentinels past k
                bool dup = false;
                for (int d = 1; d < 64; d++) dup |= (uint32_t)__shfl((int)c, (lane + d) & 63) == c;
                if (!__any(dup)) {
                    if (lane < k) sel[0] = (long long)c;
                    idx += __builtin_ctzll(__ballot(valid && rank == k - 1)) + 1;     // up to and including the k-th candidate
                    i = k;
                }
            }
        }
        // The general loop, a window of up to 64 words at a time.  CPython's `selected` set is a hash table in LDS (tab[]: open addressing,
        // 1024 slots for at most 256 + 64 values): every candidate below n of the window is inserted with one compare-and-swap chain, and
        // an insert that finds its own value there is a clash -- with an earlier window or inside this one.  A window without a clash
        // (all but ~k^2 / n of the draws) is taken WHOLE, in stream order: rank by ballot, one LDS write per candidate into pool[] (the
        // selection in order).  A window with a clash takes its inserts back (every lane clears the slot it filled: the table is as the
        // windows before left it) and is walked word by word, one lane inserting.  (k = 256 from a million slots: ~25 us of dependent
        // per-word iterations; an all-pairs register test per window still ~15 -- the five-compare, 63-rotation test is ~2 us of issue
        // per window; as a rider this chain was the longest of the env launch.)
        int *tab = pool + FB_SAMPLE_POOL;
        const bool looped = i < k;
        if (looped) for (int t = lane; t < FB_SAMPLE_TAB; t += 64) tab[t] = -1;
        __builtin_amdgcn_wave_barrier();
        auto insert = [&](uint32_t v, int &slot) {               // true: v was there already
            uint32_t hsh = (v * 2654435761u) >> 22;
            for (;;) {
                const int old = atomicCAS(&tab[hsh], -1, (int)v);
                if (old == -1) { slot = (int)hsh; return false; }
                if (old == (int)v) return true;
                hsh = (hsh + 1) & (FB_SAMPLE_TAB - 1);
            }
        };
        while (i < k) {
            if (idx >= 624) { mt_regen(mt, lane); idx = 0; regenerated = true; }
            const int avail = 624 - (int)idx, take = avail < 64 ? avail : 64;
            const uint32_t w = lane < take ? mt_temper(mt[idx + lane]) >> shift : 0xFFFFFFFFu;
            const bool valid = lane < take && w < (uint32_t)n;
            unsigned long long mask = __ballot(valid);
            if (n < (1ll << 31)) {                                   // (values are stored as non-negative ints, -1 = empty)
                int slot = -1;
                const bool clash = valid && insert(w, slot);
                if (!__any(clash)) {
                    const int need = k - i, m = __popcll(mask), rank = __popcll(mask & ((1ull << lane) - 1ull));
                    if (valid && rank < need) pool[i + rank] = (int)w;
                    if (m >= need) { idx += __builtin_ctzll(__ballot(valid && rank == need - 1)) + 1; i = k; }      // up to and including the k-th
                    else { idx += take; i += m; }
                    continue;
                }
                if (slot >= 0) tab[slot] = -1;
                __builtin_amdgcn_wave_barrier();
            }
            int consumed = take;
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint32_t c = (uint32_t)__shfl((int)w, l);
                int dup = 0, slot = -1;
                if (lane == 0) dup = n < (1ll << 31) ? (int)insert(c, slot) : 0;
                if (n >= (1ll << 31)) {                           // (no table for such populations: compare against pool[0 .. i) directly)
                    bool d = false;
                    for (int t = lane; t < i; t += 64) d |= (uint32_t)pool[t] == c;
                    dup = __any(d);
                }
                if (__builtin_amdgcn_readfirstlane(dup)) continue;      // `while j in selected: j = randbelow(n)`
                if (lane == 0) pool[i] = (int)c;
                if (++i == k) { consumed = l + 1; break; }
            }
            __builtin_amdgcn_wave_barrier();
            idx += consumed;
        }
        if (looped) {                                             // the selection, from pool[] into the lanes' register slots
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; q++) sel[q] = q * 64 + lane < k ? (long long)(uint32_t)pool[q * 64 + lane] : -1;
        }
    } else {
        for (int i = lane; i < (int)n; i += 64) pool[i] = i;
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < k; i++) {
            const uint32_t m = (uint32_t)(n - i);
            const int nbits = 32 - __builtin_clz(m);
            uint32_t r;
            do {                                        // _randbelow_with_getrandbits
                if (idx >= 624) { mt_regen(mt, lane); idx = 0; regenerated = true; }
                r = mt_temper(mt[idx++]) >> (32 - nbits);
            } while (r >= m);
            const long long res = pool[r];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) pool[r] = pool[m - 1];
            __builtin_amdgcn_wave_barrier();
            set_sel(sel, i, lane, res);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) if (q * 64 + lane < k) out[q * 64 + lane] = sel[q];
    if (regenerated) for (int i = lane; i < 624; i += 64) P.mt->mt[i] = mt[i];       // the block only changes when it is regenerated
    if (lane == 0) P.mt->idx = idx;
    if (P.gate) {
        // split schedule (fb_vec_step): this launch is the gate of the train chain behind it.  A draw that holds none of the positions the
        // env step beside it is still writing (the newest n_envs of the deque) lets the chain start at once -- once the PREVIOUS step's
        // env launch has retired, which nothing else orders this stream behind; otherwise it waits for this step's.  The wait sits here,
        // in a launch of one wave, and not in the trunk behind it: waves that spin must not hold what the kernels they wait for need.
        bool dirty = false;
#pragma unroll
        for (int q = 0; q < 4; q++) dirty |= q * 64 + lane < k && sel[q] >= P.newest_from;
        dirty = __any(dirty);
        if (lane == 0) {
            if (!dirty) atomicAdd(&P.gate->clean_count, 1u);
            fb_flag_wait(&P.gate->env_done, dirty ? P.gate_val : P.gate_val - 1, &P.gate->timeouts[0]);
            // an acting trunk of several rounds of workgroups (2048 envs and more): the chain behind this launch starts when the trunk's
            // LAST round does -- that round is a partial one (4096 envs: 52 of 820 workgroups), most of the chip is idle beside it, and the
            // full rounds before it run undisturbed
            if (P.wait_last_round) fb_flag_wait(&P.gate->last_round, P.gate_val, &P.gate->timeouts[5]);
        }
    }
}

