// fb_replay.hip -- replay memory in HBM (gfx950): frame ring, uniform + prioritized sampling, gather.
//
// Reference semantics (paths relative to the reference checkout):
//   deque store / popleft      BrainDQN.py:36,69-72 (REPLAY_MEMORY = 50000, :26)
//   frame stack                BrainDQN.py:68,238-239
//   random.sample              BrainDQN.py:197      (CPython Lib/random.py, MT19937)
//   minibatch assembly         BrainDQN.py:198-201  ("the replay gather")
//   SumTree                    BrainPrioritizedReplyDQN.py:32-104
//   Memory                     BrainPrioritizedReplyDQN.py:107-151
//
// Layout (DESIGN.md "Replay"):
//   bits[T_f][N][100] u64   one 80x80 frame = 6400 bits (the preprocess only emits 0 / 255)
//   act/rew/term[T_f][N]    per transition
//   A vector step t pushes N transitions; transition (t, e) reads frames t-3..t (s) and t-2..t+1
//   (s').  The reference's deque position j (0 = oldest) is g = oldest + j, (t, e) = divmod(g, N):
//   with N = 1 this is exactly the reference's deque, with N > 1 it is the deque that N envs
//   appending in env order would build.  T_f = ceil(cap / N) + 6 time slots keep every frame a
//   live transition needs.
//   PER: tree/maxt/mint f64[2*cap-1] array heaps.  `tree` is the reference's SumTree, updated with
//   the same sequence of floating-point operations (so its bytes are the reference's bytes);
//   maxt / mint replace the reference's O(capacity) np.max / min scans (max and min are exact and
//   order independent, so they give the identical value).
#include <math.h>
#include "fb_common.h"

namespace {

constexpr int WORDS = 100;           // u64 words per frame
constexpr int MAXB = 256;            // largest minibatch the single-workgroup sampler kernels take
constexpr int MAXH = 23;             // tree depths handled (capacity < 2^22)

struct ReplayDev {                   // mutable state, device resident (graph replay safe)
    long long steps;                 // vector pushes so far
    long long per_pointer, per_size; // SumTree.data_pointer / size
    double beta;                     // Memory.beta
    unsigned int done_ctr;           // last-block-done counter of the push kernel
    unsigned int philox_calls;
    int error;                       // sticky: bad index / batch larger than memory
};

struct ReplayParams {
    long long cap;
    int n_envs, t_f, kind;
    unsigned long long *bits;
    uint8_t *act; float *rew; uint8_t *term;
    ReplayDev *dev;
    double *tree, *maxt, *mint;
    FbMT *mt;
    int rng_kind;
    uint32_t seed_lo, seed_hi;
};

#include "fb_sampler.h"
#include "fb_gather.h"
static_assert(MAXB == FB_SAMPLE_MAXB, "sampler batch limit");

__device__ __forceinline__ size_t frame_off(const ReplayParams &P, long long f, int e) {
    if (f < 0) f = 0;                                   // setInitState: the first frame four times
    return ((size_t)(f % P.t_f) * P.n_envs + e) * WORDS;
}

// ------------------------------------------------------------------ store
__global__ __launch_bounds__(64) void reset_kernel(ReplayParams P, const uint8_t *__restrict__ frames,
                                                   const unsigned long long *__restrict__ fbits) {
    const int e = blockIdx.x, lane = threadIdx.x;
    unsigned long long *dst = P.bits + frame_off(P, 0, e);
    if (fbits) {
        for (int w = lane; w < WORDS; w += 64) dst[w] = fbits[(size_t)e * WORDS + w];
    } else {
        for (int w = 0; w < WORDS; w++) {
            const unsigned long long m = __ballot(frames[(size_t)e * 6400 + w * 64 + lane] != 0);
            if (lane == 0) dst[w] = m;
        }
    }
    if (e == 0 && lane == 0) {
        P.dev->steps = 0; P.dev->per_pointer = 0; P.dev->per_size = 0; P.dev->beta = 0.4;
        P.dev->done_ctr = 0; P.dev->error = 0;
    }
}

// `steps` (the index of the transition this push writes) is passed by value: the host handle counts pushes,
// so no workgroup has to read the device counter and the single workgroup that publishes steps + 1 at the end
// races with nobody -- no fence, no atomics.  (Consequence: a captured hipGraph must not contain pushes.)
__global__ __launch_bounds__(64) void push_kernel(ReplayParams P, long long steps, const uint8_t *__restrict__ frames,
                                                  const unsigned long long *__restrict__ fbits,
                                                  const uint8_t *__restrict__ a, const float *__restrict__ r,
                                                  const uint8_t *__restrict__ t) {
    const int lane = threadIdx.x;
    for (int e = blockIdx.x; e < P.n_envs; e += gridDim.x) {
        unsigned long long *dst = P.bits + frame_off(P, steps + 1, e);
        if (fbits) {
            for (int w = lane; w < WORDS; w += 64) dst[w] = fbits[(size_t)e * WORDS + w];
        } else {
            for (int w = 0; w < WORDS; w++) {
                const unsigned long long m = __ballot(frames[(size_t)e * 6400 + w * 64 + lane] != 0);
                if (lane == 0) dst[w] = m;
            }
        }
        if (lane == 0) {
            const size_t mo = (size_t)(steps % P.t_f) * P.n_envs + e;
            P.act[mo] = a[e]; P.rew[mo] = r[e]; P.term[mo] = t[e];
        }
    }
    if (blockIdx.x == 0 && lane == 0) P.dev->steps = steps + 1;
}

// ------------------------------------------------------------------ gather (minibatch assembly): fb_gather.h
__host__ __device__ __forceinline__ FbGatherCtx gather_ctx(const ReplayParams &P) {
    return FbGatherCtx{P.cap, P.n_envs, P.t_f, P.kind, P.bits, P.act, P.rew, P.term, &P.dev->error};
}

template <bool CURRENT>
__global__ __launch_bounds__(256) void gather_kernel(ReplayParams P, long long steps, int B, const long long *__restrict__ idx,
                                                     uint4 *__restrict__ s, uint4 *__restrict__ s2,
                                                     uint8_t *__restrict__ a, float *__restrict__ r,
                                                     uint8_t *__restrict__ t) {
    gather_body<CURRENT>(gather_ctx(P), steps, B, idx, s, s2, a, r, t, (long long)blockIdx.x * 256 + threadIdx.x);
}

// ------------------------------------------------------------------ MT19937 on one wave + random.sample: fb_sampler.h
__device__ __forceinline__ FbSampleCtx sample_ctx(const ReplayParams &P, long long steps) {
    const long long total = steps * P.n_envs;
    return FbSampleCtx{P.mt, &P.dev->error, total < P.cap ? total : P.cap};
}

__global__ __launch_bounds__(64) void sample_cpython_kernel(ReplayParams P, int k, long long setsize,
                                                            long long *__restrict__ out) {
    __shared__ uint32_t mt[624];
    __shared__ int pool[FB_SAMPLE_POOL + FB_SAMPLE_TAB];
    sample_cpython_body(sample_ctx(P, P.dev->steps), k, setsize, out, mt, pool);
}

// the draw of a step whose push is still on its way on another stream (fb_vec_step's split schedule): population size and gate from the host
__global__ __launch_bounds__(64) void sample_gated_kernel(FbSampleRider r) {
    __shared__ uint32_t mt[624];
    __shared__ int pool[FB_SAMPLE_POOL + FB_SAMPLE_TAB];
    sample_cpython_body(r.ctx, r.k, r.setsize, r.out, mt, pool);
}

// fb_replay_push_sample: Memory append + random.sample in ONE launch.  The sample only needs the size the
// memory will have after the push (known on the host), not the pushed data, so it rides as one extra
// workgroup beside the copy workgroups: the ~10 us single-wave sampler leaves the step's critical path.
__global__ __launch_bounds__(64) void push_sample_kernel(ReplayParams P, long long steps, const uint8_t *__restrict__ frames,
                                                         const unsigned long long *__restrict__ fbits,
                                                         const uint8_t *__restrict__ a, const float *__restrict__ r,
                                                         const uint8_t *__restrict__ t, int k, long long setsize,
                                                         long long *__restrict__ out) {
    __shared__ uint32_t mt[624];
    __shared__ int pool[FB_SAMPLE_POOL + FB_SAMPLE_TAB];
    if (blockIdx.x == gridDim.x - 1) { sample_cpython_body(sample_ctx(P, steps + 1), k, setsize, out, mt, pool); return; }
    const int lane = threadIdx.x;
    for (int e = blockIdx.x; e < P.n_envs; e += gridDim.x - 1) {
        unsigned long long *dst = P.bits + frame_off(P, steps + 1, e);
        if (fbits) {
            for (int w = lane; w < WORDS; w += 64) dst[w] = fbits[(size_t)e * WORDS + w];
        } else {
            for (int w = 0; w < WORDS; w++) {
                const unsigned long long m = __ballot(frames[(size_t)e * 6400 + w * 64 + lane] != 0);
                if (lane == 0) dst[w] = m;
            }
        }
        if (lane == 0) {
            const size_t mo = (size_t)(steps % P.t_f) * P.n_envs + e;
            P.act[mo] = a[e]; P.rew[mo] = r[e]; P.term[mo] = t[e];
        }
    }
    if (blockIdx.x == 0 && lane == 0) P.dev->steps = steps + 1;
}

__global__ void sample_philox_kernel(ReplayParams P, int k, long long *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = P.dev->steps * P.n_envs;
    const long long n = total < P.cap ? total : P.cap;
    const unsigned int call = P.dev->philox_calls;
    if (i < k) {
        const fb_u4 o = fb_philox(P.seed_lo, P.seed_hi, (uint32_t)i, call, FB_STREAM_SAMPLE, 0u);
        out[i] = n > 0 ? (long long)(((unsigned long long)o.x * (unsigned long long)n) >> 32) : 0;
    }
    __syncthreads();
    if (i == 0) { P.dev->philox_calls = call + 1; if (n <= 0) P.dev->error = 2; }
}

// ------------------------------------------------------------------ prioritized replay
__global__ void fill_f64_kernel(double *p, long long n, double v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

__device__ __forceinline__ int node_depth(long long node) {          // root = 0
    return 63 - __builtin_clzll((unsigned long long)(node + 1));
}
// ancestor of leaf ti (depth D) at depth d <= D
__device__ __forceinline__ long long anc(long long ti, int D, int d) { return ((ti + 1) >> (D - d)) - 1; }

// Exact SumTree.update for a list of (tree_idx, p) applied IN ORDER (BrainPrioritizedReplyDQN.py:62-68).
// The values of all touched nodes are first loaded side by side (no dependent global latency), then ONE wave walks the list in
// lockstep, lane d owning the nodes at depth d: change_j = p_j - leaf_j (the leaf's lane), every lane above adds change_j to its
// ancestor.  A node touched by several list entries takes the value its previous occurrence left (prev[][]), which is exactly the
// read-modify-write order of the reference's loop -- so the tree bytes are the reference's.
// What surrounds the walk is organised around ONE table (round 3; it was an O(n) ancestor search per (entry, depth), a depth-serial
// write-back and 21 barrier-separated passes over global memory for the max / min heaps: 27.5 us at n = 32, depth 21):
//   lcd[j][q] = depth of the lowest common ancestor of the leaves of entries j and q (the heap index + 1 IS the root-to-node path:
//   align the two to one depth, xor, count leading zeros).  From row j, in one pass over q:
//     prev[j][d]   = the largest q < j with lcd >= d   (the previous occurrence of j's depth-d ancestor: the duplicate link)
//     last bit d   = no q > j with lcd >= d            (j writes that node back: the last occurrence wins, like the reference's loop)
//     sib[j][d]    = some q with lcd == d exactly      (j's depth-(d+1) SIBLING subtree is touched too: its max / min come from q)
//   and the heaps are finished in shared memory: thread j carries (max, min) of its path node from its leaf to the root, one level per
//   step, taking the sibling's pair from the sibling entry's LDS word when that subtree was touched and from a value pre-loaded into
//   registers otherwise (all 2 x 22 loads of a thread in flight together); entries sharing a node compute and store the same pair.
#ifndef PU_EXIT
#define PU_EXIT 0                    // ablation: return behind phase k of per_apply_updates (tools/time_per.py against -DPU_EXIT=k builds)
#endif
struct UpdLayout {                   // byte offsets inside the dynamic LDS block, for a list of n entries
    int val, prev, sib, ti, p, lcd, last, rep, mask, cur, bytes;
    __host__ __device__ explicit UpdLayout(int n) {
        int o = 0;
        val = o; o += n * MAXH * 8;
        ti = o; o += n * 8;
        p = o; o += n * 8;
        cur = o; o += 4 * n * 8;
        prev = o; o += n * MAXH * 2;
        sib = o; o += n * MAXH * 2;
        last = o; o += n * 4;
        rep = o; o += n * 4;
        mask = o; o += 16;
        lcd = o; o += n * n;
        bytes = (o + 15) & ~15;
    }
};

// 4) of per_apply_updates: the max / min heaps.  Heap thread hj (active: it owns entry j = hj) climbs from its leaf (the LAST occurrence's
// p: later duplicates win) to the root.  Entries exchange values only at levels where some sibling subtree is touched as well (`readers`:
// for a random batch in a big tree that is the top few levels): a level with readers is fed by an LDS write + ONE synchronisation at the
// end of the level below it; the other levels are thread-local.  WAVE: all heap threads sit in one wave (n <= 64) -- waiting for the
// wave's own LDS traffic is the whole synchronisation, and the climb can run BESIDE the ordered walk of another wave; otherwise a
// workgroup barrier that waits for LDS traffic only (__syncthreads() would also drain the global stores of every level -- nobody in
// this workgroup reads them -- one store round trip per level).
template <bool WAVE>
__device__ __forceinline__ void per_heaps(const ReplayParams &P, int n, bool active, int j, const long long *ti_s, const double *p_s,
                                          const int *rep_s, short (*sib)[MAXH], unsigned readers, double *cur,
                                          const double (&smx)[MAXH - 1], const double (&smn)[MAXH - 1]) {
    auto level_sync = []() {
        if (WAVE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    const long long ti = ti_s[j];
    const int D = node_depth(ti);
    double mx = p_s[rep_s[j]], mn = mx;                              // the leaf's final value: p of the last entry with this leaf
    if (active) { P.maxt[ti] = mx; P.mint[ti] = mn; }
    const int Dtop = node_depth(2 * P.cap - 2);                      // the deeper leaf level: the loop below is uniform over the block
    int par = 0;
    const int dfirst = Dtop - 1;
    if (dfirst >= 0 && ((readers >> dfirst) & 1u)) {                 // (uniform)
        if (active) { cur[(par * 2 + 0) * n + j] = mx; cur[(par * 2 + 1) * n + j] = mn; }
        level_sync();
    }
#pragma unroll
    for (int d = MAXH - 2; d >= 0; d--) {
        if (d < Dtop) {                                              // (uniform)
            if (active && d < D) {
                const int sq = sib[j][d];
                const double omx = sq >= 0 ? cur[(par * 2 + 0) * n + sq] : smx[d], omn = sq >= 0 ? cur[(par * 2 + 1) * n + sq] : smn[d];
                mx = omx > mx ? omx : mx;
                mn = omn < mn ? omn : mn;
                const long long node = anc(ti, D, d);
                P.maxt[node] = mx; P.mint[node] = mn;
            }
            if (d > 0 && ((readers >> (d - 1)) & 1u)) {              // the next level has readers: publish this level's values
                par ^= 1;
                if (active) { cur[(par * 2 + 0) * n + j] = mx; cur[(par * 2 + 1) * n + j] = mn; }
                level_sync();
            }
        }
    }
}

__device__ void per_apply_updates(const ReplayParams &P, unsigned char *smem, int n, int tid, int nthreads) {
    const UpdLayout Lo(n);
    double (*val)[MAXH] = reinterpret_cast<double (*)[MAXH]>(smem + Lo.val);
    short (*prev)[MAXH] = reinterpret_cast<short (*)[MAXH]>(smem + Lo.prev);
    short (*sib)[MAXH] = reinterpret_cast<short (*)[MAXH]>(smem + Lo.sib);
    const long long *ti_s = reinterpret_cast<const long long *>(smem + Lo.ti);
    const double *p_s = reinterpret_cast<const double *>(smem + Lo.p);
    unsigned char *lcd = smem + Lo.lcd;
    unsigned *lastm = reinterpret_cast<unsigned *>(smem + Lo.last);
    int *rep_s = reinterpret_cast<int *>(smem + Lo.rep);                 // the last entry with the same leaf (its p is the leaf's final value)
    unsigned *sibmask = reinterpret_cast<unsigned *>(smem + Lo.mask);    // bit d: some entry's depth-(d + 1) sibling subtree is touched
    double *cur = reinterpret_cast<double *>(smem + Lo.cur);             // [parity][max | min][n]
    if (tid == 0) *sibmask = 0u;
    // 0) the common-ancestor depths of all pairs
    for (int it = tid; it < n * n; it += nthreads) {
        const int j = it / n, q = it - j * n;
        const long long a = ti_s[j] + 1, b = ti_s[q] + 1;               // 1-prefixed root-to-leaf paths
        const int Da = node_depth(ti_s[j]), Db = node_depth(ti_s[q]), Dm = Da < Db ? Da : Db;
        const unsigned long long z = (unsigned long long)((a >> (Da - Dm)) ^ (b >> (Db - Dm)));
        lcd[it] = (unsigned char)(z == 0 ? Dm : Dm - (64 - __builtin_clzll(z)));
    }
    __syncthreads();
    if (PU_EXIT == 1) return;
    // 1) links, one thread per entry: one pass down over q < j (prev: first hit per depth wins = the largest q), one pass over q > j
    //    (last), one over all q (sib); then the values of first occurrences, all (entry, depth) loads in flight together
    for (int j = tid; j < n; j += nthreads) {
        const int D = node_depth(ti_s[j]);
        const unsigned char *row = lcd + j * n;
        int cov = -1;                                                    // depths 0 .. cov already have their previous occurrence
        for (int d = 0; d < MAXH; d++) { prev[j][d] = -1; sib[j][d] = -1; }
        for (int q = j - 1; q >= 0 && cov < D; q--) {
            const int l = row[q];
            for (int d = cov + 1; d <= l; d++) prev[j][d] = (short)q;
            cov = l > cov ? l : cov;
        }
        int covn = -1, rep = j;                                          // depths 0 .. covn have a LATER occurrence
        for (int q = j + 1; q < n; q++) { const int l = row[q]; covn = l > covn ? l : covn; rep = l == D ? q : rep; }      // (lcd == D: the same leaf)
        lastm[j] = covn + 1 > D ? 0u : (~0u << (covn + 1));              // bit d set: entry j is the last one touching its depth-d ancestor
        rep_s[j] = rep;
        unsigned sm = 0u;
        for (int q = 0; q < n; q++) { const int l = row[q]; if (q != j && l < D) { sib[j][l] = (short)q; sm |= 1u << l; } }
        if (sm) atomicOr(sibmask, sm);
    }
    __syncthreads();
    if (PU_EXIT == 2) return;
    for (int it = tid; it < n * MAXH; it += nthreads) {
        const int j = it / MAXH, d = it - j * MAXH;
        const long long ti = ti_s[j];
        const int D = node_depth(ti);
        double v = 0;
        if (d <= D && prev[j][d] < 0) v = P.tree[anc(ti, D, d)];
        val[j][d] = v;
    }
    // (the heaps' pre-loads: thread j's off-path children, every level, requested now and used after the walk)
    // (n <= 64: the heap threads are the first n of wave 1, so that they can climb while wave 0 walks; longer lists: threads 0 .. n - 1)
    const bool hfast = n <= 64 && nthreads >= 128;
    const int hj = hfast ? tid - 64 : tid;
    const bool hact = hj >= 0 && hj < n;
    double smx[MAXH - 1], smn[MAXH - 1];
    {
        const int j = hact ? hj : 0;
        const long long ti = ti_s[j];
        const int D = node_depth(ti);
#pragma unroll
        for (int d = 0; d < MAXH - 1; d++) {
            const bool need = hact && d < D && sib[j][d] < 0;
            const long long on = d < D ? anc(ti, D, d + 1) : 1;          // j's node at depth d + 1; its sibling is the other child of the parent
            const long long off = need ? (((on + 1) ^ 1) - 1) : 0;
            smx[d] = P.maxt[off]; smn[d] = P.mint[off];
        }
    }
    __syncthreads();
    if (PU_EXIT == 3) { if (smx[3] == 1.2345 && smn[7] == 2.5) P.dev->error = 7; return; }
    // 2) the ordered walk, first wave only.  Per entry the dependent chain is: ONE LDS read (the node's current value) -> subtract / add
    //    -> one LDS write; everything else is off it -- the duplicate links and the leaf depths of 32 entries at a time are fetched into
    //    registers up front, and the leaf lane's change reaches the others through v_readlane (the leaf depth is wave-uniform), not
    //    through the LDS crossbar.  (With the links and depths read inside the loop a step took ~190 ns: 6 us for 32 entries.)
    if (tid < 64) {
        const int d = tid;
        for (int jb = 0; jb < n; jb += 32) {
            int pvr[32], Dl = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) pvr[k] = jb + k < n && d < MAXH ? (int)prev[jb + k][d] : -1;
            if (jb + (d & 31) < n) Dl = node_depth(ti_s[jb + (d & 31)]);           // lane k (and k + 32) holds entry jb + k's leaf depth
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int j = jb + k;
                if (j < n) {                                             // (uniform)
                    const int D = __builtin_amdgcn_readlane(Dl, k);
                    const int pv = pvr[k];
                    double old = 0;
                    if (d <= D) old = pv >= 0 ? val[pv][d] : val[j][d];
                    double change = p_s[j] - old;                        // meaningful on the leaf's lane (:63)
                    const int lo = __builtin_amdgcn_readlane(__double2loint(change), D), hi = __builtin_amdgcn_readlane(__double2hiint(change), D);
                    change = __hiloint2double(hi, lo);
                    if (d == D) val[j][d] = p_s[j];                      // :64
                    else if (d < D) val[j][d] = old + change;            // :66-68
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    // (n <= 64) 4) the heaps, in wave 1, BESIDE the walk: they need the links and the final leaf values, not the sums
    if (hfast && tid >= 64 && tid < 128)
        per_heaps<true>(P, n, hact, hact ? hj : 0, ti_s, p_s, rep_s, sib, *sibmask, cur, smx, smn);
    __syncthreads();
    if (PU_EXIT == 4) { if (smx[3] == 1.2345 && smn[7] == 2.5) P.dev->error = 7; return; }
    // 3) write back: the last occurrence of every touched node
    for (int it = tid; it < n * MAXH; it += nthreads) {
        const int j = it / MAXH, d = it - j * MAXH;
        const long long ti = ti_s[j];
        const int D = node_depth(ti);
        if (d <= D && ((lastm[j] >> d) & 1u)) P.tree[anc(ti, D, d)] = val[j][d];
    }
    // (longer lists) 4) the heaps, all waves
    if (!hfast) per_heaps<false>(P, n, hact, hact ? hj : 0, ti_s, p_s, rep_s, sib, *sibmask, cur, smx, smn);
}

// Memory.store for `count` new transitions (BrainPrioritizedReplyDQN.py:121-125, add :50-60), reference order, parallel over
// tree NODES.  One vector step stores `count` consecutive data slots, all with the same priority p (np.max over the leaves; storing
// the maximum never changes it).  The reference walks leaf by leaf: change_j = p - tree[leaf_j], then every ancestor += change_j.
// The order dependence of those running sums is per NODE: node a ends as ((tree[a] + c_j1) + c_j2) + .. over the leaves under it
// in increasing j -- a contiguous range of j -- and no node reads another one.  So every touched node gets a thread that adds
// its range of changes in list order: the same sequence of fp64 additions the reference executes for that node, hence the same
// bytes, with a critical path of the root's `count` additions instead of count x depth dependent walks (1024 envs, 1 M slots:
// 457 us -> ~10 us).  max / min heaps: every stored leaf holds p, the global maximum, so maxt = p on every touched node and
// mint = p on every node whose leaves are all new; the (at most two per level) partially covered edge nodes take the minimum of
// their children, walked bottom-up by one thread with the untouched siblings' values loaded up front.
// TH threads, CHUNK leaves per pass (their changes live in LDS), GRP = the group size of the ordered-add pipeline.  Two shapes:
//   <1024, 4096, 16>  the in-line launch (fb_replay_push on the caller's stream): as wide as a workgroup gets
//   < 256, 2048,  8>  the RUN-AHEAD launch (fb_replay_per_store_ahead): one wave per SIMD, <= 64 registers, 17 KB of LDS -- the shape that
//                     fits on a CU BESIDE a workgroup of the acting trunk (2 waves x 224 registers per SIMD, 133 KB of LDS), so the store
//                     starts the moment its stream is released instead of waiting for that launch to drain.  The root's chain of ordered
//                     fp64 additions -- the critical path -- is the same length either way; the parallel parts are a few per cent of it.
template <int TH, int CHUNK, int GRP>
__device__ __forceinline__ void per_store_body(const ReplayParams &P, int count, double *chg, double *sibv, long long *lvl_lo, int *lvl_off) {
    const int tid = threadIdx.x;
    long long pointer = P.dev->per_pointer, size = P.dev->per_size;
    // first data slot whose leaf sits on the deeper level (heap index >= 2^D - 1)
    const int Dmax = node_depth(2 * P.cap - 2);
    const long long deep0 = ((1ll << Dmax) - 1) - (P.cap - 1);   // may be <= 0: all leaves on one level
    int done = 0;
    while (done < count) {
        long long n = count - done;
        if (n > CHUNK) n = CHUNK;
        if (n > P.cap - pointer) n = P.cap - pointer;             // do not cross the ring wrap
        if (pointer < deep0 && pointer + n > deep0) n = deep0 - pointer;   // nor the leaf depth change: one depth per pass
        // np.max over all leaves: storing max_p never changes the maximum, re-read per pass anyway
        double max_p = P.maxt[0];
        if (max_p == 0) max_p = 1.0;                             // abs_err_upper
        __syncthreads();                                         // everybody holds the old root maximum; the previous pass is complete
        const long long leaf_lo = pointer + P.cap - 1, leaf_hi = leaf_lo + n - 1;
        const int D = node_depth(leaf_lo);
        // 1) the leaves: change_j = p - old (:63), leaf = p (:64)
        for (int j = tid; j < (int)n; j += TH) {
            const long long leaf = leaf_lo + j;
            chg[j] = max_p - P.tree[leaf];
            P.tree[leaf] = max_p; P.maxt[leaf] = max_p; P.mint[leaf] = max_p;
        }
        if (tid < D) lvl_lo[tid] = anc(leaf_lo, D, tid);
        if (tid >= TH / 2 && tid < TH / 2 + 4 * D) {             // (level d, edge e, child k): one load each, all in flight together --
            const int q = tid - TH / 2, d = q >> 2, e = (q >> 1) & 1, k = (q & 1) + 1;      // the walk in 3) then never waits for memory
            const long long a = e ? anc(leaf_hi, D, d) : anc(leaf_lo, D, d), c = 2 * a + k;
            const int sh1 = D - d - 1;
            const long long Lc = ((c + 1) << sh1) - 1, Rc = ((c + 2) << sh1) - 2;
            sibv[q] = (Rc < leaf_lo || Lc > leaf_hi) ? P.mint[c] : 0.0;
        }
        if (tid == 0) {                                          // node tasks flattened root first (the long ones land on different threads)
            int off = 0;
            for (int d = 0; d < D; d++) { lvl_off[d] = off; off += (int)(anc(leaf_hi, D, d) - anc(leaf_lo, D, d)) + 1; }
            lvl_off[D] = off;
        }
        __syncthreads();
        // 2) every touched internal node adds the changes of the leaves under it, in list order (:66-68)
        const int total = lvl_off[D];
        for (int task = tid; task < total; task += TH) {
            int d = 0;
            while (d + 1 < D && lvl_off[d + 1] <= task) d++;
            const long long a = lvl_lo[d] + (task - lvl_off[d]);
            const int sh = D - d;
            const long long L = ((a + 1) << sh) - 1, R = ((a + 2) << sh) - 2;      // its descendants on the leaves' level
            const int jlo = (int)((L > leaf_lo ? L : leaf_lo) - leaf_lo), jhi = (int)((R < leaf_hi ? R : leaf_hi) - leaf_lo);
            double v = P.tree[a];
            int j = jlo;
            // groups of GRP: the LDS reads of the NEXT group are in flight while this group's dependent fp64 adds run (the adds stay
            // in list order; the root's chain is the launch's critical path: one LDS latency per 8 adds made it ~15 ns per leaf)
            if (j + GRP <= jhi + 1) {
                // (two register sets used alternately -- no copies: v_mov_b64 costs as much as the v_add_f64 it would sit beside)
                double c[GRP], nx[GRP];
#pragma unroll
                for (int q = 0; q < GRP; q++) c[q] = chg[j + q];
                while (j + GRP <= jhi + 1) {
                    const bool m1 = j + 2 * GRP <= jhi + 1;
#pragma unroll
                    for (int q = 0; q < GRP; q++) nx[q] = chg[m1 ? j + GRP + q : j + q];
#pragma unroll
                    for (int q = 0; q < GRP; q++) v += c[q];
                    j += GRP;
                    if (!m1) break;
                    const bool m2 = j + 2 * GRP <= jhi + 1;
#pragma unroll
                    for (int q = 0; q < GRP; q++) c[q] = chg[m2 ? j + GRP + q : j + q];
#pragma unroll
                    for (int q = 0; q < GRP; q++) v += nx[q];
                    j += GRP;
                    if (!m2) break;
                }
            }
            for (; j <= jhi; j++) v += chg[j];
            P.tree[a] = v;
            P.maxt[a] = max_p;
            if (L >= leaf_lo && R <= leaf_hi) P.mint[a] = max_p;     // all its leaves are new
        }
        // 3) the minimum heap on the partially covered edge nodes (<= 2 per level), bottom-up, by one thread beside 2)
        if (tid == TH - 1) {
            double m_lo = max_p, m_hi = max_p;                   // values of the level below's edge nodes (the leaves: p)
            for (int d = D - 1; d >= 0; d--) {
                const long long lo = anc(leaf_lo, D, d), hi = anc(leaf_hi, D, d), clo = anc(leaf_lo, D, d + 1), chi = anc(leaf_hi, D, d + 1);
                const int sh1 = D - d - 1;
                double nv[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const long long a = e ? hi : lo;
                    double val = INFINITY;
#pragma unroll
                    for (int k = 1; k <= 2; k++) {
                        const long long c = 2 * a + k;
                        const long long Lc = ((c + 1) << sh1) - 1, Rc = ((c + 2) << sh1) - 2;
                        double cv;
                        if (Rc < leaf_lo || Lc > leaf_hi) cv = sibv[(d * 2 + e) * 2 + (k - 1)];  // untouched: the stored value stands (fetched above)
                        else if (Lc >= leaf_lo && Rc <= leaf_hi) cv = max_p;                 // all new
                        else cv = c == clo ? m_lo : m_hi;                                     // the level below's edge node
                        (void)chi;
                        val = cv < val ? cv : val;
                    }
                    nv[e] = val;
                    P.mint[a] = val;
                }
                m_lo = nv[0]; m_hi = nv[1];
            }
        }
        __threadfence_block();
        pointer = (pointer + n) % P.cap;
        size = size + n < P.cap ? size + n : P.cap;
        done += (int)n;
    }
    __syncthreads();
    if (tid == 0) { P.dev->per_pointer = pointer; P.dev->per_size = size; }
}

__global__ __launch_bounds__(1024) void per_store_kernel(ReplayParams P, int count) {
    __shared__ double chg[4096];
    __shared__ double sibv[4 * MAXH];       // min-heap values of the untouched children of the edge nodes, fetched up front
    __shared__ long long lvl_lo[MAXH];
    __shared__ int lvl_off[MAXH + 1];
    per_store_body<1024, 4096, 16>(P, count, chg, sibv, lvl_lo, lvl_off);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void per_store_slim_kernel(ReplayParams P, int count) {
    __shared__ double chg[2048];
    __shared__ double sibv[4 * MAXH];
    __shared__ long long lvl_lo[MAXH];
    __shared__ int lvl_off[MAXH + 1];
    per_store_body<256, 2048, 8>(P, count, chg, sibv, lvl_lo, lvl_off);
}

// ---- FB_PER_FAST: the same heaps kept by RECOMPUTATION instead of the reference's running sums.  A node is
// set to (left + right) of its children, level by level above the touched leaves, so the tree is the exactly
// rounded pairwise sum of its leaves and independent of the order of the updates -- the low bits differ from
// the reference's history-dependent tree (which is why FB_PER_EXACT stays the default and the parity mode),
// but a store of N leaves costs ~21 level passes of one workgroup instead of N ordered tree walks
// (1024 envs: 457 us -> tens of us).
__device__ __forceinline__ void per_refresh_node(const ReplayParams &P, long long node) {
    const double a = P.tree[2 * node + 1], b = P.tree[2 * node + 2];
    const double c = P.maxt[2 * node + 1], d = P.maxt[2 * node + 2];
    const double e = P.mint[2 * node + 1], f = P.mint[2 * node + 2];
    P.tree[node] = a + b;
    P.maxt[node] = c > d ? c : d;
    P.mint[node] = e < f ? e : f;
}

// Memory.store for `count` consecutive data slots: the touched leaves are one or two contiguous heap ranges
// (ring wrap), and so are their ancestors on every level.
__global__ __launch_bounds__(1024) void per_store_fast_kernel(ReplayParams P, int count) {
    const int tid = threadIdx.x;
    long long pointer = P.dev->per_pointer, size = P.dev->per_size;
    double max_p = P.maxt[0];
    if (max_p == 0) max_p = 1.0;                                 // abs_err_upper
    __syncthreads();                                             // everybody has read the old root
    // a segment never crosses the ring wrap nor the slot where the leaves change depth (deep0), so every range below
    // holds nodes of ONE depth and a pass never reads a node that the same pass writes
    const int Dmax = node_depth(2 * P.cap - 2);
    const long long deep0 = ((1ll << Dmax) - 1) - (P.cap - 1);
    long long left = count < P.cap ? count : P.cap, lo = pointer;
    while (left > 0) {
        long long n = left < P.cap - lo ? left : P.cap - lo;
        if (lo < deep0 && lo + n > deep0) n = deep0 - lo;
        long long a = lo + P.cap - 1, b = a + n - 1;             // heap range of the leaves
        for (long long i = a + tid; i <= b; i += 1024) { P.tree[i] = max_p; P.maxt[i] = max_p; P.mint[i] = max_p; }
        while (a > 0) {
            __threadfence_block();
            __syncthreads();
            a = (a - 1) >> 1; b = (b - 1) >> 1;
            for (long long i = a + tid; i <= b; i += 1024) per_refresh_node(P, i);
        }
        __threadfence_block();
        __syncthreads();
        left -= n; lo = (lo + n) % P.cap;
    }
    if (tid == 0) {
        P.dev->per_pointer = (pointer + count) % P.cap;
        P.dev->per_size = size + count < P.cap ? size + count : P.cap;
    }
}

// Memory.batch_update: later duplicates win (like the reference's loop), then the ancestors level by level
// (duplicate ancestors are refreshed redundantly with identical values)
__global__ __launch_bounds__(256) void per_update_fast_kernel(ReplayParams P, int n, const long long *__restrict__ idx,
                                                              float *__restrict__ abs_err, const float *__restrict__ prio, int write_back) {
    __shared__ long long ti_s[MAXB];
    const int tid = threadIdx.x;
    double ps = 0;
    if (tid < n) {
        long long ti = idx[tid];
        if (ti < P.cap - 1 || ti > 2 * P.cap - 2) { P.dev->error = 1; ti = P.cap - 1; }
        ti_s[tid] = ti;
        float pf;
        if (prio) pf = prio[tid];
        else {
            float e = abs_err[tid] + 0.01f;
            if (write_back) abs_err[tid] = e;
            const float c = e < 1.0f ? e : 1.0f;
            pf = (float)pow((double)c, (double)0.6f);
        }
        ps = (double)pf;
    }
    __syncthreads();
    if (tid < n) {
        bool last = true;
        for (int q = tid + 1; q < n; q++) if (ti_s[q] == ti_s[tid]) { last = false; break; }
        if (last) { P.tree[ti_s[tid]] = ps; P.maxt[ti_s[tid]] = ps; P.mint[ti_s[tid]] = ps; }
    }
    for (int d = MAXH - 2; d >= 0; d--) {
        __threadfence_block();
        __syncthreads();
        if (tid < n) {
            const int D = node_depth(ti_s[tid]);
            if (d < D) per_refresh_node(P, anc(ti_s[tid], D, d));
        }
    }
}

// Memory.batch_update (BrainPrioritizedReplyDQN.py:146-151)
// write_back = 0: the caller's abs_err array is left alone (the run-ahead form: the kernel runs on the memory's side stream while the
// caller's stream may be reading that array)
__global__ __launch_bounds__(256) void per_update_kernel(ReplayParams P, int n, const long long *__restrict__ idx,
                                                         float *__restrict__ abs_err,
                                                         const float *__restrict__ prio, int write_back) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const UpdLayout Lo(n);
    long long *ti_w = reinterpret_cast<long long *>(smem + Lo.ti);
    double *p_w = reinterpret_cast<double *>(smem + Lo.p);
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += 256) {
        long long ti = idx[j];
        if (ti < P.cap - 1 || ti > 2 * P.cap - 2) { P.dev->error = 1; ti = P.cap - 1; }
        ti_w[j] = ti;
        float ps;
        if (prio) ps = prio[j];
        else {
            float e = abs_err[j] + 0.01f;                        // abs_errors += epsilon (in place, fp32)
            if (write_back) abs_err[j] = e;
            const float c = e < 1.0f ? e : 1.0f;                 // np.minimum(.., abs_err_upper)
            ps = (float)pow((double)c, (double)0.6f);            // np.power(fp32, 0.6) -> fp32
        }
        p_w[j] = (double)ps;
    }
    __syncthreads();
    per_apply_updates(P, smem, n, tid, 256);
}

// Memory.sample (BrainPrioritizedReplyDQN.py:127-144)
__global__ __launch_bounds__(256) void per_sample_kernel(ReplayParams P, int n, const double *__restrict__ uni,
                                                         long long *__restrict__ idx_out,
                                                         double *__restrict__ isw_out, float *__restrict__ isw32_out) {
    __shared__ uint32_t mt[624];
    __shared__ uint32_t words[2 * MAXB];
    const int tid = threadIdx.x;
    const double old_beta = P.dev->beta;
    const double nb = old_beta + 0.001;
    const double beta = nb < 1.0 ? nb : 1.0;                     // np.min([1., beta + 0.001])
    unsigned int call = P.dev->philox_calls;
    if (!uni && P.rng_kind == FB_RNG_NUMPY) {                    // 2 words per np.random.uniform, in order
        if (tid < 64) {
            for (int i = tid; i < 624; i += 64) mt[i] = P.mt->mt[i];
            uint32_t idx = P.mt->idx;
            __builtin_amdgcn_wave_barrier();
            int produced = 0;
            while (produced < 2 * n) {
                if (idx >= 624) { mt_regen(mt, tid); idx = 0; }
                int avail = 624 - (int)idx, want = 2 * n - produced;
                int take = avail < want ? avail : want;
                for (int i = tid; i < take; i += 64) words[produced + i] = mt_temper(mt[idx + i]);
                produced += take; idx += take;
                __builtin_amdgcn_wave_barrier();
            }
            for (int i = tid; i < 624; i += 64) P.mt->mt[i] = mt[i];
            if (tid == 0) P.mt->idx = idx;
        }
    }
    __syncthreads();
    if (tid < n) {
        const double total = P.tree[0];
        const double seg = total / n;                            // pri_seg
        const double a = seg * tid, b = seg * (tid + 1);
        double u;
        if (uni) u = uni[tid];
        else if (P.rng_kind == FB_RNG_NUMPY) {
            const uint32_t x = words[2 * tid] >> 5, y = words[2 * tid + 1] >> 6;
            u = (x * 67108864.0 + y) / 9007199254740992.0;
        } else {
            const fb_u4 o = fb_philox(P.seed_lo, P.seed_hi, (uint32_t)tid, call, FB_STREAM_PER, 0u);
            u = ((o.x >> 5) * 67108864.0 + (o.y >> 6)) / 9007199254740992.0;
        }
        double v = a + (b - a) * u;                              // np.random.uniform(a, b)
        const long long len = 2 * P.cap - 1;
        long long parent = 0;
        for (;;) {                                               // get_leaf, :85-100
            const long long cl = 2 * parent + 1;
            if (cl >= len) break;
            const double left = P.tree[cl];
            if (v <= left) parent = cl;
            else { v -= left; parent = cl + 1; }
        }
        const double prob = P.tree[parent] / total;
        const double min_prob = P.mint[0] / total;               // get_min_prob over the filled leaves
        idx_out[tid] = parent;
        const double wgt = pow(prob / min_prob, -beta);
        isw_out[tid] = wgt;
        if (isw32_out) isw32_out[tid] = (float)wgt;               // (the float32 ISWeights placeholder of the loss, :245)
    }
    if (tid == 0) { P.dev->beta = beta; P.dev->philox_calls = call + 1; }
}

}  // namespace

struct fb_replay {
    ReplayParams P;
    FbMT *h_mt;
    long long host_steps;            // pushes since the last reset (mirror of ReplayDev::steps)
    int per_mode;                    // FB_PER_EXACT | FB_PER_FAST
    // prioritized memories: Memory.store's TREE part of the coming push can run ahead of it on a stream of its own
    // (fb_replay_per_store_ahead): it needs the tree and the number of envs, nothing of the frames
    hipStream_t side;
    hipEvent_t ev_fork, ev_store, ev_upd;
    bool store_ahead;                // the tree part of the next fb_replay_push has been issued already; that push joins it
    bool upd_pending;                // a batch_update runs on the side stream (fb_replay_update_priorities_ahead): whatever touches the tree next joins it
    bool store_forked;               // the run-ahead store was ordered behind the caller's stream by an event of its own (a sample behind it needs no second one)
    const void *side_for; bool side_checked; int side_ok; int side_prio;      // the caller's stream `side` has been checked against (fb_side_stream_beside; NULL is a stream too), the verdict, its priority
};

// The run-ahead tree work only pays when `side` really runs BESIDE the caller's stream; HIP may have put both on one hardware queue
// (fb_common.h).  Checked once per caller's stream, with nothing of this memory in flight on `side`; a side stream that fails is
// replaced by a fresh one (up to six tries), and if none passes the tree work stays in line.
static bool per_side_usable(fb_replay *h, void *stream) {
    if (!h->side) return false;
    if (h->side_checked && h->side_for == stream) return h->side_ok != 0;
    if (h->store_ahead || h->upd_pending) return h->side_checked && h->side_ok != 0;       // (work of this memory is on `side`: not now)
    h->side = fb_side_stream_beside(fb_stream(stream), h->side_prio, h->side, &h->side_ok);
    h->side_for = stream; h->side_checked = true;
    return h->side_ok != 0;
}


// every entry point that reads or writes the tree on a caller's stream passes here first
static int per_join(fb_replay *h, hipStream_t st) {
    if (h->upd_pending) {
        FB_CHECK_HIP(hipStreamWaitEvent(st, h->ev_upd, 0));
        h->upd_pending = false;
    }
    return FB_OK;
}

extern "C" int fb_replay_create(int64_t capacity, int n_envs, int kind, fb_replay_t *out) {
    FB_REQUIRE(out, "fb_replay_create: out is NULL");
    FB_REQUIRE(capacity >= 1 && capacity < (1ll << 22), "fb_replay_create: capacity %lld out of range", (long long)capacity);
    FB_REQUIRE(n_envs >= 1 && n_envs <= (1 << 22), "fb_replay_create: n_envs out of range");
    FB_REQUIRE(kind == FB_REPLAY_UNIFORM || kind == FB_REPLAY_PER, "fb_replay_create: kind must be 0 or 1");
    fb_replay *h = new fb_replay();
    memset(h, 0, sizeof(*h));
    ReplayParams &P = h->P;
    P.cap = capacity; P.n_envs = n_envs; P.kind = kind;
    P.t_f = (int)((capacity + n_envs - 1) / n_envs) + 6;
    const size_t slots = (size_t)P.t_f * n_envs;
    // (the ring and its rows start as zeros: slots that have never been pushed to hold nothing of an earlier allocation, and equal
    // memories save equal checkpoint blobs)
    hipError_t e = hipMalloc(&P.bits, slots * WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(P.bits, 0, slots * WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&P.act, slots);
    if (e == hipSuccess) e = hipMemset(P.act, 0, slots);
    if (e == hipSuccess) e = hipMalloc(&P.rew, slots * sizeof(float));
    if (e == hipSuccess) e = hipMemset(P.rew, 0, slots * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&P.term, slots);
    if (e == hipSuccess) e = hipMemset(P.term, 0, slots);
    if (e == hipSuccess) e = hipMalloc(&P.dev, sizeof(ReplayDev));
    if (e == hipSuccess) e = hipMemset(P.dev, 0, sizeof(ReplayDev));
    if (e == hipSuccess) e = hipMalloc(&P.mt, sizeof(FbMT));
    if (e == hipSuccess && kind == FB_REPLAY_PER) {
        const size_t nb = sizeof(double) * (size_t)(2 * capacity - 1);
        e = hipMalloc(&P.tree, nb);
        if (e == hipSuccess) e = hipMalloc(&P.maxt, nb);
        if (e == hipSuccess) e = hipMalloc(&P.mint, nb);
    }
    if (e != hipSuccess) {
        fb_set_error(e == hipErrorOutOfMemory ? FB_ERR_NOMEM : FB_ERR_HIP, "fb_replay_create: %s", hipGetErrorString(e));
        fb_replay_destroy(h);
        return e == hipErrorOutOfMemory ? FB_ERR_NOMEM : FB_ERR_HIP;
    }
    if (kind == FB_REPLAY_PER) {
        // (highest priority: the store is ONE workgroup; when it becomes ready together with the acting forward's thousand it must get its
        // CU first -- behind them it would wait until that launch drains, and most of what it could hide would be over)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        h->side_prio = prio_hi;
        if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, h->side_prio) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_store, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_upd, hipEventDisableTiming) != hipSuccess) {
            fb_set_error(FB_ERR_HIP, "fb_replay_create: stream / event creation failed");
            fb_replay_destroy(h);
            return FB_ERR_HIP;
        }
    }
    *out = h;
    int rc = fb_replay_seed(h, kind == FB_REPLAY_PER ? FB_RNG_NUMPY : FB_RNG_CPYTHON, 0);
    if (rc != FB_OK) return rc;
    return fb_replay_reset(h, nullptr, nullptr, nullptr);
}

extern "C" int fb_replay_destroy(fb_replay_t h) {
    if (!h) return FB_OK;
    ReplayParams &P = h->P;
    void *ptrs[] = {P.bits, P.act, P.rew, P.term, P.dev, P.mt, P.tree, P.maxt, P.mint};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_store) (void)hipEventDestroy(h->ev_store);
    if (h->ev_upd) (void)hipEventDestroy(h->ev_upd);
    delete h;
    return FB_OK;
}

extern "C" int fb_replay_seed(fb_replay_t h, int rng_kind, uint64_t seed) {
    FB_REQUIRE(h, "fb_replay_seed: NULL handle");
    FB_REQUIRE(rng_kind >= 0 && rng_kind <= 2, "fb_replay_seed: rng_kind must be 0 (cpython), 1 (philox) or 2 (numpy)");
    FbMT s;
    if (rng_kind == FB_RNG_NUMPY) fb_mt_init_genrand_host(&s, (uint32_t)seed);       // np.random.seed(int)
    else {                                                                           // random.seed(int)
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        fb_mt_init_by_array_host(&s, key, key[1] ? 2 : 1);
    }
    FB_CHECK_HIP(hipDeviceSynchronize());
    FB_CHECK_HIP(hipMemcpy(h->P.mt, &s, sizeof(FbMT), hipMemcpyHostToDevice));
    h->P.rng_kind = rng_kind;
    h->P.seed_lo = (uint32_t)seed; h->P.seed_hi = (uint32_t)(seed >> 32);
    return FB_OK;
}

extern "C" int fb_replay_reset(fb_replay_t h, const uint8_t *frames, const uint64_t *frame_bits, void *stream) {
    FB_REQUIRE(h, "fb_replay_reset: NULL handle");
    h->host_steps = 0;
    ReplayParams &P = h->P;
    hipStream_t st = fb_stream(stream);
    if (P.kind == FB_REPLAY_PER) {
        if (h->side) {                   // (nothing of an earlier life may still be on its way to the tree)
            FB_CHECK_HIP(hipStreamSynchronize(h->side));
            h->store_ahead = false; h->upd_pending = false;
        }
        const long long nn = 2 * P.cap - 1;
        FB_CHECK_HIP(hipMemsetAsync(P.tree, 0, sizeof(double) * nn, st));
        FB_CHECK_HIP(hipMemsetAsync(P.maxt, 0, sizeof(double) * nn, st));
        hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, P.mint, nn, (double)INFINITY);
        FB_LAUNCH_CHECK();
    }
    if (!frames && !frame_bits) {       // create-time reset: empty memory, blank first frame
        FB_CHECK_HIP(hipMemsetAsync(P.bits, 0, (size_t)P.n_envs * WORDS * sizeof(unsigned long long), st));
        FB_CHECK_HIP(hipMemsetAsync(P.dev, 0, sizeof(ReplayDev), st));
        ReplayDev d; memset(&d, 0, sizeof(d)); d.beta = 0.4;
        FB_CHECK_HIP(hipStreamSynchronize(st));
        FB_CHECK_HIP(hipMemcpy(P.dev, &d, sizeof(d), hipMemcpyHostToDevice));
        return FB_OK;
    }
    hipLaunchKernelGGL(reset_kernel, dim3(P.n_envs), dim3(64), 0, st, P, frames, (const unsigned long long *)frame_bits);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_replay_push(fb_replay_t h, const uint8_t *frames, const uint64_t *frame_bits, const uint8_t *actions,
                              const float *rewards, const uint8_t *terminals, void *stream) {
    FB_REQUIRE(h && actions && rewards && terminals, "fb_replay_push: NULL argument");
    FB_REQUIRE((frames != nullptr) != (frame_bits != nullptr), "fb_replay_push: give exactly one of frames / frame_bits");
    ReplayParams &P = h->P;
    hipStream_t st = fb_stream(stream);
    hipLaunchKernelGGL(push_kernel, dim3(P.n_envs < 512 ? P.n_envs : 512), dim3(64), 0, st, P, h->host_steps, frames, (const unsigned long long *)frame_bits,
                       actions, rewards, terminals);
    FB_LAUNCH_CHECK();
    h->host_steps += 1;
    return fb_replay_finish_push(h, stream);
}

// Memory.store's tree part of a push whose frames and scalars are on their way (push_kernel above, or the env launch that carried them:
// fb_replay_begin_push_rider).  Uniform memory: nothing to do.
int fb_replay_finish_push(fb_replay_t h, void *stream) {
    ReplayParams &P = h->P;
    hipStream_t st = fb_stream(stream);
    if (P.kind == FB_REPLAY_PER) {
        if (h->store_ahead) {                    // the tree part ran ahead on the side stream: whatever follows this push waits for it
            h->store_ahead = false;
            h->upd_pending = false;              // (a batch_update in front of it on that stream is covered by the same wait)
            FB_CHECK_HIP(hipStreamWaitEvent(st, h->ev_store, 0));
            return FB_OK;
        }
        const int rcj = per_join(h, st);
        if (rcj != FB_OK) return rcj;
        if (h->per_mode == FB_PER_FAST) hipLaunchKernelGGL(per_store_fast_kernel, dim3(1), dim3(1024), 0, st, P, P.n_envs);
        else hipLaunchKernelGGL(per_store_kernel, dim3(1), dim3(1024), 0, st, P, P.n_envs);
        FB_LAUNCH_CHECK();
    }
    return FB_OK;
}

// Memory.store's tree work for the NEXT fb_replay_push, issued now on the memory's side stream behind everything `stream` holds so far
// (BrainPrioritizedReplyDQN.py:121-125: p = max over the leaves, add(p, .) for each new transition -- nothing of the transition itself
// enters the tree).  One vector step's store is one workgroup walking the root's chain of n_envs ordered fp64 additions (17 us at
// 1024 envs, 64 us at 4096): issued here, at the top of fb_vec_step, it runs beside the acting forward and the env step instead of
// between the env step and Memory.sample.  The push that follows joins it.  Returns 1 when issued (prioritized memories), else 0.
int fb_replay_per_store_ahead(fb_replay_t h, void *stream) {
    // (reference-order mode only: the level-wise FB_PER_FAST store takes ~8 us, less than the two cross-stream hops cost)
    if (!h || h->P.kind != FB_REPLAY_PER || h->store_ahead || h->per_mode == FB_PER_FAST) return 0;
    if (!per_side_usable(h, stream)) return 0;
    hipStream_t st = fb_stream(stream);
    // (behind a run-ahead batch_update the side stream is already ordered after the caller's last touch of the tree -- that update's own
    // fork -- and nothing on the caller's stream has touched the tree since: no second fork)
    h->store_forked = !h->upd_pending;
    if (!h->upd_pending && (hipEventRecord(h->ev_fork, st) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_fork, 0) != hipSuccess)) return 0;
    if (h->per_mode == FB_PER_FAST) hipLaunchKernelGGL(per_store_fast_kernel, dim3(1), dim3(1024), 0, h->side, h->P, h->P.n_envs);
    else hipLaunchKernelGGL(per_store_slim_kernel, dim3(1), dim3(256), 0, h->side, h->P, h->P.n_envs);     // (the shape that fits beside the acting trunk)
    if (hipEventRecord(h->ev_store, h->side) != hipSuccess) { (void)hipStreamWaitEvent(st, h->ev_fork, 0); return 0; }
    h->store_ahead = true;
    return 1;
}

// Memory.sample of the step whose Memory.store has just been put on the side stream (fb_replay_per_store_ahead returned 1): the draw needs
// the tree as that store leaves it and the memory's own generator -- nothing of the env step -- so it follows the store there and the
// push's join covers it (ev_store is recorded again behind it).  Returns 1 when issued.
int fb_replay_sample_ahead(fb_replay_t h, int batch, int64_t *idx, double *isw, float *isw32, void *stream) {
    static const bool on = !(getenv("FB_PER_SAMPLE_AHEAD") && atoi(getenv("FB_PER_SAMPLE_AHEAD")) == 0);      // A/B knob
    if (!on || !h || h->P.kind != FB_REPLAY_PER || !h->store_ahead || !idx || !isw || batch < 1 || batch > MAXB) return 0;
    // the draw WRITES the caller's idx / isw buffers: it goes behind whatever the caller's stream holds so far (readers of the previous
    // step's indices) -- the store in front of it may have been issued without a fork of its own (behind a run-ahead batch_update)
    // (fb_vec_step issues it right behind the store: that store's own fork, if it made one, covers both)
    if (!h->store_forked && (hipEventRecord(h->ev_fork, fb_stream(stream)) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_fork, 0) != hipSuccess)) return 0;
    hipLaunchKernelGGL(per_sample_kernel, dim3(1), dim3(256), 0, h->side, h->P, batch, (const double *)nullptr, (long long *)idx, isw, isw32);
    if (hipGetLastError() != hipSuccess || hipEventRecord(h->ev_store, h->side) != hipSuccess) {
        (void)hipStreamSynchronize(h->side);                 // (the store's own record stands; whatever was issued is done before anyone goes on)
        return 0;
    }
    return 1;
}

static long long cpython_setsize(int batch) {
    // Lib/random.py: setsize = 21; if k > 5: setsize += 4 ** _ceil(_log(k * 3, 4))
    long long setsize = 21;
    if (batch > 5) setsize += (long long)pow(4.0, ceil(log((double)batch * 3.0) / log(4.0)));
    return setsize;
}

extern "C" int fb_replay_push_sample(fb_replay_t h, const uint8_t *frames, const uint64_t *frame_bits, const uint8_t *actions,
                                     const float *rewards, const uint8_t *terminals, int batch, int64_t *idx, void *stream) {
    FB_REQUIRE(h && actions && rewards && terminals && idx, "fb_replay_push_sample: NULL argument");
    FB_REQUIRE((frames != nullptr) != (frame_bits != nullptr), "fb_replay_push_sample: give exactly one of frames / frame_bits");
    FB_REQUIRE(batch >= 1 && batch <= MAXB, "fb_replay_push_sample: batch must be in 1..%d", MAXB);
    ReplayParams &P = h->P;
    if (P.kind != FB_REPLAY_UNIFORM || P.rng_kind != FB_RNG_CPYTHON) {       // nothing to fuse: the two calls in a row
        int rc = fb_replay_push(h, frames, frame_bits, actions, rewards, terminals, stream);
        if (rc != FB_OK) return rc;
        return fb_replay_sample(h, batch, nullptr, idx, nullptr, stream);
    }
    hipLaunchKernelGGL(push_sample_kernel, dim3((P.n_envs < 512 ? P.n_envs : 512) + 1), dim3(64), 0, fb_stream(stream), P, h->host_steps,
                       frames, (const unsigned long long *)frame_bits, actions, rewards, terminals, batch, cpython_setsize(batch),
                       (long long *)idx);
    FB_LAUNCH_CHECK();
    h->host_steps += 1;
    return FB_OK;
}

int fb_replay_num_envs(fb_replay_t h) { return h ? h->P.n_envs : 0; }

int fb_replay_is_prioritized(fb_replay_t h) { return h && h->P.kind == FB_REPLAY_PER; }

int fb_replay_begin_push_rider(fb_replay_t h, FbPushRider *push) {
    ReplayParams &P = h->P;
    const long long steps = h->host_steps;               // push_kernel: frame steps + 1, meta row steps
    push->bits = P.bits + (size_t)((steps + 1) % P.t_f) * P.n_envs * WORDS;
    const size_t mo = (size_t)(steps % P.t_f) * P.n_envs;
    push->act = P.act + mo; push->rew = P.rew + mo; push->term = P.term + mo;
    push->steps_dev = &P.dev->steps; push->steps_new = steps + 1;
    h->host_steps += 1;
    return 1;
}

int fb_replay_gather_rider(fb_replay_t h, int batch, const int64_t *idx, uint8_t *s, uint8_t *s2, uint8_t *a, float *r, uint8_t *t,
                           FbGatherRider *rider) {
    if (!h || !idx || !s || !s2 || !a || !r || !t || batch < 1) return 0;
    rider->c = gather_ctx(h->P); rider->steps = h->host_steps; rider->B = batch; rider->idx = (const long long *)idx;
    rider->s = s; rider->s2 = s2; rider->a = a; rider->r = r; rider->t = t;
    return 1;
}

int fb_replay_sample_rider(fb_replay_t h, int batch, int64_t *idx, FbSampleRider *rider, int pushes_ahead) {
    const ReplayParams &P = h->P;
    if (P.kind != FB_REPLAY_UNIFORM || P.rng_kind != FB_RNG_CPYTHON || batch < 1 || batch > MAXB || !idx) return 0;
    const long long total = (h->host_steps + pushes_ahead) * P.n_envs;
    rider->ctx = FbSampleCtx{P.mt, &P.dev->error, total < P.cap ? total : P.cap};
    rider->k = batch; rider->setsize = cpython_setsize(batch); rider->out = (long long *)idx;
    return 1;
}

int fb_replay_sample_gated(fb_replay_t h, int batch, int64_t *idx, const FbSplitCtx *ctx, void *stream, int wait_last_round) {
    FbSampleRider r;
    if (!h || !ctx || !fb_replay_sample_rider(h, batch, idx, &r, 1)) return 0;
    r.ctx.gate = ctx->f; r.ctx.gate_val = ctx->seq; r.ctx.wait_last_round = wait_last_round;
    r.ctx.newest_from = r.ctx.n - h->P.n_envs;          // deque positions of the transitions the coming push appends
    hipLaunchKernelGGL(sample_gated_kernel, dim3(1), dim3(64), 0, fb_stream(stream), r);
    return hipGetLastError() == hipSuccess;
}

extern "C" int fb_replay_current_state(fb_replay_t h, uint8_t *states, void *stream) {
    FB_REQUIRE(h && states, "fb_replay_current_state: NULL argument");
    ReplayParams &P = h->P;
    const long long threads = (long long)P.n_envs * 1600;
    hipLaunchKernelGGL(gather_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, fb_stream(stream), P,
                       h->host_steps, P.n_envs, (const long long *)nullptr, (uint4 *)states, (uint4 *)nullptr, (uint8_t *)nullptr,
                       (float *)nullptr, (uint8_t *)nullptr);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_replay_sample(fb_replay_t h, int batch, const double *uniforms, int64_t *idx, double *isw, void *stream) {
    return fb_replay_sample_f32(h, batch, uniforms, idx, isw, nullptr, stream);
}

int fb_replay_sample_f32(fb_replay_t h, int batch, const double *uniforms, int64_t *idx, double *isw, float *isw32, void *stream) {
    FB_REQUIRE(h && idx, "fb_replay_sample: NULL argument");
    FB_REQUIRE(batch >= 1 && batch <= MAXB, "fb_replay_sample: batch must be in 1..%d", MAXB);
    ReplayParams &P = h->P;
    hipStream_t st = fb_stream(stream);
    if (P.kind == FB_REPLAY_PER) {
        FB_REQUIRE(isw, "fb_replay_sample: PER needs isw");
        const int rcj = per_join(h, st);
        if (rcj != FB_OK) return rcj;
        hipLaunchKernelGGL(per_sample_kernel, dim3(1), dim3(256), 0, st, P, batch, uniforms, (long long *)idx, isw, isw32);
    } else if (P.rng_kind == FB_RNG_PHILOX) {
        hipLaunchKernelGGL(sample_philox_kernel, dim3(1), dim3(256), 0, st, P, batch, (long long *)idx);
    } else {
        hipLaunchKernelGGL(sample_cpython_kernel, dim3(1), dim3(64), 0, st, P, batch, cpython_setsize(batch), (long long *)idx);
    }
    FB_LAUNCH_CHECK();
    return FB_OK;
}

int fb_replay_ring_src(fb_replay_t h, int batch, const int64_t *idx, uint8_t *a, float *r, uint8_t *t, FbRingSrc *out) {
    FB_REQUIRE(h && idx && a && r && t && out, "fb_replay_ring_src: NULL argument");
    FB_REQUIRE(batch >= 1 && batch <= MAXB, "fb_replay_ring_src: batch out of range");
    *out = FbRingSrc{gather_ctx(h->P), h->host_steps, (const long long *)idx, a, r, t};
    return FB_OK;
}

extern "C" int fb_replay_gather(fb_replay_t h, int batch, const int64_t *idx, uint8_t *s, uint8_t *s2, uint8_t *a, float *r,
                                uint8_t *t, void *stream) {
    FB_REQUIRE(h && idx && s && s2 && a && r && t, "fb_replay_gather: NULL argument");
    FB_REQUIRE(batch >= 1 && batch <= (1 << 20), "fb_replay_gather: batch out of range");
    const long long threads = (long long)batch * 1600;
    hipLaunchKernelGGL(gather_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, fb_stream(stream), h->P,
                       h->host_steps, batch, (const long long *)idx, (uint4 *)s, (uint4 *)s2, a, r, t);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

// Measurement aid for bench.py: the same gather launched `reps` times back to back on `stream`, so that HIP
// events see the kernel and not the per-call host overhead of a Python loop.
extern "C" int fb_replay_profile_gather(fb_replay_t h, int batch, const int64_t *idx, uint8_t *s, uint8_t *s2, uint8_t *a, float *r,
                                        uint8_t *t, int reps, void *stream) {
    FB_REQUIRE(reps >= 1, "fb_replay_profile_gather: reps must be >= 1");
    for (int i = 0; i < reps; i++) {
        int rc = fb_replay_gather(h, batch, idx, s, s2, a, r, t, stream);
        if (rc != FB_OK) return rc;
    }
    return FB_OK;
}

static int launch_per_update(fb_replay *h, int batch, const int64_t *idx, float *abs_err, const float *prio, int write_back, hipStream_t st) {
    const size_t bytes = (size_t)UpdLayout(batch).bytes;                 // 10 KB at a batch of 32, 150 KB at 256
    static size_t allowed = 48 * 1024;                                   // (dynamic LDS beyond the default limit has to be asked for, once)
    if (bytes > allowed) {
        FB_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(per_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        allowed = bytes;
    }
    hipLaunchKernelGGL(per_update_kernel, dim3(1), dim3(256), bytes, st, h->P, batch, (const long long *)idx, abs_err, prio, write_back);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_replay_update_priorities(fb_replay_t h, int batch, const int64_t *idx, float *abs_err,
                                           const float *priorities_or_null, void *stream) {
    FB_REQUIRE(h && idx && (abs_err || priorities_or_null), "fb_replay_update_priorities: NULL argument");
    FB_REQUIRE(h->P.kind == FB_REPLAY_PER, "fb_replay_update_priorities: not a prioritized memory");
    FB_REQUIRE(batch >= 1 && batch <= MAXB, "fb_replay_update_priorities: batch must be in 1..%d", MAXB);
    const int rcj = per_join(h, fb_stream(stream));
    if (rcj != FB_OK) return rcj;
    if (h->per_mode == FB_PER_FAST) {
        hipLaunchKernelGGL(per_update_fast_kernel, dim3(1), dim3(256), 0, fb_stream(stream), h->P, batch, (const long long *)idx, abs_err,
                           priorities_or_null, 1);
        FB_LAUNCH_CHECK();
        return FB_OK;
    }
    return launch_per_update(h, batch, idx, abs_err, priorities_or_null, 1, fb_stream(stream));
}

// fb_vec_step's own Memory.batch_update when it stays in line: the same update, but the caller's abs_err array is left as the loss wrote
// it -- |TD error| -- exactly as in the run-ahead form (fb_replay_update_priorities_ahead), so that what fb_vec_step leaves in abs_err
// does not depend on which of the two forms the env count selected (the reference's in-place `abs_errors += epsilon`,
// BrainPrioritizedReplyDQN.py:147, stays inside the kernel; the stand-alone fb_replay_update_priorities does it in place as before)
int fb_replay_update_priorities_keep(fb_replay_t h, int batch, const int64_t *idx, const float *abs_err, void *stream) {
    FB_REQUIRE(h && idx && abs_err && h->P.kind == FB_REPLAY_PER && batch >= 1 && batch <= MAXB, "fb_replay_update_priorities_keep: bad argument");
    const int rcj = per_join(h, fb_stream(stream));
    if (rcj != FB_OK) return rcj;
    if (h->per_mode == FB_PER_FAST) {
        hipLaunchKernelGGL(per_update_fast_kernel, dim3(1), dim3(256), 0, fb_stream(stream), h->P, batch, (const long long *)idx, const_cast<float *>(abs_err),
                           (const float *)nullptr, 0);
        FB_LAUNCH_CHECK();
        return FB_OK;
    }
    return launch_per_update(h, batch, idx, const_cast<float *>(abs_err), nullptr, 0, fb_stream(stream));
}

// Memory.batch_update on the memory's SIDE stream, behind everything `stream` holds so far (reference-order mode; returns 1 when issued,
// 0 when the caller has to make the ordinary call).  What follows batch_update in the loop -- the next step's acting forward and env
// step -- does not touch the tree; the next thing that does is Memory.store of the next step, which fb_replay_per_store_ahead puts on the
// same stream right behind this, and the push / sample that follow join that.  The kernel (17.6 us at a batch of 32, one workgroup)
// leaves the critical path; abs_err is read, not written (the in-place `abs_errors += epsilon` of :147 stays inside the kernel).
int fb_replay_update_priorities_ahead(fb_replay_t h, int batch, const int64_t *idx, const float *abs_err, void *stream) {
    // Only where the acting phase it hides behind is long enough: the kernel (139 registers per lane) cannot share a CU with a workgroup of
    // the acting trunk (2 waves x 224 registers per SIMD), so it starts when the trunk's first round of workgroups retires -- 30 us in --
    // and at 1024 / 2048 envs (one / two rounds) the side stream then finishes AFTER the env step: measured 185 / 206 us per step against
    // 164 / 196 in line; at 4096 envs (four rounds) 272 against 286.  FB_PER_UPDATE_AHEAD=0 / 1 forces the in-line / run-ahead form.
    static const int knob = getenv("FB_PER_UPDATE_AHEAD") ? atoi(getenv("FB_PER_UPDATE_AHEAD")) : -1;
    if (knob == 0 || !h || h->P.kind != FB_REPLAY_PER || h->per_mode != FB_PER_EXACT || !h->side || !idx || !abs_err || batch < 1 || batch > MAXB) return 0;
    if (knob < 0 && h->P.n_envs < 4096) return 0;
    if (h->store_ahead || h->upd_pending) return 0;                        // (not in the loop's order: take the ordinary path)
    if (!per_side_usable(h, stream)) return 0;
    hipStream_t st = fb_stream(stream);
    if (hipEventRecord(h->ev_fork, st) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_fork, 0) != hipSuccess) return 0;
    if (launch_per_update(h, batch, idx, const_cast<float *>(abs_err), nullptr, 0, h->side) != FB_OK ||
        hipEventRecord(h->ev_upd, h->side) != hipSuccess) {
        (void)hipStreamSynchronize(h->side);                               // whatever did get issued is done before anyone goes on
        return 0;
    }
    h->upd_pending = true;
    return 1;
}

extern "C" int fb_replay_set_per_mode(fb_replay_t h, int mode) {
    FB_REQUIRE(h && (mode == FB_PER_EXACT || mode == FB_PER_FAST), "fb_replay_set_per_mode: mode must be FB_PER_EXACT or FB_PER_FAST");
    FB_REQUIRE(h->P.kind == FB_REPLAY_PER, "fb_replay_set_per_mode: not a prioritized memory");
    h->per_mode = mode;
    return FB_OK;
}

extern "C" int fb_replay_size(fb_replay_t h, int64_t *size_host) {
    FB_REQUIRE(h && size_host, "fb_replay_size: NULL argument");
    FB_CHECK_HIP(hipDeviceSynchronize());
    ReplayDev d;
    FB_CHECK_HIP(hipMemcpy(&d, h->P.dev, sizeof(d), hipMemcpyDeviceToHost));
    if (d.error) {
        const int code = d.error;
        d.error = 0;
        FB_CHECK_HIP(hipMemcpy(h->P.dev, &d, sizeof(d), hipMemcpyHostToDevice));
        return fb_set_error(code == 2 ? FB_ERR_STATE : FB_ERR_INVALID,
                            code == 2 ? "replay: Sample larger than population or is negative"
                                      : "replay: an index handed to gather/update was out of range");
    }
    const long long total = d.steps * h->P.n_envs;
    *size_host = total < h->P.cap ? total : h->P.cap;
    return FB_OK;
}

extern "C" int fb_replay_per_tree(fb_replay_t h, double *tree_host, int64_t *data_pointer, int64_t *size, double *beta) {
    FB_REQUIRE(h && h->P.kind == FB_REPLAY_PER, "fb_replay_per_tree: not a prioritized memory");
    FB_CHECK_HIP(hipDeviceSynchronize());
    ReplayDev d;
    FB_CHECK_HIP(hipMemcpy(&d, h->P.dev, sizeof(d), hipMemcpyDeviceToHost));
    if (tree_host) FB_CHECK_HIP(hipMemcpy(tree_host, h->P.tree, sizeof(double) * (size_t)(2 * h->P.cap - 1), hipMemcpyDeviceToHost));
    if (data_pointer) *data_pointer = d.per_pointer;
    if (size) *size = d.per_size;
    if (beta) *beta = d.beta;
    return FB_OK;
}

// ------------------------------------------------------------------ checkpoint of the memory
// The reference saves the network and three scalars and FORGETS the replay memory (BrainDQN.py:176-192,227-233: a resumed run
// observes for 1 000 steps again).  These three calls let a caller keep it: one opaque host blob holds the frame ring, the
// per-transition action / reward / terminal rows, the device-side counters (steps, SumTree pointer / size / beta, Philox call
// counter), the sampler's MT19937 state and, for a prioritized memory, the three heaps -- everything a later
// fb_replay_sample / gather / push needs to continue bit for bit.  Synchronous; the blob is only valid for a memory created with
// the same capacity, env count and kind.
namespace {
struct ReplayBlobHeader {
    uint64_t magic, version;
    int64_t cap, host_steps;
    int32_t n_envs, kind, t_f, rng_kind, per_mode, pad;
    uint32_t seed_lo, seed_hi;
    uint64_t total_bytes;
};
constexpr uint64_t REPLAY_BLOB_MAGIC = 0x5052424644514e46ull;        // "FNQDFBRP"
struct BlobPart { void *dev; size_t bytes; };
static int blob_parts(fb_replay *h, BlobPart *parts) {
    const ReplayParams &P = h->P;
    const size_t slots = (size_t)P.t_f * P.n_envs;
    int n = 0;
    parts[n++] = BlobPart{P.dev, sizeof(ReplayDev)};
    parts[n++] = BlobPart{P.mt, sizeof(FbMT)};
    parts[n++] = BlobPart{P.bits, slots * WORDS * sizeof(unsigned long long)};
    parts[n++] = BlobPart{P.act, slots};
    parts[n++] = BlobPart{P.rew, slots * sizeof(float)};
    parts[n++] = BlobPart{P.term, slots};
    if (P.kind == FB_REPLAY_PER) {
        const size_t nb = sizeof(double) * (size_t)(2 * P.cap - 1);
        parts[n++] = BlobPart{P.tree, nb}; parts[n++] = BlobPart{P.maxt, nb}; parts[n++] = BlobPart{P.mint, nb};
    }
    return n;
}
}  // namespace

extern "C" int fb_replay_state_bytes(fb_replay_t h, size_t *bytes_host) {
    FB_REQUIRE(h && bytes_host, "fb_replay_state_bytes: NULL argument");
    BlobPart parts[9];
    const int n = blob_parts(h, parts);
    size_t total = sizeof(ReplayBlobHeader);
    for (int i = 0; i < n; i++) total += (parts[i].bytes + 15) & ~(size_t)15;
    *bytes_host = total;
    return FB_OK;
}

extern "C" int fb_replay_save_state(fb_replay_t h, void *blob_host, size_t bytes) {
    FB_REQUIRE(h && blob_host, "fb_replay_save_state: NULL argument");
    size_t need = 0;
    fb_replay_state_bytes(h, &need);
    FB_REQUIRE(bytes >= need, "fb_replay_save_state: the blob needs %zu bytes, %zu given", need, bytes);
    FB_CHECK_HIP(hipDeviceSynchronize());
    ReplayBlobHeader hd;
    memset(&hd, 0, sizeof(hd));
    hd.magic = REPLAY_BLOB_MAGIC; hd.version = 1; hd.cap = h->P.cap; hd.host_steps = h->host_steps; hd.n_envs = h->P.n_envs; hd.kind = h->P.kind;
    hd.t_f = h->P.t_f; hd.rng_kind = h->P.rng_kind; hd.per_mode = h->per_mode; hd.seed_lo = h->P.seed_lo; hd.seed_hi = h->P.seed_hi; hd.total_bytes = need;
    memcpy(blob_host, &hd, sizeof(hd));
    BlobPart parts[9];
    const int n = blob_parts(h, parts);
    char *o = (char *)blob_host + sizeof(hd);
    for (int i = 0; i < n; i++) {
        FB_CHECK_HIP(hipMemcpy(o, parts[i].dev, parts[i].bytes, hipMemcpyDeviceToHost));
        const size_t padded = (parts[i].bytes + 15) & ~(size_t)15;
        memset(o + parts[i].bytes, 0, padded - parts[i].bytes);          // (alignment gaps: equal memories give equal blobs)
        o += padded;
    }
    return FB_OK;
}

extern "C" int fb_replay_load_state(fb_replay_t h, const void *blob_host, size_t bytes) {
    FB_REQUIRE(h && blob_host && bytes >= sizeof(ReplayBlobHeader), "fb_replay_load_state: bad argument");
    ReplayBlobHeader hd;
    memcpy(&hd, blob_host, sizeof(hd));
    size_t need = 0;
    fb_replay_state_bytes(h, &need);
    FB_REQUIRE(hd.magic == REPLAY_BLOB_MAGIC && hd.version == 1, "fb_replay_load_state: not a replay checkpoint");
    FB_REQUIRE(hd.cap == h->P.cap && hd.n_envs == h->P.n_envs && hd.kind == h->P.kind && hd.t_f == h->P.t_f && hd.total_bytes == need && bytes >= need,
               "fb_replay_load_state: the checkpoint is of a memory with capacity %lld, %d envs, kind %d; this one has %lld, %d, %d",
               (long long)hd.cap, hd.n_envs, hd.kind, (long long)h->P.cap, h->P.n_envs, h->P.kind);
    FB_CHECK_HIP(hipDeviceSynchronize());
    BlobPart parts[9];
    const int n = blob_parts(h, parts);
    const char *o = (const char *)blob_host + sizeof(hd);
    for (int i = 0; i < n; i++) {
        FB_CHECK_HIP(hipMemcpy(parts[i].dev, o, parts[i].bytes, hipMemcpyHostToDevice));
        o += (parts[i].bytes + 15) & ~(size_t)15;
    }
    h->host_steps = hd.host_steps; h->P.rng_kind = hd.rng_kind; h->per_mode = hd.per_mode; h->P.seed_lo = hd.seed_lo; h->P.seed_hi = hd.seed_hi;
    return FB_OK;
}
