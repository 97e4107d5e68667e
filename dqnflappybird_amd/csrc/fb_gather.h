// Minibatch assembly from the packed frame ring -- device code shared by the replay kernels (fb_replay.hip) and the Adam
// kernel, which can carry the NEXT train step's gather as a rider (fb_qnet.hip, fb_train_steps).  Included inside each
// translation unit's anonymous namespace.
#pragma once

__device__ __forceinline__ size_t fb_frame_off(const FbGatherCtx &P, long long f, int e) {
    if (f < 0) f = 0;                                   // setInitState: the first frame four times
    return ((size_t)(f % P.t_f) * P.n_envs + e) * 100;
}

// Where transition `j` of a sampled minibatch lives: (tt, e) = (time slot, env) of its frame t; its window is frames tt - 3 .. tt + 1 of
// env e.  j is a deque position (uniform memory, 0 = oldest) or a SumTree leaf index (prioritized memory); an index outside the
// filled part raises the memory's error flag (when `flag`) and reads transition 0.
__device__ __forceinline__ void fb_ring_locate(const FbGatherCtx &P, long long steps, long long j, bool flag, long long &tt, int &e) {
    const long long total = steps * P.n_envs;
    long long g;
    if (P.kind == FB_REPLAY_PER) {
        long long d = j - (P.cap - 1);
        if (d < 0 || d >= P.cap || d >= total) { if (flag) *P.error = 1; d = 0; }
        g = d + P.cap * ((total - 1 - d) / P.cap);  // newest transition living in data slot d
    } else {
        const long long size = total < P.cap ? total : P.cap;
        if (j < 0 || j >= size) { if (flag) *P.error = 1; j = 0; }
        g = total - size + j;                       // deque position j, 0 = oldest
    }
    tt = g / P.n_envs; e = (int)(g - tt * P.n_envs);
}

// One thread expands 4 pixels x 4 stacked frames = 16 contiguous bytes of s (and of s').
__device__ __forceinline__ uint32_t expand4(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3, int q) {
    return (((n0 >> q) & 1u) * 0xFFu) | (((n1 >> q) & 1u) * 0xFF00u) | (((n2 >> q) & 1u) * 0xFF0000u) |
           (((n3 >> q) & 1u) * 0xFF000000u);
}

// thread tid of B * 1600: sample tid / 1600, 16-byte chunk tid % 1600 of s (and of s')
template <bool CURRENT>
__device__ __forceinline__ void gather_body(const FbGatherCtx &P, long long steps, int B, const long long *__restrict__ idx,
                                            uint4 *__restrict__ s, uint4 *__restrict__ s2, uint8_t *__restrict__ a,
                                            float *__restrict__ r, uint8_t *__restrict__ t, long long tid) {
    // `steps` (pushes so far) comes by value from the host's mirror: one dependent global round trip less than reading
    // ReplayDev::steps here
    if (tid >= (long long)B * 1600) return;
    const int b = (int)(tid / 1600), chunk = (int)(tid - (long long)b * 1600);
    long long tt; int e;
    if (CURRENT) { tt = steps; e = b; }
    else fb_ring_locate(P, steps, idx[b], chunk == 0, tt, e);
    const int p = chunk * 4, w = p >> 6, sh = p & 63;
    uint32_t n[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (CURRENT && k == 4) { n[k] = 0; break; }
        n[k] = (uint32_t)(P.bits[fb_frame_off(P, tt - 3 + k, e) + w] >> sh) & 0xFu;
    }
    uint4 o;
    o.x = expand4(n[0], n[1], n[2], n[3], 0); o.y = expand4(n[0], n[1], n[2], n[3], 1);
    o.z = expand4(n[0], n[1], n[2], n[3], 2); o.w = expand4(n[0], n[1], n[2], n[3], 3);
    s[tid] = o;
    if (!CURRENT) {
        o.x = expand4(n[1], n[2], n[3], n[4], 0); o.y = expand4(n[1], n[2], n[3], n[4], 1);
        o.z = expand4(n[1], n[2], n[3], n[4], 2); o.w = expand4(n[1], n[2], n[3], n[4], 3);
        s2[tid] = o;
        if (chunk == 0) {
            const size_t mo = (size_t)(tt % P.t_f) * P.n_envs + e;
            a[b] = P.act[mo]; r[b] = P.rew[mo]; t[b] = P.term[mo];
        }
    }
}
