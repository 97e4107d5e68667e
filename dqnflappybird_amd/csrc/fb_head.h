// fc2 / dueling head (+ epsilon-greedy action for the acting path) of ONE state on one wave, from the fc1 partial sums
// hfp[ks][state][FC] -- device code shared by head_kernel (fb_qnet.hip) and the env step kernel, which can carry it as a
// rider (fb_env.hip, fb_vec_step).  Included inside each translation unit's anonymous namespace.
#pragma once

constexpr int MAXA = 8;              // actions
constexpr int FC1_KS = 5;            // most fc1 K slices any forward path produces

__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// P = the network's flat parameters, smp = row of the state in hf / q / actions.  Returns the action (0 without C.actions).
// AT = the number of actions when it is known at compile time (2: the game's), MAXA = read it from C.A.
// No load sits under a branch (hipcc waits vmcnt(0) at the join: with a run-time `if (a < A)` around every fc2 weight this
// was a chain of ~40 dependent L2 round trips): every parameter is fetched from a clamped, valid address -- the tail
// parameters (b_q, b_v) together with the first round's partial sums -- and masked by selects.
template <int AT>
__device__ __forceinline__ int head_one_t(const HeadCore &C, const float *__restrict__ P, int smp, int lane) {
    const int A = AT == MAXA ? C.A : AT;
    int action = 0;
    float acc[AT + 1], bqv[AT];
#pragma unroll
    for (int a = 0; a <= AT; a++) acc[a] = 0.f;
#pragma unroll
    for (int a = 0; a < AT; a++) bqv[a] = P[C.off.bq + (a < A ? a : 0)];
    const float bvv = P[C.dueling ? C.off.bv : C.off.bq];
    const int wv_off = C.dueling ? C.off.wv : C.off.bf1;         // valid either way
    // a lane takes 4 consecutive units per round (FC % 128 == 0): the nks partial sums and the bias arrive as float4,
    // all rounds' loads in flight together; the per-unit arithmetic and its order are those of fc1_out
    for (int j0 = 4 * lane; j0 < C.FC; j0 += 256) {
        float4 t[FC1_KS];
#pragma unroll
        for (int ks = 0; ks < FC1_KS; ks++)
            t[ks] = sel4(ks < C.nks, *reinterpret_cast<const float4 *>(C.hf + ((size_t)(ks < C.nks ? ks : 0) * C.stot + smp) * C.FC + j0));
        const float4 bv = *reinterpret_cast<const float4 *>(P + C.off.bf1 + j0);
        float w[4][AT];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int a = 0; a < AT; a++) w[e][a] = P[C.off.wq + (j0 + e) * A + (a < A ? a : 0)];
        float wvv[4];                                             // (W_v starts 4-byte aligned only: no float4)
#pragma unroll
        for (int e = 0; e < 4; e++) wvv[e] = P[wv_off + j0 + e];
        float4 v = t[0];
#pragma unroll
        for (int ks = 1; ks < FC1_KS; ks++) { v.x += t[ks].x; v.y += t[ks].y; v.z += t[ks].z; v.w += t[ks].w; }
        const float x4[4] = {fmaxf(v.x + bv.x, 0.f), fmaxf(v.y + bv.y, 0.f), fmaxf(v.z + bv.z, 0.f), fmaxf(v.w + bv.w, 0.f)};
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int a = 0; a < AT; a++) acc[a] = fmaf(x4[e], w[e][a], acc[a]);          // columns a >= A are never read
            acc[AT] = fmaf(x4[e], wvv[e], acc[AT]);                                      // used by the dueling head only
        }
    }
#pragma unroll
    for (int a = 0; a <= AT; a++)
        for (int o = 32; o > 0; o >>= 1) acc[a] += __shfl_xor(acc[a], o);
    float qv[AT];
    float mean = 0.f;
#pragma unroll
    for (int a = 0; a < AT; a++) { qv[a] = a < A ? acc[a] + bqv[a] : 0.f; if (a < A) mean += qv[a]; }
    if (C.dueling) {                                             // Q = V + (A - mean_a A)
        const float V = acc[AT] + bvv;
        mean /= (float)A;
#pragma unroll
        for (int a = 0; a < AT; a++) qv[a] = V + (qv[a] - mean);
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < AT; a++) if (a < A) C.q[(size_t)smp * A + a] = qv[a];
        if (C.actions) {                                         // BrainDQN.py:103-108
            int best = 0;
#pragma unroll
            for (int a = 1; a < AT; a++) if (a < A && qv[a] > qv[best]) best = a;           // np.argmax: first maximum
            const fb_u4 o = fb_philox(C.seed_lo, C.seed_hi, (uint32_t)smp, C.step_lo, FB_STREAM_EPS, C.step_hi);
            const float u = (float)(o.x >> 8) * (1.0f / 16777216.0f);                      // random.random()
            if (u <= C.epsilon) best = (int)(((unsigned long long)o.y * (unsigned)A) >> 32);   // randrange(A)
            C.actions[smp] = (uint8_t)best;
            action = best;
        }
    }
    return __builtin_amdgcn_readfirstlane(action);
}

__device__ __forceinline__ int head_one(const HeadCore &C, const float *__restrict__ P, int smp, int lane) {
    return C.A == 2 ? head_one_t<2>(C, P, smp, lane) : head_one_t<MAXA>(C, P, smp, lane);
}
