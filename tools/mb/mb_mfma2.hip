// Micro-benchmark 2: conv2 shape (n = 1024): does (a) a k-quad packed weight layout (float4 B loads) and
// (b) wave-private LDS staging of the activation runs (coalesced global loads) pay?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BMODE, int AMODE>   // BMODE 0: dword column loads, 1: float4 from packed [k/4][n][4];  AMODE 0: per-lane float4 runs, 1: LDS staged
__global__ __launch_bounds__(256) void k(const float *__restrict__ p1, const float *__restrict__ w, const float *__restrict__ wp,
                                         float *__restrict__ out, int M) {
    __shared__ float As[AMODE ? 4 * 64 * 36 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, i = lane & 31, j = lane & 31;
    const int tile = blockIdx.x * 4 + wave, n0 = blockIdx.y * 32;
    if (tile * 32 >= M) return;
    const int m = tile * 32 + i, b = m / 25, rem = m - b * 25, oy = rem / 5, ox = rem - oy * 5;
    int sb[4], soy[4], sox[4];
    for (int u = 0; u < 4; u++) { const int mm = tile * 32 + (lane >> 3) + 8 * u; sb[u] = mm / 25; const int r2 = mm - sb[u] * 25; soy[u] = r2 / 5; sox[u] = r2 - soy[u] * 5; }
    f32x16 acc = {0};
#pragma unroll 2
    for (int c = 0; c < 8; c++) {
        const int ky = c >> 1, kx = 2 * (c & 1) + hl;
        float a[32], bb[32];
        if (AMODE == 0) {
            const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
            const bool ok = m < M && iy >= 0 && iy < 10 && ix >= 0 && ix < 10;
            const float *arun = p1 + (((size_t)b * 10 + (ok ? iy : 0)) * 10 + (ok ? ix : 0)) * 32;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = reinterpret_cast<const float4 *>(arun)[q];
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
        } else {
            float *A = As + wave * 64 * 36;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int u = q & 3, half = q >> 2, kx2 = 2 * (c & 1) + half;
                const int iy = soy[u] * 2 + ky - 1, ix = sox[u] * 2 + kx2 - 1;
                const bool ok = iy >= 0 && iy < 10 && ix >= 0 && ix < 10 && sb[u] * 25 < M;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4 *>(p1 + (((size_t)sb[u] * 10 + iy) * 10 + ix) * 32 + (lane & 7) * 4);
                *reinterpret_cast<float4 *>(A + (q * 8 + (lane >> 3)) * 36 + (lane & 7) * 4) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float4 v = *reinterpret_cast<const float4 *>(A + (i + 32 * hl) * 36 + 4 * q);
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (BMODE == 0) {
            const float *bcol = w + ((ky * 4 + kx) * 32) * 64 + n0 + j;
#pragma unroll
            for (int t = 0; t < 32; t++) bb[t] = bcol[t * 64];
        } else {
            const float4 *bq = reinterpret_cast<const float4 *>(wp) + (size_t)((ky * 4 + kx) * 8) * 64 + n0 + j;
#pragma unroll
            for (int q = 0; q < 8; q++) { const float4 v = bq[q * 64]; bb[4 * q] = v.x; bb[4 * q + 1] = v.y; bb[4 * q + 2] = v.z; bb[4 * q + 3] = v.w; }
        }
#pragma unroll
        for (int t = 0; t < 32; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bb[t], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int mr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
        if (mr < M) out[(size_t)mr * 64 + n0 + j] = acc[r];
    }
}

template <int BMODE, int AMODE>
float run(const float *p1, const float *w, const float *wp, float *out, int n) {
    const int M = n * 25, tiles = (M + 31) / 32;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<BMODE, AMODE>), dim3((tiles + 3) / 4, 2), dim3(256), 0, 0, p1, w, wp, out, M);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<BMODE, AMODE>), dim3((tiles + 3) / 4, 2), dim3(256), 0, 0, p1, w, wp, out, M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 20;
}

int main() {
    const int n = 1024;
    float *p1, *w, *wp, *out;
    hipMalloc(&p1, (size_t)n * 3200 * 4); hipMalloc(&w, 32768 * 4); hipMalloc(&wp, 32768 * 4); hipMalloc(&out, (size_t)n * 1600 * 4);
    hipMemset(p1, 0, (size_t)n * 3200 * 4); hipMemset(w, 0, 32768 * 4); hipMemset(wp, 0, 32768 * 4);
    printf("B dword, A direct  : %7.1f us\n", run<0, 0>(p1, w, wp, out, n));
    printf("B float4 packed    : %7.1f us\n", run<1, 0>(p1, w, wp, out, n));
    printf("A via private LDS  : %7.1f us\n", run<0, 1>(p1, w, wp, out, n));
    printf("both               : %7.1f us\n", run<1, 1>(p1, w, wp, out, n));
    return 0;
}
