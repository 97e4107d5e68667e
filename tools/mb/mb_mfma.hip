// Micro-benchmark: what bounds a one-wave-per-tile fp32-MFMA implicit GEMM (conv2 shape, n = 1024 samples)?
// Variants drop the A loads, the B loads, or both, and vary waves per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int NACC>   // MODE bit0: load A, bit1: load B
__global__ __launch_bounds__(256) void k(const float *__restrict__ p1, const float *__restrict__ w, float *__restrict__ out, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, i = lane & 31, j = lane & 31;
    const int tile = blockIdx.x * 4 + wave, n0 = blockIdx.y * 32;
    if (tile * 32 >= M) return;
    const int m = tile * 32 + i, b = m / 25, rem = m - b * 25, oy = rem / 5, ox = rem - oy * 5;
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; q++) acc[q] = (f32x16){0};
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int ky = c >> 1, kx = 2 * (c & 1) + hl;
        const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
        const bool ok = m < M && iy >= 0 && iy < 10 && ix >= 0 && ix < 10;
        const float *arun = p1 + (((size_t)b * 10 + (ok ? iy : 0)) * 10 + (ok ? ix : 0)) * 32;
        const float *bcol = w + ((ky * 4 + kx) * 32) * 64 + n0 + j;
        float a[32], bb[32];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
            if ((MODE & 1) && ok) v = reinterpret_cast<const float4 *>(arun)[q];
            a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < 32; t++) bb[t] = (MODE & 2) ? bcol[t * 64] : (float)(t + lane);
#pragma unroll
        for (int t = 0; t < 32; t++) acc[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bb[t], acc[t % NACC], 0, 0, 0);
    }
    f32x16 s = acc[0];
    for (int q = 1; q < NACC; q++) s += acc[q];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int mr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
        if (mr < M) out[(size_t)mr * 64 + n0 + j] = s[r];
    }
}

template <int MODE, int NACC>
float run(const float *p1, const float *w, float *out, int n) {
    const int M = n * 25, tiles = (M + 31) / 32;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, NACC>), dim3((tiles + 3) / 4, 2), dim3(256), 0, 0, p1, w, out, M);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<MODE, NACC>), dim3((tiles + 3) / 4, 2), dim3(256), 0, 0, p1, w, out, M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 20;
}

int main() {
    const int n = 1024;
    float *p1, *w, *out;
    CK(hipMalloc(&p1, (size_t)n * 3200 * 4)); CK(hipMalloc(&w, 32768 * 4)); CK(hipMalloc(&out, (size_t)n * 1600 * 4));
    CK(hipMemset(p1, 0, (size_t)n * 3200 * 4)); CK(hipMemset(w, 0, 32768 * 4));
    const double flop = 2.0 * n * 25 * 64 * 512;
    printf("conv2 shape, n=%d, ideal at 157.3 TF: %.1f us\n", n, flop / 157.3e6);
    printf("A+B loads, 1 acc : %7.1f us\n", run<3, 1>(p1, w, out, n));
    printf("A only           : %7.1f us\n", run<1, 1>(p1, w, out, n));
    printf("B only           : %7.1f us\n", run<2, 1>(p1, w, out, n));
    printf("no loads, 1 acc  : %7.1f us\n", run<0, 1>(p1, w, out, n));
    printf("no loads, 2 acc  : %7.1f us\n", run<0, 2>(p1, w, out, n));
    printf("no loads, 4 acc  : %7.1f us\n", run<0, 4>(p1, w, out, n));
    printf("A+B loads, 4 acc : %7.1f us\n", run<3, 4>(p1, w, out, n));
    return 0;
}
