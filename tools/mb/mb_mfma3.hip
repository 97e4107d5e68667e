// Micro-benchmark 4: sustained v_mfma_f32_32x32x16_bf16 rate of one wave per SIMD (4-wave workgroups), chains of 6
// dependent MFMAs alternating between two accumulators -- the issue pattern of conv23_sp_kernel -- on random
// bf16 operands and on zeros, for 205 and 256 workgroups.  ns per MFMA -> effective clock (32 cycles each).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, float *__restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    uint4 a[6], b[6];
    for (int i = 0; i < 6; i++) { a[i] = in[(t * 12 + i) & 65535]; b[i] = in[(t * 12 + 6 + i) & 65535]; }
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[i]), acc0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; i++) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[i]), __builtin_bit_cast(bf16x8, a[i]), acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
    out[t] = s;
}

int main() {
    uint4 *in; float *out;
    CK(hipMalloc(&in, 65536 * 16)); CK(hipMalloc(&out, 256 * 256 * 4 * 4));
    uint32_t *h = (uint32_t *)malloc(65536 * 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int i = 0; i < 65536 * 4; i++) {      // two bf16 per word, values in +-[0.5, 2) (random) or 0
            uint32_t lo = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15), hi = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15);
            h[i] = mode ? 0u : (lo | hi << 16);
        }
        CK(hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice));
        for (int grid : {205, 256, 1024}) {
            const int iters = 2000; float ms;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters);
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double per = ms * 1e6 / (iters * 12.0) / (grid > 256 ? grid / 256.0 : 1.0);
            printf("%s grid %4d: %.2f ns per MFMA per SIMD -> %.2f GHz at 32 cycles\n", mode ? "zeros " : "random", grid, per, 32.0 / per);
        }
    }
    return 0;
}
