// Micro-benchmark 5: what a launch costs as a function of its shape.  A kernel whose workgroups do next to nothing (one global load,
// one store), launched back to back on one stream (each launch depends on the previous one: the kernel boundary is included), for
// grids of G workgroups x T threads with L bytes of LDS and a register budget of V VGPRs.  us per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T, int V>
__global__ __launch_bounds__(T) void k(const float *__restrict__ in, float *__restrict__ out, int lds_words) {
    extern __shared__ float sm[];
    float v = in[blockIdx.x & 1023];
    if (V > 64) {                        // hold > 64 registers live
        float r[96];
#pragma unroll
        for (int i = 0; i < 96; i++) r[i] = v * (float)(i + threadIdx.x);
#pragma unroll
        for (int i = 0; i < 96; i++) asm volatile("" : "+v"(r[i]));
#pragma unroll
        for (int i = 0; i < 96; i++) v += r[i];
    }
    if (lds_words && threadIdx.x == 0) sm[lds_words - 1] = v;
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

template <int T, int V>
int run(const float *in, float *out, int G, int lds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k<T, V>), dim3(G), dim3(T), lds, 0, in, out, lds / 4);
    CK(hipEventRecord(e0, 0));
    const int N = 400;
    for (int i = 0; i < N; i++) hipLaunchKernelGGL((k<T, V>), dim3(G), dim3(T), lds, 0, in, out, lds / 4);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("G %5d x T %4d  lds %6d  vgpr<=%3d  waves %6d : %6.2f us per launch\n", G, T, lds, V, G * T / 64, ms * 1e3 / N);
    return 0;
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 65536 * 4)); CK(hipMemset(in, 0, 4096));
    for (int lds : {0, 30000, 60000}) {
        for (int G : {1, 64, 256, 1024, 4096}) {
            run<64, 64>(in, out, G, lds); run<256, 64>(in, out, G, lds); run<512, 64>(in, out, G, lds); run<1024, 64>(in, out, G, lds);
        }
    }
    for (int G : {256, 1024}) { run<256, 128>(in, out, G, 0); run<512, 128>(in, out, G, 0); }
    return 0;
}
