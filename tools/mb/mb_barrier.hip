// Micro-benchmark 3: what does a phase boundary cost on gfx950?
//  (a) a dependent kernel launch in a stream, (b) the same inside a hipGraph,
//  (c) a grid-wide barrier inside one persistent kernel (agent-scope atomics + fences),
//      with every workgroup writing a line before the barrier and reading another group's line after it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float *buf, int it) {
    const int g = blockIdx.x, n = gridDim.x;
    if (threadIdx.x < 16) buf[((it & 1) * n + g) * 16 + threadIdx.x] = buf[(((it + 1) & 1) * n + (g + 1) % n) * 16 + threadIdx.x] + 1.f;
}

__device__ inline void grid_barrier(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                       // agent scope: write back this XCD's dirty L2 lines
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__global__ void persistent(float *buf, unsigned *ctr, int iters) {
    const int g = blockIdx.x, n = gridDim.x;
    for (int it = 0; it < iters; it++) {
        if (threadIdx.x < 16) buf[((it & 1) * n + g) * 16 + threadIdx.x] = buf[(((it + 1) & 1) * n + (g + 1) % n) * 16 + threadIdx.x] + 1.f;
        grid_barrier(ctr, (unsigned)(it + 1) * n);
    }
}

int main() {
    float *buf; unsigned *ctr;
    CK(hipMalloc(&buf, 2 * 4096 * 16 * 4)); CK(hipMemset(buf, 0, 2 * 4096 * 16 * 4)); CK(hipMalloc(&ctr, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {64, 256, 512, 1024};
    for (int threads : {64, 256, 512}) for (int grid : grids) {
        const int K = 200; float ms;
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(tiny, dim3(grid), dim3(threads), 0, s, buf, i);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(tiny, dim3(grid), dim3(threads), 0, s, buf, i);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
        const float us_stream = ms * 1000.f / K;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(tiny, dim3(grid), dim3(threads), 0, s, buf, i);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float us_graph = ms * 1000.f / K;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        float us_bar = -1.f;
        if (grid <= 512) {   // all workgroups must be co-resident: 256 CUs x >= 2 groups of <= 512 threads
            CK(hipMemsetAsync(ctr, 0, 64, s));
            hipLaunchKernelGGL(persistent, dim3(grid), dim3(threads), 0, s, buf, ctr, 10);
            CK(hipMemsetAsync(ctr, 0, 64, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(persistent, dim3(grid), dim3(threads), 0, s, buf, ctr, K);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
            us_bar = ms * 1000.f / K;
        }
        printf("threads %3d grid %4d: stream launch %.2f us  graph node %.2f us  grid barrier %.2f us\n", threads, grid, us_stream, us_graph, us_bar);
    }
    return 0;
}
