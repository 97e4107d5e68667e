// Micro-benchmark 6: a stream gated on a device word (hipStreamWaitValue32 on signal memory) that a KERNEL on another stream opens.
//   1. does a kernel's store to hipMallocSignalMemory memory release a hipStreamWaitValue32(GTE) on another stream?
//   2. what does the gate cost when it is already open (back-to-back kernels with a satisfied wait between them)?
//   3. latency from the opening store to the start of the gated kernel.
// Safety: a host watchdog writes the value itself after 2 s, so the gated stream always drains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void open_gate(volatile unsigned *gate, unsigned v, long long *stamp, int spin) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    *gate = v;
    __threadfence_system();
    stamp[0] = wall_clock64();
}
__global__ void gated(long long *stamp, float *out) { stamp[1] = wall_clock64(); out[threadIdx.x] = 1.f; }
__global__ void tiny(float *out) { out[threadIdx.x] += 1.f; }

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned *gate; long long *stamp; float *out;
    CK(hipExtMallocWithFlags((void **)&gate, 8, hipMallocSignalMemory));
    CK(hipMalloc(&stamp, 64)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(stamp, 0, 64));
    *gate = 0;                                       // (signal memory is host-visible)
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    // 1 + 3
    for (unsigned round = 1; round <= 5; round++) {
        CK(hipStreamWaitValue32(sb, gate, round, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(gated, dim3(1), dim3(64), 0, sb, stamp, out);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        hipLaunchKernelGGL(open_gate, dim3(1), dim3(1), 0, sa, gate, round, stamp, 100000);       // spins 1 ms (100 MHz clock), then opens
        bool by_watchdog = false;
        auto t0 = std::chrono::steady_clock::now();
        while (hipStreamQuery(sb) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { *gate = round; by_watchdog = true; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        long long st[2];
        CK(hipMemcpy(st, stamp, 16, hipMemcpyDeviceToHost));
        printf("round %u: gate opened by %s; store -> gated kernel start: %.2f us\n", round, by_watchdog ? "the HOST WATCHDOG (kernel store not seen)" : "the kernel",
               (st[1] - st[0]) / 100.0);
    }
    // 2: chain of tiny kernels with / without an open gate in front of each
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int with = 0; with < 2; with++) {
        const int N = 500;
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, out);
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < N; i++) {
            if (with) CK(hipStreamWaitValue32(sb, gate, 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, out);
        }
        CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("chain of tiny kernels %s an open gate in front of each: %.2f us per kernel\n", with ? "WITH" : "without", ms * 1e3 / N);
    }
    // cross-stream event wait, already satisfied, for comparison
    {
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CK(hipEventRecord(ev, sa)); CK(hipStreamSynchronize(sa));
        const int N = 500;
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < N; i++) { CK(hipStreamWaitEvent(sb, ev, 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, out); }
        CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("chain of tiny kernels with a SATISFIED cross-stream event wait in front of each: %.2f us per kernel\n", ms * 1e3 / N);
    }
    // ping-pong between two streams through events: the cost of a cross-stream hop on the critical path
    {
        hipEvent_t ea, eb; CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
        const int N = 300;
        CK(hipEventRecord(e0, sa));
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, out); CK(hipEventRecord(ea, sa)); CK(hipStreamWaitEvent(sb, ea, 0));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, out + 64); CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0));
        }
        CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("ping-pong of tiny kernels between two streams through events: %.2f us per kernel (hop included)\n", ms * 1e3 / (2 * N));
    }
    return 0;
}
