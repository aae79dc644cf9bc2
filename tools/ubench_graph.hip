// ubench_graph.hip -- what a hipGraph of a proof's launch sequence can rely on, on this runtime (ROCm 7.2, gfx950):
//   1. stream capture over TWO streams (fork by event wait, join before the end), with hipMemsetAsync, device-to-pinned-host
//      copies and kernels whose launch configuration uses dynamic LDS;
//   2. hipEventRecordWithFlags(.., hipEventRecordExternal) inside the capture: are the events re-recorded by every launch of
//      the instantiated graph, and does hipEventElapsedTime give the span between two of them;
//   3. the wall time of a chain of short dependent kernels launched eagerly against the same chain replayed as a graph.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_graph tools/ubench_graph.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(unsigned* out, unsigned add, long long cycles) {
    extern __shared__ unsigned lds[];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { lds[0] = out[0] + add; out[0] = lds[0]; }
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned *d1, *d2, *host;
    CK(hipMalloc(&d1, 256)); CK(hipMalloc(&d2, 256));
    CK(hipHostMalloc((void**)&host, 256, hipHostMallocDefault));
    hipEvent_t fork, join, t0, t1, u0, u1;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&u0)); CK(hipEventCreate(&u1));
    const long long C100 = 100 * 100;        // ~100 us: wall_clock64 (s_memrealtime) counts at 100 MHz

    // ---- 1 + 2: capture over two streams with external event records
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(d1, 0, 256, s1));
    CK(hipEventRecord(fork, s1));
    CK(hipStreamWaitEvent(s2, fork, 0));                     // s2 joins the capture
    CK(hipEventRecordWithFlags(t0, s1, hipEventRecordExternal));
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 1024, s1, d1, 1u, C100);
    CK(hipEventRecordWithFlags(t1, s1, hipEventRecordExternal));
    CK(hipMemsetAsync(d2, 0, 256, s2));
    CK(hipEventRecordWithFlags(u0, s2, hipEventRecordExternal));
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 70000, s2, d2, 10u, C100);
    CK(hipEventRecordWithFlags(u1, s2, hipEventRecordExternal));
    CK(hipMemcpyAsync(host + 8, d2, 4, hipMemcpyDeviceToHost, s2));
    CK(hipEventRecord(join, s2));
    CK(hipStreamWaitEvent(s1, join, 0));
    CK(hipMemcpyAsync(host, d1, 4, hipMemcpyDeviceToHost, s1));
    CK(hipStreamEndCapture(s1, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    size_t nodes = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nodes));
    printf("captured graph: %zu nodes\n", nodes);
    for (int it = 0; it < 4; it++) {
        host[0] = host[8] = 0xdead;
        auto w0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(exec, s1));
        CK(hipStreamSynchronize(s1));
        double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        float a = -1, b = -1;
        hipError_t ea = hipEventElapsedTime(&a, t0, t1), eb = hipEventElapsedTime(&b, u0, u1);
        printf("launch %d: results %u %u (want 3 50)  wall %.3f ms  span s1 %.3f ms (%s; want ~0.3)  span s2 %.3f ms (%s; want ~0.5)\n", it, host[0], host[8],
               wall, a, hipGetErrorString(ea), b, hipGetErrorString(eb));
    }
    (void)hipGetLastError();

    // ---- 3: a chain of 64 short kernels, eager against graph
    const int N = 64;
    const long long C5 = 5 * 100;             // ~5 us each
    auto chain = [&](hipStream_t s) { for (int k = 0; k < N; k++) hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(256), 0, s, d1, 1u, C5); };
    for (int rep = 0; rep < 3; rep++) {
        auto w0 = std::chrono::steady_clock::now();
        chain(s1);
        CK(hipStreamSynchronize(s1));
        printf("eager chain of %d x ~5 us kernels: %.3f ms\n", N, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    }
    hipGraph_t g2; hipGraphExec_t e2;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    chain(s1);
    CK(hipStreamEndCapture(s1, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        auto w0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(e2, s1));
        CK(hipStreamSynchronize(s1));
        printf("graph chain of %d x ~5 us kernels: %.3f ms\n", N, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    }
    // the same chain with an external event pair around every fourth kernel (what per-kernel timing inside a graph would add)
    std::vector<hipEvent_t> ev(2 * (N / 4));
    for (auto& e : ev) CK(hipEventCreate(&e));
    hipGraph_t g3; hipGraphExec_t e3;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < N; k++) {
        if (k % 4 == 0) CK(hipEventRecordWithFlags(ev[2 * (k / 4)], s1, hipEventRecordExternal));
        hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(256), 0, s1, d1, 1u, C5);
        if (k % 4 == 0) CK(hipEventRecordWithFlags(ev[2 * (k / 4) + 1], s1, hipEventRecordExternal));
    }
    CK(hipStreamEndCapture(s1, &g3));
    CK(hipGraphInstantiate(&e3, g3, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        auto w0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(e3, s1));
        CK(hipStreamSynchronize(s1));
        float ms = -1;
        hipError_t e = hipEventElapsedTime(&ms, ev[0], ev[1]);
        printf("graph chain with %zu external event records: %.3f ms (first pair: %.4f ms, %s)\n", ev.size(),
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(), ms, hipGetErrorString(e));
    }
    return 0;
}
