// probe_graph_events.hip -- which stream-capture patterns this runtime accepts for external event records (hipEventRecordWithFlags
// .. hipEventRecordExternal): capture mode, events created inside the capture, records on the forked stream, a second edge between
// the streams before the record, the same event pair of an earlier eager use. Prints the result of every call.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess || verbose) printf("  %-74s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
static int verbose = 0, explicit_nodes = 0;
// the same record as an explicit event-record node behind the stream's current capture dependencies (what the library does: the
// HIP runtime a PyTorch wheel bundles, ROCm 7.0, refuses hipEventRecordExternal, the system's 7.2 takes it)
static hipError_t record(hipEvent_t ev, hipStream_t s) {
    if (!explicit_nodes) return hipEventRecordWithFlags(ev, s, hipEventRecordExternal);
    hipStreamCaptureStatus st; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t n = 0;
    hipError_t e = hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &n);
    if (e != hipSuccess) return e;
    if (st != hipStreamCaptureStatusActive) return hipErrorStreamCaptureInvalidated;
    hipGraphNode_t node = nullptr;
    e = hipGraphAddEventRecordNode(&node, g, deps, n, ev);
    if (e != hipSuccess) return e;
    return hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
}
__global__ void k(unsigned* o) { if (threadIdx.x == 0) o[0]++; }
int main(int argc, char** argv) {
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    verbose = argc > 2;
    explicit_nodes = (variant & 32) != 0;
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    unsigned* d; (void)hipMalloc(&d, 256);
    hipEvent_t fork, fork2, order2; (void)hipEventCreateWithFlags(&fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&fork2, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&order2, hipEventDisableTiming);
    printf("variant %d: mode %s, %s first, events created %s, second edge %s, eager use of the order events before %s, records as %s\n", variant,
           variant & 1 ? "relaxed" : "thread-local", variant & 2 ? "memset" : "nothing", variant & 4 ? "inside" : "before", variant & 8 ? "yes" : "no",
           variant & 16 ? "yes" : "no", variant & 32 ? "explicit event-record nodes" : "hipEventRecordExternal");
    if (variant & 16) {           // the order events have an eager history, as a prover's have after its first proof
        CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0)); CK(hipEventRecord(order2, s2)); CK(hipStreamWaitEvent(s1, order2, 0));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    }
    CK(hipStreamBeginCapture(s1, variant & 1 ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
    if (variant & 2) CK(hipMemsetAsync(d, 0, 4, s1));
    CK(hipEventRecord(fork, s1));
    CK(hipStreamWaitEvent(s2, fork, 0));
    hipEvent_t e[4];
    if (!(variant & 4)) for (auto& x : e) CK(hipEventCreate(&x));
    if (variant & 4) { CK(hipEventCreate(&e[0])); CK(hipEventCreate(&e[1])); }
    CK(record(e[0], s1));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d);
    CK(record(e[1], s1));
    if (variant & 8) { CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0)); }
    if (variant & 4) { CK(hipEventCreate(&e[2])); CK(hipEventCreate(&e[3])); }
    CK(record(e[2], s2));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d + 8);
    CK(record(e[3], s2));
    CK(hipEventRecord(order2, s2));
    CK(hipStreamWaitEvent(s1, order2, 0));
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
    CK(hipStreamEndCapture(s1, &g));
    if (g) {
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0)); CK(hipGraphLaunch(ex, s1)); CK(hipStreamSynchronize(s1));
        float a = -1, b = -1; CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[2], e[3]));
        printf("  spans %.4f ms, %.4f ms\n", a, b);
    }
    return 0;
}
