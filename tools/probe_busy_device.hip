// probe_busy_device.hip -- what a host thread can do on this runtime while a long kernel (a window-table build) occupies the device:
// does hipMalloc return, do copies and short kernels on other streams (of which priority class) get through, does hipFree wait?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void busy(unsigned* o, long long ticks) {          // every workgroup spins: grid >> chip, so the kernel holds every CU for its whole duration
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) o[0]++;
}
__global__ void tiny(unsigned* o) { if (threadIdx.x == 0) o[0]++; }
int main(int argc, char** argv) {
    const int busy_prio = argc > 1 ? atoi(argv[1]) : 0;       // -1 lowest, 0 normal
    int least, greatest; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sb, sl, sn;
    if (busy_prio < 0) CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, least)); else CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sl, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    unsigned *d, *d2; CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&d2, 256 << 20));
    char* pinned; CK(hipHostMalloc((void**)&pinned, 64 << 20, hipHostMallocDefault));
    void* early; CK(hipMalloc(&early, 1 << 20));
    // ~1 ms per workgroup, 256 CUs x 8 workgroups resident -> 2048 at a time; 400 rounds ~ 0.4 s
    hipLaunchKernelGGL(busy, dim3(2048 * 400), dim3(256), 0, sb, d, 100000LL);
    const double t0 = now();
    printf("busy kernel on a %s-priority stream queued; then, from the host:\n", busy_prio < 0 ? "lowest" : "normal");
    void* big = nullptr; CK(hipMalloc(&big, (size_t)8 << 30));
    printf("  hipMalloc(8 GiB) returned after           %8.2f ms\n", now() - t0);
    double t = now(); CK(hipMemcpyAsync(d2, pinned, 64 << 20, hipMemcpyHostToDevice, sl)); CK(hipStreamSynchronize(sl));
    printf("  64 MiB H2D on a lowest-priority stream     %8.2f ms (done %8.2f ms after the start)\n", now() - t, now() - t0);
    t = now(); CK(hipMemcpyAsync(d2, pinned, 64 << 20, hipMemcpyHostToDevice, sn)); CK(hipStreamSynchronize(sn));
    printf("  64 MiB H2D on a normal-priority stream     %8.2f ms (done %8.2f)\n", now() - t, now() - t0);
    t = now(); hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, sl, d + 64); CK(hipStreamSynchronize(sl));
    printf("  tiny kernel on a lowest-priority stream    %8.2f ms (done %8.2f)\n", now() - t, now() - t0);
    t = now(); hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, sn, d + 128); CK(hipStreamSynchronize(sn));
    printf("  tiny kernel on a normal-priority stream    %8.2f ms (done %8.2f)\n", now() - t, now() - t0);
    t = now(); CK(hipFree(early));
    printf("  hipFree of an unrelated 1 MiB buffer       %8.2f ms (done %8.2f)\n", now() - t, now() - t0);
    CK(hipStreamSynchronize(sb));
    printf("  busy kernel finished                       %8.2f ms after the start\n", now() - t0);
    return 0;
}
