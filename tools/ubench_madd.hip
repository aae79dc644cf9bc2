// ubench_madd.hip -- the arithmetic core of the MSM accumulation alone: xyzz_madd on register-resident operands, no memory,
// no branches on data. Compares with the real kernel (segment_accumulate_kernel<G1Cfg>: 201 M additions in 15.3 ms =
// 13.1 G madd/s) to tell how much of its time is the instruction stream and how much is everything around it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iultragroth_amd/csrc tools/ubench_madd.hip -o tools/ubench_madd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec.hpp"
using namespace ug;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <class F, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void madd_chain(u32* data, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int W = sizeof(F) / 4;
    F x, y;
    u32* px = reinterpret_cast<u32*>(&x); u32* py = reinterpret_cast<u32*>(&y);
    for (int i = 0; i < W; i++) { px[i] = data[(size_t)t * 2 * W + i] & (MASK29 >> 4); py[i] = data[(size_t)t * 2 * W + W + i] & (MASK29 >> 4); }
    XYZZ<F> acc = xyzz_from_affine(x, y);
    for (int i = 0; i < iters; i++) {
        acc = xyzz_madd(acc, y, x);           // (not points of the curve: the formulas are straight-line, the work is the same)
        acc = xyzz_madd(acc, x, y);
    }
    u32* pa = reinterpret_cast<u32*>(&acc);
    u32 o = 0;
    for (int i = 0; i < 4 * W; i++) o ^= pa[i];
    data[(size_t)t * 2 * W] = o;
}

template <class F, int WAVES> int run(const char* name, double mads_per_madd) {
    const int blocks = 256 * 4 * WAVES, threads = 256, iters = 600;
    constexpr int W = sizeof(F) / 4;
    u32* d;
    size_t n = (size_t)blocks * threads * 2 * W;
    CK(hipMalloc(&d, n * 4));
    std::vector<u32> h(n);
    unsigned long long s = 777;
    for (size_t i = 0; i < n; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (u32)(s >> 33); }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((madd_chain<F, WAVES>), dim3(blocks), dim3(threads), 0, 0, d, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double madds = (double)blocks * threads * iters * 2.0;
        if (rep == 2) printf("%-28s %d waves/SIMD: %7.1f ms  %6.2f G madd/s  = %5.2f T mad/s (%.0f mads per addition) = %4.1f %% of 29 T mad/s\n",
                             name, WAVES, ms, madds / ms / 1e6, madds * mads_per_madd / ms / 1e9, mads_per_madd, madds * mads_per_madd / ms / 1e9 / 29.0 * 100);
    }
    CK(hipFree(d));
    return 0;
}

int main() {
    if (run<Fq, 3>("G1 xyzz_madd", 1467)) return 1;
    if (run<Fq, 2>("G1 xyzz_madd", 1467)) return 1;
    if (run<Fq, 4>("G1 xyzz_madd", 1467)) return 1;
    if (run<Fq2, 2>("G2 xyzz_madd", 4470)) return 1;
    if (run<Fq2, 1>("G2 xyzz_madd", 4470)) return 1;
    return 0;
}
