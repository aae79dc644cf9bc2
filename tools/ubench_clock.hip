// ubench_clock.hip -- what clock does the chip HOLD under a sustained F29 multiply-add load, and what is the sustained
// v_mad_u64_u32 rate? (tools/ubench_int.hip measures 0.3 ms bursts, which run at the 2.4 GHz maximum: DVFS has not reacted.)
// Method of MI355X_MICROARCH.md ("DVFS give-back", item 6): clock = delta s_memtime / delta s_memrealtime x 100 MHz, stamped
// around the loop in a diagnostic kernel, median over workgroups, after >= 1 s of back-to-back launches on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iultragroth_amd/csrc tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "ff.hpp"
using namespace ug;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// each lane: a chain of Montgomery products on its own random element (162 mads + the reduction's carry work each)
__global__ __launch_bounds__(256) void mul_chain(u32* data, int iters, unsigned long long* stamps) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq a, b;
    for (int i = 0; i < NL; i++) { a.l[i] = data[(size_t)t * 18 + i] & MASK29; b.l[i] = data[(size_t)t * 18 + 9 + i] & MASK29; }
    a.l[NL - 1] &= 0xfffff; b.l[NL - 1] &= 0xfffff;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) { a = mul(a, b); b = mul(b, a); }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < NL; i++) data[(size_t)t * 18 + i] = a.l[i] ^ b.l[i];
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int blocks = 256 * 12, threads = 256;          // 12 workgroups of 4 waves per CU: 3 waves per SIMD resident, 4 rounds
    const int iters = 20000;
    u32* d; unsigned long long* st;
    size_t n = (size_t)blocks * threads * 18;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&st, blocks * 16));
    std::vector<u32> h(n);
    unsigned long long s = 12345;
    for (size_t i = 0; i < n; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (u32)(s >> 33); }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mul_chain, dim3(blocks), dim3(threads), 0, 0, d, iters, st);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> hs(2 * blocks);
        CK(hipMemcpy(hs.data(), st, blocks * 16, hipMemcpyDeviceToHost));
        std::vector<double> clk(blocks);
        for (int b = 0; b < blocks; b++) clk[b] = (double)hs[2 * b] / (double)hs[2 * b + 1] * 100.0;       // MHz
        std::sort(clk.begin(), clk.end());
        double muls = (double)blocks * threads * iters * 2.0;
        printf("run %d: %.1f ms  %.1f G modmul/s  %.2f T mad/s (162 per product)  in-kernel clock median %.0f MHz (min %.0f, max %.0f)\n",
               rep, ms, muls / ms / 1e6, muls * 162.0 / ms / 1e9, clk[blocks / 2], clk.front(), clk.back());
    }
    return 0;
}
