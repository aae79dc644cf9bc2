// ubench_fma52.hip -- a Montgomery product on 5 x 52-bit limbs held as doubles, partial products by pairs of v_fma_f64
// in round-toward-zero mode (Emmart, Zheng, Weems: hi = fma(a, b, 2^104), lo = fma(a, b, 2^104 + 2^52 - hi); the bit
// patterns of hi and lo are summed as 64-bit integers), against the F29 product of ff.hpp (9 x 29-bit limbs on
// v_mad_u64_u32). VERDICT r1 item 4 asked for this measurement: "adopt only if the microbenchmark and the G1 kernel both win".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iultragroth_amd/csrc tools/ubench_fma52.hip -o tools/ubench_fma52
// Both products are checked against each other first (the two Montgomery radices differ by one bit: R52 = 2^260,
// R29 = 2^261, so mont52(a, b) = 2 * mont29(a, b) mod q), then timed as dependent chains, CHAINS of them per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "ff.hpp"
using namespace ug;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef unsigned long long ull;
constexpr ull E52 = (ull)(1023 + 52) << 52;         // bit pattern of 2^52
constexpr ull E104 = (ull)(1023 + 104) << 52;       // bit pattern of 2^104
constexpr ull M52 = ((ull)1 << 52) - 1;

struct F52 { double l[5]; };                        // exact integers in [0, 2^52)
struct Mod52 { double q[5]; double qp; };           // modulus limbs, -q^-1 mod 2^52

__device__ __forceinline__ double as_d(ull b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ ull as_u(double d) { return (ull)__double_as_longlong(d); }
// Every f64 operation of the product is inline assembly: the compiler keeps the rounding mode of the instructions IT emits
// at round-to-nearest (it puts a mode switch back after any s_setreg it sees), so v_fma_f64 / v_add_f64 written in C++
// would never run in the round-toward-zero mode the hi / lo trick needs.
__device__ __forceinline__ double fma_asm(double a, double b, double c) {
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double sub_asm(double a, double b) {
    double d;
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double limb_to_double(ull limb) { return sub_asm(as_d(limb | E52), 0x1p52); }      // exact for limb < 2^52

// col[k] += lo pattern, col[k + 1] += hi pattern of x * y (x, y exact integers below 2^52; rounding mode RZ)
__device__ __forceinline__ void pp(ull* col, int k, double x, double y) {
    const double hi = fma_asm(x, y, 0x1p104);                           // 2^104 + floor(xy / 2^52) * 2^52
    const double lo = fma_asm(x, y, sub_asm(0x1p104 + 0x1p52, hi));     // 2^52 + (xy mod 2^52)
    col[k] += as_u(lo);
    col[k + 1] += as_u(hi);
}

__device__ __forceinline__ void set_round_toward_zero_f64() {
    // MODE[3:2] = rounding of f64 / f16 operations: 3 = toward zero     hwreg(HW_REG_MODE = 1, offset 2, size 2)
    // (set at every product: the compiler's own mode bookkeeping assumes round-to-nearest at function entry and puts that
    // back after any instruction it had to switch the mode for -- one scalar instruction per product is the price)
    __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);
}

__device__ __forceinline__ F52 mont52(const F52& a, const F52& b, const Mod52& m) {
    set_round_toward_zero_f64();
    ull col[11];
    int nlo[11], nhi[11];                           // how many lo / hi patterns each column holds (constants after unrolling)
#pragma unroll
    for (int k = 0; k < 11; k++) { col[k] = 0; nlo[k] = 0; nhi[k] = 0; }
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) { pp(col, i + j, a.l[i], b.l[j]); nlo[i + j]++; nhi[i + j + 1]++; }
    ull carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        // true value of column i so far, its low 52 bits, the Montgomery digit
        const ull t = col[i] - (ull)nlo[i] * E52 - (ull)nhi[i] * E104 + carry;
        const double td = limb_to_double(t & M52);
        const double h = fma_asm(td, m.qp, 0x1p104);
        const double md = sub_asm(fma_asm(td, m.qp, sub_asm(0x1p104 + 0x1p52, h)), 0x1p52);      // (t * q') mod 2^52
        ull before = col[i];
#pragma unroll
        for (int j = 0; j < 5; j++) { pp(col, i + j, md, m.q[j]); nlo[i + j]++; nhi[i + j + 1]++; }
        // column i is now a multiple of 2^52: its carry goes on
        const ull lo_mq0 = col[i] - before - E52;                        // the one lo pattern this round added to column i
        carry = (t + lo_mq0) >> 52;
    }
    F52 r;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const ull t = col[5 + k] - (ull)nlo[5 + k] * E52 - (ull)nhi[5 + k] * E104 + carry;
        r.l[k] = limb_to_double(t & M52);
        carry = t >> 52;
    }
    return r;                                       // < 2q for inputs < 2q (R = 2^260 > 4q)
}

__device__ __forceinline__ F52 words_to_f52(const u32* w) {        // 256-bit little-endian integer -> 5 x 52-bit limbs
    ull v[4];
    for (int i = 0; i < 4; i++) v[i] = (ull)w[2 * i] | ((ull)w[2 * i + 1] << 32);
    F52 r;
    for (int k = 0; k < 5; k++) {
        int bit = 52 * k, word = bit >> 6, sh = bit & 63;
        ull x = v[word] >> sh;
        if (sh > 12 && word + 1 < 4) x |= v[word + 1] << (64 - sh);
        r.l[k] = limb_to_double(x & M52);
    }
    return r;
}
__device__ __forceinline__ void f52_to_words(u32* w, const F52& a) {
    ull v[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
        ull x = as_u(sub_asm(a.l[k], -0x1p52)) & M52;       // (exact: the limb is an integer below 2^52)
        int bit = 52 * k, word = bit >> 6, sh = bit & 63;
        v[word] |= x << sh;
        if (sh > 12) v[word + 1] |= x >> (64 - sh);
    }
    for (int i = 0; i < 4; i++) { w[2 * i] = (u32)v[i]; w[2 * i + 1] = (u32)(v[i] >> 32); }
}

__global__ void check_kernel(const u32* a_words, const u32* b_words, Mod52 m, u32* out52, u32* out29, int n) {
    set_round_toward_zero_f64();
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    F52 r = mont52(words_to_f52(a_words + 8 * t), words_to_f52(b_words + 8 * t), m);
    f52_to_words(out52 + 8 * t, r);
    Fq x = unpack256<FqParams>(a_words + 8 * t), y = unpack256<FqParams>(b_words + 8 * t);
    pack256(out29 + 8 * t, cond_sub_q(mul(x, y)));
}

template <int CHAINS> __global__ void chain52_kernel(Mod52 m, double* out, int iters) {
    set_round_toward_zero_f64();
    F52 x[CHAINS], y[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int k = 0; k < 5; k++) { x[c].l[k] = (double)((threadIdx.x * 977 + c * 131 + k * 7 + 3) & 0xFFFFF); y[c].l[k] = (double)((blockIdx.x * 61 + c * 17 + k + 5) & 0xFFFFF); }
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int c = 0; c < CHAINS; c++) { x[c] = mont52(x[c], y[c], m); y[c] = mont52(y[c], x[c], m); }
    double acc = 0;
    for (int c = 0; c < CHAINS; c++) for (int k = 0; k < 5; k++) acc += x[c].l[k] + y[c].l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int CHAINS> __global__ void chain29_kernel(u32* out, int iters) {
    Fq x[CHAINS], y[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int k = 0; k < NL; k++) { x[c].l[k] = (threadIdx.x * 977 + c * 131 + k * 7 + 3) & 0xFFFFF; y[c].l[k] = (blockIdx.x * 61 + c * 17 + k + 5) & 0xFFFFF; }
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int c = 0; c < CHAINS; c++) { x[c] = mul(x[c], y[c]); y[c] = mul(y[c], x[c]); }
    u32 acc = 0;
    for (int c = 0; c < CHAINS; c++) for (int k = 0; k < NL; k++) acc ^= x[c].l[k] ^ y[c].l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void host_mod52(Mod52& m) {
    // q as 256-bit words (ff.hpp: FqParams::q32), limbs of 52 bits, q' = -q^-1 mod 2^52 by Newton iteration
    ull v[4];
    for (int i = 0; i < 4; i++) v[i] = (ull)FqParams::q32[2 * i] | ((ull)FqParams::q32[2 * i + 1] << 32);
    ull limb[5];
    for (int k = 0; k < 5; k++) {
        int bit = 52 * k, word = bit >> 6, sh = bit & 63;
        ull x = v[word] >> sh;
        if (sh > 12 && word + 1 < 4) x |= v[word + 1] << (64 - sh);
        limb[k] = x & M52; m.q[k] = (double)limb[k];
    }
    ull inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - limb[0] * inv;              // q^-1 mod 2^64
    m.qp = (double)((0 - inv) & M52);
}

template <int CHAINS> static int timed(const Mod52& m, int blocks, int iters) {
    double* o52; u32* o29;
    CK(hipMalloc(&o52, (size_t)blocks * 256 * 8)); CK(hipMalloc(&o29, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms52 = 0, ms29 = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain52_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, m, o52, iters); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms52, e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain29_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, o29, iters); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms29, e0, e1));
    }
    const double n = (double)blocks * 256 * iters * CHAINS * 2;
    printf("%6d blocks x 256 lanes, %d chains per lane:   fma52 %8.2f G products/s   F29 %8.2f G products/s   ratio %.2f\n", blocks, CHAINS,
           n / (ms52 * 1e-3) / 1e9, n / (ms29 * 1e-3) / 1e9, ms29 / ms52);
    CK(hipFree(o52)); CK(hipFree(o29));
    return 0;
}

int main() {
    Mod52 m;
    host_mod52(m);
    // ---- agreement -------------------------------------------------------------------------------------------
    const int n = 4096;
    std::vector<u32> a(8 * n), b(8 * n);
    ull s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (u32)(s >> 16); };
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 8; k++) { a[8 * i + k] = rnd(); b[8 * i + k] = rnd(); }
        a[8 * i + 7] &= 0x1FFFFFFF; b[8 * i + 7] &= 0x1FFFFFFF;      // < 2^253 < q
        if (i == 0) for (int k = 0; k < 8; k++) a[k] = 0;
        if (i == 1) for (int k = 0; k < 8; k++) { a[8 + k] = FqParams::q32[k]; b[8 + k] = FqParams::q32[k]; }   // q * q (top of the lazy range)
        if (i == 1) { a[8] -= 1; b[8] -= 1; }
    }
    u32 *da, *db, *d52, *d29;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&d52, a.size() * 4)); CK(hipMalloc(&d29, a.size() * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, m, d52, d29, n);
    std::vector<u32> r52(8 * n), r29(8 * n);
    CK(hipMemcpy(r52.data(), d52, r52.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r29.data(), d29, r29.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        Fq x = unpack256<FqParams>(&r29[8 * i]);
        u32 twice[8], got[8];
        pack256(twice, cond_sub_q(canon(add(x, x))));
        pack256(got, cond_sub_q(canon(unpack256<FqParams>(&r52[8 * i]))));
        if (memcmp(twice, got, 32)) { if (bad < 3) printf("mismatch at %d\n", i); bad++; }
    }
    printf("agreement of %d products (mont52 = 2 * mont29 mod q): %s\n", n, bad ? "FAILED" : "OK");
    if (bad) return 1;
    // ---- throughput -------------------------------------------------------------------------------------------
    const int iters = 400;
    timed<1>(m, 256, iters); timed<1>(m, 512, iters); timed<1>(m, 1024, iters); timed<1>(m, 2048, iters);
    timed<2>(m, 512, iters); timed<2>(m, 1024, iters); timed<4>(m, 512, iters); timed<4>(m, 2048, iters);
    return 0;
}
