// tools/ubench_int.hip -- integer-issue microbenchmarks for gfx950 (MI355X).
// Measures the sustained chip-wide rate of the instructions a 254-bit Montgomery product is
// made of, so that kernel reports can quote "mads/s vs measured peak" (SURVEY.md section 8d).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o tools/ubench_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define ITERS 4096

// 8 independent chains per lane, each a dependent sequence of the instruction under test
__global__ void k_mad64(u64* out, u32 a, u32 b) {
    u64 c[8]; u32 x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = (u64)(u32)c[j] * y + c[j] + x;   // v_mad_u64_u32 + add
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64_pure(u64* out, u32 a, u32 b) {
    u64 c[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[j]) : "v"(y), "v"((u32)(j + 3)) : "vcc");
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_addc(u64* out, u32 a, u32 b) {
    u64 c[8]; u32 h[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) { c[j] = j + threadIdx.x + a; h[j] = j; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                         : "+v"(c[j]), "+v"(h[j]) : "v"(y), "v"((u32)(j + 3)) : "vcc");
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j] + h[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(u64* out, u32 a, u32 b) {
    u32 c[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[j]) : "v"(y));
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(u64* out, u32 a, u32 b) {
    u32 c[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c[j]) : "v"(y));
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(u64* out, u32 a, u32 b) {
    u32 c[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[j]) : "v"(y));
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addc(u64* out, u32 a, u32 b) {
    u32 c[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(c[j]) : "v"(y) : "vcc");
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshladd64(u64* out, u32 a, u32 b) {
    u64 c[8]; u64 y = ((u64)b << 20) ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[j]) : "v"(y));
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(u64* out, u32 a, u32 b) {
    u32 c[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c[j]) : "v"(y));
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dfma(u64* out, u32 a, u32 b) {
    double c[8]; double y = 1.0 + 1e-9 * (b ^ threadIdx.x);
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(c[j]) : "v"(y));
    }
    double s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)s;
}
__global__ void k_fma32(u64* out, u32 a, u32 b) {
    float c[8]; float y = 1.0f + 1e-7f * (b ^ threadIdx.x);
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(c[j]) : "v"(y));
    }
    float s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)s;
}


#define UB_KERNEL32(NAME, ASM)                                                              \
__global__ void NAME(u64* out, u32 a, u32 b) {                                              \
    u32 c[8]; u32 y = b ^ threadIdx.x;                                                      \
    for (int j = 0; j < 8; j++) c[j] = j + threadIdx.x + a;                                 \
    for (int i = 0; i < ITERS; i++) {                                                       \
        _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(ASM : "+v"(c[j]) : "v"(y)); \
    }                                                                                       \
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
}
#define UB_KERNEL64(NAME, ASM)                                                              \
__global__ void NAME(u64* out, u32 a, u32 b) {                                              \
    u64 c[8]; u32 y = (b ^ threadIdx.x) & 31;                                               \
    for (int j = 0; j < 8; j++) c[j] = ((u64)(j + threadIdx.x + a) << 33) | 12345;          \
    for (int i = 0; i < ITERS; i++) {                                                       \
        _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(ASM : "+v"(c[j]) : "v"(y)); \
    }                                                                                       \
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j];                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
}
UB_KERNEL32(k_and32, "v_and_b32 %0, %0, %1")
UB_KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
UB_KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %1")
UB_KERNEL32(k_lshladd32, "v_lshl_add_u32 %0, %0, 3, %1")
UB_KERNEL32(k_lshr32, "v_lshrrev_b32 %0, 3, %0")
UB_KERNEL32(k_or3, "v_or3_b32 %0, %0, %1, %1")
UB_KERNEL32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
UB_KERNEL32(k_sub32, "v_sub_u32 %0, %0, %1")
UB_KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 3, %0")
UB_KERNEL64(k_ashr64, "v_ashrrev_i64 %0, 3, %0")
UB_KERNEL64(k_lshl64, "v_lshlrev_b64 %0, 1, %0")
// one mad followed by N independent adds: do the adds hide under the mad?
template <int NADD>
__global__ void k_mad_plus_adds(u64* out, u32 a, u32 b) {
    u64 c[8]; u32 d[8]; u32 y = b ^ threadIdx.x;
    for (int j = 0; j < 8; j++) { c[j] = j + threadIdx.x + a; d[j] = j * 7 + a; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[j]) : "v"(y), "v"((u32)(j + 3)) : "vcc");
#pragma unroll
            for (int k = 0; k < NADD; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d[(j + k) & 7]) : "v"(y));
        }
    }
    u64 s = 0; for (int j = 0; j < 8; j++) s += c[j] + d[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static int run(const char* name, K kern, double ops_per_thread, int blocks, int threads, u64* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 6789u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 6789u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double total = ops_per_thread * (double)blocks * threads;
    double rate = total / (best * 1e-3);
    // lane-ops per clock per CU at 2.4 GHz nominal, and cycles per wave64 instruction per SIMD
    double per_clk_cu = rate / 256.0 / 2.4e9;
    printf("%-14s blocks=%5d thr=%4d  %8.3f ms  %9.2f Gop/s  %6.1f lane-op/clk/CU  => %5.2f cyc/wave-instr/SIMD (at 2.4GHz)\n",
           name, blocks, threads, best, rate * 1e-9, per_clk_cu, 64.0 * 4.0 / per_clk_cu);
    return 0;
}

int main() {
    u64* out; CK(hipMalloc(&out, sizeof(u64) * 256 * 32 * 1024));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    double ops = (double)ITERS * 8;
    for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
        int threads = 256, blocks = 256 * wavesPerSimd * 4;   // 4 rounds of full occupancy at this level
        printf("--- %d wave(s)/SIMD resident (launch_bounds not forced; blocks=%d)\n", wavesPerSimd, blocks);
        if (run("mad_u64_u32+add", k_mad64, ops, blocks, threads, out)) return 1;
        if (run("mad_u64_u32", k_mad64_pure, ops, blocks, threads, out)) return 1;
        if (run("mad+addc pair", k_mad_addc, ops, blocks, threads, out)) return 1;
        if (run("mul_lo_u32", k_mullo, ops, blocks, threads, out)) return 1;
        if (run("mul_hi_u32", k_mulhi, ops, blocks, threads, out)) return 1;
        if (run("add_u32", k_add32, ops, blocks, threads, out)) return 1;
        if (run("addc_co_u32", k_addc, ops, blocks, threads, out)) return 1;
        if (run("lshl_add_u64", k_lshladd64, ops, blocks, threads, out)) return 1;
        if (run("mad_u32_u24", k_mad24, ops, blocks, threads, out)) return 1;
        if (run("fma_f64", k_dfma, ops, blocks, threads, out)) return 1;
        if (run("fma_f32", k_fma32, ops, blocks, threads, out)) return 1;
        if (run("and_b32", k_and32, ops, blocks, threads, out)) return 1;
        if (run("alignbit_b32", k_alignbit, ops, blocks, threads, out)) return 1;
        if (run("add3_u32", k_add3, ops, blocks, threads, out)) return 1;
        if (run("lshl_add_u32", k_lshladd32, ops, blocks, threads, out)) return 1;
        if (run("lshrrev_b32", k_lshr32, ops, blocks, threads, out)) return 1;
        if (run("or3_b32", k_or3, ops, blocks, threads, out)) return 1;
        if (run("add_co_u32", k_addco, ops, blocks, threads, out)) return 1;
        if (run("sub_u32", k_sub32, ops, blocks, threads, out)) return 1;
        if (run("lshrrev_b64", k_lshr64, ops, blocks, threads, out)) return 1;
        if (run("ashrrev_i64", k_ashr64, ops, blocks, threads, out)) return 1;
        if (run("lshlrev_b64", k_lshl64, ops, blocks, threads, out)) return 1;
        printf("--- one v_mad_u64_u32 followed by N independent v_add_u32 (rate counted per mad)\n");
        if (run("mad+0add", k_mad_plus_adds<0>, ops, blocks, threads, out)) return 1;
        if (run("mad+1add", k_mad_plus_adds<1>, ops, blocks, threads, out)) return 1;
        if (run("mad+2add", k_mad_plus_adds<2>, ops, blocks, threads, out)) return 1;
        if (run("mad+4add", k_mad_plus_adds<4>, ops, blocks, threads, out)) return 1;
        break;  // occupancy is set by the hardware (8 waves/SIMD at these register counts)
    }
    CK(hipFree(out));
    return 0;
}
