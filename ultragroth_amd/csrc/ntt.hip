// ntt.hip -- radix-2 number-theoretic transform over BN254 Fr for gfx950.
//
// Replaces FFT<Fr>::fft / ifft / root of the reference's un-vendored ffiasm submodule, as used by
// the H-polynomial block of the prover (src/groth16.cpp:110-140; ctor src/groth16.hpp:109):
//     ifft(n) ; x[i] *= root(log2 n + 1, i) ; fft(n)      for each of a, b, c.
// Conventions (pinned by SURVEY.md Appendix A): omega_{2^s} = 5^((r-1)/2^s); ifft scales by 1/n.
//
// Structure: every transform is decimation-in-time -- input in bit-reversed order, output in
// natural order -- done in passes of up to 11 stages. One workgroup stages a tile through LDS
// as 9 limb planes (conflict-free 4-byte accesses), runs its stages with one butterfly per thread
// per stage, and writes the tile back. DIT is used for both directions because its butterfly
// (a + w b, a - w b) adds a freshly reduced product at every stage, so lazily reduced values grow
// by at most 2q per stage and one cheap contraction per pass suffices (ff.hpp, dev_common.hpp).
// Elements live in HBM as 32-byte packed device-Montgomery values; twiddles are precomputed
// tables in HBM (n/2 entries per direction), read through L2.
//
// Algorithmic HBM bytes of one size-n transform: 64 n (read + write once); a transform of
// 2^24 points makes 3 passes, so measured traffic is about 3x that plus twiddles.
#include "dev_common.hpp"
#include "internal.hpp"

namespace ug {

namespace {

constexpr int NTT_MAX_LOG_TILE = 11;      // 2^11 elements * 36 B = 72 KiB of LDS, 1024 threads

struct PassArgs {
    const u32* in;
    u32* out;
    const u32* tw;        // packed twiddles, stage-major: entry (2^s - 1 + j) = omega_{2^(s+1)}^j, j < 2^s
    const u32* post;      // optional: out[i] *= post[i]   (natural index), nullptr if none
    const u32* post_const;// optional: out[i] *= *post_const
    int logn;
    int s0;               // first stage of this pass
    int k;                // stages in this pass
    int j;                // log2(tiles per workgroup)
    int gather_bitrev;    // first pass only: read element idx from in[bitrev(idx)]
    int scatter_bitrev;   // last pass only: write element idx to out[bitrev(idx)]
};

__global__ __launch_bounds__(1024) void ntt_pass_kernel(PassArgs a) {
    extern __shared__ u32 lds[];
    const int E = 1 << (a.k + a.j);               // elements per workgroup
    const int tid = threadIdx.x, nth = blockDim.x;
    const u32 bid = blockIdx.x;
    const int k = a.k, j = a.j, s0 = a.s0;
    const u32 emask = (1u << k) - 1, tmask = (1u << j) - 1;

    // global index of local element `pos`
    u32 hi = 0, lo0 = 0, base = 0;
    if (s0 == 0) base = bid << (k + j);
    else { u32 nlo = 1u << (s0 - j); hi = bid / nlo; lo0 = (bid % nlo) << j; }
    auto gidx = [&](u32 pos) -> u32 {
        if (s0 == 0) return base + pos;
        u32 t = pos & tmask, e = pos >> j;
        return (hi << (s0 + k)) | (e << s0) | (lo0 + t);
    };

    for (u32 pos = tid; pos < (u32)E; pos += nth) {
        u32 g = gidx(pos);
        u32 src = a.gather_bitrev ? bit_reverse(g, a.logn) : g;
        Fr x = ld_packed<FrParams>(a.in + (size_t)src * 8);
#pragma unroll
        for (int l = 0; l < NL; l++) lds[l * E + pos] = x.l[l];
    }
    __syncthreads();

    for (int d = 0; d < k; d++) {
        for (u32 bf = tid; bf < (u32)(E >> 1); bf += nth) {
            u32 pos0, pos1, twi;
            if (s0 == 0) {
                u32 t = bf >> (k - 1), p = bf & ((1u << (k - 1)) - 1);
                u32 elow = p & ((1u << d) - 1);
                u32 e0 = ((p >> d) << (d + 1)) | elow;
                pos0 = (t << k) | e0; pos1 = pos0 + (1u << d);
                twi = ((1u << d) - 1) + elow;
            } else {
                u32 t = bf & tmask, p = bf >> j;
                u32 elow = p & ((1u << d) - 1);
                u32 e0 = ((p >> d) << (d + 1)) | elow;
                pos0 = (e0 << j) | t; pos1 = pos0 + (1u << (d + j));
                twi = ((1u << (s0 + d)) - 1) + ((elow << s0) | (lo0 + t));
            }
            Fr x0, x1;
#pragma unroll
            for (int l = 0; l < NL; l++) { x0.l[l] = lds[l * E + pos0]; x1.l[l] = lds[l * E + pos1]; }
            Fr w = ld_packed<FrParams>(a.tw + (size_t)twi * 8);
            Fr t = mul(x1, w);                       // < 2q for x1 < 169 q
            Fr y0 = add(x0, t);                      // grows by < 2q per stage
            Fr y1 = sub<2>(x0, t);
#pragma unroll
            for (int l = 0; l < NL; l++) { lds[l * E + pos0] = y0.l[l]; lds[l * E + pos1] = y1.l[l]; }
        }
        __syncthreads();
    }

    for (u32 pos = tid; pos < (u32)E; pos += nth) {
        Fr x;
#pragma unroll
        for (int l = 0; l < NL; l++) x.l[l] = lds[l * E + pos];
        u32 g = gidx(pos);
        if (a.post) x = mul(x, ld_packed<FrParams>(a.post + (size_t)g * 8));   // strict, < 2q
        else if (a.post_const) x = mul(x, ld_packed<FrParams>(a.post_const));
        else x = contract(x);                                                   // < 2.01 q, strict
        u32 dst = a.scatter_bitrev ? bit_reverse(g, a.logn) : g;
        st_packed(a.out + (size_t)dst * 8, x);
    }
}

__global__ void bitrev_copy_kernel(u32* out, const u32* in, int logn) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << logn)) return;
    u32 w[8];
    load8(w, in + (size_t)i * 8);
    store8(out + (size_t)bit_reverse(i, logn) * 8, w);
}

// table[i] = scale * base^i for i < count, packed canonical; thread t fills a run of RUN entries
constexpr int POW_RUN = 64;
__global__ void power_table_kernel(u32* table, const u32* base_packed, const u32* scale_packed, u64 count) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 start = t * POW_RUN;
    if (start >= count) return;
    Fr b = ld_packed<FrParams>(base_packed);
    // b^start by square-and-multiply
    Fr acc = ld_packed<FrParams>(scale_packed);
    Fr sq = b;
    for (u64 e = start; e; e >>= 1) {
        if (e & 1) acc = mul(acc, sq);
        sq = sqr(sq);
    }
    for (int i = 0; i < POW_RUN && start + i < count; i++) {
        st_packed(table + (start + i) * 8, cond_sub_q(acc));
        acc = mul(acc, b);
    }
}

// Fr constants on the host, in device form
Fr host_fr_from_u64(u64 v) {
    u32 w[8] = {(u32)v, (u32)(v >> 32), 0, 0, 0, 0, 0, 0};
    return from_normal<FrParams>(w);
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------

// omega_{2^s} = 5^((r-1)/2^s) in device Montgomery form, canonical
Fr fr_root_of_unity(int s) {
    u32 e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::q32[i];
    e[0] -= 1;
    for (int k = 0; k < s; k++)
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? e[i + 1] << 31 : 0);
    return cond_sub_q(pow256(host_fr_from_u64(FrParams::generator), e));
}

void NttPlan::init(int logn_, hipStream_t stream) {
    release();
    logn = logn_;
    // per device, and idempotent: set whenever a plan is made on the current device
    UG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (1 << NTT_MAX_LOG_TILE) * NL * 4));
    u64 n = (u64)1 << logn;
    // Stage-major twiddle tables: stage s (butterfly span 2^s) reads omega_{2^(s+1)}^j at consecutive j, so the
    // lanes of a wave touch consecutive 32-byte entries at every stage (a single table of omega_n^i indexed with a
    // stage-dependent stride makes the late stages hit one L2 channel with 64 KiB strides).
    u64 entries = n > 1 ? n - 1 : 1;
    UG_HIP(hipMalloc(&tw_fwd, entries * 32));
    UG_HIP(hipMalloc(&tw_inv, entries * 32));
    UG_HIP(hipMalloc(&twist, n * 32));
    UG_HIP(hipMalloc(&ninv, 32));
    Fr w2n = fr_root_of_unity(logn + 1);
    Fr n_inverse = cond_sub_q(inv(host_fr_from_u64(n)));
    Fr one = cond_sub_q(fp_one<FrParams>());
    std::vector<u32> consts((size_t)(3 + 2 * (logn + 1)) * 8);
    pack256(consts.data(), w2n); pack256(consts.data() + 8, n_inverse); pack256(consts.data() + 16, one);
    for (int st = 0; st < logn; st++) {
        Fr w = fr_root_of_unity(st + 1);
        pack256(consts.data() + (size_t)(3 + 2 * st) * 8, w);
        pack256(consts.data() + (size_t)(4 + 2 * st) * 8, cond_sub_q(inv(w)));
    }
    u32* d_consts;
    UG_HIP(hipMalloc(&d_consts, consts.size() * 4));
    UG_HIP(hipMemcpyAsync(d_consts, consts.data(), consts.size() * 4, hipMemcpyHostToDevice, stream));
    auto launch = [&](u32* table, int base_i, int scale_i, u64 count) {
        u64 threads = (count + POW_RUN - 1) / POW_RUN;
        unsigned blocks = (unsigned)((threads + 255) / 256);
        hipLaunchKernelGGL(power_table_kernel, dim3(blocks), dim3(256), 0, stream, table, d_consts + 8 * base_i, d_consts + 8 * scale_i, count);
        UG_KERNEL_CHECK();
    };
    for (int st = 0; st < logn; st++) {
        u64 off = ((u64)1 << st) - 1;
        launch(tw_fwd + off * 8, 3 + 2 * st, 2, (u64)1 << st);
        launch(tw_inv + off * 8, 4 + 2 * st, 2, (u64)1 << st);
    }
    launch(twist, 0, 1, n);                    // n^-1 * omega_{2n}^i
    UG_HIP(hipMemcpyAsync(ninv, d_consts + 8, 32, hipMemcpyDeviceToDevice, stream));
    UG_HIP(hipStreamSynchronize(stream));
    UG_HIP(hipFree(d_consts));
}

void NttPlan::release() {
    if (tw_fwd) hipFree(tw_fwd);
    if (tw_inv) hipFree(tw_inv);
    if (twist) hipFree(twist);
    if (ninv) hipFree(ninv);
    tw_fwd = tw_inv = twist = ninv = nullptr;
}

// One DIT transform. `in` holds the input in bit-reversed order unless gather_bitrev is set (then
// natural order, gathered on the fly); output natural order, or bit-reversed if scatter_bitrev.
// post (optional) multiplies output element i (natural index) by post[i] in the last pass.
void NttPlan::transform(u32* out, const u32* in, bool inverse, bool gather_bitrev, bool scatter_bitrev,
                        const u32* post, const u32* post_const, hipStream_t stream, MsmStats* stats) const {
    if (logn == 0) {
        if (out != in) UG_HIP(hipMemcpyAsync(out, in, 32, hipMemcpyDeviceToDevice, stream));
        return;   // size-1 transform is the identity (n^-1 = 1, omega_2^0 = 1)
    }
    if (gather_bitrev && scatter_bitrev) throw std::invalid_argument("ntt: gather and scatter together not supported");
    if ((gather_bitrev || scatter_bitrev) && out == in) throw std::invalid_argument("ntt: permuting transform must be out of place");
    // split the stages: first pass contiguous (up to 11 stages), then strided passes with 2^j tiles
    int stages[8], nj[8], np = 0, rem = logn;
    int first = rem < NTT_MAX_LOG_TILE ? rem : NTT_MAX_LOG_TILE;
    stages[np] = first; nj[np] = 0; np++; rem -= first;
    while (rem > 0) {
        // strided pass: k <= 8 stages and 2^j adjacent tiles, k + j = 11 (runs of 2^j * 32 bytes)
        int npass_left = (rem + 7) / 8;
        int k = (rem + npass_left - 1) / npass_left;
        stages[np] = k; nj[np] = NTT_MAX_LOG_TILE - k; np++; rem -= k;
    }
    // Buffer plan: a scatter pass (last) reads `in`-resident data and writes `out`; otherwise the
    // first pass moves in -> out and the rest run in place on `out`. `in` is clobbered when scattering.
    int s0 = 0;
    for (int p = 0; p < np; p++) {
        PassArgs a;
        bool last = (p == np - 1);
        if (scatter_bitrev) { a.in = in; a.out = last ? out : const_cast<u32*>(in); }
        else { a.in = (p == 0) ? in : out; a.out = out; }
        a.tw = inverse ? tw_inv : tw_fwd; a.logn = logn; a.s0 = s0; a.k = stages[p];
        a.j = nj[p];
        if (s0 > 0 && a.j > s0) a.j = s0;
        a.gather_bitrev = (p == 0 && gather_bitrev) ? 1 : 0;
        a.scatter_bitrev = (last && scatter_bitrev) ? 1 : 0;
        a.post = last ? post : nullptr;
        a.post_const = last ? post_const : nullptr;
        int E = 1 << (a.k + a.j);
        unsigned blocks = (unsigned)(((u64)1 << logn) >> (a.k + a.j));
        int threads = E / 2 > 1024 ? 1024 : (E / 2 < 64 ? 64 : E / 2);
        size_t lds = (size_t)E * NL * 4;
        int slot = stats ? stats->begin(stream, (u64)1 << logn) : -1;
        hipLaunchKernelGGL(ntt_pass_kernel, dim3(blocks), dim3(threads), lds, stream, a);
        UG_KERNEL_CHECK();
        if (stats) stats->end(slot, stream);
        s0 += stages[p];
    }
}

void bitrev_copy(u32* out, const u32* in, int logn, hipStream_t stream) {
    u64 n = (u64)1 << logn;
    hipLaunchKernelGGL(bitrev_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, in, logn);
    UG_KERNEL_CHECK();
}

}  // namespace ug
