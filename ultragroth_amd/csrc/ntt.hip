// ntt.hip -- radix-2 number-theoretic transform over BN254 Fr for gfx950.
//
// Replaces FFT<Fr>::fft / ifft / root of the reference's un-vendored ffiasm submodule, as used by
// the H-polynomial block of the prover (src/groth16.cpp:110-140; ctor src/groth16.hpp:109):
//     ifft(n) ; x[i] *= root(log2 n + 1, i) ; fft(n)      for each of a, b, c.
// Conventions (pinned by SURVEY.md Appendix A): omega_{2^s} = 5^((r-1)/2^s); ifft scales by 1/n.
//
// Structure: every transform is decimation-in-time -- input in bit-reversed order, output in
// natural order -- done in passes of up to 10 stages. One workgroup stages a tile through LDS
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

// 2^10 elements * 36 B = 36 KiB of LDS: four workgroups per CU, each in its own phase (load | butterfly steps | store), keep
// memory and vector issue busy together: 0.79 ms per pass at 2^24 against 0.84 with 2^11-element tiles (two workgroups per
// CU) and 1.01 with 2^12 (one); 2^9 measures the same as 2^10. 2^24 = 10 + 7 + 7 stages, still three passes.
constexpr int NTT_MAX_LOG_TILE = 10;
constexpr int NTT_THREADS = 256;          // one radix-4 butterfly per lane per step at the full tile

// Twiddles live in HBM UNPACKED: 9 limbs of 29 bits in 12 words (48 bytes: three aligned 16-byte loads, no shifts or masks at
// the point of use -- a packed 32-byte entry cost ~26 vector instructions to unpack, three times per radix-4 step).
constexpr int TW_WORDS = 12;
__device__ __forceinline__ Fr ld_twiddle(const u32* tw, size_t entry) {
    const uint4* q = reinterpret_cast<const uint4*>(tw + entry * TW_WORDS);
    const uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = tw[entry * TW_WORDS + 8];
    return r;
}

__device__ __forceinline__ Fr as_fr(const u32* x) { return fp_from<FrParams>(x); }
// Shoup form of a twiddle (UG_NTT_SHOUP, the default since round 5): the PLAIN constant w beside wq = floor(w 2^261 / q), 20 words
// (80 bytes: five aligned 16-byte loads): [w0..w7 | wq0..wq7 | w8 wq8 0 0]. x * w mod q then costs 143 multiply-adds instead of
// the Montgomery product's 162 + 9 (ff.hpp: mul_shoup) and the data keep their Montgomery form.
constexpr int TWS_WORDS = 20;
struct ShoupTw { u32 w[NL], wq[NL]; };
__device__ __forceinline__ ShoupTw ld_twiddle_shoup(const u32* tw, size_t entry) {
    const uint4* q = reinterpret_cast<const uint4*>(tw + entry * TWS_WORDS);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    ShoupTw r;
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w; r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
    r.wq[0] = c.x; r.wq[1] = c.y; r.wq[2] = c.z; r.wq[3] = c.w; r.wq[4] = d.x; r.wq[5] = d.y; r.wq[6] = d.z; r.wq[7] = d.w;
    r.w[8] = e.x; r.wq[8] = e.y;
    return r;
}
// the twiddle product of a butterfly in either form: result strict, < 2q (Montgomery) / < 3q (Shoup)
template <bool SHOUP> struct Twiddle;
template <> struct Twiddle<false> {
    Fr w;
    __device__ __forceinline__ static Twiddle load(const u32* tw, size_t entry) { Twiddle t; t.w = ld_twiddle(tw, entry); return t; }
    __device__ __forceinline__ Fr times(const u32* x) const { return mul(as_fr(x), w); }
};
template <> struct Twiddle<true> {
    ShoupTw t;
    __device__ __forceinline__ static Twiddle load(const u32* tw, size_t entry) { Twiddle r; r.t = ld_twiddle_shoup(tw, entry); return r; }
    __device__ __forceinline__ Fr times(const u32* x) const { return mul_shoup<FrParams>(x, t.w, t.wq); }
};

struct PassArgs {
    const u32* in;
    u32* out;
    const u32* tw;        // unpacked twiddles (TW_WORDS each), stage-major: entry (2^s - 1 + j) = omega_{2^(s+1)}^j, j < 2^s
    const u32* post;      // optional: out[i] *= post[i]   (natural index), nullptr if none
    const u32* post_const;// optional: out[i] *= *post_const
    const u32* in2;       // optional, first pass: the input element is in[i] * in2[i]
    const u32* fin_a;     // optional, last pass: out[i] = plain(fin_a[i] * fin_b[i] - x[i]) instead of x[i]
    const u32* fin_b;
    int logn;
    int s0;               // first stage of this pass
    int k;                // stages in this pass
    int j;                // log2(tiles per workgroup)
    int gather_bitrev;    // first pass only: read element idx from in[bitrev(idx)]
    int scatter_bitrev;   // last pass only: write element idx to out[bitrev(idx)]
#ifdef UG_MEASURE
    int fuse_steps;       // UG_NTT_FUSE_STEPS=1 (WRONG results, timing only): two radix-4 steps per LDS round trip -- see the kernel
#endif
};

// Raw limb-wise sum / difference WITHOUT the carry pass (limbs may exceed 29 bits): the radix-4 step below feeds them
// straight into a product or into the next sum. sub adds the 2q table padded by 2 * 2^29 per limb (ff.hpp: sub<K>), so no
// limb goes negative for a subtrahend with strict limbs.
__device__ __forceinline__ void raw_add(u32* r, const u32* a, const Fr& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = a[i] + b.l[i];
}
__device__ __forceinline__ void raw_sub2q(u32* r, const u32* a, const Fr& b) {
    const u32* kq = kq_padded<FrParams, 2>();
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = a[i] + kq[i] - b.l[i];
}
// a + Kq - b for a product b below K q (K = 2 for Montgomery products, 3 for Shoup products)
template <int K> __device__ __forceinline__ void raw_subkq(u32* r, const u32* a, const Fr& b) {
    const u32* kq = kq_padded<FrParams, K>();
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = a[i] + kq[i] - b.l[i];
}

// One pass = k consecutive DIT stages on a tile staged in LDS (9 limb planes, conflict-free 4-byte accesses). Stages are
// taken two at a time as radix-4 steps held in registers: 4 elements and 3 twiddles in, 4 products, 4 elements out per
// lane -- half the LDS round trips and address arithmetic of radix-2 steps, and only ONE carry pass per element per two
// stages: the first stage's sums stay un-normalised (limbs < 2^31 + 16, which the column sums of a product with a strict
// twiddle still hold: 9 * 2^60 + 9 * 2^58 < 2^64). An odd k starts with one radix-2 step. Bounds in units of q: every
// stage adds a product or its negation + K q -- Montgomery products are below 2q (K = 2), Shoup products below 3q (K = 3) --
// so a pass of 10 stages takes a packed input (< 2^256 < 5.3 q) to below 36 q (Shoup; 26 q Montgomery): under the 64 q the
// contraction of the last loop takes (to < 2.01 q for packing) and the 170 q a Shoup product's operand may have.
// up to three transforms of the same size in one launch (blockIdx.y picks one): the three chains of the H-polynomial block
struct PassBatch { PassArgs a[3]; };

template <bool SHOUP>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(PassBatch batch) {
    constexpr int KQ = SHOUP ? 3 : 2;             // the twiddle products are below KQ q
    extern __shared__ u32 lds[];
    const PassArgs& a = batch.a[blockIdx.y];
    const int E = 1 << (a.k + a.j);               // elements per workgroup
    const int tid = threadIdx.x, nth = blockDim.x;
    const u32 bid = blockIdx.x;
    const int k = a.k, j = a.j, s0 = a.s0;
    const u32 tmask = (1u << j) - 1;

    // global index of local element `pos`
    u32 hi = 0, lo0 = 0, base = 0;
    if (s0 == 0) base = bid << (k + j);
    else { u32 nlo = 1u << (s0 - j); hi = bid / nlo; lo0 = (bid % nlo) << j; }
    auto gidx = [&](u32 pos) -> u32 {
        if (s0 == 0) return base + pos;
        u32 t = pos & tmask, e = pos >> j;
        return (hi << (s0 + k)) | (e << s0) | (lo0 + t);
    };
    // local position of stage index e in tile t, and the twiddle-table entry of (stage s0 + d, local exponent elow)
    auto lpos = [&](u32 e, u32 t) -> u32 { return s0 == 0 ? (t << k) | e : (e << j) | t; };
    auto twidx = [&](int d, u32 elow, u32 t) -> u32 {
        return s0 == 0 ? ((1u << d) - 1) + elow : ((1u << (s0 + d)) - 1) + ((elow << s0) | (lo0 + t));
    };
    // (a bank swizzle of the positions was measured: the early steps' 2- to 4-way conflicts cost nothing visible, the
    // kernel is bound by vector issue -- 162 multiply-adds and ~190 other vector instructions per butterfly)
    auto ld = [&](u32* x, u32 pos) {
#pragma unroll
        for (int l = 0; l < NL; l++) x[l] = lds[l * E + pos];
    };
    auto st = [&](u32 pos, const Fr& y) {
#pragma unroll
        for (int l = 0; l < NL; l++) lds[l * E + pos] = y.l[l];
    };

    for (u32 pos = tid; pos < (u32)E; pos += nth) {
        u32 g = gidx(pos);
        u32 src = a.gather_bitrev ? bit_reverse(g, a.logn) : g;
        Fr x = ld_packed<FrParams>(a.in + (size_t)src * 8);
        if (a.in2) x = mul(x, ld_packed<FrParams>(a.in2 + (size_t)src * 8));      // fused c = a o b (first pass of chain c)
        st(pos, x);
    }
    __syncthreads();

    int d = 0;
    if (k & 1) {                                  // odd stage count: one radix-2 step first
        for (u32 bf = tid; bf < (u32)(E >> 1); bf += nth) {
            u32 t, p;
            if (s0 == 0) { t = bf >> (k - 1); p = bf & ((1u << (k - 1)) - 1); } else { t = bf & tmask; p = bf >> j; }
            const u32 e0 = p << 1;                // d = 0: elow = 0
            const u32 pos0 = lpos(e0, t), pos1 = lpos(e0 + 1, t);
            u32 x0[NL], x1[NL], y[NL];
            ld(x0, pos0); ld(x1, pos1);
            const Twiddle<SHOUP> w = Twiddle<SHOUP>::load(a.tw, twidx(0, 0, t));
            Fr tt = w.times(x1);
            raw_add(y, x0, tt); st(pos0, norm_weak<FrParams>(y));
            raw_subkq<KQ>(y, x0, tt); st(pos1, norm_weak<FrParams>(y));
        }
        __syncthreads();
        d = 1;
    }
#ifdef UG_MEASURE
    // Measurement only (round 4, VERDICT item 6: what would a radix-8 / radix-16 step buy?): two radix-4 steps on ONE set of
    // registers per LDS round trip -- the second step takes the first one's outputs as if they were its own four elements (they
    // are not: the results are WRONG) -- i.e. the instruction stream of four stages with half the LDS loads, stores, address
    // arithmetic and barriers and the same products, carry passes and twiddle loads: an upper bound of what ANY higher radix
    // can save (a real radix-16 step would also need 16 elements per lane in registers).
    if (!SHOUP && a.fuse_steps) {
        for (; d + 3 < k; d += 4) {
            for (u32 bf = tid; bf < (u32)(E >> 2); bf += nth) {
                u32 t, p;
                if (s0 == 0) { t = bf >> (k - 2); p = bf & ((1u << (k - 2)) - 1); } else { t = bf & tmask; p = bf >> j; }
                u32 x0[NL], x1[NL], x2[NL], x3[NL];
                u32 q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                for (int half = 0; half < 2; half++) {
                    const int dd = d + 2 * half;
                    const u32 D = 1u << dd;
                    const u32 elow = p & (D - 1);
                    const u32 e0 = ((p >> dd) << (dd + 2)) | elow;
                    q0 = lpos(e0, t); q1 = lpos(e0 + D, t); q2 = lpos(e0 + 2 * D, t); q3 = lpos(e0 + 3 * D, t);
                    Fr wa = ld_twiddle(a.tw, twidx(dd, elow, t));
                    Fr wb0 = ld_twiddle(a.tw, twidx(dd + 1, elow, t));
                    Fr wb1 = ld_twiddle(a.tw, twidx(dd + 1, elow + D, t));
                    if (half == 0) { ld(x0, q0); ld(x1, q1); ld(x2, q2); ld(x3, q3); }
                    Fr t1 = mul(as_fr(x1), wa);
                    Fr t3 = mul(as_fr(x3), wa);
                    u32 a0[NL], a1[NL], a2[NL], a3[NL];
                    raw_add(a0, x0, t1); raw_sub2q(a1, x0, t1);
                    raw_add(a2, x2, t3); raw_sub2q(a3, x2, t3);
                    Fr u2 = mul(as_fr(a2), wb0);
                    Fr u3 = mul(as_fr(a3), wb1);
                    u32 y[NL];
                    Fr r0, r1, r2, r3;
                    raw_add(y, a0, u2); r0 = norm_weak<FrParams>(y);
                    raw_sub2q(y, a0, u2); r2 = norm_weak<FrParams>(y);
                    raw_add(y, a1, u3); r1 = norm_weak<FrParams>(y);
                    raw_sub2q(y, a1, u3); r3 = norm_weak<FrParams>(y);
#pragma unroll
                    for (int l = 0; l < NL; l++) { x0[l] = r0.l[l]; x1[l] = r1.l[l]; x2[l] = r2.l[l]; x3[l] = r3.l[l]; }
                }
                st(q0, as_fr(x0)); st(q1, as_fr(x1)); st(q2, as_fr(x2)); st(q3, as_fr(x3));
            }
            __syncthreads();
        }
    }
#endif
    for (; d < k; d += 2) {
        const u32 D = 1u << d;
        for (u32 bf = tid; bf < (u32)(E >> 2); bf += nth) {
            u32 t, p;
            if (s0 == 0) { t = bf >> (k - 2); p = bf & ((1u << (k - 2)) - 1); } else { t = bf & tmask; p = bf >> j; }
            const u32 elow = p & (D - 1);
            const u32 e0 = ((p >> d) << (d + 2)) | elow;
            const u32 p0 = lpos(e0, t), p1 = lpos(e0 + D, t), p2 = lpos(e0 + 2 * D, t), p3 = lpos(e0 + 3 * D, t);
            // twiddles first: their latency (L2 / HBM) hides behind the LDS reads
            const Twiddle<SHOUP> wa = Twiddle<SHOUP>::load(a.tw, twidx(d, elow, t));
            const Twiddle<SHOUP> wb0 = Twiddle<SHOUP>::load(a.tw, twidx(d + 1, elow, t));
            const Twiddle<SHOUP> wb1 = Twiddle<SHOUP>::load(a.tw, twidx(d + 1, elow + D, t));
            u32 x0[NL], x1[NL], x2[NL], x3[NL];
            ld(x0, p0); ld(x1, p1); ld(x2, p2); ld(x3, p3);
            // stage d: (x0, x1) and (x2, x3), both with wa; sums left raw
            Fr t1 = wa.times(x1);
            Fr t3 = wa.times(x3);
            u32 a0[NL], a1[NL], a2[NL], a3[NL];
            raw_add(a0, x0, t1); raw_subkq<KQ>(a1, x0, t1);
            raw_add(a2, x2, t3); raw_subkq<KQ>(a3, x2, t3);
            // stage d + 1: (a0, a2) with wb0, (a1, a3) with wb1 = wb0 * omega_4
            Fr u2 = wb0.times(a2);
            Fr u3 = wb1.times(a3);
            u32 y[NL];
            raw_add(y, a0, u2); st(p0, norm_weak<FrParams>(y));
            raw_subkq<KQ>(y, a0, u2); st(p2, norm_weak<FrParams>(y));
            raw_add(y, a1, u3); st(p1, norm_weak<FrParams>(y));
            raw_subkq<KQ>(y, a1, u3); st(p3, norm_weak<FrParams>(y));
        }
        __syncthreads();
    }

    // (a scattering last pass walks its tile so that adjacent lanes store to adjacent places: local element (e, t) goes to
    // bitrev(g) = (bitrev(lo0 + t) << k) | bitrev_k(e), so the lanes take e in bit-reversed order -- 2 KB per wave store
    // instead of 64 pieces of 32 bytes in as many lines; round 3: 0.94 -> see profiles/r03_variants_ab.txt item 11)
    const bool by_place = a.scatter_bitrev && s0 != 0;
    for (u32 idx = tid; idx < (u32)E; idx += nth) {
        const u32 pos = by_place ? lpos(bit_reverse(idx & ((1u << k) - 1), k), idx >> k) : idx;
        Fr x;
        ld(x.l, pos);
        u32 g = gidx(pos);
        u32 dst = a.scatter_bitrev ? bit_reverse(g, a.logn) : g;
        // (post is indexed by the PLACE an element is stored at: a table for a scattering transform is kept in bit-reversed
        // order, NttPlan::twist, so that its reads are as coalesced as the stores)
        if (a.post) x = mul(x, ld_packed<FrParams>(a.post + (size_t)dst * 8));  // strict, < 2q
        else if (a.post_const) x = mul(x, ld_packed<FrParams>(a.post_const));
        else x = contract(x);                                                   // < 2.01 q, strict
        if (a.fin_a) {
            // fused h = a o b - c, plain integers (last pass of the third chain; S9, src/groth16.cpp:142-148)
            Fr tt = mul(ld_packed<FrParams>(a.fin_a + (size_t)dst * 8), ld_packed<FrParams>(a.fin_b + (size_t)dst * 8));   // < 2q
            u32 w[8];
            to_normal(w, sub<6>(tt, x));                                        // x < 2.01 q here
            store8(a.out + (size_t)dst * 8, w);
        } else {
            st_packed(a.out + (size_t)dst * 8, x);
        }
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
// unpacked: entries of TW_WORDS words holding the 9 limbs (twiddle tables); else packed 32-byte entries
__global__ void power_table_kernel(u32* table, const u32* base_packed, const u32* scale_packed, u64 count, int unpacked) {
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
        const Fr c = cond_sub_q(acc);
        if (unpacked == 2) {
            // Shoup entry: the plain constant and its quotient (ld_twiddle_shoup's layout)
            Fr lone = fp_zero<FrParams>();
            lone.l[0] = 1;
            const Fr wp = cond_sub_q(mul(c, lone));                    // c / 2^261: the plain value
            const Fr wq = shoup_quotient(c);
            u32* o = table + (start + i) * TWS_WORDS;
#pragma unroll
            for (int l = 0; l < 8; l++) { o[l] = wp.l[l]; o[8 + l] = wq.l[l]; }
            o[16] = wp.l[8]; o[17] = wq.l[8]; o[18] = 0; o[19] = 0;
        } else if (unpacked) {
            u32* o = table + (start + i) * TW_WORDS;
#pragma unroll
            for (int l = 0; l < NL; l++) o[l] = c.l[l];
            o[9] = 0; o[10] = 0; o[11] = 0;
        } else st_packed(table + (start + i) * 8, c);
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
    UG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (1 << NTT_MAX_LOG_TILE) * NL * 4));
    UG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (1 << NTT_MAX_LOG_TILE) * NL * 4));
    // twiddle products in Shoup form (tuning knob UG_NTT_SHOUP=0: the Montgomery form of rounds 1-4; both are exact)
    {
        const char* e = getenv("UG_NTT_SHOUP");
        shoup = !(e && e[0] == '0');
    }
    const size_t tw_words = shoup ? TWS_WORDS : TW_WORDS;
    u64 n = (u64)1 << logn;
    // Stage-major twiddle tables: stage s (butterfly span 2^s) reads omega_{2^(s+1)}^j at consecutive j, so the
    // lanes of a wave touch consecutive 32-byte entries at every stage (a single table of omega_n^i indexed with a
    // stage-dependent stride makes the late stages hit one L2 channel with 64 KiB strides).
    u64 entries = n > 1 ? n - 1 : 1;
    UG_HIP(hipMalloc(&tw_fwd, entries * tw_words * 4));
    UG_HIP(hipMalloc(&tw_inv, entries * tw_words * 4));
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
    auto launch = [&](u32* table, int base_i, int scale_i, u64 count, int unpacked) {
        u64 threads = (count + POW_RUN - 1) / POW_RUN;
        unsigned blocks = (unsigned)((threads + 255) / 256);
        hipLaunchKernelGGL(power_table_kernel, dim3(blocks), dim3(256), 0, stream, table, d_consts + 8 * base_i, d_consts + 8 * scale_i, count, unpacked);
        UG_KERNEL_CHECK();
    };
    for (int st = 0; st < logn; st++) {
        u64 off = ((u64)1 << st) - 1;
        launch(tw_fwd + off * tw_words, 3 + 2 * st, 2, (u64)1 << st, shoup ? 2 : 1);
        launch(tw_inv + off * tw_words, 4 + 2 * st, 2, (u64)1 << st, shoup ? 2 : 1);
    }
    // n^-1 * omega_{2n}^i (packed: one product per element, in the last pass of the inverse transform), stored at bitrev(i):
    // that pass scatters element i to place bitrev(i) and multiplies there
    u32* natural = nullptr;
    UG_HIP(hipMalloc(&natural, n * 32));
    launch(natural, 0, 1, n, 0);
    hipLaunchKernelGGL(bitrev_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, twist, natural, logn);
    UG_KERNEL_CHECK();
    UG_HIP(hipMemcpyAsync(ninv, d_consts + 8, 32, hipMemcpyDeviceToDevice, stream));
    UG_HIP(hipStreamSynchronize(stream));
    UG_HIP(hipFree(natural));
    UG_HIP(hipFree(d_consts));
}

void NttPlan::release() {
    if (tw_fwd) hipFree(tw_fwd);
    if (tw_inv) hipFree(tw_inv);
    if (twist) hipFree(twist);
    if (ninv) hipFree(ninv);
    tw_fwd = tw_inv = twist = ninv = nullptr;
}

// The passes of one DIT transform. `in` holds the input in bit-reversed order unless gather_bitrev is set (then
// natural order, gathered on the fly); output natural order, or bit-reversed if scatter_bitrev.
// post (optional) multiplies the output element stored at place p by post[p] in the last pass (for a scattering transform the
// table is therefore in bit-reversed order).
// Buffers: the first pass reads `in` (and fuse->in2); intermediate passes run in place on `mid` = fuse->work when given
// (then `in` is only read), else `in` itself for a scattering transform (which is clobbered) and `out` otherwise; the last
// pass writes `out`.
int NttPlan::passes(NttPass* list, u32* out, const u32* in, bool inverse, bool gather_bitrev, bool scatter_bitrev,
                    const u32* post, const u32* post_const, const NttFusion* fuse) const {
    const u32* in2 = fuse ? fuse->in2 : nullptr;
    u32* work = fuse ? fuse->work : nullptr;
    if (logn <= 0) throw std::invalid_argument("ntt: a pass list needs at least two points");
    if (gather_bitrev && scatter_bitrev) throw std::invalid_argument("ntt: gather and scatter together not supported");
    if ((gather_bitrev || scatter_bitrev) && out == in && !work) throw std::invalid_argument("ntt: permuting transform must be out of place");
    // split the stages: first pass contiguous (up to NTT_MAX_LOG_TILE stages), then strided passes with 2^j adjacent elements per row
    int stages[NTT_MAX_PASSES], nj[NTT_MAX_PASSES], np = 0, rem = logn;
    int first = rem < NTT_MAX_LOG_TILE ? rem : NTT_MAX_LOG_TILE;
    stages[np] = first; nj[np] = 0; np++; rem -= first;
    while (rem > 0) {
        // strided pass: k <= 8 stages and 2^j adjacent tiles, k + j = NTT_MAX_LOG_TILE (runs of 2^j * 32 bytes)
        int npass_left = (rem + 7) / 8;
        int k = (rem + npass_left - 1) / npass_left;
        stages[np] = k; nj[np] = NTT_MAX_LOG_TILE - k; np++; rem -= k;
    }
    if (np == 1 && (gather_bitrev || scatter_bitrev) && out == in) throw std::invalid_argument("ntt: a single permuting pass must be out of place");
    u32* mid = work ? work : (scatter_bitrev ? const_cast<u32*>(in) : out);
    if ((gather_bitrev || scatter_bitrev) && np > 1 && out == mid && scatter_bitrev) throw std::invalid_argument("ntt: the scattering pass must leave its work buffer");
    int s0 = 0;
    for (int p = 0; p < np; p++) {
        NttPass& a = list[p];
        bool last = (p == np - 1);
        a.in = (p == 0) ? in : mid;
        a.out = last ? out : mid;
        a.in2 = (p == 0) ? in2 : nullptr;
        a.fin_a = (last && fuse) ? fuse->fin_a : nullptr;
        a.fin_b = (last && fuse) ? fuse->fin_b : nullptr;
        a.tw = inverse ? tw_inv : tw_fwd; a.logn = logn; a.s0 = s0; a.k = stages[p];
        a.j = nj[p];
        if (s0 > 0 && a.j > s0) a.j = s0;
        a.gather_bitrev = (p == 0 && gather_bitrev) ? 1 : 0;
        a.scatter_bitrev = (last && scatter_bitrev) ? 1 : 0;
        a.post = last ? post : nullptr;
        a.post_const = last ? post_const : nullptr;
        s0 += stages[p];
    }
    return np;
}

// pass `p` of up to three transforms of this plan's size in ONE launch (their pass lists share the geometry)
void NttPlan::launch(const NttPass* const* lists, int count, int p, hipStream_t stream, MsmStats* stats) const {
    if (count < 1 || count > 3) throw std::logic_error("ntt: batch size");
    PassBatch b;
    for (int c = 0; c < count; c++) {
        const NttPass& n = lists[c][p];
        PassArgs& a = b.a[c];
        a.in = n.in; a.out = n.out; a.tw = n.tw; a.post = n.post; a.post_const = n.post_const; a.in2 = n.in2; a.fin_a = n.fin_a; a.fin_b = n.fin_b;
        a.logn = n.logn; a.s0 = n.s0; a.k = n.k; a.j = n.j; a.gather_bitrev = n.gather_bitrev; a.scatter_bitrev = n.scatter_bitrev;
#ifdef UG_MEASURE
        static const int fuse_steps = getenv("UG_NTT_FUSE_STEPS") ? atoi(getenv("UG_NTT_FUSE_STEPS")) : 0;
        a.fuse_steps = fuse_steps;
#endif
    }
    for (int c = count; c < 3; c++) b.a[c] = b.a[0];
    const PassArgs& a = b.a[0];
    int E = 1 << (a.k + a.j);
    unsigned blocks = (unsigned)(((u64)1 << logn) >> (a.k + a.j));
    int threads = E / 4 > NTT_THREADS ? NTT_THREADS : (E / 4 < 64 ? 64 : E / 4);
    size_t lds = (size_t)E * NL * 4;
    int slot = stats ? stats->begin(stream, ((u64)1 << logn) * (u64)count) : -1;
    if (shoup) hipLaunchKernelGGL(ntt_pass_kernel<true>, dim3(blocks, (unsigned)count), dim3(threads), lds, stream, b);
    else hipLaunchKernelGGL(ntt_pass_kernel<false>, dim3(blocks, (unsigned)count), dim3(threads), lds, stream, b);
    UG_KERNEL_CHECK();
    if (stats) stats->end(slot, stream);
}

void NttPlan::transform(u32* out, const u32* in, bool inverse, bool gather_bitrev, bool scatter_bitrev,
                        const u32* post, const u32* post_const, hipStream_t stream, MsmStats* stats, const NttFusion* fuse) const {
    if (logn == 0) {
        if (fuse && (fuse->in2 || fuse->fin_a)) throw std::invalid_argument("ntt: fused forms need at least two points");
        if (out != in) UG_HIP(hipMemcpyAsync(out, in, 32, hipMemcpyDeviceToDevice, stream));
        return;   // size-1 transform is the identity (n^-1 = 1, omega_2^0 = 1)
    }
    if ((gather_bitrev || scatter_bitrev) && out == in) throw std::invalid_argument("ntt: permuting transform must be out of place");
    NttPass list[NTT_MAX_PASSES];
    const int np = passes(list, out, in, inverse, gather_bitrev, scatter_bitrev, post, post_const, fuse);
    const NttPass* one[1] = {list};
    for (int p = 0; p < np; p++) launch(one, 1, p, stream, stats);
}

void bitrev_copy(u32* out, const u32* in, int logn, hipStream_t stream) {
    u64 n = (u64)1 << logn;
    hipLaunchKernelGGL(bitrev_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, in, logn);
    UG_KERNEL_CHECK();
}

}  // namespace ug
