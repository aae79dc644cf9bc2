// ff.hpp -- BN254 field arithmetic for gfx950, shared by device kernels and host glue.
//
// Replaces the reference's field layer on the prover hot path:
//   F{r,q}_rawMMul / rawAdd / rawSub / rawNeg / To/FromMontgomery
//   (build/fr_raw_generic.cpp:11-39,68-80,107-148,192-232; x86 form build/fr.asm:372-538).
//
// MI355X-first representation ("F29"): an element is 9 limbs of 29 bits held in uint32, value
// < 2^261, Montgomery radix R' = 2^261. A 29x29-bit product is < 2^58, so the 64-bit column
// sums of a schoolbook product (<= 18 products + 9 reduction products per column) never
// overflow: one v_mad_u64_u32 per product and NO carry instructions. Measured on MI355X
// (tools/ubench_int.hip): v_mad_u64_u32 and v_addc_co_u32 both issue at ~5 cycles per wave, so
// the usual 8x32-bit carry-chain form spends as much on carries as on multiplies; this form
// runs at the v_mad_u64_u32 issue peak (171 G modmul/s vs 97 G for the compiler's 8x32 CIOS).
//
// Lazy reduction: values are kept only "below a small multiple of q" (never above 2^261 ~ 170 q)
// and are NOT conditionally reduced after add/sub. mul() contracts the range again:
//     mul(a,b) = a*b/R' mod q,   result < a*b/R' + q,   i.e. < 2q whenever (a/q)*(b/q) < 170.
// Every formula that uses these ops states the bound it relies on. canon() gives the unique
// representative in [0,q) for output, storage as 32 bytes, and zero tests.
//
// Limb discipline: "strict" = every limb < 2^29; "weak" = every limb <= 2^29 + 15 (top limb free).
// mul/sqr accept weak inputs and return strict outputs; add/sub return weak outputs.
#pragma once
#include <cstdint>
#include "ff_consts.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UG_HD __host__ __device__ __forceinline__
#else
#define UG_HD inline
#endif

namespace ug {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int NL = 9;
constexpr int LB = 29;
constexpr u32 MASK29 = 0x1fffffffu;

template <class P>
struct Fp {
    u32 l[NL];
    typedef P Params;
};

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

// Host-only range checker for the lazy-reduction bounds (tests build with -DUG_CHECK_BOUNDS).
#if defined(UG_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
}  // namespace ug
#include <cstdio>
#include <cstdlib>
namespace ug {
template <class P> inline void check_lt_kq(const Fp<P>& a, int k, const char* what) {
    // strict-normalise a and k*q, compare from the top limb
    u64 av[NL], kv[NL], ca = 0, ck = 0;
    for (int i = 0; i < NL; i++) {
        u64 x = (u64)a.l[i] + ca; av[i] = (i < NL - 1) ? (x & MASK29) : x; ca = (i < NL - 1) ? (x >> LB) : 0;
        u64 y = (u64)P::q[i] * (u64)k + ck; kv[i] = (i < NL - 1) ? (y & MASK29) : y; ck = (i < NL - 1) ? (y >> LB) : 0;
    }
    for (int i = NL - 1; i >= 0; i--) {
        if (av[i] < kv[i]) return;
        if (av[i] > kv[i]) break;
    }
    fprintf(stderr, "UG_CHECK_BOUNDS: value not below %d*q in %s\n", k, what);
    abort();
}
#define UG_BOUND(a, k, what) check_lt_kq(a, k, what)
#else
#define UG_BOUND(a, k, what) ((void)0)
#endif

template <class P> UG_HD Fp<P> fp_from(const u32* c) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c[i];
    return r;
}
template <class P> UG_HD Fp<P> fp_zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    return r;
}
template <class P> UG_HD Fp<P> fp_one() { return fp_from<P>(P::one); }      // Montgomery 1

// ---- column products -------------------------------------------------------------------------
// c[0..17] += a * b (schoolbook, 81 mads). Column bound: each product < (2^29+16)^2 < 2^58.01.
template <class P> UG_HD void cols_mul(u64* c, const Fp<P>& a, const Fp<P>& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (u64)a.l[i] * b.l[j];
    }
}
// c += a * a (45 mads): diagonal + doubled off-diagonal.
template <class P> UG_HD void cols_sqr(u64* c, const Fp<P>& a) {
    u32 d[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) d[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        c[2 * i] += (u64)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < NL; j++) c[i + j] += (u64)a.l[i] * d[j];
    }
}
UG_HD void cols_zero(u64* c) {
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) c[k] = 0;
}
// Montgomery reduction of 18 columns (value T = sum c[k] 2^(29k) < 170 q * 2^261):
// returns T / 2^261 mod q as a strict element, result < T/2^261 + q. 81 mads + 9 mul_lo.
template <class P> UG_HD Fp<P> redc(u64* c) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        u32 m = ((u32)c[i] * P::np) & MASK29;
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (u64)m * P::q[j];
        c[i + 1] += c[i] >> LB;
    }
    Fp<P> r;
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
        r.l[k - NL] = (u32)c[k] & MASK29;
        c[k + 1] += c[k] >> LB;
    }
    r.l[NL - 1] = (u32)c[2 * NL - 1];
    return r;
}

// a*b/R' mod q. Requires (a/q)*(b/q) < 169; result strict, < a*b/R' + q.
template <class P> UG_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
    u64 c[2 * NL];
    cols_zero(c);
    cols_mul(c, a, b);
    return redc<P>(c);
}
template <class P> UG_HD Fp<P> sqr(const Fp<P>& a) {
    u64 c[2 * NL];
    cols_zero(c);
    cols_sqr(c, a);
    return redc<P>(c);
}
// a*b + c*d, one reduction. Requires (a/q)(b/q) + (c/q)(d/q) < 169.
template <class P> UG_HD Fp<P> mul_add(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c_, const Fp<P>& d) {
    u64 c[2 * NL];
    cols_zero(c);
    cols_mul(c, a, b);
    cols_mul(c, c_, d);
    return redc<P>(c);
}

// ---- Shoup products: one operand is a TABLE constant (twiddles) --------------------------------------
// x * w mod q for a constant w < q kept beside wq = floor(w * 2^261 / q):
//     quot = floor(x * wq / 2^261)  (from the columns >= 7 of the product: 53 multiply-adds),
//     r    = (x * w + quot * (2^261 - q)) mod 2^261  (low columns only: 90 multiply-adds)
// -- 143 multiply-adds against the Montgomery product's 162 + 9 mul_lo, and the data keep whatever form they have (a value
// x R' times a PLAIN constant w is x w R': Montgomery-form data stay Montgomery-form).
// x: limbs below 2^31 + 2^30 (un-normalised sums are fine), value below 2^261 (~170 q). w, wq: strict limbs.
// quot never exceeds floor(x w / q) and falls short of it by at most 2 (wq is a floor: < 1; the dropped columns 0..6 of x * wq
// are worth < 2^-23; the floor: < 1), so the result is x w - quot q in [0, 3q), strict limbs.
template <class P> UG_HD Fp<P> mul_shoup(const u32* x, const u32* w, const u32* wq) {
    u64 h[NL + 1];                                   // columns 7 .. 16 of x * wq
#pragma unroll
    for (int k = 0; k <= NL; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++)
            if (i + j >= 7) h[i + j - 7] += (u64)x[i] * wq[j];
    }
    u32 qt[NL];
    u64 carry = ((h[0] >> LB) + h[1]) >> LB;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        const u64 v = h[2 + k] + carry;
        qt[k] = (u32)v & MASK29;
        carry = v >> LB;
    }
    qt[NL - 1] = (u32)carry;
    u64 c[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++)
            if (i + j < NL) c[i + j] += (u64)x[i] * w[j] + (u64)qt[i] * P::qc[j];
    }
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        r.l[k] = (u32)c[k] & MASK29;
        c[k + 1] += c[k] >> LB;
    }
    r.l[NL - 1] = (u32)c[NL - 1] & MASK29;           // (modulo 2^261: the true value is below 3q < 2^256)
    return r;
}
// wq = floor(w * 2^261 / q) from cm = w * 2^261 mod q (canonical: the Montgomery form of w) -- w 2^261 = wq q + cm, so
// wq = -cm * q^-1 modulo 2^261, and wq < 2^261 makes that the value itself. cm != 0.
template <class P> UG_HD Fp<P> shoup_quotient(const Fp<P>& cm) {
    u32 n[NL];                                       // 2^261 - cm
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const u32 v = (0u - cm.l[i] - borrow) & MASK29;
        borrow = (cm.l[i] + borrow) ? 1u : 0u;
        n[i] = v;
    }
    u64 c[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++)
            if (i + j < NL) c[i + j] += (u64)n[i] * P::qinv[j];
    }
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        r.l[k] = (u32)c[k] & MASK29;
        c[k + 1] += c[k] >> LB;
    }
    r.l[NL - 1] = (u32)c[NL - 1] & MASK29;
    return r;
}

// ---- carry handling ----------------------------------------------------------------------------
// One parallel carry pass: inputs with limbs < 2^32 (non-negative), output weak.
template <class P> UG_HD Fp<P> norm_weak(const u32* x) {
    Fp<P> r;
    r.l[0] = x[0] & MASK29;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.l[i] = (x[i] & MASK29) + (x[i - 1] >> LB);
    r.l[NL - 1] = x[NL - 1] + (x[NL - 2] >> LB);
    return r;
}
// Serial carry propagation: any limbs < 2^32 -> strict.
template <class P> UG_HD Fp<P> norm_strict(const Fp<P>& a) {
    Fp<P> r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        u32 v = a.l[i] + carry;
        r.l[i] = v & MASK29;
        carry = v >> LB;
    }
    r.l[NL - 1] = a.l[NL - 1] + carry;
    return r;
}

// a + b  (value adds; no reduction)
template <class P> UG_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = a.l[i] + b.l[i];
    return norm_weak<P>(x);
}
template <class P> UG_HD Fp<P> dbl(const Fp<P>& a) {
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = a.l[i] << 1;
    return norm_weak<P>(x);
}
template <class P> UG_HD Fp<P> triple(const Fp<P>& a) {
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = a.l[i] * 3u;
    return norm_weak<P>(x);
}

template <class P, int K> UG_HD const u32* kq_padded() {
    static_assert(K >= 1 && K <= 16, "multiple of q out of table");
    return K == 1 ? P::kq1 : K == 2 ? P::kq2 : K == 3 ? P::kq3 : K == 4 ? P::kq4
         : K == 5 ? P::kq5 : K == 6 ? P::kq6 : K == 7 ? P::kq7 : K == 8 ? P::kq8
         : K == 9 ? P::kq9 : K == 10 ? P::kq10 : K == 11 ? P::kq11 : K == 12 ? P::kq12
         : K == 13 ? P::kq13 : K == 14 ? P::kq14 : K == 15 ? P::kq15 : P::kq16;
}
// a - b + K*q. Requires b < K*q (b weak); result value = a + Kq - b > 0, weak.
// The table holds K*q with every limb below the top raised by 2*2^29 (next limb lowered by 2), so
// no limb below the top ever goes negative; the top limb is exact modulo 2^32.
template <int K, class P> UG_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
    const u32* kq = kq_padded<P, K>();
    UG_BOUND(b, K, "sub");
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = a.l[i] + kq[i] - b.l[i];
    return norm_weak<P>(x);
}
// K*q - a
template <int K, class P> UG_HD Fp<P> neg(const Fp<P>& a) {
    const u32* kq = kq_padded<P, K>();
    UG_BOUND(a, K, "neg");
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = kq[i] - a.l[i];
    return norm_weak<P>(x);
}

// a - b - 2c + 5q in one pass (X3 = R^2 - PPP - 2Q of the addition formulas). Requires b + 2c < 5q.
// Uses the table padded by 4*2^29 per limb, so limbs stay non-negative for weak b, c.
template <class P> UG_HD Fp<P> sub_b_2c_5q(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c) {
    UG_BOUND(add(b, dbl(c)), 5, "sub_b_2c_5q");
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = a.l[i] + P::kqw5[i] - b.l[i] - (c.l[i] << 1);
    return norm_weak<P>(x);
}
// a*b + c*(K q - d) with one reduction  (= a*b - c*d mod q). Requires d < K q.
// Result < (A B + C K)/170 q + q for operands below A q, B q, C q.
template <int K, class P> UG_HD Fp<P> mul_sub(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c_, const Fp<P>& d) {
    u64 c[2 * NL];
    Fp<P> nd = neg<K>(d);
    cols_zero(c);
    cols_mul(c, a, b);
    cols_mul(c, c_, nd);
    return redc<P>(c);
}

// ---- exact forms ---------------------------------------------------------------------------------
template <class P> UG_HD bool limbs_all_zero(const Fp<P>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= a.l[i];
    return o == 0;
}
template <class P> UG_HD bool limbs_equal_q(const Fp<P>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= a.l[i] ^ P::q[i];
    return o == 0;
}
// For a STRICT value known to be < 2q (e.g. any mul/sqr/redc output): is it 0 mod q ?
template <class P> UG_HD bool is_zero_lt2q(const Fp<P>& a) { return limbs_all_zero(a) || limbs_equal_q(a); }

// strict a < 2q  ->  canonical [0,q)
template <class P> UG_HD Fp<P> cond_sub_q(const Fp<P>& a) {
    Fp<P> d;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = (int32_t)a.l[i] - (int32_t)P::q[i] + borrow;
        d.l[i] = (u32)v & MASK29;
        borrow = v >> LB;            // 0 or -1
    }
    // borrow == 0  <=>  a >= q ; the top limb of d is < 2^29 in that case
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
}
// any a < 169 q  ->  canonical [0,q), same residue
template <class P> UG_HD Fp<P> canon(const Fp<P>& a) { return cond_sub_q(mul(a, fp_one<P>())); }
template <class P> UG_HD bool is_zero(const Fp<P>& a) { return limbs_all_zero(canon(a)); }
template <class P> UG_HD bool equal(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> x = canon(a), y = canon(b);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= x.l[i] ^ y.l[i];
    return o == 0;
}

// ---- 256-bit (8 x u32, little-endian) <-> limbs ----------------------------------------------------
template <class P> UG_HD Fp<P> unpack256(const u32* w) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, j = bit >> 5, sh = bit & 31;
        u64 two = (u64)w[j] | ((u64)(j + 1 < 8 ? w[j + 1] : 0u) << 32);
        r.l[i] = (u32)(two >> sh) & MASK29;
    }
    return r;
}
// requires strict limbs and value < 2^256
template <class P> UG_HD void pack256(u32* w, const Fp<P>& a) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, i = bit / LB, sh = bit - i * LB;    // limb i holds bit `bit` at offset sh
        u64 v = (u64)a.l[i] >> sh;
        v |= (u64)a.l[i + 1] << (LB - sh);
        if (i + 2 < NL) v |= (u64)a.l[i + 2] << (2 * LB - sh);
        w[j] = (u32)v;
    }
}

// reference Montgomery form (x * 2^256 mod q, 32 bytes)  ->  device form (x * 2^261), < 2q strict
template <class P> UG_HD Fp<P> from_mont256(const u32* w) { return mul(unpack256<P>(w), fp_from<P>(P::from256)); }
// device form -> reference Montgomery form, canonical, packed
template <class P> UG_HD void to_mont256(u32* w, const Fp<P>& a) {
    pack256(w, cond_sub_q(mul(a, fp_from<P>(P::to256))));
}
// plain integer (32 bytes, any value < 2^256)  ->  device form
template <class P> UG_HD Fp<P> from_normal(const u32* w) { return mul(unpack256<P>(w), fp_from<P>(P::r2)); }
// device form -> canonical plain integer, packed
template <class P> UG_HD void to_normal(u32* w, const Fp<P>& a) {
    Fp<P> o = fp_zero<P>();
    o.l[0] = 1;
    pack256(w, cond_sub_q(mul(a, o)));
}
// device form, value < 2^256 (e.g. < 5 q), strict  ->  32 bytes and back (no change of residue/form)
template <class P> UG_HD void store_packed(u32* w, const Fp<P>& a) { pack256(w, a); }
template <class P> UG_HD Fp<P> load_packed(const u32* w) { return unpack256<P>(w); }

// a^e for a 256-bit exponent (8 x u32 LE), Montgomery in/out; a < 8q; result < 2q strict
template <class P> UG_HD Fp<P> pow256(const Fp<P>& a, const u32* e) {
    Fp<P> acc = fp_one<P>();
    bool started = false;
    for (int i = 255; i >= 0; i--) {
        if (started) acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            if (started) acc = mul(acc, a); else { acc = a; started = true; }
        }
    }
    return mul(acc, fp_one<P>());
}
// a^-1 = a^(q-2); Montgomery in/out (same contract as RawFr::inv, build/fr.cpp:238-250). inv(0) = 0.
// (Fermat form: 254 squarings + ~127 products. Kept as the cross-check of inv() below.)
template <class P> UG_HD Fp<P> inv_fermat(const Fp<P>& a) {
    u32 e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::q32[i];
    e[0] -= 2;                                   // q is odd and q32[0] >= 2: no borrow
    return pow256(a, e);
}

// ---- inversion by divsteps (Bernstein-Yang "safegcd"), constant time, no data-dependent branch ----------------------
// The same contract as inv_fermat at about a tenth of its cost on this ISA (round 3): the loop works on 32-bit words with
// selects, and per 29 divsteps it needs 90 multiply-adds for the two matrix-vector updates instead of the 29 squarings
// (3 650 multiply-adds) a Fermat chain spends on the same 29 bits. Lanes never diverge: every value takes 21 x 29 = 609
// divsteps (590 suffice for any odd modulus below 2^256 with the half-delta start used here).
// State: f, g signed integers of 9 limbs x 29 bits (the top limb carries the sign); d, e in (-2q, q) with
// d * x = f and e * x = g (mod q) throughout; at the end g = 0, f = +-1, so x^-1 = +-d.
namespace safegcd {
struct S29 { int32_t v[NL]; };
struct Mat { int32_t u, v, q, r; };

// 29 divsteps on the low 32 bits of f and g; t = the transition matrix scaled by 2^29 (entries within +-2^29)
UG_HD int32_t divsteps29(int32_t zeta, u32 f0, u32 g0, Mat& t) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < LB; i++) {
        u32 c1 = (u32)(zeta >> 31);              // all ones while zeta < 0
        const u32 c2 = 0u - (g & 1u);            // all ones when g is odd
        const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;      // (f, u, v) negated when zeta < 0
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                // swap-and-negate step only when zeta < 0 and g odd
        zeta = (zeta ^ (int32_t)c1) - 1;
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// (f, g) <- t * (f, g) / 2^29  (exact)
UG_HD void update_fg(S29& f, S29& g, const Mat& t) {
    int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0];
    int64_t cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
    cf >>= LB; cg >>= LB;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        cf += (int64_t)t.u * f.v[i] + (int64_t)t.v * g.v[i];
        cg += (int64_t)t.q * f.v[i] + (int64_t)t.r * g.v[i];
        f.v[i - 1] = (int32_t)((u32)cf & MASK29); cf >>= LB;
        g.v[i - 1] = (int32_t)((u32)cg & MASK29); cg >>= LB;
    }
    f.v[NL - 1] = (int32_t)cf; g.v[NL - 1] = (int32_t)cg;
}
// (d, e) <- t * (d, e) / 2^29 mod q: multiples md, me of q are added so that the low 29 bits vanish
template <class P> UG_HD void update_de(S29& d, S29& e, const Mat& t) {
    const u32 qinv = (0u - P::np) & MASK29;                       // q^-1 mod 2^29
    const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);      // + q for a negative d, e: keeps them in (-2q, q)
    int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0];
    int64_t ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
    md -= (int32_t)((qinv * (u32)cd + (u32)md) & MASK29);
    me -= (int32_t)((qinv * (u32)ce + (u32)me) & MASK29);
    cd += (int64_t)(int32_t)P::q[0] * md; ce += (int64_t)(int32_t)P::q[0] * me;
    cd >>= LB; ce >>= LB;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        cd += (int64_t)t.u * d.v[i] + (int64_t)t.v * e.v[i] + (int64_t)(int32_t)P::q[i] * md;
        ce += (int64_t)t.q * d.v[i] + (int64_t)t.r * e.v[i] + (int64_t)(int32_t)P::q[i] * me;
        d.v[i - 1] = (int32_t)((u32)cd & MASK29); cd >>= LB;
        e.v[i - 1] = (int32_t)((u32)ce & MASK29); ce >>= LB;
    }
    d.v[NL - 1] = (int32_t)cd; e.v[NL - 1] = (int32_t)ce;
}
// r in (-2q, q), negated when `negate` is all ones  ->  [0, q), limbs strict
template <class P> UG_HD void normalize(S29& r, int32_t negate) {
    int32_t add = r.v[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = ((r.v[i] + ((int32_t)P::q[i] & add)) ^ negate) - negate;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { r.v[i + 1] += r.v[i] >> LB; r.v[i] &= (int32_t)MASK29; }
    add = r.v[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] += (int32_t)P::q[i] & add;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { r.v[i + 1] += r.v[i] >> LB; r.v[i] &= (int32_t)MASK29; }
}
// x^-1 mod q for a canonical x (plain integers in and out; 0 -> 0)
template <class P> UG_HD Fp<P> inv_plain(const Fp<P>& x) {
    S29 f, g, d, e;
#pragma unroll
    for (int i = 0; i < NL; i++) { f.v[i] = (int32_t)P::q[i]; g.v[i] = (int32_t)x.l[i]; d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    int32_t zeta = -1;                           // zeta = -(delta + 1/2), delta starts at 1/2
#pragma unroll 1
    for (int it = 0; it < 21; it++) {
        Mat t;
        zeta = divsteps29(zeta, (u32)f.v[0] | ((u32)f.v[1] << LB), (u32)g.v[0] | ((u32)g.v[1] << LB), t);
        update_de<P>(d, e, t);
        update_fg(f, g, t);
    }
    normalize<P>(d, f.v[NL - 1] >> 31);          // f = +-1 (or +-q when x = 0, where d = 0 anyway)
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = (u32)d.v[i];
    return r;
}
}  // namespace safegcd

// a^-1, Montgomery in/out (same contract as RawFr::inv, build/fr.cpp:238-250): a < 169 q in; result strict, < 2q. inv(0) = 0.
// The stored value is X = a R'; its plain inverse X^-1 = a^-1 R'^-1 is lifted back by two products with R'^2.
template <class P> UG_HD Fp<P> inv(const Fp<P>& a) {
    const Fp<P> v = safegcd::inv_plain<P>(canon(a));
    const Fp<P> r2 = fp_from<P>(P::r2);
    return mul(mul(v, r2), r2);
}

// ---- Fp2 = Fp[u]/(u^2+1) ----------------------------------------------------------------------------
// element a + b u stored (a, b): the order of zkey G2 records and proof JSON (src/groth16.cpp:228-233)
template <class P>
struct Fp2 {
    Fp<P> a, b;
    typedef P Params;
};
typedef Fp2<FqParams> Fq2;

template <class P> UG_HD Fp2<P> f2_zero() { Fp2<P> r; r.a = fp_zero<P>(); r.b = fp_zero<P>(); return r; }
template <class P> UG_HD Fp2<P> f2_one() { Fp2<P> r; r.a = fp_one<P>(); r.b = fp_zero<P>(); return r; }
template <class P> UG_HD Fp2<P> add(const Fp2<P>& x, const Fp2<P>& y) { Fp2<P> r; r.a = add(x.a, y.a); r.b = add(x.b, y.b); return r; }
template <class P> UG_HD Fp2<P> dbl(const Fp2<P>& x) { Fp2<P> r; r.a = dbl(x.a); r.b = dbl(x.b); return r; }
template <class P> UG_HD Fp2<P> triple(const Fp2<P>& x) { Fp2<P> r; r.a = triple(x.a); r.b = triple(x.b); return r; }
template <int K, class P> UG_HD Fp2<P> sub(const Fp2<P>& x, const Fp2<P>& y) { Fp2<P> r; r.a = sub<K>(x.a, y.a); r.b = sub<K>(x.b, y.b); return r; }
template <int K, class P> UG_HD Fp2<P> neg(const Fp2<P>& x) { Fp2<P> r; r.a = neg<K>(x.a); r.b = neg<K>(x.b); return r; }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u : 4 column products, 2 reductions.
// -a1 b1 is formed as a1 * (KB q - b1), so y.b must be < KB q. With components of x below Bx q and
// of y below By q:   r.a < (Bx By + Bx KB)/170 q + q,   r.b < 2 Bx By/170 q + q,   both strict.
template <int KB, class P> UG_HD Fp2<P> mulk(const Fp2<P>& x, const Fp2<P>& y) {
    Fp2<P> r;
    u64 c[2 * NL];
    Fp<P> nb1 = neg<KB>(y.b);
    cols_zero(c); cols_mul(c, x.a, y.a); cols_mul(c, x.b, nb1); r.a = redc<P>(c);
    cols_zero(c); cols_mul(c, x.a, y.b); cols_mul(c, x.b, y.a); r.b = redc<P>(c);
    return r;
}
// (a0 + a1 u)^2 = (a0^2 + a1 (KB q - a1)) + (a0 * 2 a1) u : 3 column products (one a square), 2 reductions.
// Components below B q (B <= KB):  r.a < (B^2 + B KB)/170 q + q,  r.b < 2 B^2/170 q + q, strict.
template <int KB, class P> UG_HD Fp2<P> sqrk(const Fp2<P>& x) {
    Fp2<P> r;
    u64 c[2 * NL];
    Fp<P> nb = neg<KB>(x.b);
    cols_zero(c); cols_sqr(c, x.a); cols_mul(c, x.b, nb); r.a = redc<P>(c);
    cols_zero(c); cols_mul(c, x.a, dbl(x.b)); r.b = redc<P>(c);
    return r;
}
// a*b - c*d with one reduction per component: real = a0 b0 + a1 (KB q - b1) + c0 (K q - d0) + c1 d1,
// imag = a0 b1 + a1 b0 + c0 (K q - d1) + c1 (K q - d0). Requires b.b < KB q and d < K q (both components).
// Four column products share a reduction: 4 * 9 * 2^58 + 9 * 2^58 < 2^64.
template <int KB, int K, class P> UG_HD Fp2<P> mul_subk(const Fp2<P>& a, const Fp2<P>& b, const Fp2<P>& c_, const Fp2<P>& d) {
    Fp2<P> r;
    u64 c[2 * NL];
    Fp<P> nb1 = neg<KB>(b.b), nd0 = neg<K>(d.a), nd1 = neg<K>(d.b);
    cols_zero(c); cols_mul(c, a.a, b.a); cols_mul(c, a.b, nb1); cols_mul(c, c_.a, nd0); cols_mul(c, c_.b, d.b); r.a = redc<P>(c);
    cols_zero(c); cols_mul(c, a.a, b.b); cols_mul(c, a.b, b.a); cols_mul(c, c_.a, nd1); cols_mul(c, c_.b, nd0); r.b = redc<P>(c);
    return r;
}
template <int KB, int K, class P> UG_HD Fp<P> mul_subk(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c_, const Fp<P>& d) {
    return mul_sub<K>(a, b, c_, d);
}
template <class P> UG_HD Fp2<P> sub_b_2c_5q(const Fp2<P>& a, const Fp2<P>& b, const Fp2<P>& c) {
    Fp2<P> r; r.a = sub_b_2c_5q(a.a, b.a, c.a); r.b = sub_b_2c_5q(a.b, b.b, c.b); return r;
}
// same spelling for the base field, where no negation multiple is needed
template <int KB, class P> UG_HD Fp<P> mulk(const Fp<P>& a, const Fp<P>& b) { return mul(a, b); }
template <int KB, class P> UG_HD Fp<P> sqrk(const Fp<P>& a) { return sqr(a); }
template <class P> UG_HD Fp2<P> mul(const Fp2<P>& x, const Fp2<P>& y) { return mulk<8>(x, y); }
template <class P> UG_HD Fp2<P> sqr(const Fp2<P>& x) { return sqrk<8>(x); }
// x * k for k in Fp
template <class P> UG_HD Fp2<P> mul_fp(const Fp2<P>& x, const Fp<P>& k) { Fp2<P> r; r.a = mul(x.a, k); r.b = mul(x.b, k); return r; }
template <class P> UG_HD Fp2<P> canon(const Fp2<P>& x) { Fp2<P> r; r.a = canon(x.a); r.b = canon(x.b); return r; }
template <class P> UG_HD bool is_zero_lt2q(const Fp2<P>& x) { return is_zero_lt2q(x.a) && is_zero_lt2q(x.b); }
template <class P> UG_HD bool limbs_all_zero(const Fp2<P>& x) { return limbs_all_zero(x.a) && limbs_all_zero(x.b); }
template <class P> UG_HD Fp2<P> inv(const Fp2<P>& x) {
    Fp<P> n = add(sqr(x.a), sqr(x.b));          // < 4q
    Fp<P> ni = inv(n);
    Fp2<P> r;
    r.a = mul(x.a, ni);
    r.b = neg<2>(mul(x.b, ni));
    return r;
}
// Zero test for a STRICT value known to be < 3q (mul / sqr outputs, including Fp2 sqr's 2.51 q bound).
template <class P> UG_HD bool is_zero_small(const Fp<P>& a) {
    u32 o0 = 0, o1 = 0, o2 = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { o0 |= a.l[i]; o1 |= a.l[i] ^ P::q[i]; o2 |= a.l[i] ^ P::twoq[i]; }
    return o0 == 0 || o1 == 0 || o2 == 0;
}
template <class P> UG_HD bool is_zero_small(const Fp2<P>& x) { return is_zero_small(x.a) && is_zero_small(x.b); }

}  // namespace ug
