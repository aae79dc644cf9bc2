// hostmath_test.cpp -- host build of ff.hpp / ec.hpp for the CPU ("not gpu") test-suite.
// The same headers are compiled for gfx950 by the kernels; here they are compiled by g++ with
// -DUG_CHECK_BOUNDS so that every lazy-reduction bound stated in ec.hpp is asserted while the
// tests compare results with the oracle. Interface: reference byte formats (32-byte little-endian
// Montgomery R=2^256 values, zkey affine records, (0,0) = infinity).
#include <cstring>
#include <initializer_list>
#include <cstdint>
#include "ec.hpp"
#include "segmap.hpp"

using namespace ug;

namespace {
template <class P> Fp<P> ld(const uint8_t* p) { u32 w[8]; memcpy(w, p, 32); return from_mont256<P>(w); }
template <class P> void st(uint8_t* p, const Fp<P>& a) { u32 w[8]; to_mont256(w, a); memcpy(p, w, 32); }

bool g1_load(const uint8_t* p, Fq& x, Fq& y) {
    bool z = true; for (int i = 0; i < 64; i++) z &= (p[i] == 0);
    if (z) return false;
    x = canon(ld<FqParams>(p)); y = canon(ld<FqParams>(p + 32));
    return true;
}
bool g2_load(const uint8_t* p, Fq2& x, Fq2& y) {
    bool z = true; for (int i = 0; i < 128; i++) z &= (p[i] == 0);
    if (z) return false;
    x.a = canon(ld<FqParams>(p)); x.b = canon(ld<FqParams>(p + 32));
    y.a = canon(ld<FqParams>(p + 64)); y.b = canon(ld<FqParams>(p + 96));
    return true;
}
void g1_store(uint8_t* out, const G1XYZZ& p) {
    if (is_inf(p)) { memset(out, 0, 64); return; }
    Fq x, y; xyzz_to_affine(x, y, p);
    st(out, x); st(out + 32, y);
}
void g2_store(uint8_t* out, const G2XYZZ& p) {
    if (is_inf(p)) { memset(out, 0, 128); return; }
    Fq2 x, y; xyzz_to_affine(x, y, p);
    st(out, x.a); st(out + 32, x.b); st(out + 64, y.a); st(out + 96, y.b);
}
}  // namespace

extern "C" {

// op: 0 mul, 1 add, 2 sub, 3 neg(a), 4 inv(a), 5 sqr(a), 6 from/to normal round trip of a, 7 pack/unpack,
//     8 inv_fermat(a) (the a^(q-2) form inv() replaced), 9 inv of the lazily reduced a + 5q
int ugt_f_op(int which, int op, uint8_t* out, const uint8_t* a, const uint8_t* b) {
    if (which == 0) {
        typedef FrParams P;
        Fp<P> x = ld<P>(a), y = b ? ld<P>(b) : fp_zero<P>(), r;
        switch (op) {
            case 0: r = mul(x, y); break;
            case 1: r = add(x, y); break;
            case 2: r = sub<2>(x, y); break;
            case 3: r = neg<2>(x); break;
            case 4: r = inv(x); break;
            case 5: r = sqr(x); break;
            case 6: { u32 w[8]; to_normal(w, x); r = from_normal<P>(w); break; }
            case 7: { u32 w[8]; Fp<P> c = canon(x); pack256(w, c); r = unpack256<P>(w); break; }
            case 8: r = inv_fermat(x); break;
            case 9: r = inv(sub<5>(x, fp_zero<P>())); break;
            default: return 1;
        }
        st(out, r);
    } else {
        typedef FqParams P;
        Fp<P> x = ld<P>(a), y = b ? ld<P>(b) : fp_zero<P>(), r;
        switch (op) {
            case 0: r = mul(x, y); break;
            case 1: r = add(x, y); break;
            case 2: r = sub<2>(x, y); break;
            case 3: r = neg<2>(x); break;
            case 4: r = inv(x); break;
            case 5: r = sqr(x); break;
            case 6: { u32 w[8]; to_normal(w, x); r = from_normal<P>(w); break; }
            case 7: { u32 w[8]; Fp<P> c = canon(x); pack256(w, c); r = unpack256<P>(w); break; }
            case 8: r = inv_fermat(x); break;
            case 9: r = inv(sub<5>(x, fp_zero<P>())); break;
            default: return 1;
        }
        st(out, r);
    }
    return 0;
}

// long lazy chains: r = ((a*b + a - b)^2 * a - b) ... exercised `rounds` times, Fq
void ugt_fq_chain(uint8_t* out, const uint8_t* a, const uint8_t* b, int rounds) {
    Fq x = ld<FqParams>(a), y = ld<FqParams>(b);
    Fq acc = x;
    for (int i = 0; i < rounds; i++) {
        Fq t = add(mul(acc, y), x);          // < 3
        t = sub<2>(t, y);                    // < 5
        t = sqr(t);                          // < 2
        t = sub<2>(mul(t, x), y);            // < 4
        acc = add(t, dbl(acc));              // grows: acc < 4 + 2*acc_prev  -> contract again
        acc = mul(acc, fp_one<FqParams>());  // < 2
    }
    st(out, acc);
}
// Fq2: out = x*y, x^2, inv(x)  (components 32 bytes each: a then b)
int ugt_fq2_op(int op, uint8_t* out, const uint8_t* x_, const uint8_t* y_) {
    Fq2 x, y, r;
    x.a = ld<FqParams>(x_); x.b = ld<FqParams>(x_ + 32);
    if (y_) { y.a = ld<FqParams>(y_); y.b = ld<FqParams>(y_ + 32); } else y = f2_zero<FqParams>();
    switch (op) {
        case 0: r = mul(x, y); break;
        case 1: r = sqr(x); break;
        case 2: r = inv(x); break;
        case 3: r = sub<2>(add(x, y), y); break;
        default: return 1;
    }
    st(out, r.a); st(out + 32, r.b);
    return 0;
}

// sum of n affine points with optional signs (sign[i] != 0 -> subtract), via repeated mixed adds
void ugt_g1_sum(uint8_t out[64], const uint8_t* pts, const uint8_t* sign, size_t n) {
    G1XYZZ acc = xyzz_inf<Fq>();
    for (size_t i = 0; i < n; i++) {
        Fq x, y;
        if (!g1_load(pts + 64 * i, x, y)) continue;
        if (sign && sign[i]) y = neg<1>(y);
        acc = xyzz_madd(acc, x, y);
    }
    g1_store(out, acc);
}
void ugt_g2_sum(uint8_t out[128], const uint8_t* pts, const uint8_t* sign, size_t n) {
    G2XYZZ acc = xyzz_inf<Fq2>();
    for (size_t i = 0; i < n; i++) {
        Fq2 x, y;
        if (!g2_load(pts + 128 * i, x, y)) continue;
        if (sign && sign[i]) y = neg<1>(y);
        acc = xyzz_madd(acc, x, y);
    }
    g2_store(out, acc);
}
// tree sum with general XYZZ + XYZZ adds (exercises xyzz_add and its doubling branch)
void ugt_g1_tree_sum(uint8_t out[64], const uint8_t* pts, size_t n) {
    G1XYZZ* v = new G1XYZZ[n ? n : 1];
    for (size_t i = 0; i < n; i++) {
        Fq x, y;
        v[i] = g1_load(pts + 64 * i, x, y) ? xyzz_from_affine(x, y) : xyzz_inf<Fq>();
    }
    for (size_t w = n; w > 1; w = (w + 1) / 2)
        for (size_t i = 0; i < w / 2; i++) v[i] = xyzz_add(v[i], v[w - 1 - i]);
    g1_store(out, n ? v[0] : xyzz_inf<Fq>());
    delete[] v;
}
void ugt_g2_tree_sum(uint8_t out[128], const uint8_t* pts, size_t n) {
    G2XYZZ* v = new G2XYZZ[n ? n : 1];
    for (size_t i = 0; i < n; i++) {
        Fq2 x, y;
        v[i] = g2_load(pts + 128 * i, x, y) ? xyzz_from_affine(x, y) : xyzz_inf<Fq2>();
    }
    for (size_t w = n; w > 1; w = (w + 1) / 2)
        for (size_t i = 0; i < w / 2; i++) v[i] = xyzz_add(v[i], v[w - 1 - i]);
    g2_store(out, n ? v[0] : xyzz_inf<Fq2>());
    delete[] v;
}
void ugt_g1_mul(uint8_t out[64], const uint8_t base[64], const uint8_t scalar[32]) {
    Fq x, y; u32 k[8]; memcpy(k, scalar, 32);
    if (!g1_load(base, x, y)) { memset(out, 0, 64); return; }
    g1_store(out, xyzz_mul_scalar(xyzz_from_affine(x, y), k, 256));
}
void ugt_g2_mul(uint8_t out[128], const uint8_t base[128], const uint8_t scalar[32]) {
    Fq2 x, y; u32 k[8]; memcpy(k, scalar, 32);
    if (!g2_load(base, x, y)) { memset(out, 0, 128); return; }
    g2_store(out, xyzz_mul_scalar(xyzz_from_affine(x, y), k, 256));
}

// Shoup product of the NTT's twiddle multiplications (ff.hpp: mul_shoup, shoup_quotient): out = canonical((xa + xb + xc + xd) * w),
// everything plain 32-byte integers; the four addends are summed limb-wise WITHOUT a carry pass, the un-normalised form the
// radix-4 butterfly feeds into its second product. Also returns 1 when the raw result was below 3q, as the kernel's bounds assume.
int ugt_fr_mul_shoup(uint8_t out[32], const uint8_t xa[32], const uint8_t xb[32], const uint8_t xc[32], const uint8_t xd[32], const uint8_t w[32]) {
    u32 a[8], b[8], c[8], d[8], ww[8];
    memcpy(a, xa, 32); memcpy(b, xb, 32); memcpy(c, xc, 32); memcpy(d, xd, 32); memcpy(ww, w, 32);
    const Fr fa = unpack256<FrParams>(a), fb = unpack256<FrParams>(b), fc = unpack256<FrParams>(c), fd = unpack256<FrParams>(d);
    u32 x[NL];
    for (int i = 0; i < NL; i++) x[i] = fa.l[i] + fb.l[i] + fc.l[i] + fd.l[i];          // limbs up to 2^31
    const Fr wp = unpack256<FrParams>(ww);                                             // plain, < q
    const Fr wq = shoup_quotient(cond_sub_q(from_normal<FrParams>(ww)));
    const Fr r = mul_shoup<FrParams>(x, wp.l, wq.l);
    int below3q = 1;
    {   // r < 3q ?  compare strict limbs from the top with 3q
        u64 t[NL], carry = 0;
        for (int i = 0; i < NL; i++) { u64 v = (u64)FrParams::q[i] * 3 + carry; t[i] = i < NL - 1 ? (v & MASK29) : v; carry = i < NL - 1 ? (v >> LB) : 0; }
        for (int i = NL - 1; i >= 0; i--) { if (r.l[i] < t[i]) break; if (r.l[i] > t[i]) { below3q = 0; break; } if (i == 0) below3q = 0; }
    }
    u32 o[8];
    pack256(o, canon(r));
    memcpy(out, o, 32);
    return below3q;
}

// the windowed form the provers' host parts use (ec.hpp: xyzz_mul_scalar_w4)
void ugt_g1_mul_w4(uint8_t out[64], const uint8_t base[64], const uint8_t scalar[32]) {
    Fq x, y; u32 k[8]; memcpy(k, scalar, 32);
    if (!g1_load(base, x, y)) { memset(out, 0, 64); return; }
    g1_store(out, xyzz_mul_scalar_w4(xyzz_from_affine(x, y), k));
}
void ugt_g2_mul_w4(uint8_t out[128], const uint8_t base[128], const uint8_t scalar[32]) {
    Fq2 x, y; u32 k[8]; memcpy(k, scalar, 32);
    if (!g2_load(base, x, y)) { memset(out, 0, 128); return; }
    g2_store(out, xyzz_mul_scalar_w4(xyzz_from_affine(x, y), k));
}

// SegMap (segmap.hpp): every invariant the kernels of msm.hip rely on, for one (n_valid, log_a, log_b); 0 = all hold
int ugt_segmap_check(uint64_t n_valid, uint64_t total, int log_a, int log_b) {
    const SegMap m = SegMap::make(n_valid, log_a, log_b);
    if (m.seg0 % 64 || m.split % ((uint64_t)64 << log_a) || m.split > n_valid + ((uint64_t)64 << log_a)) return 1;
    if (log_b < log_a && m.split > n_valid) return 2;
    // segments tile [0, n_valid) in order, each entry belongs to the segment seg_of names, lengths as log_len says
    uint64_t pos = 0; uint32_t t = 0;
    while (pos < n_valid) {
        if (m.first_entry(t) != pos) return 3;
        const uint64_t len = (uint64_t)1 << m.log_len(t);
        if (m.log_len(t) != (t < m.seg0 ? log_a : log_b)) return 4;
        const uint64_t probes[3] = {pos, pos + len / 2, pos + len - 1};
        for (uint64_t p : probes) if (p < n_valid && m.seg_of((uint32_t)p) != t) return 5;
        // lane-transposed positions: inside the tile of the segment's wave, distinct for the 64 segments of the tile
        const uint64_t tile0 = m.first_entry(t & ~63u), tile_len = (uint64_t)64 << m.log_len(t);
        const uint32_t ks[2] = {0u, (uint32_t)(len - 1)};
        for (uint32_t k : ks) {
            const uint64_t q = m.transposed(t, k);
            if (q < tile0 || q >= tile0 + tile_len || (q - tile0) % 64 != (t & 63) || (q - tile0) / 64 != k) return 6;
        }
        pos += len; t++;
    }
    if (n_valid > total) return 7;
    if (t > SegMap::max_segments(total, log_a, log_b)) return 8;        // the host's grid / slot bound covers every segment
    if (m.first_entry(t) < n_valid) return 9;
    return 0;
}

}  // extern "C"
