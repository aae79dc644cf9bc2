// ec.hpp -- BN254 G1 / G2 group law on lazily reduced F29 coordinates (see ff.hpp).
//
// Replaces, on the prover hot path, the curve layer the reference gets from the un-vendored
// iden3/ffiasm submodule: Curve::add / sub / dbl / copy / mulByScalar and the point types
// {x,y} (affine, (0,0) = infinity) and {x,y,zz,zzz} (field names visible at
// src/groth16.cpp:379-410; call sites src/groth16.cpp:55-64,154,168-200).
// Curves: G1 y^2 = x^3 + 3 over Fq, G2 y^2 = x^3 + 3/(9+u) over Fq2 -- both a = 0.
//
// Coordinates are extended Jacobian (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), formulas from the EFD
// (madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1). Results are only ever compared or
// emitted after conversion to affine, which is canonical, so the choice of formulas cannot change
// any output bit.
//
// RANGE INVARIANT (in units of q, per Fq component) kept by every function below:
//     X < 7,   Y < 4,   ZZ < 2,   ZZZ < 2,    infinity  <=>  all limbs of ZZ are zero.
// Affine inputs are canonical (< 1). Each line notes the bound of what it produces; "K" template
// arguments are the multiples of q that keep subtractions positive (ff.hpp sub<K>/neg<K>/mulk<K>).
// tests/test_host_math.py builds this header with -DUG_CHECK_BOUNDS, which asserts every one.
#pragma once
#include "ff.hpp"

namespace ug {

template <class F> struct Affine { F x, y; };          // canonical components; inf handled by callers
template <class F> struct XYZZ { F x, y, zz, zzz; };

template <class P> UG_HD Fp<P> field_zero(const Fp<P>*) { return fp_zero<P>(); }
template <class P> UG_HD Fp2<P> field_zero(const Fp2<P>*) { return f2_zero<P>(); }
template <class P> UG_HD Fp<P> field_one(const Fp<P>*) { return fp_one<P>(); }
template <class P> UG_HD Fp2<P> field_one(const Fp2<P>*) { return f2_one<P>(); }

template <class F> UG_HD bool is_inf(const XYZZ<F>& p) { return limbs_all_zero(p.zz); }
template <class F> UG_HD XYZZ<F> xyzz_inf() {
    XYZZ<F> r;
    r.x = field_zero((F*)0); r.y = field_zero((F*)0); r.zz = field_zero((F*)0); r.zzz = field_zero((F*)0);
    return r;
}
template <class F> UG_HD XYZZ<F> xyzz_from_affine(const F& x, const F& y) {
    XYZZ<F> r;
    r.x = x; r.y = y; r.zz = field_one((F*)0); r.zzz = field_one((F*)0);
    return r;
}

// 2 * (x, y), affine in (mdbl-2008-s-1). BN254 has prime order: no point with y = 0.
template <class F> UG_HD XYZZ<F> xyzz_dbl_affine(const F& x, const F& y) {
    XYZZ<F> r;
    F u = dbl(y);                                   // < 2
    F v = sqrk<2>(u);                               // < 1.1
    F w = mulk<8>(u, v);                            // < 1.2
    F s = mulk<8>(x, v);                            // < 1.1
    F m = triple(sqrk<1>(x));                       // < 3.1
    r.x = add(sqrk<4>(m), neg<3>(dbl(s)));          // < 1.2 + 3 = 4.2
    F t = sub<5>(s, r.x);                           // < 6.1
    r.y = mul_subk<7, 1>(m, t, w, y);               // M t - W y: < 1.3
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * p (dbl-2008-s-1)
template <class F> UG_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    if (is_inf(p)) return p;
    XYZZ<F> r;
    F u = dbl(p.y);                                 // < 8
    F v = sqrk<8>(u);                               // < 1.76
    F w = mulk<8>(u, v);                            // < 1.5
    F s = mulk<8>(p.x, v);                          // < 1.41
    F m = triple(sqrk<7>(p.x));                     // < 3 * 1.58 = 4.74
    r.x = add(sqrk<5>(m), neg<3>(dbl(s)));          // < 1.3 + 3 = 4.3
    F t = sub<5>(s, r.x);                           // < 6.41
    r.y = mul_subk<7, 4>(m, t, w, p.y);             // M t - W Y: < (31 + 34 + 6 + 6)/170 + 1 = 1.5
    r.zz = mulk<8>(v, p.zz);
    r.zzz = mulk<8>(w, p.zzz);
    return r;
}

// p + (x2, y2), (x2, y2) affine and not infinity (madd-2008-s), all exceptional cases handled
template <class F> UG_HD XYZZ<F> xyzz_madd(const XYZZ<F>& p, const F& x2, const F& y2) {
    if (is_inf(p)) return xyzz_from_affine(x2, y2);
    F u2 = mulk<8>(x2, p.zz);                       // < 1.1
    F s2 = mulk<8>(y2, p.zzz);                      // < 1.1
    F pp_ = sub<7>(u2, p.x);                        // P  < 8.1
    F rr_ = sub<4>(s2, p.y);                        // R  < 5.1
    F pp = sqrk<9>(pp_);                            // PP < 1.92
    F r2 = sqrk<6>(rr_);                            // R^2 < 1.5
    if (is_zero_small(pp)) {
        if (is_zero_small(r2)) return xyzz_dbl_affine(x2, y2);
        return xyzz_inf<F>();
    }
    XYZZ<F> r;
    F ppp = mulk<8>(pp_, pp);                       // < 1.52
    F q = mulk<8>(p.x, pp);                         // < 1.41
    r.x = sub_b_2c_5q(r2, ppp, q);                  // R^2 - PPP - 2Q + 5q  < 1.5 + 5 = 6.5   (PPP + 2Q < 4.4)
    F t = sub<7>(q, r.x);                           // < 8.41
    r.y = mul_subk<9, 2>(rr_, t, p.y, ppp);         // R t - Y PPP, one reduction: < (43 + 46 + 8 + 6)/170 + 1 = 1.6
    r.zz = mulk<8>(p.zz, pp);
    r.zzz = mulk<8>(p.zzz, ppp);
    return r;
}

// p1 + p2 (add-2008-s), all exceptional cases handled
template <class F> UG_HD XYZZ<F> xyzz_add(const XYZZ<F>& p1, const XYZZ<F>& p2) {
    if (is_inf(p2)) return p1;
    if (is_inf(p1)) return p2;
    F u1 = mulk<8>(p1.x, p2.zz);                    // < 1.42
    F u2 = mulk<8>(p2.x, p1.zz);
    F s1 = mulk<8>(p1.y, p2.zzz);                   // < 1.24
    F s2 = mulk<8>(p2.y, p1.zzz);
    F pp_ = sub<2>(u2, u1);                         // P < 3.42
    F rr_ = sub<2>(s2, s1);                         // R < 3.24
    F pp = sqrk<4>(pp_);                            // < 1.15
    F r2 = sqrk<4>(rr_);                            // < 1.15
    if (is_zero_small(pp)) {
        if (is_zero_small(r2)) return xyzz_dbl(p1);
        return xyzz_inf<F>();
    }
    XYZZ<F> r;
    F ppp = mulk<8>(pp_, pp);                       // < 1.2
    F q = mulk<8>(u1, pp);                          // < 1.1
    r.x = sub_b_2c_5q(r2, ppp, q);                  // < 1.15 + 5 = 6.15   (PPP + 2Q < 3.4)
    F t = sub<7>(q, r.x);                           // < 8.1
    r.y = mul_subk<9, 2>(rr_, t, s1, ppp);          // R t - S1 PPP: < (27 + 30 + 3 + 2)/170 + 1 = 1.4
    r.zz = mulk<8>(mulk<8>(p1.zz, p2.zz), pp);
    r.zzz = mulk<8>(mulk<8>(p1.zzz, p2.zzz), ppp);
    return r;
}

template <class F> UG_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) {
    XYZZ<F> r = p;
    if (!is_inf(p)) r.y = sub<4>(field_zero((F*)0), p.y);   // 4q - Y  in (0, 4q)
    return r;
}

// canonical affine coordinates (device Montgomery form, each component in [0,q)); p must not be inf
template <class F> UG_HD void xyzz_to_affine(F& x, F& y, const XYZZ<F>& p) {
    F izzz = inv(p.zzz);                            // 1/zzz
    F iz = mulk<8>(izzz, p.zz);                     // zz/zzz = 1/z
    F izz = sqrk<8>(iz);                            // 1/zz
    x = canon(mulk<8>(p.x, izz));
    y = canon(mulk<8>(p.y, izzz));
}

// k * p, k = nbits-bit little-endian integer in 32-bit words (MSB-first double-and-add).
// Replaces Curve::mulByScalar for the seven blinding products of src/groth16.cpp:172-194.
template <class F> UG_HD XYZZ<F> xyzz_mul_scalar(const XYZZ<F>& p, const u32* k, int nbits) {
    XYZZ<F> acc = xyzz_inf<F>();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = xyzz_add(acc, p);
    }
    return acc;
}

// k * p for a 256-bit k on the host: signed 4-bit windows over the multiples 1..8 of p (256 doublings + 64 additions + 7 for the
// table, against 256 + ~128 for the bit-by-bit form of ec.hpp, which stays the device's). The seven products of S12 are what a
// proof's host part is made of once the device has answered (0.5 of 1.3 ms at 2^20, tools/phase_times.py `finish`). Host only.
template <class F> inline XYZZ<F> xyzz_mul_scalar_w4(const XYZZ<F>& p, const u32 k[8]) {
    if (is_inf(p)) return p;
    XYZZ<F> tab[8];
    tab[0] = p; tab[1] = xyzz_dbl(p);
    for (int i = 2; i < 8; i++) tab[i] = xyzz_add(tab[i - 1], p);
    int digit[65];
    int carry = 0;
    for (int w = 0; w < 64; w++) {
        int d = (int)((k[w >> 3] >> ((w & 7) * 4)) & 15) + carry;
        carry = d > 8;
        digit[w] = carry ? d - 16 : d;
    }
    digit[64] = carry;
    XYZZ<F> acc = xyzz_inf<F>();
    for (int w = 64; w >= 0; w--) {
        if (w != 64) for (int j = 0; j < 4; j++) acc = xyzz_dbl(acc);
        const int d = digit[w];
        if (d > 0) acc = xyzz_add(acc, tab[d - 1]);
        else if (d < 0) acc = xyzz_add(acc, xyzz_neg(tab[-d - 1]));
    }
    return acc;
}

typedef XYZZ<Fq> G1XYZZ;
typedef XYZZ<Fq2> G2XYZZ;

}  // namespace ug
