/*
 * oracle/field.h -- TEST INFRASTRUCTURE ONLY (CPU oracle), never linked into the product.
 *
 * Plain-C restatement of the reference's 4x64-bit Montgomery field layer for the
 * BN254 scalar field Fr and base field Fq:
 *   - CIOS Montgomery product, R = 2^256, one conditional subtract
 *       reference: build/fr_raw_generic.cpp:107-148 (Fr_rawMMul), build/fr.asm:372-538
 *   - add / sub / neg with the reference's reduction rules
 *       reference: build/fr_raw_generic.cpp:11-19 (rawAdd), :31-39 (rawSub), :68-80 (rawNeg)
 *   - toMontgomery = MMul by R^2, fromMontgomery = MMul by 1
 *       reference: build/fr_raw_generic.cpp:192-232
 *   - constants q, R^2, np: build/fr_raw_generic.cpp:5-8, build/fq_raw_generic.cpp:5-8;
 *     R^3: build/fr_generic.cpp:7, build/fq_generic.cpp:7
 * build/fq*.cpp is build/fr*.cpp with the names and constants swapped, so one
 * parameterised implementation (struct fctx) serves both.
 */
#ifndef UGO_FIELD_H
#define UGO_FIELD_H

#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 v[4]; } fe;          /* one field element, little-endian limbs */

typedef struct fctx {
    u64 q[4];      /* modulus                        */
    u64 r2[4];     /* R^2 mod q                      */
    u64 r3[4];     /* R^3 mod q                      */
    u64 one[4];    /* R mod q  (Montgomery form of 1) */
    u64 np;        /* -q^-1 mod 2^64                 */
} fctx;

static const fctx UGO_FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
    {0x5e94d8e1b4bf0040ULL, 0x2a489cbe1cfbb6b8ULL, 0x893cc664a19fcfedULL, 0x0cf8594b7fcc657cULL},
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    0xc2e1f593efffffffULL
};

static const fctx UGO_FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
    {0xb1cd6dafda1530dfULL, 0x62f210e6a7283db6ULL, 0xef7f0b0c0ada0afbULL, 0x20fd6e902d592544ULL},
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    0x87d20782e4866389ULL
};

static inline int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
static inline void fe_zero(fe *a) { memset(a, 0, sizeof *a); }

/* a >= q ? */
static inline int fe_geq_q(const u64 a[4], const fctx *f) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > f->q[i]) return 1;
        if (a[i] < f->q[i]) return 0;
    }
    return 1;
}
static inline u64 limbs_sub(u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static inline u64 limbs_add(u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a[i] + b[i] + carry;
        r[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
    return carry;
}

/* reference: Fr_rawAdd -- add, then subtract q when carry out or result >= q */
static inline void fe_add(fe *r, const fe *a, const fe *b, const fctx *f) {
    u64 t[4];
    u64 carry = limbs_add(t, a->v, b->v);
    if (carry || fe_geq_q(t, f)) limbs_sub(t, t, f->q);
    memcpy(r->v, t, 32);
}
/* reference: Fr_rawSub -- subtract, add q back on borrow */
static inline void fe_sub(fe *r, const fe *a, const fe *b, const fctx *f) {
    u64 t[4];
    if (limbs_sub(t, a->v, b->v)) limbs_add(t, t, f->q);
    memcpy(r->v, t, 32);
}
/* reference: Fr_rawNeg -- 0 stays 0 */
static inline void fe_neg(fe *r, const fe *a, const fctx *f) {
    if (fe_is_zero(a)) { fe_zero(r); return; }
    u64 t[4];
    limbs_sub(t, f->q, a->v);
    memcpy(r->v, t, 32);
}
static inline void fe_dbl(fe *r, const fe *a, const fctx *f) { fe_add(r, a, a, f); }

/* reference: Fr_rawMMul -- interleaved (CIOS) Montgomery product, portable form */
static inline void fe_mul_c(fe *r, const fe *a, const fe *b, const fctx *f) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5;
    const u64 *q = f->q;
    for (int i = 0; i < 4; i++) {
        u64 ai = a->v[i];
        u128 c;
        c = (u128)ai * b->v[0] + t0;             t0 = (u64)c; c >>= 64;
        c += (u128)ai * b->v[1] + t1;            t1 = (u64)c; c >>= 64;
        c += (u128)ai * b->v[2] + t2;            t2 = (u64)c; c >>= 64;
        c += (u128)ai * b->v[3] + t3;            t3 = (u64)c; c >>= 64;
        c += t4;                                 t4 = (u64)c; t5 = (u64)(c >> 64);
        u64 m = t0 * f->np;
        c = (u128)m * q[0] + t0;                 c >>= 64;
        c += (u128)m * q[1] + t1;                t0 = (u64)c; c >>= 64;
        c += (u128)m * q[2] + t2;                t1 = (u64)c; c >>= 64;
        c += (u128)m * q[3] + t3;                t2 = (u64)c; c >>= 64;
        c += t4;                                 t3 = (u64)c; t4 = t5 + (u64)(c >> 64);
    }
    u64 t[4] = {t0, t1, t2, t3};
    if (t4 || fe_geq_q(t, f)) limbs_sub(t, t, q);
    memcpy(r->v, t, 32);
}
#if defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__)
/* The same product the way the reference's x86_64 back-end computes it (build/fr.asm:372-538 is mulx with the adcx / adox
 * carry chains): own code, one asm block per CIOS round -- multiply-accumulate a[i] * b on the two chains, m = t0 * np,
 * multiply-accumulate m * q, drop the low word. q < 2^254, so the running top word cannot overflow. The CPU baseline of
 * bench.py runs on this form, so that "port" is not softer than the assembly it stands for (measured on the build host,
 * Xeon 2.1 GHz: 20.3 ns per product against 27-31 ns for the portable form; tests/test_oracle.py checks both against the
 * reference's own field layer and against each other).
 * The single running top word holds as long as b < 2^254 (then t + a_i b + m q < 2^319); an UNREDUCED second operand -- the
 * prover multiplies zkey coefficients as they come, up to 2^256 - 1 -- takes the portable form, which carries a sixth word. */
static inline void fe_mul(fe *r, const fe *a, const fe *b, const fctx *f) {
    if (b->v[3] >> 62) { fe_mul_c(r, a, b, f); return; }
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4, lo, hi;
    const u64 *A = a->v, *B = b->v, *Q = f->q;
    const u64 np = f->np;
#define UGO_ROUND(I)                                                                                   \
    __asm__("xorl %k[t4], %k[t4]\n\t"         /* t4 = 0, CF = OF = 0 */                                 \
            "movq %[ai], %%rdx\n\t"                                                                     \
            "mulxq 0(%[b]), %[lo], %[hi]\n\t adoxq %[lo], %[t0]\n\t adcxq %[hi], %[t1]\n\t"              \
            "mulxq 8(%[b]), %[lo], %[hi]\n\t adoxq %[lo], %[t1]\n\t adcxq %[hi], %[t2]\n\t"              \
            "mulxq 16(%[b]), %[lo], %[hi]\n\t adoxq %[lo], %[t2]\n\t adcxq %[hi], %[t3]\n\t"             \
            "mulxq 24(%[b]), %[lo], %[hi]\n\t adoxq %[lo], %[t3]\n\t adcxq %[hi], %[t4]\n\t"             \
            "movl $0, %k[lo]\n\t adoxq %[lo], %[t4]\n\t"                                                 \
            "movq %[t0], %%rdx\n\t imulq %[np], %%rdx\n\t"                                               \
            "xorl %k[lo], %k[lo]\n\t"         /* CF = OF = 0 */                                         \
            "mulxq 0(%[q]), %[lo], %[hi]\n\t adoxq %[lo], %[t0]\n\t adcxq %[hi], %[t1]\n\t"              \
            "mulxq 8(%[q]), %[lo], %[hi]\n\t adoxq %[lo], %[t1]\n\t adcxq %[hi], %[t2]\n\t"              \
            "mulxq 16(%[q]), %[lo], %[hi]\n\t adoxq %[lo], %[t2]\n\t adcxq %[hi], %[t3]\n\t"             \
            "mulxq 24(%[q]), %[lo], %[hi]\n\t adoxq %[lo], %[t3]\n\t adcxq %[hi], %[t4]\n\t"             \
            "movl $0, %k[lo]\n\t adoxq %[lo], %[t4]\n\t"                                                 \
            : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "=&r"(t4), [lo] "=&r"(lo), [hi] "=&r"(hi)  \
            : [ai] "m"(A[I]), [b] "r"(B), [q] "r"(Q), [np] "r"(np)                                        \
            : "rdx", "cc", "memory");                                                                    \
    t0 = t1; t1 = t2; t2 = t3; t3 = t4;
    UGO_ROUND(0) UGO_ROUND(1) UGO_ROUND(2) UGO_ROUND(3)
#undef UGO_ROUND
    u64 t[4] = {t0, t1, t2, t3};
    if (fe_geq_q(t, f)) limbs_sub(t, t, Q);
    memcpy(r->v, t, 32);
}
#else
static inline void fe_mul(fe *r, const fe *a, const fe *b, const fctx *f) { fe_mul_c(r, a, b, f); }
#endif
static inline void fe_sqr(fe *r, const fe *a, const fctx *f) { fe_mul(r, a, a, f); }

static inline void fe_to_mont(fe *r, const fe *a, const fctx *f) {
    fe r2; memcpy(r2.v, f->r2, 32);
    fe_mul(r, a, &r2, f);
}
static inline void fe_from_mont(fe *r, const fe *a, const fctx *f) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(r, a, &one, f);
}
static inline void fe_set_one(fe *r, const fctx *f) { memcpy(r->v, f->one, 32); }

/* a^e, e given as 4 little-endian limbs (plain integer), a and result in Montgomery form */
static inline void fe_pow(fe *r, const fe *a, const u64 e[4], const fctx *f) {
    fe acc; fe_set_one(&acc, f);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) fe_sqr(&acc, &acc, f);
        if ((e[i >> 6] >> (i & 63)) & 1) {
            if (started) fe_mul(&acc, &acc, a, f); else { acc = *a; started = 1; }
        }
    }
    *r = acc;
}
/* Montgomery in, Montgomery out (same contract as RawFr::inv, build/fr.cpp:238-250);
 * computed as a^(q-2) instead of GMP mpz_invert: the value is unique, so identical. */
static inline void fe_inv(fe *r, const fe *a, const fctx *f) {
    u64 e[4]; u64 two[4] = {2, 0, 0, 0};
    limbs_sub(e, f->q, two);
    fe_pow(r, a, e, f);
}

#endif
