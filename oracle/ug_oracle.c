/*
 * oracle/ug_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + OpenMP) of the Groth16 / UltraGroth prover hot path of
 * rarimo/ultragroth, used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker / reported baseline. The product (ultragroth_amd/csrc) never links,
 * loads or calls anything in this directory.
 *
 * Parity status: PINNED for Groth16 by
 *   (1) the reference's own fixture testdata/{circuit_final.zkey,witness.wtns,
 *       verification_key.json}: the proof this oracle emits verifies against the reference's
 *       verification key (the reference's only acceptance test, .github/workflows/build.yml:69-81);
 *   (2) the known answers of SURVEY.md Appendix A (omega, h[], raw MSMs, proof.json sha256);
 *   (3) the reference's own field layer compiled here into oracle/_ref (build/f{r,q}*.cpp).
 * UltraGroth: PARITY UNPINNED -- the reference ships no UltraGroth fixture or test
 * (SURVEY.md section 4); the restatement below follows src/ultra_groth.cpp line by line.
 *
 * What follows what:
 *   binfile container          src/binfile_utils.cpp:32-80
 *   zkey header (groth16)      src/zkey_utils.cpp:42-76     (ultragroth: :123-163)
 *   wtns header                src/wtns_utils.cpp:13-26
 *   prove() S1..S13            src/groth16.cpp:48-203
 *   proof / public JSON        src/groth16.cpp:217-250, src/prover.cpp:106-117
 *   FFT<Fr> (fft, ifft, root)  depends/ffiasm (absent): radix-2, omega_{2^s} = 5^((r-1)/2^s),
 *                              ifft scales by 1/n  (conventions pinned by SURVEY.md Appendix A)
 *   UltraGroth                 src/ultra_groth.cpp:24-462, src/keccak256.cpp
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "field.h"
#include "ug_oracle.h"

/* ------------------------------------------------------------------ Fq / Fq2 glue */

static inline void q_mul(fe *r, const fe *a, const fe *b) { fe_mul(r, a, b, &UGO_FQ); }
static inline void q_sqr(fe *r, const fe *a) { fe_mul(r, a, a, &UGO_FQ); }
static inline void q_add(fe *r, const fe *a, const fe *b) { fe_add(r, a, b, &UGO_FQ); }
static inline void q_sub(fe *r, const fe *a, const fe *b) { fe_sub(r, a, b, &UGO_FQ); }
static inline void q_neg(fe *r, const fe *a) { fe_neg(r, a, &UGO_FQ); }
static inline void q_dbl(fe *r, const fe *a) { fe_add(r, a, a, &UGO_FQ); }
static inline void q_one(fe *r) { fe_set_one(r, &UGO_FQ); }
static inline void q_inv(fe *r, const fe *a) { fe_inv(r, a, &UGO_FQ); }

/* Fq2 = Fq[u]/(u^2+1); element a + b*u stored (a, b) -- the order of the zkey G2 records
 * (x.a, x.b, y.a, y.b) and of the proof JSON (src/groth16.cpp:228-233). */
typedef struct { fe a, b; } fe2;
static inline int f2_is_zero(const fe2 *x) { return fe_is_zero(&x->a) && fe_is_zero(&x->b); }
static inline void f2_zero(fe2 *x) { fe_zero(&x->a); fe_zero(&x->b); }
static inline void f2_one(fe2 *x) { q_one(&x->a); fe_zero(&x->b); }
static inline void f2_add(fe2 *r, const fe2 *x, const fe2 *y) { q_add(&r->a, &x->a, &y->a); q_add(&r->b, &x->b, &y->b); }
static inline void f2_sub(fe2 *r, const fe2 *x, const fe2 *y) { q_sub(&r->a, &x->a, &y->a); q_sub(&r->b, &x->b, &y->b); }
static inline void f2_neg(fe2 *r, const fe2 *x) { q_neg(&r->a, &x->a); q_neg(&r->b, &x->b); }
static inline void f2_dbl(fe2 *r, const fe2 *x) { f2_add(r, x, x); }
static inline void f2_mul(fe2 *r, const fe2 *x, const fe2 *y) {
    fe aa, bb, s, t, m;
    q_mul(&aa, &x->a, &y->a);
    q_mul(&bb, &x->b, &y->b);
    q_add(&s, &x->a, &x->b);
    q_add(&t, &y->a, &y->b);
    q_mul(&m, &s, &t);
    q_sub(&m, &m, &aa);
    q_sub(&m, &m, &bb);
    q_sub(&r->a, &aa, &bb);
    r->b = m;
}
static inline void f2_sqr(fe2 *r, const fe2 *x) {
    fe s, d, m;
    q_add(&s, &x->a, &x->b);
    q_sub(&d, &x->a, &x->b);
    q_mul(&m, &x->a, &x->b);
    q_mul(&r->a, &s, &d);
    q_dbl(&r->b, &m);
}
static inline void f2_inv(fe2 *r, const fe2 *x) {
    fe n, t, ni;
    q_sqr(&n, &x->a); q_sqr(&t, &x->b); q_add(&n, &n, &t);
    q_inv(&ni, &n);
    q_mul(&r->a, &x->a, &ni);
    q_mul(&t, &x->b, &ni);
    q_neg(&r->b, &t);
}

/* ------------------------------------------------------------------ curve instances */

#define T fe
#define N(name) g1_##name
#define F_mul q_mul
#define F_sqr q_sqr
#define F_add q_add
#define F_sub q_sub
#define F_neg q_neg
#define F_dbl q_dbl
#define F_is_zero fe_is_zero
#define F_set_zero fe_zero
#define F_set_one q_one
#define F_inv q_inv
#include "curve_tmpl.inc"
#undef T
#undef N
#undef F_mul
#undef F_sqr
#undef F_add
#undef F_sub
#undef F_neg
#undef F_dbl
#undef F_is_zero
#undef F_set_zero
#undef F_set_one
#undef F_inv

#define T fe2
#define N(name) g2_##name
#define F_mul f2_mul
#define F_sqr f2_sqr
#define F_add f2_add
#define F_sub f2_sub
#define F_neg f2_neg
#define F_dbl f2_dbl
#define F_is_zero f2_is_zero
#define F_set_zero f2_zero
#define F_set_one f2_one
#define F_inv f2_inv
#include "curve_tmpl.inc"

/* ------------------------------------------------------------------ exported field ops */

static const fctx *pick(int which) { return which == UGO_FIELD_FQ ? &UGO_FQ : &UGO_FR; }

void ugo_f_mul(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_mul(&z, &x, &y, pick(which)); memcpy(r, &z, 32);
}
/* the portable form of the product (field.h: fe_mul_c), for the cross-check of the two in tests/test_oracle.py */
void ugo_f_mul_portable(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_mul_c(&z, &x, &y, pick(which)); memcpy(r, &z, 32);
}
void ugo_f_add(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_add(&z, &x, &y, pick(which)); memcpy(r, &z, 32);
}
void ugo_f_sub(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_sub(&z, &x, &y, pick(which)); memcpy(r, &z, 32);
}
void ugo_f_neg(int which, uint64_t *r, const uint64_t *a) {
    fe x, z; memcpy(&x, a, 32); fe_neg(&z, &x, pick(which)); memcpy(r, &z, 32);
}
void ugo_f_to_mont(int which, uint64_t *r, const uint64_t *a) {
    fe x, z; memcpy(&x, a, 32); fe_to_mont(&z, &x, pick(which)); memcpy(r, &z, 32);
}
void ugo_f_from_mont(int which, uint64_t *r, const uint64_t *a) {
    fe x, z; memcpy(&x, a, 32); fe_from_mont(&z, &x, pick(which)); memcpy(r, &z, 32);
}
void ugo_f_inv(int which, uint64_t *r, const uint64_t *a) {
    fe x, z; memcpy(&x, a, 32); fe_inv(&z, &x, pick(which)); memcpy(r, &z, 32);
}
/* vector forms used by the parity tests (n elements of 32 bytes) */
void ugo_f_mul_vec(int which, uint64_t *r, const uint64_t *a, const uint64_t *b, size_t n) {
    for (size_t i = 0; i < n; i++) ugo_f_mul(which, r + 4 * i, a + 4 * i, b + 4 * i);
}

/* ------------------------------------------------------------------ exported curve ops */

void ugo_g1_msm(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, size_t n) {
    g1_xyzz acc; g1_aff a;
    g1_msm(&acc, bases, scalars, 32, n);
    g1_to_aff(&a, &acc); memcpy(out, &a, 64);
}
void ugo_g1_msm_naive(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, size_t n) {
    g1_xyzz acc; g1_aff a;
    g1_msm_naive(&acc, bases, scalars, 32, n);
    g1_to_aff(&a, &acc); memcpy(out, &a, 64);
}
void ugo_g2_msm(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, size_t n) {
    g2_xyzz acc; g2_aff a;
    g2_msm(&acc, bases, scalars, 32, n);
    g2_to_aff(&a, &acc); memcpy(out, &a, 128);
}
void ugo_g2_msm_naive(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, size_t n) {
    g2_xyzz acc; g2_aff a;
    g2_msm_naive(&acc, bases, scalars, 32, n);
    g2_to_aff(&a, &acc); memcpy(out, &a, 128);
}
void ugo_g1_mul(uint8_t out[64], const uint8_t base[64], const uint8_t scalar[32]) {
    g1_aff b, a; g1_xyzz p, r;
    memcpy(&b, base, 64); g1_from_aff(&p, &b);
    g1_mul_scalar(&r, &p, scalar, 32);
    g1_to_aff(&a, &r); memcpy(out, &a, 64);
}
void ugo_g2_mul(uint8_t out[128], const uint8_t base[128], const uint8_t scalar[32]) {
    g2_aff b, a; g2_xyzz p, r;
    memcpy(&b, base, 128); g2_from_aff(&p, &b);
    g2_mul_scalar(&r, &p, scalar, 32);
    g2_to_aff(&a, &r); memcpy(out, &a, 128);
}
void ugo_g1_add(uint8_t out[64], const uint8_t p[64], const uint8_t q[64]) {
    g1_aff a, b, c; g1_xyzz x;
    memcpy(&a, p, 64); memcpy(&b, q, 64);
    g1_from_aff(&x, &a); g1_add_mixed(&x, &x, &b);
    g1_to_aff(&c, &x); memcpy(out, &c, 64);
}
void ugo_g2_add(uint8_t out[128], const uint8_t p[128], const uint8_t q[128]) {
    g2_aff a, b, c; g2_xyzz x;
    memcpy(&a, p, 128); memcpy(&b, q, 128);
    g2_from_aff(&x, &a); g2_add_mixed(&x, &x, &b);
    g2_to_aff(&c, &x); memcpy(out, &c, 128);
}
/* y^2 == x^3 + b ?  (b = 3 for G1, 3/(9+u) for G2); infinity counts as on-curve */
int ugo_g1_on_curve(const uint8_t p[64]) {
    g1_aff a; memcpy(&a, p, 64);
    if (g1_aff_is_inf(&a)) return 1;
    fe three = {{3, 0, 0, 0}}, b, l, r;
    fe_to_mont(&b, &three, &UGO_FQ);
    q_sqr(&l, &a.y);
    q_sqr(&r, &a.x); q_mul(&r, &r, &a.x); q_add(&r, &r, &b);
    return fe_eq(&l, &r);
}
int ugo_g2_on_curve(const uint8_t p[128]) {
    g2_aff a; memcpy(&a, p, 128);
    if (g2_aff_is_inf(&a)) return 1;
    fe three = {{3, 0, 0, 0}}, nine = {{9, 0, 0, 0}};
    fe2 xi, b, l, r;
    fe_to_mont(&xi.a, &nine, &UGO_FQ); q_one(&xi.b);          /* 9 + u */
    f2_inv(&b, &xi);
    fe t3; fe_to_mont(&t3, &three, &UGO_FQ);
    q_mul(&b.a, &b.a, &t3); q_mul(&b.b, &b.b, &t3);           /* 3/(9+u) */
    f2_sqr(&l, &a.y);
    f2_sqr(&r, &a.x); f2_mul(&r, &r, &a.x); f2_add(&r, &r, &b);
    return fe_eq(&l.a, &r.a) && fe_eq(&l.b, &r.b);
}

/* ------------------------------------------------------------------ NTT over Fr */

static inline void r_mul(fe *r, const fe *a, const fe *b) { fe_mul(r, a, b, &UGO_FR); }
static inline void r_add(fe *r, const fe *a, const fe *b) { fe_add(r, a, b, &UGO_FR); }
static inline void r_sub(fe *r, const fe *a, const fe *b) { fe_sub(r, a, b, &UGO_FR); }

/* omega_{2^s} = 5^((r-1)/2^s), Montgomery form (5 = smallest quadratic non-residue of Fr) */
void ugo_fr_root_of_unity(uint64_t out[4], int s) {
    u64 e[4]; u64 one[4] = {1, 0, 0, 0};
    limbs_sub(e, UGO_FR.q, one);
    for (int k = 0; k < s; k++) {               /* e >>= 1 */
        for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
    }
    fe five = {{5, 0, 0, 0}}, g, w;
    fe_to_mont(&g, &five, &UGO_FR);
    fe_pow(&w, &g, e, &UGO_FR);
    memcpy(out, &w, 32);
}

static unsigned bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* t[i] = w^i for i < count (Montgomery form): blocks of 2^12 entries, each started with one exponentiation, in parallel */
static void power_table(fe *t, const fe *w, size_t count) {
    const size_t B = (size_t)1 << 12;
    size_t nblocks = (count + B - 1) / B;
#pragma omp parallel for schedule(static) if (count >= 2 * B)
    for (size_t blk = 0; blk < nblocks; blk++) {
        size_t i0 = blk * B, i1 = i0 + B < count ? i0 + B : count;
        u64 e[4] = {(u64)i0, 0, 0, 0};
        if (i0 == 0) fe_set_one(&t[0], &UGO_FR); else fe_pow(&t[i0], w, e, &UGO_FR);
        for (size_t i = i0 + 1; i < i1; i++) r_mul(&t[i], &t[i - 1], w);
    }
}

/* In-place radix-2 transform, natural order in and out.
 * inverse = 0:  X[k] = sum_j x[j] w^(jk),  w = omega_n
 * inverse = 1:  x[j] = n^-1 sum_k X[k] w^(-jk)      (reference: fft->fft / fft->ifft,
 *               src/groth16.cpp:112,120) */
void ugo_fr_ntt(uint64_t *data, int logn, int inverse) {
    size_t n = (size_t)1 << logn;
    fe *x = (fe *)data;
    fe w; ugo_fr_root_of_unity(w.v, logn);
    if (inverse) fe_inv(&w, &w, &UGO_FR);
    /* twiddles w^0 .. w^(n/2-1) */
    size_t half = n / 2 ? n / 2 : 1;
    fe *tw = (fe *)malloc(half * sizeof(fe));
    power_table(tw, &w, half);
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t i = 0; i < n; i++) {            /* each pair is swapped by the thread that owns its smaller index */
        size_t j = bitrev((unsigned)i, logn);
        if (i < j) { fe t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (int s = 0; s < logn; s++) {
        size_t m = (size_t)1 << s;              /* half-block */
        size_t step = half >> s;
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t k = 0; k < n / 2; k++) {
            size_t blk = k >> s, j = k & (m - 1);
            size_t i0 = (blk << (s + 1)) + j, i1 = i0 + m;
            fe t, u = x[i0];
            r_mul(&t, &x[i1], &tw[j * step]);
            r_add(&x[i0], &u, &t);
            r_sub(&x[i1], &u, &t);
        }
    }
    if (inverse) {
        fe nn = {{(u64)n, 0, 0, 0}}, ninv;
        fe_to_mont(&nn, &nn, &UGO_FR);
        fe_inv(&ninv, &nn, &UGO_FR);
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t i = 0; i < n; i++) r_mul(&x[i], &x[i], &ninv);
    }
    free(tw);
}

/* ------------------------------------------------------------------ H polynomial (S5..S9) */

#pragma pack(push, 1)
typedef struct { uint32_t m, c, s; fe coef; } coef_rec;   /* src/groth16.hpp:41-49, 44 bytes */
#pragma pack(pop)

/* coefs: nCoefs packed 44-byte records (pointer already past the 4-byte count prefix,
 * src/groth16.cpp:38); wtns: nVars normal-form values; out h: N normal-form values.
 * Returns 0, or 1 when a record indexes outside [0,N) x [0,nVars). */
int ugo_hpoly(uint64_t *h_out, const uint8_t *coefs, uint64_t ncoefs, const uint8_t *wtns,
              uint32_t nvars, uint32_t domain_size, uint64_t *abc_coset_out) {
    size_t n = domain_size;
    int logn = 0; while (((size_t)1 << logn) < n) logn++;
    fe *a = (fe *)calloc(n, sizeof(fe)), *b = (fe *)calloc(n, sizeof(fe)), *c = (fe *)malloc(n * sizeof(fe));
    /* S6: scatter-add under striped locks, as the reference does it (src/groth16.cpp:70-99: NLOCKS = 1024 mutexes,
     * lock[c % NLOCKS]); the sum is order-independent */
    int bad = 0;
#ifdef _OPENMP
    enum { NLOCKS = 1024 };
    static omp_lock_t locks[NLOCKS];
    static int locks_ready = 0;
#pragma omp critical(ugo_locks_init)
    if (!locks_ready) { for (int i = 0; i < NLOCKS; i++) omp_init_lock(&locks[i]); locks_ready = 1; }
#endif
#pragma omp parallel for schedule(static) reduction(|:bad) if (ncoefs >= 4096)
    for (uint64_t i = 0; i < ncoefs; i++) {
        coef_rec rec; memcpy(&rec, coefs + i * 44, 44);
        if (rec.c >= n || rec.s >= nvars) { bad |= 1; continue; }
        fe w, aux; memcpy(&w, wtns + (size_t)rec.s * 32, 32);
        r_mul(&aux, &w, &rec.coef);
        fe *ab = rec.m == 0 ? a : b;
#ifdef _OPENMP
        omp_set_lock(&locks[rec.c % NLOCKS]);
#endif
        r_add(&ab[rec.c], &ab[rec.c], &aux);
#ifdef _OPENMP
        omp_unset_lock(&locks[rec.c % NLOCKS]);
#endif
    }
    if (bad) { free(a); free(b); free(c); return 1; }
    /* S7 */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) r_mul(&c[i], &a[i], &b[i]);
    /* S8: ifft, twist by omega_{2n}^i, fft */
    fe w2n; ugo_fr_root_of_unity(w2n.v, logn + 1);
    fe *tw = (fe *)malloc(n * sizeof(fe));
    power_table(tw, &w2n, n);
    fe *polys[3] = {a, b, c};
    for (int p = 0; p < 3; p++) {
        fe *x = polys[p];
        ugo_fr_ntt((uint64_t *)x, logn, 1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) r_mul(&x[i], &x[i], &tw[i]);
        ugo_fr_ntt((uint64_t *)x, logn, 0);
    }
    if (abc_coset_out) {
        memcpy(abc_coset_out, a, n * 32);
        memcpy(abc_coset_out + 4 * n, b, n * 32);
        memcpy(abc_coset_out + 8 * n, c, n * 32);
    }
    /* S9 */
    fe *h = (fe *)h_out;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fe t;
        r_mul(&t, &a[i], &b[i]);
        r_sub(&t, &t, &c[i]);
        fe_from_mont(&h[i], &t, &UGO_FR);
    }
    free(a); free(b); free(c); free(tw);
    return 0;
}

/* ------------------------------------------------------------------ binfile / zkey / wtns */

typedef struct { const uint8_t *p; uint64_t size; } section;

#define MAX_SEC 32
typedef struct { section s[MAX_SEC]; int present[MAX_SEC]; } binfile;

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

static int seterr(char *err, size_t errsz, const char *msg) {
    if (err && errsz) { strncpy(err, msg, errsz - 1); err[errsz - 1] = 0; }
    return 1;
}

/* src/binfile_utils.cpp:32-80 */
static int binfile_parse(binfile *bf, const uint8_t *buf, uint64_t size, const char *type,
                         uint32_t max_version, char *err, size_t errsz) {
    memset(bf, 0, sizeof *bf);
    if (size < 12) return seterr(err, errsz, "File is too short.");
    if (memcmp(buf, type, 4) != 0) return seterr(err, errsz, "Invalid file type.");
    if (rd32(buf + 4) > max_version) return seterr(err, errsz, "Invalid version.");
    uint32_t ns = rd32(buf + 8);
    if (size < 12 + (uint64_t)ns * 12) return seterr(err, errsz, "File is too short to contain sections.");
    uint64_t pos = 12;
    for (uint32_t i = 0; i < ns; i++) {
        if (pos + 12 > size) return seterr(err, errsz, "File pos is too big.");
        uint32_t t = rd32(buf + pos); uint64_t sz = rd64(buf + pos + 4);
        pos += 12;
        if (t < MAX_SEC && !bf->present[t]) { bf->s[t].p = buf + pos; bf->s[t].size = sz; bf->present[t] = 1; }
        pos += sz;
        if (pos > size) return seterr(err, errsz, "Section is invalid.");
    }
    return 0;
}

typedef struct {
    uint32_t n8q, n8r, nvars, npublic, domain;
    uint64_t ncoefs;
    const uint8_t *alpha1, *beta1, *beta2, *gamma2, *delta1, *delta2;
    /* ultragroth */
    uint32_t n_idx_c1, n_idx_c2, rand_indx;
    const uint8_t *round_delta1, *round_delta2;
} zkey_hdr;

static const uint8_t FR_PRIME_LE[32] = {
    0x01,0x00,0x00,0xf0,0x93,0xf5,0xe1,0x43,0x91,0x70,0xb9,0x79,0x48,0xe8,0x33,0x28,
    0x5d,0x58,0x81,0x81,0xb6,0x45,0x50,0xb8,0x29,0xa0,0x31,0xe1,0x72,0x4e,0x64,0x30};

/* src/zkey_utils.cpp:42-76 (protocol 1) and :123-163 (protocol 1337) */
static int zkey_header(zkey_hdr *h, const binfile *bf, int ultra, char *err, size_t errsz) {
    memset(h, 0, sizeof *h);
    if (!bf->present[1] || !bf->present[2] || !bf->present[4]) return seterr(err, errsz, "Section does not exist");
    uint32_t proto = rd32(bf->s[1].p);
    if (!ultra && proto != 1) return seterr(err, errsz, "zkey file is not groth16");
    if (ultra && proto != 1337) return seterr(err, errsz, "zkey file is not ultragroth");
    const uint8_t *p = bf->s[2].p;
    h->n8q = rd32(p); p += 4; p += h->n8q;
    h->n8r = rd32(p); p += 4;
    if (h->n8q != 32 || h->n8r != 32 || memcmp(p, FR_PRIME_LE, 32) != 0) return seterr(err, errsz, "zkey curve not supported");
    p += h->n8r;
    h->nvars = rd32(p); p += 4;
    h->npublic = rd32(p); p += 4;
    h->domain = rd32(p); p += 4;
    if (ultra) {
        h->n_idx_c1 = rd32(p); p += 4;
        h->n_idx_c2 = rd32(p); p += 4;
        h->rand_indx = rd32(p); p += 4;
    }
    h->alpha1 = p; p += 64;
    h->beta1 = p; p += 64;
    h->beta2 = p; p += 128;
    h->gamma2 = p; p += 128;
    if (ultra) {
        h->round_delta1 = p; p += 64;
        h->round_delta2 = p; p += 128;
    }
    h->delta1 = p; p += 64;      /* ultragroth: final_delta1 */
    h->delta2 = p; p += 128;     /* ultragroth: final_delta2 */
    h->ncoefs = bf->s[4].size / 44;
    return 0;
}

int ugo_zkey_info(const uint8_t *zkey, uint64_t size, uint32_t out[4], uint64_t *ncoefs) {
    binfile bf; zkey_hdr h; char e[64];
    if (binfile_parse(&bf, zkey, size, "zkey", 1, e, sizeof e)) return 1;
    int ultra = bf.present[1] && rd32(bf.s[1].p) == 1337;
    if (zkey_header(&h, &bf, ultra, e, sizeof e)) return 1;
    out[0] = h.nvars; out[1] = h.npublic; out[2] = h.domain; out[3] = ultra;
    *ncoefs = h.ncoefs;
    return 0;
}
/* offset and size of a zkey/wtns section inside the buffer (for tests that slice fixtures) */
int ugo_section(const uint8_t *buf, uint64_t size, const char *type, uint32_t id, uint64_t *off, uint64_t *sz) {
    binfile bf; char e[64];
    if (binfile_parse(&bf, buf, size, type, 2, e, sizeof e)) return 1;
    if (id >= MAX_SEC || !bf.present[id]) return 1;
    *off = (uint64_t)(bf.s[id].p - buf); *sz = bf.s[id].size;
    return 0;
}

/* ------------------------------------------------------------------ decimal strings / JSON */

/* 256-bit little-endian limbs -> decimal, no leading zeros */
static int to_dec(char *out, const u64 v[4]) {
    u64 t[4] = {v[0], v[1], v[2], v[3]};
    char tmp[80]; int n = 0;
    if ((t[0] | t[1] | t[2] | t[3]) == 0) { out[0] = '0'; out[1] = 0; return 1; }
    while (t[0] | t[1] | t[2] | t[3]) {
        u128 rem = 0;
        for (int i = 3; i >= 0; i--) {
            u128 cur = (rem << 64) | t[i];
            t[i] = (u64)(cur / 10); rem = cur % 10;
        }
        tmp[n++] = (char)('0' + (int)rem);
    }
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    out[n] = 0;
    return n;
}
/* E.f1.toString: de-Montgomerise, print decimal (build/fq.cpp toString) */
static int fq_str(char *out, const fe *a) { fe t; fe_from_mont(&t, a, &UGO_FQ); return to_dec(out, t.v); }

/* nlohmann dump(): keys in lexicographic order, no whitespace (src/groth16.cpp:217-250) */
static int proof_json(char *out, const g1_aff *A, const g2_aff *B, const g1_aff *C) {
    char ax[80], ay[80], bxa[80], bxb[80], bya[80], byb[80], cx[80], cy[80];
    fq_str(ax, &A->x); fq_str(ay, &A->y);
    fq_str(bxa, &B->x.a); fq_str(bxb, &B->x.b); fq_str(bya, &B->y.a); fq_str(byb, &B->y.b);
    fq_str(cx, &C->x); fq_str(cy, &C->y);
    return sprintf(out,
        "{\"pi_a\":[\"%s\",\"%s\",\"1\"],\"pi_b\":[[\"%s\",\"%s\"],[\"%s\",\"%s\"],[\"1\",\"0\"]],"
        "\"pi_c\":[\"%s\",\"%s\",\"1\"],\"protocol\":\"groth16\"}",
        ax, ay, bxa, bxb, bya, byb, cx, cy);
}

/* ------------------------------------------------------------------ Groth16 prove */

/* S11-S13 with caller-supplied blinding scalars (the reference draws 31 random bytes each,
 * src/groth16.cpp:158-166; r and s here are 32-byte LE with byte 31 == 0). */
static void blind(g1_aff *A, g2_aff *B, g1_aff *C,
                  g1_xyzz pi_a, g1_xyzz pib1, g2_xyzz pi_b, g1_xyzz pi_c, g1_xyzz pih,
                  const zkey_hdr *h, const uint8_t r[32], const uint8_t s[32]) {
    g1_aff alpha1, beta1, delta1; g2_aff beta2, delta2;
    memcpy(&alpha1, h->alpha1, 64); memcpy(&beta1, h->beta1, 64); memcpy(&delta1, h->delta1, 64);
    memcpy(&beta2, h->beta2, 128); memcpy(&delta2, h->delta2, 128);
    g1_xyzz d1, p1; g2_xyzz d2, p2;
    g1_from_aff(&d1, &delta1); g2_from_aff(&d2, &delta2);

    g1_add_mixed(&pi_a, &pi_a, &alpha1);                         /* :171 */
    g1_mul_scalar(&p1, &d1, r, 32); g1_add(&pi_a, &pi_a, &p1);  /* :172-173 */

    g2_add_mixed(&pi_b, &pi_b, &beta2);                          /* :175 */
    g2_mul_scalar(&p2, &d2, s, 32); g2_add(&pi_b, &pi_b, &p2);  /* :176-177 */

    g1_add_mixed(&pib1, &pib1, &beta1);                          /* :179 */
    g1_mul_scalar(&p1, &d1, s, 32); g1_add(&pib1, &pib1, &p1);  /* :180-181 */

    g1_add(&pi_c, &pi_c, &pih);                                  /* :183 */
    g1_mul_scalar(&p1, &pi_a, s, 32); g1_add(&pi_c, &pi_c, &p1);/* :185-186 */
    g1_mul_scalar(&p1, &pib1, r, 32); g1_add(&pi_c, &pi_c, &p1);/* :188-189 */

    /* :191-192  rs = toMontgomery(MMul(r, s)) = r*s mod q as a plain integer */
    fe fr_, fs_, rs; memcpy(&fr_, r, 32); memcpy(&fs_, s, 32);
    r_mul(&rs, &fr_, &fs_); fe_to_mont(&rs, &rs, &UGO_FR);
    g1_mul_scalar(&p1, &d1, (const uint8_t *)rs.v, 32);          /* :194 */
    g1_sub(&pi_c, &pi_c, &p1);                                   /* :195 */

    g1_to_aff(A, &pi_a); g2_to_aff(B, &pi_b); g1_to_aff(C, &pi_c);
}

/* public signals: w[1..nPublic] as decimal strings (src/prover.cpp:106-117; toMontgomery
 * followed by toString's fromMontgomery is the identity on the normal-form value) */
static int public_json(char *out, const uint8_t *w, uint32_t npublic, uint32_t skip) {
    int n = 0; out[n++] = '[';
    int first = 1;
    for (uint32_t i = 1; i <= npublic; i++) {
        if (i == skip) continue;
        fe v; memcpy(&v, w + (size_t)i * 32, 32);
        if (fe_geq_q(v.v, &UGO_FR)) limbs_sub(v.v, v.v, UGO_FR.q);
        if (!first) out[n++] = ',';
        first = 0;
        out[n++] = '"'; n += to_dec(out + n, v.v); out[n++] = '"';
    }
    if (first) { strcpy(out, "null"); return 4; }        /* nlohmann: empty json dumps as null */
    out[n++] = ']'; out[n] = 0;
    return n;
}

/* S11-S13 and the JSON texts from the five MSM results given as affine records (A 64 | B1 64 | B2 128 | C 64 | H 64):
 * lets a test assemble the expected proof of a circuit too large for a CPU MSM from sums known in closed form.
 * `zkey` needs its sections 1, 2 and 4 only (header; a 4-byte section 4 will do); public signals from w[1..nPublic]. */
int ugo_groth16_finish(const uint8_t *zkey, uint64_t zkey_size, const uint8_t sums[384], const uint8_t *public_w,
                       const uint8_t r[32], const uint8_t s[32], char *proof_out, uint64_t proof_cap,
                       char *public_out, uint64_t public_cap, char *err, uint64_t errsz) {
    binfile zf; zkey_hdr h;
    if (binfile_parse(&zf, zkey, zkey_size, "zkey", 1, err, errsz)) return 1;
    if (zkey_header(&h, &zf, 0, err, errsz)) return 1;
    g1_aff a, b1, c, hh; g2_aff b2;
    memcpy(&a, sums, 64); memcpy(&b1, sums + 64, 64); memcpy(&b2, sums + 128, 128); memcpy(&c, sums + 256, 64); memcpy(&hh, sums + 320, 64);
    g1_xyzz pi_a, pib1, pi_c, pih; g2_xyzz pi_b;
    g1_from_aff(&pi_a, &a); g1_from_aff(&pib1, &b1); g2_from_aff(&pi_b, &b2); g1_from_aff(&pi_c, &c); g1_from_aff(&pih, &hh);
    g1_aff A, C; g2_aff B;
    blind(&A, &B, &C, pi_a, pib1, pi_b, pi_c, pih, &h, r, s);
    char pj[1024]; int pl = proof_json(pj, &A, &B, &C);
    char *pub = (char *)malloc((size_t)h.npublic * 82 + 8);
    int ql = public_json(pub, public_w, h.npublic, 0);
    if ((uint64_t)pl + 1 > proof_cap || (uint64_t)ql + 1 > public_cap) { free(pub); seterr(err, errsz, "buffer too short"); return 2; }
    memcpy(proof_out, pj, (size_t)pl + 1); memcpy(public_out, pub, (size_t)ql + 1);
    free(pub);
    return 0;
}

int ugo_groth16_prove(const uint8_t *zkey, uint64_t zkey_size, const uint8_t *wtns, uint64_t wtns_size,
                      const uint8_t r[32], const uint8_t s[32],
                      char *proof_out, uint64_t proof_cap, char *public_out, uint64_t public_cap,
                      uint8_t *raw_out /* optional: MSM_A(64) B1(64) B2(128) C(64) H(64) affine */,
                      double *timings /* optional: [msm_s, fft_s] */,
                      char *err, uint64_t errsz) {
    binfile zf, wf; zkey_hdr h;
    if (binfile_parse(&zf, zkey, zkey_size, "zkey", 1, err, errsz)) return 1;
    if (zkey_header(&h, &zf, 0, err, errsz)) return 1;
    for (int id = 5; id <= 9; id++) if (!zf.present[id]) return seterr(err, errsz, "Section does not exist");
    if (binfile_parse(&wf, wtns, wtns_size, "wtns", 2, err, errsz)) return 1;
    if (!wf.present[1] || !wf.present[2]) return seterr(err, errsz, "Section does not exist");
    const uint8_t *wp = wf.s[1].p;
    uint32_t n8 = rd32(wp);
    if (n8 != 32 || memcmp(wp + 4, FR_PRIME_LE, 32) != 0) return seterr(err, errsz, "different wtns curve");
    uint32_t wn = rd32(wp + 4 + n8);
    if (wn != h.nvars) {
        char m[128]; snprintf(m, sizeof m, "Invalid witness length. Circuit: %u, witness: %u", h.nvars, wn);
        seterr(err, errsz, m); return 3;
    }
    const uint8_t *w = wf.s[2].p;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    g1_xyzz pi_a, pib1, pi_c, pih; g2_xyzz pi_b;
    g1_msm(&pi_a, zf.s[5].p, w, 32, h.nvars);                                     /* S1 :55 */
    g1_msm(&pib1, zf.s[6].p, w, 32, h.nvars);                                     /* S2 :58 */
    g2_msm(&pi_b, zf.s[7].p, w, 32, h.nvars);                                     /* S3 :61 */
    g1_msm(&pi_c, zf.s[8].p, w + (size_t)(h.npublic + 1) * 32, 32, h.nvars - h.npublic - 1); /* S4 :64 */
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    uint64_t *hh = (uint64_t *)malloc((size_t)h.domain * 32);
    if (ugo_hpoly(hh, zf.s[4].p + 4, h.ncoefs, w, h.nvars, h.domain, NULL)) {   /* S5-S9 */
        free(hh); return seterr(err, errsz, "coefficient index out of range");
    }
#ifdef _OPENMP
    t2 = omp_get_wtime();
#endif
    g1_msm(&pih, zf.s[9].p, (const uint8_t *)hh, 32, h.domain);                   /* S10 :154 */
    free(hh);
#ifdef _OPENMP
    t3 = omp_get_wtime();
#endif
    if (timings) { timings[0] = (t1 - t0) + (t3 - t2); timings[1] = t2 - t1; }
    if (raw_out) {
        g1_aff a1; g2_aff a2;
        g1_to_aff(&a1, &pi_a); memcpy(raw_out, &a1, 64);
        g1_to_aff(&a1, &pib1); memcpy(raw_out + 64, &a1, 64);
        g2_to_aff(&a2, &pi_b); memcpy(raw_out + 128, &a2, 128);
        g1_to_aff(&a1, &pi_c); memcpy(raw_out + 256, &a1, 64);
        g1_to_aff(&a1, &pih);  memcpy(raw_out + 320, &a1, 64);
    }
    g1_aff A, C; g2_aff B;
    blind(&A, &B, &C, pi_a, pib1, pi_b, pi_c, pih, &h, r, s);
    char pj[1024]; int pl = proof_json(pj, &A, &B, &C);
    char *pub = (char *)malloc((size_t)h.npublic * 82 + 8);
    int ql = public_json(pub, w, h.npublic, 0);
    if ((uint64_t)pl + 1 > proof_cap || (uint64_t)ql + 1 > public_cap) { free(pub); seterr(err, errsz, "buffer too short"); return 2; }
    memcpy(proof_out, pj, (size_t)pl + 1);
    memcpy(public_out, pub, (size_t)ql + 1);
    free(pub);
    return 0;
}

/* ------------------------------------------------------------------ Keccak-256 (UltraGroth) */

/* Keccak-f[1600], rate 1088, Ethereum padding 0x01 (src/keccak256.cpp:8) */
static const u64 KRC[24] = {
    0x0000000000000001ULL,0x0000000000008082ULL,0x800000000000808aULL,0x8000000080008000ULL,
    0x000000000000808bULL,0x0000000080000001ULL,0x8000000080008081ULL,0x8000000000008009ULL,
    0x000000000000008aULL,0x0000000000000088ULL,0x0000000080008009ULL,0x000000008000000aULL,
    0x000000008000808bULL,0x800000000000008bULL,0x8000000000008089ULL,0x8000000000008003ULL,
    0x8000000000008002ULL,0x8000000000000080ULL,0x000000000000800aULL,0x800000008000000aULL,
    0x8000000080008081ULL,0x8000000000008080ULL,0x0000000080000001ULL,0x8000000080008008ULL};
static const int KROT[24] = {1,3,6,10,15,21,28,36,45,55,2,14,27,41,56,8,25,43,62,18,39,61,20,44};
static const int KPIL[24] = {10,7,11,17,18,3,5,16,8,21,24,4,15,23,19,13,12,2,20,14,22,9,6,1};
static void keccakf(u64 st[25]) {
    for (int round = 0; round < 24; round++) {
        u64 bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = KPIL[i]; u64 b = st[j];
            st[j] = (t << KROT[i]) | (t >> (64 - KROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KRC[round];
    }
}
void ugo_keccak256(uint8_t out[32], const uint8_t *in, uint64_t len) {
    u64 st[25]; memset(st, 0, sizeof st);
    uint8_t *sb = (uint8_t *)st;
    const uint64_t rate = 136;
    while (len >= rate) {
        for (uint64_t i = 0; i < rate; i++) sb[i] ^= in[i];
        keccakf(st); in += rate; len -= rate;
    }
    for (uint64_t i = 0; i < len; i++) sb[i] ^= in[i];
    sb[len] ^= 0x01; sb[rate - 1] ^= 0x80;
    keccakf(st);
    memcpy(out, sb, 32);
}

/* ------------------------------------------------------------------ UltraGroth */

/* src/ultra_groth.cpp:33-58: keccak256(x_BE32 || y_BE32) read as a big-endian integer and
 * brought into Fr (fromMpz: Montgomery product with R^2 reduces values >= r). Returns Montgomery form. */
static void derive_challenge(fe *rand_m, const g1_aff *commit) {
    uint8_t buf[64], ch[32];
    fe x, y; fe_from_mont(&x, &commit->x, &UGO_FQ); fe_from_mont(&y, &commit->y, &UGO_FQ);
    for (int i = 0; i < 32; i++) {
        buf[i] = (uint8_t)(x.v[3 - (i >> 3)] >> (56 - 8 * (i & 7)));
        buf[32 + i] = (uint8_t)(y.v[3 - (i >> 3)] >> (56 - 8 * (i & 7)));
    }
    ugo_keccak256(ch, buf, 64);
    fe v; fe_zero(&v);
    for (int i = 0; i < 32; i++) v.v[3 - (i >> 3)] |= (u64)ch[i] << (56 - 8 * (i & 7));
    fe_to_mont(rand_m, &v, &UGO_FR);
}
void ugo_derive_challenge(uint8_t out_normal[32], const uint8_t commit_aff[64]) {
    g1_aff c; fe m, nrm; memcpy(&c, commit_aff, 64);
    derive_challenge(&m, &c); fe_from_mont(&nrm, &m, &UGO_FR); memcpy(out_normal, &nrm, 32);
}

/* RawFr::set(int) (build/fr.cpp:209-223): the Montgomery form of a C int, negative values taken as value + r. */
static void fr_set_int(fe *r, int32_t value) {
    fe m = {{(u64)(value < 0 ? -(int64_t)value : (int64_t)value), 0, 0, 0}};
    fe_to_mont(&m, &m, &UGO_FR);
    if (value < 0) fe_neg(&m, &m, &UGO_FR);
    *r = m;
}
/* One row of the lookup table, src/ultra_groth.cpp:72-79:
 *     sum = field.add(i, rand); inv(inv, sum); prod = field.mul(frequencies[i], inv);
 * `i` (an int loop variable) and `frequencies[i]` (a uint32_t) both bind to the (int, Element) overloads
 * (build/fr.hpp:249-251), so each goes through RawFr::set(int): an unsigned frequency >= 2^31 converts to the
 * negative int freq - 2^32 and enters the product as freq - 2^32 + r. Outputs are normal form (copy_digits :24-31). */
static void lookup_row(fe *inv2_out, fe *prod_out, uint32_t i, uint32_t frequency, const fe *rand_m) {
    fe im, sum, inv, fm, pr;
    fr_set_int(&im, (int32_t)i);
    r_add(&sum, &im, rand_m);
    fe_inv(&inv, &sum, &UGO_FR);              /* inverse of 0 is 0 (mpz_invert leaves 0) */
    fe_from_mont(inv2_out, &inv, &UGO_FR);
    fr_set_int(&fm, (int32_t)frequency);
    r_mul(&pr, &fm, &inv);
    fe_from_mont(prod_out, &pr, &UGO_FR);
}
void ugo_lookup_row(uint64_t inv2_out[4], uint64_t prod_out[4], uint32_t i, uint32_t frequency, const uint64_t rand_mont[4]) {
    fe a, b, rm; memcpy(&rm, rand_mont, 32);
    lookup_row(&a, &b, i, frequency, &rm);
    memcpy(inv2_out, &a, 32); memcpy(prod_out, &b, 32);
}

/* src/ultra_groth.cpp:62-106. signals: nVars normal-form values, modified in place. */
static void compute_lookup(uint8_t *signals, const uint32_t *chunks, uint32_t chunks_total,
                           const uint32_t *freq, uint32_t lookup_size,
                           const uint32_t *wtns_idx, const uint32_t *push_idx, uint32_t n_idx,
                           const fe *rand_m) {
    size_t total = (size_t)2 * lookup_size + chunks_total + 1;
    fe *push = (fe *)malloc(total * sizeof(fe));
    fe *inv1 = push + 1, *inv2 = inv1 + chunks_total, *prod = inv2 + lookup_size;
    fe_from_mont(&push[0], rand_m, &UGO_FR);
    for (uint32_t i = 0; i < lookup_size; i++) lookup_row(&inv2[i], &prod[i], i, freq[i], rand_m);
    for (uint32_t i = 0; i < chunks_total; i++) inv1[i] = inv2[chunks[i]];
    for (uint32_t i = 0; i < n_idx; i++) memcpy(signals + (size_t)wtns_idx[i] * 32, &push[push_idx[i]], 32);
    free(push);
}

static int ultra_proof_json(char *out, const g1_aff *A, const g2_aff *B, const g1_aff *F, const g1_aff *Rr) {
    /* keys pi_a, pi_b, pi_f, pi_r, protocol (src/ultra_groth.cpp:476-513), lexicographic dump */
    char ax[80], ay[80], bxa[80], bxb[80], bya[80], byb[80], fx[80], fy[80], rx[80], ry[80];
    fq_str(ax, &A->x); fq_str(ay, &A->y);
    fq_str(bxa, &B->x.a); fq_str(bxb, &B->x.b); fq_str(bya, &B->y.a); fq_str(byb, &B->y.b);
    fq_str(fx, &F->x); fq_str(fy, &F->y); fq_str(rx, &Rr->x); fq_str(ry, &Rr->y);
    return sprintf(out,
        "{\"pi_a\":[\"%s\",\"%s\",\"1\"],\"pi_b\":[[\"%s\",\"%s\"],[\"%s\",\"%s\"],[\"1\",\"0\"]],"
        "\"pi_f\":[\"%s\",\"%s\",\"1\"],\"pi_r\":[\"%s\",\"%s\",\"1\"],\"protocol\":\"ultragroth\"}",
        ax, ay, bxa, bxb, bya, byb, fx, fy, rx, ry);
}

/* src/ultra_groth.cpp:401-462 with the three blinding draws (r_k, r, s) supplied by the caller */
int ugo_ultra_groth_prove(const uint8_t *zkey, uint64_t zkey_size, const uint8_t *wtns, uint64_t wtns_size,
                          const uint8_t rk[32], const uint8_t r[32], const uint8_t s[32],
                          char *proof_out, uint64_t proof_cap, char *public_out, uint64_t public_cap,
                          char *err, uint64_t errsz) {
    binfile zf, wf; zkey_hdr h;
    if (binfile_parse(&zf, zkey, zkey_size, "zkey", 1, err, errsz)) return 1;
    if (zkey_header(&h, &zf, 1, err, errsz)) return 1;
    for (int id = 5; id <= 12; id++) if (!zf.present[id]) return seterr(err, errsz, "Section does not exist");
    if (binfile_parse(&wf, wtns, wtns_size, "wtns", 2, err, errsz)) return 1;
    for (int id = 1; id <= 6; id++) if (!wf.present[id]) return seterr(err, errsz, "Section does not exist");
    const uint8_t *wp = wf.s[1].p;
    uint32_t n8 = rd32(wp);
    if (n8 != 32 || memcmp(wp + 4, FR_PRIME_LE, 32) != 0) return seterr(err, errsz, "different wtns curve");
    uint32_t wn = rd32(wp + 4 + n8);
    if (wn != h.nvars) {
        char m[128]; snprintf(m, sizeof m, "Invalid witness length. Circuit: %u, witness: %u", h.nvars, wn);
        seterr(err, errsz, m); return 3;
    }
    uint8_t *sig = (uint8_t *)malloc((size_t)h.nvars * 32);
    memcpy(sig, wf.s[2].p, (size_t)h.nvars * 32);
    const uint32_t *ridx = (const uint32_t *)zf.s[10].p, *fidx = (const uint32_t *)zf.s[11].p;
    uint32_t *ridx_c = (uint32_t *)malloc((size_t)h.n_idx_c1 * 4 + 4), *fidx_c = (uint32_t *)malloc((size_t)h.n_idx_c2 * 4 + 4);
    memcpy(ridx_c, ridx, (size_t)h.n_idx_c1 * 4); memcpy(fidx_c, fidx, (size_t)h.n_idx_c2 * 4);

    /* round 1 (:415-419, execute_round :161-184) */
    uint8_t *rw = (uint8_t *)malloc((size_t)h.n_idx_c1 * 32 + 32);
    for (uint32_t i = 0; i < h.n_idx_c1; i++) memcpy(rw + (size_t)i * 32, sig + (size_t)ridx_c[i] * 32, 32);
    g1_xyzz commit, tmp, fd1; g1_aff fdelta1, rdelta1, commit_aff;
    g1_msm(&commit, zf.s[8].p, rw, 32, h.n_idx_c1);
    memcpy(&fdelta1, h.delta1, 64); memcpy(&rdelta1, h.round_delta1, 64);
    g1_from_aff(&fd1, &fdelta1);
    g1_mul_scalar(&tmp, &fd1, rk, 32); g1_add(&commit, &commit, &tmp);      /* :176 blinds with final_delta1 */
    g1_to_aff(&commit_aff, &commit);
    free(rw);

    fe rand_m; derive_challenge(&rand_m, &commit_aff);                       /* :428 */

    /* copy the four u32 index sections out of the (possibly unaligned) buffer */
    uint32_t nch = (uint32_t)(wf.s[3].size >> 2), nfr = (uint32_t)(wf.s[4].size >> 2), nix = (uint32_t)(wf.s[5].size >> 2);
    uint32_t *chunks = (uint32_t *)malloc((size_t)nch * 4 + 4), *freq = (uint32_t *)malloc((size_t)nfr * 4 + 4);
    uint32_t *widx = (uint32_t *)malloc((size_t)nix * 4 + 4), *pidx = (uint32_t *)malloc((size_t)nix * 4 + 4);
    memcpy(chunks, wf.s[3].p, (size_t)nch * 4); memcpy(freq, wf.s[4].p, (size_t)nfr * 4);
    memcpy(widx, wf.s[5].p, (size_t)nix * 4); memcpy(pidx, wf.s[6].p, (size_t)nix * 4);
    compute_lookup(sig, chunks, nch, freq, nfr, widx, pidx, nix, &rand_m);   /* :437 */
    free(chunks); free(freq); free(widx); free(pidx);

    uint8_t *fw = (uint8_t *)malloc((size_t)h.n_idx_c2 * 32 + 32);
    for (uint32_t i = 0; i < h.n_idx_c2; i++) memcpy(fw + (size_t)i * 32, sig + (size_t)fidx_c[i] * 32, 32);

    /* final round (:187-399) */
    g1_xyzz pi_a, pib1, pi_c, pih; g2_xyzz pi_b;
    g1_msm(&pi_a, zf.s[5].p, sig, 32, h.nvars);
    g1_msm(&pib1, zf.s[6].p, sig, 32, h.nvars);
    g2_msm(&pi_b, zf.s[7].p, sig, 32, h.nvars);
    g1_msm(&pi_c, zf.s[9].p, fw, 32, h.n_idx_c2);
    free(fw);
    uint64_t *hh = (uint64_t *)malloc((size_t)h.domain * 32);
    if (ugo_hpoly(hh, zf.s[4].p + 4, h.ncoefs, sig, h.nvars, h.domain, NULL)) {
        free(hh); free(sig); free(ridx_c); free(fidx_c); return seterr(err, errsz, "coefficient index out of range");
    }
    g1_msm(&pih, zf.s[12].p, (const uint8_t *)hh, 32, h.domain);
    free(hh);
    /* :386-388  pi_c -= round_random_factor * round_delta1  (folded in before the affine step) */
    g1_xyzz rd1, p1; g1_from_aff(&rd1, &rdelta1);
    g1_mul_scalar(&p1, &rd1, rk, 32);
    g1_sub(&pi_c, &pi_c, &p1);
    g1_aff A, C; g2_aff B;
    blind(&A, &B, &C, pi_a, pib1, pi_b, pi_c, pih, &h, r, s);

    char pj[1600]; int pl = ultra_proof_json(pj, &A, &B, &C, &commit_aff);
    char *pub = (char *)malloc((size_t)h.npublic * 82 + 8);
    int ql = public_json(pub, sig, h.npublic, h.rand_indx);                  /* prover.cpp:89-105 */
    free(sig); free(ridx_c); free(fidx_c);
    if ((uint64_t)pl + 1 > proof_cap || (uint64_t)ql + 1 > public_cap) { free(pub); seterr(err, errsz, "buffer too short"); return 2; }
    memcpy(proof_out, pj, (size_t)pl + 1);
    memcpy(public_out, pub, (size_t)ql + 1);
    free(pub);
    return 0;
}

int ugo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* cap the OpenMP team (bench.py: the box's CPU share may be smaller than the cores OpenMP sees) */
void ugo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* sum_i scalars[i] * (seed + i) mod r, plain integers in and out: the discrete log of an MSM over the
 * synthetic base points P_i = (seed + i) * G (ultragroth_amd/synth.py), for full-size checks in the exponent */
void ugo_fr_dot_walk(uint64_t out[4], const uint8_t *scalars, uint64_t n, uint64_t seed) {
    fe total; fe_zero(&total);
#pragma omp parallel
    {
        fe acc; fe_zero(&acc);
#pragma omp for schedule(static)
        for (uint64_t i = 0; i < n; i++) {
            fe s, k = {{seed + i, 0, 0, 0}}, t;
            memcpy(&s, scalars + i * 32, 32);
            fe_to_mont(&s, &s, &UGO_FR);          /* also reduces values >= r */
            fe_mul(&t, &s, &k, &UGO_FR);          /* (s R)(k)/R = s k */
            fe_add(&acc, &acc, &t, &UGO_FR);
        }
#pragma omp critical
        fe_add(&total, &total, &acc, &UGO_FR);
    }
    memcpy(out, &total, 32);
}
