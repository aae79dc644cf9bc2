/*
 * oracle/ug_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle C entry points).
 * See ug_oracle.c for what each function restates (reference file:line).
 * Every field element is 4 little-endian uint64 limbs; "Montgomery" = value * 2^256 mod p.
 * G1 affine = (x, y) Montgomery, 64 bytes; G2 affine = (x.a, x.b, y.a, y.b), 128 bytes;
 * all-zero record = point at infinity (zkey convention).
 */
#ifndef UG_ORACLE_H
#define UG_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#define UGO_FIELD_FR 0
#define UGO_FIELD_FQ 1

void ugo_f_mul(int which, uint64_t *r, const uint64_t *a, const uint64_t *b);
void ugo_f_mul_portable(int which, uint64_t *r, const uint64_t *a, const uint64_t *b);
void ugo_f_add(int which, uint64_t *r, const uint64_t *a, const uint64_t *b);
void ugo_f_sub(int which, uint64_t *r, const uint64_t *a, const uint64_t *b);
void ugo_f_neg(int which, uint64_t *r, const uint64_t *a);
void ugo_f_to_mont(int which, uint64_t *r, const uint64_t *a);
void ugo_f_from_mont(int which, uint64_t *r, const uint64_t *a);
void ugo_f_inv(int which, uint64_t *r, const uint64_t *a);
void ugo_f_mul_vec(int which, uint64_t *r, const uint64_t *a, const uint64_t *b, size_t n);

void ugo_g1_msm(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, size_t n);
void ugo_g1_msm_naive(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, size_t n);
void ugo_g2_msm(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, size_t n);
void ugo_g2_msm_naive(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, size_t n);
void ugo_g1_mul(uint8_t out[64], const uint8_t base[64], const uint8_t scalar[32]);
void ugo_g2_mul(uint8_t out[128], const uint8_t base[128], const uint8_t scalar[32]);
void ugo_g1_add(uint8_t out[64], const uint8_t p[64], const uint8_t q[64]);
void ugo_g2_add(uint8_t out[128], const uint8_t p[128], const uint8_t q[128]);
int  ugo_g1_on_curve(const uint8_t p[64]);
int  ugo_g2_on_curve(const uint8_t p[128]);

void ugo_fr_root_of_unity(uint64_t out[4], int s);
void ugo_fr_ntt(uint64_t *data, int logn, int inverse);
int  ugo_hpoly(uint64_t *h_out, const uint8_t *coefs, uint64_t ncoefs, const uint8_t *wtns,
               uint32_t nvars, uint32_t domain_size, uint64_t *abc_coset_out);

int  ugo_zkey_info(const uint8_t *zkey, uint64_t size, uint32_t out[4], uint64_t *ncoefs);
int  ugo_section(const uint8_t *buf, uint64_t size, const char *type, uint32_t id, uint64_t *off, uint64_t *sz);

int  ugo_groth16_prove(const uint8_t *zkey, uint64_t zkey_size, const uint8_t *wtns, uint64_t wtns_size,
                       const uint8_t r[32], const uint8_t s[32],
                       char *proof_out, uint64_t proof_cap, char *public_out, uint64_t public_cap,
                       uint8_t *raw_out, double *timings, char *err, uint64_t errsz);
int  ugo_groth16_finish(const uint8_t *zkey, uint64_t zkey_size, const uint8_t sums[384], const uint8_t *public_w,
                        const uint8_t r[32], const uint8_t s[32], char *proof_out, uint64_t proof_cap,
                        char *public_out, uint64_t public_cap, char *err, uint64_t errsz);
int  ugo_ultra_groth_prove(const uint8_t *zkey, uint64_t zkey_size, const uint8_t *wtns, uint64_t wtns_size,
                           const uint8_t rk[32], const uint8_t r[32], const uint8_t s[32],
                           char *proof_out, uint64_t proof_cap, char *public_out, uint64_t public_cap,
                           char *err, uint64_t errsz);

void ugo_keccak256(uint8_t out[32], const uint8_t *in, uint64_t len);
void ugo_derive_challenge(uint8_t out_normal[32], const uint8_t commit_aff[64]);
void ugo_lookup_row(uint64_t inv2_out[4], uint64_t prod_out[4], uint32_t i, uint32_t frequency, const uint64_t rand_mont[4]);
int  ugo_num_threads(void);
void ugo_set_num_threads(int n);
void ugo_fr_dot_walk(uint64_t out[4], const uint8_t *scalars, uint64_t n, uint64_t seed);
#endif
