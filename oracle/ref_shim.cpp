// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" wrappers over the REFERENCE's own field layer, compiled from the sources
// where they lie (/root/reference/build/{fr,fq}{,_generic,_raw_generic}.cpp) by
// oracle/Makefile into oracle/_ref/libref_field.so. Nothing from the reference is copied
// into this repository; this file only calls RawFr / RawFq (build/fr.hpp:206-281).
// Used to validate oracle/field.h (and through it the HIP field code) against the
// reference's arithmetic: mul, add, sub, neg, to/fromMontgomery, inv, toString.
//
// Two UltraGroth steps are pinned the same way -- each wrapper below makes the very calls the reference makes, on the
// reference's own classes (RawFr / RawFq of build/fr.cpp, FIPS202_KECCAK_256 of src/keccak256.cpp, GMP):
//   ref_lookup_row        one iteration of compute_lookup's table loop   (src/ultra_groth.cpp:72-79, copy_digits :24-31)
//   ref_derive_challenge  derive_challenge                               (src/ultra_groth.cpp:33-58)
#include <cstring>
#include <string>
#include <gmp.h>
#include "fr.hpp"
#include "fq.hpp"
#include "keccak256.h"

extern "C" {

void ref_fr_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) { Fr_rawMMul(r, a, b); }
void ref_fr_add(uint64_t *r, const uint64_t *a, const uint64_t *b) { Fr_rawAdd(r, a, b); }
void ref_fr_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) { Fr_rawSub(r, a, b); }
void ref_fr_neg(uint64_t *r, const uint64_t *a) { Fr_rawNeg(r, a); }
void ref_fr_to_mont(uint64_t *r, const uint64_t *a) {
    RawFr::Element x, y; memcpy(x.v, a, 32); RawFr::field.toMontgomery(y, x); memcpy(r, y.v, 32);
}
void ref_fr_from_mont(uint64_t *r, const uint64_t *a) {
    RawFr::Element x, y; memcpy(x.v, a, 32); RawFr::field.fromMontgomery(y, x); memcpy(r, y.v, 32);
}
void ref_fr_inv(uint64_t *r, const uint64_t *a) {
    RawFr::Element x, y; memcpy(x.v, a, 32); RawFr::field.inv(y, x); memcpy(r, y.v, 32);
}
int ref_fr_to_string(char *out, int cap, const uint64_t *a) {
    RawFr::Element x; memcpy(x.v, a, 32);
    std::string s = RawFr::field.toString(x);
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1); return (int)s.size();
}

void ref_fq_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) { Fq_rawMMul(r, a, b); }
void ref_fq_add(uint64_t *r, const uint64_t *a, const uint64_t *b) { Fq_rawAdd(r, a, b); }
void ref_fq_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) { Fq_rawSub(r, a, b); }
void ref_fq_neg(uint64_t *r, const uint64_t *a) { Fq_rawNeg(r, a); }
void ref_fq_to_mont(uint64_t *r, const uint64_t *a) {
    RawFq::Element x, y; memcpy(x.v, a, 32); RawFq::field.toMontgomery(y, x); memcpy(r, y.v, 32);
}
void ref_fq_from_mont(uint64_t *r, const uint64_t *a) {
    RawFq::Element x, y; memcpy(x.v, a, 32); RawFq::field.fromMontgomery(y, x); memcpy(r, y.v, 32);
}
void ref_fq_inv(uint64_t *r, const uint64_t *a) {
    RawFq::Element x, y; memcpy(x.v, a, 32); RawFq::field.inv(y, x); memcpy(r, y.v, 32);
}
int ref_fq_to_string(char *out, int cap, const uint64_t *a) {
    RawFq::Element x; memcpy(x.v, a, 32);
    std::string s = RawFq::field.toString(x);
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1); return (int)s.size();
}


// src/ultra_groth.cpp:72-79 with i and frequency typed as the reference types them (int loop variable, uint32_t
// array element): overload resolution picks add(int, Element) / mul(int, Element) exactly as it does there
void ref_lookup_row(uint64_t *inv2_digits, uint64_t *prod_digits, int i, uint32_t frequency, const uint64_t *rand_mont) {
    RawFr::Element rand; memcpy(rand.v, rand_mont, 32);
    RawFr::Element sum = RawFr::field.add(i, rand);
    RawFr::Element inv;
    RawFr::field.inv(inv, sum);
    RawFr::Element tmp;
    Fr_rawFromMontgomery(tmp.v, inv.v); memcpy(inv2_digits, tmp.v, 32);
    RawFr::Element prod = RawFr::field.mul(frequency, inv);
    Fr_rawFromMontgomery(tmp.v, prod.v); memcpy(prod_digits, tmp.v, 32);
}

// src/ultra_groth.cpp:33-58 on a commitment given as two Montgomery Fq coordinates; E.f1 = RawFq, E.fr = RawFr
void ref_derive_challenge(uint64_t *rand_mont_out, const uint64_t *x_mont, const uint64_t *y_mont) {
    RawFq::Element x, y; memcpy(x.v, x_mont, 32); memcpy(y.v, y_mont, 32);
    uint8_t buffer[2 * 32];
    uint8_t challenge[32];
    memset(buffer, 0, sizeof buffer);          // (the reference leaves it uninitialised; only matters below 2^192)
    mpz_t coordinate_buffer;
    mpz_init(coordinate_buffer);
    RawFq::field.toMpz(coordinate_buffer, x);
    mpz_export(buffer + 0, NULL, 1, 8, 1, 0, coordinate_buffer);
    RawFq::field.toMpz(coordinate_buffer, y);
    mpz_export(buffer + 32, NULL, 1, 8, 1, 0, coordinate_buffer);
    FIPS202_KECCAK_256(buffer, 32 * 2, challenge);
    RawFr::Element rand;
    mpz_t v;
    mpz_init(v);
    mpz_import(v, 32, 0, 1, -1, 0, challenge);
    RawFr::field.fromMpz(rand, v);
    memcpy(rand_mont_out, rand.v, 32);
    mpz_clear(v); mpz_clear(coordinate_buffer);
}

}
