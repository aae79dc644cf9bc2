// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" wrappers over the REFERENCE's own field layer, compiled from the sources
// where they lie (/root/reference/build/{fr,fq}{,_generic,_raw_generic}.cpp) by
// oracle/Makefile into oracle/_ref/libref_field.so. Nothing from the reference is copied
// into this repository; this file only calls RawFr / RawFq (build/fr.hpp:206-281).
// Used to validate oracle/field.h (and through it the HIP field code) against the
// reference's arithmetic: mul, add, sub, neg, to/fromMontgomery, inv, toString.
#include <cstring>
#include <string>
#include "fr.hpp"
#include "fq.hpp"

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

}
