// verifier_api.cpp -- groth16_verify / ultra_groth_verify (include/verifier.h), host code.
//
// Replaces src/verifier.cpp (JSON front end, error strings, return codes) and the verifier halves of
// src/groth16.cpp:298-690 / src/ultra_groth.cpp:565-975, whose arithmetic (Fq6/Fq12 tower, G2, mulByScalar) lives in
// the absent ffiasm submodule. Verification is milliseconds of CPU work in the reference and stays on the host here:
// SURVEY.md section 8(f) lists it as the first row after the prover hot path. The checks are the reference's:
//     groth16:    vkX = IC[0] + sum_i input_i IC[i+1];   e(A,B) e(-alpha1,beta2) e(-vkX,gamma2) e(-C,delta2) == 1
//     ultragroth: vkX = IC[0] + sum_i input_i IC[i+1] + derive_challenge(pi_r) IC_rand;
//                 e(A,B) e(-alpha1,beta2) e(-vkX,gamma2) e(-pi_f,delta_c2_2) e(-pi_r,delta_c1_2) == 1
// Pairs with a point at infinity are skipped (pairingCheck, src/groth16.cpp:673-690).
//
// The pairing is restated from the definition, not from the reference's line functions: Fq12 = Fq[w]/(w^12 - 18 w^6 + 82)
// as 12 coefficients (u = w^6 - 9), D-type twist (x, y) -> (x w^2, y w^3), optimal-ate Miller loop over 6t+2 with affine
// G2 steps, the two Frobenius lines at the end. The result is a yes/no, so any correct pairing gives the reference's
// answer. Final exponentiation: with G = (f^(p^2) f)^((p^4 - p^2 + 1)/r) the full power f^((p^12-1)/r) equals
// conj(G)/G (conj = the p^6 Frobenius, w -> -w), which is 1 exactly when G lies in Fq6, i.e. when its odd coefficients
// vanish: one 761-bit exponentiation, no Fq12 inversion.
#include <cstdint>
#include <cstring>
#include <future>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "ec.hpp"
#include "host_util.hpp"
#include "../../include/verifier.h"

using namespace ug;
using ughost::keccak256;

namespace {

// ---- canonical field wrappers (always in [0, q), device Montgomery form) -------------------------------------------
struct F1 { Fq v; };
F1 f1_zero() { return F1{fp_zero<FqParams>()}; }
F1 f1_one() { return F1{canon(fp_one<FqParams>())}; }
// sums and differences of canonical values are below 2q, products of canonical values are strict and below 2q:
// one conditional subtraction restores [0, q)
F1 operator+(const F1& a, const F1& b) { return F1{cond_sub_q(norm_strict(add(a.v, b.v)))}; }
F1 operator-(const F1& a, const F1& b) { return F1{cond_sub_q(norm_strict(sub<1>(a.v, b.v)))}; }
F1 operator*(const F1& a, const F1& b) { return F1{cond_sub_q(mul(a.v, b.v))}; }
F1 operator-(const F1& a) { return F1{cond_sub_q(norm_strict(neg<1>(a.v)))}; }
bool is0(const F1& a) { return limbs_all_zero(a.v); }
bool operator==(const F1& a, const F1& b) {
    for (int i = 0; i < NL; i++) if (a.v.l[i] != b.v.l[i]) return false;
    return true;
}
F1 f1_inv(const F1& a) { return F1{canon(inv(a.v))}; }
F1 f1_small(u32 k) { u32 w[8] = {k, 0, 0, 0, 0, 0, 0, 0}; return F1{canon(from_normal<FqParams>(w))}; }

// decimal string -> value mod q (E.f1.fromString); anything but digits is an error
F1 f1_from_decimal(const std::string& s) {
    if (s.empty()) throw std::invalid_argument("not a number");
    const F1 ten = f1_small(10);
    F1 acc = f1_zero();
    for (char ch : s) {
        if (ch < '0' || ch > '9') throw std::invalid_argument("not a number");
        acc = acc * ten + f1_small((u32)(ch - '0'));
    }
    return acc;
}
void f1_to_plain(u32 out[8], const F1& a) { to_normal(out, a.v); }

struct F2 { F1 a, b; };                                            // a + b u, u^2 = -1
F2 f2_zero() { return F2{f1_zero(), f1_zero()}; }
F2 operator+(const F2& x, const F2& y) { return F2{x.a + y.a, x.b + y.b}; }
F2 operator-(const F2& x, const F2& y) { return F2{x.a - y.a, x.b - y.b}; }
F2 operator-(const F2& x) { return F2{-x.a, -x.b}; }
F2 operator*(const F2& x, const F2& y) { return F2{x.a * y.a - x.b * y.b, x.a * y.b + x.b * y.a}; }
F2 f2_scale(const F2& x, const F1& k) { return F2{x.a * k, x.b * k}; }
F2 f2_conj(const F2& x) { return F2{x.a, -x.b}; }
bool is0(const F2& x) { return is0(x.a) && is0(x.b); }
bool operator==(const F2& x, const F2& y) { return x.a == y.a && x.b == y.b; }
F2 f2_inv(const F2& x) {
    F1 n = f1_inv(x.a * x.a + x.b * x.b);
    return F2{x.a * n, -(x.b * n)};
}

// ---- Fq12, 12 coefficients in w ----------------------------------------------------------------------------------------
struct F12 { F1 c[12]; };
F12 f12_one() { F12 r; for (auto& x : r.c) x = f1_zero(); r.c[0] = f1_one(); return r; }
// Products are accumulated lazily: up to 6 limb-column products of canonical operands share one Montgomery reduction
// ((6 * 9 + 9) * 2^58 < 2^64 per column; 6 q^2 / 2^261 + q < 2q for the value), so a full product costs 144 column
// products and ~40 reductions instead of 144 of each.
struct LazySum {
    u64 c[2 * NL];
    int terms = 0;
    F1 total = f1_zero();
    LazySum() { cols_zero(c); }
    void flush() {
        if (!terms) return;
        total = total + F1{cond_sub_q(redc<FqParams>(c))};
        cols_zero(c);
        terms = 0;
    }
    void add(const F1& x, const F1& y) {
        if (terms == 6) flush();
        cols_mul(c, x.v, y.v);
        terms++;
    }
    F1 value() { flush(); return total; }
};
F12 f12_reduce(F1* t) {                                            // w^12 = 18 w^6 - 82
    static const F1 k18 = f1_small(18), k82 = f1_small(82);
    for (int k = 22; k >= 12; k--) {
        if (is0(t[k])) continue;
        t[k - 6] = t[k - 6] + k18 * t[k];
        t[k - 12] = t[k - 12] - k82 * t[k];
    }
    F12 r;
    for (int i = 0; i < 12; i++) r.c[i] = t[i];
    return r;
}
F12 f12_mul(const F12& a, const F12& b) {
    bool za[12], zb[12];
    for (int i = 0; i < 12; i++) { za[i] = is0(a.c[i]); zb[i] = is0(b.c[i]); }
    F1 t[23];
    for (int k = 0; k < 23; k++) {
        LazySum sum;
        for (int i = (k > 11 ? k - 11 : 0); i <= (k < 11 ? k : 11); i++)
            if (!za[i] && !zb[k - i]) sum.add(a.c[i], b.c[k - i]);
        t[k] = sum.value();
    }
    return f12_reduce(t);
}
F12 f12_sqr(const F12& a) {
    F1 t[23];
    for (int k = 0; k < 23; k++) {
        LazySum cross;                                              // sum over i < j, i + j = k  (at most 6 pairs)
        for (int i = (k > 11 ? k - 11 : 0); 2 * i < k; i++) cross.add(a.c[i], a.c[k - i]);
        F1 s = cross.value();
        t[k] = s + s;
        if (!(k & 1)) t[k] = t[k] + a.c[k >> 1] * a.c[k >> 1];
    }
    return f12_reduce(t);
}
// (a + b u) w^k with u = w^6 - 9, added into f
void f12_add_embedded(F12& f, const F2& c, int k) {
    static const F1 k9 = f1_small(9);
    f.c[k] = f.c[k] + (c.a - k9 * c.b);
    f.c[k + 6] = f.c[k + 6] + c.b;
}

// ---- curve points, affine with an infinity flag -----------------------------------------------------------------------
struct G1A { F1 x, y; bool inf; };
struct G2A { F2 x, y; bool inf; };

bool g1_on_curve(const G1A& p) { return p.inf || p.y * p.y == p.x * p.x * p.x + f1_small(3); }
bool g2_on_curve(const G2A& q) {
    if (q.inf) return true;
    static const F2 b = f2_scale(f2_inv(F2{f1_small(9), f1_small(1)}), f1_small(3));     // 3 / (9 + u)
    return q.y * q.y == q.x * q.x * q.x + b;
}
G2A g2_dbl(const G2A& p) {
    if (p.inf || is0(p.y)) return G2A{f2_zero(), f2_zero(), true};
    F2 m = f2_scale(p.x * p.x, f1_small(3)) * f2_inv(p.y + p.y);
    F2 x = m * m - (p.x + p.x);
    return G2A{x, m * (p.x - x) - p.y, false};
}
G2A g2_add(const G2A& p, const G2A& q) {
    if (p.inf) return q;
    if (q.inf) return p;
    if (p.x == q.x) {
        if (p.y == q.y) return g2_dbl(p);
        return G2A{f2_zero(), f2_zero(), true};
    }
    F2 m = (q.y - p.y) * f2_inv(q.x - p.x);
    F2 x = m * m - p.x - q.x;
    return G2A{x, m * (p.x - x) - p.y, false};
}
// line through the twist points t1, t2 (tangent when equal), evaluated at the G1 point pt
F12 line(const G2A& t1, const G2A& t2, const G1A& pt) {
    F12 f;
    for (auto& x : f.c) x = f1_zero();
    F2 m;
    if (!(t1.x == t2.x)) m = (t2.y - t1.y) * f2_inv(t2.x - t1.x);
    else if (t1.y == t2.y && !is0(t1.y)) m = f2_scale(t1.x * t1.x, f1_small(3)) * f2_inv(t1.y + t1.y);
    else {                                                          // vertical: xP - x1 w^2
        f.c[0] = pt.x;
        f12_add_embedded(f, -t1.x, 2);
        return f;
    }
    f.c[0] = -pt.y;                                                 // -yP + (m xP) w + (y1 - m x1) w^3
    f12_add_embedded(f, f2_scale(m, pt.x), 1);
    f12_add_embedded(f, t1.y - m * t1.x, 3);
    return f;
}

struct Consts {
    F2 g12, g13;           // xi^((p-1)/3), xi^((p-1)/2)
    F1 g22, g23;           // xi^((p^2-1)/3), xi^((p^2-1)/2) (both in Fq)
    F1 gamma[12];          // gamma^k, gamma = 82^((p-1)/6): the p^2 Frobenius maps w^k to gamma^k w^k
    Consts() {
        g12 = F2{f1_from_decimal("21575463638280843010398324269430826099269044274347216827212613867836435027261"),
                 f1_from_decimal("10307601595873709700152284273816112264069230130616436755625194854815875713954")};
        g13 = F2{f1_from_decimal("2821565182194536844548159561693502659359617185244120367078079554186484126554"),
                 f1_from_decimal("3505843767911556378687030309984248845540243509899259641013678093033130930403")};
        g22 = f1_from_decimal("21888242871839275220042445260109153167277707414472061641714758635765020556616");
        g23 = f1_from_decimal("21888242871839275222246405745257275088696311157297823662689037894645226208582");
        F1 g = f1_from_decimal("21888242871839275220042445260109153167277707414472061641714758635765020556617");
        gamma[0] = f1_one();
        for (int k = 1; k < 12; k++) gamma[k] = gamma[k - 1] * g;
    }
};
const Consts& consts() { static const Consts c; return c; }

constexpr unsigned __int128 ATE_LOOP = ((unsigned __int128)1 << 64) + 0x9d797039be763ba8ull;       // 6 t + 2 = 29793968203157093288

F12 miller(const G2A& q, const G1A& pt) {
    const Consts& k = consts();
    F12 f = f12_one();
    G2A r = q;
    for (int i = 63; i >= 0; i--) {                                 // bit 64 is the leading one
        f = f12_mul(f12_sqr(f), line(r, r, pt));
        r = g2_dbl(r);
        if ((ATE_LOOP >> i) & 1) {
            f = f12_mul(f, line(r, q, pt));
            r = g2_add(r, q);
        }
    }
    G2A q1{f2_conj(q.x) * k.g12, f2_conj(q.y) * k.g13, false};
    G2A nq2{f2_scale(q.x, k.g22), -f2_scale(q.y, k.g23), false};
    f = f12_mul(f, line(r, q1, pt));
    r = g2_add(r, q1);
    f = f12_mul(f, line(r, nq2, pt));
    return f;
}

// (p^4 - p^2 + 1) / r, 761 bits, little-endian words
const u32 HARD_EXPONENT[24] = {
    0xccdf42b1u, 0xe81bb482u, 0xf49c36d4u, 0x5abf5cc4u, 0x1da014fdu, 0xf1154e7eu, 0x87cdbacfu, 0xdcc7b44cu,
    0x954bcf8au, 0xaaa441e3u, 0xd5095f23u, 0x6b887d56u, 0xf3fd90c6u, 0x79581e16u, 0xd189227du, 0x3b1b1355u,
    0x61876f6bu, 0x4e529a58u, 0xd5b12278u, 0x6c0eb522u, 0x83177fafu, 0x331ec151u, 0x0b0759adu, 0x01baaa71u};

// f^((p^12 - 1)/r) == 1 ?  (see the header comment)
bool final_exponentiation_is_one(const F12& f) {
    bool zero = true;
    for (const auto& x : f.c) zero = zero && is0(x);
    if (zero) return false;
    const Consts& k = consts();
    F12 fp2;
    for (int i = 0; i < 12; i++) fp2.c[i] = f.c[i] * k.gamma[i];
    const F12 base = f12_mul(fp2, f);
    F12 g = f12_one();
    bool started = false;
    for (int i = 760; i >= 0; i--) {
        if (started) g = f12_sqr(g);
        if ((HARD_EXPONENT[i >> 5] >> (i & 31)) & 1) {
            g = started ? f12_mul(g, base) : base;
            started = true;
        }
    }
    for (int i = 1; i < 12; i += 2) if (!is0(g.c[i])) return false;
    return true;
}

// the four / five Miller loops are independent: one host thread each
bool pairing_check(const std::vector<G1A>& a, const std::vector<G2A>& b) {
    std::vector<std::future<F12>> parts;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i].inf || b[i].inf) continue;                         // src/groth16.cpp:679-681
        parts.push_back(std::async(std::launch::async, [&, i] { return miller(b[i], a[i]); }));
    }
    F12 acc = f12_one();
    for (auto& p : parts) acc = f12_mul(acc, p.get());
    return final_exponentiation_is_one(acc);
}

// ---- G1 arithmetic for vkX (ec.hpp, host) -------------------------------------------------------------------------------
G1XYZZ to_xyzz(const G1A& p) { return p.inf ? xyzz_inf<Fq>() : xyzz_from_affine(p.x.v, p.y.v); }
G1A from_xyzz(const G1XYZZ& p) {
    if (is_inf(p)) return G1A{f1_zero(), f1_zero(), true};
    Fq x, y;
    xyzz_to_affine(x, y, p);
    return G1A{F1{x}, F1{y}, false};
}
G1A g1_neg(const G1A& p) { return p.inf ? p : G1A{p.x, -p.y, false}; }

// ---- minimal JSON (objects, arrays, strings, numbers, literals): what nlohmann::json::parse accepts of it --------------
struct JVal {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    std::string str;                       // String: the text; Number: its literal
    bool integral = false;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal& at(const char* key) const {
        if (type != Object) throw std::invalid_argument("not an object");
        const JVal* hit = nullptr;
        for (const auto& kv : obj) if (kv.first == key) hit = &kv.second;          // later duplicates win, as nlohmann
        if (!hit) throw std::invalid_argument("missing key");
        return *hit;
    }
    const JVal& at(size_t i) const {
        if (type != Array || i >= arr.size()) throw std::invalid_argument("not an array element");
        return arr[i];
    }
    const std::string& string() const {
        if (type != String) throw std::invalid_argument("not a string");
        return str;
    }
};
struct JParser {
    const char* p;
    explicit JParser(const char* s) : p(s) {}
    void ws() { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++; }
    [[noreturn]] void bad() { throw std::invalid_argument("malformed json"); }
    JVal parse_document() {
        JVal v = value(0);
        ws();
        if (*p) bad();
        return v;
    }
    JVal value(int depth) {
        if (depth > 64) bad();
        ws();
        JVal v;
        if (*p == '{') {
            v.type = JVal::Object; p++; ws();
            if (*p == '}') { p++; return v; }
            for (;;) {
                ws();
                if (*p != '"') bad();
                std::string key = string_literal();
                ws();
                if (*p != ':') bad();
                p++;
                v.obj.emplace_back(std::move(key), value(depth + 1));
                ws();
                if (*p == ',') { p++; continue; }
                if (*p == '}') { p++; return v; }
                bad();
            }
        }
        if (*p == '[') {
            v.type = JVal::Array; p++; ws();
            if (*p == ']') { p++; return v; }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (*p == ',') { p++; continue; }
                if (*p == ']') { p++; return v; }
                bad();
            }
        }
        if (*p == '"') { v.type = JVal::String; v.str = string_literal(); return v; }
        if (!strncmp(p, "true", 4)) { p += 4; v.type = JVal::Bool; return v; }
        if (!strncmp(p, "false", 5)) { p += 5; v.type = JVal::Bool; return v; }
        if (!strncmp(p, "null", 4)) { p += 4; return v; }
        if (*p == '-' || (*p >= '0' && *p <= '9')) {
            const char* s = p;
            if (*p == '-') p++;
            if (*p == '0') p++;
            else if (*p >= '1' && *p <= '9') while (*p >= '0' && *p <= '9') p++;
            else bad();
            v.integral = true;
            if (*p == '.') { v.integral = false; p++; if (*p < '0' || *p > '9') bad(); while (*p >= '0' && *p <= '9') p++; }
            if (*p == 'e' || *p == 'E') {
                v.integral = false; p++;
                if (*p == '+' || *p == '-') p++;
                if (*p < '0' || *p > '9') bad();
                while (*p >= '0' && *p <= '9') p++;
            }
            v.type = JVal::Number; v.str.assign(s, p);
            return v;
        }
        bad();
    }
    std::string string_literal() {
        std::string out;
        p++;                                                        // opening quote
        for (;;) {
            unsigned char ch = (unsigned char)*p;
            if (ch == 0 || ch < 0x20) bad();
            if (ch == '"') { p++; return out; }
            if (ch == '\\') {
                p++;
                switch (*p) {
                    case '"': out += '"'; break;   case '\\': out += '\\'; break; case '/': out += '/'; break;
                    case 'b': out += '\b'; break;  case 'f': out += '\f'; break;  case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;  case 't': out += '\t'; break;
                    case 'u': {
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; i++) {
                            char h = p[i];
                            cp <<= 4;
                            if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
                            else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
                            else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
                            else bad();
                        }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xc0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3f)); }
                        else { out += (char)(0xe0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3f)); out += (char)(0x80 | (cp & 0x3f)); }
                        break;
                    }
                    default: bad();
                }
                p++;
                continue;
            }
            out += (char)ch;
            p++;
        }
    }
};

// G1PointAffineFromJson / G2PointAffineFromJson (src/groth16.cpp:254-268): x and y only; (0, 0) is the point at infinity
G1A g1_from_json(const JVal& v) {
    G1A p{f1_from_decimal(v.at((size_t)0).string()), f1_from_decimal(v.at((size_t)1).string()), false};
    p.inf = is0(p.x) && is0(p.y);
    return p;
}
G2A g2_from_json(const JVal& v) {
    G2A q{F2{f1_from_decimal(v.at((size_t)0).at((size_t)0).string()), f1_from_decimal(v.at((size_t)0).at((size_t)1).string())},
          F2{f1_from_decimal(v.at((size_t)1).at((size_t)0).string()), f1_from_decimal(v.at((size_t)1).at((size_t)1).string())}, false};
    q.inf = is0(q.x) && is0(q.y);
    return q;
}

// decimal string -> value mod r as a plain 256-bit integer (E.fr.fromString, then fromMontgomery in verify())
void fr_plain_from_decimal(u32 out[8], const std::string& s) {
    if (s.empty()) throw std::invalid_argument("not a number");
    u32 ten[8] = {10, 0, 0, 0, 0, 0, 0, 0};
    const Fr t = from_normal<FrParams>(ten);
    Fr acc = fp_zero<FrParams>();
    for (char ch : s) {
        if (ch < '0' || ch > '9') throw std::invalid_argument("not a number");
        u32 d[8] = {(u32)(ch - '0'), 0, 0, 0, 0, 0, 0, 0};
        acc = canon(add(canon(mul(acc, t)), from_normal<FrParams>(d)));
    }
    to_normal(out, acc);
}

struct Inputs { std::vector<std::vector<u32>> plain; };
Inputs parse_inputs(const char* text) {                             // src/verifier.cpp:59-86
    Inputs in;
    try {
        JVal j = JParser(text).parse_document();
        if (j.type != JVal::Array || j.arr.empty()) throw std::invalid_argument("invalid inputs data");
        for (const JVal& e : j.arr) {
            std::vector<u32> w(8);
            fr_plain_from_decimal(w.data(), e.string());
            in.plain.push_back(std::move(w));
        }
    } catch (...) { throw std::invalid_argument("invalid inputs data"); }
    return in;
}

struct Groth16Proof { G1A a, c; G2A b; };
struct Groth16Key { G1A alpha; G2A beta, gamma, delta; std::vector<G1A> ic; };
struct UltraProof { G1A a, final_commit, round_commit; G2A b; };
struct UltraKey { G1A alpha, ic_rand; G2A beta, gamma, final_delta, round_delta; std::vector<G1A> ic; };

Groth16Proof parse_proof(const char* text) {                        // :16-36
    try {
        JVal j = JParser(text).parse_document();
        if (j.at("protocol").string() != "groth16") throw std::invalid_argument("invalid proof data");
        return Groth16Proof{g1_from_json(j.at("pi_a")), g1_from_json(j.at("pi_c")), g2_from_json(j.at("pi_b"))};
    } catch (...) { throw std::invalid_argument("invalid proof data"); }
}
UltraProof parse_ultra_proof(const char* text) {                    // :38-57
    try {
        JVal j = JParser(text).parse_document();
        if (j.at("protocol").string() != "ultragroth") throw std::invalid_argument("invalid proof data");
        return UltraProof{g1_from_json(j.at("pi_a")), g1_from_json(j.at("pi_f")), g1_from_json(j.at("pi_r")), g2_from_json(j.at("pi_b"))};
    } catch (...) { throw std::invalid_argument("invalid proof data"); }
}
void check_key_header(const JVal& j, const char* protocol) {
    const JVal& np = j.at("nPublic");
    if (np.type != JVal::Number) throw std::invalid_argument("nPublic");
    if (j.at("protocol").string() != protocol || j.at("curve").string() != "bn128") throw std::invalid_argument("protocol");
}
std::vector<G1A> parse_ic(const JVal& j) {
    std::vector<G1A> ic;
    const JVal& a = j.at("IC");
    if (a.type == JVal::Array) for (const JVal& e : a.arr) ic.push_back(g1_from_json(e));
    else if (a.type == JVal::Object) for (const auto& kv : a.obj) ic.push_back(g1_from_json(kv.second));   // json::items()
    if (ic.empty()) throw std::invalid_argument("IC");
    return ic;
}
Groth16Key parse_key(const char* text) {                            // :88-116
    try {
        JVal j = JParser(text).parse_document();
        check_key_header(j, "groth16");
        Groth16Key k{g1_from_json(j.at("vk_alpha_1")), g2_from_json(j.at("vk_beta_2")), g2_from_json(j.at("vk_gamma_2")),
                     g2_from_json(j.at("vk_delta_2")), {}};
        k.ic = parse_ic(j);
        return k;
    } catch (...) { throw std::invalid_argument("invalid verification key data"); }
}
UltraKey parse_ultra_key(const char* text) {                        // :118-146, src/ultra_groth.cpp:543-563
    try {
        JVal j = JParser(text).parse_document();
        check_key_header(j, "ultragroth");
        UltraKey k{g1_from_json(j.at("vk_alpha_1")), {}, g2_from_json(j.at("vk_beta_2")), g2_from_json(j.at("vk_gamma_2")),
                   g2_from_json(j.at("vk_delta_c2_2")), g2_from_json(j.at("vk_delta_c1_2")), {}};
        k.ic = parse_ic(j);
        k.ic_rand = g1_from_json(j.at("IC_rand"));
        return k;
    } catch (...) { throw std::invalid_argument("invalid verification key data"); }
}

// sum_i input_i IC[i+1]  (the loops at src/groth16.cpp:322-332, src/ultra_groth.cpp:589-601)
G1XYZZ inputs_combination(const Inputs& in, const std::vector<G1A>& ic) {
    G1XYZZ acc = xyzz_inf<Fq>();
    for (size_t i = 0; i < in.plain.size(); i++) acc = xyzz_add(acc, xyzz_mul_scalar(to_xyzz(ic[i + 1]), in.plain[i].data(), 256));
    return acc;
}

// derive_challenge (src/ultra_groth.cpp:33-58): keccak256(x_BE32 || y_BE32) of the round commitment as a big-endian
// integer, reduced mod r; returned as a plain integer
void derive_challenge_plain(u32 out[8], const G1A& commit) {
    u32 x[8], y[8];
    f1_to_plain(x, commit.x); f1_to_plain(y, commit.y);
    uint8_t buf[64], ch[32];
    for (int i = 0; i < 32; i++) {
        buf[i] = (uint8_t)(x[7 - (i >> 2)] >> (24 - 8 * (i & 3)));
        buf[32 + i] = (uint8_t)(y[7 - (i >> 2)] >> (24 - 8 * (i & 3)));
    }
    keccak256(ch, buf, 64);
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 32; i++) w[7 - (i >> 2)] |= (u32)ch[i] << (24 - 8 * (i & 3));
    to_normal(out, from_normal<FrParams>(w));                       // from_normal reduces values >= r
}

bool all_on_curve(std::initializer_list<const G1A*> g1, std::initializer_list<const G2A*> g2) {
    for (const G1A* p : g1) if (!g1_on_curve(*p)) return false;
    for (const G2A* q : g2) if (!g2_on_curve(*q)) return false;
    return true;
}

void copy_error(char* dst, unsigned long cap, const char* msg) {
    if (dst && cap) strncpy(dst, msg, cap);                         // as the reference: no terminator added beyond strncpy's
}

}  // namespace

extern "C" {

int groth16_verify(const char* proof, const char* inputs, const char* verification_key, char* error_msg,
                   unsigned long error_msg_maxsize) {
    try {
        if (!proof || !inputs || !verification_key) throw std::invalid_argument("null argument");
        Groth16Proof pr = parse_proof(proof);
        Inputs in = parse_inputs(inputs);
        Groth16Key key = parse_key(verification_key);
        if (in.plain.size() + 1 != key.ic.size()) throw std::invalid_argument("len(inputs)+1 != len(vk.IC)");     // src/groth16.cpp:318-320
        // points that are not on their curve can only come from malformed data: no pairing is defined for them
        if (!all_on_curve({&pr.a, &pr.c, &key.alpha}, {&pr.b, &key.beta, &key.gamma, &key.delta})) return VERIFIER_INVALID_PROOF;
        for (const G1A& p : key.ic) if (!g1_on_curve(p)) return VERIFIER_INVALID_PROOF;
        G1A vkx = from_xyzz(xyzz_add(inputs_combination(in, key.ic), to_xyzz(key.ic[0])));
        bool ok = pairing_check({pr.a, g1_neg(key.alpha), g1_neg(vkx), g1_neg(pr.c)}, {pr.b, key.beta, key.gamma, key.delta});
        return ok ? VERIFIER_VALID_PROOF : VERIFIER_INVALID_PROOF;
    } catch (std::exception& e) {
        copy_error(error_msg, error_msg_maxsize, e.what());
        return VERIFIER_ERROR;
    } catch (...) {
        copy_error(error_msg, error_msg_maxsize, "unknown error");
        return VERIFIER_ERROR;
    }
}

int ultra_groth_verify(const char* proof, const char* inputs, const char* verification_key, char* error_msg,
                       unsigned long error_msg_maxsize) {
    try {
        if (!proof || !inputs || !verification_key) throw std::invalid_argument("null argument");
        UltraProof pr = parse_ultra_proof(proof);
        Inputs in = parse_inputs(inputs);
        UltraKey key = parse_ultra_key(verification_key);
        if (in.plain.size() + 1 != key.ic.size()) throw std::invalid_argument("len(inputs) != len(vk.IC)");       // src/ultra_groth.cpp:585-587
        if (!all_on_curve({&pr.a, &pr.final_commit, &pr.round_commit, &key.alpha, &key.ic_rand},
                          {&pr.b, &key.beta, &key.gamma, &key.final_delta, &key.round_delta})) return VERIFIER_INVALID_PROOF;
        for (const G1A& p : key.ic) if (!g1_on_curve(p)) return VERIFIER_INVALID_PROOF;
        u32 rand[8];
        derive_challenge_plain(rand, pr.round_commit);                                                             // :603-609
        G1XYZZ vk = xyzz_add(inputs_combination(in, key.ic), to_xyzz(key.ic[0]));
        vk = xyzz_add(vk, xyzz_mul_scalar(to_xyzz(key.ic_rand), rand, 256));                                       // :609-612
        G1A vkx = from_xyzz(vk);
        bool ok = pairing_check({pr.a, g1_neg(key.alpha), g1_neg(vkx), g1_neg(pr.final_commit), g1_neg(pr.round_commit)},
                                {pr.b, key.beta, key.gamma, key.final_delta, key.round_delta});
        return ok ? VERIFIER_VALID_PROOF : VERIFIER_INVALID_PROOF;
    } catch (std::exception& e) {
        copy_error(error_msg, error_msg_maxsize, e.what());
        return VERIFIER_ERROR;
    } catch (...) {
        copy_error(error_msg, error_msg_maxsize, "unknown error");
        return VERIFIER_ERROR;
    }
}

}  // extern "C"
