// prover_api.cpp -- outer C-ABI (include/prover.h): Groth16 and UltraGroth provers on top of the inner
// ug_* device ABI. Host side of the hot path only: parsing, orchestration, blinding, JSON.
//
// Follows, step for step, the reference's
//   Groth16Prover / UltraGrothProver wrappers     src/prover.cpp:144-309
//   Groth16::Prover::prove                        src/groth16.cpp:48-203      (S1..S13)
//   UltraGroth::Prover::{execute_round, execute_final_round, prove, compute_lookup, derive_challenge}
//                                                 src/ultra_groth.cpp:33-106,161-462
//   Proof::toJson, BuildPublicString              src/groth16.cpp:217-250, src/ultra_groth.cpp:476-513,
//                                                 src/prover.cpp:89-117
//   extern "C" entry points and error mapping     src/prover.cpp:311-891
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <exception>
#include <functional>
#include <future>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "ec.hpp"
#include "host_util.hpp"
#include "../../include/prover.h"
#include "../../include/ultragroth_hip.h"

using namespace ug;
using namespace ughost;

namespace {

class ShortBufferException : public std::invalid_argument {
public:
    explicit ShortBufferException(const std::string& m) : std::invalid_argument(m) {}
};
// a rank of a bucket-class layout could not get its window tables: the one failure a many-device create answers with another layout
class TablesUnavailable : public std::runtime_error {
public:
    explicit TablesUnavailable(const std::string& m) : std::runtime_error(m) {}
};
class InvalidWitnessLengthException : public std::invalid_argument {
public:
    explicit InvalidWitnessLengthException(const std::string& m) : std::invalid_argument(m) {}
};

void copyError(char* error_msg, unsigned long long maxsize, const char* what) {
    if (error_msg) strncpy(error_msg, what, maxsize);
}
void ugCheck(int rc) {
    if (rc != UG_OK) throw std::runtime_error(ug_last_error());
}

// Work queued on the prover's contexts must not outlive a failing call: the queued MSMs hold pointers into the caller's
// stack frame and the kernels read the leased witness buffer. Armed while a call has work in flight; on unwind it waits
// for the device and drops what was queued (ug_ctx_abandon), before the witness lease and the turn are given back.
struct QueueGuard {
    ug_ctx *a, *b;
    bool armed = true;
    QueueGuard(ug_ctx* a_, ug_ctx* b_ = nullptr) : a(a_), b(b_) {}
    ~QueueGuard() { if (armed) { ug_ctx_abandon(a); ug_ctx_abandon(b); } }
    void done() { armed = false; }
    QueueGuard(const QueueGuard&) = delete;
    QueueGuard& operator=(const QueueGuard&) = delete;
};
// `around(false)` of ProverBase::proveTurn is owed once `around(true)` has been called, also when the proof throws
struct AroundGuard {
    const std::function<void(bool)>& fn;
    bool open = false;
    explicit AroundGuard(const std::function<void(bool)>& f) : fn(f) {}
    void begin() { if (fn) { fn(true); open = true; } }
    void end() { if (open) { open = false; fn(false); } }
    ~AroundGuard() { if (open) { try { fn(false); } catch (...) {} } }
};

constexpr unsigned long long PROOF_MIN_GROTH16 = 810, PROOF_MIN_ULTRA = 1400;
unsigned long long publicMin(unsigned long long count) { return count * 82 + 4; }

void checkBufferSizes(unsigned long long proofCalc, const unsigned long long* proofSize,
                      unsigned long long publicCalc, const unsigned long long* publicSize, const std::string& type) {
    if (*proofSize < proofCalc)
        throw ShortBufferException("Proof buffer is too short. " + type + " size: " + std::to_string(proofCalc) +
                                   ", actual size: " + std::to_string(*proofSize));
    if (*publicSize < publicCalc)
        throw ShortBufferException("Public buffer is too short. " + type + " size: " + std::to_string(publicCalc) +
                                   ", actual size: " + std::to_string(*publicSize));
}

// ---- host curve helpers on reference-format records ---------------------------------------------------
G1XYZZ g1FromRecord(const uint8_t* rec) {
    u32 w[16];
    memcpy(w, rec, 64);
    u32 o = 0;
    for (int i = 0; i < 16; i++) o |= w[i];
    if (!o) return xyzz_inf<Fq>();
    return xyzz_from_affine(from_mont256<FqParams>(w), from_mont256<FqParams>(w + 8));
}
G2XYZZ g2FromRecord(const uint8_t* rec) {
    u32 w[32];
    memcpy(w, rec, 128);
    u32 o = 0;
    for (int i = 0; i < 32; i++) o |= w[i];
    if (!o) return xyzz_inf<Fq2>();
    Fq2 x, y;
    x.a = from_mont256<FqParams>(w); x.b = from_mont256<FqParams>(w + 8);
    y.a = from_mont256<FqParams>(w + 16); y.b = from_mont256<FqParams>(w + 24);
    return xyzz_from_affine(x, y);
}
void g1ToRecord(uint8_t* rec, const G1XYZZ& p) {
    if (is_inf(p)) { memset(rec, 0, 64); return; }
    Fq x, y;
    xyzz_to_affine(x, y, p);
    u32 w[16];
    to_mont256(w, x); to_mont256(w + 8, y);
    memcpy(rec, w, 64);
}
void g2ToRecord(uint8_t* rec, const G2XYZZ& p) {
    if (is_inf(p)) { memset(rec, 0, 128); return; }
    Fq2 x, y;
    xyzz_to_affine(x, y, p);
    u32 w[32];
    to_mont256(w, x.a); to_mont256(w + 8, x.b); to_mont256(w + 16, y.a); to_mont256(w + 24, y.b);
    memcpy(rec, w, 128);
}
// E.f1.toString of a Montgomery coordinate: plain value, decimal
std::string coordString(const uint8_t* mont32) {
    u32 w[8], n[8];
    memcpy(w, mont32, 32);
    to_normal(n, from_mont256<FqParams>(w));
    return toDecimal(reinterpret_cast<const uint8_t*>(n));
}
std::string g1Json(const uint8_t* rec) {
    return "[\"" + coordString(rec) + "\",\"" + coordString(rec + 32) + "\",\"1\"]";
}
std::string g2Json(const uint8_t* rec) {
    return "[[\"" + coordString(rec) + "\",\"" + coordString(rec + 32) + "\"],[\"" + coordString(rec + 64) + "\",\"" +
           coordString(rec + 96) + "\"],[\"1\",\"0\"]]";
}

// one blinding scalar: 31 random bytes, top byte zero (src/groth16.cpp:158-166)
void drawBlinding(uint8_t out[32]) {
    memset(out, 0, 32);
    randomBytes(out, 31);
}

// S12: the seven single scalar multiplications and the sums (src/groth16.cpp:168-195). Four of the seven need only
// the verification key and the blinding scalars: blindingTerms() forms them on four host threads, and a prover that
// has drawn r and s before its device work starts runs it concurrently with that work.
struct BlindingTerms {
    G1XYZZ rDelta1, sDelta1, rsDelta1;
    G2XYZZ sDelta2;
};
BlindingTerms blindingTerms(const ZkeyHeader& h, const uint8_t r[32], const uint8_t s[32]) {
    u32 rw[8], sw[8], rsw[8];
    memcpy(rw, r, 32); memcpy(sw, s, 32);
    // :191-192  rs = toMontgomery(MMul(r, s)) = r * s mod q as a plain integer
    to_normal(rsw, mul(from_normal<FrParams>(rw), from_normal<FrParams>(sw)));
    const G1XYZZ delta1 = g1FromRecord(h.delta1);
    const G2XYZZ delta2 = g2FromRecord(h.delta2);
    BlindingTerms t;
    auto f2 = std::async(std::launch::async, [&] { return xyzz_mul_scalar_w4(delta2, sw); });        // :176-177
    auto fr = std::async(std::launch::async, [&] { return xyzz_mul_scalar_w4(delta1, rw); });        // :172-173
    auto fs = std::async(std::launch::async, [&] { return xyzz_mul_scalar_w4(delta1, sw); });        // :180-181
    t.rsDelta1 = xyzz_mul_scalar_w4(delta1, rsw);                                                    // :194-195
    t.sDelta1 = fs.get(); t.rDelta1 = fr.get(); t.sDelta2 = f2.get();
    return t;
}
// All five sums come in as affine records; outputs are the affine records of pi_a, pi_b, pi_c.
void blind(uint8_t* outA, uint8_t* outB, uint8_t* outC, const uint8_t* sumA, const uint8_t* sumB1, const uint8_t* sumB2,
           const uint8_t* sumC, const uint8_t* sumH, const ZkeyHeader& h, const uint8_t r[32], const uint8_t s[32],
           const BlindingTerms& t, const G1XYZZ* extraSubtract) {
    u32 rw[8], sw[8];
    memcpy(rw, r, 32); memcpy(sw, s, 32);
    G1XYZZ pi_a = g1FromRecord(sumA), pib1 = g1FromRecord(sumB1), pi_c = g1FromRecord(sumC), pih = g1FromRecord(sumH);
    G2XYZZ pi_b = g2FromRecord(sumB2);
    pi_a = xyzz_add(xyzz_add(pi_a, g1FromRecord(h.alpha1)), t.rDelta1);   // :171-173
    pi_b = xyzz_add(xyzz_add(pi_b, g2FromRecord(h.beta2)), t.sDelta2);    // :175-177
    pib1 = xyzz_add(xyzz_add(pib1, g1FromRecord(h.beta1)), t.sDelta1);    // :179-181
    auto fa = std::async(std::launch::async, [&] { return xyzz_mul_scalar_w4(pi_a, sw); });          // :185-186
    G1XYZZ rB1 = xyzz_mul_scalar_w4(pib1, rw);                             // :188-189
    pi_c = xyzz_add(pi_c, pih);                                           // :183
    pi_c = xyzz_add(pi_c, fa.get());
    pi_c = xyzz_add(pi_c, rB1);
    pi_c = xyzz_add(pi_c, xyzz_neg(t.rsDelta1));                          // :194-195
    if (extraSubtract) pi_c = xyzz_add(pi_c, xyzz_neg(*extraSubtract));   // ultra_groth.cpp:386-388
    g1ToRecord(outA, pi_a); g2ToRecord(outB, pi_b); g1ToRecord(outC, pi_c);   // :197-200
}

// BuildPublicString (src/prover.cpp:106-117; UltraGroth variant :89-105 skips rand_indx):
// toMontgomery + toString = the value reduced mod r, printed in decimal. An empty list dumps as "null".
std::string publicJson(const uint8_t* w, uint32_t nPublic, uint32_t skip) {
    std::string out = "[";
    bool first = true;
    for (uint32_t i = 1; i <= nPublic; i++) {
        if (i == skip) continue;
        u32 v[8], n[8];
        memcpy(v, w + (size_t)i * 32, 32);
        to_normal(n, from_normal<FrParams>(v));
        if (!first) out += ",";
        first = false;
        out += "\"" + toDecimal(reinterpret_cast<const uint8_t*>(n)) + "\"";
    }
    if (first) return "null";
    return out + "]";
}

int deviceFromEnv() {
    const char* e = getenv("ULTRAGROTH_DEVICE");
    return e ? atoi(e) : 0;
}

struct Range { uint64_t lo, hi; };
Range shardRange(uint64_t n, int rank, int count) {
    return Range{n * (uint64_t)rank / (uint64_t)count, n * (uint64_t)(rank + 1) / (uint64_t)count};
}
// What one rank of a many-device Groth16 prover owns (made by shardLayouts below). Two ways to cut the
// witness products (S1-S4, src/groth16.cpp:55-64) over the ranks, and any product of the two:
//   by base-point range   rank k holds the points and scalars of a contiguous range `w` of the witness-indexed sections
//   by bucket class       the ranks of a group hold the SAME range `w` (all of it when there is one group) with its whole window
//                         tables, and each takes the entries of its residues [r0, r0 + cnt) of the bucket ids mod 2^qLog
//                         (ug_schedule_set_classes; the lowest bucket ids by scalar range `sp` instead)
// The H product (S10, :154) is always cut by base-point range `h` (h is made in slices); chain ranks may get none of it.
struct ShardLayout {
    Range w{0, 0}, h{0, 0};
    int qLog = 0;
    uint32_t r0 = 0, cnt = 1;
    Range sp{0, 0};
    unsigned chains = 0;          // bit k: this rank runs chain k of the H polynomial (:66-140)
};
constexpr uint32_t SHARD_SPECIALS = 16;      // digits 1 .. 16 of every window are owned by scalar range (bits and small constants of a witness)
// bucket classes per group of B ranks: about sixteen residues per rank (its share of the entries in steps of ~ 6 % of itself, and
// 2 x 16 + 1 points per result block), 128 at most
inline int shardQLog(int B) { int q = 4; while (q < 7 && (1 << q) < 16 * B) q++; return q; }

// ---- common device-side state of a prover -------------------------------------------------------------------
struct DeviceProver {
    ug_ctx* ctx = nullptr;
    ug_ctx* ctx2 = nullptr;      // second stream (Groth16): the H-polynomial branch runs beside the witness MSMs
    ug_bases *A = nullptr, *B1 = nullptr, *B2 = nullptr, *C = nullptr, *H = nullptr, *roundC = nullptr;
    ug_bases* G = nullptr;       // the G1 sets that share the witness scalars as ONE interleaved group ([A | B1 | C] for Groth16,
                                 // [A | B1] for UltraGroth; A, B1 (and C) are then null): one gather and one accumulation
                                 // launch for all of them (ug_bases_create_group_g1). ULTRAGROTH_FUSED=0 keeps separate sets.
    // SPARSE B (Groth16, one device): in real circuits many signals never appear on the B side of a constraint, so B1 / B2 hold
    // points at infinity for them. When at most three quarters of the B points are real, the prover keeps only those (Bc1, Bc2:
    // compacted sets), the list of their signal numbers (bIdx) and a schedule of its own over the gathered scalars (wB, sB);
    // the G1 group is then [A | C]. The dense form (B1 in the group, B2 over the witness schedule) stays for everything else.
    ug_bases *Bc1 = nullptr, *Bc2 = nullptr;
    ug_index* bIdx = nullptr;
    ug_dvec* wB = nullptr;
    ug_schedule* sB = nullptr;
    int tableB = 0;              // window width of the compacted sets' tables (0: classic windows)
    ug_hpoly* hp = nullptr;
    ug_dvec *w = nullptr, *h = nullptr, *aux = nullptr;
    ug_dvec* w2 = nullptr;       // second witness buffer (Groth16): the next proof's witness is staged here while a proof runs
                                 // (written once, under the prover's slot lock; the device part reads Groth16Prover::wCur_)
    ug_schedule *sw = nullptr, *sh = nullptr, *saux = nullptr;
    ug_index *roundIdx = nullptr, *finalIdx = nullptr;      // UltraGroth: zkey sections 10 and 11, resident
    ~DeviceProver() {
        ug_index_destroy(roundIdx); ug_index_destroy(finalIdx); ug_index_destroy(bIdx);
        ug_schedule_destroy(sw); ug_schedule_destroy(sh); ug_schedule_destroy(saux); ug_schedule_destroy(sB);
        ug_dvec_destroy(w); ug_dvec_destroy(w2); ug_dvec_destroy(h); ug_dvec_destroy(aux); ug_dvec_destroy(wB);
        ug_bases_destroy(Bc1); ug_bases_destroy(Bc2);
        ug_hpoly_destroy(hp);
        ug_bases_destroy(A); ug_bases_destroy(B1); ug_bases_destroy(B2); ug_bases_destroy(C); ug_bases_destroy(H);
        ug_bases_destroy(roundC); ug_bases_destroy(G);
        ug_ctx_destroy(ctx);
        ug_ctx_destroy(ctx2);
    }
};

// ---- two witness buffers per prover: the next call's witness is staged while a proof runs -----------------------------
// A buffer is leased from the moment a host thread starts to fill it until the proof that reads it has left the device.
// The second buffer (nVars * 32 bytes) is allocated the first time two calls overlap; if that fails the calls simply take
// turns on the first. Lock order everywhere: witness lease -> stageMutex (released again) -> the card's lock -> proveMutex.
struct StagedWitness {
    ug_dvec* buf = nullptr; bool leased = false;
    std::vector<uint8_t> publicPart; double uploadMs = 0;
    std::vector<uint32_t> chunks, freq, wIdx, pIdx;            // UltraGroth: uwtns sections 3-6
};
class WitnessBuffers {
public:
    // first / second: the two owners in the prover's DeviceProver (w, w2); *first holds the buffer made at create
    void attach(ug_ctx* ctx, uint64_t n, ug_dvec** first, ug_dvec** second) {
        ctx_ = ctx; n_ = n; own_[0] = first; own_[1] = second; slots_[0].buf = *first;
    }
    int lease() {
        std::unique_lock<std::mutex> lk(mutex_);
        for (;;) {
            for (int k = 0; k < 2; k++) {
                StagedWitness& sl = slots_[k];
                if (sl.leased) continue;
                if (!sl.buf) {
                    if (allocFailed_) continue;
                    if (ug_dvec_create(ctx_, n_, &sl.buf) != UG_OK) { sl.buf = nullptr; allocFailed_ = true; continue; }
                    (*own_[0] ? *own_[1] : *own_[0]) = sl.buf;            // (the owner that is free)
                }
                sl.leased = true;
                return k;
            }
            free_.wait(lk);
        }
    }
    void release(int k) {
        { std::lock_guard<std::mutex> lk(mutex_); slots_[k].leased = false; }
        free_.notify_all();
    }
    StagedWitness& operator[](int k) { return slots_[k]; }
    // the buffer that is neither the current one nor being filled goes (it comes back with the next overlap); the caller
    // holds the prover's proveMutex
    void trim(const ug_dvec* current) {
        std::lock_guard<std::mutex> lk(mutex_);
        for (StagedWitness& sl : slots_) {
            if (!sl.buf || sl.leased || sl.buf == current) continue;
            if (*own_[0] == sl.buf) { *own_[0] = *own_[1]; *own_[1] = nullptr; } else if (*own_[1] == sl.buf) *own_[1] = nullptr;
            ug_dvec_destroy(sl.buf);
            sl.buf = nullptr;
            allocFailed_ = false;
        }
    }
    std::mutex stageMutex;          // one staging copy at a time (the uploader is the context's)
private:
    ug_ctx* ctx_ = nullptr; uint64_t n_ = 0; ug_dvec** own_[2] = {nullptr, nullptr};
    StagedWitness slots_[2];
    std::mutex mutex_;
    std::condition_variable free_;
    bool allocFailed_ = false;
};
struct WitnessLease {
    WitnessBuffers& b; int slot;
    explicit WitnessLease(WitnessBuffers& b_) : b(b_), slot(b_.lease()) {}
    ~WitnessLease() { b.release(slot); }
    WitnessLease(const WitnessLease&) = delete;
    WitnessLease& operator=(const WitnessLease&) = delete;
    StagedWitness& operator*() { return b[slot]; }
};

// ---- fixed-base window tables ---------------------------------------------------------------------------------
// A zkey's points never change between proofs, so a created prover may trade HBM for work: with tables 2^(c j) P_i
// every window digit of an MSM lands in one bucket set (ultragroth_hip.h: ug_bases_precompute). A group is the base
// sets that share one schedule (they must share the window width). Tables are built when ULTRAGROTH_TABLES is not
// "0", a group has at least 2^14 scalars (below that the classic windows are cheaper) and everything fits the free
// device memory with room left for schedules, buckets and NTT vectors; otherwise the classic path runs.
struct TableGroup {
    std::vector<ug_bases*> g1, g2;     // sets of the group with their point counts
    std::vector<uint64_t> n1, n2;
    uint64_t scalars = 0;              // scalars per schedule
    int* c = nullptr;                  // out: window width, 0 = classic
};
constexpr uint64_t TABLES_MIN_SCALARS = (uint64_t)1 << 14, TABLES_MAX_SCALARS = (uint64_t)1 << 26;

thread_local bool g_oneShotProver = false;      // create + one prove + destroy (groth16_prover, the CLIs): tables cannot pay

thread_local bool g_registryCreate = false;     // the resident multi-circuit prover decides about tables itself, after create

thread_local bool g_deferTables = false;        // set around groth16_prover_create's constructor call: the one prover that may hand out
                                                // proofs before its window tables exist (every caller of it goes through proveTurn)

// what the tables of all groups would take (0: no group qualifies) and the schedule workspace they imply
uint64_t tablesNeed(const std::vector<TableGroup>& groups, std::vector<int>& width, uint64_t* workspaceOut) {
    uint64_t need = 0, workspace = (uint64_t)2 << 30;
    width.assign(groups.size(), 0);
    for (size_t k = 0; k < groups.size(); k++) {
        const TableGroup& g = groups[k];
        if (g.scalars < TABLES_MIN_SCALARS || g.scalars > TABLES_MAX_SCALARS) continue;
        int c = ug_msm_table_window(g.scalars);
        width[k] = c;
        for (uint64_t n : g.n1) need += ug_bases_tables_bytes(n, 0, c);
        for (uint64_t n : g.n2) need += ug_bases_tables_bytes(n, 1, c);
        workspace += 24 * g.scalars * (uint64_t)((255 + c - 1) / c);       // sorted entries + segment slots
    }
    if (workspaceOut) *workspaceOut = workspace;
    return need;
}
// builds the tables when the environment allows them and `need + workspace` fits the free device memory and `limit`
// (bytes the caller is ready to spend, ~0 = no limit of its own); returns the bytes taken (0: classic windows stay)
uint64_t planWindowTables(ug_ctx* ctx, std::vector<TableGroup>& groups, uint64_t limit = ~(uint64_t)0, bool force = false) {
    for (auto& g : groups) *g.c = 0;
    const char* e = getenv("ULTRAGROTH_TABLES");
    if (e && e[0] == '0') return 0;
    if (!force && (g_registryCreate || (g_oneShotProver && !(e && e[0] == '2')))) return 0;     // "2": tables even for one-shot calls
    std::vector<int> width;
    uint64_t workspace = 0;
    const uint64_t need = tablesNeed(groups, width, &workspace);
    if (!need) return 0;
    uint64_t freeB = 0, totalB = 0;
    ugCheck(ug_ctx_mem_info(ctx, &freeB, &totalB));
    if (need + workspace > freeB || need > limit) return 0;
    uint64_t taken = 0;
    for (size_t k = 0; k < groups.size(); k++) {
        if (!width[k]) continue;
        bool ok = true;
        for (ug_bases* b : groups[k].g1) ok = ok && ug_bases_precompute(b, width[k]) == 0;
        for (ug_bases* b : groups[k].g2) ok = ok && ug_bases_precompute(b, width[k]) == 0;
        if (ok) {
            *groups[k].c = width[k];
            for (uint64_t n : groups[k].n1) taken += ug_bases_tables_bytes(n, 0, width[k]);
            for (uint64_t n : groups[k].n2) taken += ug_bases_tables_bytes(n, 1, width[k]);
        } else {                               // a group left half built gives its memory back and keeps the classic windows
            for (ug_bases* b : groups[k].g1) ug_bases_drop_tables(b);
            for (ug_bases* b : groups[k].g2) ug_bases_drop_tables(b);
        }
    }
    return taken;
}
// The same decision taken BEFORE the base sets exist (only their sizes are known): widths[k] = the table width of group k,
// or 0. `otherBytes` = what the caller is still going to allocate besides the tables (the points themselves, matrix,
// vectors). With the widths known, every set is created together with its tables (ug_bases_create_tables_*), whose build
// overlaps the upload of the next section.
std::vector<int> planTableWidthsAhead(ug_ctx* ctx, const std::vector<TableGroup>& groups, uint64_t otherBytes) {
    std::vector<int> none(groups.size(), 0);
    const char* e = getenv("ULTRAGROTH_TABLES");
    if (e && e[0] == '0') return none;
    if (g_registryCreate || (g_oneShotProver && !(e && e[0] == '2'))) return none;
    std::vector<int> width;
    uint64_t workspace = 0;
    const uint64_t need = tablesNeed(groups, width, &workspace);
    if (!need) return none;
    uint64_t freeB = 0, totalB = 0;
    ugCheck(ug_ctx_mem_info(ctx, &freeB, &totalB));
    if (need + workspace + otherBytes > freeB) return none;
    return width;
}
bool fusedGroups() {
    const char* e = getenv("ULTRAGROTH_FUSED");
    return !(e && e[0] == '0');
}
// The witness products of one schedule, queued on ctx (results after ug_ctx_collect): the G1 sets -- as the interleaved group
// d.G when the prover holds one (outC null: a two-member group), else d.A, d.B1 and, with outC, d.C shifted by shiftC --
// and the G2 set d.B2; g2First queues the G2 product ahead of the G1 ones.
void buildSchedule(ug_schedule* s, const ug_dvec* scalars, uint64_t first, uint64_t count, int tableC);
void enqueueWitnessProducts(DeviceProver& d, ug_ctx* ctx, const ug_schedule* sw, uint8_t* outA, uint8_t* outB1, uint8_t* outB2, uint8_t* outC,
                            int64_t shiftC, bool g2First, const ug_dvec* witness = nullptr) {
    if (d.Bc2) {
        // sparse B: [A | C] over the witness schedule; the B scalars gathered by signal number, a schedule over them, B1 and B2
        // over that one (src/groth16.cpp:58,61 with the points at infinity left out: the sums are the same)
        if (!witness) throw std::logic_error("sparse B: the witness vector is needed");
        if (d.G) {                                      // Groth16: [A | C]
            void* outsG[2] = {outA, outC};
            ugCheck(ug_msm_group_enqueue(ctx, d.G, sw, outsG));
        } else {                                        // UltraGroth: A alone (its C sets have schedules of their own)
            const ug_bases* setA[1] = {d.A};
            void* outsA[1] = {outA};
            ugCheck(ug_msm_batch_enqueue(ctx, 1, setA, sw, nullptr, outsA));
        }
        ugCheck(ug_dvec_gather_index(d.wB, witness, d.bIdx));
        buildSchedule(d.sB, d.wB, 0, ug_dvec_size(d.wB), d.tableB);
        const ug_bases* sets[2] = {d.Bc1, d.Bc2};
        void* outsB[2] = {outB1, outB2};
        ugCheck(ug_msm_batch_enqueue(ctx, 2, sets, d.sB, nullptr, outsB));
        return;
    }
    auto g2 = [&] {
        const ug_bases* sets[1] = {d.B2};
        void* outs[1] = {outB2};
        ugCheck(ug_msm_batch_enqueue(ctx, 1, sets, sw, nullptr, outs));
    };
    // ULTRAGROTH_TAILS=split: a group and a G2 set in one call that runs their tails side by side on two streams
    // (ug_msm_witness_enqueue). Built, exact and measured in round 5 -- a wash (one rank of eight at 2^24: 22.06-22.16 ms against
    // 21.96-22.10; 2^24: 130.96-131.37 against 130.90-131.97): the tails keep the chip's issue slots busy, they are not waiting. The
    // default stays one product after the other.
    static const bool splitTails = [] { const char* e = getenv("ULTRAGROTH_TAILS"); return e && !strcmp(e, "split"); }();
    if (splitTails && !g2First && d.G && d.B2) {
        void* outs[3] = {outA, outB1, outC};
        ugCheck(ug_msm_witness_enqueue(ctx, d.G, d.B2, sw, outs, outB2));
        return;
    }
    if (g2First) g2();
    if (d.G) {
        void* outs[3] = {outA, outB1, outC};
        ugCheck(ug_msm_group_enqueue(ctx, d.G, sw, outs));
    } else {
        const ug_bases* sets[3] = {d.A, d.B1, d.C};
        const int64_t shifts[3] = {0, 0, shiftC};
        void* outs[3] = {outA, outB1, outC};
        ugCheck(ug_msm_batch_enqueue(ctx, outC ? 3 : 2, sets, sw, shifts, outs));
    }
    if (!g2First) g2();
}
void buildSchedule(ug_schedule* s, const ug_dvec* scalars, uint64_t first, uint64_t count, int tableC) {
    if (tableC) ugCheck(ug_schedule_build_tables(s, scalars, first, count, tableC));
    else ugCheck(ug_schedule_build(s, scalars, first, count));
}

// SPARSE B (see DeviceProver): which of the n signals have a real point in B1 or B2? Decided on a sample of every 257th signal first
// (a dense circuit -- every synthetic benchmark circuit -- pays 65 k record tests and nothing else), then counted exactly on eight
// host threads. True when at most three quarters of the points are real: `support` then lists their signal numbers in order and
// b1c / b2c hold the compacted records. A signal is left out only when BOTH its records are all zero: exact for any zkey.
bool sparseBSupport(const uint8_t* pB1, const uint8_t* pB2, uint64_t n, std::vector<uint32_t>& support, std::vector<uint8_t>& b1c,
                    std::vector<uint8_t>& b2c) {
    const char* sb = getenv("ULTRAGROTH_SPARSE_B");
    if ((sb && sb[0] == '0') || !fusedGroups() || n < ((uint64_t)1 << 14) || n >= ((uint64_t)1 << 32)) return false;
    auto real = [&](uint64_t i) {
        // (the first words of x decide for any real point; the whole records are compared only when they are zero)
        uint64_t a, b;
        memcpy(&a, pB2 + i * 128, 8); memcpy(&b, pB1 + i * 64, 8);
        if (a | b) return true;
        return !(ug_points_all_infinity(pB2 + i * 128, 1, 128) && ug_points_all_infinity(pB1 + i * 64, 1, 64));
    };
    uint64_t seen = 0, hit = 0;
    for (uint64_t i = 0; i < n; i += 257) { seen++; hit += real(i) ? 1 : 0; }
    if (hit * 10 > seen * 8) return false;            // the sample says more than ~80 % real
    const int T = 8;
    std::vector<std::vector<uint32_t>> found(T);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const uint64_t lo = n * (uint64_t)t / T, hi = n * (uint64_t)(t + 1) / T;
                found[t].reserve((size_t)((hi - lo) * hit / seen + 1024));
                for (uint64_t i = lo; i < hi; i++) if (real(i)) found[t].push_back((uint32_t)i);
            });
        for (auto& x : th) x.join();
    }
    size_t total = 0;
    for (auto& v : found) total += v.size();
    if (!total || (uint64_t)total * 4 > n * 3) return false;
    support.clear(); support.reserve(total);
    for (auto& v : found) support.insert(support.end(), v.begin(), v.end());
    b1c.resize(total * 64); b2c.resize(total * 128);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (size_t j = total * (size_t)t / T; j < total * (size_t)(t + 1) / T; j++) {
                memcpy(b1c.data() + j * 64, pB1 + (uint64_t)support[j] * 64, 64);
                memcpy(b2c.data() + j * 128, pB2 + (uint64_t)support[j] * 128, 128);
            }
        });
    for (auto& x : th) x.join();
    return true;
}

struct ProverBase {        // what the extern "C" layer stores behind the opaque handle
    std::mutex proveMutex;
    virtual ~ProverBase() {}
    virtual void prove(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub) = 0;
    // One proof for a caller that may share the object with other host threads (the extern "C" prove calls and the
    // registry come through here): the calls take turns on the device -- `device`, when given, is the lock of everything
    // on the card (registry), `around(true/false)` brackets the device part. A prover may do the part of a proof that
    // needs no device turn (parsing the witness, copying it into a buffer no running proof uses) BEFORE it takes its turn:
    // Groth16Prover does, so that two host threads proving on one object hide the PCIe copy of one proof behind the
    // kernels of the other.
    typedef std::function<void(bool)> Around;
    virtual void proveTurn(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub,
                           std::mutex* device = nullptr, const Around& around = Around()) {
        std::unique_lock<std::mutex> card;
        if (device) card = std::unique_lock<std::mutex>(*device);
        std::lock_guard<std::mutex> turn(proveMutex);
        AroundGuard bracket(around);
        bracket.begin();
        prove(wtns, wtnsSize, proof, pub);
        bracket.end();
    }
    virtual unsigned long long proofBufferMinSize() const = 0;
    virtual unsigned long long publicBufferMinSize() const = 0;
    virtual void timings(double* msm, double* fft, double* total) const = 0;
    virtual ug_ctx* ctx() = 0;
    double uploadMs = 0;             // host wall time of the last loadWitness
    // HBM residency control (the resident multi-circuit prover, ug_registry_*): fixed-base tables are optional and the
    // first thing given back; the per-proof workspaces (schedules, buckets) are re-grown by the next proof
    uint64_t tableBytes = 0;
    virtual std::vector<TableGroup> tableGroups() = 0;
    uint64_t tablesWouldTake() { std::vector<TableGroup> g = tableGroups(); std::vector<int> w; return tablesNeed(g, w, nullptr); }
    virtual bool tablesReady(bool /*wait*/) { return true; }       // (provers that build their tables in the background override)
    bool buildTables(uint64_t limit) {
        if (tableBytes) return true;
        std::lock_guard<std::mutex> turn(proveMutex);
        std::vector<TableGroup> g = tableGroups();
        tableBytes = planWindowTables(ctx(), g, limit, /*force*/ true);
        return tableBytes != 0;
    }
    void dropTables() {
        tablesReady(true);
        std::lock_guard<std::mutex> turn(proveMutex);
        std::vector<TableGroup> g = tableGroups();
        for (auto& grp : g) {
            for (ug_bases* b : grp.g1) ug_bases_drop_tables(b);
            for (ug_bases* b : grp.g2) ug_bases_drop_tables(b);
            *grp.c = 0;
        }
        tableBytes = 0;
    }
    virtual void trimWorkspaces() = 0;
    // phases of a sharded proof (include/prover.h); both provers implement them
    [[noreturn]] static void noPhase() { throw std::invalid_argument("this prover object does not support the call"); }
    virtual void loadWitness(const void*, unsigned long long) { noPhase(); }
    virtual void loadWitnessPart(const void*, unsigned long long, int) { noPhase(); }
    virtual void run(uint8_t*) { noPhase(); }
    virtual void runWitnessMsm(uint8_t*, bool = true) { noPhase(); }
    virtual void witnessMsmBegin() { noPhase(); }
    virtual void witnessMsmEnd(uint8_t*) { noPhase(); }
    virtual void runHMsm(uint8_t*) { noPhase(); }
    virtual void hpolyChain(int, void*) { noPhase(); }
    virtual void hpolyCombine(void*, void*, void*) { noPhase(); }
    virtual void hRange(unsigned long long*, unsigned long long*, unsigned long long*) const { noPhase(); }
    virtual void finish(const uint8_t*, std::string&, std::string&) { noPhase(); }
    virtual void proveResident(std::string&, std::string&) { noPhase(); }
    virtual void roundCommit(uint8_t*) { noPhase(); }
    virtual void roundFinish(const uint8_t*, uint8_t*) { noPhase(); }
    virtual void applyCommitment(const uint8_t*) { noPhase(); }
    virtual int kernelStats(int which, double* avgMs, unsigned long long* launches, unsigned long long* entries, int reset) {
        uint64_t l = 0, e = 0;
        int rc = ug_ctx_kernel_stats(ctx(), which, avgMs, &l, &e, reset);
        if (launches) *launches = l;
        if (entries) *entries = e;
        return rc;
    }
};

const uint8_t* checkedSection(const BinFile& f, uint32_t id, uint64_t needBytes) {
    if (f.sectionSize(id) < needBytes)
        throw std::range_error("Section " + std::to_string(id) + " is shorter than the header implies");
    return f.sectionData(id);
}

}  // namespace

// =================================================================================================================
class Groth16Prover : public ProverBase {
public:
    // What a prover is made from: the zkey header (section 2), the coefficient records and the five point sections --
    // whole sections of a zkey buffer, or (sliced) only this rank's slice of each, so that a rank of a many-GPU prover
    // never holds the whole zkey (37.6 GB at 2^26) in host memory.
    struct Sources {
        bool sliced = false;
        const uint8_t* coefs = nullptr; uint64_t nCoefs = 0; bool haveCoefs = true;     // !haveCoefs: this rank runs no NTT chain
        const uint8_t *pA = nullptr, *pB1 = nullptr, *pB2 = nullptr, *pC = nullptr, *pH = nullptr;
        const unsigned long long* sliceBytes = nullptr;       // sliced: the caller's byte counts of pA .. pH, checked against the ranges
    };
    struct Ranges { Range w, c, h; };
    // the slices of rank `rank` of `count` (witnessRange: chosen by the caller, else the even split); C follows the witness slice
    static Ranges shardRanges(uint64_t M, uint64_t nPublic, uint64_t N, int rank, int count, const Range* witnessRange,
                              const Range* hRange = nullptr) {
        if (count < 1 || rank < 0 || rank >= count) throw std::invalid_argument("invalid shard rank / count");
        if (M < nPublic + 1) throw std::invalid_argument("zkey header: nVars smaller than nPublic + 1");
        Ranges r;
        r.w = shardRange(M, rank, count);
        if (witnessRange) {
            if (witnessRange->lo > witnessRange->hi || witnessRange->hi > M) throw std::invalid_argument("witness range outside [0, nVars]");
            r.w = *witnessRange;
        }
        r.h = shardRange(N, rank, count);
        if (hRange) {
            if (hRange->lo > hRange->hi || hRange->hi > N) throw std::invalid_argument("h range outside [0, domainSize]");
            r.h = *hRange;
        }
        const uint64_t shift = nPublic + 1, nC = M - nPublic - 1;
        r.c.lo = r.w.lo > shift ? r.w.lo - shift : 0;
        r.c.hi = r.w.hi > shift ? r.w.hi - shift : 0;
        if (r.c.hi > nC) r.c.hi = nC;
        if (r.c.lo > r.c.hi) r.c.lo = r.c.hi;
        return r;
    }

    // witnessRange: the slice of the witness-indexed sets (A, B1, B2, C) this rank owns, when the caller balances the
    // ranks itself (ug_groth16_prover_create_sharded_range); nullptr = the even split
    // runsChain = false: this rank will never be asked for an H-polynomial chain (hpolyChain / run), so it keeps no coefficient
    // matrix, twiddles or NTT vectors (ranks 3 and up of a many-device prover)
    // layout (optional): this rank's part of a many-device layout -- overrides the witness range, gives the h range and the
    // bucket classes of the witness products
    Groth16Prover(const void* zkey, unsigned long long zkeySize, int device, int rank, int count, const Range* witnessRange = nullptr,
                  bool runsChain = true, const ShardLayout* layout = nullptr)
        : rank_(rank), count_(count) {
        if (layout) layout_ = *layout;
        haveLayout_ = layout != nullptr;
        if (count < 1 || rank < 0 || rank >= count) throw std::invalid_argument("invalid shard rank / count");
        BinFile f(zkey, zkeySize, "zkey", 1);
        hdr_ = loadZkeyHeader(f, false);
        if (!hdr_.rIsBn254) throw std::invalid_argument("zkey curve not supported");
        if (hdr_.nVars < hdr_.nPublic + 1) throw std::invalid_argument("zkey header: nVars smaller than nPublic + 1");
        const uint64_t M = hdr_.nVars, N = hdr_.domainSize, nC = M - hdr_.nPublic - 1;
        Sources src;
        src.coefs = checkedSection(f, 4, 4 + hdr_.nCoefs * 44) + 4;       // src/groth16.cpp:38
        src.nCoefs = hdr_.nCoefs;
        src.pA = checkedSection(f, 5, M * 64);
        src.pB1 = checkedSection(f, 6, M * 64);
        src.pB2 = checkedSection(f, 7, M * 128);
        src.pC = checkedSection(f, 8, nC * 64);
        src.pH = checkedSection(f, 9, N * 64);
        src.haveCoefs = runsChain;
        init(src, device, witnessRange);
    }
    // from the header section and this rank's slices (ug_groth16_prover_create_sharded_slices)
    Groth16Prover(const void* header, unsigned long long headerSize, const Sources& slices, int device, int rank, int count,
                  const Range* witnessRange, const ShardLayout* layout = nullptr)
        : rank_(rank), count_(count) {
        if (layout) layout_ = *layout;
        haveLayout_ = layout != nullptr;
        // a header-only container so that the one header parser serves both forms
        std::vector<uint8_t> mini;
        auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) mini.push_back((uint8_t)(v >> (8 * k))); };
        auto put64 = [&](uint64_t v) { for (int k = 0; k < 8; k++) mini.push_back((uint8_t)(v >> (8 * k))); };
        mini.insert(mini.end(), {'z', 'k', 'e', 'y'});
        put32(1); put32(3);
        put32(1); put64(4); put32(1);
        put32(2); put64(headerSize);
        mini.insert(mini.end(), static_cast<const uint8_t*>(header), static_cast<const uint8_t*>(header) + headerSize);
        put32(4); put64(0);
        BinFile f(mini.data(), mini.size(), "zkey", 1);
        hdr_ = loadZkeyHeader(f, false);
        if (!hdr_.rIsBn254) throw std::invalid_argument("zkey curve not supported");
        hdr_.nCoefs = slices.nCoefs;
        Sources src = slices;
        src.sliced = true;
        init(src, device, witnessRange);       // (copies the verification-key points out of `mini` before it goes away)
    }

private:
    void init(const Sources& src, int device, const Range* witnessRange) {
        const int rank = rank_, count = count_;
        // the prover keeps its own copy of the verification-key points it needs (the reference keeps pointers)
        vk_.assign(hdr_.alpha1, hdr_.alpha1 + 64 + 64 + 128 + 128 + 64 + 128);
        hdr_.alpha1 = vk_.data(); hdr_.beta1 = vk_.data() + 64; hdr_.beta2 = vk_.data() + 128; hdr_.gamma2 = vk_.data() + 256;
        hdr_.delta1 = vk_.data() + 384; hdr_.delta2 = vk_.data() + 448;

        const uint64_t M = hdr_.nVars, N = hdr_.domainSize;
        const uint8_t *coefs = src.coefs, *pA = src.pA, *pB1 = src.pB1, *pB2 = src.pB2, *pC = src.pC, *pH = src.pH;
        haveHpoly_ = src.haveCoefs;

        const Ranges rg = haveLayout_ ? shardRanges(M, hdr_.nPublic, N, rank, count, &layout_.w, &layout_.h)
                                      : shardRanges(M, hdr_.nPublic, N, rank, count, witnessRange);
        wr_ = rg.w;                              // witness scalars (and A/B1/B2 points) of this rank
        hr_ = rg.h;                              // h scalars (and H points) of this rank
        const uint64_t cLo = rg.c.lo, cHi = rg.c.hi;
        if (!src.sliced) {                       // whole sections: step to this rank's slice
            pA += wr_.lo * 64; pB1 += wr_.lo * 64; pB2 += wr_.lo * 128; pC += cLo * 64; pH += hr_.lo * 64;
        } else {                                 // slices: each must hold the points of this rank's range (a short buffer would be read past its end)
            const uint64_t need[5] = {(wr_.hi - wr_.lo) * 64, (wr_.hi - wr_.lo) * 64, (wr_.hi - wr_.lo) * 128, (cHi - cLo) * 64, (hr_.hi - hr_.lo) * 64};
            const uint8_t* ptr[5] = {pA, pB1, pB2, pC, pH};
            static const char* const what[5] = {"points_a", "points_b1", "points_b2", "points_c", "points_h"};
            for (int k = 0; k < 5; k++) {
                if (need[k] && !ptr[k]) throw std::invalid_argument(std::string("Null ") + what[k] + " slice");
                if (src.sliceBytes && src.sliceBytes[k] < need[k])
                    throw std::invalid_argument(std::string(what[k]) + " slice is shorter than this rank's range: " + std::to_string(src.sliceBytes[k]) +
                                                " bytes, needed " + std::to_string(need[k]));
            }
            if (src.haveCoefs && !src.coefs && src.nCoefs) throw std::invalid_argument("Null coefficient records");
        }

        ugCheck(ug_ctx_create(&d_.ctx, device));
        // COLD START: the prover groth16_prover_create makes does not wait for its window tables. Create returns
        // after the 0.25 s of uploads; a thread of the prover (tableBuilder) then builds the tables in pieces of ~10 ms of device
        // time, each under the prover's turn, and steps back whenever a caller wants the turn: proofs that arrive meanwhile use
        // the classic windows on table 0 (which the upload filled) and wait for one piece at most; when the last piece is done
        // the widths become the prover's (under the turn: a proof boundary). ULTRAGROTH_TABLES_BG=0 builds them inside create as
        // rounds 1-4 did; sharded ranks and phase callers always get that form (they hold no turn the builder could share).
        {
            const char* bg = getenv("ULTRAGROTH_TABLES_BG");
            bgTables_ = g_deferTables && count == 1 && !haveLayout_ && !src.sliced && !(bg && bg[0] == '0');
        }
        // the H branch (coefficient mat-vec, NTT chains, h schedule, H MSM) gets its own stream. ULTRAGROTH_H_PRIORITY = h | n | l
        // gives it a stream priority class (ug_ctx_create_priority). Measured on MI355X / ROCm 7.2 (tools/run_r3_order.sh, rank 0 of
        // an 8-way shard at 2^24): the class changes NOTHING -- a chain queued beside the witness products takes 17.4 ms with the
        // high class and 17.2 ms with the normal one (7.1 ms alone): workgroups of both streams take the slots that retiring
        // workgroups free in turn. So the default stays normal, and the ORDER of the calls decides who runs first
        // (MultiGroth16Prover::prove, bench.py).
        {
            const char* pe = getenv("ULTRAGROTH_H_PRIORITY");
            const int cls = pe ? (pe[0] == 'h' ? 1 : pe[0] == 'l' ? -1 : 0) : 0;
            ugCheck(ug_ctx_create_priority(&d_.ctx2, device, cls));
        }
        if (const char* mr = getenv("ULTRAGROTH_MAX_RANGE")) {
            uint64_t v = strtoull(mr, nullptr, 10);
            if (v >= 1 && v < MAX_RANGE) maxRange_ = v;
        }
        cLo_ = cLo; cHi_ = cHi;
        // Window tables are decided ahead, from the sizes alone, so that every set is created WITH its tables: the table
        // kernel of one section runs while the next section is uploaded (zkey ingest: the 0.25 s of copies disappear
        // behind the 2.4 s of table building at 2^24).
        const uint64_t nw = wr_.hi - wr_.lo, nh = hr_.hi - hr_.lo;
        const uint64_t otherBytes = nw * (64 * 2 + 128) + (cHi - cLo) * 64 + nh * 64 + (haveHpoly_ ? hdr_.nCoefs * 80 + N * 32 * 5 + N * 100 : 0) +
                                    M * 32 + N * 32;
        // (a section of nothing but points at infinity -- B1 and C of a "G1 MSM + NTT only" circuit, BASELINE configs[1] -- stays
        // out of the group: as a set of its own it is marked empty and its product costs nothing)
        const bool anyEmpty = ug_points_all_infinity(pA, nw, 64) || ug_points_all_infinity(pB1, nw, 64) || ug_points_all_infinity(pC, cHi - cLo, 64);
        groupG1_ = fusedGroups() && !anyEmpty;
        // SPARSE B (see DeviceProver, sparseBSupport)
        std::vector<uint32_t> bSupport;
        std::vector<uint8_t> b1c, b2c;
        // (a rank of a many-device prover does the same over ITS range of the signals -- pB1 / pB2 are its slices here --, except in a
        // bucket-class layout, whose classes belong to the witness schedule)
        sparseB_ = !(haveLayout_ && layout_.qLog) && groupG1_ && nw <= maxRange_ && sparseBSupport(pB1, pB2, nw, bSupport, b1c, b2c);
        nB_ = sparseB_ ? bSupport.size() : 0;
        if (sparseB_ && wr_.lo) for (uint32_t& i : bSupport) i += (uint32_t)wr_.lo;          // signal numbers of the whole witness
        std::vector<int> ahead = planTableWidthsAhead(d_.ctx, tableGroups(), otherBytes);
        ahead.resize(3, 0);
        bool withTables = true;
        if (bgTables_ && (ahead[0] || ahead[1])) {
            ugCheck(ug_ctx_defer_tables(d_.ctx, 1));
            ugCheck(ug_ctx_defer_tables(d_.ctx2, 1));
        } else bgTables_ = false;
        auto create = [&](ug_ctx* ctx, bool g2, const uint8_t* pts, uint64_t n, uint64_t first, int width, ug_bases** out) {
            if (width && withTables) {
                int rc = g2 ? ug_bases_create_tables_g2(ctx, pts, n, first, width, out) : ug_bases_create_tables_g1(ctx, pts, n, first, width, out);
                if (rc == UG_OK) return;
                withTables = false;                 // memory ran short after all: this set and the rest without tables
            }
            ugCheck(g2 ? ug_bases_create_g2(ctx, pts, n, first, out) : ug_bases_create_g1(ctx, pts, n, first, out));
        };
        traceStep("create: contexts made, point sets next");
        if (groupG1_ && sparseB_) {
            // [A | C]; B1 and B2 as compacted sets over the signals that have a real B point
            const void* hosts[2] = {pA, pC};
            const uint64_t counts[2] = {nw, cHi - cLo}, firsts[2] = {wr_.lo, cLo + hdr_.nPublic + 1};
            int rc = UG_ERROR;
            if (ahead[0]) rc = ug_bases_create_group_g1(d_.ctx, 2, hosts, counts, firsts, wr_.lo, nw, ahead[0], &d_.G);
            if (rc != UG_OK) {
                if (ahead[0]) withTables = false;
                ugCheck(ug_bases_create_group_g1(d_.ctx, 2, hosts, counts, firsts, wr_.lo, nw, 0, &d_.G));
            }
            create(d_.ctx, false, b1c.data(), nB_, 0, ahead[2], &d_.Bc1);
            create(d_.ctx, true, b2c.data(), nB_, 0, ahead[2], &d_.Bc2);
            ugCheck(ug_index_create(d_.ctx, bSupport.data(), nB_, &d_.bIdx));
            ugCheck(ug_dvec_create(d_.ctx, nB_, &d_.wB));
            ugCheck(ug_schedule_create(d_.ctx, &d_.sB));
            b1c = std::vector<uint8_t>(); b2c = std::vector<uint8_t>();
        } else if (groupG1_) {
            // A, B1 and C (with its index shift folded into the slot numbers) as one interleaved group
            const void* hosts[3] = {pA, pB1, pC};
            const uint64_t counts[3] = {nw, nw, cHi - cLo}, firsts[3] = {wr_.lo, wr_.lo, cLo + hdr_.nPublic + 1};
            int rc = UG_ERROR;
            if (ahead[0]) rc = ug_bases_create_group_g1(d_.ctx, 3, hosts, counts, firsts, wr_.lo, nw, ahead[0], &d_.G);
            if (rc != UG_OK) {
                if (ahead[0]) withTables = false;           // memory ran short after all
                ugCheck(ug_bases_create_group_g1(d_.ctx, 3, hosts, counts, firsts, wr_.lo, nw, 0, &d_.G));
            }
        } else {
            create(d_.ctx, false, pA, nw, wr_.lo, ahead[0], &d_.A);
            create(d_.ctx, false, pB1, nw, wr_.lo, ahead[0], &d_.B1);
        }
        traceStep("create: G1 sets of the witness uploaded");
        if (!sparseB_) create(d_.ctx, true, pB2, nw, wr_.lo, ahead[0], &d_.B2);
        traceStep("create: B2 uploaded");
        if (!d_.G) create(d_.ctx, false, pC, cHi - cLo, cLo, ahead[0], &d_.C);
        const bool group0 = withTables && ahead[0];
        create(d_.ctx2, false, pH, nh, hr_.lo, ahead[1], &d_.H);
        traceStep("create: H uploaded");
        const bool group1 = withTables && ahead[1];
        const bool group2 = sparseB_ && withTables && ahead[2];
        if (ahead[0] && !group0) {
            for (ug_bases* b : {d_.G, d_.A, d_.B1, d_.B2, d_.C}) if (b) ug_bases_drop_tables(b);
        }
        if (sparseB_ && ahead[2] && !group2) { ug_bases_drop_tables(d_.Bc1); ug_bases_drop_tables(d_.Bc2); }
        tableW_ = group0 ? ahead[0] : 0;
        tableH_ = group1 ? ahead[1] : 0;
        d_.tableB = group2 ? ahead[2] : 0;
        if (group0) tableBytes += (d_.G ? ug_bases_tables_bytes((sparseB_ ? 2 : 3) * nw, 0, tableW_) : ug_bases_tables_bytes(nw, 0, tableW_) * 2 + ug_bases_tables_bytes(cHi - cLo, 0, tableW_)) +
                                  (sparseB_ ? 0 : ug_bases_tables_bytes(nw, 1, tableW_));
        if (group1) tableBytes += ug_bases_tables_bytes(nh, 0, tableH_);
        if (group2) tableBytes += ug_bases_tables_bytes(nB_, 0, d_.tableB) + ug_bases_tables_bytes(nB_, 1, d_.tableB);
        if (bgTables_) {          // classic windows until tableBuilder() has finished them
            pendingW_.store(tableW_); pendingH_.store(tableH_); pendingB_.store(d_.tableB);
            tableW_ = tableH_ = 0; d_.tableB = 0;
        }
        if (haveHpoly_ && !d_.hp) ugCheck(ug_hpoly_create(d_.ctx2, coefs, hdr_.nCoefs, hdr_.domainSize, hdr_.nVars, &d_.hp));
        ugCheck(ug_dvec_create(d_.ctx, M, &d_.w));
        wCur_ = d_.w;
        witness_.attach(d_.ctx, M, &d_.w, &d_.w2);
        ugCheck(ug_dvec_create(d_.ctx2, N, &d_.h));
        ugCheck(ug_schedule_create(d_.ctx, &d_.sw));
        ugCheck(ug_schedule_create(d_.ctx2, &d_.sh));
        if (haveLayout_ && layout_.qLog) {
            // Bucket classes: this rank's witness schedule keeps its residues' entries only. They exist for the window-table form
            // (ONE bucket set per product, so that a class is a slice of it): without the tables every window would bring its own
            // sets of every owned residue, and the result blocks would not hold them.
            if (!tableW_) throw TablesUnavailable("a bucket-class layout needs the fixed-base window tables (device memory short, or ULTRAGROTH_TABLES=0)");
            if (layout_.sp.lo > layout_.sp.hi || layout_.sp.hi > M) throw std::invalid_argument("special-bucket range outside [0, nVars]");
            ugCheck(ug_schedule_set_classes(d_.sw, layout_.qLog, layout_.r0, layout_.cnt, SHARD_SPECIALS, layout_.sp.lo, layout_.sp.hi - layout_.sp.lo));
        }
        traceStep("create: vectors and schedules made");
        ugCheck(ug_ctx_sync(d_.ctx));                // table builds queued by the creates above end here: a finished prover (deferred
        ugCheck(ug_ctx_sync(d_.ctx2));               // builds have queued nothing: their thread starts now)
        if (bgTables_) builder_ = std::thread([this] { tableBuilder(); });
        traceStep("create: done");
    }

public:
    // ---- deferred window tables: built by this thread in pieces between the proofs (init) ----
    // A caller that wants the turn says so (TurnRequest) and the builder stays away from the mutex meanwhile: a plain mutex
    // would let the builder take turn after turn.
    struct TurnRequest {
        std::atomic<int>& n;
        explicit TurnRequest(std::atomic<int>& n_) : n(n_) { n.fetch_add(1); }
        ~TurnRequest() { n.fetch_sub(1); }
    };
    // ... and they alternate: under a steady stream of callers the builder still gets one piece between two of their turns (a
    // proof then waits ~10 ms; the tables arrive after ~190 proofs at 2^24 instead of never), and never two in a row while
    // somebody waits
    void callerHasTheTurn() { builderHadLast_.store(false); turnsTaken_.fetch_add(1); }
    void tableBuilder() {
        // pieces of about 10 ms: the G1 table kernel makes ~60 k points per ms, the G2 one ~20 k (2^24: 840 ms for 50 M points of
        // the A | B1 | C group, 836 ms for 16.7 M points of B2)
        const uint64_t pieceG1 = (uint64_t)1 << 19, pieceG2 = (uint64_t)3 << 16;
        try {
            // The tables' memory (72 GiB at 2^24) is not allocated by create either: the first allocation of tens of GiB on a device
            // can take a second (seen: 36 GiB in 1.09 s on a box fresh from boot, 0.3 ms afterwards) and holds up every other HIP
            // call of the process meanwhile. So the builder lets the FIRST proof through before it asks -- or starts after a quarter
            // of a second without any caller --, allocates the room of all sets in one go, and swaps each in under the turn.
            {
                const auto born = std::chrono::steady_clock::now();
                for (;;) {
                    if (stopBuilder_.load()) return;
                    const bool quiet = std::chrono::steady_clock::now() - born > std::chrono::milliseconds(250);
                    if ((turnsTaken_.load() > 0 || quiet) && wantTurn_.load() == 0 && proveMutex.try_lock()) { proveMutex.unlock(); break; }
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                }
                ug_bases* all[8] = {d_.G, d_.A, d_.B1, d_.C, d_.B2, d_.H, d_.Bc1, d_.Bc2};
                void* mem[8] = {nullptr};
                for (int k = 0; k < 8; k++) if (all[k]) ugCheck(ug_bases_tables_alloc(all[k], &mem[k]));
                for (int k = 0; k < 8; k++) {
                    if (!all[k]) continue;
                    while (wantTurn_.load() > 0 && builderHadLast_.load() && !stopBuilder_.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
                    std::lock_guard<std::mutex> turn(proveMutex);
                    ugCheck(ug_bases_tables_adopt(all[k], mem[k]));
                    mem[k] = nullptr;
                    dropGraphs();
                    builderHadLast_.store(true);
                }
                traceStep("room for the window tables allocated and swapped in");
            }
            for (int group = 0; group < 3; group++) {
                ug_bases* sets[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
                if (group == 0) { if (!pendingW_.load()) continue; sets[0] = d_.G; sets[1] = d_.A; sets[2] = d_.B1; sets[3] = d_.C; sets[4] = d_.B2; }
                else if (group == 1) { if (!pendingH_.load()) continue; sets[0] = d_.H; }
                else { if (!pendingB_.load()) continue; sets[0] = d_.Bc1; sets[4] = d_.Bc2; }
                for (ug_bases* b : sets) {
                    if (!b) continue;
                    for (;;) {
                        while (wantTurn_.load() > 0 && builderHadLast_.load() && !stopBuilder_.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
                        if (stopBuilder_.load()) return;
                        std::lock_guard<std::mutex> turn(proveMutex);
                        // one piece per turn while the prover is idle (a caller that arrives waits ~5 ms on average); four in the
                        // turn that follows a caller's proof or when callers queue -- under a steady load the tables are then in
                        // use after ~50 proofs (~9 s at 2^24, a proof up to 40 ms later meanwhile) instead of ~140 (one piece in
                        // the gap between two calls): a proof on the tables is 20 ms faster, so finishing the build pays
                        const int pieces = (wantTurn_.load() > 0 || !builderHadLast_.load()) ? 4 : 1;
                        uint64_t left = 1;
                        for (int i = 0; i < pieces && left; i++) ugCheck(ug_bases_tables_step(b, (b == d_.B2 || b == d_.Bc2) ? pieceG2 : pieceG1, &left));
                        builderHadLast_.store(true);
                        if (!left) break;
                    }
                }
                std::lock_guard<std::mutex> turn(proveMutex);             // a proof boundary: the next proof uses the tables
                if (group == 0) { tableW_ = pendingW_.load(); pendingW_.store(0); }
                else if (group == 1) { tableH_ = pendingH_.load(); pendingH_.store(0); }
                else { d_.tableB = pendingB_.load(); pendingB_.store(0); }
                dropGraphs();
                traceStep(group == 0 ? "window tables of the witness sets in use" : group == 1 ? "window tables of H in use" : "window tables of the B sets in use");
            }
        } catch (const std::exception& e) {
            builderError_ = e.what();                                     // the prover stays on the classic windows
        }
        builderDone_.store(true);
    }
    void stopTableBuilder() {
        stopBuilder_.store(true);
        if (builder_.joinable()) builder_.join();
    }
    bool tablesReady(bool wait) override {
        if (!bgTables_) return true;
        while (wait && !builderDone_.load() && !stopBuilder_.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        // (no turn taken: a query must not wait for the builder's pieces; the error string is complete once builderDone_ is set)
        if (wait && builderDone_.load() && !builderError_.empty()) throw std::runtime_error("window tables: " + builderError_);
        return !pendingW_.load() && !pendingH_.load() && !pendingB_.load();
    }
    std::vector<TableGroup> tableGroups() override {
        std::vector<TableGroup> groups(2);
        if (d_.G || (!d_.A && groupG1_)) { groups[0].g1 = {d_.G}; groups[0].n1 = {3 * (wr_.hi - wr_.lo)}; }      // (also before the sets exist)
        else { groups[0].g1 = {d_.A, d_.B1, d_.C}; groups[0].n1 = {wr_.hi - wr_.lo, wr_.hi - wr_.lo, cHi_ - cLo_}; }
        groups[0].g2 = {d_.B2}; groups[0].n2 = {wr_.hi - wr_.lo};
        groups[0].scalars = wr_.hi - wr_.lo; groups[0].c = &tableW_;
        if (wr_.hi - wr_.lo > maxRange_) groups[0].scalars = 0;          // proved in pieces: classic windows per piece
        groups[1].g1 = {d_.H}; groups[1].n1 = {hr_.hi - hr_.lo};
        groups[1].scalars = hr_.hi - hr_.lo; groups[1].c = &tableH_;
        if (hr_.hi - hr_.lo > maxRange_) groups[1].scalars = 0;
        if (sparseB_) {               // the group of the witness holds [A | C] only; B1 and B2 live compacted, with a schedule of their own
            groups[0].g1 = {d_.G}; groups[0].n1 = {2 * (wr_.hi - wr_.lo)};
            groups[0].g2.clear(); groups[0].n2.clear();
            groups.resize(3);
            groups[2].g1 = {d_.Bc1}; groups[2].n1 = {nB_};
            groups[2].g2 = {d_.Bc2}; groups[2].n2 = {nB_};
            groups[2].scalars = nB_; groups[2].c = &d_.tableB;
        }
        return groups;
    }
    void trimWorkspaces() override {
        TurnRequest mine(wantTurn_);
        std::lock_guard<std::mutex> turn(proveMutex);
        dropGraphs();                                   // (they hold pointers into what goes now)
        ug_schedule_trim(d_.sw); ug_schedule_trim(d_.sh);
        if (d_.sB) ug_schedule_trim(d_.sB);
        ug_ctx_trim(d_.ctx); ug_ctx_trim(d_.ctx2);
        witness_.trim(wCur_);
    }

    const ZkeyHeader& header() const { return hdr_; }
    bool sparseB() const { return sparseB_; }

    // the whole witness section of a .wtns buffer, checked against the circuit (src/prover.cpp:183-196)
    const uint8_t* witnessData(const BinFile& f) const {
        WtnsHeader wh = loadWtnsHeader(f);
        if (hdr_.nVars != wh.nVars)
            throw InvalidWitnessLengthException("Invalid witness length. Circuit: " + std::to_string(hdr_.nVars) +
                                                ", witness: " + std::to_string(wh.nVars));
        if (!wh.primeIsBn254) throw std::invalid_argument("different wtns curve");
        return checkedSection(f, 2, (uint64_t)hdr_.nVars * 32);
    }
    // phase call (the caller drives the phases of one proof from one thread and holds no lock): staged like a proof's
    void loadWitness(const void* wtns, unsigned long long wtnsSize) override {
        if (witnessQueued_ == 1) throw std::invalid_argument("the queued witness products still read the witness (ug_groth16_prover_witness_msm_end)");
        WitnessLease lease(witness_);
        stage(*lease, wtns, wtnsSize);
        std::unique_lock<std::mutex> turn;
        { TurnRequest mine(wantTurn_); turn = std::unique_lock<std::mutex>(proveMutex); }
        adopt(*lease);
    }
    // staging: parse the .wtns and copy the witness into the leased buffer; needs no turn on the device, only the
    // uploader (one copy at a time)
    void stage(StagedWitness& sl, const void* wtns, unsigned long long wtnsSize) {
        std::lock_guard<std::mutex> st(witness_.stageMutex);
        auto u0 = std::chrono::steady_clock::now();
        BinFile f(wtns, wtnsSize, "wtns", 2);
        const uint8_t* data = witnessData(f);
        sl.publicPart.assign(data, data + ((size_t)hdr_.nPublic + 1) * 32);
        ugCheck(ug_dvec_upload_idle(sl.buf, data, hdr_.nVars));
        sl.uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u0).count();
    }
    // the staged witness becomes the prover's (proveMutex held)
    void adopt(StagedWitness& sl) {
        wCur_ = sl.buf;
        publicPart_.swap(sl.publicPart);
        uploadMs = sl.uploadMs;
        resetTimings();
        witnessLoaded_ = true; witnessComplete_ = true;
    }
    // ULTRAGROTH_TRACE=1: when each step of a call happens, in ms of a process-wide clock, on stderr (two callers' lines
    // interleave: how long the device waits between one proof's last kernel and the next one's first)
    static void traceStep(const char* what) {
        static const bool on = getenv("ULTRAGROTH_TRACE") && atoi(getenv("ULTRAGROTH_TRACE")) != 0;
        if (!on) return;
        static const auto origin = std::chrono::steady_clock::now();
        fprintf(stderr, "[groth16 %5.5zx] %9.3f ms  %s\n", std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xfffff,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - origin).count(), what);
    }
    void proveTurn(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub, std::mutex* device,
                   const Around& around) override {
        auto t0 = std::chrono::steady_clock::now();
        traceStep("call");
        WitnessLease lease(witness_);
        stage(*lease, wtns, wtnsSize);
        traceStep("witness staged");
        std::unique_lock<std::mutex> card;
        if (device) card = std::unique_lock<std::mutex>(*device);
        std::unique_lock<std::mutex> turn;
        { TurnRequest mine(wantTurn_); turn = std::unique_lock<std::mutex>(proveMutex); }
        callerHasTheTurn();
        traceStep("turn on the device");
        AroundGuard bracket(around);
        bracket.begin();
        adopt(*lease);
        // the turn ends with the device part: blinding and JSON of this proof run on the host while the next caller's
        // kernels start (0.7 ms of idle device per proof otherwise)
        proveLoaded(proof, pub, [&] {
            totalMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            bracket.end();
            turn.unlock();
            if (card.owns_lock()) card.unlock();
        });
        traceStep("proof finished");
    }
    // The witness in two parts, for a rank of a sharded prover: part 0 = the scalars of this rank's MSM slice (what
    // runWitnessMsm reads), part 1 = everything else (only the H-polynomial mat-vec reads it; a rank that runs no
    // chain never calls it). Part 1 may be uploaded from a second host thread while the MSMs of part 0 run.
    void loadWitnessPart(const void* wtns, unsigned long long wtnsSize, int part) override {
        auto t0 = std::chrono::steady_clock::now();
        BinFile f(wtns, wtnsSize, "wtns", 2);
        WtnsHeader wh = loadWtnsHeader(f);
        if (hdr_.nVars != wh.nVars)
            throw InvalidWitnessLengthException("Invalid witness length. Circuit: " + std::to_string(hdr_.nVars) +
                                                ", witness: " + std::to_string(wh.nVars));
        if (!wh.primeIsBn254) throw std::invalid_argument("different wtns curve");
        const uint8_t* data = checkedSection(f, 2, (uint64_t)hdr_.nVars * 32);
        if (part == 0 && witnessQueued_ == 1)      // (part 1 lies outside the range the queued products read, and goes to the other stream)
            throw std::invalid_argument("the queued witness products still read the witness (ug_groth16_prover_witness_msm_end)");
        if (part == 0) {
            resetTimings();
            publicPart_.assign(data, data + ((size_t)hdr_.nPublic + 1) * 32);
            ugCheck(ug_dvec_upload_range(wCur_, data + wr_.lo * 32, wr_.lo, wr_.hi - wr_.lo, 0));
            witnessLoaded_ = true; witnessComplete_ = false;
            uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        } else if (part == 1) {
            // on the H branch's context: its stream orders the copy before the mat-vec, and the MSM stream is left alone
            if (wr_.lo) ugCheck(ug_dvec_upload_range(wCur_, data, 0, wr_.lo, d_.ctx2));
            if (wr_.hi < hdr_.nVars) ugCheck(ug_dvec_upload_range(wCur_, data + wr_.hi * 32, wr_.hi, hdr_.nVars - wr_.hi, d_.ctx2));
            witnessComplete_ = true;
        } else throw std::invalid_argument("witness part must be 0 or 1");
    }

    // S1-S4 on this rank's slice; partials = A | B1 | B2 | C | H affine records (H left at infinity)
    void runWitnessMsm(uint8_t* partials, bool standalone = true) override {
        if (!witnessLoaded_) throw std::invalid_argument("no witness loaded");
        if (witnessQueued_ == 1) throw std::invalid_argument("witness products are queued on this prover (ug_groth16_prover_witness_msm_end)");
        // A schedule holds at most 2^31 (scalar, window) entries; ranges above MAX_RANGE scalars (only the
        // reference's largest legal domain, 2^27, needs it) are proved in pieces whose partial sums are added.
        memset(partials, 0, UG_GROTH16_PARTIALS_SIZE);
        uint8_t part[UG_GROTH16_PARTIALS_SIZE];
        QueueGuard inFlight(d_.ctx);
        for (uint64_t lo = wr_.lo; lo < wr_.hi; lo += maxRange_) {
            uint64_t n = std::min<uint64_t>(maxRange_, wr_.hi - lo);
            uint8_t* out = (lo == wr_.lo) ? partials : part;
            memset(part, 0, sizeof part);
            buildSchedule(d_.sw, wCur_, lo, n, tableW_);
            // S1-S4 (src/groth16.cpp:55,58,61,64): A, B1, B2, C over the witness schedule, queued back to back
            enqueueWitnessProducts(d_, d_.ctx, d_.sw, out, out + 64, out + 128, out + 256, (int64_t)hdr_.nPublic + 1, false, wCur_);
            ugCheck(ug_ctx_collect(d_.ctx));
            if (out == part && ug_groth16_partials_add(partials, part) != PROVER_OK) throw std::runtime_error("partial sum failed");
        }
        inFlight.done();
        if (standalone) collectTimings(1);
    }
    // The same in two calls (ug_groth16_prover_witness_msm_begin / _end): begin queues S1-S4 on the witness stream and returns
    // without a host wait, so that the caller drives this rank's H branch -- its chains, the exchange of the evaluation slices
    // with the other ranks, hpolyCombine, runHMsm: all on the second stream -- while the products run; end waits for them.
    // Rank 0 also draws r and s here (S11, still r then s, before the device work as in proveLoaded) and forms the multiples
    // that need only them on host threads beside the device; finish() then takes those.
    void witnessMsmBegin() override {
        if (!witnessLoaded_) throw std::invalid_argument("no witness loaded");
        if (witnessQueued_) throw std::invalid_argument("the witness products are already queued (ug_groth16_prover_witness_msm_end)");
        memset(queuedParts_, 0, sizeof queuedParts_);
        if (rank_ == 0) {
            if (earlyTerms_.valid()) earlyTerms_.wait();       // (a proof that was begun and never finished: its draw is dropped)
            drawBlinding(earlyR_); drawBlinding(earlyS_);
            earlyTerms_ = std::async(std::launch::async, [this] { return blindingTerms(hdr_, earlyR_, earlyS_); });
        }
        const uint64_t n = wr_.hi - wr_.lo;
        if (n == 0 || n > maxRange_) {             // nothing to queue / proved in pieces (partial sums added between them): done here
            if (n) runWitnessMsm(queuedParts_, true);
            witnessQueued_ = 2;
            return;
        }
        QueueGuard inFlight(d_.ctx);
        buildSchedule(d_.sw, wCur_, wr_.lo, n, tableW_);
        enqueueWitnessProducts(d_, d_.ctx, d_.sw, queuedParts_, queuedParts_ + 64, queuedParts_ + 128, queuedParts_ + 256,
                               (int64_t)hdr_.nPublic + 1, false, wCur_);
        inFlight.done();
        witnessQueued_ = 1;
    }
    void witnessMsmEnd(uint8_t* partials) override {
        if (!witnessQueued_) throw std::invalid_argument("no witness products queued (ug_groth16_prover_witness_msm_begin)");
        const int how = witnessQueued_;
        witnessQueued_ = 0;
        if (how == 1) {
            QueueGuard inFlight(d_.ctx);
            ugCheck(ug_ctx_collect(d_.ctx));
            inFlight.done();
            collectTimings(1);
        }
        memcpy(partials, queuedParts_, UG_GROTH16_PARTIALS_SIZE);
    }
    // a begun proof is given up: wait for the device, drop the queued products (no result is written)
    void witnessMsmAbandon() {
        if (witnessQueued_ == 1) ug_ctx_abandon(d_.ctx);
        witnessQueued_ = 0;
        if (earlyTerms_.valid()) { earlyTerms_.wait(); earlyTerms_ = std::future<BlindingTerms>(); }     // the abandoned proof's r, s are dropped with it
    }
    // S10 on this rank's slice of h (which must be in d_.h); only the H record of partials is written
    void runHMsm(uint8_t* partials) override { runHMsmImpl(partials, true); }
    void runHMsmImpl(uint8_t* partials, bool standalone) {
        memset(partials, 0, UG_GROTH16_PARTIALS_SIZE);
        uint8_t part[UG_GROTH16_PARTIALS_SIZE];
        QueueGuard inFlight(d_.ctx2);
        for (uint64_t lo = hr_.lo; lo < hr_.hi; lo += maxRange_) {
            uint64_t n = std::min<uint64_t>(maxRange_, hr_.hi - lo);
            uint8_t* out = (lo == hr_.lo) ? partials : part;
            memset(part, 0, sizeof part);
            buildSchedule(d_.sh, d_.h, lo, n, tableH_);
            const ug_bases* sets[1] = {d_.H};
            void* outs[1] = {out + 320};
            ugCheck(ug_msm_batch(d_.ctx2, 1, sets, d_.sh, nullptr, outs));                     // S10 :154
            if (out == part && ug_groth16_partials_add(partials, part) != PROVER_OK) throw std::runtime_error("partial sum failed");
        }
        inFlight.done();
        if (standalone) collectTimings(2);
    }
    // device time per branch since the witness was loaded; each branch's figures are touched only by the host thread that
    // drives that branch (a sharded rank runs its chains from a second thread, see include/prover.h)
    void collectTimings(int which) {
        if (which & 1) ugCheck(ug_ctx_timings(d_.ctx, &m1_, &f1_, 0));
        if (which & 2) ugCheck(ug_ctx_timings(d_.ctx2, &m2_, &f2_, 0));
    }
    void resetTimings() {
        ugCheck(ug_ctx_timings(d_.ctx, nullptr, nullptr, 1)); ugCheck(ug_ctx_timings(d_.ctx2, nullptr, nullptr, 1));
        m1_ = f1_ = m2_ = f2_ = 0;
    }
    void hpolyChain(int which, void* deviceOut) override {
        if (!witnessLoaded_ || !witnessComplete_) throw std::invalid_argument("the whole witness has not been loaded");
        if (!haveHpoly_) throw std::invalid_argument("this rank was created without the coefficient matrix");
        ug_dvec* v = nullptr;
        ugCheck(ug_dvec_wrap(d_.ctx2, deviceOut, hdr_.domainSize, &v));
        int rc = ug_hpoly_chain(d_.hp, wCur_, which, v);
        ug_dvec_destroy(v);
        ugCheck(rc);
        collectTimings(2);
    }
    void hpolyCombine(void* da, void* db, void* dc) override {
        uint64_t cnt = hr_.hi - hr_.lo;
        if (!cnt) return;                            // (a chain rank of a bucket-class layout takes no part in the H product)
        ug_dvec *a = nullptr, *b = nullptr, *c = nullptr;
        ugCheck(ug_dvec_wrap(d_.ctx2, da, cnt, &a));
        ugCheck(ug_dvec_wrap(d_.ctx2, db, cnt, &b));
        ugCheck(ug_dvec_wrap(d_.ctx2, dc, cnt, &c));
        int rc = ug_hpoly_combine(d_.hp, a, b, c, hr_.lo, cnt, d_.h);
        ug_dvec_destroy(a); ug_dvec_destroy(b); ug_dvec_destroy(c);
        ugCheck(rc);
        collectTimings(2);
    }
    void hRange(unsigned long long* first, unsigned long long* count, unsigned long long* domain) const override {
        if (first) *first = hr_.lo;
        if (count) *count = hr_.hi - hr_.lo;
        if (domain) *domain = hdr_.domainSize;
    }

    // S1-S10 on this rank's slices with the H-polynomial block computed locally (replicated when sharded): everything is
    // queued on the two streams -- S1-S4 on one, S5-S10 (H polynomial, its schedule, the H MSM) on the other -- and the host
    // waits ONCE, at the end, before it finishes the five results. By default the second stream is ordered behind the first
    // on the device (ug_ctx_wait), so kernels run one after the other and their durations are clean; ULTRAGROTH_OVERLAP=1
    // drops that edge: the memory-bound kernels of one branch then overlap the integer-bound kernels of the other
    // (measured in round 1: 148 -> 144 ms per 2^24 proof, 49.9 -> 46.7 ms at 2^22) at the price of blurred per-kernel times.
    void run(uint8_t* partials) override {
        if (!witnessLoaded_ || !witnessComplete_) throw std::invalid_argument("no witness loaded");
        if (!haveHpoly_) throw std::invalid_argument("this rank was created without the coefficient matrix");
        if (witnessQueued_) throw std::invalid_argument("witness products are queued on this prover (ug_groth16_prover_witness_msm_end)");
        const uint64_t nw = wr_.hi - wr_.lo, nh = hr_.hi - hr_.lo;
        if (nw > maxRange_ || nh > maxRange_) {                 // proved in pieces: partial sums are added between them
            uint8_t hpart[UG_GROTH16_PARTIALS_SIZE];
            runWitnessMsm(partials, /*standalone*/ false);
            QueueGuard hBranch(d_.ctx2);
            ugCheck(ug_hpoly_run(d_.hp, wCur_, d_.h));                                          // S5-S9 :66-148
            runHMsmImpl(hpart, false);                                                         // S10   :154
            hBranch.done();
            memcpy(partials + 320, hpart + 320, 64);
            collectTimings(3);
            return;
        }
        // ULTRAGROTH_OVERLAP: 1 (default since round 5: -2 % at 2^24, -9 % at 2^20) no edge between the streams: the H branch runs
        // beside the witness products; 0 the second stream waits for the first (one kernel on the chip at a time: what a caller
        // that wants clean per-kernel times sets, bench.py for its roofline steps); 2 the G2 product goes first and the H
        // branch's mat-vec, NTT passes and schedule run beside it, its MSM after the witness branch
        const char* ov = getenv("ULTRAGROTH_OVERLAP");
        const int overlap = ov ? atoi(ov) : 1;
        memset(partials, 0, UG_GROTH16_PARTIALS_SIZE);
        memset(runParts_, 0, sizeof runParts_);
        QueueGuard inFlight(d_.ctx, d_.ctx2);                   // from here to the collects below work is queued on both streams
        // the whole device part: queued eagerly, or -- ULTRAGROTH_GRAPH=1 -- recorded once per witness buffer and replayed (the
        // queued products write to runParts_, a member, so that a replay finds the same addresses)
        auto queueAll = [&] {
            uint8_t* out = runParts_;
            buildSchedule(d_.sw, wCur_, wr_.lo, nw, tableW_);
            // S1-S4 (src/groth16.cpp:55,58,61,64): A, B1, B2, C over the witness schedule, queued back to back
            enqueueWitnessProducts(d_, d_.ctx, d_.sw, out, out + 64, out + 128, out + 256, (int64_t)hdr_.nPublic + 1, overlap == 2, wCur_);
            if (overlap == 0) ugCheck(ug_ctx_wait(d_.ctx2, d_.ctx));
            ugCheck(ug_hpoly_run(d_.hp, wCur_, d_.h));                                          // S5-S9 :66-148
            buildSchedule(d_.sh, d_.h, hr_.lo, nh, tableH_);
            if (overlap == 2) ugCheck(ug_ctx_wait(d_.ctx2, d_.ctx));
            const ug_bases* sets[1] = {d_.H};
            void* outs[1] = {out + 320};
            ugCheck(ug_msm_batch_enqueue(d_.ctx2, 1, sets, d_.sh, nullptr, outs));             // S10 :154
        };
        ug_graph* g = graphsWanted() ? graphFor(wCur_, overlap) : nullptr;
        if (g) {
            ugCheck(ug_graph_launch(g));
        } else if (graphsWanted() && graphWarm(wCur_, overlap)) {
            // second proof on this witness buffer with this geometry: every workspace has its size, record the sequence
            ugCheck(ug_graph_begin(d_.ctx, d_.ctx2));
            try { queueAll(); ugCheck(ug_graph_end(d_.ctx, &g)); }
            catch (...) { ug_graph_abort(d_.ctx); throw; }
            graphs_.push_back(GraphSlot{wCur_, overlap, g});
            ugCheck(ug_graph_launch(g));
        } else {
            queueAll();
        }
        traceStep("device part fully queued");
        if (g) {                                                // a graph runs on the first context's stream: that one is waited for
            ugCheck(ug_ctx_collect(d_.ctx));
            ugCheck(ug_ctx_collect(d_.ctx2));
        } else {
            ugCheck(ug_ctx_collect(d_.ctx2));                   // the one host wait of the device part ...
            ugCheck(ug_ctx_collect(d_.ctx));                    // (... this one returns at once unless the streams overlap)
        }
        inFlight.done();
        memcpy(partials, runParts_, UG_GROTH16_PARTIALS_SIZE);
        collectTimings(3);
    }
    // ---- one hipGraph per (witness buffer, overlap mode) of a created prover: ULTRAGROTH_GRAPH=1 ----
    struct GraphSlot { const ug_dvec* w; int overlap; ug_graph* g; };
    static bool graphsWanted() { const char* e = getenv("ULTRAGROTH_GRAPH"); return e && atoi(e) != 0; }      // (read per proof)
    // a graph that is still valid for this witness buffer; stale ones (a workspace was re-allocated since) are dropped
    ug_graph* graphFor(const ug_dvec* w, int overlap) {
        for (size_t i = 0; i < graphs_.size();) {
            if (!ug_graph_valid(graphs_[i].g)) { ug_graph_destroy(graphs_[i].g); graphs_.erase(graphs_.begin() + (long)i); warm_.clear(); continue; }
            if (graphs_[i].w == w && graphs_[i].overlap == overlap) return graphs_[i].g;
            i++;
        }
        return nullptr;
    }
    // true from the second proof on (w, overlap) on: the first one ran eagerly and sized every buffer
    bool graphWarm(const ug_dvec* w, int overlap) {
        for (auto& k : warm_) if (k.first == w && k.second == overlap) return true;
        warm_.push_back({w, overlap});
        return false;
    }
    void dropGraphs(bool keepWarm = false) {
        for (auto& s : graphs_) ug_graph_destroy(s.g);
        graphs_.clear();
        if (!keepWarm) warm_.clear();
    }
    int kernelStats(int which, double* avgMs, unsigned long long* launches, unsigned long long* entries, int reset) override {
        double a1 = 0, a2 = 0;
        uint64_t l1 = 0, l2 = 0, e1 = 0, e2 = 0;
        if (reset && !statsOn_) {                   // the statistics start here: sequences recorded without their event pairs are recorded again
            TurnRequest mine(wantTurn_);
            std::lock_guard<std::mutex> turn(proveMutex);
            dropGraphs(/*keepWarm*/ true);
            statsOn_ = true;
        }
        if (ug_ctx_kernel_stats(d_.ctx, which, &a1, &l1, &e1, reset) != UG_OK) return UG_ERROR;
        if (ug_ctx_kernel_stats(d_.ctx2, which, &a2, &l2, &e2, reset) != UG_OK) return UG_ERROR;
        if (avgMs) *avgMs = (l1 + l2) ? (a1 * (double)l1 + a2 * (double)l2) / (double)(l1 + l2) : 0.0;
        if (launches) *launches = l1 + l2;
        if (entries) *entries = e1 + e2;
        return UG_OK;
    }

    void finish(const uint8_t* sums, std::string& proof, std::string& pub) override {
        if (earlyTerms_.valid()) {                 // r, s and their multiples were made beside the device work (witnessMsmBegin)
            const BlindingTerms terms = earlyTerms_.get();
            finishWith(sums, earlyR_, earlyS_, terms, proof, pub);
            return;
        }
        uint8_t r[32], s[32];
        drawBlinding(r); drawBlinding(s);                                                      // S11 :158-166
        finishWith(sums, r, s, blindingTerms(hdr_, r, s), proof, pub);
    }
    void finishWith(const uint8_t* sums, const uint8_t r[32], const uint8_t s[32], const BlindingTerms& terms, std::string& proof,
                    std::string& pub, const std::vector<uint8_t>* publicPart = nullptr) {
        uint8_t A[64], B[128], C[64];
        blind(A, B, C, sums, sums + 64, sums + 128, sums + 256, sums + 320, hdr_, r, s, terms, nullptr);
        // nlohmann dump(): keys in lexicographic order, no whitespace (src/groth16.cpp:217-250)
        proof = "{\"pi_a\":" + g1Json(A) + ",\"pi_b\":" + g2Json(B) + ",\"pi_c\":" + g1Json(C) + ",\"protocol\":\"groth16\"}";
        pub = publicJson((publicPart ? *publicPart : publicPart_).data(), hdr_.nPublic, 0);
    }

    // (every caller comes through proveTurn, which this class overrides; kept for the interface)
    void prove(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub) override {
        proveTurn(wtns, wtnsSize, proof, pub, nullptr, Around());
    }
    // one whole proof (S1-S13) of the witness that is already resident (loadWitness): what prove does after its staging --
    // r and s drawn first, the multiples that need only them formed on host threads beside the device work
    void proveResident(std::string& proof, std::string& pub) override {
        std::unique_lock<std::mutex> turn;
        { TurnRequest mine(wantTurn_); turn = std::unique_lock<std::mutex>(proveMutex); }
        callerHasTheTurn();
        if (!witnessLoaded_ || !witnessComplete_) throw std::invalid_argument("no witness loaded");
        proveLoaded(proof, pub);
    }
    // deviceDone (optional) is called once the device part has left the device: nothing after it touches the prover's
    // per-proof state (the public signals are copied first), so the caller may give up its turn there
    void proveLoaded(std::string& proof, std::string& pub, const std::function<void()>& deviceDone = std::function<void()>()) {
        // S11 (:158-166) drawn up front, in the reference's order: the multiples of delta that need only r and s are
        // formed on host threads while the device runs S1-S10
        uint8_t r[32], s[32];
        drawBlinding(r); drawBlinding(s);
        auto terms = std::async(std::launch::async, [&] { return blindingTerms(hdr_, r, s); });
        uint8_t partials[UG_GROTH16_PARTIALS_SIZE];
        traceStep("device part queued from here");
        run(partials);                                   // (the future joins in its destructor if this throws)
        traceStep("device part done");
        const std::vector<uint8_t> publicPart = publicPart_;
        if (deviceDone) deviceDone();
        finishWith(partials, r, s, terms.get(), proof, pub, &publicPart);
    }

    unsigned long long proofBufferMinSize() const override { return PROOF_MIN_GROTH16; }
    unsigned long long publicBufferMinSize() const override { return publicMin(hdr_.nPublic); }
    void timings(double* msm, double* fft, double* total) const override {
        if (msm) *msm = m1_ + m2_;
        if (fft) *fft = f1_ + f2_;
        if (total) *total = totalMs_;
    }
    ug_ctx* ctx() override { return d_.ctx; }
    ug_ctx* ctx2() { return d_.ctx2; }              // the H branch's context (its stream orders hpolyChain / hpolyCombine / runHMsm)
    // a many-device prover collects a chain rank's witness from its peers: the vector the device part reads, this rank's range of
    // it, and the word that the rest has arrived (on the H branch's stream, as loadWitnessPart(.., 1) leaves it)
    ug_dvec* witnessVec() { return wCur_; }
    Range witnessRange() const { return wr_; }
    void witnessGathered() {
        if (!witnessLoaded_) throw std::invalid_argument("the rank's own part of the witness comes first");
        witnessComplete_ = true;
    }

private:
    static constexpr uint64_t MAX_RANGE = (uint64_t)1 << 26;       // 2^26 scalars * <= 16 windows < 2^31 entries
    uint64_t maxRange_ = MAX_RANGE;    // ULTRAGROTH_MAX_RANGE lowers it (tests: the piecewise path without a 2^27 circuit)
    int tableW_ = 0, tableH_ = 0;      // window widths of the fixed-base tables (0: classic windows), planWindowTables
    std::atomic<int> pendingW_{0}, pendingH_{0}, pendingB_{0};      // ... of tables that are still being built (tableBuilder)
    bool sparseB_ = false;             // B1 / B2 kept compacted over the signals with a real B point (DeviceProver)
    uint64_t nB_ = 0;
    bool bgTables_ = false;
    std::thread builder_;
    std::atomic<bool> stopBuilder_{false}, builderDone_{false};
    std::atomic<int> wantTurn_{0};
    std::atomic<uint64_t> turnsTaken_{0};
    std::atomic<bool> builderHadLast_{true};       // (the first turns, before any caller: one piece each)
    std::string builderError_;
    int rank_, count_;
    ShardLayout layout_;               // this rank's part of a many-device layout (haveLayout_), else unused
    bool haveLayout_ = false;
    ZkeyHeader hdr_;
    std::vector<uint8_t> vk_, publicPart_;
    Range wr_{0, 0}, hr_{0, 0};
    uint64_t cLo_ = 0, cHi_ = 0;       // this rank's slice of the C section
    DeviceProver d_;
    bool witnessLoaded_ = false, witnessComplete_ = false, haveHpoly_ = true;
    bool groupG1_ = true;              // A | B1 | C kept as one interleaved group (not when a member is all infinity, or ULTRAGROTH_FUSED=0)
    ug_dvec* wCur_ = nullptr;          // the witness the device part reads: one of the two buffers (d_.w, d_.w2 own them)
    WitnessBuffers witness_;
    double m1_ = 0, f1_ = 0, m2_ = 0, f2_ = 0, totalMs_ = 0;      // device ms of the MSM / FFT parts per stream
    // witnessMsmBegin .. witnessMsmEnd: the queued products write here (the queue holds these addresses until it is collected)
    int witnessQueued_ = 0;            // 1: queued on the witness stream, 2: already complete in queuedParts_
    uint8_t queuedParts_[UG_GROTH16_PARTIALS_SIZE] = {};
    uint8_t earlyR_[32] = {}, earlyS_[32] = {};
    std::future<BlindingTerms> earlyTerms_;
    uint8_t runParts_[UG_GROTH16_PARTIALS_SIZE] = {};       // where run()'s queued products write (fixed: a replayed graph's too)
    std::vector<GraphSlot> graphs_;
    std::vector<std::pair<const ug_dvec*, int>> warm_;
    bool statsOn_ = false;
public:
    ~Groth16Prover() override { stopTableBuilder(); dropGraphs(); }      // (before d_ goes: both refer to its contexts)
};

// =================================================================================================================
class UltraGrothProver : public ProverBase {
public:
    // What a prover is made from: whole sections of a zkey buffer, or (sliced) only this rank's slice of each point section
    // and of the two index lists, so that a rank of a many-GPU prover never holds the whole zkey in host memory
    struct Sources {
        bool sliced = false;
        const uint8_t* coefs = nullptr; uint64_t nCoefs = 0; bool haveCoefs = true;     // !haveCoefs: this rank runs no NTT chain
        const uint8_t *pA = nullptr, *pB1 = nullptr, *pB2 = nullptr, *pRoundC = nullptr, *pFinalC = nullptr, *pH = nullptr;
        const uint8_t *idx1 = nullptr, *idx2 = nullptr;
        const unsigned long long* sliceBytes = nullptr;       // sliced: byte counts of pA, pB1, pB2, pRoundC, pFinalC, pH, idx1, idx2
    };
    struct Ranges { Range w, h, c1, c2; };
    static Ranges shardRanges(uint64_t M, uint64_t N, uint64_t nC1, uint64_t nC2, int rank, int count) {
        if (count < 1 || rank < 0 || rank >= count) throw std::invalid_argument("invalid shard rank / count");
        return Ranges{shardRange(M, rank, count), shardRange(N, rank, count), shardRange(nC1, rank, count), shardRange(nC2, rank, count)};
    }
    // rank `rank` of `count`: the witness-indexed sets (A, B1, B2), the round set (C1 with round_indexes), the final set
    // (C2 with final_round_indexes) and H are each cut into `count` contiguous slices; every rank keeps the whole witness
    UltraGrothProver(const void* zkey, unsigned long long zkeySize, int device, int rank = 0, int count = 1, bool runsChain = true) {
        if (count < 1 || rank < 0 || rank >= count) throw std::invalid_argument("invalid shard rank / count");
        BinFile f(zkey, zkeySize, "zkey", 1);
        hdr_ = loadZkeyHeader(f, true);
        if (!hdr_.rIsBn254) throw std::invalid_argument("zkey curve not supported");
        const uint64_t M = hdr_.nVars, N = hdr_.domainSize;
        Sources src;
        src.coefs = checkedSection(f, 4, 4 + hdr_.nCoefs * 44) + 4;
        src.nCoefs = hdr_.nCoefs;
        src.haveCoefs = runsChain;
        // section map of protocol 1337 (src/prover.cpp:242-259)
        src.pA = checkedSection(f, 5, M * 64);
        src.pB1 = checkedSection(f, 6, M * 64);
        src.pB2 = checkedSection(f, 7, M * 128);
        src.pRoundC = checkedSection(f, 8, (uint64_t)hdr_.numIndexesC1 * 64);
        src.pFinalC = checkedSection(f, 9, (uint64_t)hdr_.numIndexesC2 * 64);
        src.idx1 = checkedSection(f, 10, (uint64_t)hdr_.numIndexesC1 * 4);
        src.idx2 = checkedSection(f, 11, (uint64_t)hdr_.numIndexesC2 * 4);
        src.pH = checkedSection(f, 12, N * 64);
        init(src, device, rank, count);
    }
    // from the header section and this rank's slices (ug_ultra_groth_prover_create_sharded_slices)
    UltraGrothProver(const void* header, unsigned long long headerSize, const Sources& slices, int device, int rank, int count) {
        if (count < 1 || rank < 0 || rank >= count) throw std::invalid_argument("invalid shard rank / count");
        std::vector<uint8_t> mini;             // a header-only container, so that the one header parser serves both forms
        auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) mini.push_back((uint8_t)(v >> (8 * k))); };
        auto put64 = [&](uint64_t v) { for (int k = 0; k < 8; k++) mini.push_back((uint8_t)(v >> (8 * k))); };
        mini.insert(mini.end(), {'z', 'k', 'e', 'y'});
        put32(1); put32(3);
        put32(1); put64(4); put32(1337);
        put32(2); put64(headerSize);
        mini.insert(mini.end(), static_cast<const uint8_t*>(header), static_cast<const uint8_t*>(header) + headerSize);
        put32(4); put64(0);
        BinFile f(mini.data(), mini.size(), "zkey", 1);
        hdr_ = loadZkeyHeader(f, true);
        hdr_.nCoefs = slices.nCoefs;
        Sources src = slices;
        src.sliced = true;
        init(src, device, rank, count);        // (copies the verification-key points out of `mini` before it goes away)
    }

private:
    void init(const Sources& src, int device, int rank, int count) {
        if (!hdr_.rIsBn254) throw std::invalid_argument("zkey curve not supported");
        vk_.assign(hdr_.alpha1, hdr_.alpha1 + 64 + 64 + 128 + 128 + 64 + 128 + 64 + 128);
        uint8_t* v = vk_.data();
        hdr_.alpha1 = v; hdr_.beta1 = v + 64; hdr_.beta2 = v + 128; hdr_.gamma2 = v + 256;
        hdr_.roundDelta1 = v + 384; hdr_.roundDelta2 = v + 448; hdr_.delta1 = v + 576; hdr_.delta2 = v + 640;

        const uint64_t M = hdr_.nVars, N = hdr_.domainSize;
        const Ranges rg = shardRanges(M, N, hdr_.numIndexesC1, hdr_.numIndexesC2, rank, count);
        wr_ = rg.w; hr_ = rg.h;
        const Range c1 = rg.c1, c2 = rg.c2;
        const uint8_t *coefs = src.coefs, *pA = src.pA, *pB1 = src.pB1, *pB2 = src.pB2, *pRoundC = src.pRoundC, *pFinalC = src.pFinalC,
                      *pH = src.pH, *idx1 = src.idx1, *idx2 = src.idx2;
        haveHpoly_ = src.haveCoefs;
        if (!src.sliced) {                       // whole sections: step to this rank's slices
            pA += wr_.lo * 64; pB1 += wr_.lo * 64; pB2 += wr_.lo * 128; pRoundC += c1.lo * 64; pFinalC += c2.lo * 64; pH += hr_.lo * 64;
            idx1 += c1.lo * 4; idx2 += c2.lo * 4;
        } else {                                 // slices: each must hold this rank's range (a short buffer would be read past its end)
            const uint64_t need[8] = {(wr_.hi - wr_.lo) * 64, (wr_.hi - wr_.lo) * 64, (wr_.hi - wr_.lo) * 128, (c1.hi - c1.lo) * 64,
                                      (c2.hi - c2.lo) * 64, (hr_.hi - hr_.lo) * 64, (c1.hi - c1.lo) * 4, (c2.hi - c2.lo) * 4};
            const uint8_t* ptr[8] = {pA, pB1, pB2, pRoundC, pFinalC, pH, idx1, idx2};
            static const char* const what[8] = {"points_a", "points_b1", "points_b2", "points_round_c", "points_final_c", "points_h",
                                                "round_indexes", "final_round_indexes"};
            for (int k = 0; k < 8; k++) {
                if (need[k] && !ptr[k]) throw std::invalid_argument(std::string("Null ") + what[k] + " slice");
                if (src.sliceBytes && src.sliceBytes[k] < need[k])
                    throw std::invalid_argument(std::string(what[k]) + " slice is shorter than this rank's range: " + std::to_string(src.sliceBytes[k]) +
                                                " bytes, needed " + std::to_string(need[k]));
            }
            if (src.haveCoefs && !src.coefs && src.nCoefs) throw std::invalid_argument("Null coefficient records");
        }
        roundIdx_.resize(c1.hi - c1.lo); finalIdx_.resize(c2.hi - c2.lo);
        if (!roundIdx_.empty()) memcpy(roundIdx_.data(), idx1, roundIdx_.size() * 4);
        if (!finalIdx_.empty()) memcpy(finalIdx_.data(), idx2, finalIdx_.size() * 4);
        for (uint32_t i : roundIdx_) if (i >= M) throw std::range_error("round index outside the witness");
        for (uint32_t i : finalIdx_) if (i >= M) throw std::range_error("final round index outside the witness");

        ugCheck(ug_ctx_create(&d_.ctx, device));
        // the H branch (H polynomial, its schedule, the H product) has a stream of its own, as in Groth16Prover: a rank of a
        // many-GPU prover drives it beside its queued witness products (witnessMsmBegin), and ULTRAGROTH_OVERLAP=1 lets it run
        // beside them on one GPU as well; by default the second stream is ordered behind the first on the device
        ugCheck(ug_ctx_create(&d_.ctx2, device));
        {   // SPARSE B (DeviceProver, sparseBSupport): as for Groth16 -- B1 and B2 compacted over the signals with a real B point
            std::vector<uint32_t> bSupport;
            std::vector<uint8_t> b1c, b2c;
            sparseB_ = count == 1 && !src.sliced && sparseBSupport(pB1, pB2, wr_.hi - wr_.lo, bSupport, b1c, b2c);
            nB_ = sparseB_ ? bSupport.size() : 0;
            if (sparseB_) {
                ugCheck(ug_bases_create_g1(d_.ctx, pA, wr_.hi - wr_.lo, wr_.lo, &d_.A));
                ugCheck(ug_bases_create_g1(d_.ctx, b1c.data(), nB_, 0, &d_.Bc1));
                ugCheck(ug_bases_create_g2(d_.ctx, b2c.data(), nB_, 0, &d_.Bc2));
                ugCheck(ug_index_create(d_.ctx, bSupport.data(), nB_, &d_.bIdx));
                ugCheck(ug_dvec_create(d_.ctx, nB_, &d_.wB));
                ugCheck(ug_schedule_create(d_.ctx, &d_.sB));
            }
        }
        if (sparseB_) {
        } else if (fusedGroups()) {                     // A and B1 share the witness scalars: one interleaved group
            const void* hosts[2] = {pA, pB1};
            const uint64_t counts[2] = {wr_.hi - wr_.lo, wr_.hi - wr_.lo}, firsts[2] = {wr_.lo, wr_.lo};
            ugCheck(ug_bases_create_group_g1(d_.ctx, 2, hosts, counts, firsts, wr_.lo, wr_.hi - wr_.lo, 0, &d_.G));
        } else {
            ugCheck(ug_bases_create_g1(d_.ctx, pA, wr_.hi - wr_.lo, wr_.lo, &d_.A));
            ugCheck(ug_bases_create_g1(d_.ctx, pB1, wr_.hi - wr_.lo, wr_.lo, &d_.B1));
        }
        if (!sparseB_) ugCheck(ug_bases_create_g2(d_.ctx, pB2, wr_.hi - wr_.lo, wr_.lo, &d_.B2));
        // the round / final sets are multiplied with GATHERED scalars (position k of the slice's index list), so their
        // slices count from 0
        ugCheck(ug_bases_create_g1(d_.ctx, pFinalC, c2.hi - c2.lo, 0, &d_.C));
        ugCheck(ug_bases_create_g1(d_.ctx, pRoundC, c1.hi - c1.lo, 0, &d_.roundC));
        ugCheck(ug_bases_create_g1(d_.ctx2, pH, hr_.hi - hr_.lo, hr_.lo, &d_.H));
        if (haveHpoly_ && !d_.hp) ugCheck(ug_hpoly_create(d_.ctx2, coefs, hdr_.nCoefs, hdr_.domainSize, hdr_.nVars, &d_.hp));
        ugCheck(ug_dvec_create(d_.ctx, M, &d_.w));
        wCur_ = d_.w;
        witness_.attach(d_.ctx, M, &d_.w, &d_.w2);
        ugCheck(ug_dvec_create(d_.ctx2, N, &d_.h));
        uint64_t auxN = std::max<uint64_t>(roundIdx_.size(), finalIdx_.size());
        ugCheck(ug_dvec_create(d_.ctx, auxN ? auxN : 1, &d_.aux));
        ugCheck(ug_schedule_create(d_.ctx, &d_.sw));
        ugCheck(ug_schedule_create(d_.ctx2, &d_.sh));
        ugCheck(ug_schedule_create(d_.ctx, &d_.saux));
        ugCheck(ug_index_create(d_.ctx, roundIdx_.data(), roundIdx_.size(), &d_.roundIdx));
        ugCheck(ug_index_create(d_.ctx, finalIdx_.data(), finalIdx_.size(), &d_.finalIdx));
        std::vector<TableGroup> groups = tableGroups();
        tableBytes = planWindowTables(d_.ctx, groups);
    }

public:
    std::vector<TableGroup> tableGroups() override {
        std::vector<TableGroup> groups(4);
        const uint64_t nw = wr_.hi - wr_.lo;
        if (d_.G) { groups[0].g1 = {d_.G}; groups[0].n1 = {2 * nw}; } else { groups[0].g1 = {d_.A, d_.B1}; groups[0].n1 = {nw, nw}; }
        groups[0].g2 = {d_.B2}; groups[0].n2 = {nw};
        groups[0].scalars = nw; groups[0].c = &tableW_;
        groups[1].g1 = {d_.roundC}; groups[1].n1 = {roundIdx_.size()}; groups[1].scalars = roundIdx_.size(); groups[1].c = &tableC1_;
        groups[2].g1 = {d_.C}; groups[2].n1 = {finalIdx_.size()}; groups[2].scalars = finalIdx_.size(); groups[2].c = &tableC2_;
        groups[3].g1 = {d_.H}; groups[3].n1 = {hr_.hi - hr_.lo}; groups[3].scalars = hr_.hi - hr_.lo; groups[3].c = &tableH_;
        if (sparseB_) {               // A alone over the witness schedule; B1 and B2 compacted, with a schedule of their own
            groups[0].g1 = {d_.A}; groups[0].n1 = {nw};
            groups[0].g2.clear(); groups[0].n2.clear();
            groups.resize(5);
            groups[4].g1 = {d_.Bc1}; groups[4].n1 = {nB_};
            groups[4].g2 = {d_.Bc2}; groups[4].n2 = {nB_};
            groups[4].scalars = nB_; groups[4].c = &d_.tableB;
        }
        return groups;
    }
    void trimWorkspaces() override {
        std::lock_guard<std::mutex> turn(proveMutex);
        ug_schedule_trim(d_.sw); ug_schedule_trim(d_.sh); ug_schedule_trim(d_.saux);
        if (d_.sB) ug_schedule_trim(d_.sB);
        ug_ctx_trim(d_.ctx); ug_ctx_trim(d_.ctx2);
        witness_.trim(wCur_);
    }

    const ZkeyHeader& header() const { return hdr_; }

    // Staging: parse the .uwtns and copy the signals into the leased buffer -- no turn on the device needed, so the witness
    // of a waiting call is copied while another proof runs (as Groth16Prover::proveTurn).
    void stage(StagedWitness& sl, const void* wtns, unsigned long long wtnsSize) {
        std::lock_guard<std::mutex> st(witness_.stageMutex);
        auto tLoad0 = std::chrono::steady_clock::now();
        BinFile f(wtns, wtnsSize, "wtns", 2);
        WtnsHeader wh = loadWtnsHeader(f);
        if (hdr_.nVars != wh.nVars)
            throw InvalidWitnessLengthException("Invalid witness length. Circuit: " + std::to_string(hdr_.nVars) +
                                                ", witness: " + std::to_string(wh.nVars));
        if (!wh.primeIsBn254) throw std::invalid_argument("different wtns curve");
        const uint64_t M = hdr_.nVars;
        const uint8_t* signals0 = checkedSection(f, 2, M * 32);       // the reference copies these (prover.cpp:283-285);
                                                                      // here the copy lives in HBM and is patched there
        auto u32Section = [&](uint32_t id) {
            std::vector<uint32_t> v(f.sectionSize(id) >> 2);
            memcpy(v.data(), f.sectionData(id), v.size() * 4);
            return v;
        };
        sl.chunks = u32Section(3); sl.freq = u32Section(4); sl.wIdx = u32Section(5); sl.pIdx = u32Section(6);
        if (sl.wIdx.size() != sl.pIdx.size()) throw std::range_error("uwtns: wtns_indxs and push_indxs differ in length");
        sl.publicPart.assign(signals0, signals0 + ((size_t)hdr_.nPublic + 1) * 32);
        ugCheck(ug_dvec_upload_idle(sl.buf, signals0, M));
        sl.uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tLoad0).count();
    }
    // A rank of a many-device prover (MultiUltraGrothProver): the .uwtns parsed as stage() does, but only the signals
    // [lo, hi) copied over this rank's PCIe link, into the current buffer -- the rest arrives from the peers' HBM
    // (witnessVec / ug_dvec_copy_via), after which witnessGathered() makes the witness the prover's. One thread, no turn.
    void loadWitnessSlice(const void* wtns, unsigned long long wtnsSize, uint64_t lo, uint64_t hi) {
        if (witnessQueued_ == 1) throw std::invalid_argument("the queued witness products still read the witness (ug_groth16_prover_witness_msm_end)");
        auto tLoad0 = std::chrono::steady_clock::now();
        BinFile f(wtns, wtnsSize, "wtns", 2);
        WtnsHeader wh = loadWtnsHeader(f);
        if (hdr_.nVars != wh.nVars)
            throw InvalidWitnessLengthException("Invalid witness length. Circuit: " + std::to_string(hdr_.nVars) +
                                                ", witness: " + std::to_string(wh.nVars));
        if (!wh.primeIsBn254) throw std::invalid_argument("different wtns curve");
        const uint64_t M = hdr_.nVars;
        if (lo > hi || hi > M) throw std::invalid_argument("witness slice outside [0, nVars]");
        const uint8_t* signals0 = checkedSection(f, 2, M * 32);
        auto u32Section = [&](uint32_t id) {
            std::vector<uint32_t> v(f.sectionSize(id) >> 2);
            memcpy(v.data(), f.sectionData(id), v.size() * 4);
            return v;
        };
        StagedWitness sl;
        sl.buf = wCur_;
        sl.chunks = u32Section(3); sl.freq = u32Section(4); sl.wIdx = u32Section(5); sl.pIdx = u32Section(6);
        if (sl.wIdx.size() != sl.pIdx.size()) throw std::range_error("uwtns: wtns_indxs and push_indxs differ in length");
        sl.publicPart.assign(signals0, signals0 + ((size_t)hdr_.nPublic + 1) * 32);
        if (hi > lo) ugCheck(ug_dvec_upload_range(wCur_, signals0 + lo * 32, lo, hi - lo, nullptr));
        sl.uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tLoad0).count();
        adopt(sl);
        witnessLoaded_ = false;                      // (until the peers' slices are here: witnessGathered)
    }
    ug_dvec* witnessVec() { return wCur_; }
    void witnessGathered() { witnessLoaded_ = true; }
    // the staged witness becomes the prover's (proveMutex held)
    void adopt(StagedWitness& sl) {
        wCur_ = sl.buf;
        chunks_.swap(sl.chunks); freq_.swap(sl.freq); wIdx_.swap(sl.wIdx); pIdx_.swap(sl.pIdx);
        publicPart_.swap(sl.publicPart);
        uploadMs = sl.uploadMs;
        ugCheck(ug_ctx_timings(d_.ctx, nullptr, nullptr, 1)); ugCheck(ug_ctx_timings(d_.ctx2, nullptr, nullptr, 1));
        witnessLoaded_ = true; committed_ = false; haveRoundScalar_ = false;
        if (trace_) fprintf(stderr, "[ultragroth] %-28s %8.3f ms\n", "parse uwtns + witness upload", uploadMs);
        tPhase_ = std::chrono::steady_clock::now();
    }
    // ---- phases (a sharded proof calls them one by one, see include/prover.h; proveTurn() below strings them together) ----
    // (the caller drives the phases of one proof from one thread and holds no lock)
    void loadWitness(const void* wtns, unsigned long long wtnsSize) override {
        if (witnessQueued_ == 1) throw std::invalid_argument("the queued witness products still read the witness (ug_groth16_prover_witness_msm_end)");
        WitnessLease lease(witness_);
        stage(*lease, wtns, wtnsSize);
        std::lock_guard<std::mutex> turn(proveMutex);
        adopt(*lease);
    }

    // round 1: this rank's part of the commitment to the round witnesses (ultra_groth.cpp:415-419, execute_round :161-184)
    void roundCommit(uint8_t* out64) override {
        if (!witnessLoaded_) throw std::invalid_argument("no witness loaded");
        ugCheck(ug_dvec_gather_index(d_.aux, wCur_, d_.roundIdx));
        mark("round gather");
        buildSchedule(d_.saux, d_.aux, 0, roundIdx_.size(), tableC1_);
        ugCheck(ug_msm_g1(d_.ctx, d_.roundC, d_.saux, 0, out64));
        mark("round MSM");
    }
    // on ONE rank, with the sum of all parts: draws the round randomness (:173) and blinds the commitment (:176)
    void roundFinish(const uint8_t* total64, uint8_t* commit64) override {
        uint8_t rk[32];
        drawBlinding(rk);
        memcpy(rkw_, rk, 32);
        haveRoundScalar_ = true;
        G1XYZZ commit = xyzz_add(g1FromRecord(total64), xyzz_mul_scalar_w4(g1FromRecord(hdr_.delta1), rkw_));   // final_delta1
        g1ToRecord(commit64, commit);
    }
    // on EVERY rank, with the blinded commitment: Fiat-Shamir challenge and the lookup signals it determines
    void applyCommitment(const uint8_t* commit64) override {
        if (!witnessLoaded_) throw std::invalid_argument("no witness loaded");
        memcpy(commitRec_, commit64, 64);
        // derive_challenge (:33-58): keccak256(x_BE32 || y_BE32) as a big-endian integer
        u32 cx[8], cy[8], w8[8];
        memcpy(w8, commitRec_, 32); to_normal(cx, from_mont256<FqParams>(w8));
        memcpy(w8, commitRec_ + 32, 32); to_normal(cy, from_mont256<FqParams>(w8));
        uint8_t buf[64], ch[32];
        for (int i = 0; i < 32; i++) {
            buf[i] = (uint8_t)(cx[7 - (i >> 2)] >> (24 - 8 * (i & 3)));
            buf[32 + i] = (uint8_t)(cy[7 - (i >> 2)] >> (24 - 8 * (i & 3)));
        }
        keccak256(ch, buf, 64);
        u32 chw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 32; i++) chw[7 - (i >> 2)] |= (u32)ch[i] << (24 - 8 * (i & 3));
        Fr rand = from_normal<FrParams>(chw);              // reduces values >= r, like fromMpz + toMontgomery
        mark("commit + challenge");
        // compute_lookup (:62-106): the new values are written into the device copy of the witness (last write wins)
        applyLookup(publicPart_, chunks_, freq_, wIdx_, pIdx_, rand);
        mark("lookup");
        committed_ = true;
    }
    // final round (execute_final_round :187-399), this rank's slices: A | B1 | B2 | C2 records of the partials block
    void runWitnessMsm(uint8_t* partials, bool = true) override {
        if (!committed_) throw std::invalid_argument("the round commitment has not been applied");
        if (witnessQueued_) throw std::invalid_argument("witness products are queued on this prover (ug_groth16_prover_witness_msm_end)");
        memset(partials, 0, UG_GROTH16_PARTIALS_SIZE);
        buildSchedule(d_.sw, wCur_, wr_.lo, wr_.hi - wr_.lo, tableW_);
        enqueueWitnessProducts(d_, d_.ctx, d_.sw, partials, partials + 64, partials + 128, nullptr, 0, false, wCur_);      // MSM1-3 :201,214,227
        ugCheck(ug_ctx_collect(d_.ctx));
        mark("A, B1, B2 MSMs");
        ugCheck(ug_dvec_gather_index(d_.aux, wCur_, d_.finalIdx));                           // :439-445
        mark("final gather");
        buildSchedule(d_.saux, d_.aux, 0, finalIdx_.size(), tableC2_);
        ugCheck(ug_msm_g1(d_.ctx, d_.C, d_.saux, 0, partials + 256));                       // MSM4 :234
        mark("C MSM");
    }
    // The same in two calls, as Groth16Prover: _begin queues MSM1-3, the gather of the final witnesses and MSM4 on the witness
    // stream and returns; the caller drives the H branch (second stream) meanwhile; _end waits and returns the four sums.
    void witnessMsmBegin() override {
        if (!committed_) throw std::invalid_argument("the round commitment has not been applied");
        if (witnessQueued_) throw std::invalid_argument("the witness products are already queued (ug_groth16_prover_witness_msm_end)");
        memset(queuedParts_, 0, sizeof queuedParts_);
        QueueGuard inFlight(d_.ctx);
        queueFinalRoundProducts(queuedParts_);
        inFlight.done();
        witnessQueued_ = 1;
    }
    void witnessMsmEnd(uint8_t* partials) override {
        if (!witnessQueued_) throw std::invalid_argument("no witness products queued (ug_groth16_prover_witness_msm_begin)");
        witnessQueued_ = 0;
        QueueGuard inFlight(d_.ctx);
        ugCheck(ug_ctx_collect(d_.ctx));
        inFlight.done();
        collectTimings(1);
        memcpy(partials, queuedParts_, UG_GROTH16_PARTIALS_SIZE);
    }
    void witnessMsmAbandon() {
        if (witnessQueued_ == 1) ug_ctx_abandon(d_.ctx);
        witnessQueued_ = 0;
    }
    // MSM1-3 (:201,214,227), the gather of the final witnesses (:439-445) and MSM4 (:234), queued on the witness stream
    void queueFinalRoundProducts(uint8_t* sums) {
        buildSchedule(d_.sw, wCur_, wr_.lo, wr_.hi - wr_.lo, tableW_);
        enqueueWitnessProducts(d_, d_.ctx, d_.sw, sums, sums + 64, sums + 128, nullptr, 0, false, wCur_);
        ugCheck(ug_dvec_gather_index(d_.aux, wCur_, d_.finalIdx));
        buildSchedule(d_.saux, d_.aux, 0, finalIdx_.size(), tableC2_);
        const ug_bases* setC[1] = {d_.C};
        void* outC[1] = {sums + 256};
        ugCheck(ug_msm_batch_enqueue(d_.ctx, 1, setC, d_.saux, nullptr, outC));
    }
    void hpolyChain(int which, void* deviceOut) override {
        if (!committed_) throw std::invalid_argument("the round commitment has not been applied");
        if (!haveHpoly_) throw std::invalid_argument("this rank was created without the coefficient matrix");
        ug_dvec* v = nullptr;
        ugCheck(ug_dvec_wrap(d_.ctx2, deviceOut, hdr_.domainSize, &v));
        int rc = ug_hpoly_chain(d_.hp, wCur_, which, v);
        ug_dvec_destroy(v);
        ugCheck(rc);
        ugCheck(ug_ctx_sync(d_.ctx2));               // the vector is complete when the call returns (the caller sends it on)
    }
    void hpolyCombine(void* da, void* db, void* dc) override {
        uint64_t cnt = hr_.hi - hr_.lo;
        ug_dvec *a = nullptr, *b = nullptr, *c = nullptr;
        ugCheck(ug_dvec_wrap(d_.ctx2, da, cnt, &a));
        ugCheck(ug_dvec_wrap(d_.ctx2, db, cnt, &b));
        ugCheck(ug_dvec_wrap(d_.ctx2, dc, cnt, &c));
        int rc = ug_hpoly_combine(d_.hp, a, b, c, hr_.lo, cnt, d_.h);
        ug_dvec_destroy(a); ug_dvec_destroy(b); ug_dvec_destroy(c);
        ugCheck(rc);
    }
    void hRange(unsigned long long* first, unsigned long long* count, unsigned long long* domain) const override {
        if (first) *first = hr_.lo;
        if (count) *count = hr_.hi - hr_.lo;
        if (domain) *domain = hdr_.domainSize;
    }
    // MSM5 (:322) on this rank's slice of h (which must be in d_.h); only the H record of partials is written
    void runHMsm(uint8_t* partials) override {
        memset(partials, 0, UG_GROTH16_PARTIALS_SIZE);
        buildSchedule(d_.sh, d_.h, hr_.lo, hr_.hi - hr_.lo, tableH_);
        ugCheck(ug_msm_g1(d_.ctx2, d_.H, d_.sh, 0, partials + 320));
        mark("H MSM");
        collectTimings(witnessQueued_ == 1 ? 2 : 3);          // (queued witness products: their stream is read when they are collected)
    }
    // device time of both streams since the witness was adopted; which: 1 = witness stream, 2 = H branch (each waits for its stream)
    void collectTimings(int which) {
        if (which & 1) ugCheck(ug_ctx_timings(d_.ctx, &m1_, &f1_, 0));
        if (which & 2) ugCheck(ug_ctx_timings(d_.ctx2, &m2_, &f2_, 0));
        msmMs_ = m1_ + m2_; fftMs_ = f1_ + f2_;
    }
    // on the rank that ran roundFinish, with the summed partials: r and s (:345-346), the blinded proof, the JSON texts
    void finish(const uint8_t* sums, std::string& proof, std::string& pub) override {
        uint8_t r[32], s[32];
        drawBlinding(r); drawBlinding(s);
        HostTerms t = hostTerms(r, s);
        finishWith(sums, r, s, t, proof, pub);
    }

    struct HostTerms { BlindingTerms b; G1XYZZ roundTerm; };
    HostTerms hostTerms(const uint8_t r[32], const uint8_t s[32]) {
        if (!haveRoundScalar_) throw std::invalid_argument("finish on a rank that did not close the round");
        HostTerms t;
        auto fr = std::async(std::launch::async, [&] { return xyzz_mul_scalar_w4(g1FromRecord(hdr_.roundDelta1), rkw_); });   // :386-388
        t.b = blindingTerms(hdr_, r, s);
        t.roundTerm = fr.get();
        return t;
    }
    void finishWith(const uint8_t* sums, const uint8_t r[32], const uint8_t s[32], const HostTerms& t, std::string& proof,
                    std::string& pub) {
        uint8_t A[64], B[128], C[64];
        blind(A, B, C, sums, sums + 64, sums + 128, sums + 256, sums + 320, hdr_, r, s, t.b, &t.roundTerm);
        // keys pi_a, pi_b, pi_f, pi_r, protocol (src/ultra_groth.cpp:476-513)
        proof = "{\"pi_a\":" + g1Json(A) + ",\"pi_b\":" + g2Json(B) + ",\"pi_f\":" + g1Json(C) + ",\"pi_r\":" + g1Json(commitRec_) +
                ",\"protocol\":\"ultragroth\"}";
        pub = publicJson(publicPart_.data(), hdr_.nPublic, hdr_.randIndx);                  // prover.cpp:89-105
        mark("blinding + JSON");
    }

    // (every caller comes through proveTurn, which this class overrides; kept for the interface)
    void prove(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub) override {
        proveTurn(wtns, wtnsSize, proof, pub, nullptr, Around());
    }
    void proveTurn(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub, std::mutex* device,
                   const Around& around) override {
        auto t0 = std::chrono::steady_clock::now();
        WitnessLease lease(witness_);
        stage(*lease, wtns, wtnsSize);
        std::unique_lock<std::mutex> card;
        if (device) card = std::unique_lock<std::mutex>(*device);
        std::lock_guard<std::mutex> turn(proveMutex);
        AroundGuard bracket(around);
        bracket.begin();
        // ULTRAGROTH_TRACE=1: host wall-clock per phase on stderr (where the non-MSM, non-FFT time of a proof goes)
        trace_ = getenv("ULTRAGROTH_TRACE") && atoi(getenv("ULTRAGROTH_TRACE")) != 0;
        if (!haveHpoly_) throw std::invalid_argument("this rank was created without the coefficient matrix");
        adopt(*lease);
        QueueGuard inFlight(d_.ctx, d_.ctx2);            // (declared before `terms`: the host threads join first, then the device is drained)
        uint8_t part[64], commit[64];
        roundCommit(part);
        roundFinish(part, commit);
        applyCommitment(commit);
        // r and s (:345-346) are drawn here, still after the round randomness as in the reference, so that the multiples
        // of the deltas that need only the blinding scalars are formed on host threads beside the device work
        uint8_t r[32], s[32];
        drawBlinding(r); drawBlinding(s);
        auto terms = std::async(std::launch::async, [&] { return hostTerms(r, s); });
        // (a std::async future joins in its destructor, and r, s are declared before it: they outlive the threads)
        uint8_t sums[UG_GROTH16_PARTIALS_SIZE];
        if (trace_) {                                    // phase by phase, with a host wait (and a line on stderr) after each
            uint8_t hpart[UG_GROTH16_PARTIALS_SIZE];
            runWitnessMsm(sums);
            ugCheck(ug_hpoly_run(d_.hp, wCur_, d_.h));                                       // FFT block :243-320
            mark("H polynomial");
            runHMsm(hpart);
            memcpy(sums + 320, hpart + 320, 64);
        } else {
            // the whole final round queued on the stream, ONE host wait: MSM1-3 (:201,214,227), the gather of the final
            // witnesses (:439-445) and MSM4 (:234), the FFT block (:243-320), MSM5 (:322)
            // (two streams since round 3: the H branch has its own; by default it is ordered behind the witness products on the
            // device, ULTRAGROTH_OVERLAP=1 drops that edge -- the witness is complete either way: the lookup writes end with a
            // host wait)
            memset(sums, 0, sizeof sums);
            queueFinalRoundProducts(sums);
            const char* ov = getenv("ULTRAGROTH_OVERLAP");
            if (ov && atoi(ov) == 0) ugCheck(ug_ctx_wait(d_.ctx2, d_.ctx));      // (default: beside, as for Groth16)
            ugCheck(ug_hpoly_run(d_.hp, wCur_, d_.h));
            buildSchedule(d_.sh, d_.h, hr_.lo, hr_.hi - hr_.lo, tableH_);
            const ug_bases* setH[1] = {d_.H};
            void* outH[1] = {sums + 320};
            ugCheck(ug_msm_batch_enqueue(d_.ctx2, 1, setH, d_.sh, nullptr, outH));
            ugCheck(ug_ctx_collect(d_.ctx2));
            ugCheck(ug_ctx_collect(d_.ctx));
            collectTimings(3);
        }
        inFlight.done();
        finishWith(sums, r, s, terms.get(), proof, pub);
        totalMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        bracket.end();
    }

    unsigned long long proofBufferMinSize() const override { return PROOF_MIN_ULTRA; }
    unsigned long long publicBufferMinSize() const override { return publicMin((unsigned long long)hdr_.nPublic - 1); }
    void timings(double* msm, double* fft, double* total) const override {
        if (msm) *msm = msmMs_;
        if (fft) *fft = fftMs_;
        if (total) *total = totalMs_;
    }
    ug_ctx* ctx() override { return d_.ctx; }

private:
    static void putPlain(uint8_t* dst, const Fr& v) { u32 w[8]; to_normal(w, v); memcpy(dst, w, 32); }
    // compute_lookup (src/ultra_groth.cpp:62-106). The table -- inv2[i] = 1 / (i + rand) (0 when the sum is 0) and
    // prod[i] = freq[i] * inv2[i], with the reference's (int, Element) overload semantics -- is made on the device
    // (ug_fr_lookup_table, one lane per row), and so are the writes into the witness (ug_dvec_apply_lookup; the
    // reference's push_vector with its per-chunk copies of inv2 is never materialised).
    // publicPart receives the writes that land on public signals (they go into public.json).
    void applyLookup(std::vector<uint8_t>& publicPart, const std::vector<uint32_t>& chunks, const std::vector<uint32_t>& freq,
                     const std::vector<uint32_t>& wIdx, const std::vector<uint32_t>& pIdx, const Fr& rand) {
        const size_t L = freq.size(), Cn = chunks.size();
        std::vector<uint8_t> table((2 * L + 1) * 32);                 // [rand | inv2 | prod], plain integers
        uint8_t randPlain[32];
        putPlain(randPlain, rand);
        ugCheck(ug_fr_lookup_table(d_.ctx, randPlain, freq.data(), L, table.data()));     // one lane per row on the device
        ugCheck(ug_dvec_apply_lookup(wCur_, wIdx.data(), pIdx.data(), wIdx.size(), chunks.data(), Cn, table.data(), L));
        // the same writes for the public signals, in order (a later write overwrites an earlier one)
        for (size_t i = 0; i < wIdx.size(); i++) {
            if (wIdx[i] > hdr_.nPublic) continue;
            const uint64_t p = pIdx[i];
            const uint64_t t = p == 0 ? 0 : p <= Cn ? 1 + (uint64_t)chunks[p - 1] : p - Cn;
            memcpy(publicPart.data() + (size_t)wIdx[i] * 32, table.data() + t * 32, 32);
        }
    }

    ZkeyHeader hdr_;
    std::vector<uint8_t> vk_;
    void mark(const char* what) {
        if (!trace_) return;
        ug_ctx_sync(d_.ctx); ug_ctx_sync(d_.ctx2);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ultragroth] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tPhase_).count());
        tPhase_ = now;
    }
    std::vector<uint32_t> roundIdx_, finalIdx_;                    // this rank's slices of the zkey's two index lists
    std::vector<uint32_t> chunks_, freq_, wIdx_, pIdx_;            // uwtns sections 3-6 of the loaded witness
    ug_dvec* wCur_ = nullptr;                                      // the witness the device part reads and patches (d_.w, d_.w2 own the two buffers)
    WitnessBuffers witness_;
    std::vector<uint8_t> publicPart_;                              // signals 0..nPublic, patched by the lookup writes
    Range wr_{0, 0}, hr_{0, 0};
    uint8_t commitRec_[64] = {0};
    u32 rkw_[8] = {0};
    bool witnessLoaded_ = false, committed_ = false, haveRoundScalar_ = false, trace_ = false, haveHpoly_ = true;
    std::chrono::steady_clock::time_point tPhase_;
    int tableW_ = 0, tableC1_ = 0, tableC2_ = 0, tableH_ = 0;      // fixed-base table widths per schedule group (0: classic)
    bool sparseB_ = false;             // B1 / B2 kept compacted over the signals with a real B point (DeviceProver)
    uint64_t nB_ = 0;
    DeviceProver d_;
    double msmMs_ = 0, fftMs_ = 0, totalMs_ = 0;
    double m1_ = 0, f1_ = 0, m2_ = 0, f2_ = 0;                     // device ms per stream (witness stream, H branch)
    int witnessQueued_ = 0;                                        // witnessMsmBegin .. witnessMsmEnd
    uint8_t queuedParts_[UG_GROTH16_PARTIALS_SIZE] = {};
public:
    ug_ctx* ctx2() { return d_.ctx2; }                             // the H branch's context
};


// =================================================================================================================
// One proof on SEVERAL devices of the node behind the reference's own API (ULTRAGROTH_DEVICES=0,1,2,...): the prover the
// extern "C" create calls return then holds one sharded Groth16Prover per listed device -- rank k owns the k-th contiguous
// range of every base-point section (what BASELINE.json's north_star asks: "shard by base-point range across the GPUs of one
// node") -- and groth16_prover_prove drives them from one host thread per device:
//   part 0 of the witness (the rank's scalars) -> its A | B1 | B2 | C partial sums          src/groth16.cpp:55-64
//   ranks 0..2 (k mod R) also upload the rest of the witness and run one iFFT / twist / FFT chain each      :66-140
//   every rank copies its slice of the three evaluation vectors from the chain ranks' devices (peer copies over xGMI,
//   ug_dvec_copy), forms its slice of h and its H partial sum                                               :142-154
//   the 384-byte partial blocks are added on the host (an EC addition, not an RCCL reduction: the data is five points per
//   rank) and rank 0 blinds and serialises                                                                  :158-250
// No Python, no torch, no launcher: `prover <zkey> <wtns> <proof.json> <public.json>` uses the node as it is. (bench.py's
// one-process-per-GPU torchrun path stays the one the driver measures; both call the same phase functions.)
// A device may be listed more than once: "0,0,0,0" rehearses four ranks on one GPU (tests).
std::vector<int> devicesFromEnv() {
    std::vector<int> out;
    const char* e = getenv("ULTRAGROTH_DEVICES");
    if (!e || !*e) return out;
    const char* p = e;
    while (*p) {
        char* end = nullptr;
        long v = strtol(p, &end, 10);
        if (end == p || v < 0) throw std::invalid_argument(std::string("ULTRAGROTH_DEVICES: not a list of device numbers: ") + e);
        out.push_back((int)v);
        p = end;
        if (*p == ',') p++;
        else if (*p) throw std::invalid_argument(std::string("ULTRAGROTH_DEVICES: not a list of device numbers: ") + e);
    }
    if (out.size() > 64) throw std::invalid_argument("ULTRAGROTH_DEVICES: more than 64 ranks");
    return out;
}

// One iFFT/twist/FFT chain costs about this fraction of ALL the witness MSMs of a proof (2^24, tools/phase_times.py), so a
// rank that also runs chains gets fewer points (the same split bench.py makes for its ranks). (End of round 3: a chain is 5.9 ms
// since the tiled mat-vec, 7.1 before; with 0.063 / 0.063 / 0.080 a chain rank of eight ended after 21.1 ms, a plain one after
// 22.1: per point a rank's products cost 6.0 ms per million on top of 4.5 ms that do not depend on the slice.)
std::vector<Range> balancedWitnessRanges(uint64_t nVars, int count) {
    static const double CHAIN_SHARE[3] = {0.058, 0.058, 0.070};
    std::vector<double> extra(count, 0.0), share(count);
    for (int c = 0; c < 3; c++) extra[c % count] += CHAIN_SHARE[c];
    double base = 1.0, tot = 0;
    for (double x : extra) base += x;
    base /= count;
    for (int k = 0; k < count; k++) { share[k] = std::max(base - extra[k], 0.0); tot += share[k]; }
    std::vector<Range> out(count);
    double run = 0;
    uint64_t lo = 0;
    for (int k = 0; k < count; k++) {
        run += share[k];
        uint64_t hi = k == count - 1 ? nVars : std::min<uint64_t>(nVars, (uint64_t)((double)nVars * run / tot + 0.5));
        if (hi < lo) hi = lo;
        out[k] = Range{lo, hi};
        lo = hi;
    }
    return out;
}

// ---- the layout of a many-device Groth16 prover: which rank owns what (ShardLayout) -----------------------------------------
// R ranks in P groups of B = R / P: group p holds one base-point range of the witness-indexed sections, its B ranks the bucket
// classes of that range's products (B = 1: base-point ranges only, the layout of rounds 1-3). Chains k of the H polynomial run on
// ranks k mod R. The shares are chosen so that the ranks finish together, from a cost model in units of "all witness products of
// a proof" measured at 2^24 on one rank of eight (tools/phase_times.py, profiles/r04_rank_phases_*.txt): a chain costs
// CHAIN_SHARE; a rank's witness products cost CLASS_FIXED + its share of the entries (the part that does not shrink with the
// share: every rank recodes all scalars of its range, and the kernels behind the accumulation have their latency floor); its H
// product H_FIXED + H_PART * its share of h. With five ranks or more the chain ranks take no part of the H product at all.
constexpr double CLASS_FIXED = 0.027, H_FIXED = 0.014, H_PART = 0.175;
static const double CHAIN_SHARE_CLASSES[3] = {0.058, 0.058, 0.070};
// device memory a rank needs for the window tables of n witness points and their schedules (planTableWidthsAhead's arithmetic)
uint64_t classTablesNeed(uint64_t n) {
    if (n < TABLES_MIN_SCALARS || n > TABLES_MAX_SCALARS) return ~(uint64_t)0;
    const int c = ug_msm_table_window(n);
    return ug_bases_tables_bytes(3 * n, 0, c) + ug_bases_tables_bytes(n, 1, c) + n * (3 * 64 + 128) + 24 * n * (uint64_t)((255 + c - 1) / c) + n * 32;
}
// P for `R` ranks (0 = decide here): ULTRAGROTH_SHARD=PxB when it fits R, else BASE-POINT RANGES (P = R). Bucket classes are
// built, tested and kept as an option, not chosen: measured on one rank of eight at 2^24 (profiles/r04_rank_phases_classes.txt),
// a class rank's witness products take 21.1 ms against 20.2 ms for the base-point rank -- it saves the thirteenth window and
// half the buckets (2.7 ms of accumulation) and pays them back recoding ALL scalars of its range for its eighth of the entries
// (schedule 3.3 instead of 0.8 ms); the per-entry rate of the accumulation depends on the entries in flight, not on the window
// width. ULTRAGROTH_SHARD=auto takes classes from four ranks on when the tables of a group's range fit a device (hbmBytes,
// 0 = 256 GiB), in as few groups as that allows.
int pointRangeGroups(uint64_t M, int R, int wanted, uint64_t hbmBytes) {
    auto valid = [&](int P) { return P >= 1 && P <= R && R % P == 0; };
    if (wanted) {                                   // a caller's own choice is taken or refused, never silently replaced
        if (!valid(wanted)) throw std::invalid_argument("point_ranges " + std::to_string(wanted) + " does not divide shard_count " + std::to_string(R));
        return wanted;
    }
    const char* e = getenv("ULTRAGROTH_SHARD");
    if (!e || !*e) return R;
    int P = 0, B = 0;
    if (sscanf(e, "%dx%d", &P, &B) == 2 && valid(P) && P * B == R) return P;
    if (strcmp(e, "auto") != 0 || R < 4) return R;
    if (!hbmBytes) hbmBytes = (uint64_t)256 << 30;
    for (P = 1; P < R; P *= 2) {
        if (!valid(P)) continue;
        const uint64_t need = classTablesNeed(M / (uint64_t)P + 1);
        if (need != ~(uint64_t)0 && need + ((uint64_t)8 << 30) <= hbmBytes - hbmBytes / 8) return P;
    }
    return R;
}
std::vector<ShardLayout> shardLayouts(uint64_t M, uint64_t N, int R, int P) {
    if (R < 1 || P < 1 || R % P) throw std::invalid_argument("invalid shard layout");
    std::vector<ShardLayout> out(R);
    for (int c = 0; c < 3; c++) out[c % R].chains |= 1u << c;
    const int B = R / P;
    if (B == 1) {                                   // base-point ranges only: chain ranks get fewer points, H split evenly
        const std::vector<Range> wr = balancedWitnessRanges(M, R);
        for (int k = 0; k < R; k++) { out[k].w = wr[k]; out[k].h = shardRange(N, k, R); out[k].sp = wr[k]; }
        return out;
    }
    // who takes part in the H product, and every rank's share of the witness entries
    std::vector<double> chain(R, 0.0), hs(R, 0.0), f(R, 0.0);
    for (int c = 0; c < 3; c++) chain[c % R] += CHAIN_SHARE_CLASSES[c];
    int takers = 0;
    for (int k = 0; k < R; k++) if (R < 5 || !out[k].chains) takers++;
    double total = 1.0;
    for (int k = 0; k < R; k++) {
        hs[k] = (R < 5 || !out[k].chains) ? 1.0 / takers : 0.0;
        total += chain[k] + CLASS_FIXED + (hs[k] > 0 ? H_FIXED + H_PART * hs[k] : 0.0);
    }
    const double T = total / R;
    double fsum = 0;
    for (int k = 0; k < R; k++) {
        f[k] = std::max(T - chain[k] - CLASS_FIXED - (hs[k] > 0 ? H_FIXED + H_PART * hs[k] : 0.0), 0.02);
        fsum += f[k];
    }
    for (double& x : f) x /= fsum;
    {   // h ranges: contiguous, in rank order
        double run = 0;
        uint64_t lo = 0;
        for (int k = 0; k < R; k++) {
            run += hs[k];
            const uint64_t hi = k == R - 1 ? N : std::min<uint64_t>(N, (uint64_t)((double)N * run + 0.5));
            out[k].h = Range{lo, std::max(lo, hi)};
            lo = out[k].h.hi;
        }
    }
    const int qLog = shardQLog(B);
    const uint32_t Q = 1u << qLog;
    if ((uint32_t)B > Q)                            // (every rank of a group owns at least one residue; Q is capped at 2^7)
        throw std::invalid_argument("bucket-class layout: " + std::to_string(B) + " ranks per point range, at most " + std::to_string(Q) + " residues");
    double before = 0;
    uint64_t wlo = 0;
    for (int p = 0; p < P; p++) {
        double F = 0;
        for (int k = p * B; k < (p + 1) * B; k++) F += f[k];
        const uint64_t whi = p == P - 1 ? M : std::min<uint64_t>(M, (uint64_t)((double)M * (before + F) + 0.5));
        before += F;
        // the group's residues: cnt_k ~ Q f_k / F, at least one each, by largest remainder
        std::vector<uint32_t> cnt(B, 1);
        std::vector<std::pair<double, int>> rem;
        uint32_t given = B;
        for (int j = 0; j < B; j++) {
            const double want = (double)(Q - B) * f[p * B + j] / F;          // (beyond the one every rank has)
            cnt[j] += (uint32_t)want; given += (uint32_t)want;
            rem.push_back({want - (double)(uint32_t)want, j});
        }
        std::sort(rem.begin(), rem.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
        for (size_t q = 0; given < Q; q = (q + 1) % rem.size()) { cnt[rem[q].second]++; given++; }
        uint32_t r0 = 0;
        const uint64_t nw = whi - wlo;
        for (int j = 0; j < B; j++) {
            ShardLayout& L = out[p * B + j];
            L.w = Range{wlo, whi};
            L.qLog = qLog; L.r0 = r0; L.cnt = cnt[j];
            L.sp = Range{wlo + nw * r0 / Q, wlo + nw * (r0 + cnt[j]) / Q};       // the special buckets: the same shares, by scalar range
            r0 += cnt[j];
        }
        wlo = whi;
    }
    return out;
}

class MultiGroth16Prover : public ProverBase {
public:
    MultiGroth16Prover(const void* zkey, unsigned long long zkeySize, const std::vector<int>& devices) {
        const int R = (int)devices.size();
        BinFile f(zkey, zkeySize, "zkey", 1);
        ZkeyHeader h = loadZkeyHeader(f, false);
        if (!h.rIsBn254) throw std::invalid_argument("zkey curve not supported");
        nPublic_ = h.nPublic; domain_ = h.domainSize; nVars_ = h.nVars;
        // the layout (shardLayouts): base-point ranges, unless ULTRAGROTH_SHARD asks for bucket classes (PxB, or auto: when the
        // tables of a group's range fit the devices, asked from the first one); if a class rank then cannot build its tables after
        // all, everything is created once more with base-point ranges
        const bool oneShot = g_oneShotProver;          // (thread-local: handed to the creating threads by value)
        uint64_t freeB = 0, totalB = 0;
        { ug_ctx* probe = nullptr; ugCheck(ug_ctx_create(&probe, devices[0])); ug_ctx_mem_info(probe, &freeB, &totalB); ug_ctx_destroy(probe); }
        const char* te = getenv("ULTRAGROTH_TABLES");
        const bool tablesOff = (te && te[0] == '0') || (oneShot && !(te && te[0] == '2'));
        int P = tablesOff ? R : pointRangeGroups(h.nVars, R, 0, freeB);
        for (;;) {
            const std::vector<ShardLayout> lay = shardLayouts(h.nVars, h.domainSize, R, P);
            ranks_.clear();
            ranks_.resize(R);
            // every rank uploads and converts its slices on its own device, all at once (ranks that share a device -- rehearsals --
            // and hold whole tables each: one after the other, or the devices' memory would be asked for R times over)
            std::vector<std::future<void>> jobs;
            auto make = [&, oneShot](int k) {
                g_oneShotProver = oneShot;
                ranks_[k].reset(new Groth16Prover(zkey, zkeySize, devices[k], k, R, nullptr, /*runsChain*/ lay[k].chains != 0, &lay[k]));
            };
            std::exception_ptr failure;
            for (int k = 0; k < R; k++) {
                bool shared = false;
                for (int q = 0; q < k; q++) shared = shared || devices[q] == devices[k];
                if (P < R && shared) { for (auto& j : jobs) { try { j.get(); } catch (...) { if (!failure) failure = std::current_exception(); } } jobs.clear(); }
                jobs.push_back(std::async(std::launch::async, make, k));
            }
            for (auto& j : jobs) { try { j.get(); } catch (...) { if (!failure) failure = std::current_exception(); } }
            if (!failure) { layouts_ = lay; break; }
            ranks_.clear();
            // once more with base-point ranges ONLY when a class rank could not get its window tables or its memory; a bad
            // zkey, a HIP error or a failed device fails the same way again, seconds to minutes later: rethrown as it is
            bool tablesOrMemory = false;
            try { std::rethrow_exception(failure); }
            catch (const TablesUnavailable&) { tablesOrMemory = true; }
            catch (const std::bad_alloc&) { tablesOrMemory = true; }
            catch (const std::exception& e) { tablesOrMemory = strstr(e.what(), "not enough device memory") || strstr(e.what(), "out of memory"); }
            catch (...) {}
            if (P == R || !tablesOrMemory) std::rethrow_exception(failure);
            P = R;
        }
        // the evaluation vectors of the three chains (on the devices of ranks k mod R) and every rank's slices of them
        for (int c = 0; c < 3; c++) {
            ugCheck(ug_dvec_create(ranks_[c % R]->ctx2(), domain_, &full_[c]));
        }
        slices_.resize(R);
        for (int k = 0; k < R; k++) {
            unsigned long long first = 0, cnt = 0;
            ranks_[k]->hRange(&first, &cnt, nullptr);
            for (int c = 0; c < 3; c++) ugCheck(ug_dvec_create(ranks_[k]->ctx2(), cnt ? cnt : 1, &slices_[k].v[c]));
        }
    }
    ~MultiGroth16Prover() override {
        for (auto& s : slices_) for (ug_dvec* v : s.v) ug_dvec_destroy(v);
        for (ug_dvec* v : full_) ug_dvec_destroy(v);
    }
    void prove(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub) override {
        const auto t0 = std::chrono::steady_clock::now();
        const int R = (int)ranks_.size();
        // checked once here (every rank checks again): the reference's error for a wrong witness comes before any device work
        { BinFile f(wtns, wtnsSize, "wtns", 2); (void)ranks_[0]->witnessData(f); }
        std::vector<std::array<uint8_t, UG_GROTH16_PARTIALS_SIZE>> parts(R);
        std::vector<std::exception_ptr> errs(R);
        // One host thread per rank, and on every rank both streams busy: the witness products are queued first and left to
        // run (witnessMsmBegin); the rank's chains, the copies of its slices of the three evaluation vectors, its combine and
        // its H product go to its second, high-priority stream meanwhile. The only edges between ranks are the three "chain c
        // is complete" events: a rank that waits for them spends that time on its witness products.
        std::promise<void> chainDone[3];
        std::shared_future<void> chainReady[3];
        for (int c = 0; c < 3; c++) chainReady[c] = chainDone[c].get_future().share();
        // THE WITNESS ON A NODE: every rank copies its own slice of the scalars over its own PCIe link (all links at once); a
        // chain rank, whose mat-vec reads the WHOLE witness, then collects the other slices from its peers' HBM over xGMI
        // (ug_dvec_copy_via: hipMemcpyPeerAsync) instead of pulling all of it through its one link -- 512 MiB at 2^24: 9.4 ms by
        // PCIe, ~1.3 ms for seven 64 MiB pieces over seven links. The ranks' ranges tile the witness in rank order (base-point
        // layouts), or repeat inside a group (bucket classes): pieces already covered are skipped. ULTRAGROTH_WITNESS_GATHER=0
        // keeps the round-4 form (a chain rank uploads the rest itself).
        const char* wg = getenv("ULTRAGROTH_WITNESS_GATHER");
        const bool gather = R > 1 && !(wg && wg[0] == '0');
        std::vector<std::promise<void>> sliceUp(R);
        std::vector<std::shared_future<void>> sliceReady(R);
        for (int k = 0; k < R; k++) sliceReady[k] = sliceUp[k].get_future().share();
        {
            std::vector<std::thread> th;
            for (int k = 0; k < R; k++)
                th.emplace_back([&, k] {
                    Groth16Prover& p = *ranks_[k];
                    bool told[3] = {false, false, false};
                    bool toldSlice = false;
                    try {
                        p.loadWitnessPart(wtns, wtnsSize, 0);
                        toldSlice = true;
                        sliceUp[k].set_value();
                        // with many ranks most of them wait for the chains: a chain rank then runs its chain first, alone (a
                        // chain beside the products gets ~40 % of the chip whatever the stream priorities say, 17 ms instead of
                        // 7 at 2^24 / 8 ranks); with few ranks it runs beside the products (bench.py: UG_BENCH_CHAIN_ORDER)
                        const bool chainFirst = R >= 5;
                        if (!chainFirst) p.witnessMsmBegin();
                        if (k < 3) {                                       // chains k, k + R, ... of the three
                            if (gather) {
                                uint64_t covered = 0;                      // [0, covered) of the witness is on this rank or on its way
                                const Range mine = p.witnessRange();
                                for (int q = 0; q < R; q++) {
                                    const Range w = q == k ? mine : ranks_[q]->witnessRange();
                                    const uint64_t lo = std::max(w.lo, covered);
                                    if (w.hi <= lo) continue;
                                    if (q != k && !(lo >= mine.lo && w.hi <= mine.hi)) {
                                        sliceReady[q].get();               // (a peer's failed upload is rethrown here)
                                        // the part of it this rank does not hold itself
                                        const uint64_t a = lo, b = w.hi;
                                        const uint64_t a1 = std::min(b, std::max(a, mine.lo)), b1 = std::max(a, std::min(b, mine.hi));
                                        if (a1 > a) ugCheck(ug_dvec_copy_via(p.witnessVec(), a, ranks_[q]->witnessVec(), a, a1 - a, p.ctx2()));
                                        if (b > b1) ugCheck(ug_dvec_copy_via(p.witnessVec(), std::max(b1, a), ranks_[q]->witnessVec(), std::max(b1, a), b - std::max(b1, a), p.ctx2()));
                                    }
                                    covered = std::max(covered, w.hi);
                                }
                                if (covered < nVars_) throw std::logic_error("the ranks' witness ranges do not cover the witness");
                                p.witnessGathered();
                            } else p.loadWitnessPart(wtns, wtnsSize, 1);
                            for (int c = k; c < 3; c += R) {
                                p.hpolyChain(c, ug_dvec_device_ptr(full_[c]));
                                told[c] = true;
                                chainDone[c].set_value();
                            }
                        }
                        if (chainFirst) p.witnessMsmBegin();
                        for (int c = 0; c < 3; c++) chainReady[c].get();    // (a chain rank's failure is rethrown here)
                        unsigned long long first = 0, cnt = 0;
                        p.hRange(&first, &cnt, nullptr);
                        for (int c = 0; c < 3 && cnt; c++) ugCheck(ug_dvec_copy(slices_[k].v[c], 0, full_[c], first, cnt));
                        p.hpolyCombine(ug_dvec_device_ptr(slices_[k].v[0]), ug_dvec_device_ptr(slices_[k].v[1]), ug_dvec_device_ptr(slices_[k].v[2]));
                        uint8_t hpart[UG_GROTH16_PARTIALS_SIZE];
                        p.runHMsm(hpart);
                        p.witnessMsmEnd(parts[k].data());
                        memcpy(parts[k].data() + 320, hpart + 320, 64);
                    } catch (...) {
                        errs[k] = std::current_exception();
                        if (!toldSlice) sliceUp[k].set_exception(errs[k]);  // nobody may wait for a slice or a chain that will not come
                        for (int c = k; c < 3; c += R)
                            if (k < 3 && !told[c]) chainDone[c].set_exception(errs[k]);
                        p.witnessMsmAbandon();
                    }
                });
            for (auto& t : th) t.join();
        }
        for (auto& e : errs) if (e) std::rethrow_exception(e);
        for (int k = 1; k < R; k++)
            if (ug_groth16_partials_add(parts[0].data(), parts[k].data()) != PROVER_OK) throw std::runtime_error("partial sum failed");
        ranks_[0]->finish(parts[0].data(), proof, pub);
        msm_ = fft_ = 0;
        for (auto& r : ranks_) { double m = 0, f = 0; r->timings(&m, &f, nullptr); msm_ = std::max(msm_, m); fft_ = std::max(fft_, f); }
        total_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    unsigned long long proofBufferMinSize() const override { return PROOF_MIN_GROTH16; }
    unsigned long long publicBufferMinSize() const override { return publicMin(nPublic_); }
    void timings(double* msm, double* fft, double* total) const override {      // the slowest rank's device time per part
        if (msm) *msm = msm_;
        if (fft) *fft = fft_;
        if (total) *total = total_;
    }
    ug_ctx* ctx() override { return ranks_[0]->ctx(); }
    std::vector<TableGroup> tableGroups() override { return {}; }
    void trimWorkspaces() override { for (auto& r : ranks_) r->trimWorkspaces(); }
    int kernelStats(int which, double* avgMs, unsigned long long* launches, unsigned long long* entries, int reset) override {
        return ranks_[0]->kernelStats(which, avgMs, launches, entries, reset);
    }
    int rankCount() const { return (int)ranks_.size(); }

private:
    struct Slices { ug_dvec* v[3] = {nullptr, nullptr, nullptr}; };
    std::vector<std::unique_ptr<Groth16Prover>> ranks_;
    std::vector<ShardLayout> layouts_;
    ug_dvec* full_[3] = {nullptr, nullptr, nullptr};
    std::vector<Slices> slices_;
    uint32_t nPublic_ = 0, domain_ = 0;
    uint64_t nVars_ = 0;
    double msm_ = 0, fft_ = 0, total_ = 0;
};

// The same for UltraGroth (src/ultra_groth.cpp:401-462): every rank keeps the whole witness and owns slices of the
// witness-indexed sets, the round set, the final set and H. Per proof: the 64-byte parts of the round commitment are added on
// the host, rank 0 closes the round (:173-176), every rank derives the challenge and completes its witness (:33-106), then
// the final round runs as the Groth16 phases do (witness products queued, the H branch on its second stream beside them);
// rank 0 blinds and serialises.
class MultiUltraGrothProver : public ProverBase {
public:
    MultiUltraGrothProver(const void* zkey, unsigned long long zkeySize, const std::vector<int>& devices) {
        const int R = (int)devices.size();
        BinFile f(zkey, zkeySize, "zkey", 1);
        ZkeyHeader h = loadZkeyHeader(f, true);
        if (!h.rIsBn254) throw std::invalid_argument("zkey curve not supported");
        nPublic_ = h.nPublic; domain_ = h.domainSize; nVars_ = h.nVars;
        ranks_.resize(R);
        const bool oneShot = g_oneShotProver;
        std::vector<std::future<void>> jobs;
        for (int k = 0; k < R; k++)
            jobs.push_back(std::async(std::launch::async, [&, k, oneShot] {
                g_oneShotProver = oneShot;
                ranks_[k].reset(new UltraGrothProver(zkey, zkeySize, devices[k], k, R, /*runsChain*/ k < 3));
            }));
        std::exception_ptr failure;
        for (auto& j : jobs) { try { j.get(); } catch (...) { if (!failure) failure = std::current_exception(); } }
        if (failure) { ranks_.clear(); std::rethrow_exception(failure); }
        for (int c = 0; c < 3; c++) ugCheck(ug_dvec_create(ranks_[c % R]->ctx2(), domain_, &full_[c]));      // (the H branch's stream)
        slices_.resize(R);
        for (int k = 0; k < R; k++) {
            unsigned long long first = 0, cnt = 0;
            ranks_[k]->hRange(&first, &cnt, nullptr);
            for (int c = 0; c < 3; c++) ugCheck(ug_dvec_create(ranks_[k]->ctx2(), cnt ? cnt : 1, &slices_[k].v[c]));
        }
    }
    ~MultiUltraGrothProver() override {
        for (auto& s : slices_) for (ug_dvec* v : s.v) ug_dvec_destroy(v);
        for (ug_dvec* v : full_) ug_dvec_destroy(v);
    }
    void prove(const void* wtns, unsigned long long wtnsSize, std::string& proof, std::string& pub) override {
        const auto t0 = std::chrono::steady_clock::now();
        const int R = (int)ranks_.size();
        std::vector<std::exception_ptr> errs(R);
        auto everyRank = [&](const std::function<void(int)>& body) {
            std::vector<std::thread> th;
            for (int k = 0; k < R; k++) th.emplace_back([&, k] { try { body(k); } catch (...) { errs[k] = std::current_exception(); } });
            for (auto& t : th) t.join();
            for (auto& e : errs) if (e) std::rethrow_exception(e);
        };
        // round 1: the commitment to the round witnesses, part by part. Every rank needs the whole witness (the lookup completion
        // rewrites signals anywhere in it): each copies ITS R-th over its own PCIe link, then collects the others from its peers'
        // HBM (peer copies over xGMI; 128 MiB at 2^22: 3.1 ms per rank through one link, ~0.9 ms this way).
        // ULTRAGROTH_WITNESS_GATHER=0: every rank uploads all of it, as round 4 did.
        std::vector<std::array<uint8_t, 64>> cparts(R);
        const char* wg = getenv("ULTRAGROTH_WITNESS_GATHER");
        if (R > 1 && !(wg && wg[0] == '0')) {
            const uint64_t M = nVars_;
            auto slice = [&](int k) { return Range{M * (uint64_t)k / (uint64_t)R, M * (uint64_t)(k + 1) / (uint64_t)R}; };
            everyRank([&](int k) { const Range w = slice(k); ranks_[k]->loadWitnessSlice(wtns, wtnsSize, w.lo, w.hi); });
            everyRank([&](int k) {
                for (int q = 0; q < R; q++) {
                    const Range w = slice(q);
                    if (q != k && w.hi > w.lo) ugCheck(ug_dvec_copy(ranks_[k]->witnessVec(), w.lo, ranks_[q]->witnessVec(), w.lo, w.hi - w.lo));
                }
                ranks_[k]->witnessGathered();
                ranks_[k]->roundCommit(cparts[k].data());
            });
        } else {
            everyRank([&](int k) { ranks_[k]->loadWitness(wtns, wtnsSize); ranks_[k]->roundCommit(cparts[k].data()); });
        }
        for (int k = 1; k < R; k++) if (ug_g1_record_add(cparts[0].data(), cparts[k].data()) != PROVER_OK) throw std::runtime_error("partial sum failed");
        uint8_t commit[64];
        ranks_[0]->roundFinish(cparts[0].data(), commit);
        // final round: as MultiGroth16Prover -- the witness products of every rank queued (witnessMsmBegin), its H branch driven
        // beside them on its second stream; the only edges between the ranks are the three "chain c is complete" events
        std::vector<std::array<uint8_t, UG_GROTH16_PARTIALS_SIZE>> parts(R);
        std::promise<void> chainDone[3];
        std::shared_future<void> chainReady[3];
        for (int c = 0; c < 3; c++) chainReady[c] = chainDone[c].get_future().share();
        everyRank([&](int k) {
            UltraGrothProver& p = *ranks_[k];
            bool told[3] = {false, false, false};
            try {
                p.applyCommitment(commit);
                const bool chainFirst = R >= 5;                         // (see MultiGroth16Prover::prove)
                if (!chainFirst) p.witnessMsmBegin();
                for (int c = k; c < 3; c += R) {
                    p.hpolyChain(c, ug_dvec_device_ptr(full_[c]));
                    told[c] = true;
                    chainDone[c].set_value();
                }
                if (chainFirst) p.witnessMsmBegin();
                for (int c = 0; c < 3; c++) chainReady[c].get();        // (a chain rank's failure is rethrown here)
                unsigned long long first = 0, cnt = 0;
                p.hRange(&first, &cnt, nullptr);
                for (int c = 0; c < 3; c++) ugCheck(ug_dvec_copy(slices_[k].v[c], 0, full_[c], first, cnt));
                p.hpolyCombine(ug_dvec_device_ptr(slices_[k].v[0]), ug_dvec_device_ptr(slices_[k].v[1]), ug_dvec_device_ptr(slices_[k].v[2]));
                uint8_t hpart[UG_GROTH16_PARTIALS_SIZE];
                p.runHMsm(hpart);
                p.witnessMsmEnd(parts[k].data());
                memcpy(parts[k].data() + 320, hpart + 320, 64);
            } catch (...) {
                for (int c = k; c < 3; c += R)                          // nobody may wait for a chain that will not come
                    if (!told[c]) chainDone[c].set_exception(std::current_exception());
                p.witnessMsmAbandon();
                throw;
            }
        });
        for (int k = 1; k < R; k++)
            if (ug_groth16_partials_add(parts[0].data(), parts[k].data()) != PROVER_OK) throw std::runtime_error("partial sum failed");
        ranks_[0]->finish(parts[0].data(), proof, pub);
        msm_ = fft_ = 0;
        for (auto& r : ranks_) { double m = 0, f = 0; r->timings(&m, &f, nullptr); msm_ = std::max(msm_, m); fft_ = std::max(fft_, f); }
        total_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    unsigned long long proofBufferMinSize() const override { return PROOF_MIN_ULTRA; }
    unsigned long long publicBufferMinSize() const override { return publicMin((unsigned long long)nPublic_ - 1); }
    void timings(double* msm, double* fft, double* total) const override {
        if (msm) *msm = msm_;
        if (fft) *fft = fft_;
        if (total) *total = total_;
    }
    ug_ctx* ctx() override { return ranks_[0]->ctx(); }
    std::vector<TableGroup> tableGroups() override { return {}; }
    void trimWorkspaces() override { for (auto& r : ranks_) r->trimWorkspaces(); }

private:
    struct Slices { ug_dvec* v[3] = {nullptr, nullptr, nullptr}; };
    std::vector<std::unique_ptr<UltraGrothProver>> ranks_;
    ug_dvec* full_[3] = {nullptr, nullptr, nullptr};
    std::vector<Slices> slices_;
    uint32_t nPublic_ = 0, domain_ = 0;
    uint64_t nVars_ = 0;
    double msm_ = 0, fft_ = 0, total_ = 0;
};

ProverBase* newUltraGrothProver(const void* zkey, unsigned long long size) {
    const std::vector<int> devices = devicesFromEnv();
    if (devices.size() > 1) return new MultiUltraGrothProver(zkey, size, devices);
    return new UltraGrothProver(zkey, size, devices.size() == 1 ? devices[0] : deviceFromEnv());
}
ProverBase* newGroth16Prover(const void* zkey, unsigned long long size) {
    const std::vector<int> devices = devicesFromEnv();
    if (devices.size() > 1) return new MultiGroth16Prover(zkey, size, devices);
    struct Flag { Flag() { g_deferTables = true; } ~Flag() { g_deferTables = false; } } deferred;
    return new Groth16Prover(zkey, size, devices.size() == 1 ? devices[0] : deviceFromEnv(), 0, 1);
}

// =================================================================================================================
// Resident multi-circuit prover: the GPU form of FullProver's map<circuit, Prover> (src/fullprover.cpp:21-63), without
// its HTTP shell and witness calculator. Several created provers share one device under an HBM budget. What is given
// back when the budget is short, in this order, always from the least recently used circuit first:
//   1. fixed-base window tables (optional per circuit: a circuit without them proves with the classic windows),
//   2. per-proof workspaces (schedules, buckets: re-grown by the circuit's next proof),
//   3. whole circuits (a circuit loaded from a file comes back on demand; one loaded from a buffer must be loaded again).
// Residency is measured, not estimated: bytes in use = free device memory at registry creation - free now.
class Registry {
public:
    Registry(int device, uint64_t budget) : device_(device), budget_(budget) {
        ugCheck(ug_ctx_create(&probe_, device));
        uint64_t total = 0;
        ugCheck(ug_ctx_mem_info(probe_, &baselineFree_, &total));
        if (!budget_ || budget_ > baselineFree_) budget_ = baselineFree_;
    }
    ~Registry() {
        entries_.clear();
        ug_ctx_destroy(probe_);
    }
    static std::string circuitName(const std::string& path) {              // getfilename(), src/fullprover.cpp:14-19
        std::string f = path.substr(path.find_last_of("/\\") + 1);
        return f.substr(0, f.find_last_of('.'));
    }
    void load(const std::string& name, const void* zkey, uint64_t size, const std::string& path) {
        std::lock_guard<std::mutex> lock(mutex_);
        if (name.empty()) throw std::invalid_argument("empty circuit name");
        {
            auto it = entries_.find(name);
            if (it != entries_.end() && it->second->busy) throw std::invalid_argument("circuit is proving, it cannot be replaced now: " + name);
        }
        // the header tells the protocol (1 = groth16, 1337 = ultragroth, src/zkey_utils.cpp:48-50,129-131)
        BinFile f(zkey, size, "zkey", 1);
        if (f.sectionSize(1) < 4) throw std::range_error("Invalid section size");
        uint32_t protocol;
        memcpy(&protocol, f.sectionData(1), 4);
        // The replacement is built BESIDE the resident circuit of that name, which goes only once the new one exists: a bad or
        // oversized zkey leaves the working circuit as it was. (Both must fit the device for that moment; a caller that
        // replaces a circuit too large for that evicts it first.)
        const uint64_t before = used();
        std::unique_ptr<Entry> e(new Entry());
        e->name = name; e->path = path; e->ultra = protocol == 1337;
        g_registryCreate = true;
        try {
            if (e->ultra) e->prover.reset(new UltraGrothProver(zkey, size, device_));
            else e->prover.reset(new Groth16Prover(zkey, size, device_, 0, 1));
        } catch (...) { g_registryCreate = false; throw; }
        g_registryCreate = false;
        e->coreBytes = used() - before;
        e->lastUsed = ++tick_;
        if (e->coreBytes > budget_) throw std::runtime_error("circuit " + name + " does not fit the HBM budget");
        Entry* raw = e.get();
        std::unique_ptr<Entry> old;
        {
            auto it = entries_.find(name);
            if (it != entries_.end()) { old = std::move(it->second); entries_.erase(it); }
        }
        old.reset();                                   // the circuit this one replaces gives its memory back first
        entries_[name] = std::move(e);
        evictedPaths_.erase(name);                     // a stale path of an earlier, evicted circuit of that name must not come back
        if (!makeRoom(0, raw)) {
            entries_.erase(name);
            throw std::runtime_error("circuit " + name + " does not fit the HBM budget");
        }
        growTables();
    }
    void loadFile(const std::string& path) {
        FileMap m(path);
        load(circuitName(path), m.data(), m.size(), path);
    }
    void prove(const std::string& name, const void* wtns, uint64_t wtnsSize, std::string& proof, std::string& pub,
               unsigned long long* proofSize, unsigned long long* publicSize) {
        Entry* e = nullptr;
        {
            std::unique_lock<std::mutex> lock(mutex_);
            auto it = entries_.find(name);
            if (it == entries_.end()) {
                auto ev = evictedPaths_.find(name);
                if (ev == evictedPaths_.end()) throw std::invalid_argument("circuit not loaded: " + name);
                const std::string path = ev->second;          // evicted under memory pressure: its file brings it back
                evictedPaths_.erase(ev);
                lock.unlock();
                loadFile(path);
                lock.lock();
                it = entries_.find(name);
                if (it == entries_.end()) throw std::runtime_error("circuit could not be reloaded: " + name);
            }
            e = it->second.get();
            e->lastUsed = ++tick_;
            checkBufferSizes(e->prover->proofBufferMinSize(), proofSize, e->prover->publicBufferMinSize(), publicSize, "Minimum");
            // room for the proof's workspaces (known after the circuit's first proof; a guess from its size before); when even
            // that cannot be had the proof is still tried -- its own allocations decide, and a failure leaves nothing queued
            (void)makeRoom(e->workBytes ? 0 : e->coreBytes / 2, e);
            e->busy++;                                                     // from here on nobody may take it away
        }
        std::exception_ptr failure;
        {
            // one proof on the device at a time (deviceMutex_), as fullprover's `busy`; the witness of a waiting call may be
            // copied to the device meanwhile (ProverBase::proveTurn)
            try {
                uint64_t before = 0;
                e->prover->proveTurn(wtns, wtnsSize, proof, pub, &deviceMutex_, [&](bool begin) {
                    if (begin) { before = used(); return; }
                    const uint64_t after = used();
                    if (after > before) e->workBytes += after - before;      // (atomics: info() reads them under mutex_, which this
                    e->proofs++;                                             //  callback -- it runs inside the prover's turn -- must not take)
                    proofsTotal_++;
                });
            } catch (...) { failure = std::current_exception(); }
        }
        std::lock_guard<std::mutex> lock(mutex_);
        e->busy--;
        if (failure) std::rethrow_exception(failure);
        makeRoom(0, e);
        growTables();
    }
    void evict(const std::string& name) {
        std::lock_guard<std::mutex> lock(mutex_);
        auto it = entries_.find(name);
        if (it != entries_.end() && it->second->busy) throw std::invalid_argument("circuit is proving, it cannot be evicted now: " + name);
        const size_t resident = entries_.erase(name), remembered = evictedPaths_.erase(name);      // both, unconditionally
        if (!resident && !remembered) throw std::invalid_argument("circuit not loaded: " + name);
    }
    // name empty: totals. state: 0 not loaded, 1 resident without tables, 2 resident with tables, 3 evicted (reloadable)
    void info(const std::string& name, unsigned long long* bytes, int* state, unsigned long long* proofs) {
        std::lock_guard<std::mutex> lock(mutex_);
        if (name.empty()) {
            if (bytes) *bytes = used();
            if (state) *state = (int)entries_.size();
            if (proofs) *proofs = proofsTotal_;            // (all proofs made here, also those of circuits evicted since)
            return;
        }
        auto it = entries_.find(name);
        if (it == entries_.end()) {
            if (bytes) *bytes = 0;
            if (proofs) *proofs = 0;
            if (state) *state = evictedPaths_.count(name) ? 3 : 0;
            return;
        }
        Entry& e = *it->second;
        if (bytes) *bytes = e.coreBytes + e.prover->tableBytes + e.workBytes;
        if (state) *state = e.prover->tableBytes ? 2 : 1;
        if (proofs) *proofs = e.proofs;
    }

private:
    struct Entry {
        std::string name, path;
        bool ultra = false;
        std::unique_ptr<ProverBase> prover;
        uint64_t coreBytes = 0, lastUsed = 0;
        std::atomic<uint64_t> workBytes{0}, proofs{0};       // written by the proof's bracket callback, outside mutex_
        uint64_t tablesDroppedAt = 0;      // tick at which its tables were taken away (it gets them back only after it was used again)
        int busy = 0;                      // proofs in flight on it: a busy circuit is never trimmed, evicted or replaced
    };
    uint64_t used() {
        uint64_t freeB = 0, total = 0;
        ugCheck(ug_ctx_mem_info(probe_, &freeB, &total));
        return baselineFree_ > freeB ? baselineFree_ - freeB : 0;
    }
    Entry* lru(const Entry* keep, bool (*has)(const Entry&)) {
        Entry* best = nullptr;
        for (auto& kv : entries_) {
            Entry* e = kv.second.get();
            if (e == keep || e->busy || !has(*e)) continue;
            if (!best || e->lastUsed < best->lastUsed) best = e;
        }
        return best;
    }
    // frees memory until used() + extra <= budget; `keep` is never evicted as a whole (its tables and workspaces may go last)
    bool makeRoom(uint64_t extra, Entry* keep) {
        while (used() + extra > budget_) {
            Entry* e = lru(keep, [](const Entry& x) { return x.prover->tableBytes != 0; });
            if (e) { e->prover->dropTables(); e->tablesDroppedAt = ++tick_; continue; }
            e = lru(keep, [](const Entry& x) { return x.workBytes != 0; });
            if (e) { e->prover->trimWorkspaces(); e->workBytes = 0; continue; }
            e = lru(keep, [](const Entry&) { return true; });
            if (e) {
                if (!e->path.empty()) evictedPaths_[e->name] = e->path;
                entries_.erase(e->name);
                continue;
            }
            if (keep && keep->prover->tableBytes) { keep->prover->dropTables(); keep->tablesDroppedAt = ++tick_; continue; }
            if (keep && keep->workBytes) { keep->prover->trimWorkspaces(); keep->workBytes = 0; continue; }
            return false;
        }
        return true;
    }
    // tables for the most recently used circuits that lack them, while they fit the budget (with room for a proof's workspaces)
    void growTables() {
        std::vector<Entry*> order;
        for (auto& kv : entries_) if (!kv.second->prover->tableBytes && !kv.second->busy) order.push_back(kv.second.get());
        std::sort(order.begin(), order.end(), [](Entry* a, Entry* b) { return a->lastUsed > b->lastUsed; });
        for (Entry* e : order) {
            if (e->tablesDroppedAt && e->lastUsed < e->tablesDroppedAt) continue;      // no thrashing: not before its next use
            const uint64_t need = e->prover->tablesWouldTake();
            if (!need) continue;
            const uint64_t now = used(), reserve = e->workBytes ? 0 : e->coreBytes / 2;
            if (now + need + reserve > budget_ - budget_ / 10) continue;               // and only into a comfortable margin
            e->prover->buildTables(budget_ - now - reserve);
        }
    }
    int device_;
    uint64_t budget_, baselineFree_ = 0, tick_ = 0;
    ug_ctx* probe_ = nullptr;
    std::mutex mutex_, deviceMutex_;
    std::atomic<uint64_t> proofsTotal_{0};
    std::map<std::string, std::unique_ptr<Entry>> entries_;
    std::map<std::string, std::string> evictedPaths_;
};

// =================================================================================================================
// extern "C" surface. Error mapping as in src/prover.cpp:556-576.
#define API_TRY try {
#define API_CATCH                                                                                              \
    } catch (InvalidWitnessLengthException& e) { copyError(error_msg, error_msg_maxsize, e.what()); return PROVER_INVALID_WITNESS_LENGTH; } \
    catch (ShortBufferException& e) { copyError(error_msg, error_msg_maxsize, e.what()); return PROVER_ERROR_SHORT_BUFFER; }              \
    catch (std::exception& e) { copyError(error_msg, error_msg_maxsize, e.what()); return PROVER_ERROR; }                                 \
    catch (...) { copyError(error_msg, error_msg_maxsize, "unknown error"); return PROVER_ERROR; }                                        \
    return PROVER_OK;

namespace {
int proveImpl(void* prover_object, const void* wtns_buffer, unsigned long long wtns_size, char* proof_buffer,
              unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size, char* error_msg,
              unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (wtns_buffer == NULL) throw std::invalid_argument("Null witness buffer");
    if (proof_buffer == NULL) throw std::invalid_argument("Null proof buffer");
    if (proof_size == NULL) throw std::invalid_argument("Null proof size");
    if (public_buffer == NULL) throw std::invalid_argument("Null public buffer");
    if (public_size == NULL) throw std::invalid_argument("Null public size");
    ProverBase* prover = static_cast<ProverBase*>(prover_object);
    checkBufferSizes(prover->proofBufferMinSize(), proof_size, prover->publicBufferMinSize(), public_size, "Minimum");
    std::string stringProof, stringPublic;
    {
        // the reference's prover keeps no per-proof state, so callers may prove from several threads on one object;
        // here the object owns the device buffers of a proof: concurrent calls take turns on the device (and the witness
        // of the next call is copied there while they wait)
        prover->proveTurn(wtns_buffer, wtns_size, stringProof, stringPublic);
    }
    checkBufferSizes(stringProof.length(), proof_size, stringPublic.length(), public_size, "Required");
    std::strncpy(proof_buffer, stringProof.c_str(), *proof_size);
    std::strncpy(public_buffer, stringPublic.c_str(), *public_size);
    API_CATCH
}
int publicSizeImpl(const void* zkey, unsigned long long size, bool ultra, unsigned long long* public_size, char* error_msg,
                   unsigned long long error_msg_maxsize) {
    API_TRY
    BinFile f(zkey, size, "zkey", 1);
    ZkeyHeader h = loadZkeyHeader(f, ultra);
    *public_size = publicMin(h.nPublic);
    API_CATCH
}
}  // namespace

extern "C" {

int groth16_public_size_for_zkey_buf(const void* zkey_buffer, unsigned long long zkey_size, unsigned long long* public_size,
                                     char* error_msg, unsigned long long error_msg_maxsize) {
    return publicSizeImpl(zkey_buffer, zkey_size, false, public_size, error_msg, error_msg_maxsize);
}
int ultra_groth_public_size_for_zkey_buf(const void* zkey_buffer, unsigned long long zkey_size, unsigned long long* public_size,
                                         char* error_msg, unsigned long long error_msg_maxsize) {
    return publicSizeImpl(zkey_buffer, zkey_size, true, public_size, error_msg, error_msg_maxsize);
}
int groth16_public_size_for_zkey_file(const char* zkey_fname, unsigned long long* public_size, char* error_msg,
                                      unsigned long long error_msg_maxsize) {
    API_TRY
    FileMap m(zkey_fname);
    BinFile f(m.data(), m.size(), "zkey", 1);
    *public_size = publicMin(loadZkeyHeader(f, false).nPublic);
    API_CATCH
}
int ultra_groth_public_size_for_zkey_file(const char* zkey_fname, unsigned long long* public_size, char* error_msg,
                                          unsigned long long error_msg_maxsize) {
    API_TRY
    FileMap m(zkey_fname);
    BinFile f(m.data(), m.size(), "zkey", 1);
    *public_size = publicMin(loadZkeyHeader(f, true).nPublic);
    API_CATCH
}

void groth16_proof_size(unsigned long long* proof_size) { *proof_size = PROOF_MIN_GROTH16; }
void ultra_groth_proof_size(unsigned long long* proof_size) { *proof_size = PROOF_MIN_ULTRA; }

int groth16_prover_create(void** prover_object, const void* zkey_buffer, unsigned long long zkey_size, char* error_msg,
                          unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_buffer == NULL) throw std::invalid_argument("Null zkey buffer");
    *prover_object = newGroth16Prover(zkey_buffer, zkey_size);
    API_CATCH
}
int ultra_groth_prover_create(void** prover_object, const void* zkey_buffer, unsigned long long zkey_size, char* error_msg,
                              unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_buffer == NULL) throw std::invalid_argument("Null zkey buffer");
    *prover_object = newUltraGrothProver(zkey_buffer, zkey_size);
    API_CATCH
}
int groth16_prover_create_zkey_file(void** prover_object, const char* zkey_file_path, char* error_msg,
                                    unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    FileMap m(zkey_file_path);
    *prover_object = newGroth16Prover(m.data(), m.size());
    API_CATCH
}
int ultra_groth_prover_create_zkey_file(void** prover_object, const char* zkey_file_path, char* error_msg,
                                        unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    FileMap m(zkey_file_path);
    *prover_object = newUltraGrothProver(m.data(), m.size());
    API_CATCH
}

int groth16_prover_prove(void* prover_object, const void* wtns_buffer, unsigned long long wtns_size, char* proof_buffer,
                         unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size, char* error_msg,
                         unsigned long long error_msg_maxsize) {
    return proveImpl(prover_object, wtns_buffer, wtns_size, proof_buffer, proof_size, public_buffer, public_size,
                                    error_msg, error_msg_maxsize);
}
int ultra_groth_prover_prove(void* prover_object, const void* wtns_buffer, unsigned long long wtns_size, char* proof_buffer,
                             unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size, char* error_msg,
                             unsigned long long error_msg_maxsize) {
    return proveImpl(prover_object, wtns_buffer, wtns_size, proof_buffer, proof_size, public_buffer,
                                       public_size, error_msg, error_msg_maxsize);
}

void groth16_prover_destroy(void* prover_object) { delete static_cast<ProverBase*>(prover_object); }
void ultra_groth_prover_destroy(void* prover_object) { delete static_cast<ProverBase*>(prover_object); }

int groth16_prover(const void* zkey_buffer, unsigned long long zkey_size, const void* wtns_buffer, unsigned long long wtns_size,
                   char* proof_buffer, unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size,
                   char* error_msg, unsigned long long error_msg_maxsize) {
    void* prover = NULL;
    g_oneShotProver = true;
    int error = groth16_prover_create(&prover, zkey_buffer, zkey_size, error_msg, error_msg_maxsize);
    g_oneShotProver = false;
    if (error != PROVER_OK) return error;
    error = groth16_prover_prove(prover, wtns_buffer, wtns_size, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                                 error_msg_maxsize);
    groth16_prover_destroy(prover);
    return error;
}
int ultra_groth_prover(const void* zkey_buffer, unsigned long long zkey_size, const void* wtns_buffer, unsigned long long wtns_size,
                       char* proof_buffer, unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size,
                       char* error_msg, unsigned long long error_msg_maxsize) {
    void* prover = NULL;
    g_oneShotProver = true;
    int error = ultra_groth_prover_create(&prover, zkey_buffer, zkey_size, error_msg, error_msg_maxsize);
    g_oneShotProver = false;
    if (error != PROVER_OK) return error;
    error = ultra_groth_prover_prove(prover, wtns_buffer, wtns_size, proof_buffer, proof_size, public_buffer, public_size,
                                     error_msg, error_msg_maxsize);
    ultra_groth_prover_destroy(prover);
    return error;
}

int groth16_prover_zkey_file(const char* zkey_file_path, const void* wtns_buffer, unsigned long long wtns_size, char* proof_buffer,
                             unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size, char* error_msg,
                             unsigned long long error_msg_maxsize) {
    std::unique_ptr<FileMap> m;
    try { m.reset(new FileMap(zkey_file_path)); }
    catch (std::exception& e) { copyError(error_msg, error_msg_maxsize, e.what()); return PROVER_ERROR; }
    return groth16_prover(m->data(), m->size(), wtns_buffer, wtns_size, proof_buffer, proof_size, public_buffer, public_size,
                          error_msg, error_msg_maxsize);
}
int ultra_groth_prover_zkey_file(const char* zkey_file_path, const void* wtns_buffer, unsigned long long wtns_size,
                                 char* proof_buffer, unsigned long long* proof_size, char* public_buffer,
                                 unsigned long long* public_size, char* error_msg, unsigned long long error_msg_maxsize) {
    std::unique_ptr<FileMap> m;
    try { m.reset(new FileMap(zkey_file_path)); }
    catch (std::exception& e) { copyError(error_msg, error_msg_maxsize, e.what()); return PROVER_ERROR; }
    return ultra_groth_prover(m->data(), m->size(), wtns_buffer, wtns_size, proof_buffer, proof_size, public_buffer, public_size,
                              error_msg, error_msg_maxsize);
}

// ---- additions ----------------------------------------------------------------------------------------------
int ug_test_set_blinding(const void* bytes, unsigned long long n) { return setRandomOverride(bytes, (size_t)n) ? PROVER_OK : PROVER_ERROR; }

int ug_registry_create(void** registry, int device, unsigned long long hbm_budget_bytes, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (registry == NULL) throw std::invalid_argument("Null registry pointer");
    *registry = new Registry(device, hbm_budget_bytes);
    API_CATCH
}
int ug_registry_load(void* registry, const char* circuit, const void* zkey_buffer, unsigned long long zkey_size, char* error_msg,
                     unsigned long long error_msg_maxsize) {
    API_TRY
    if (registry == NULL) throw std::invalid_argument("Null registry object");
    if (circuit == NULL) throw std::invalid_argument("Null circuit name");
    if (zkey_buffer == NULL) throw std::invalid_argument("Null zkey buffer");
    static_cast<Registry*>(registry)->load(circuit, zkey_buffer, zkey_size, "");
    API_CATCH
}
int ug_registry_load_file(void* registry, const char* zkey_file_path, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (registry == NULL) throw std::invalid_argument("Null registry object");
    if (zkey_file_path == NULL) throw std::invalid_argument("Null zkey path");
    static_cast<Registry*>(registry)->loadFile(zkey_file_path);
    API_CATCH
}
int ug_registry_prove(void* registry, const char* circuit, const void* wtns_buffer, unsigned long long wtns_size, char* proof_buffer,
                      unsigned long long* proof_size, char* public_buffer, unsigned long long* public_size, char* error_msg,
                      unsigned long long error_msg_maxsize) {
    API_TRY
    if (registry == NULL) throw std::invalid_argument("Null registry object");
    if (circuit == NULL) throw std::invalid_argument("Null circuit name");
    if (wtns_buffer == NULL) throw std::invalid_argument("Null witness buffer");
    if (proof_buffer == NULL) throw std::invalid_argument("Null proof buffer");
    if (proof_size == NULL) throw std::invalid_argument("Null proof size");
    if (public_buffer == NULL) throw std::invalid_argument("Null public buffer");
    if (public_size == NULL) throw std::invalid_argument("Null public size");
    std::string stringProof, stringPublic;
    static_cast<Registry*>(registry)->prove(circuit, wtns_buffer, wtns_size, stringProof, stringPublic, proof_size, public_size);
    checkBufferSizes(stringProof.length(), proof_size, stringPublic.length(), public_size, "Required");
    std::strncpy(proof_buffer, stringProof.c_str(), *proof_size);
    std::strncpy(public_buffer, stringPublic.c_str(), *public_size);
    API_CATCH
}
int ug_registry_evict(void* registry, const char* circuit, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (registry == NULL) throw std::invalid_argument("Null registry object");
    if (circuit == NULL) throw std::invalid_argument("Null circuit name");
    static_cast<Registry*>(registry)->evict(circuit);
    API_CATCH
}
int ug_registry_info(void* registry, const char* circuit, unsigned long long* resident_bytes, int* state, unsigned long long* proofs) {
    if (registry == NULL) return PROVER_ERROR;
    try { static_cast<Registry*>(registry)->info(circuit ? circuit : "", resident_bytes, state, proofs); }
    catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
void ug_registry_destroy(void* registry) { delete static_cast<Registry*>(registry); }

int ug_prover_last_timings(void* prover_object, double* msm_ms, double* fft_ms, double* total_ms) {
    if (!prover_object) return PROVER_ERROR;
    static_cast<ProverBase*>(prover_object)->timings(msm_ms, fft_ms, total_ms);
    return PROVER_OK;
}
int ug_prover_kernel_stats(void* prover_object, int which, double* launch_ms_avg, unsigned long long* launches,
                           unsigned long long* units, int reset) {
    if (!prover_object) return PROVER_ERROR;
    int rc = static_cast<ProverBase*>(prover_object)->kernelStats(which, launch_ms_avg, launches, units, reset);
    return rc == UG_OK ? PROVER_OK : PROVER_ERROR;
}
int ug_prover_tables_ready(void* prover_object, int wait) {
    if (!prover_object) return -1;
    try { return static_cast<ProverBase*>(prover_object)->tablesReady(wait != 0) ? 1 : 0; }
    catch (...) { return -1; }
}
int ug_prover_last_upload_ms(void* prover_object, double* upload_ms) {
    if (!prover_object || !upload_ms) return PROVER_ERROR;
    *upload_ms = static_cast<ProverBase*>(prover_object)->uploadMs;
    return PROVER_OK;
}

int ug_groth16_prover_create_sharded(void** prover_object, const void* zkey_buffer, unsigned long long zkey_size, int device,
                                     int shard_rank, int shard_count, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_buffer == NULL) throw std::invalid_argument("Null zkey buffer");
    *prover_object = static_cast<ProverBase*>(new Groth16Prover(zkey_buffer, zkey_size, device, shard_rank, shard_count));
    API_CATCH
}
int ug_groth16_prover_create_sharded_range(void** prover_object, const void* zkey_buffer, unsigned long long zkey_size, int device,
                                           int shard_rank, int shard_count, unsigned long long witness_first,
                                           unsigned long long witness_end, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_buffer == NULL) throw std::invalid_argument("Null zkey buffer");
    Range wr{witness_first, witness_end};
    *prover_object = static_cast<ProverBase*>(new Groth16Prover(zkey_buffer, zkey_size, device, shard_rank, shard_count, &wr));
    API_CATCH
}
int ug_groth16_shard_ranges(unsigned long long n_vars, unsigned long long n_public, unsigned long long domain_size, int shard_rank,
                            int shard_count, const unsigned long long* witness_range, unsigned long long out[6]) {
    try {
        Range wr{0, 0};
        if (witness_range) { wr.lo = witness_range[0]; wr.hi = witness_range[1]; }
        Groth16Prover::Ranges r = Groth16Prover::shardRanges(n_vars, n_public, domain_size, shard_rank, shard_count, witness_range ? &wr : nullptr);
        out[0] = r.w.lo; out[1] = r.w.hi; out[2] = r.c.lo; out[3] = r.c.hi; out[4] = r.h.lo; out[5] = r.h.hi;
    } catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
int ug_groth16_shard_layout(unsigned long long n_vars, unsigned long long n_public, unsigned long long domain_size, int shard_rank,
                            int shard_count, int point_ranges, unsigned long long hbm_bytes, unsigned long long out[12]) {
    try {
        if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count || !out || point_ranges < 0) return PROVER_ERROR;
        const int P = pointRangeGroups(n_vars, shard_count, point_ranges, hbm_bytes);
        const ShardLayout L = shardLayouts(n_vars, domain_size, shard_count, P)[shard_rank];
        Groth16Prover::Ranges r = Groth16Prover::shardRanges(n_vars, n_public, domain_size, shard_rank, shard_count, &L.w, &L.h);
        out[0] = r.w.lo; out[1] = r.w.hi; out[2] = r.c.lo; out[3] = r.c.hi; out[4] = r.h.lo; out[5] = r.h.hi;
        out[6] = (unsigned long long)L.qLog; out[7] = L.r0; out[8] = L.cnt; out[9] = L.sp.lo; out[10] = L.sp.hi; out[11] = L.chains;
    } catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
int ug_groth16_prover_create_sharded_layout(void** prover_object, const void* zkey_header, unsigned long long zkey_header_size,
                                            const void* coefs, unsigned long long n_coefs, const void* points_a, const void* points_b1,
                                            const void* points_b2, const void* points_c, const void* points_h,
                                            const unsigned long long slice_bytes[5], int device, int shard_rank, int shard_count,
                                            const unsigned long long layout[12], char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_header == NULL) throw std::invalid_argument("Null zkey buffer");
    if (slice_bytes == NULL) throw std::invalid_argument("Null slice sizes");
    if (layout == NULL) throw std::invalid_argument("Null layout");
    Groth16Prover::Sources src;
    src.sliceBytes = slice_bytes;
    src.coefs = static_cast<const uint8_t*>(coefs); src.nCoefs = coefs ? n_coefs : 0; src.haveCoefs = coefs != NULL;
    src.pA = static_cast<const uint8_t*>(points_a); src.pB1 = static_cast<const uint8_t*>(points_b1);
    src.pB2 = static_cast<const uint8_t*>(points_b2); src.pC = static_cast<const uint8_t*>(points_c);
    src.pH = static_cast<const uint8_t*>(points_h);
    ShardLayout L;
    L.w = Range{layout[0], layout[1]}; L.h = Range{layout[4], layout[5]};
    if (layout[6] > 8 || layout[7] > 255 || layout[8] > 256) throw std::invalid_argument("invalid bucket classes in the layout");
    L.qLog = (int)layout[6]; L.r0 = (uint32_t)layout[7]; L.cnt = (uint32_t)layout[8]; L.sp = Range{layout[9], layout[10]};
    L.chains = (unsigned)layout[11];
    *prover_object = static_cast<ProverBase*>(new Groth16Prover(zkey_header, zkey_header_size, src, device, shard_rank, shard_count, nullptr, &L));
    API_CATCH
}
int ug_groth16_balanced_witness_range(unsigned long long n_vars, int shard_rank, int shard_count, unsigned long long out[2]) {
    try {
        if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count || !out) return PROVER_ERROR;
        const std::vector<Range> r = balancedWitnessRanges(n_vars, shard_count);
        out[0] = r[shard_rank].lo; out[1] = r[shard_rank].hi;
    } catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
int ug_groth16_prover_create_sharded_slices(void** prover_object, const void* zkey_header, unsigned long long zkey_header_size,
                                            const void* coefs, unsigned long long n_coefs, const void* points_a, const void* points_b1,
                                            const void* points_b2, const void* points_c, const void* points_h,
                                            const unsigned long long slice_bytes[5], int device, int shard_rank,
                                            int shard_count, const unsigned long long* witness_range, char* error_msg,
                                            unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_header == NULL) throw std::invalid_argument("Null zkey buffer");
    if (slice_bytes == NULL) throw std::invalid_argument("Null slice sizes");
    Groth16Prover::Sources src;
    src.sliceBytes = slice_bytes;
    src.coefs = static_cast<const uint8_t*>(coefs); src.nCoefs = coefs ? n_coefs : 0; src.haveCoefs = coefs != NULL;
    src.pA = static_cast<const uint8_t*>(points_a); src.pB1 = static_cast<const uint8_t*>(points_b1);
    src.pB2 = static_cast<const uint8_t*>(points_b2); src.pC = static_cast<const uint8_t*>(points_c);
    src.pH = static_cast<const uint8_t*>(points_h);
    Range wr{0, 0};
    if (witness_range) { wr.lo = witness_range[0]; wr.hi = witness_range[1]; }
    *prover_object = static_cast<ProverBase*>(new Groth16Prover(zkey_header, zkey_header_size, src, device, shard_rank, shard_count,
                                                                witness_range ? &wr : nullptr));
    API_CATCH
}
int ug_groth16_prover_load_witness_part(void* prover_object, const void* wtns_buffer, unsigned long long wtns_size, int part,
                                        char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (wtns_buffer == NULL) throw std::invalid_argument("Null witness buffer");
    static_cast<ProverBase*>(prover_object)->loadWitnessPart(wtns_buffer, wtns_size, part);
    API_CATCH
}
int ug_groth16_prover_load_witness(void* prover_object, const void* wtns_buffer, unsigned long long wtns_size, char* error_msg,
                                   unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (wtns_buffer == NULL) throw std::invalid_argument("Null witness buffer");
    static_cast<ProverBase*>(prover_object)->loadWitness(wtns_buffer, wtns_size);
    API_CATCH
}
int ug_groth16_prover_prove_resident(void* prover_object, char* proof_buffer, unsigned long long* proof_size, char* public_buffer,
                                     unsigned long long* public_size, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (!proof_buffer || !proof_size || !public_buffer || !public_size) throw std::invalid_argument("Null buffer");
    ProverBase* prover = static_cast<ProverBase*>(prover_object);
    checkBufferSizes(prover->proofBufferMinSize(), proof_size, prover->publicBufferMinSize(), public_size, "Minimum");
    std::string stringProof, stringPublic;
    prover->proveResident(stringProof, stringPublic);
    checkBufferSizes(stringProof.length(), proof_size, stringPublic.length(), public_size, "Required");
    std::strncpy(proof_buffer, stringProof.c_str(), *proof_size);
    std::strncpy(public_buffer, stringPublic.c_str(), *public_size);
    API_CATCH
}
int ug_groth16_prover_run(void* prover_object, void* partials_out, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (partials_out == NULL) throw std::invalid_argument("Null partials buffer");
    static_cast<ProverBase*>(prover_object)->run(static_cast<uint8_t*>(partials_out));
    API_CATCH
}
int ug_groth16_prover_run_witness_msm(void* prover_object, void* partials_out, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || partials_out == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->runWitnessMsm(static_cast<uint8_t*>(partials_out));
    API_CATCH
}
int ug_groth16_prover_witness_msm_begin(void* prover_object, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->witnessMsmBegin();
    API_CATCH
}
int ug_groth16_prover_witness_msm_end(void* prover_object, void* partials_out, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || partials_out == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->witnessMsmEnd(static_cast<uint8_t*>(partials_out));
    API_CATCH
}
int ug_groth16_prover_run_h_msm(void* prover_object, void* partials_out, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || partials_out == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->runHMsm(static_cast<uint8_t*>(partials_out));
    API_CATCH
}
int ug_groth16_prover_hpoly_chain(void* prover_object, int which, void* device_out, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || device_out == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->hpolyChain(which, device_out);
    API_CATCH
}
int ug_groth16_prover_hpoly_combine(void* prover_object, void* device_a, void* device_b, void* device_c, char* error_msg,
                                    unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || !device_a || !device_b || !device_c) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->hpolyCombine(device_a, device_b, device_c);
    API_CATCH
}
int ug_groth16_prover_h_range(void* prover_object, unsigned long long* first, unsigned long long* count, unsigned long long* domain_size) {
    if (prover_object == NULL) return PROVER_ERROR;
    static_cast<ProverBase*>(prover_object)->hRange(first, count, domain_size);
    return PROVER_OK;
}
int ug_ultra_groth_prover_create_sharded(void** prover_object, const void* zkey_buffer, unsigned long long zkey_size, int device,
                                         int shard_rank, int shard_count, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_buffer == NULL) throw std::invalid_argument("Null zkey buffer");
    *prover_object = static_cast<ProverBase*>(new UltraGrothProver(zkey_buffer, zkey_size, device, shard_rank, shard_count));
    API_CATCH
}
int ug_ultra_groth_shard_ranges(unsigned long long n_vars, unsigned long long domain_size, unsigned long long n_round_indexes,
                                unsigned long long n_final_indexes, int shard_rank, int shard_count, unsigned long long out[8]) {
    try {
        if (!out) return PROVER_ERROR;
        const UltraGrothProver::Ranges r = UltraGrothProver::shardRanges(n_vars, domain_size, n_round_indexes, n_final_indexes, shard_rank, shard_count);
        out[0] = r.w.lo; out[1] = r.w.hi; out[2] = r.c1.lo; out[3] = r.c1.hi; out[4] = r.c2.lo; out[5] = r.c2.hi; out[6] = r.h.lo; out[7] = r.h.hi;
    } catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
int ug_ultra_groth_prover_create_sharded_slices(void** prover_object, const void* zkey_header, unsigned long long zkey_header_size,
                                                const void* coefs, unsigned long long n_coefs, const void* points_a, const void* points_b1,
                                                const void* points_b2, const void* points_round_c, const void* points_final_c,
                                                const void* points_h, const void* round_indexes, const void* final_round_indexes,
                                                const unsigned long long slice_bytes[8], int device, int shard_rank, int shard_count,
                                                char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (zkey_header == NULL) throw std::invalid_argument("Null zkey buffer");
    if (slice_bytes == NULL) throw std::invalid_argument("Null slice sizes");
    UltraGrothProver::Sources src;
    src.sliceBytes = slice_bytes;
    src.coefs = static_cast<const uint8_t*>(coefs); src.nCoefs = coefs ? n_coefs : 0; src.haveCoefs = coefs != NULL;
    src.pA = static_cast<const uint8_t*>(points_a); src.pB1 = static_cast<const uint8_t*>(points_b1);
    src.pB2 = static_cast<const uint8_t*>(points_b2); src.pRoundC = static_cast<const uint8_t*>(points_round_c);
    src.pFinalC = static_cast<const uint8_t*>(points_final_c); src.pH = static_cast<const uint8_t*>(points_h);
    src.idx1 = static_cast<const uint8_t*>(round_indexes); src.idx2 = static_cast<const uint8_t*>(final_round_indexes);
    *prover_object = static_cast<ProverBase*>(new UltraGrothProver(zkey_header, zkey_header_size, src, device, shard_rank, shard_count));
    API_CATCH
}
int ug_ultra_groth_prover_round_commit(void* prover_object, void* commit_part_out, char* error_msg, unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || commit_part_out == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->roundCommit(static_cast<uint8_t*>(commit_part_out));
    API_CATCH
}
int ug_ultra_groth_prover_round_finish(void* prover_object, const void* commit_sum, void* commitment_out, char* error_msg,
                                       unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || commit_sum == NULL || commitment_out == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->roundFinish(static_cast<const uint8_t*>(commit_sum), static_cast<uint8_t*>(commitment_out));
    API_CATCH
}
int ug_ultra_groth_prover_apply_commitment(void* prover_object, const void* commitment, char* error_msg,
                                           unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL || commitment == NULL) throw std::invalid_argument("Null argument");
    static_cast<ProverBase*>(prover_object)->applyCommitment(static_cast<const uint8_t*>(commitment));
    API_CATCH
}
int ug_g1_record_add(void* acc, const void* other) {
    try {
        uint8_t* a = static_cast<uint8_t*>(acc);
        g1ToRecord(a, xyzz_add(g1FromRecord(a), g1FromRecord(static_cast<const uint8_t*>(other))));
    } catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
int ug_groth16_partials_add(void* partials_acc, const void* partials_other) {
    try {
        uint8_t* a = static_cast<uint8_t*>(partials_acc);
        const uint8_t* b = static_cast<const uint8_t*>(partials_other);
        const int g1off[4] = {0, 64, 256, 320};
        for (int k = 0; k < 4; k++) g1ToRecord(a + g1off[k], xyzz_add(g1FromRecord(a + g1off[k]), g1FromRecord(b + g1off[k])));
        g2ToRecord(a + 128, xyzz_add(g2FromRecord(a + 128), g2FromRecord(b + 128)));
    } catch (...) { return PROVER_ERROR; }
    return PROVER_OK;
}
int ug_groth16_prover_finish(void* prover_object, const void* partials_sum, char* proof_buffer, unsigned long long* proof_size,
                             char* public_buffer, unsigned long long* public_size, char* error_msg,
                             unsigned long long error_msg_maxsize) {
    API_TRY
    if (prover_object == NULL) throw std::invalid_argument("Null prover object");
    if (!partials_sum || !proof_buffer || !proof_size || !public_buffer || !public_size) throw std::invalid_argument("Null buffer");
    ProverBase* prover = static_cast<ProverBase*>(prover_object);
    checkBufferSizes(prover->proofBufferMinSize(), proof_size, prover->publicBufferMinSize(), public_size, "Minimum");
    std::string stringProof, stringPublic;
    prover->finish(static_cast<const uint8_t*>(partials_sum), stringProof, stringPublic);
    checkBufferSizes(stringProof.length(), proof_size, stringPublic.length(), public_size, "Required");
    std::strncpy(proof_buffer, stringProof.c_str(), *proof_size);
    std::strncpy(public_buffer, stringPublic.c_str(), *public_size);
    API_CATCH
}

}  // extern "C"
