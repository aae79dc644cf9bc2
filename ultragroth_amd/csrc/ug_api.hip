// ug_api.hip -- implementation of the inner C-ABI declared in include/ultragroth_hip.h.
// Owns device memory, the stream and the reusable workspaces; translates between the reference's byte
// formats and the device forms; catches every exception at the boundary.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <exception>
#include "dev_common.hpp"
#include "internal.hpp"
#include "../../include/ultragroth_hip.h"

using namespace ug;

namespace {
thread_local std::string g_last_error;
int fail(const std::string& msg) { g_last_error = msg; return UG_ERROR; }
#define UG_TRY try {
#define UG_CATCH                                                       \
    } catch (const std::exception& e) { return fail(e.what()); }       \
    catch (...) { return fail("unknown error"); }                      \
    return UG_OK;
}  // namespace

// Host -> HBM copies of large caller buffers (the witness of every proof: 512 MiB at 2^24; the zkey sections at create:
// 9.4 GB). The caller's memory is pageable (the reference's API hands over plain pointers, src/prover.h:127-138; a zkey
// file is an mmap, src/fileloader.cpp:23-51), and one hipMemcpy from pageable memory runs at about half the PCIe rate
// (measured 27 GB/s). Here LANES host threads each copy 8 MiB chunks into their own pinned buffers and queue the DMA on
// their own stream, DEPTH chunks deep, so page-touching memcpy and DMA overlap; `after` (optional) is queued behind each
// chunk's DMA on the same stream, e.g. the conversion kernel for that chunk's records. Blocking: returns when every
// chunk and every `after` has completed.
struct StagedUploader {
    static constexpr size_t CHUNK = (size_t)8 << 20;
    static constexpr int MAX_LANES = 8, DEPTH = 2;
    static constexpr size_t MIN_BYTES = (size_t)32 << 20;      // below this a plain copy is as fast
    struct Lane { hipStream_t stream = nullptr; uint8_t* buf[DEPTH] = {nullptr, nullptr}; hipEvent_t done[DEPTH] = {nullptr, nullptr}; };
    Lane lanes[MAX_LANES];
    int n_lanes = 0;
    std::mutex turn;            // one upload at a time: the lanes' pinned buffers are the uploader's (a witness staged for the
                                // next proof from a second host thread meets the uploads of the running one here)
    typedef std::function<void(size_t offset, size_t bytes, hipStream_t stream)> After;
    void init() {
        if (n_lanes) return;
        int want = 4;
        if (const char* e = getenv("ULTRAGROTH_UPLOAD_THREADS")) want = atoi(e);
        want = want < 1 ? 1 : want > MAX_LANES ? MAX_LANES : want;
        // The lanes' streams are created with the LOWEST priority, i.e. from a pool of hardware queues of their own: the
        // runtime maps the streams of one priority onto a few hardware queues (four by default), and a copy whose stream
        // shares a queue with a stream that holds a whole proof's kernels waits for all of them. Measured at 2^24, a witness
        // staged beside a running proof: normal priority (two of the four lanes share the compute streams' queues) 146 ms,
        // i.e. the whole proof; highest priority 10 ms, but the proof beside it takes 6 ms longer (its dispatches yield to
        // the copies' packets); lowest priority 38 ms, the proof 2.5 ms longer -- the best of the three for the pair. With
        // the device idle all three copy at the same 53 GB/s.
        int least = 0, greatest = 0;
        UG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (int l = 0; l < want; l++) {
            const char* pe = getenv("UG_UPLOAD_PRIORITY");      // tuning knob: l(ow, default) | n(ormal) | h(igh)
            const int prio = (pe && pe[0] == 'h') ? greatest : (pe && pe[0] == 'n') ? 0 : least;
            UG_HIP(hipStreamCreateWithPriority(&lanes[l].stream, hipStreamNonBlocking, prio));
            for (int d = 0; d < DEPTH; d++) {
                UG_HIP(hipHostMalloc((void**)&lanes[l].buf[d], CHUNK, hipHostMallocDefault));
                UG_HIP(hipEventCreateWithFlags(&lanes[l].done[d], hipEventDisableTiming));
            }
            n_lanes = l + 1;
        }
    }
    void upload(int device, void* dst, const void* src, size_t bytes, const After& after = After()) {
        std::lock_guard<std::mutex> one(turn);
        init();
        const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
        std::exception_ptr errs[MAX_LANES];
        auto work = [&](int l) {
            try {
                UG_HIP(hipSetDevice(device));
                Lane& ln = lanes[l];
                size_t turn = 0;
                for (size_t c = (size_t)l; c < n_chunks; c += (size_t)n_lanes, turn++) {
                    const int d = (int)(turn % DEPTH);
                    if (turn >= DEPTH) UG_HIP(hipEventSynchronize(ln.done[d]));      // the DMA out of this buffer has finished
                    const size_t off = c * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
                    memcpy(ln.buf[d], static_cast<const uint8_t*>(src) + off, len);
                    UG_HIP(hipMemcpyAsync(static_cast<uint8_t*>(dst) + off, ln.buf[d], len, hipMemcpyHostToDevice, ln.stream));
                    UG_HIP(hipEventRecord(ln.done[d], ln.stream));
                    if (after) after(off, len, ln.stream);
                }
                UG_HIP(hipStreamSynchronize(ln.stream));
            } catch (...) { errs[l] = std::current_exception(); }
        };
        std::vector<std::thread> th;
        for (int l = 1; l < n_lanes && (size_t)l < n_chunks; l++) th.emplace_back(work, l);
        work(0);
        for (auto& t : th) t.join();
        for (int l = 0; l < n_lanes; l++) if (errs[l]) std::rethrow_exception(errs[l]);
    }
    void release() {
        for (int l = 0; l < n_lanes; l++) {
            for (int d = 0; d < DEPTH; d++) { if (lanes[l].buf[d]) hipHostFree(lanes[l].buf[d]); if (lanes[l].done[d]) hipEventDestroy(lanes[l].done[d]); }
            if (lanes[l].stream) hipStreamDestroy(lanes[l].stream);
            lanes[l] = Lane();
        }
        n_lanes = 0;
    }
};

struct ug_ctx {
    int device = 0;
    StagedUploader uploader;
    hipStream_t stream = nullptr;
    MsmWorkspace ws_g1, ws_g2;
    MsmStats stats[4];                     // [0] G1, [1] G2 bucket-accumulation launches, [2] NTT pass launches, [3] G1 group accumulation
    // Per-kernel statistics are OFF until somebody asks for them (ug_ctx_kernel_stats with reset != 0 switches them on for the
    // context): an event pair around a launch costs 10-25 us of idle device on this runtime (rocprofv3 trace of a 2^20 proof,
    // profiles/r05_variants_ab.txt item 1) -- 0.2 ms per proof that only a measuring caller should pay.
    bool kernel_stats_on = false;
    MsmStats* stat(int k) { return kernel_stats_on ? &stats[k] : nullptr; }
    double msm_ms = 0, fft_ms = 0;
    // stream-time accounting without host waits: every timed span is an event pair that is resolved (elapsed time added
    // to its accumulator) the next time the stream is known to be idle -- ug_ctx_collect, ug_ctx_timings, ug_ctx_sync
    struct Span { hipEvent_t e0, e1; double* acc; };
    std::vector<Span> spans_free, spans_pending;
    // MSMs queued by ug_msm_batch_enqueue whose results are still on their way (pinned_results slot k <-> pending_msm[k])
    struct QueuedMsm { MsmPending pend; void* out; bool g2; };
    std::vector<QueuedMsm> pending_msm;
    hipEvent_t order_event = nullptr;      // ug_ctx_wait
    hipStream_t side_stream = nullptr;     // ug_msm_witness_enqueue: the G2 tail runs here beside the G1 tail
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    bool defer_tables = false;             // ug_ctx_defer_tables: sets are created with room for their window tables, which are
                                           // then built piece by piece (ug_bases_tables_step)
    ug_graph* recording = nullptr;         // the stream is being captured into this graph (ug_graph_begin .. ug_graph_end)
    std::vector<ug_graph*> launched;       // graphs launched on this stream whose event pairs are not accounted yet (resolve_spans)
    NttPlan raw_ntt;                       // cache for ug_fr_ntt
    u32* lookup_last = nullptr; u64 lookup_last_n = 0;   // zeroed scratch of ug_dvec_apply_lookup
    u32* lookup_stage = nullptr; size_t lookup_stage_bytes = 0;   // staging of the lookup calls (kept: two allocations and
                                                                  // two frees -- each a device-wide wait -- per proof otherwise)
    u32* stage_for_lookup(size_t bytes) {
        if (bytes > lookup_stage_bytes) {
            if (lookup_stage) hipFree(lookup_stage);
            lookup_stage = nullptr; lookup_stage_bytes = 0;
            UG_HIP(hipMalloc(&lookup_stage, bytes));
            lookup_stage_bytes = bytes;
        }
        return lookup_stage;
    }
    u32* pinned_results = nullptr;         // MsmStats::SLOTS result blocks of queued MSMs (ug_msm_batch)
    std::vector<void*> deferred_free;      // staging memory of queued set-up work: hipFree waits for the whole device, so it is
                                           // freed the next time the stream is idle anyway (ug_ctx_sync, destroy)
    void free_deferred() { for (void* p : deferred_free) hipFree(p); deferred_free.clear(); }
    void use() const { UG_HIP(hipSetDevice(device)); }
};
struct ug_bases {
    ug_ctx* ctx; bool g2; u64 n; u64 global_first; u32* pts;
    int table_c = 0;          // window width of the precomputed tables (0: none, pts holds the n points only)
    int members = 1;          // > 1: a group -- n = slots * members records, record slot * members + m is member m's point of
    u64 slots = 0;            //      scalar global_first + slot (ug_bases_create_group_g1)
    bool empty = false;       // every record is the point at infinity (e.g. the B2 section of a circuit without B-side wires): its
                              // products are the point at infinity and no kernel is launched for them
    u64 tables_built = ~(u64)0;   // deferred build (ug_ctx_defer_tables): points [0, tables_built) have their tables; >= n: all of them
    int deferred_c = 0;           // ... the width the set will get: until ug_bases_tables_adopt the set holds its n points only
    bool tables_usable() const { return !table_c || tables_built >= n; }
};
struct ug_dvec {
    ug_ctx* ctx; u64 n; u32* data; bool owns;
};
struct ug_index {
    ug_ctx* ctx; u64 n; u32* data;
};
struct ug_schedule {
    ug_ctx* ctx; MsmSchedule sched; u64 first = 0;
    BucketClasses cls;                    // ug_schedule_set_classes: applied by every later build
    u64 sp_first = 0, sp_end = 0;         // ... its special-bucket scalar range, in the scalar vector's (global) indices
    // the classes as the build of scalars [first, first + count) sees them (the scalar range clipped and made local)
    BucketClasses classes_for(u64 first_, u64 count) const {
        BucketClasses k = cls;
        if (!k.on()) return k;
        const u64 lo = sp_first > first_ ? sp_first - first_ : 0, hi = sp_end > first_ ? sp_end - first_ : 0;
        k.sp_lo = (u32)(lo < count ? lo : count); k.sp_hi = (u32)(hi < count ? hi : count);
        return k;
    }
};
struct ug_hpoly {
    ug_ctx* ctx; CoefMatrix mat; NttPlan ntt; u32 domain = 0, nvars = 0;
    u32 *a = nullptr, *b = nullptr, *c = nullptr, *t = nullptr, *t2 = nullptr;     // domain elements each
};

// A fixed launch sequence of one or two contexts of a device, captured once and replayed (ug_graph_*). It owns the event pairs
// that were recorded inside it -- the timed spans of the contexts' MSM | FFT accumulators and the per-kernel statistics -- and a
// copy of the products that were queued while it was captured (their result blocks arrive in the same pinned memory every time).
struct ug_graph {
    ug_ctx* c[2] = {nullptr, nullptr};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<ug_ctx::QueuedMsm> pend[2];
    struct Timed { hipEvent_t e0, e1; double* acc; };
    std::vector<Timed> spans;
    std::vector<CapturedSpan> cap[2][4];       // per context and kernel class (ug_ctx::stats)
    uint64_t epoch = 0;                        // alloc_epoch() when the capture ended
    size_t nodes = 0;
    bool capturing = false;
};

namespace {
// blocking copy of a caller buffer into device memory; `after` as in StagedUploader::upload (called once for a small copy)
// fresh: dst was allocated just now, nothing queued on the device refers to it -- the copy then neither waits for the
// context's stream nor uses it (a window-table build of the previous section may be running there: the upload of the next
// section overlaps it)
void host_to_device(ug_ctx* c, void* dst, const void* src, size_t bytes, const StagedUploader::After& after = StagedUploader::After(),
                    bool fresh = false) {
    if (!bytes) return;
    if (fresh) { c->uploader.upload(c->device, dst, src, bytes, after); return; }
    UG_HIP(hipStreamSynchronize(c->stream));                   // nothing queued earlier may still read or write dst
    if (bytes >= StagedUploader::MIN_BYTES) { c->uploader.upload(c->device, dst, src, bytes, after); return; }
    UG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    if (after) after(0, bytes, c->stream);
    UG_HIP(hipStreamSynchronize(c->stream));
}
// Test hook (ug_test_inject_fault): the `after`-th next pass through fault point `site` throws. Only in processes started
// with ULTRAGROTH_TEST_HOOKS=1 (the same gate as the blinding hook); otherwise fault_point() is one relaxed load.
std::atomic<int> g_fault_site{0}, g_fault_after{0};
bool test_hooks_on() {
    static const bool on = [] { const char* e = getenv("ULTRAGROTH_TEST_HOOKS"); return e && e[0] == '1' && !e[1]; }();
    return on;
}
void fault_point(int site) {
    if (g_fault_site.load(std::memory_order_relaxed) != site) return;
    if (g_fault_after.fetch_sub(1) != 1) return;
    g_fault_site.store(0);
    throw std::runtime_error("injected fault (test hook) at site " + std::to_string(site));
}
// ULTRAGROTH_TRACE=1: where the host time of the set-up calls goes (stderr, ms of a process-wide clock)
void trace_step(const char* what) {
    static const bool on = getenv("ULTRAGROTH_TRACE") && atoi(getenv("ULTRAGROTH_TRACE")) != 0;
    if (!on) return;
    static const auto origin = std::chrono::steady_clock::now();
    fprintf(stderr, "[ug_api] %9.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - origin).count(), what);
}
struct ScopedTimer {       // stream time between construction and stop() goes to *acc -- later: see ug_ctx::Span
    ug_ctx* c; ug_ctx::Span span; bool stopped = false, captured = false;
    ScopedTimer(const ScopedTimer&) = delete;
    ScopedTimer& operator=(const ScopedTimer&) = delete;
    ~ScopedTimer() {                                                      // the timed section threw: the event pair goes back unused
        if (stopped) return;
        if (captured) { hipEventDestroy(span.e0); hipEventDestroy(span.e1); }
        else c->spans_free.push_back(span);
    }
    ScopedTimer(ug_ctx* c_, double* acc_) : c(c_) {
        if (c->recording) {                          // a pair of the graph's own, recorded by every launch of it
            span = ug_ctx::Span{nullptr, nullptr, acc_};
            UG_HIP(hipEventCreate(&span.e0));
            if (hipEventCreate(&span.e1) != hipSuccess) { hipEventDestroy(span.e0); throw HipError("HIP error: hipEventCreate"); }
            captured = true;
            try { record_in_capture(span.e0, c->stream); }
            catch (...) { hipEventDestroy(span.e0); hipEventDestroy(span.e1); stopped = true; throw; }
            return;
        }
        if (c->spans_pending.size() >= 64) {        // a caller that never waits: account what has finished, without waiting
            std::vector<ug_ctx::Span> still;
            for (auto& sp : c->spans_pending) {
                float ms = 0;
                if (hipEventQuery(sp.e1) == hipSuccess && hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess) { *sp.acc += ms; c->spans_free.push_back(sp); }
                else still.push_back(sp);
            }
            (void)hipGetLastError();
            c->spans_pending.swap(still);
        }
        if (c->spans_free.empty()) {
            ug_ctx::Span sp{nullptr, nullptr, nullptr};
            UG_HIP(hipEventCreate(&sp.e0)); UG_HIP(hipEventCreate(&sp.e1));
            c->spans_free.push_back(sp);
        }
        span = c->spans_free.back();
        span.acc = acc_;
        UG_HIP(hipEventRecord(span.e0, c->stream));
        c->spans_free.pop_back();
    }
    void stop() {
        if (captured) {
            record_in_capture(span.e1, c->stream);
            c->recording->spans.push_back(ug_graph::Timed{span.e0, span.e1, span.acc});
            stopped = true;
            return;
        }
        UG_HIP(hipEventRecord(span.e1, c->stream));
        c->spans_pending.push_back(span);
        stopped = true;
    }
};
// after the stream has been synchronised: account the finished spans and the kernel statistics
void resolve_spans(ug_ctx* c) {
    for (auto& sp : c->spans_pending) {
        float ms = 0;
        UG_HIP(hipEventElapsedTime(&ms, sp.e0, sp.e1));
        *sp.acc += ms;
        c->spans_free.push_back(sp);
    }
    c->spans_pending.clear();
    for (int k = 0; k < 4; k++) c->stats[k].collect();
    // graphs launched on this stream have completed as well: one launch of each is accounted (their event pairs are re-recorded
    // by every launch, so a graph is resolved before it is launched again: ug_graph_launch sees to that)
    for (ug_graph* g : c->launched) {
        for (const ug_graph::Timed& t : g->spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) *t.acc += ms; else (void)hipGetLastError();
        }
        for (int q = 0; q < 2; q++)
            if (g->c[q]) for (int k = 0; k < 4; k++) g->c[q]->stats[k].account(g->cap[q][k]);
    }
    c->launched.clear();
}
void sync_and_resolve(ug_ctx* c) {
    UG_HIP(hipStreamSynchronize(c->stream));
    resolve_spans(c);
    c->free_deferred();
}
template <class F> void store_mont256(uint8_t* out, const F& v);
template <> void store_mont256<Fq>(uint8_t* out, const Fq& v) { u32 w[8]; to_mont256(w, v); memcpy(out, w, 32); }
}  // namespace

extern "C" {

const char* ug_last_error(void) { return g_last_error.c_str(); }

int ug_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int ug_ctx_create(ug_ctx** out, int device) { return ug_ctx_create_priority(out, device, 0); }
int ug_ctx_create_priority(ug_ctx** out, int device, int priority_class) {
    UG_TRY
    if (!out) throw std::invalid_argument("null ctx pointer");
    int n = 0;
    UG_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) throw std::invalid_argument("no such HIP device: " + std::to_string(device));
    ug_ctx* c = new ug_ctx();
    c->device = device;
    c->use();
    if (priority_class) {
        int least = 0, greatest = 0;                // (numerically: greatest priority = lowest number)
        UG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        UG_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, priority_class > 0 ? greatest : least));
    } else {
        UG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    UG_HIP(hipEventCreateWithFlags(&c->order_event, hipEventDisableTiming));
    for (int k = 0; k < 4; k++) c->stats[k].create();
    UG_HIP(hipHostMalloc((void**)&c->pinned_results, (size_t)MsmStats::MAX_BATCH * MSM_PENDING_WORDS * 4, hipHostMallocDefault));
    *out = c;
    UG_CATCH
}
void ug_ctx_destroy(ug_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    c->free_deferred();
    c->ws_g1.release(); c->ws_g2.release(); c->raw_ntt.release(); c->uploader.release();
    if (c->lookup_last) hipFree(c->lookup_last);
    if (c->lookup_stage) hipFree(c->lookup_stage);
    for (auto& sp : c->spans_free) { hipEventDestroy(sp.e0); hipEventDestroy(sp.e1); }
    for (auto& sp : c->spans_pending) { hipEventDestroy(sp.e0); hipEventDestroy(sp.e1); }
    if (c->order_event) hipEventDestroy(c->order_event);
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    if (c->side_fork) hipEventDestroy(c->side_fork);
    if (c->side_join) hipEventDestroy(c->side_join);
    for (int k = 0; k < 4; k++) c->stats[k].destroy();
    if (c->pinned_results) hipHostFree(c->pinned_results);
    hipStreamDestroy(c->stream);
    delete c;
}
int ug_ctx_sync(ug_ctx* c) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    c->use();
    sync_and_resolve(c);
    UG_CATCH
}
// everything queued on `waiter` after this call starts only when everything queued on `signal` before it has finished
// (device-side ordering, no host wait)
int ug_ctx_wait(ug_ctx* waiter, ug_ctx* signal) {
    UG_TRY
    if (!waiter || !signal) throw std::invalid_argument("null argument");
    if (waiter->device != signal->device) throw std::invalid_argument("contexts on different devices");
    waiter->use();
    UG_HIP(hipEventRecord(signal->order_event, signal->stream));
    UG_HIP(hipStreamWaitEvent(waiter->stream, signal->order_event, 0));
    UG_CATCH
}

// table_c != 0: the set is created WITH its fixed-base window tables -- room for all of them is allocated at once, the points
// go straight into table 0, and the table kernel is queued on the context's stream without a host wait, so that the upload
// of the caller's next section (staged through the uploader's own streams) runs beside it. Everything queued on the
// context later is ordered behind the tables; ug_ctx_sync ends the build.
static bool host_all_zero(const void* p, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i < bytes && ((uintptr_t)(b + i) & 7); i++) if (b[i]) return false;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); if (w) return false; }
    for (; i < bytes; i++) if (b[i]) return false;
    return true;
}
static int bases_create(ug_ctx* c, const void* host, u64 n, u64 global_first, bool g2, int table_c, ug_bases** out) {
    UG_TRY
    if (!c || !out || (!host && n)) throw std::invalid_argument("null argument");
    c->use();
    int windows = 1;
    if (table_c) {
        windows = MsmGeometry::choose_tables(n, table_c).windows;      // validates the width
        if (n > ((u64)1 << TABLE_INDEX_BITS)) throw std::invalid_argument("window tables need at most 2^27 points per set");
    }
    ug_bases* b = new ug_bases{c, g2, n, global_first, nullptr, 0};
    const size_t rec = g2 ? 128 : 64, bytes = (size_t)n * rec;
    b->empty = n != 0 && host_all_zero(host, bytes);          // (stops at the first point that is not infinity: the first record, normally)
    const bool defer = table_c && c->defer_tables;            // the room for the tables comes later too (ug_bases_tables_alloc / _adopt):
    if (defer) windows = 1;                                   // a first allocation of tens of GiB can take a second by itself
    if (hipMalloc(&b->pts, bytes ? bytes * (size_t)windows : 4) != hipSuccess) {
        (void)hipGetLastError();
        delete b;
        throw std::runtime_error(table_c ? "not enough device memory for the window tables" : "not enough device memory for the base points");
    }
    if (n) {
        // each chunk's records are converted to the device form behind its own DMA (chunks are whole records: 8 MiB / 128)
        u32* pts = b->pts;
        try {
            host_to_device(c, pts, host, bytes, [pts, rec, g2](size_t off, size_t len, hipStream_t st) {
                u32* p = pts + off / 4;
                if (g2) convert_points_g2(p, len / rec, st); else convert_points_g1(p, len / rec, st);
            }, /*fresh*/ true);
            if (table_c && !defer) build_window_tables(g2, pts, n, table_c, windows, c->stream);
        } catch (...) { hipFree(b->pts); delete b; throw; }
    }
    if (defer) b->deferred_c = table_c; else b->table_c = table_c;
    *out = b;
    UG_CATCH
}
int ug_bases_create_g1(ug_ctx* c, const void* host, uint64_t n, uint64_t gf, ug_bases** out) { return bases_create(c, host, n, gf, false, 0, out); }
int ug_bases_create_g2(ug_ctx* c, const void* host, uint64_t n, uint64_t gf, ug_bases** out) { return bases_create(c, host, n, gf, true, 0, out); }
int ug_bases_create_tables_g1(ug_ctx* c, const void* host, uint64_t n, uint64_t gf, int table_c, ug_bases** out) { return bases_create(c, host, n, gf, false, table_c, out); }
int ug_bases_create_tables_g2(ug_ctx* c, const void* host, uint64_t n, uint64_t gf, int table_c, ug_bases** out) { return bases_create(c, host, n, gf, true, table_c, out); }

// A group of `members` (2 or 3) G1 sets that are always multiplied by the same scalars, as ONE array of members-point
// records (msm.hip: segment_accumulate_group_kernel). Member m brings n[m] points; its first point belongs to the scalar
// with global index first[m]. The group covers the scalars [group_first, group_first + slots); slots a member has no point
// for hold infinity. With table_c the window tables are built over the interleaved array (a plain G1 array of slots * members
// points as far as the table kernel is concerned), queued on the context's stream as for ug_bases_create_tables_g1.
int ug_bases_create_group_g1(ug_ctx* c, int members, const void* const* host, const uint64_t* n, const uint64_t* first,
                             uint64_t group_first, uint64_t slots, int table_c, ug_bases** out) {
    UG_TRY
    if (!c || !out || !host || !n || !first) throw std::invalid_argument("null argument");
    if (members < 2 || members > 3) throw std::invalid_argument("a base group has 2 or 3 members");
    for (int m = 0; m < members; m++) {
        if (!host[m] && n[m]) throw std::invalid_argument("null argument");
        if (n[m] && (first[m] < group_first || first[m] + n[m] > group_first + slots)) throw std::invalid_argument("group member outside the group's scalar range");
    }
    c->use();
    int windows = 1;
    if (table_c) windows = MsmGeometry::choose_tables(slots, table_c).windows;      // validates the width
    if (slots > ((u64)1 << TABLE_INDEX_BITS)) throw std::invalid_argument("a base group holds at most 2^27 scalars");
    ug_bases* b = new ug_bases{c, false, slots * (u64)members, group_first, nullptr, 0};
    b->members = members; b->slots = slots;
    const bool defer = table_c && c->defer_tables;            // (as bases_create)
    if (defer) windows = 1;
    const size_t bytes = (size_t)b->n * 64;
    u64 most = 0;
    for (int m = 0; m < members; m++) most = n[m] > most ? n[m] : most;
    u32* stage = nullptr;
    trace_step("group: allocating");
    if (hipMalloc(&b->pts, bytes ? bytes * (size_t)windows : 4) != hipSuccess || hipMalloc(&stage, most ? (size_t)most * 64 : 4) != hipSuccess) {
        (void)hipGetLastError();
        if (b->pts) hipFree(b->pts);
        delete b;
        throw std::runtime_error(table_c ? "not enough device memory for the window tables" : "not enough device memory for the base points");
    }
    try {
        trace_step("group: allocated");
        if (bytes) {
            UG_HIP(hipMemsetAsync(b->pts, 0, bytes, c->stream));            // slots without a point: infinity
            UG_HIP(hipStreamSynchronize(c->stream));
        }
        trace_step("group: table 0 cleared");
        u32* pts = b->pts;
        for (int m = 0; m < members; m++) {
            trace_step("group: member upload");
            // chunks of whole records: converted to the device form and moved to their slots behind their own DMA
            if (!n[m]) continue;
            const u64 slot0 = first[m] - group_first;
            host_to_device(c, stage, host[m], (size_t)n[m] * 64, [=](size_t off, size_t len, hipStream_t st) {
                convert_points_g1(stage + off / 4, len / 64, st);
                interleave_points_g1(pts, stage + off / 4, len / 64, members, m, slot0 + off / 64, st);
            }, /*fresh*/ true);
        }
        if (table_c && b->n && !defer) build_window_tables(false, pts, b->n, table_c, windows, c->stream);
    } catch (...) { hipFree(stage); hipFree(b->pts); delete b; throw; }
    c->deferred_free.push_back(stage);      // (not hipFree here: it would wait for the table build just queued, and the caller's next
                                            // section could no longer be uploaded beside it)
    if (defer) b->deferred_c = table_c; else b->table_c = table_c;
    *out = b;
    UG_CATCH
}
int ug_bases_members(const ug_bases* b) { return b ? b->members : 0; }
int ug_points_all_infinity(const void* host_points, uint64_t n, uint64_t record_bytes) {
    return (host_points && n) ? (host_all_zero(host_points, (size_t)n * (size_t)record_bytes) ? 1 : 0) : 0;
}
int ug_msm_table_window(uint64_t n) { return MsmGeometry::table_window(n); }
uint64_t ug_bases_tables_bytes(uint64_t n, int g2, int c) {
    if (c < TABLE_MIN_C || c > TABLE_MAX_C) return 0;
    return (uint64_t)((255 + c - 1) / c - 1) * n * (g2 ? 128 : 64);
}
int ug_bases_precompute(ug_bases* b, int c) {
    UG_TRY
    if (!b) throw std::invalid_argument("null argument");
    if (b->table_c) throw std::invalid_argument("bases already hold window tables");
    MsmGeometry g = MsmGeometry::choose_tables(b->n, c);             // validates c
    if ((b->members > 1 ? b->slots : b->n) > ((u64)1 << TABLE_INDEX_BITS)) throw std::invalid_argument("window tables need at most 2^27 points per set");
    ug_ctx* ctx = b->ctx;
    ctx->use();
    if (b->n) {
        size_t rec = b->g2 ? 128 : 64;
        u32* all = nullptr;
        if (hipMalloc(&all, (size_t)g.windows * b->n * rec) != hipSuccess) {
            (void)hipGetLastError();
            throw std::runtime_error("not enough device memory for the window tables");
        }
        UG_HIP(hipMemcpyAsync(all, b->pts, (size_t)b->n * rec, hipMemcpyDeviceToDevice, ctx->stream));
        build_window_tables(b->g2, all, b->n, c, g.windows, ctx->stream);
        UG_HIP(hipStreamSynchronize(ctx->stream));
        alloc_epoch_bump();                          // (captured launch sequences that read the old array are stale now)
        hipFree(b->pts);
        b->pts = all;
    }
    b->table_c = c;
    UG_CATCH
}
int ug_bases_drop_tables(ug_bases* b) {
    UG_TRY
    if (!b) throw std::invalid_argument("null argument");
    b->deferred_c = 0;                               // (a deferred build that has not got its room yet is simply called off)
    if (!b->table_c) return UG_OK;
    ug_ctx* ctx = b->ctx;
    ctx->use();
    UG_HIP(hipStreamSynchronize(ctx->stream));
    size_t bytes = (size_t)b->n * (b->g2 ? 128 : 64);
    u32* small = nullptr;
    UG_HIP(hipMalloc(&small, bytes ? bytes : 4));
    if (bytes) UG_HIP(hipMemcpy(small, b->pts, bytes, hipMemcpyDeviceToDevice));
    alloc_epoch_bump();
    hipFree(b->pts);
    b->pts = small;
    b->table_c = 0;
    UG_CATCH
}
int ug_bases_table_window(const ug_bases* b) { return b ? b->table_c : 0; }
// DEFERRED TABLE BUILDS (cold start of a created prover). After ug_ctx_defer_tables(ctx, 1) the sets made on the context with a
// table width get the room for their tables and table 0, nothing else: ug_bases_tables_step builds the tables of the next
// `max_points` points on the context's stream and WAITS for them (a bounded piece of device time: the caller interleaves the
// pieces with proofs, which use the classic windows on table 0 until *remaining reaches 0).
int ug_ctx_defer_tables(ug_ctx* c, int on) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    c->defer_tables = on != 0;
    UG_CATCH
}
// The room for a deferred set's tables, in two calls: ug_bases_tables_alloc is nothing but the allocation -- it touches no stream and
// may run beside proofs (a first allocation of 36 GiB was seen to take 1.1 s, which is why create no longer makes it) -- and
// ug_bases_tables_adopt, for a moment when nothing is queued on the set's context (the caller's turn), moves the points into table
// 0 of the new array, swaps it in and frees the old one. The pieces (ug_bases_tables_step) come after it.
int ug_bases_tables_alloc(ug_bases* b, void** mem) {
    UG_TRY
    if (!b || !mem) throw std::invalid_argument("null argument");
    *mem = nullptr;
    if (!b->deferred_c || !b->n) return UG_OK;                       // nothing deferred (or nothing to build)
    b->ctx->use();
    const int windows = MsmGeometry::choose_tables(b->n, b->deferred_c).windows;
    if (hipMalloc(mem, (size_t)windows * b->n * (b->g2 ? 128 : 64)) != hipSuccess) {
        (void)hipGetLastError();
        *mem = nullptr;
        throw std::runtime_error("not enough device memory for the window tables");
    }
    UG_CATCH
}
int ug_bases_tables_adopt(ug_bases* b, void* mem) {
    UG_TRY
    if (!b) throw std::invalid_argument("null argument");
    if (!b->deferred_c) { if (mem) hipFree(mem); return UG_OK; }
    ug_ctx* c = b->ctx;
    c->use();
    if (b->n) {
        if (!mem) throw std::invalid_argument("null argument");
        UG_HIP(hipMemcpyAsync(mem, b->pts, (size_t)b->n * (b->g2 ? 128 : 64), hipMemcpyDeviceToDevice, c->stream));
        UG_HIP(hipStreamSynchronize(c->stream));
        alloc_epoch_bump();                          // (captured launch sequences that read the old array are stale now)
        hipFree(b->pts);
        b->pts = static_cast<u32*>(mem);
    }
    b->table_c = b->deferred_c;
    b->deferred_c = 0;
    b->tables_built = b->n ? 0 : ~(u64)0;
    UG_CATCH
}
int ug_bases_tables_step(ug_bases* b, uint64_t max_points, uint64_t* remaining) {
    UG_TRY
    if (!b) throw std::invalid_argument("null argument");
    if (b->deferred_c) throw std::logic_error("the set's tables have no room yet (ug_bases_tables_alloc / ug_bases_tables_adopt)");
    if (b->table_c && b->tables_built < b->n) {
        ug_ctx* c = b->ctx;
        c->use();
        const u64 first = b->tables_built, count = max_points < b->n - first ? max_points : b->n - first;
        const int windows = MsmGeometry::choose_tables(b->n, b->table_c).windows;
        // (a group is one array of n = slots * members records; a lane of the table kernel carries 4 (G1) / 2 (G2) consecutive
        // points, so pieces start at multiples of 4)
        const u64 take = count >= b->n - first ? b->n - first : (count + 3) / 4 * 4;
        build_window_tables(b->g2, b->pts, b->n, b->table_c, windows, c->stream, first, take);
        UG_HIP(hipStreamSynchronize(c->stream));
        b->tables_built = first + take >= b->n ? ~(u64)0 : first + take;
    }
    if (remaining) *remaining = (b->table_c && b->tables_built < b->n) ? b->n - b->tables_built : 0;
    UG_CATCH
}
// 1: a schedule with window tables may use this set (it has none to wait for, or they are complete); 0: a deferred build is not finished
int ug_bases_tables_ready(const ug_bases* b) { return b && b->tables_usable() ? 1 : 0; }
int ug_schedule_trim(ug_schedule* s) {
    UG_TRY
    if (!s) throw std::invalid_argument("null argument");
    s->ctx->use();
    UG_HIP(hipStreamSynchronize(s->ctx->stream));
    s->sched.release();
    UG_CATCH
}
int ug_ctx_trim(ug_ctx* c) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    c->use();
    UG_HIP(hipStreamSynchronize(c->stream));
    c->ws_g1.release(); c->ws_g2.release(); c->raw_ntt.release();
    if (c->lookup_last) { hipFree(c->lookup_last); c->lookup_last = nullptr; c->lookup_last_n = 0; }
    if (c->lookup_stage) { hipFree(c->lookup_stage); c->lookup_stage = nullptr; c->lookup_stage_bytes = 0; }
    UG_CATCH
}
int ug_ctx_mem_info(ug_ctx* c, uint64_t* free_bytes, uint64_t* total_bytes) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    c->use();
    size_t f = 0, t = 0;
    UG_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    UG_CATCH
}
void ug_bases_destroy(ug_bases* b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    alloc_epoch_bump();
    hipFree(b->pts);
    delete b;
}

int ug_dvec_create(ug_ctx* c, uint64_t n, ug_dvec** out) {
    UG_TRY
    if (!c || !out) throw std::invalid_argument("null argument");
    c->use();
    ug_dvec* v = new ug_dvec{c, n, nullptr, true};
    UG_HIP(hipMalloc(&v->data, n ? (size_t)n * 32 : 32));
    *out = v;
    UG_CATCH
}
int ug_dvec_upload(ug_dvec* v, const void* host, uint64_t n) {
    UG_TRY
    if (!v || (!host && n)) throw std::invalid_argument("null argument");
    if (n > v->n) throw std::invalid_argument("upload larger than the vector");
    v->ctx->use();
    host_to_device(v->ctx, v->data, host, (size_t)n * 32);
    UG_CATCH
}
int ug_dvec_upload_range(ug_dvec* v, const void* host, uint64_t first, uint64_t n, ug_ctx* via) {
    UG_TRY
    if (!v || (!host && n)) throw std::invalid_argument("null argument");
    if (first + n > v->n) throw std::invalid_argument("upload outside the vector");
    ug_ctx* c = via ? via : v->ctx;
    if (c->device != v->ctx->device) throw std::invalid_argument("upload context on another device");
    c->use();
    host_to_device(c, v->data + first * 8, host, (size_t)n * 32);
    UG_CATCH
}
int ug_dvec_upload_idle(ug_dvec* v, const void* host, uint64_t n) {
    UG_TRY
    if (!v || (!host && n)) throw std::invalid_argument("null argument");
    if (n > v->n) throw std::invalid_argument("upload larger than the vector");
    v->ctx->use();
    // the caller vouches that nothing queued on the device reads or writes v: the copy runs on the uploader's own streams and
    // neither waits for the context's stream nor touches it (a proof may be running there)
    host_to_device(v->ctx, v->data, host, (size_t)n * 32, StagedUploader::After(), /*fresh*/ true);
    UG_CATCH
}
int ug_dvec_download(const ug_dvec* v, void* host, uint64_t first, uint64_t n) {
    UG_TRY
    if (!v || (!host && n)) throw std::invalid_argument("null argument");
    if (first + n > v->n) throw std::invalid_argument("download outside the vector");
    v->ctx->use();
    if (n) UG_HIP(hipMemcpyAsync(host, v->data + first * 8, (size_t)n * 32, hipMemcpyDeviceToHost, v->ctx->stream));
    UG_HIP(hipStreamSynchronize(v->ctx->stream));
    UG_CATCH
}
int ug_dvec_gather(ug_dvec* out, const ug_dvec* src, const uint32_t* host_index, uint64_t n) {
    UG_TRY
    if (!out || !src || (!host_index && n)) throw std::invalid_argument("null argument");
    if (n > out->n) throw std::invalid_argument("gather larger than the output vector");
    ug_ctx* c = out->ctx;
    c->use();
    u32* idx = nullptr;
    UG_HIP(hipMalloc(&idx, n ? (size_t)n * 4 : 4));
    if (n) UG_HIP(hipMemcpyAsync(idx, host_index, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    gather_elements(out->data, src->data, idx, n, src->n, c->stream);
    UG_HIP(hipStreamSynchronize(c->stream));
    hipFree(idx);
    UG_CATCH
}
// index lists that stay on the device (UltraGroth's round_indexes / final_round_indexes are part of the zkey)
int ug_index_create(ug_ctx* c, const uint32_t* host_index, uint64_t n, ug_index** out) {
    UG_TRY
    if (!c || !out || (!host_index && n)) throw std::invalid_argument("null argument");
    c->use();
    ug_index* ix = new ug_index{c, n, nullptr};
    UG_HIP(hipMalloc(&ix->data, n ? (size_t)n * 4 : 4));
    if (n) UG_HIP(hipMemcpyAsync(ix->data, host_index, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    UG_HIP(hipStreamSynchronize(c->stream));
    *out = ix;
    UG_CATCH
}
void ug_index_destroy(ug_index* ix) {
    if (!ix) return;
    hipSetDevice(ix->ctx->device);
    hipFree(ix->data);
    delete ix;
}
int ug_dvec_gather_index(ug_dvec* out, const ug_dvec* src, const ug_index* index) {
    UG_TRY
    if (!out || !src || !index) throw std::invalid_argument("null argument");
    if (index->n > out->n) throw std::invalid_argument("gather larger than the output vector");
    ug_ctx* c = out->ctx;
    c->use();
    gather_elements(out->data, src->data, index->data, index->n, src->n, c->stream);      // queued; the stream orders its users
    UG_CATCH
}
int ug_dvec_scatter(ug_dvec* dst, const uint32_t* host_index, const void* host_values, uint64_t n) {
    UG_TRY
    if (!dst || ((!host_index || !host_values) && n)) throw std::invalid_argument("null argument");
    ug_ctx* c = dst->ctx;
    c->use();
    u32 *idx = nullptr, *val = nullptr;
    UG_HIP(hipMalloc(&idx, n ? (size_t)n * 4 : 4));
    UG_HIP(hipMalloc(&val, n ? (size_t)n * 32 : 32));
    if (n) {
        UG_HIP(hipMemcpyAsync(idx, host_index, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        UG_HIP(hipMemcpyAsync(val, host_values, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    }
    scatter_elements(dst->data, idx, val, n, dst->n, c->stream);
    UG_HIP(hipStreamSynchronize(c->stream));
    hipFree(idx); hipFree(val);
    UG_CATCH
}
int ug_dvec_apply_lookup(ug_dvec* dst, const uint32_t* w_idx, const uint32_t* p_idx, uint64_t n, const uint32_t* chunks,
                         uint64_t n_chunks, const void* table, uint64_t lookup_size) {
    UG_TRY
    if (!dst || ((!w_idx || !p_idx) && n) || (!chunks && n_chunks) || !table) throw std::invalid_argument("null argument");
    if (n >= 0xffffffffull) throw std::invalid_argument("too many lookup writes");
    const uint64_t total = 1 + n_chunks + 2 * lookup_size;          // length of the reference's push_vector
    for (uint64_t j = 0; j < n_chunks; j++)
        if (chunks[j] >= lookup_size) throw std::range_error("uwtns: chunk index outside the lookup table");
    for (uint64_t i = 0; i < n; i++)
        if (w_idx[i] >= dst->n || p_idx[i] >= total) throw std::range_error("uwtns: lookup index out of range");
    if (!n) return UG_OK;
    ug_ctx* c = dst->ctx;
    c->use();
    if (c->lookup_last_n < dst->n) {
        if (c->lookup_last) hipFree(c->lookup_last);
        c->lookup_last = nullptr; c->lookup_last_n = 0;
        UG_HIP(hipMalloc(&c->lookup_last, (size_t)dst->n * 4));
        UG_HIP(hipMemsetAsync(c->lookup_last, 0, (size_t)dst->n * 4, c->stream));
        c->lookup_last_n = dst->n;
    }
    // one staging allocation: w_idx | p_idx | chunks | table
    const size_t tbytes = (size_t)(1 + 2 * lookup_size) * 32;
    u32* stage = c->stage_for_lookup((size_t)(2 * n + n_chunks) * 4 + tbytes);
    u32 *d_w = stage, *d_p = stage + n, *d_c = stage + 2 * n, *d_t = stage + 2 * n + n_chunks;
    UG_HIP(hipMemcpyAsync(d_w, w_idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    UG_HIP(hipMemcpyAsync(d_p, p_idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (n_chunks) UG_HIP(hipMemcpyAsync(d_c, chunks, (size_t)n_chunks * 4, hipMemcpyHostToDevice, c->stream));
    UG_HIP(hipMemcpyAsync(d_t, table, tbytes, hipMemcpyHostToDevice, c->stream));
    apply_lookup(dst->data, c->lookup_last, d_w, d_p, n, d_c, n_chunks, d_t, c->stream);
    UG_HIP(hipStreamSynchronize(c->stream));
    UG_CATCH
}
int ug_fr_lookup_table(ug_ctx* c, const void* rand_plain, const uint32_t* frequencies, uint64_t lookup_size, void* table_out) {
    UG_TRY
    if (!c || !rand_plain || (!frequencies && lookup_size) || !table_out) throw std::invalid_argument("null argument");
    c->use();
    const size_t tbytes = (size_t)(1 + 2 * lookup_size) * 32;
    u32* table = c->stage_for_lookup(tbytes + (lookup_size ? (size_t)lookup_size * 4 : 4));
    u32* freq = table + tbytes / 4;
    UG_HIP(hipMemcpyAsync(table, rand_plain, 32, hipMemcpyHostToDevice, c->stream));
    if (lookup_size) UG_HIP(hipMemcpyAsync(freq, frequencies, (size_t)lookup_size * 4, hipMemcpyHostToDevice, c->stream));
    lookup_table(table, freq, lookup_size, c->stream);
    UG_HIP(hipMemcpyAsync(table_out, table, tbytes, hipMemcpyDeviceToHost, c->stream));
    UG_HIP(hipStreamSynchronize(c->stream));
    UG_CATCH
}
uint64_t ug_dvec_size(const ug_dvec* v) { return v ? v->n : 0; }
void* ug_dvec_device_ptr(const ug_dvec* v) { return v ? v->data : nullptr; }
// dst[dst_first .. + count) = src[src_first .. + count); the vectors may live on different devices of the node (peer copy over
// xGMI, or through the host when peer access is not there: hipMemcpyPeer decides). Blocking; both vectors' earlier work must be
// complete (the phase calls that produce them are).
int ug_dvec_copy(ug_dvec* dst, uint64_t dst_first, const ug_dvec* src, uint64_t src_first, uint64_t count) {
    return ug_dvec_copy_via(dst, dst_first, src, src_first, count, nullptr);
}
// the same with the copy queued on `via`'s stream (a context of dst's device; NULL = dst's own): the witness of a chain rank is
// collected from its peers on the H branch's stream while its witness products are queued on the other
int ug_dvec_copy_via(ug_dvec* dst, uint64_t dst_first, const ug_dvec* src, uint64_t src_first, uint64_t count, ug_ctx* via) {
    UG_TRY
    if (!dst || !src) throw std::invalid_argument("null argument");
    if (dst_first + count > dst->n || src_first + count > src->n) throw std::invalid_argument("copy outside the vectors");
    if (!count) return UG_OK;
    ug_ctx* c = via ? via : dst->ctx;
    if (c->device != dst->ctx->device) throw std::invalid_argument("copy context on another device than the destination");
    c->use();
    const size_t bytes = (size_t)count * 32;
    if (src->ctx->device == c->device)
        UG_HIP(hipMemcpyAsync(dst->data + dst_first * 8, src->data + src_first * 8, bytes, hipMemcpyDeviceToDevice, c->stream));
    else {
        // two devices of the node: direct access over xGMI is switched on once per ordered pair (the copy works without it, staged
        // by the runtime; "already enabled" and "not supported" are not errors here)
        static std::atomic<bool> tried[64][64];
        const int a = c->device, b = src->ctx->device;
        if (a >= 0 && a < 64 && b >= 0 && b < 64 && !tried[a][b].exchange(true)) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(b, 0);      // (current device is a: c->use())
            (void)hipGetLastError();
        }
        UG_HIP(hipMemcpyPeerAsync(dst->data + dst_first * 8, c->device, src->data + src_first * 8, src->ctx->device, bytes, c->stream));
    }
    UG_HIP(hipStreamSynchronize(c->stream));
    UG_CATCH
}
int ug_dvec_wrap(ug_ctx* c, void* device_ptr, uint64_t n, ug_dvec** out) {
    UG_TRY
    if (!c || !out || (!device_ptr && n)) throw std::invalid_argument("null argument");
    if ((uintptr_t)device_ptr & 15) throw std::invalid_argument("device pointer must be 16-byte aligned");
    *out = new ug_dvec{c, n, static_cast<u32*>(device_ptr), false};
    UG_CATCH
}
void ug_dvec_destroy(ug_dvec* v) {
    if (!v) return;
    if (v->owns) { hipSetDevice(v->ctx->device); alloc_epoch_bump(); hipFree(v->data); }
    delete v;
}

int ug_schedule_create(ug_ctx* c, ug_schedule** out) {
    UG_TRY
    if (!c || !out) throw std::invalid_argument("null argument");
    ug_schedule* s = new ug_schedule();
    s->ctx = c;
    *out = s;
    UG_CATCH
}
int ug_schedule_build(ug_schedule* s, const ug_dvec* scalars, uint64_t first, uint64_t count) {
    UG_TRY
    if (!s || !scalars) throw std::invalid_argument("null argument");
    if (first + count > scalars->n) throw std::invalid_argument("schedule range outside the scalar vector");
    ug_ctx* c = s->ctx;
    c->use();
    fault_point(UG_FAULT_SCHEDULE_BUILD);
    ScopedTimer tm(c, &c->msm_ms);
    s->first = first;
    MsmGeometry g = MsmGeometry::choose(count);
    g.set_classes(s->classes_for(first, count));
    s->sched.build(scalars->data + first * 8, g, c->stream);
    tm.stop();
    UG_CATCH
}
int ug_schedule_build_tables(ug_schedule* s, const ug_dvec* scalars, uint64_t first, uint64_t count, int c) {
    UG_TRY
    if (!s || !scalars) throw std::invalid_argument("null argument");
    if (first + count > scalars->n) throw std::invalid_argument("schedule range outside the scalar vector");
    ug_ctx* ctx = s->ctx;
    ctx->use();
    fault_point(UG_FAULT_SCHEDULE_BUILD);
    MsmGeometry g = MsmGeometry::choose_tables(count, c);
    g.set_classes(s->classes_for(first, count));
    ScopedTimer tm(ctx, &ctx->msm_ms);
    s->first = first;
    s->sched.build(scalars->data + first * 8, g, ctx->stream);
    tm.stop();
    UG_CATCH
}
// Bucket classes for every later build of this schedule (include/ultragroth_hip.h; csrc/internal.hpp: BucketClasses)
int ug_schedule_set_classes(ug_schedule* s, int q_log, uint32_t first_residue, uint32_t residues, uint32_t specials,
                            uint64_t special_first, uint64_t special_count) {
    UG_TRY
    if (!s) throw std::invalid_argument("null argument");
    if (q_log < 0 || q_log > 8) throw std::invalid_argument("bucket classes: q_log outside [0, 8]");
    BucketClasses k;
    if (q_log) {
        if (residues < 1 || (uint64_t)first_residue + residues > ((uint64_t)1 << q_log)) throw std::invalid_argument("bucket classes: residues outside [0, 2^q_log)");
        if (specials > MSM_MAX_SPECIALS) throw std::invalid_argument("bucket classes: more than 64 special buckets");
        k.q_log = q_log; k.r0 = first_residue; k.cnt = residues; k.specials = specials;
    }
    s->cls = k;
    s->sp_first = special_first; s->sp_end = special_first + special_count;
    UG_CATCH
}
void ug_schedule_destroy(ug_schedule* s) {
    if (!s) return;
    hipSetDevice(s->ctx->device);
    s->sched.release();
    delete s;
}

static void affine_out_g1(uint8_t* out, const G1XYZZ& p) {
    if (is_inf(p)) { memset(out, 0, 64); return; }
    Fq x, y;
    xyzz_to_affine(x, y, p);
    u32 w[16];
    to_mont256(w, x); to_mont256(w + 8, y);
    memcpy(out, w, 64);
}
static void affine_out_g2(uint8_t* out, const G2XYZZ& p) {
    if (is_inf(p)) { memset(out, 0, 128); return; }
    Fq2 x, y;
    xyzz_to_affine(x, y, p);
    u32 w[32];
    to_mont256(w, x.a); to_mont256(w + 8, x.b); to_mont256(w + 16, y.a); to_mont256(w + 24, y.b);
    memcpy(out, w, 128);
}

// a schedule built for window tables needs bases that hold tables of the same width (a classic schedule reads table 0 only)
static void check_tables(const ug_bases* b, const ug_schedule* s) {
    if (s->sched.geo.tables && !b->tables_usable())
        throw std::logic_error("the window tables of this base set are not complete yet (ug_bases_tables_step)");
    if (s->sched.geo.tables && s->sched.geo.c != b->table_c)
        throw std::invalid_argument("schedule built for window tables of width " + std::to_string(s->sched.geo.c) +
                                    " but the bases hold " + (b->table_c ? "tables of width " + std::to_string(b->table_c) : std::string("no tables")));
}
int ug_msm_g1(ug_ctx* c, const ug_bases* b, const ug_schedule* s, int64_t index_shift, void* out) {
    UG_TRY
    if (!c || !b || !s || !out) throw std::invalid_argument("null argument");
    if (b->g2) throw std::invalid_argument("ug_msm_g1 called with G2 bases");
    if (b->members > 1) throw std::invalid_argument("a base group is multiplied with ug_msm_group_enqueue");
    check_tables(b, s);
    c->use();
    ScopedTimer tm(c, &c->msm_ms);
    int64_t delta = (int64_t)s->first - index_shift - (int64_t)b->global_first;
    if (!c->pending_msm.empty()) throw std::logic_error("collect the queued MSMs first (ug_ctx_collect)");
    G1XYZZ r = msm_g1(s->sched, c->ws_g1, b->pts, b->empty ? 0 : b->n, delta, c->stream, c->stat(0));      // synchronises the stream
    tm.stop();
    sync_and_resolve(c);
    affine_out_g1((uint8_t*)out, r);
    UG_CATCH
}
int ug_msm_g2(ug_ctx* c, const ug_bases* b, const ug_schedule* s, int64_t index_shift, void* out) {
    UG_TRY
    if (!c || !b || !s || !out) throw std::invalid_argument("null argument");
    if (!b->g2) throw std::invalid_argument("ug_msm_g2 called with G1 bases");
    check_tables(b, s);
    c->use();
    ScopedTimer tm(c, &c->msm_ms);
    int64_t delta = (int64_t)s->first - index_shift - (int64_t)b->global_first;
    if (!c->pending_msm.empty()) throw std::logic_error("collect the queued MSMs first (ug_ctx_collect)");
    G2XYZZ r = msm_g2(s->sched, c->ws_g2, b->pts, b->empty ? 0 : b->n, delta, c->stream, c->stat(1));      // synchronises the stream
    tm.stop();
    sync_and_resolve(c);
    affine_out_g2((uint8_t*)out, r);
    UG_CATCH
}

// Several MSMs over one schedule, queued back to back on the stream with ONE host synchronisation at the end: the
// latency-bound tail of one product (bucket reduction, tree sums, result copy) no longer leaves the device idle while the
// host converts the previous result (A, B1, B2, C of src/groth16.cpp:55-64 share the witness schedule).
int ug_msm_batch_enqueue(ug_ctx* c, int count, const ug_bases* const* bases, const ug_schedule* s, const int64_t* index_shifts,
                         void* const* outs) {
    UG_TRY
    if (!c || !s || (count && (!bases || !outs))) throw std::invalid_argument("null argument");
    if (count < 0 || c->pending_msm.size() + (size_t)count > (size_t)MsmStats::MAX_BATCH)
        throw std::invalid_argument("at most 8 products may be queued before ug_ctx_collect");
    for (int k = 0; k < count; k++) {
        if (!bases[k] || !outs[k]) throw std::invalid_argument("null argument");
        if (bases[k]->members > 1) throw std::invalid_argument("a base group is multiplied with ug_msm_group_enqueue");
        check_tables(bases[k], s);
    }
    c->use();
    ScopedTimer tm(c, &c->msm_ms);
    // the G1 products of the call form one batch, the G2 products another (msm.hip: msm_enqueue_multi): one accumulation
    // launch per product, the tail kernels once per batch; within a curve the caller's order is kept
    const size_t first_slot = c->pending_msm.size();
    for (int k = 0; k < count; k++) {
        ug_ctx::QueuedMsm q;
        q.g2 = bases[k]->g2; q.out = outs[k];
        c->pending_msm.push_back(q);
    }
    try {
    for (int g2 = 0; g2 < 2; g2++) {
        int idx[MSM_BATCH_MAX], n = 0;
        auto flush = [&] {
            if (!n) return;
            const u32* pts[MSM_BATCH_MAX]; u64 nb[MSM_BATCH_MAX]; int64_t delta[MSM_BATCH_MAX]; u32* host[MSM_BATCH_MAX]; MsmPending pend[MSM_BATCH_MAX];
            for (int q = 0; q < n; q++) {
                const ug_bases* b = bases[idx[q]];
                pts[q] = b->pts; nb[q] = b->empty ? 0 : b->n;      // (an all-infinity set takes no part: msm_enqueue_multi)
                delta[q] = (int64_t)s->first - (index_shifts ? index_shifts[idx[q]] : 0) - (int64_t)b->global_first;
                host[q] = c->pinned_results + (first_slot + idx[q]) * MSM_PENDING_WORDS;
            }
            if (g2) msm_enqueue_batch_g2(s->sched, c->ws_g2, n, pts, nb, delta, c->stream, c->stat(1), host, pend);
            else msm_enqueue_batch_g1(s->sched, c->ws_g1, n, pts, nb, delta, c->stream, c->stat(0), host, pend);
            for (int q = 0; q < n; q++) c->pending_msm[first_slot + idx[q]].pend = pend[q];
            n = 0;
        };
        for (int k = 0; k < count; k++) {
            if ((bases[k]->g2 ? 1 : 0) != g2) continue;
            idx[n++] = k;
            if (n == MSM_BATCH_MAX) flush();
        }
        flush();
    }
    } catch (...) { c->pending_msm.resize(first_slot); throw; }      // nothing of a failed call stays queued
    tm.stop();
    UG_CATCH
}
// The K products of a base group over one schedule, queued like ug_msm_batch_enqueue (results after ug_ctx_collect): ONE
// accumulation launch gathers each K-point record once and keeps K accumulators; outs[m] receives member m's sum (64 bytes).
int ug_msm_group_enqueue(ug_ctx* c, const ug_bases* group, const ug_schedule* s, void* const* outs) {
    UG_TRY
    if (!c || !group || !s || !outs) throw std::invalid_argument("null argument");
    if (group->members < 2) throw std::invalid_argument("not a base group");
    const int K = group->members;
    if (c->pending_msm.size() + (size_t)K > (size_t)MsmStats::MAX_BATCH) throw std::invalid_argument("at most 8 products may be queued before ug_ctx_collect");
    for (int m = 0; m < K; m++) if (!outs[m]) throw std::invalid_argument("null argument");
    check_tables(group, s);
    c->use();
    ScopedTimer tm(c, &c->msm_ms);
    const size_t first_slot = c->pending_msm.size();
    for (int m = 0; m < K; m++) {
        ug_ctx::QueuedMsm q;
        q.g2 = false; q.out = outs[m];
        c->pending_msm.push_back(q);
    }
    try {
        u32* host[MSM_BATCH_MAX]; MsmPending pend[MSM_BATCH_MAX];
        for (int m = 0; m < K; m++) host[m] = c->pinned_results + (first_slot + m) * MSM_PENDING_WORDS;
        const int64_t delta = (int64_t)s->first - (int64_t)group->global_first;
        msm_enqueue_group_g1(s->sched, c->ws_g1, K, group->pts, group->slots, delta, c->stream, c->stat(3), host, pend);
        for (int m = 0; m < K; m++) c->pending_msm[first_slot + m].pend = pend[m];
    } catch (...) { c->pending_msm.resize(first_slot); throw; }
    tm.stop();
    UG_CATCH
}
// The witness products of a proof -- the K products of a base group (A | B1 | C, or A | B1) and the G2 product B2 over ONE schedule
// (src/groth16.cpp:55-64) -- queued so that their latency-bound ends overlap: both accumulations back to back on the context's
// stream, then the G1 tail (fix-up of cut buckets, bucket reduction, tree sums, result copy) on that stream and the G2 tail on a
// side stream beside it. The two tails are chains of dependent EC additions in kernels of a few thousand waves: one after the other
// they leave most of the chip idle, most of all on a rank of a many-device prover (round 5: one rank of eight at 2^24 22.4 -> see
// profiles/r05_variants_ab.txt item 10). Results as after ug_msm_group_enqueue + ug_msm_batch_enqueue (ug_ctx_collect).
int ug_msm_witness_enqueue(ug_ctx* c, const ug_bases* group, const ug_bases* g2set, const ug_schedule* s, void* const* outs_group, void* out_g2) {
    UG_TRY
    if (!c || !group || !g2set || !s || !outs_group || !out_g2) throw std::invalid_argument("null argument");
    if (group->members < 2 || !g2set->g2 || g2set->members > 1) throw std::invalid_argument("a base group and a G2 set are expected");
    const int K = group->members;
    if (c->pending_msm.size() + (size_t)K + 1 > (size_t)MsmStats::MAX_BATCH) throw std::invalid_argument("at most 8 products may be queued before ug_ctx_collect");
    for (int m = 0; m < K; m++) if (!outs_group[m]) throw std::invalid_argument("null argument");
    check_tables(group, s); check_tables(g2set, s);
    c->use();
    if (!c->side_stream) {
        UG_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        UG_HIP(hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming));
        UG_HIP(hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming));
    }
    ScopedTimer tm(c, &c->msm_ms);
    const size_t first_slot = c->pending_msm.size();
    for (int m = 0; m <= K; m++) {
        ug_ctx::QueuedMsm q;
        q.g2 = m == K; q.out = m == K ? out_g2 : outs_group[m];
        c->pending_msm.push_back(q);
    }
    try {
        u32* host[MSM_BATCH_MAX]; MsmPending pend[MSM_BATCH_MAX];
        for (int m = 0; m < K; m++) host[m] = c->pinned_results + (first_slot + m) * MSM_PENDING_WORDS;
        u32* host2[1] = {c->pinned_results + (first_slot + K) * MSM_PENDING_WORDS};
        MsmPending pend2[1];
        const int64_t delta = (int64_t)s->first - (int64_t)group->global_first;
        const u32* pts2[1] = {g2set->pts};
        const u64 nb2[1] = {g2set->empty ? 0 : g2set->n};
        const int64_t delta2[1] = {(int64_t)s->first - (int64_t)g2set->global_first};
        // both accumulations, back to back
        msm_enqueue_group_g1(s->sched, c->ws_g1, K, group->pts, group->slots, delta, c->stream, c->stat(3), host, pend, MSM_PHASE_ACCUMULATE);
        msm_enqueue_batch_g2(s->sched, c->ws_g2, 1, pts2, nb2, delta2, c->stream, c->stat(1), host2, pend2, MSM_PHASE_ACCUMULATE);
        // the G2 tail on the side stream, the G1 tail on the context's, then the context's stream waits for the side
        UG_HIP(hipEventRecord(c->side_fork, c->stream));
        UG_HIP(hipStreamWaitEvent(c->side_stream, c->side_fork, 0));
        msm_enqueue_batch_g2(s->sched, c->ws_g2, 1, pts2, nb2, delta2, c->side_stream, nullptr, host2, pend2, MSM_PHASE_TAIL);
        msm_enqueue_group_g1(s->sched, c->ws_g1, K, group->pts, group->slots, delta, c->stream, nullptr, host, pend, MSM_PHASE_TAIL);
        UG_HIP(hipEventRecord(c->side_join, c->side_stream));
        UG_HIP(hipStreamWaitEvent(c->stream, c->side_join, 0));
        for (int m = 0; m < K; m++) c->pending_msm[first_slot + m].pend = pend[m];
        c->pending_msm[first_slot + K].pend = pend2[0];
    } catch (...) {
        c->pending_msm.resize(first_slot);
        if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
        throw;
    }
    tm.stop();
    UG_CATCH
}
// ONE host wait for everything queued on the context; then the queued MSM results are finished on the host
// (Horner over the window sums, affine conversion) and written to their `out` records
int ug_ctx_collect(ug_ctx* c) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    c->use();
    std::vector<ug_ctx::QueuedMsm> done;
    done.swap(c->pending_msm);                                  // whatever happens below, nothing stays queued
    sync_and_resolve(c);
    for (auto& q : done) {
        if (q.g2) affine_out_g2((uint8_t*)q.out, msm_collect_g2(q.pend));
        else affine_out_g1((uint8_t*)q.out, msm_collect_g1(q.pend));
    }
    UG_CATCH
}
// A caller that queued work (ug_msm_batch_enqueue, schedules, ug_hpoly_run) and then failed before ug_ctx_collect: waits for
// whatever is still running on the context -- the kernels read the caller's witness buffer and write the pinned result
// blocks -- and forgets the queued products, whose `out` pointers may be gone with the caller's stack frame. Never throws.
void ug_ctx_abandon(ug_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->recording) ug_graph_abort(c->recording->c[0]);      // a capture that failed half way: nothing of it was ever queued
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    c->pending_msm.clear();
    try { resolve_spans(c); } catch (...) { c->spans_pending.clear(); }
}
// the pass plan of the schedule sort for keys below 2^bits (sort.hip: radix_plan): host arithmetic only, for the CPU test-suite
int ug_sort_plan(int bits, int shift[4], int bins_log[4]) {
    if (bits < 1 || bits > 32 || !shift || !bins_log) return -1;
    return radix_plan(bits, shift, bins_log);
}
int ug_test_inject_fault(int site, int after) {
    if (!test_hooks_on()) return UG_ERROR;
    g_fault_after.store(after < 1 ? 1 : after);
    g_fault_site.store(site);
    return UG_OK;
}
int ug_msm_batch(ug_ctx* c, int count, const ug_bases* const* bases, const ug_schedule* s, const int64_t* index_shifts,
                 void* const* outs) {
    int rc = ug_msm_batch_enqueue(c, count, bases, s, index_shifts, outs);
    if (rc != UG_OK) { if (c) c->pending_msm.clear(); return rc; }
    return ug_ctx_collect(c);
}

int ug_hpoly_create(ug_ctx* c, const void* host_coefs, uint64_t n_coefs, uint32_t domain, uint32_t n_vars, ug_hpoly** out) {
    UG_TRY
    if (!c || !out || (!host_coefs && n_coefs)) throw std::invalid_argument("null argument");
    if (domain == 0 || (domain & (domain - 1))) throw std::invalid_argument("domain size is not a power of two");
    if (domain > (1u << 27)) throw std::invalid_argument("domain size above 2^27 (Fr has 2-adicity 28)");
    c->use();
    std::unique_ptr<ug_hpoly, void (*)(ug_hpoly*)> hp(new ug_hpoly(), ug_hpoly_destroy);   // frees everything on any throw below
    hp->ctx = c; hp->domain = domain; hp->nvars = n_vars;
    uint8_t* raw = nullptr;
    UG_HIP(hipMalloc(&raw, n_coefs ? (size_t)n_coefs * 44 : 4));
    bool ok = false;
    try {
        host_to_device(c, raw, host_coefs, (size_t)n_coefs * 44);
        ok = hp->mat.build(raw, n_coefs, domain, n_vars, c->stream);
    } catch (...) { hipFree(raw); throw; }
    hipFree(raw);
    if (!ok) throw std::invalid_argument("coefficient record out of range (m, row or signal index)");
    hp->ntt.init(hp->mat.logn, c->stream);
    size_t bytes = (size_t)domain * 32;
    UG_HIP(hipMalloc(&hp->a, bytes)); UG_HIP(hipMalloc(&hp->b, bytes));
    UG_HIP(hipMalloc(&hp->c, bytes)); UG_HIP(hipMalloc(&hp->t, bytes)); UG_HIP(hipMalloc(&hp->t2, bytes));
    *out = hp.release();
    UG_CATCH
}

int ug_hpoly_run(ug_hpoly* hp, const ug_dvec* w, ug_dvec* h_out) {
    UG_TRY
    if (!hp || !w || !h_out) throw std::invalid_argument("null argument");
    if (w->n < hp->nvars) throw std::invalid_argument("witness vector shorter than nVars");
    if (h_out->n < hp->domain) throw std::invalid_argument("h vector shorter than the domain");
    ug_ctx* c = hp->ctx;
    c->use();
    fault_point(UG_FAULT_HPOLY_RUN);
    ScopedTimer tm(c, &c->fft_ms);
    hipStream_t st = c->stream;
    u64 n = hp->domain;
    // S5-S6: a = A.w, b = B.w   (rows stored bit-reversed)            src/groth16.cpp:66-99
    coef_matvec(hp->a, hp->b, hp->mat, w->data, 3, st);
    if (hp->mat.logn == 0) {
        // one-point domain: every transform is the identity (n^-1 = 1, omega_2^0 = 1), nothing to fold into
        fr_mul_pointwise(hp->c, hp->a, hp->b, n, st);
        fr_h_final(h_out->data, hp->a, hp->b, hp->c, n, st);
        tm.stop();
        return UG_OK;
    }
    static const bool batched = !(measure_env("UG_NTT_BATCH") && atoi(measure_env("UG_NTT_BATCH")) == 0);      // A/B switch (-DUG_MEASURE)
    NttPass pa[NTT_MAX_PASSES], pb[NTT_MAX_PASSES], pc[NTT_MAX_PASSES];
    NttFusion cinv; cinv.in2 = hp->b; cinv.work = hp->c;
    NttFusion cfwd; cfwd.work = hp->c; cfwd.fin_a = hp->a; cfwd.fin_b = hp->b;
    const int np = hp->ntt.passes(pc, hp->t2, hp->a, /*inverse*/ true, false, /*scatter_bitrev*/ true, hp->ntt.twist, nullptr, &cinv);
    if (batched && np > 1) {
        // The three chains side by side, pass by pass: ONE launch per pass index holds the same pass of all three transforms
        // (blockIdx.y), 7 launches instead of 18 for a three-pass size, and the ends of the launches fill with the other
        // chains' workgroups. Buffers: every first pass only READS a and b (chain c's forms a o b from them, S7 :100-108), so
        // chains a and b take their intermediate passes to work buffers (hp->t and the h vector, which is written last of
        // all) and scatter their twisted coefficients back into a and b; chain c works on hp->c and leaves them in hp->t2.
        NttFusion wa; wa.work = hp->t;
        NttFusion wb; wb.work = h_out->data;
        hp->ntt.passes(pa, hp->a, hp->a, true, false, true, hp->ntt.twist, nullptr, &wa);      // S8 ifft + twist :110-128
        hp->ntt.passes(pb, hp->b, hp->b, true, false, true, hp->ntt.twist, nullptr, &wb);
        const NttPass* inv3[3] = {pa, pb, pc};
        for (int p = 0; p < np; p++) hp->ntt.launch(inv3, 3, p, st, c->stat(2));
        // S8 fft :130-140, in place on a and b; chain c's last pass also forms h = a o b - c (S9 :142-148) from the FINISHED
        // a and b, so it goes after theirs
        hp->ntt.passes(pa, hp->a, hp->a, false, false, false, nullptr, nullptr, nullptr);
        hp->ntt.passes(pb, hp->b, hp->b, false, false, false, nullptr, nullptr, nullptr);
        hp->ntt.passes(pc, h_out->data, hp->t2, false, false, false, nullptr, nullptr, &cfwd);
        const NttPass* fwd3[3] = {pa, pb, pc};
        for (int p = 0; p + 1 < np; p++) hp->ntt.launch(fwd3, 3, p, st, c->stat(2));
        hp->ntt.launch(fwd3, 2, np - 1, st, c->stat(2));
        const NttPass* last[1] = {pc};
        hp->ntt.launch(last, 1, np - 1, st, c->stat(2));
    } else {
        // S7 + the inverse half of the third chain: c = a o b is formed inside the first pass of its ifft (:100-108), whose
        // passes run on hp->c so that a and b stay intact; the twisted coefficients wait in hp->t2
        hp->ntt.transform(hp->t2, hp->a, /*inverse*/ true, false, /*scatter_bitrev*/ true, hp->ntt.twist, nullptr, st, c->stat(2), &cinv);
        // S8: ifft, twist by omega_2n^i (with 1/n folded in), fft          :110-140
        u32* polys[2] = {hp->a, hp->b};
        for (int p = 0; p < 2; p++) {
            hp->ntt.transform(hp->t, polys[p], /*inverse*/ true, false, /*scatter_bitrev*/ true, hp->ntt.twist, nullptr, st, c->stat(2));
            hp->ntt.transform(polys[p], hp->t, /*inverse*/ false, false, false, nullptr, nullptr, st, c->stat(2));
        }
        // the forward half of the third chain, with S9 (h = a o b - c, to plain integers, :142-148) inside its last pass
        hp->ntt.transform(h_out->data, hp->t2, /*inverse*/ false, false, false, nullptr, nullptr, st, c->stat(2), &cfwd);
    }
    tm.stop();
    UG_CATCH
}

// coset evaluations of one of the three polynomials (0 = A.w, 1 = B.w, 2 = (A.w) o (B.w)) into `out`
// (domain elements, device form): the unit of work a rank of a sharded prover takes
int ug_hpoly_chain(ug_hpoly* hp, const ug_dvec* w, int which, ug_dvec* out) {
    UG_TRY
    if (!hp || !w || !out) throw std::invalid_argument("null argument");
    if (which < 0 || which > 2) throw std::invalid_argument("polynomial index out of range");
    if (w->n < hp->nvars) throw std::invalid_argument("witness vector shorter than nVars");
    if (out->n < hp->domain) throw std::invalid_argument("output vector shorter than the domain");
    ug_ctx* c = hp->ctx;
    c->use();
    ScopedTimer tm(c, &c->fft_ms);
    hipStream_t st = c->stream;
    u64 n = hp->domain;
    if (which == 2 && hp->mat.logn == 0) {
        coef_matvec(hp->a, hp->b, hp->mat, w->data, 3, st);
        fr_mul_pointwise(out->data, hp->a, hp->b, n, st);                // one-point domain: the chain is the identity
        tm.stop();
        return UG_OK;
    }
    if (which == 2) {
        coef_matvec(hp->a, hp->b, hp->mat, w->data, 3, st);
        NttFusion cinv; cinv.in2 = hp->b; cinv.work = hp->c;             // c = a o b inside the first pass
        hp->ntt.transform(hp->t, hp->a, /*inverse*/ true, false, /*scatter_bitrev*/ true, hp->ntt.twist, nullptr, st, c->stat(2), &cinv);
    } else {
        coef_matvec(hp->a, hp->b, hp->mat, w->data, 1 << which, st);
        u32* src = which ? hp->b : hp->a;
        hp->ntt.transform(hp->t, src, /*inverse*/ true, false, /*scatter_bitrev*/ true, hp->ntt.twist, nullptr, st, c->stat(2));
    }
    hp->ntt.transform(out->data, hp->t, /*inverse*/ false, false, false, nullptr, nullptr, st, c->stat(2));
    tm.stop();
    sync_and_resolve(c);                         // the caller hands the buffer to other libraries (RCCL): complete on return
    UG_CATCH
}
// h[first .. first + count) = plain(a o b - c) from the matching slices of the three coset evaluation vectors
int ug_hpoly_combine(ug_hpoly* hp, const ug_dvec* a, const ug_dvec* b, const ug_dvec* cc, uint64_t first, uint64_t count,
                     ug_dvec* h_out) {
    UG_TRY
    if (!a || !b || !cc || !h_out) throw std::invalid_argument("null argument");
    if (a->n < count || b->n < count || cc->n < count) throw std::invalid_argument("slice vectors shorter than count");
    if (first + count > h_out->n) throw std::invalid_argument("h slice outside the h vector");
    ug_ctx* c = hp ? hp->ctx : h_out->ctx;      // (a rank that runs no chain keeps no ug_hpoly: the combine needs none)
    c->use();
    ScopedTimer tm(c, &c->fft_ms);
    fr_h_final(h_out->data + first * 8, a->data, b->data, cc->data, count, c->stream);
    tm.stop();
    UG_CATCH
}

int ug_hpoly_debug_abc(ug_hpoly* hp, void* ha, void* hb, void* hc) {
    UG_TRY
    if (!hp) throw std::invalid_argument("null argument");
    ug_ctx* c = hp->ctx;
    c->use();
    void* host[3] = {ha, hb, hc};
    u32* dev[3] = {hp->a, hp->b, hp->c};
    // ug_hpoly_run folds the third chain's last pass into h; its coset evaluations are re-made from the twisted coefficients
    if (hc) hp->ntt.transform(hp->c, hp->t2, /*inverse*/ false, false, false, nullptr, nullptr, c->stream);
    for (int p = 0; p < 3; p++) {
        if (!host[p]) continue;
        fr_to_mont256(hp->t, dev[p], hp->domain, c->stream);
        UG_HIP(hipMemcpyAsync(host[p], hp->t, (size_t)hp->domain * 32, hipMemcpyDeviceToHost, c->stream));
        UG_HIP(hipStreamSynchronize(c->stream));
    }
    UG_CATCH
}

void ug_hpoly_destroy(ug_hpoly* hp) {
    if (!hp) return;
    hipSetDevice(hp->ctx->device);
    alloc_epoch_bump();
    hipFree(hp->a); hipFree(hp->b); hipFree(hp->c); hipFree(hp->t); hipFree(hp->t2);
    hp->mat.release(); hp->ntt.release();
    delete hp;
}

int ug_fr_ntt(ug_ctx* c, void* host_data, int logn, int inverse) {
    UG_TRY
    if (!c || !host_data) throw std::invalid_argument("null argument");
    if (logn < 0 || logn > 27) throw std::invalid_argument("logn out of range");
    c->use();
    if (c->raw_ntt.logn != logn) c->raw_ntt.init(logn, c->stream);
    size_t bytes = ((size_t)1 << logn) * 32;
    u32 *x = nullptr, *y = nullptr;
    UG_HIP(hipMalloc(&x, bytes)); UG_HIP(hipMalloc(&y, bytes));
    UG_HIP(hipMemcpyAsync(x, host_data, bytes, hipMemcpyHostToDevice, c->stream));
    fr_from_mont256(x, x, (u64)1 << logn, c->stream);
    if (logn == 0) UG_HIP(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, c->stream));
    else c->raw_ntt.transform(y, x, inverse != 0, /*gather_bitrev*/ true, false, nullptr, inverse ? c->raw_ntt.ninv : nullptr, c->stream);
    fr_to_mont256(y, y, (u64)1 << logn, c->stream);
    UG_HIP(hipMemcpyAsync(host_data, y, bytes, hipMemcpyDeviceToHost, c->stream));
    UG_HIP(hipStreamSynchronize(c->stream));
    hipFree(x); hipFree(y);
    UG_CATCH
}

int ug_field_op(ug_ctx* c, int field, int op, void* out, const void* a, const void* b, uint64_t n) {
    UG_TRY
    if (!c || !out || !a || !b) throw std::invalid_argument("null argument");
    if (field != UG_FIELD_FR && field != UG_FIELD_FQ) throw std::invalid_argument("unknown field");
    if (op < 0 || op > 3) throw std::invalid_argument("unknown op");
    c->use();
    size_t bytes = (size_t)n * 32;
    u32 *da = nullptr, *db = nullptr, *dout = nullptr;
    UG_HIP(hipMalloc(&da, bytes ? bytes : 4)); UG_HIP(hipMalloc(&db, bytes ? bytes : 4)); UG_HIP(hipMalloc(&dout, bytes ? bytes : 4));
    if (n) {
        UG_HIP(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c->stream));
        UG_HIP(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, c->stream));
        f_op_mont256(field, op, dout, da, db, n, c->stream);
        UG_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    UG_HIP(hipStreamSynchronize(c->stream));
    hipFree(da); hipFree(db); hipFree(dout);
    UG_CATCH
}

int ug_synth_points(ug_ctx* c, int g2, const void* generator_record, uint64_t seed, uint64_t n, void* host_out) {
    UG_TRY
    if (!c || !generator_record || (!host_out && n)) throw std::invalid_argument("null argument");
    c->use();
    size_t rec = g2 ? 128 : 64;
    u32 gen[32];
    memcpy(gen, generator_record, rec);
    const u64 CH = (u64)1 << 22;                       // stage through a bounded device buffer
    u32* dev = nullptr;
    UG_HIP(hipMalloc(&dev, (size_t)(n < CH ? (n ? n : 1) : CH) * rec));
    for (u64 done = 0; done < n; done += CH) {
        u64 m = n - done < CH ? n - done : CH;
        synth_points(g2 != 0, dev, gen, seed + done, m, c->stream);
        UG_HIP(hipMemcpy((uint8_t*)host_out + done * rec, dev, (size_t)m * rec, hipMemcpyDeviceToHost));
    }
    hipFree(dev);
    UG_CATCH
}

// ---- captured launch sequences ---------------------------------------------------------------------------------------------
// A created prover's proof is one fixed sequence of 60-75 launches on two streams, none of which depends on a host read-back:
// captured once per witness buffer, replayed with one hipGraphLaunch. ug_graph_begin puts the context's stream into capture
// (relaxed mode: other host threads -- the witness staging lanes -- go on using the runtime) and forks ctx2's stream off it;
// everything the library queues on either context until ug_graph_end becomes a node: kernels, memsets, the result copies into
// pinned memory, and the timing events as event-record nodes (dev_common.hpp: record_in_capture), so that the MSM | FFT split and the per-kernel statistics
// keep working under replay. Nothing runs during the capture. ug_graph_end joins the streams, instantiates, and takes the
// products that were queued (ug_msm_*_enqueue) out of the contexts; ug_graph_launch puts them back and launches: the caller
// then collects as after the eager calls -- ug_ctx_collect on the FIRST context first (the graph runs on its stream).
int ug_graph_begin(ug_ctx* c, ug_ctx* c2) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    if (c2 && c2->device != c->device) throw std::invalid_argument("contexts on different devices");
    if (c->recording || (c2 && c2->recording)) throw std::logic_error("a capture is already in progress on this context");
    if (!c->pending_msm.empty() || (c2 && !c2->pending_msm.empty())) throw std::logic_error("collect the queued MSMs first (ug_ctx_collect)");
    c->use();
    sync_and_resolve(c);                                   // pending spans and statistics of eager work: accounted before the switch
    if (c2) sync_and_resolve(c2);
    std::unique_ptr<ug_graph> g(new ug_graph());
    g->c[0] = c; g->c[1] = c2;
    UG_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    g->capturing = true;
    ug_graph* raw = g.release();
    c->recording = raw;
    for (int k = 0; k < 4; k++) c->stats[k].capture = &raw->cap[0][k];
    if (c2) {
        c2->recording = raw;
        for (int k = 0; k < 4; k++) c2->stats[k].capture = &raw->cap[1][k];
        hipError_t e = hipEventRecord(c->order_event, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c2->stream, c->order_event, 0);      // ctx2's stream joins the capture
        if (e != hipSuccess) { ug_graph_abort(c); UG_HIP(e); }
    }
    UG_CATCH
}
static void graph_unhook(ug_graph* g) {
    for (int q = 0; q < 2; q++) {
        if (!g->c[q]) continue;
        g->c[q]->recording = nullptr;
        for (int k = 0; k < 4; k++) g->c[q]->stats[k].capture = nullptr;
    }
}
void ug_graph_abort(ug_ctx* c) {
    if (!c || !c->recording) return;
    ug_graph* g = c->recording;
    (void)hipSetDevice(c->device);
    graph_unhook(g);
    for (int q = 0; q < 2; q++) if (g->c[q]) g->c[q]->pending_msm.clear();
    if (g->capturing) {
        hipGraph_t broken = nullptr;
        (void)hipStreamEndCapture(g->c[0]->stream, &broken);
        if (broken) (void)hipGraphDestroy(broken);
        (void)hipGetLastError();
        g->capturing = false;
    }
    ug_graph_destroy(g);
}
int ug_graph_end(ug_ctx* c, ug_graph** out) {
    UG_TRY
    if (!c || !out) throw std::invalid_argument("null argument");
    if (!c->recording || c->recording->c[0] != c) throw std::logic_error("no capture was begun on this context");
    ug_graph* g = c->recording;
    c->use();
    try {
        if (g->c[1]) {                                     // join: the first stream waits for everything queued on the second
            UG_HIP(hipEventRecord(g->c[1]->order_event, g->c[1]->stream));
            UG_HIP(hipStreamWaitEvent(c->stream, g->c[1]->order_event, 0));
        }
        hipError_t e = hipStreamEndCapture(c->stream, &g->graph);
        g->capturing = false;
        UG_HIP(e);
        if (!g->graph) throw std::runtime_error("stream capture produced no graph");
        UG_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
        (void)hipGraphGetNodes(g->graph, nullptr, &g->nodes);
    } catch (...) { ug_graph_abort(c); throw; }
    graph_unhook(g);
    for (int q = 0; q < 2; q++) if (g->c[q]) { g->pend[q].swap(g->c[q]->pending_msm); g->c[q]->pending_msm.clear(); }
    g->epoch = alloc_epoch();
    *out = g;
    UG_CATCH
}
// 1 while every buffer the captured launches refer to is still where it was (no per-proof device buffer of this process has been
// allocated or released since the capture ended)
int ug_graph_valid(const ug_graph* g) { return g && g->exec && g->epoch == alloc_epoch() ? 1 : 0; }
uint64_t ug_graph_nodes(const ug_graph* g) { return g ? g->nodes : 0; }
int ug_graph_launch(ug_graph* g) {
    UG_TRY
    if (!g || !g->exec) throw std::invalid_argument("null argument");
    if (!ug_graph_valid(g)) throw std::logic_error("the captured launch sequence is stale (device buffers were re-allocated since)");
    ug_ctx* c = g->c[0];
    c->use();
    for (int q = 0; q < 2; q++)
        if (g->c[q] && (!g->c[q]->pending_msm.empty() || g->c[q]->recording)) throw std::logic_error("collect the queued MSMs first (ug_ctx_collect)");
    // a launch whose event pairs have not been read yet would be overwritten: account it first (a caller that collects after
    // every launch never waits here)
    for (ug_graph* l : c->launched) if (l == g) { sync_and_resolve(c); break; }
    for (int q = 0; q < 2; q++) if (g->c[q]) g->c[q]->pending_msm = g->pend[q];
    hipError_t e = hipGraphLaunch(g->exec, c->stream);
    if (e != hipSuccess) { for (int q = 0; q < 2; q++) if (g->c[q]) g->c[q]->pending_msm.clear(); UG_HIP(e); }
    c->launched.push_back(g);
    UG_CATCH
}
void ug_graph_destroy(ug_graph* g) {
    if (!g) return;
    if (g->c[0]) {
        (void)hipSetDevice(g->c[0]->device);
        // a launch may still run, and the context may still want to read its event pairs
        if (!g->capturing) (void)hipStreamSynchronize(g->c[0]->stream);
        auto& l = g->c[0]->launched;
        for (size_t i = 0; i < l.size();) { if (l[i] == g) l.erase(l.begin() + (long)i); else i++; }
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    for (auto& t : g->spans) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
    for (int q = 0; q < 2; q++) for (int k = 0; k < 4; k++) for (auto& sp : g->cap[q][k]) { if (sp.e0) hipEventDestroy(sp.e0); if (sp.e1) hipEventDestroy(sp.e1); }
    (void)hipGetLastError();
    delete g;
}

int ug_ctx_timings(ug_ctx* c, double* msm_ms, double* fft_ms, int reset) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    c->use();
    sync_and_resolve(c);
    if (msm_ms) *msm_ms = c->msm_ms;
    if (fft_ms) *fft_ms = c->fft_ms;
    if (reset) { c->msm_ms = 0; c->fft_ms = 0; }
    UG_CATCH
}
int ug_ctx_kernel_stats(ug_ctx* c, int which, double* avg_ms, uint64_t* launches, uint64_t* entries, int reset) {
    UG_TRY
    if (!c) throw std::invalid_argument("null argument");
    if (which < 0 || which > 3) throw std::invalid_argument("kernel stats: 0 = G1 accumulation, 1 = G2 accumulation, 2 = NTT pass, 3 = G1 group accumulation");
    c->use();
    sync_and_resolve(c);
    MsmStats& st = c->stats[which];
    if (avg_ms) *avg_ms = st.launches ? st.accumulate_ms / (double)st.launches : 0.0;
    if (launches) *launches = st.launches;
    if (entries) *entries = st.entries;
    if (reset) { st.accumulate_ms = 0; st.launches = 0; st.entries = 0; c->kernel_stats_on = true; }
    UG_CATCH
}

}  // extern "C"
