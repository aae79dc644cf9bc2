// internal.hpp -- host-side interfaces between the translation units of libultragroth_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <vector>
#include "ec.hpp"

namespace ug {

// ---- ntt.hip ------------------------------------------------------------------------------------------
Fr fr_root_of_unity(int s);
void bitrev_copy(u32* out, const u32* in, int logn, hipStream_t stream);

// Element-wise work folded into the first / last pass of a transform (the H-polynomial block's c = a o b and
// h = a o b - c, src/groth16.cpp:100-108,142-148) and a separate work buffer so that the inputs stay intact:
//   in2    first pass: the transform's input element is in[i] * in2[i]
//   work   intermediate passes run in place here (instead of on `in` / `out`)
//   fin_a, fin_b   last pass: out[i] = plain integer of fin_a[i] * fin_b[i] - x[i] (32-byte plain, not device form)
struct NttFusion { const u32* in2 = nullptr; u32* work = nullptr; const u32* fin_a = nullptr; const u32* fin_b = nullptr; };

// one pass of a transform as the host plans it (ntt.hip turns it into kernel arguments)
struct NttPass {
    const u32* in; u32* out; const u32* tw; const u32* post; const u32* post_const; const u32* in2; const u32* fin_a; const u32* fin_b;
    int logn, s0, k, j, gather_bitrev, scatter_bitrev;
};
constexpr int NTT_MAX_PASSES = 8;

struct NttPlan {
    int logn = -1;
    u32* tw_fwd = nullptr;    // stage-major twiddles omega_{2^(s+1)}^j (ntt.hip: 12-word Montgomery or 20-word Shoup entries)
    u32* tw_inv = nullptr;    // omega_n^-i,  i < n/2
    u32* twist = nullptr;     // n^-1 * omega_{2n}^i at place bitrev(i), i < n   (coset twist with the ifft scale folded in; in the
                              // order of the scattering pass that applies it)
    u32* ninv = nullptr;      // n^-1 (one element)
    bool shoup = true;        // twiddle tables hold (w, floor(w 2^261 / q)) pairs and the butterflies make Shoup products (ntt.hip)
    void init(int logn, hipStream_t stream);
    void release();
    // DIT transform; see ntt.hip for the buffer rules. post / post_const are optional multipliers
    // applied in the last pass: out[i] *= post[i], or out[i] *= *post_const.
    // stats (optional): every pass launch is bracketed with an event pair (KernelStats::collect after a stream sync).
    // fuse (optional): see NttFusion.
    void transform(u32* out, const u32* in, bool inverse, bool gather_bitrev, bool scatter_bitrev,
                   const u32* post, const u32* post_const, hipStream_t stream, struct MsmStats* stats = nullptr,
                   const struct NttFusion* fuse = nullptr) const;
    // the same in two steps, for callers that run several transforms of this size side by side: passes() fills `list`
    // (NTT_MAX_PASSES entries) and returns the pass count -- the same for every transform of the plan --, launch() runs pass
    // p of up to three such lists in one kernel launch (logn >= 1)
    int passes(NttPass* list, u32* out, const u32* in, bool inverse, bool gather_bitrev, bool scatter_bitrev,
               const u32* post, const u32* post_const, const struct NttFusion* fuse = nullptr) const;
    void launch(const NttPass* const* lists, int count, int p, hipStream_t stream, struct MsmStats* stats = nullptr) const;
    ~NttPlan() { release(); }
};

// Every allocation or release of a per-proof device buffer (schedules, MSM workspaces, the sort's tables) moves this counter: a
// captured launch sequence (ug_graph) holds raw pointers into those buffers and is only replayed while the counter stands where
// it stood when the capture ended.
uint64_t alloc_epoch();
void alloc_epoch_bump();

// ---- msm.hip ------------------------------------------------------------------------------------------
// Bucket classes (many-device provers, DESIGN.md section 7: the witness products sharded by BUCKET instead of by base point).
// A schedule with classes keeps only the (scalar, window) digits whose bucket b = |digit| - 1 has its residue b mod Q,
// Q = 2^q_log, in [r0, r0 + cnt): one bucket set of 2^(c-1) / Q buckets per owned residue, local id b >> q_log. The ranks of
// a node own disjoint residue ranges of the SAME scalars and the SAME (whole) base tables, so a rank sorts, accumulates and
// reduces its share of the entries at the full problem's window width -- no extra window, a share of the buckets -- and the
// weights come back on the host: a digit of magnitude b + 1 = Q (k + 1) - (Q - 1 - r), k = b >> q_log, r = b mod Q, so the sum
// of a residue's set is Q * S1 - (Q - 1 - r) * S0 with S1 = sum (k + 1) B_k (what the running-sum reduction yields) and
// S0 = sum B_k (its running sum, emitted beside it).
// The lowest `specials` bucket ids of every window -- the digits 1 .. specials, which small witness values (bits, bytes,
// counters) fill with up to millions of entries: one residue's owner would get all of bucket 0 -- are not owned by residue
// but by SCALAR RANGE: the schedule keeps such a digit iff its scalar lies in [sp_lo, sp_hi) (local indices); they get
// bucket ids of their own behind the regular sets, and their weighted sum (sum (b + 1) X_b) is a third output.
struct BucketClasses {
    int q_log = 0;               // 0: no classes, the schedule owns every bucket
    u32 r0 = 0, cnt = 1;         // owned residues [r0, r0 + cnt) of Q = 2^q_log
    u32 specials = 0;            // bucket ids below this are owned by scalar range
    u32 sp_lo = 0, sp_hi = 0;    // ... the scalars [sp_lo, sp_hi) of the schedule
    bool on() const { return q_log > 0; }
};
constexpr u32 MSM_MAX_SPECIALS = 64;     // (one wave sums a window's special buckets)

// How a scalar's window digits become (bucket key, entry) pairs: the kernel argument of the digit-making kernels (sort.hip)
struct DigitPlan {
    u64 n; int c, windows; u32 buckets, sentinel; int tables;      // buckets: per set; sentinel = total_buckets(): the key of a digit that is not kept
    int q_log; u32 r0, cnt, specials, sp_lo, sp_hi, special_base;  // bucket classes (q_log = 0: none); special_base = first special bucket id
};

struct MsmGeometry {
    u64 n = 0;          // number of scalars
    int c = 0;          // window bits
    int windows = 0;    // digits per scalar: ceil(255 / c)
    u32 buckets = 0;    // per bucket set: 2^(c-1), with classes 2^(c-1-q_log)
    bool tables = false;  // fixed-base window tables: digit j of scalar i multiplies 2^(c j) P_i, read from table j, so
                          // every digit of every window lands in ONE bucket set (no Horner, 1/windows of the reduction)
    BucketClasses cls;
    static MsmGeometry choose(u64 n, int force_c = 0);
    static MsmGeometry choose_tables(u64 n, int c);
    static int table_window(u64 n);                                    // cost-model window width for the tables mode
    void set_classes(const BucketClasses& k);                          // after choose / choose_tables (divides `buckets`)
    int window_sets() const { return tables ? 1 : windows; }           // Horner steps on the host
    int class_sets() const { return cls.on() ? (int)cls.cnt : 1; }
    int bucket_windows() const { return window_sets() * class_sets(); }      // bucket sets the reduction sees: set (w, j) = w * class_sets + j
    u64 special_buckets() const { return cls.on() ? (u64)window_sets() * cls.specials : 0; }     // ids behind the regular sets
    u64 total_buckets() const { return (u64)bucket_windows() * buckets + special_buckets(); }
    // points of a product's result block: S1 per set; with classes also S0 per set and one weighted sum of the specials per window
    int result_points() const { return cls.on() ? 2 * bucket_windows() + (cls.specials ? window_sets() : 0) : bucket_windows(); }
    // entries this schedule is expected to keep (uniform digits): what the segment length is chosen for
    u64 expected_entries() const { return cls.on() ? ((n * (u64)windows) >> cls.q_log) * cls.cnt : n * (u64)windows; }
    DigitPlan digit_plan() const {
        DigitPlan d;
        d.n = n; d.c = c; d.windows = windows; d.buckets = buckets; d.sentinel = (u32)total_buckets(); d.tables = tables ? 1 : 0;
        d.q_log = cls.q_log; d.r0 = cls.r0; d.cnt = cls.on() ? cls.cnt : 1; d.specials = cls.on() ? cls.specials : 0;
        d.sp_lo = cls.sp_lo; d.sp_hi = cls.sp_hi; d.special_base = (u32)((u64)bucket_windows() * buckets);
        return d;
    }
};
constexpr int TABLE_INDEX_BITS = 27;                                   // entry = index | table << 27 | sign << 31
constexpr int TABLE_MIN_C = 16, TABLE_MAX_C = 24;                      // <= 16 tables (4 bits), <= 2^23 buckets

struct HeavyBucket { u32 bucket, first_seg, last_seg, pad; };  // a bucket cut into many segment pieces

// ---- sort.hip: the hand-written LSD radix partition that groups a schedule's pairs by bucket ------------------------
int radix_plan(int bits, int* shift, int* bins_log);          // passes; low bits first
// the unsorted pair form: keys / vals [w * n + i] = window w of scalar i (only for scalars of more than 16 windows, i.e. tiny MSMs)
void digit_pairs(const u32* scalars, const DigitPlan& plan, u32* keys, u32* vals, hipStream_t stream);
struct RadixSorter {
    u32* lookback = nullptr;      // tiles x 256 status words of the pass in flight
    u32* small = nullptr;         // 4 x 256 bin counts / starts, one tile counter per pass
    u64 tiles_cap = 0;
    void reserve(u64 n_pairs, int ipt_min);
    void release();
    // see sort.hip; returns which of the two buffer pairs holds the sorted pairs
    int sort(const u32* scalars, const MsmGeometry& geo, u32 sentinel, u64 n_pairs, int bits,
             u32* const buf_keys[2], u32* const buf_vals[2], u32* error_flag, hipStream_t stream, u32* n_valid_out = nullptr,
             bool* dropped_out = nullptr);
    ~RadixSorter() { release(); }
};

// Signed-digit decomposition of n scalars, grouped by (window, bucket): shared by every base set that
// is multiplied by the same scalars (A, B1, B2 and C all use the witness, src/groth16.cpp:55-64).
struct MsmSchedule {
    MsmGeometry geo;
    const u32* keys = nullptr;    // sorted bucket ids (sentinel = total_buckets at the end)
    const u32* vals = nullptr;    // sorted entries: scalar index | table << 27 | sign << 31
    const u32* tkeys = nullptr;   // lane-transposed copies of keys / vals (see msm.hip: transposed_index)
    const u32* tvals = nullptr;
    u32* bucket_start = nullptr;  // per bucket: first entry
    u32* bucket_count = nullptr;  // per bucket: number of entries
    u32* small_list = nullptr;    // buckets cut into 2 .. 32 segment pieces (meta[4] of them), any order
    int log_seg = 0;              // entries per lane of the segmented accumulation = 2^log_seg ...
    int log_seg_tail = 0;         // ... and 2^log_seg_tail for the segments of the last eighth of the entries (msm.hip: SegMap)
    u32* heavy_list = nullptr;    // device: HeavyBucket[heavy_cap]
    u32 heavy_cap = 0;
    u32* medium_list = nullptr;   // device: HeavyBucket[heavy_cap] for buckets of FIX_MAX < pieces <= MEDIUM_MAX
    u32* heavy_offsets = nullptr; // device: first task of each heavy bucket (n_heavy + 1 entries)
    u32* meta = nullptr;          // device: [n_heavy, n_valid, n_heavy_tasks, n_medium, n_small, ., ., sort failure flag] -- read by the
                                  // kernels; the flag travels to the host with every product's result block
    // workspace
    u32 *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;       // (the library sort's scratch: UG_SORT=cub, kept for A/B runs)
    RadixSorter sorter;
    u64 capacity_n = 0; u64 capacity_buckets = 0;
    void reserve(const MsmGeometry& g);
    void build(const u32* scalars_dev, const MsmGeometry& g, hipStream_t stream);   // scalars: n x 32 B plain integers
    void release();
    ~MsmSchedule() { release(); }
};

struct MsmWorkspace {
    u32* bucket_pts = nullptr;   // total_buckets XYZZ records
    u32* chunk_pts = nullptr;    // reduction scratch
    u32* chunk_pts2 = nullptr;
    u32* slot_pts = nullptr;     // two partial sums per segment of the accumulation
    u32* task_pts = nullptr;     // partial sums of the heavy-bucket tasks
    size_t bucket_bytes = 0, chunk_bytes = 0, slot_bytes = 0, task_bytes = 0;
    void reserve(const MsmGeometry& g, bool g2, u64 n_segments, u32 n_heavy_tasks, int sets = 1);
    void release();
    ~MsmWorkspace() { release(); }
};

// An event pair recorded INSIDE a captured launch sequence (dev_common.hpp record_in_capture: an event-record
// node of the graph, re-recorded by every launch of it), with what the elapsed time is accounted to once a launch has completed.
struct CapturedSpan { hipEvent_t e0 = nullptr, e1 = nullptr; u64 units = 0; };

struct MsmStats {                 // HIP-event timing of one kernel's launches (bucket accumulation G1 / G2, NTT passes)
    static constexpr int SLOTS = 64;                // launches that may be in flight before collect()
    static constexpr int MAX_BATCH = 8;             // MSMs queued back to back by ug_msm_batch
    hipEvent_t ev0[SLOTS] = {}, ev1[SLOTS] = {};
    u64 slot_entries[SLOTS] = {};
    int pending = 0;
    double accumulate_ms = 0; u64 launches = 0; u64 entries = 0;
    // while the stream is being captured into a graph (ug_graph_begin .. _end) the pairs go here instead, as external event
    // records that belong to the graph; account() adds one completed launch of the graph
    std::vector<CapturedSpan>* capture = nullptr;
    void account(const std::vector<CapturedSpan>& spans);
    void create();
    void destroy();
    void collect();                                 // after the stream has been synchronised
    void collect_ready();                           // the launches that have finished, without waiting
    int begin(hipStream_t stream, u64 units);       // records the start event; returns the slot (-1: none free, untimed)
    void end(int slot, hipStream_t stream);
};
typedef MsmStats KernelStats;

// An MSM whose kernels and result copy are queued on the stream; the window sums arrive in `host` (pinned memory,
// bucket_windows * PT_WORDS words) once the stream is synchronised. Several may be queued back to back: the stream
// orders their use of the shared workspace.
struct MsmPending {
    bool g2 = false;
    bool empty = true;           // nothing was queued: the sum is the point at infinity
    int c = 0, window_sets = 0, class_sets = 1;
    BucketClasses cls;           // cls.on(): the block holds [S1 per set | S0 per set | specials per window] (MsmGeometry::result_points)
    u32* host = nullptr;
};
constexpr size_t MSM_PENDING_WORDS = 128 * 72;      // room for the largest result block (<= 127 G2 points); the LAST word
                                                    // receives the schedule's failure flag (meta[7])
MsmPending msm_enqueue_g1(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream,
                          MsmStats* stats, u32* pinned_host);
MsmPending msm_enqueue_g2(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream,
                          MsmStats* stats, u32* pinned_host);
// up to MSM_BATCH_MAX products over one schedule: one accumulation launch each, the latency-bound tail kernels once for all
constexpr int MSM_BATCH_MAX = 4;
void msm_enqueue_batch_g1(const MsmSchedule& s, MsmWorkspace& ws, int count, const u32* const* bases, const u64* n_bases, const int64_t* delta,
                          hipStream_t stream, MsmStats* stats, u32* const* pinned_host, MsmPending* pend);
// phase (msm_enqueue_batch_g2, msm_enqueue_group_g1): the whole call, or its two halves made one after the other with the same
// arguments -- the accumulation launches, and everything behind them (on a stream that is ordered behind the accumulation)
constexpr int MSM_PHASE_ALL = 0, MSM_PHASE_ACCUMULATE = 1, MSM_PHASE_TAIL = 2;
void msm_enqueue_batch_g2(const MsmSchedule& s, MsmWorkspace& ws, int count, const u32* const* bases, const u64* n_bases, const int64_t* delta,
                          hipStream_t stream, MsmStats* stats, u32* const* pinned_host, MsmPending* pend, int phase = MSM_PHASE_ALL);
// `members` (2 or 3) products over ONE interleaved array of members-point records (n_slots records per window table); pinned_host /
// pend: one per member
void msm_enqueue_group_g1(const MsmSchedule& s, MsmWorkspace& ws, int members, const u32* bases, u64 n_slots, int64_t delta, hipStream_t stream,
                          MsmStats* stats, u32* const* pinned_host, MsmPending* pend, int phase = MSM_PHASE_ALL);
// 64-byte records of one member set -> record (slot0 + i) * members + member of a group array
void interleave_points_g1(u32* dst, const u32* src, u64 n, int members, int member, u64 slot0, hipStream_t stream);
G1XYZZ msm_collect_g1(const MsmPending& p);
G2XYZZ msm_collect_g2(const MsmPending& p);

// sum over the schedule's scalars (local index i) of scalar_i * base[i + delta]; bases is a device array
// of n_bases packed affine records in device Montgomery form ((0,0) = infinity); entries whose base
// index falls outside [0, n_bases) are skipped. Result is a host XYZZ point.
G1XYZZ msm_g1(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream, MsmStats* stats);
G2XYZZ msm_g2(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream, MsmStats* stats);

// zkey affine records (reference Montgomery form, R = 2^256) -> device form, in place on the device
void convert_points_g1(u32* pts, u64 n, hipStream_t stream);
void convert_points_g2(u32* pts, u64 n, hipStream_t stream);

// pts = `tables` tables of n records, table 0 filled: table j = 2^(c j) * table 0 (fixed-base window tables)
// (first, count: only the tables of the points [first, first + count) -- one piece of a deferred build; default: all of them)
void build_window_tables(bool g2, u32* pts, u64 n, int c, int tables, hipStream_t stream, u64 first = 0, u64 count = ~(u64)0);

// bench / test tooling: out[i] = (seed + i) * G as zkey-format records (device buffer); G given as a host record
void synth_points(bool g2, u32* out_dev, const u32* gen_record_host, u64 seed, u64 n, hipStream_t stream);

// ---- hpoly.hip ----------------------------------------------------------------------------------------
struct CoefMatrix {
    u64 ncoefs = 0; u32 domain = 0; int logn = 0;
    u32* row_ptr = nullptr;     // 2*domain + 1 offsets, rows = m * domain + c
    u32* sig = nullptr;         // signal index per entry
    u32* val = nullptr;         // packed coefficient per entry (device form, pre-scaled so that
                                //   mul(w_plain, val) = w * coef in device Montgomery form)
    // raw44: ncoefs packed 44-byte records {u32 m, u32 c, u32 s, 32-byte coef} already on the device
    // (src/groth16.hpp:41-49). Returns false when a record is out of range.
    bool build(const uint8_t* raw44_dev, u64 ncoefs, u32 domain, u32 nvars, hipStream_t stream);
    void release();
    ~CoefMatrix() { release(); }
};

// a_br[bitrev(c)] = sum coef * w  over m = 0 rows, b_br likewise for m = 1 (device form, packed)
void coef_matvec(u32* a_br, u32* b_br, const CoefMatrix& m, const u32* wtns_dev, int mask, hipStream_t stream);   // mask bit 0: A rows, bit 1: B rows
// out[i] = x[i] * y[i]
void fr_mul_pointwise(u32* out, const u32* x, const u32* y, u64 n, hipStream_t stream);
// h[i] = plain integer of (a[i] * b[i] - c[i])      (src/groth16.cpp:142-148)
void fr_h_final(u32* h, const u32* a, const u32* b, const u32* c, u64 n, hipStream_t stream);
// element-wise format changes, n elements of 32 bytes
void fr_from_mont256(u32* out, const u32* in, u64 n, hipStream_t stream);
void fr_to_mont256(u32* out, const u32* in, u64 n, hipStream_t stream);
void f_op_mont256(int which, int op, u32* out, const u32* a, const u32* b, u64 n, hipStream_t stream);
void gather_elements(u32* out, const u32* src, const u32* index_dev, u64 n, u64 src_n, hipStream_t stream);
void scatter_elements(u32* dst, const u32* index_dev, const u32* values_dev, u64 n, u64 dst_n, hipStream_t stream);
// dst[w_idx[i]] = push[p_idx[i]] for i < n in order (last write wins); push = [table[0] | table[1 + chunks[j]], j < n_chunks |
// table[1..]]; all pointers on the device, indices already validated; last_scratch: one zeroed u32 per element of dst
// table_dev: 1 + 2 L elements, [0] = rand (plain) on entry; fills inv2 = 1/(i + rand) and prod = freq[i] * inv2[i] (plain)
void lookup_table(u32* table_dev, const u32* freq_dev, u64 L, hipStream_t stream);
void apply_lookup(u32* dst, u32* last_scratch, const u32* w_idx, const u32* p_idx, u64 n, const u32* chunks, u64 n_chunks,
                  const u32* table, hipStream_t stream);

}  // namespace ug
