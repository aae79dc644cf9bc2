/*
 * ultragroth_hip.h -- inner C-ABI of libultragroth_hip.so: the MI355X (gfx950) replacement for the
 * arithmetic layer that rarimo/ultragroth's prover reaches through the un-vendored iden3/ffiasm
 * submodule. Plain pointers and sizes only; no exceptions cross this boundary (every function returns
 * UG_OK or UG_ERROR and ug_last_error() gives the message for the calling thread).
 *
 * Reference interface each group replaces (paths relative to the reference repository):
 *
 *   ug_bases_*  + ug_schedule_* + ug_msm_g1 / ug_msm_g2
 *       Curve::multiMulByScalarMSM(Point& r, PointAffine* bases, uint8_t* scalars, 32, n)
 *       call sites src/groth16.cpp:55,58,61,64,154 ; src/ultra_groth.cpp:168,201,214,227,234,322
 *       (the base arrays are the zkey sections handed to makeProver, src/groth16.cpp:9-46,
 *        src/prover.cpp:162-178)
 *
 *   ug_hpoly_create / ug_hpoly_run
 *       FFT<Fr>(2 * domainSize) ctor            src/groth16.hpp:109
 *       the a/b/c block of Prover::prove        src/groth16.cpp:66-148
 *         (zero, sparse coefficient scatter-add, a o b, 3 x {ifft, root(.) twist, fft}, a o b - c,
 *          fromMontgomery), twin in src/ultra_groth.cpp:243-320
 *
 *   ug_fr_ntt            FFT<Fr>::fft / ifft on a host buffer    (src/groth16.cpp:112,120)
 *   ug_field_op          F{r,q}_rawMMul / rawAdd / rawSub        (build/fr_raw_generic.cpp:11-39,107-148)
 *
 * Byte formats are the reference's: field elements are 32-byte little-endian; "Montgomery" means
 * value * 2^256 mod p (build/fr_raw_generic.cpp:192); G1 affine records are (x, y) Montgomery, 64 bytes;
 * G2 records are (x.a, x.b, y.a, y.b), 128 bytes; an all-zero record is the point at infinity; MSM
 * scalars and witness values are plain integers. Inputs may be unaligned (zkey sections start at
 * 4 mod 8); outputs are written with memcpy semantics.
 *
 * Environment variables read by libultragroth_hip.so (all optional; every setting of the first two groups gives correct
 * results):
 *
 *   deployment         ULTRAGROTH_DEVICE=n        device of a prover made by the reference's create calls (default 0)
 *                      ULTRAGROTH_DEVICES=a,b,..  one proof sharded over the listed devices behind the reference's API
 *                      ULTRAGROTH_TABLES=0|1|2    fixed-base window tables: never | created provers (default) | one-shot calls too
 *                      ULTRAGROTH_OVERLAP=0|1|2   H branch behind / beside (default since round 5) the witness products on one device
 *                                                 (0: one kernel on the chip at a time, for clean per-kernel times)
 *                      ULTRAGROTH_TABLES_BG=0     groth16_prover_create waits for its window tables (default: returns once the zkey is
 *                                                 resident; the tables are built in pieces between proofs, include/prover.h)
 *                      ULTRAGROTH_GRAPH=1         the device part of a created prover's proof recorded once per witness buffer and
 *                                                 replayed as a hipGraph (exact; measured a wash on this runtime, so off by default)
 *                      ULTRAGROTH_WITNESS_GATHER=0  ULTRAGROTH_DEVICES: a chain rank uploads the whole witness over its own PCIe link
 *                                                 (default: it collects the other ranks' slices from their HBM, peer copies)
 *                      ULTRAGROTH_SPARSE_B=0      keep B1 / B2 dense (default: when at most 3/4 of a circuit's B points are real, the prover
 *                                                 keeps only those, with a schedule of its own over their scalars: -15 % per proof at
 *                                                 2^24 with half of the B points at infinity, -26 % with three quarters)
 *                      ULTRAGROTH_TAILS=split     the G1 and G2 tails of the witness products side by side on two streams (exact; a wash)
 *                      ULTRAGROTH_FUSED=0         A, B1, C as separate base sets instead of one interleaved group
 *                      ULTRAGROTH_SHARD=PxB       many-device layout: P base-point ranges x B bucket classes
 *                      ULTRAGROTH_MAX_RANGE=n     scalars per schedule (tests: the piecewise path without a 2^27 circuit)
 *                      ULTRAGROTH_UPLOAD_THREADS=n, ULTRAGROTH_H_PRIORITY=h|n|l, ULTRAGROTH_TRACE=1 (phase times on stderr)
 *   tuning knobs       UG_MSM_C, UG_TABLE_C       window width of the classic / table schedules (default: cost model)
 *                      UG_SEG_LANES_LOG, UG_SEG_TAPER   segment length of the accumulation (default 2^20 lanes, tapered end)
 *                      UG_REDUCE_G1_LOG, UG_REDUCE_G2_LOG   lanes the bucket reduction keeps busy (default 17, 16)
 *                      UG_UPLOAD_PRIORITY=l|n|h   stream class of the witness staging lanes (default l)
 *                      UG_NTT_SHOUP=0             Montgomery-form twiddle products in the NTT passes (default: Shoup form, ff.hpp mul_shoup)
 *   test hooks         ULTRAGROTH_TEST_HOOKS=1 enables ug_test_set_blinding / ULTRAGROTH_TEST_BLINDING and ug_test_inject_fault;
 *                      without it those calls fail and the variables are ignored
 *   measurement        NOT in this library: UG_SORT=cub (library sort, links hipcub), UG_GROUP_FOLD_LOG (gathers folded into
 *                      cache: WRONG sums), UG_NTT_FUSE_STEPS (WRONG results), UG_SORT_IPT / _SPL / _DROP, UG_NTT_BATCH, UG_MATVEC_TILED exist only in the
 *                      -DUG_MEASURE build (`make -C ultragroth_amd/csrc MEASURE=1 measure` -> libultragroth_hip_measure.so,
 *                      loaded with ULTRAGROTH_LIB=<path> by the Python mirror); tests/test_abi.py checks that the product
 *                      library holds none of these names and no hipcub symbol.
 */
#ifndef ULTRAGROTH_HIP_H
#define ULTRAGROTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UG_OK 0
#define UG_ERROR 1

#define UG_FIELD_FR 0
#define UG_FIELD_FQ 1
#define UG_OP_MUL 0
#define UG_OP_ADD 1
#define UG_OP_SUB 2
#define UG_OP_SQR 3

typedef struct ug_ctx ug_ctx;            /* one device + one stream + reusable workspaces           */
typedef struct ug_bases ug_bases;        /* a G1 or G2 base-point array resident in HBM              */
typedef struct ug_dvec ug_dvec;          /* a device vector of 32-byte elements (witness, h, ...)    */
typedef struct ug_index ug_index;        /* a list of 32-bit indices resident in HBM                  */
typedef struct ug_schedule ug_schedule;  /* signed-digit bucket schedule of a range of a ug_dvec     */
typedef struct ug_hpoly ug_hpoly;        /* coefficient matrix + NTT tables of one circuit           */

/* message of the last failing call on this thread ("" if none) */
const char* ug_last_error(void);
/* number of visible HIP devices, or -1 when the HIP runtime cannot be initialised */
int ug_device_count(void);

int  ug_ctx_create(ug_ctx** ctx, int device);
/* The same with a stream priority class: +1 = the device's highest, 0 = normal, -1 = its lowest (a tuning knob: on MI355X /
 * ROCm 7.2 two compute streams of different classes were measured to share the chip exactly as two of the same class do). */
int  ug_ctx_create_priority(ug_ctx** ctx, int device, int priority_class);
void ug_ctx_destroy(ug_ctx* ctx);
/* block until all work queued on the context's stream has finished */
int  ug_ctx_sync(ug_ctx* ctx);

/* Upload n affine points (zkey record format) and convert them to the device form. The array stands for
 * the global point indices [global_first, global_first + n) of its zkey section, so that a rank of a
 * sharded prover can hold a slice. */
int  ug_bases_create_g1(ug_ctx* ctx, const void* host_points, uint64_t n, uint64_t global_first, ug_bases** out);
int  ug_bases_create_g2(ug_ctx* ctx, const void* host_points, uint64_t n, uint64_t global_first, ug_bases** out);
/* The same WITH the fixed-base window tables described below (table_c = their window width, 0 = none): the tables are
 * allocated with the set, the points are uploaded straight into table 0 and the table kernel is queued on the context
 * without a host wait, so the upload of the caller's next section overlaps it (zkey ingest). Work queued
 * on the context afterwards is ordered behind the build; ug_ctx_sync waits for it. */
int  ug_bases_create_tables_g1(ug_ctx* ctx, const void* host_points, uint64_t n, uint64_t global_first, int table_c, ug_bases** out);
int  ug_bases_create_tables_g2(ug_ctx* ctx, const void* host_points, uint64_t n, uint64_t global_first, int table_c, ug_bases** out);
/* Fixed-base window tables (no reference counterpart: the zkey's points never change between proofs, and 288 GB of
 * HBM holds them): extend the set by tables 2^(c j) * P_i, j = 1 .. ceil(255/c) - 1, c in [16, 24]. A schedule built
 * with ug_schedule_build_tables(.., c) then puts every window digit into ONE bucket set. Fails (nothing changed)
 * when device memory is short. ug_msm_table_window: the cost-model width for n scalars; ug_bases_tables_bytes: the
 * additional device memory the tables take. */
int      ug_msm_table_window(uint64_t n);
uint64_t ug_bases_tables_bytes(uint64_t n, int g2, int c);
int      ug_bases_precompute(ug_bases* b, int c);
/* give the tables' memory back (the set keeps its n points; schedules must then be built without tables) */
int      ug_bases_drop_tables(ug_bases* b);
int      ug_bases_table_window(const ug_bases* b);          /* width of the tables held, 0 = none */
/* DEFERRED TABLE BUILDS (cold start of a created prover): after ug_ctx_defer_tables(ctx, 1) the sets made by
 * ug_bases_create_tables_* / ug_bases_create_group_g1 on this context hold their points only and remember the width: nothing
 * is queued, not even the tables' memory is allocated. ug_bases_tables_step(set, max_points, &remaining) builds the tables of the next max_points points on the
 * context's stream and waits for them: a bounded piece of device time that the caller puts between proofs. Until
 * ug_bases_tables_ready(set) returns 1 the set is a plain set (schedules without tables read table 0 only; products over a table
 * schedule fail). Why pieces and not one kernel on a side stream: a table kernel's workgroups live for 17-26 ms, and whenever a
 * proof's kernel ends they take the free compute units -- measured, the first proof beside such a build took 2.0 s instead of 0.17. */
int      ug_ctx_defer_tables(ug_ctx* ctx, int on);
/* the room for a deferred set's tables: _alloc is nothing but the device allocation (no stream touched: it may run beside proofs; a
 * first allocation of 36 GiB was seen to take 1.1 s); _adopt, when nothing is queued on the set's context, moves the points into
 * table 0 of the new array, swaps it in and frees the old one. Then the pieces: */
int      ug_bases_tables_alloc(ug_bases* b, void** mem);
int      ug_bases_tables_adopt(ug_bases* b, void* mem);
int      ug_bases_tables_step(ug_bases* b, uint64_t max_points, uint64_t* remaining);
int      ug_bases_tables_ready(const ug_bases* b);
int      ug_ctx_mem_info(ug_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
void ug_bases_destroy(ug_bases* b);

/* device vectors of n 32-byte elements */
int  ug_dvec_create(ug_ctx* ctx, uint64_t n, ug_dvec** out);
int  ug_dvec_upload(ug_dvec* v, const void* host, uint64_t n);           /* host -> device, n <= capacity */
/* the same into elements [first, first + n); `via`: the context whose stream and staging buffers carry the copy (NULL =
 * the vector's own), so that two host threads can fill disjoint ranges of one vector at the same time */
int  ug_dvec_upload_range(ug_dvec* v, const void* host, uint64_t first, uint64_t n, ug_ctx* via);
/* the same as ug_dvec_upload for a vector NOTHING queued on the device refers to (the caller vouches for it): the copy
 * uses the staging streams only and does not wait for the context's stream -- the witness of the NEXT proof is staged this
 * way from a second host thread while the current proof runs (groth16_prover_prove from two threads on one prover) */
int  ug_dvec_upload_idle(ug_dvec* v, const void* host, uint64_t n);
int  ug_dvec_download(const ug_dvec* v, void* host, uint64_t first, uint64_t n);
/* out[i] = src[index[i]] for i < n (UltraGroth round / final witness gathers, src/ultra_groth.cpp:415-445) */
int  ug_dvec_gather(ug_dvec* out, const ug_dvec* src, const uint32_t* host_index, uint64_t n);
/* the same with the index list kept on the device (the two lists are part of the zkey: upload once at create) */
int  ug_index_create(ug_ctx* ctx, const uint32_t* host_index, uint64_t n, ug_index** out);
void ug_index_destroy(ug_index* index);
int  ug_dvec_gather_index(ug_dvec* out, const ug_dvec* src, const ug_index* index);
/* dst[index[i]] = values[i] for i < n; indices must be distinct (UltraGroth lookup signals written back into the
 * witness, src/ultra_groth.cpp:99-105) */
int  ug_dvec_scatter(ug_dvec* dst, const uint32_t* host_index, const void* host_values, uint64_t n);
/* UltraGroth lookup completion (the loop of compute_lookup, src/ultra_groth.cpp:99-105, with its push_vector :62-98
 * left implicit): for i < n in order  dst[w_idx[i]] = push[p_idx[i]],  where
 *   push = [ table[0] | table[1 + chunks[j]] for j < n_chunks | table[1], ..., table[2 L] ],
 * table = [challenge | inv2[0..L) | prod[0..L)] as plain 32-byte integers (L = lookup_size). Writes to one index keep
 * the last. All arrays are host arrays; indices out of range fail with nothing written. */
int  ug_dvec_apply_lookup(ug_dvec* dst, const uint32_t* w_idx, const uint32_t* p_idx, uint64_t n,
                          const uint32_t* chunks, uint64_t n_chunks, const void* table, uint64_t lookup_size);
/* The lookup table of compute_lookup (src/ultra_groth.cpp:71-80) for challenge `rand_plain` (32-byte plain integer):
 * table_out = [rand | inv2[0..L) | prod[0..L)] with inv2[i] = 1 / (i + rand) (0 when i + rand = 0) and
 * prod[i] = frequencies[i] * inv2[i], plain integers, (1 + 2 L) * 32 bytes of host memory -- the `table` argument of
 * ug_dvec_apply_lookup. Row index and frequency follow the reference's (int, Element) overloads (>= 2^31: value - 2^32). */
int  ug_fr_lookup_table(ug_ctx* ctx, const void* rand_plain, const uint32_t* frequencies, uint64_t lookup_size, void* table_out);
/* non-owning view of n 32-byte elements already in device memory (e.g. a torch / RCCL buffer) */
int  ug_dvec_wrap(ug_ctx* ctx, void* device_ptr, uint64_t n, ug_dvec** out);
uint64_t ug_dvec_size(const ug_dvec* v);
/* raw device address of a vector's first element (for the phase calls of include/prover.h that take device pointers) */
void* ug_dvec_device_ptr(const ug_dvec* v);
/* dst[dst_first .. + count) = src[src_first .. + count) between vectors of ANY two devices of the node (blocking) */
int  ug_dvec_copy(ug_dvec* dst, uint64_t dst_first, const ug_dvec* src, uint64_t src_first, uint64_t count);
/* the same with the copy queued on `via`'s stream (a context of dst's device; NULL = dst's own) */
int  ug_dvec_copy_via(ug_dvec* dst, uint64_t dst_first, const ug_dvec* src, uint64_t src_first, uint64_t count, ug_ctx* via);
void ug_dvec_destroy(ug_dvec* v);

/* Decompose scalars [first, first + count) of `scalars` (plain 32-byte integers) into signed window
 * digits grouped by bucket. One schedule serves every base set multiplied by the same scalars. */
int  ug_schedule_create(ug_ctx* ctx, ug_schedule** out);
/* release the device buffers of a schedule / of a context's MSM workspaces (they are re-allocated by the next build /
 * product): what a resident multi-circuit prover trims first when HBM is short */
int  ug_schedule_trim(ug_schedule* s);
int  ug_ctx_trim(ug_ctx* ctx);
int  ug_schedule_build(ug_schedule* s, const ug_dvec* scalars, uint64_t first, uint64_t count);
/* the same for base sets that hold window tables of width c (ug_bases_precompute); count <= 2^27 */
int  ug_schedule_build_tables(ug_schedule* s, const ug_dvec* scalars, uint64_t first, uint64_t count, int c);
/* BUCKET CLASSES (a many-device prover that shards the witness products by bucket instead of by base point; DESIGN.md section 7,
 * no counterpart in the reference): every later build of the schedule keeps only the (scalar, window) digits whose bucket
 * b = |digit| - 1 has b mod 2^q_log in [first_residue, first_residue + residues) -- except the lowest `specials` (<= 64) bucket ids
 * of every window, the digits small witness values pile up in: those are kept iff their scalar's index (in the scalar vector) lies in
 * [special_first, special_first + special_count). Schedules built over the SAME scalars and base sets whose residue ranges tile
 * [0, 2^q_log) and whose special ranges tile the scalars give products that ADD UP to the product of the plain schedule: each
 * holds its share of the entries at the plain schedule's window width and reduces only its share of the buckets. q_log = 0
 * switches the classes off. Needs a window of at least q_log + 4 bits (ug_schedule_build* fail otherwise). */
int  ug_schedule_set_classes(ug_schedule* s, int q_log, uint32_t first_residue, uint32_t residues, uint32_t specials,
                             uint64_t special_first, uint64_t special_count);
void ug_schedule_destroy(ug_schedule* s);

/* out = sum over the schedule's scalars s_i (global index i) of s_i * P_{i - index_shift}; points whose
 * global index falls outside the uploaded slice are skipped. out is an affine record in the reference
 * format (64 bytes for G1, 128 for G2; all zero = infinity). */
int  ug_msm_g1(ug_ctx* ctx, const ug_bases* bases, const ug_schedule* s, int64_t index_shift, void* out_affine);
int  ug_msm_g2(ug_ctx* ctx, const ug_bases* bases, const ug_schedule* s, int64_t index_shift, void* out_affine);
/* count <= 8 products over ONE schedule (G1 and G2 sets mixed freely; index_shifts may be NULL), queued back to back with
 * a single host synchronisation: outs[k] receives what ug_msm_g1 / ug_msm_g2 would write for bases[k]. */
int  ug_msm_batch(ug_ctx* ctx, int count, const ug_bases* const* bases, const ug_schedule* s, const int64_t* index_shifts,
                  void* const* outs_affine);
/* The same in two halves, so that a whole proof needs ONE host wait: ug_msm_batch_enqueue queues the products (up to 8
 * per context may be pending; `outs` must stay valid) and returns at once; ug_ctx_collect waits for everything queued on
 * the context, then finishes the queued results on the host (Horner, affine conversion) into their `outs`. Schedules
 * (ug_schedule_build*) and ug_hpoly_run queue their work without a host wait as well. ug_ctx_wait orders two contexts on
 * the device: what is queued on `waiter` afterwards starts when what was queued on `signal` before has finished. */
int  ug_msm_batch_enqueue(ug_ctx* ctx, int count, const ug_bases* const* bases, const ug_schedule* schedule,
                          const int64_t* index_shifts, void* const* outs);
/* A GROUP of 2 or 3 G1 base sets that are always multiplied by the same scalars -- A, B1 and C of a Groth16 proof
 * (src/groth16.cpp:55,58,64 all read the witness; C with its index shift), A and B1 of an UltraGroth one -- kept as ONE array
 * of K-point records, so that the accumulation kernel reads the entry list once and gathers the K points of an entry from
 * adjacent memory. Member m brings n[m] points (zkey format), the first of which belongs to the scalar with global index
 * first[m]; the group covers scalars [group_first, group_first + slots), a slot a member has no point for is infinity.
 * table_c != 0: with fixed-base window tables of that width (as ug_bases_create_tables_g1). ug_msm_group_enqueue queues the K
 * products over a schedule (results in outs[m], 64 bytes each, after ug_ctx_collect); the schedule's range must lie in the
 * group's. ug_bases_destroy / _precompute / _drop_tables / _table_window accept a group. */
int  ug_bases_create_group_g1(ug_ctx* ctx, int members, const void* const* host_points, const uint64_t* n, const uint64_t* first,
                              uint64_t group_first, uint64_t slots, int table_c, ug_bases** out);
int  ug_bases_members(const ug_bases* bases);
/* 1 when every one of the n records of `record_bytes` bytes at host_points is the point at infinity (all zero): such a set's products
 * are the point at infinity, no kernel runs for them (ug_bases_create_* mark the set themselves), and a prover keeps it out of a
 * base group. Stops at the first other record, i.e. at once for any real section. */
int  ug_points_all_infinity(const void* host_points, uint64_t n, uint64_t record_bytes);
int  ug_msm_group_enqueue(ug_ctx* ctx, const ug_bases* group, const ug_schedule* schedule, void* const* outs);
/* The witness products of a proof in one call -- the K products of a base group and ONE G2 set over the same schedule (S1-S4,
 * src/groth16.cpp:55-64): both accumulations back to back, then the G1 and the G2 tails (cut buckets, bucket reduction, tree sums,
 * result copies: chains of dependent EC additions) side by side on two streams. Results as after ug_msm_group_enqueue +
 * ug_msm_batch_enqueue: outs_group[m] (64 bytes each) and out_g2 (128 bytes) after ug_ctx_collect. */
int  ug_msm_witness_enqueue(ug_ctx* ctx, const ug_bases* group, const ug_bases* g2set, const ug_schedule* schedule,
                            void* const* outs_group, void* out_g2);
int  ug_ctx_collect(ug_ctx* ctx);
int  ug_ctx_wait(ug_ctx* waiter, ug_ctx* signal);
/* For a caller that queued work and then failed before ug_ctx_collect: waits for what is still running on the context (the
 * kernels read the caller's buffers) and forgets the queued products, whose `outs` may no longer exist. Never fails. */
void ug_ctx_abandon(ug_ctx* ctx);
/* Test hook, honoured only in processes started with ULTRAGROTH_TEST_HOOKS=1 (UG_ERROR otherwise): the `after`-th next call
 * that passes fault point `site` fails with "injected fault". Lets the tests take the error paths of a running proof. */
/* diagnostic: the passes (return value, <= 4; -1 on bad arguments) the schedule sort makes for keys below 2^bits, with the
 * bit offset and width of each pass's digit (csrc/sort.hip). Host arithmetic only. */
int  ug_sort_plan(int bits, int shift[4], int bins_log[4]);
#define UG_FAULT_HPOLY_RUN 1
#define UG_FAULT_SCHEDULE_BUILD 2
int  ug_test_inject_fault(int site, int after);

/* coefs: n_coefs packed 44-byte records {u32 m, u32 c, u32 s, Fr coef} (zkey section 4 past its 4-byte
 * count, src/groth16.cpp:38). Builds the row-sorted matrix and the NTT tables for domain_size. */
int  ug_hpoly_create(ug_ctx* ctx, const void* host_coefs, uint64_t n_coefs, uint32_t domain_size,
                     uint32_t n_vars, ug_hpoly** out);
/* h (domain_size plain integers) from the witness (n_vars plain integers), all on the device */
int  ug_hpoly_run(ug_hpoly* hp, const ug_dvec* witness, ug_dvec* h_out);
/* The same block in two steps, for a prover sharded over GPUs: ug_hpoly_chain computes the coset evaluations of
 * ONE of the three polynomials (which = 0: A.w, 1: B.w, 2: (A.w) o (B.w)) -- ranks take different polynomials and
 * exchange slices -- and ug_hpoly_combine turns matching slices of the three vectors into h[first, first+count).
 * The evaluation vectors are in the library's device form and only meaningful to this library. ug_hpoly_combine needs
 * no coefficient matrix: hp may be NULL (then the work runs on h_out's context). */
int  ug_hpoly_chain(ug_hpoly* hp, const ug_dvec* witness, int which, ug_dvec* out_evals);
int  ug_hpoly_combine(ug_hpoly* hp, const ug_dvec* a, const ug_dvec* b, const ug_dvec* c, uint64_t first, uint64_t count,
                      ug_dvec* h_out);
/* optional: also return the three coset evaluation vectors (reference Montgomery form), for tests */
int  ug_hpoly_debug_abc(ug_hpoly* hp, void* host_a, void* host_b, void* host_c);
void ug_hpoly_destroy(ug_hpoly* hp);

/* In-place size-2^logn transform of a host buffer of Montgomery Fr elements, natural order in and out.
 * inverse != 0 also scales by 1/n. */
int  ug_fr_ntt(ug_ctx* ctx, void* host_data, int logn, int inverse);
/* out[i] = a[i] (op) b[i] on n Montgomery elements of the chosen field (host buffers) */
int  ug_field_op(ug_ctx* ctx, int field, int op, void* out, const void* a, const void* b, uint64_t n);

/* Tooling for synthetic circuits (bench.py, tests): host_out[i] = (seed + i) * G as zkey-format affine
 * records, G given as one such record (64 bytes for G1, 128 for G2). Not part of the reference interface. */
int  ug_synth_points(ug_ctx* ctx, int g2, const void* generator_record, uint64_t seed, uint64_t n, void* host_out);

/* CAPTURED LAUNCH SEQUENCES (one hipGraph per created prover and witness buffer; no reference counterpart -- the reference's
 * prove, src/groth16.cpp:48-203, is one fixed sequence of calls, and so is this library's: no launch depends on a host read-back).
 * ug_graph_begin puts ctx's stream into capture and forks ctx2's stream (may be NULL) off it; whatever the library queues on the
 * two contexts until ug_graph_end -- schedules, products, ug_hpoly_run, ug_ctx_wait edges -- is recorded instead of run, the timing
 * events included (as event-record nodes: ug_ctx_timings / ug_ctx_kernel_stats keep working under replay). Calls that wait
 * for the device (ug_ctx_sync, _collect, blocking products, uploads) must not be made on a capturing context. ug_graph_end
 * instantiates the graph and takes the queued products out of the contexts; ug_graph_launch puts them back and launches on ctx's
 * stream: collect as after the eager calls, ctx FIRST. ug_graph_valid: 0 once any per-proof device buffer of the process has
 * been re-allocated since the capture (schedules, workspaces): the caller then drops the graph and captures again.
 * ug_graph_abort ends a capture that failed half way (nothing was queued). */
typedef struct ug_graph ug_graph;
int      ug_graph_begin(ug_ctx* ctx, ug_ctx* ctx2);
int      ug_graph_end(ug_ctx* ctx, ug_graph** out);
void     ug_graph_abort(ug_ctx* ctx);
int      ug_graph_valid(const ug_graph* g);
uint64_t ug_graph_nodes(const ug_graph* g);
int      ug_graph_launch(ug_graph* g);
void     ug_graph_destroy(ug_graph* g);

/* milliseconds of device time spent in the MSM and H-polynomial parts since the last reset
 * (the MSM | FFT split the reference prints in src/ultra_groth.cpp:199-335) */
int  ug_ctx_timings(ug_ctx* ctx, double* msm_ms, double* fft_ms, int reset);

/* HIP events (recorded on the launch stream) around the launches of the kernels the roofline is reported for:
 * which = 0 the G1 bucket accumulation, 1 the G2 bucket accumulation, 2 the NTT pass kernel. Average launch duration in
 * ms since the last reset, the launch count, and the units processed ((point, window) entries; NTT points per pass).
 * The statistics of a context are collected from its first call with reset != 0 on (an event pair around a launch costs
 * 10-25 us of idle device: a caller that never asks never pays). A captured launch sequence (ug_graph_*) holds the event pairs
 * that were on when it was captured. */
int  ug_ctx_kernel_stats(ug_ctx* ctx, int which, double* launch_ms_avg, uint64_t* launches, uint64_t* units, int reset);

#ifdef __cplusplus
}
#endif
#endif
