/*
 * prover.h -- outer C-ABI of libultragroth_hip.so.
 *
 * Drop-in for the extern "C" prover API of rarimo/ultragroth (src/prover.h): same symbol names, argument
 * order, types, return codes and error-message strings, so that every existing binding of the reference
 * (its own CLIs src/main_prover.cpp:39-65 and src/main_prover_ultra_groth.cpp:39-65, and the Go / iOS /
 * Android / React-Native / Flutter wrappers listed in README.md:158-168) links against this library
 * unchanged. The MSMs and the H-polynomial block run on one MI355X; everything else is host code.
 *
 * Reference declarations mirrored here (src/prover.h):
 *   groth16_public_size_for_zkey_buf :18-25      ultra_groth_public_size_for_zkey_buf :26-33
 *   groth16_public_size_for_zkey_file :44-50     ultra_groth_public_size_for_zkey_file :51-57
 *   groth16_proof_size :62-65                    ultra_groth_proof_size :69-72
 *   groth16_prover_create :80-87                 ultra_groth_prover_create :89-96
 *   groth16_prover_create_zkey_file :104-110     ultra_groth_prover_create_zkey_file :112-118
 *   groth16_prover_prove :127-138                ultra_groth_prover_prove :140-151
 *   groth16_prover_destroy :156-159              ultra_groth_prover_destroy :161-164
 *   groth16_prover :173-185                      ultra_groth_prover :187-199
 *   groth16_prover_zkey_file :208-219            ultra_groth_prover_zkey_file :221-232
 *
 * Behavioural notes (all as in src/prover.cpp unless stated):
 *   - error_msg receives at most error_msg_maxsize bytes via strncpy (may be unterminated when full);
 *   - proof_size / public_size are inputs only: buffers shorter than the minimum (810 / 1400 bytes for
 *     the proof, 82 * nPublic + 4 for the public signals) give PROVER_ERROR_SHORT_BUFFER;
 *   - *_create uploads and converts the zkey's base points and coefficients to HBM, so -- unlike the
 *     reference, which keeps raw pointers into the caller's zkey buffer -- the buffer may be released as
 *     soon as *_create returns (this also removes the reference's dangling-mapping defect in
 *     *_create_zkey_file, src/prover.cpp:449-473);
 *   - a prover object may be used from several host threads at once, as the reference's (whose prove
 *     allocates everything per call, src/prover.cpp:341-391): the calls take turns on the device. A
 *     prover (Groth16 or UltraGroth) copies the witness of a waiting call into its second witness buffer meanwhile, so
 *     two threads proving on one object hide the host-to-device copy of one proof behind the kernels of
 *     the other (2^24: 10 ms of a 150 ms proof). The ug_ phase calls of a sharded proof below are driven
 *     by one thread per object and must not be mixed with concurrent *_prove calls on the same object;
 *   - a prover object is bound to one HIP device (environment variable ULTRAGROTH_DEVICE, default 0).
 *
 * Additions (not in the reference), all prefixed ug_: deterministic blinding for tests, per-phase
 * timings, and the sharded entry points a multi-GPU launcher uses (one process per GPU).
 */
#ifndef ULTRAGROTH_PROVER_H
#define ULTRAGROTH_PROVER_H

#ifdef __cplusplus
extern "C" {
#endif

#define PROVER_OK                     0x0
#define PROVER_ERROR                  0x1
#define PROVER_ERROR_SHORT_BUFFER     0x2
#define PROVER_INVALID_WITNESS_LENGTH 0x3

int groth16_public_size_for_zkey_buf(const void *zkey_buffer, unsigned long long zkey_size,
                                     unsigned long long *public_size,
                                     char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_public_size_for_zkey_buf(const void *zkey_buffer, unsigned long long zkey_size,
                                         unsigned long long *public_size,
                                         char *error_msg, unsigned long long error_msg_maxsize);

int groth16_public_size_for_zkey_file(const char *zkey_fname, unsigned long long *public_size,
                                      char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_public_size_for_zkey_file(const char *zkey_fname, unsigned long long *public_size,
                                          char *error_msg, unsigned long long error_msg_maxsize);

void groth16_proof_size(unsigned long long *proof_size);
void ultra_groth_proof_size(unsigned long long *proof_size);

int groth16_prover_create(void **prover_object, const void *zkey_buffer, unsigned long long zkey_size,
                          char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_prover_create(void **prover_object, const void *zkey_buffer, unsigned long long zkey_size,
                              char *error_msg, unsigned long long error_msg_maxsize);

int groth16_prover_create_zkey_file(void **prover_object, const char *zkey_file_path,
                                    char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_prover_create_zkey_file(void **prover_object, const char *zkey_file_path,
                                        char *error_msg, unsigned long long error_msg_maxsize);

int groth16_prover_prove(void *prover_object, const void *wtns_buffer, unsigned long long wtns_size,
                         char *proof_buffer, unsigned long long *proof_size,
                         char *public_buffer, unsigned long long *public_size,
                         char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_prover_prove(void *prover_object, const void *wtns_buffer, unsigned long long wtns_size,
                             char *proof_buffer, unsigned long long *proof_size,
                             char *public_buffer, unsigned long long *public_size,
                             char *error_msg, unsigned long long error_msg_maxsize);

void groth16_prover_destroy(void *prover_object);
void ultra_groth_prover_destroy(void *prover_object);

int groth16_prover(const void *zkey_buffer, unsigned long long zkey_size,
                   const void *wtns_buffer, unsigned long long wtns_size,
                   char *proof_buffer, unsigned long long *proof_size,
                   char *public_buffer, unsigned long long *public_size,
                   char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_prover(const void *zkey_buffer, unsigned long long zkey_size,
                       const void *wtns_buffer, unsigned long long wtns_size,
                       char *proof_buffer, unsigned long long *proof_size,
                       char *public_buffer, unsigned long long *public_size,
                       char *error_msg, unsigned long long error_msg_maxsize);

int groth16_prover_zkey_file(const char *zkey_file_path,
                             const void *wtns_buffer, unsigned long long wtns_size,
                             char *proof_buffer, unsigned long long *proof_size,
                             char *public_buffer, unsigned long long *public_size,
                             char *error_msg, unsigned long long error_msg_maxsize);
int ultra_groth_prover_zkey_file(const char *zkey_file_path,
                                 const void *wtns_buffer, unsigned long long wtns_size,
                                 char *proof_buffer, unsigned long long *proof_size,
                                 char *public_buffer, unsigned long long *public_size,
                                 char *error_msg, unsigned long long error_msg_maxsize);

/* ---- additions ------------------------------------------------------------------------------------ */

/* Test hook: the next draws of blinding randomness (31 bytes each: r then s for Groth16;
 * r_k, r, s for UltraGroth -- src/groth16.cpp:165-166, src/ultra_groth.cpp:173,345-346) are taken from
 * `bytes` (cyclically) instead of the operating system. n = 0 restores OS entropy. Process-wide, and it removes
 * zero knowledge, so it is honoured only in a process whose environment holds ULTRAGROTH_TEST_HOOKS=1 when the
 * library first looks (tests, bench.py --check, smoke()); otherwise it changes nothing and returns PROVER_ERROR. */
int ug_test_set_blinding(const void *bytes, unsigned long long n);

/* Device milliseconds of the last prove on this prover object: MSM part, H-polynomial ("FFT") part, and
 * host wall-clock milliseconds of the whole prove call. */
int ug_prover_last_timings(void *prover_object, double *msm_ms, double *fft_ms, double *total_ms);
/* average launch duration (ms), launch count and units processed of the kernels the roofline is reported for, since the
 * prover was created or the counters were last reset: which = 0 G1 bucket accumulation, 1 G2 bucket accumulation
 * (units: (point, window) entries), 2 NTT pass kernel (units: points per pass) */
int ug_prover_kernel_stats(void *prover_object, int which, double *launch_ms_avg, unsigned long long *launches,
                           unsigned long long *units, int reset);
/* host wall-clock milliseconds the last prove spent bringing the witness into HBM (parse + host-to-device copy) */
int ug_prover_last_upload_ms(void *prover_object, double *upload_ms);
/* COLD START: groth16_prover_create (src/prover.h:127-138 of the reference; an unsharded prover here) returns once the zkey is
 * resident -- 0.25 s at 2^24 -- and does not wait for its fixed-base window tables (1.9 s of kernels): a thread of the prover
 * builds them in pieces of ~10 ms of device time between the proofs, which meanwhile run on the classic windows (one piece per
 * turn while the prover is idle, four after a caller's proof under load), and the first proof after the last piece uses them.
 * 1 = the tables are in use (or the prover has none to wait for), 0 = still being built; wait != 0 blocks until they are;
 * -1 = error. ULTRAGROTH_TABLES_BG=0 makes create wait as it did before. */
int ug_prover_tables_ready(void *prover_object, int wait);

/* ---- resident multi-circuit prover ---------------------------------------------------------------------
 * The GPU form of FullProver's map<circuit, Prover> (src/fullprover.cpp:21-63; its HTTP shell and witness calculator
 * are out of scope): several created provers -- Groth16 and UltraGroth, told apart by the zkey's protocol field --
 * share one device under an HBM budget (0 = all the memory free at creation). When the budget is short the registry
 * gives back, least recently used circuit first: fixed-base window tables (a circuit without them proves with the
 * classic windows, only slower), then per-proof workspaces, then whole circuits; a circuit that was loaded from a file
 * and evicted comes back by itself on its next ug_registry_prove. Proofs of different circuits take turns on the
 * device (the reference answers `busy`, src/fullprover.cpp:82-101). Return codes and messages as the prover calls. */
int ug_registry_create(void **registry, int device, unsigned long long hbm_budget_bytes, char *error_msg,
                       unsigned long long error_msg_maxsize);
int ug_registry_load(void *registry, const char *circuit, const void *zkey_buffer, unsigned long long zkey_size,
                     char *error_msg, unsigned long long error_msg_maxsize);
/* circuit name = file name without directory and extension (getfilename(), src/fullprover.cpp:14-19) */
int ug_registry_load_file(void *registry, const char *zkey_file_path, char *error_msg, unsigned long long error_msg_maxsize);
int ug_registry_prove(void *registry, const char *circuit, const void *wtns_buffer, unsigned long long wtns_size,
                      char *proof_buffer, unsigned long long *proof_size, char *public_buffer,
                      unsigned long long *public_size, char *error_msg, unsigned long long error_msg_maxsize);
int ug_registry_evict(void *registry, const char *circuit, char *error_msg, unsigned long long error_msg_maxsize);
/* circuit != NULL: its resident bytes, state (0 not loaded, 1 resident without tables, 2 resident with tables,
 * 3 evicted but reloadable from its file) and proof count. circuit == NULL: device bytes in use by the registry,
 * number of resident circuits (in *state), proofs made. */
int ug_registry_info(void *registry, const char *circuit, unsigned long long *resident_bytes, int *state,
                     unsigned long long *proofs);
void ug_registry_destroy(void *registry);

/* Sharded Groth16 proving, one process per GPU. Rank `shard_rank` of `shard_count` holds the base points
 * [rank * n / count, (rank + 1) * n / count) of every section and produces partial sums; the partial
 * sums of all ranks are added (ug_groth16_partials_add) and finished on one rank.
 * partials layout: MSM_A (64) | MSM_B1 (64) | MSM_B2 (128) | MSM_C (64) | MSM_H (64) affine records. */
#define UG_GROTH16_PARTIALS_SIZE 384
int ug_groth16_prover_create_sharded(void **prover_object, const void *zkey_buffer, unsigned long long zkey_size,
                                     int device, int shard_rank, int shard_count,
                                     char *error_msg, unsigned long long error_msg_maxsize);
/* The same with the rank's slice [witness_first, witness_end) of the witness-indexed sections (A, B1, B2 and, shifted
 * by nPublic + 1, C) chosen by the caller, whose ranks must tile [0, nVars): ranks that also run an H-polynomial chain
 * (ug_groth16_prover_hpoly_chain) can be given fewer points. The H section stays split evenly. */
int ug_groth16_prover_create_sharded_range(void **prover_object, const void *zkey_buffer, unsigned long long zkey_size,
                                           int device, int shard_rank, int shard_count,
                                           unsigned long long witness_first, unsigned long long witness_end,
                                           char *error_msg, unsigned long long error_msg_maxsize);
/* The same from this rank's SLICES only, so that no rank ever holds the whole zkey in host memory (37.6 GB at 2^26):
 * zkey_header = the bytes of zkey section 2; coefs = section 4 without its 4-byte count (NULL: this rank will run no
 * H-polynomial chain and keeps no coefficient matrix); points_* = this rank's slice of sections 5..9, starting at the
 * first point of the ranges ug_groth16_shard_ranges reports: out[6] = {witness first, end, C first, end, H first, end}
 * (witness_range: two values as in _create_sharded_range, or NULL for the even split). slice_bytes = the byte counts of the
 * caller's points_a .. points_h buffers: a slice shorter than the rank's range is refused instead of read past its end. */
/* The witness range [out[0], out[1]) a launcher should give rank `shard_rank` of `shard_count` so that the ranks finish
 * together: ranks 0..2 also run an iFFT/twist/FFT chain each (ug_groth16_prover_hpoly_chain) and get fewer points. The ranges of
 * all ranks tile [0, n_vars). (What ULTRAGROTH_DEVICES uses inside the library and bench.py between its processes.) */
int ug_groth16_balanced_witness_range(unsigned long long n_vars, int shard_rank, int shard_count, unsigned long long out[2]);
int ug_groth16_shard_ranges(unsigned long long n_vars, unsigned long long n_public, unsigned long long domain_size,
                            int shard_rank, int shard_count, const unsigned long long *witness_range,
                            unsigned long long out[6]);
int ug_groth16_prover_create_sharded_slices(void **prover_object, const void *zkey_header, unsigned long long zkey_header_size,
                                            const void *coefs, unsigned long long n_coefs,
                                            const void *points_a, const void *points_b1, const void *points_b2,
                                            const void *points_c, const void *points_h,
                                            const unsigned long long slice_bytes[5],
                                            int device, int shard_rank, int shard_count,
                                            const unsigned long long *witness_range,
                                            char *error_msg, unsigned long long error_msg_maxsize);
/* THE LAYOUT OF A MANY-DEVICE PROVER. The witness products can be cut over the ranks in two ways, and in
 * any product of the two: by BASE-POINT RANGE (a rank holds a contiguous range of the points and scalars: the functions above)
 * and by BUCKET CLASS (the ranks of a group hold the SAME range with its whole window tables, and each takes the bucket ids of
 * its residues mod 2^q_log: it sorts, accumulates and reduces a share of the entries at the full range's window width).
 * ug_groth16_shard_layout gives rank `shard_rank` of `shard_count` its part of the layout the library would use itself
 * (ULTRAGROTH_DEVICES): point_ranges = the number of base-point ranges P (shard_count / P ranks share each; P = shard_count is
 * the base-point form), 0 = chosen here -- ULTRAGROTH_SHARD=PxB, else bucket classes from four ranks on in as few ranges as let a
 * range's window tables fit a device of hbm_bytes (0 = 256 GiB). The shares are balanced for the ranks that also run an
 * H-polynomial chain (bit k of out[11]: chain k); the H product is cut by range, and with five ranks or more the chain ranks
 * take none of it (their H range is empty).
 *   out[12] = { witness first, end | C first, end | H first, end | q_log (0 = no classes), first residue, residues |
 *               first, end of the scalars whose special (lowest) buckets the rank owns | chains bitmask }
 * A pure function of its arguments and ULTRAGROTH_SHARD: every rank of a launcher computes the same layout.
 * ug_groth16_prover_create_sharded_layout: as _create_sharded_slices (the slices are those of out[0..5]), with the layout. */
int ug_groth16_shard_layout(unsigned long long n_vars, unsigned long long n_public, unsigned long long domain_size,
                            int shard_rank, int shard_count, int point_ranges, unsigned long long hbm_bytes,
                            unsigned long long out[12]);
int ug_groth16_prover_create_sharded_layout(void **prover_object, const void *zkey_header, unsigned long long zkey_header_size,
                                            const void *coefs, unsigned long long n_coefs,
                                            const void *points_a, const void *points_b1, const void *points_b2,
                                            const void *points_c, const void *points_h,
                                            const unsigned long long slice_bytes[5],
                                            int device, int shard_rank, int shard_count,
                                            const unsigned long long layout[12],
                                            char *error_msg, unsigned long long error_msg_maxsize);
/* upload the witness (wtns file buffer) to the device; returns PROVER_INVALID_WITNESS_LENGTH etc. */
int ug_groth16_prover_load_witness(void *prover_object, const void *wtns_buffer, unsigned long long wtns_size,
                                   char *error_msg, unsigned long long error_msg_maxsize);
/* The witness in two parts: part 0 = the scalars of this rank's slice (all that run_witness_msm reads), part 1 = the rest
 * (read only by hpoly_chain / run: a rank without a chain never uploads it). Part 1 travels on the H branch's stream and
 * may be called from a second host thread while the MSMs over part 0 run. */
int ug_groth16_prover_load_witness_part(void *prover_object, const void *wtns_buffer, unsigned long long wtns_size, int part,
                                        char *error_msg, unsigned long long error_msg_maxsize);
/* One whole proof (S1-S13, src/groth16.cpp:48-203) of the witness ug_groth16_prover_load_witness left resident in HBM: what
 * groth16_prover_prove does after it has parsed and copied the witness (r and s drawn first, the blinding multiples that need
 * only them formed on host threads beside the device work). For callers that prove one witness several times with fresh
 * blinding, and for bench.py, whose timed region starts with the inputs resident. Groth16 provers on one device. */
int ug_groth16_prover_prove_resident(void *prover_object, char *proof_buffer, unsigned long long *proof_size,
                                     char *public_buffer, unsigned long long *public_size,
                                     char *error_msg, unsigned long long error_msg_maxsize);
/* device part of the prove on the resident witness: the five MSMs over this rank's slice + H polynomial */
int ug_groth16_prover_run(void *prover_object, void *partials_out,
                          char *error_msg, unsigned long long error_msg_maxsize);
/* The same device part in phases, so that ranks can split the H-polynomial block (three independent
 * iFFT/twist/FFT chains) instead of replicating it:
 *   run_witness_msm   A, B1, B2, C partial sums over this rank's slice (H record left at infinity)
 *   hpoly_chain       coset evaluations of polynomial `which` (0,1,2) into a device buffer of domainSize * 32 bytes
 *   h_range           [first, first + count): the slice of h this rank multiplies
 *   hpoly_combine     device buffers holding this rank's slice of the three evaluation vectors -> h slice
 *   run_h_msm         H partial sum over this rank's slice (other records at infinity)
 * Device pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()) on the prover's device. */
int ug_groth16_prover_run_witness_msm(void *prover_object, void *partials_out, char *error_msg, unsigned long long error_msg_maxsize);
/* run_witness_msm in two calls, so that ONE host thread keeps both of a rank's streams busy: _begin queues the four products on
 * the witness stream and returns without waiting for the device; the caller then drives the H branch (hpoly_chain, the exchange
 * of the evaluation slices with the other ranks, hpoly_combine, run_h_msm -- all on the rank's second stream) while the
 * products run, and _end waits for them and returns their partial sums. Between the two calls no witness part 0 may be loaded
 * and no other witness product started. On rank 0, _begin also draws the blinding scalars r and s (src/groth16.cpp:158-166,
 * before the device work) and forms the multiples that need only them on host threads; ug_groth16_prover_finish then uses
 * those. On a node where most ranks wait for the chains (five ranks or more), a chain rank calls hpoly_chain BEFORE _begin:
 * stream priorities have no effect on this stack, so a chain queued beside the products would take 2.4 times as long. */
int ug_groth16_prover_witness_msm_begin(void *prover_object, char *error_msg, unsigned long long error_msg_maxsize);
int ug_groth16_prover_witness_msm_end(void *prover_object, void *partials_out, char *error_msg, unsigned long long error_msg_maxsize);
int ug_groth16_prover_hpoly_chain(void *prover_object, int which, void *device_out, char *error_msg, unsigned long long error_msg_maxsize);
int ug_groth16_prover_h_range(void *prover_object, unsigned long long *first, unsigned long long *count, unsigned long long *domain_size);
int ug_groth16_prover_hpoly_combine(void *prover_object, void *device_a, void *device_b, void *device_c,
                                    char *error_msg, unsigned long long error_msg_maxsize);
int ug_groth16_prover_run_h_msm(void *prover_object, void *partials_out, char *error_msg, unsigned long long error_msg_maxsize);
int ug_groth16_partials_add(void *partials_acc, const void *partials_other);
/* blinding + JSON from summed partials (host only) */
int ug_groth16_prover_finish(void *prover_object, const void *partials_sum,
                             char *proof_buffer, unsigned long long *proof_size,
                             char *public_buffer, unsigned long long *public_size,
                             char *error_msg, unsigned long long error_msg_maxsize);

/* Sharded UltraGroth proving (BASELINE configs[4]), one process per GPU. Rank `shard_rank` of `shard_count` holds
 * contiguous slices of the witness-indexed sections (A, B1, B2), of the round set with its index list, of the final set
 * with its index list, and of H; every rank keeps the whole witness. One proof, in this order:
 *   ug_groth16_prover_load_witness           every rank: the .uwtns buffer (witness + lookup sections)
 *   ug_ultra_groth_prover_round_commit       every rank: its part of the round commitment (64-byte affine record)
 *     -- exchange: all ranks' parts are added (ug_g1_record_add) --
 *   ug_ultra_groth_prover_round_finish       ONE rank: draws the round randomness, blinds the sum -> the commitment pi_r
 *     -- exchange: the commitment goes to every rank --
 *   ug_ultra_groth_prover_apply_commitment   every rank: Fiat-Shamir challenge, lookup signals written into its witness
 *   ug_groth16_prover_run_witness_msm        every rank: A | B1 | B2 | final-set partial sums (or _witness_msm_begin ... _end around
 *                                            the H-branch calls below, as for Groth16: an UltraGroth rank has two streams too)
 *   ug_groth16_prover_hpoly_chain / _h_range / _hpoly_combine / _run_h_msm    as for Groth16 (or replicate the block)
 *     -- exchange: partial blocks added (ug_groth16_partials_add) --
 *   ug_groth16_prover_finish                 the rank that closed the round: r, s, blinding, proof.json / public.json
 * The ug_groth16_prover_* phase calls accept both kinds of prover object. */
int ug_ultra_groth_prover_create_sharded(void **prover_object, const void *zkey_buffer, unsigned long long zkey_size,
                                         int device, int shard_rank, int shard_count,
                                         char *error_msg, unsigned long long error_msg_maxsize);
/* The same from this rank's SLICES only (no rank holds the whole zkey): zkey_header = the bytes of zkey section 2 (protocol 1337),
 * coefs = section 4 without its 4-byte count (NULL: this rank runs no H-polynomial chain and keeps no coefficient matrix),
 * points_* = the rank's slices of sections 5 (A), 6 (B1), 7 (B2), 8 (round points C1), 9 (final points C2), 12 (H) and
 * round_indexes / final_round_indexes its slices of the index lists 10 / 11 (u32 each), all starting at the first element of the
 * ranges ug_ultra_groth_shard_ranges reports: out[8] = {witness first, end | round set first, end | final set first, end | H first, end}.
 * slice_bytes = the byte counts of the eight buffers in that order (A, B1, B2, C1, C2, H, round_indexes, final_round_indexes): a slice
 * shorter than the rank's range is refused instead of read past its end. (Sharding of src/ultra_groth.cpp:401-462.) */
int ug_ultra_groth_shard_ranges(unsigned long long n_vars, unsigned long long domain_size, unsigned long long n_round_indexes,
                                unsigned long long n_final_indexes, int shard_rank, int shard_count, unsigned long long out[8]);
int ug_ultra_groth_prover_create_sharded_slices(void **prover_object, const void *zkey_header, unsigned long long zkey_header_size,
                                                const void *coefs, unsigned long long n_coefs,
                                                const void *points_a, const void *points_b1, const void *points_b2,
                                                const void *points_round_c, const void *points_final_c, const void *points_h,
                                                const void *round_indexes, const void *final_round_indexes,
                                                const unsigned long long slice_bytes[8],
                                                int device, int shard_rank, int shard_count,
                                                char *error_msg, unsigned long long error_msg_maxsize);
int ug_ultra_groth_prover_round_commit(void *prover_object, void *commit_part_out,
                                       char *error_msg, unsigned long long error_msg_maxsize);
int ug_ultra_groth_prover_round_finish(void *prover_object, const void *commit_sum, void *commitment_out,
                                       char *error_msg, unsigned long long error_msg_maxsize);
int ug_ultra_groth_prover_apply_commitment(void *prover_object, const void *commitment,
                                           char *error_msg, unsigned long long error_msg_maxsize);
/* acc += other for two 64-byte G1 affine records (all zero = infinity) */
int ug_g1_record_add(void *acc, const void *other);

#ifdef __cplusplus
}
#endif
#endif
