// msm.hip -- Pippenger bucket multi-scalar multiplication in BN254 G1 and G2 for gfx950.
//
// Replaces Curve::multiMulByScalarMSM(r, bases, scalars, 32, n) of the reference's un-vendored
// ffiasm submodule at its five call sites in the prover (src/groth16.cpp:55,58,61,64,154;
// UltraGroth twins src/ultra_groth.cpp:168,201,214,227,234,322). Scalars are 32-byte plain integers.
//
// Pipeline (one "schedule" per scalar vector, shared by every base set multiplied by it):
//   1. signed digits         scalar -> signed c-bit digits (carry recoding), one (key, val) pair per window, made inside the first
//                            pass of the sort (sort.hip; digit_pairs only for scalars of more than 16 windows):
//                            key = bucket id (one bucket set for all windows when the base set has window tables, else
//                            window * 2^(c-1) + |digit| - 1), val = index | table << 27 | sign << 31; zero digits get a
//                            sentinel key and fall off the end of the sort
//   2. radix partition       sort.hip: hand-written LSD passes with a decoupled look-back group the pairs by bucket
//                            (UG_SORT=cub: the library sort of rounds 1-2, kept for A/B runs)
//   3. bucket_bounds / bucket_counts   first entry and entry count of every bucket; buckets cut by segment boundaries are
//                            listed by size class. Counts stay on the device (meta): no host read-back
//   4. transpose_entries     lane-transposed copy of the entries: the sorted list is cut into segments (segmap.hpp), one
//                            lane each, so that every lane of a wave does the same number of additions
//   5. segment_accumulate    the dominant kernels: gather affine bases, mixed-add into XYZZ registers; whole buckets go to
//                            bucket_pts, runs cut by a segment boundary to two slots per lane. G1 sets that share their scalars
//                            (A | B1 | C) are one interleaved group: segment_accumulate_group_kernel keeps K accumulators
//      bucket_fixup / medium_bucket / heavy_partial + heavy_final   add the pieces of cut buckets (a lane, a wave or
//                            workgroup tasks per bucket: witnesses are full of 0/1 values, i.e. million-entry buckets)
//   6. bucket_chunk_reduce   running-sum trick on chunks of up to 32 buckets + ec_sum_groups / ec_sum_wave tree
//   7. host                  Horner over the window sums (none with window tables), affine conversion
// Steps 5b-6 run once for all products of a curve queued together (msm_enqueue_multi).
//
// Arithmetic volume dominates: each of the n * windows gathered bases costs one mixed addition
// (G1: 8 mul + 2 sqr in Fq, G2: the same in Fq2). Algorithmic HBM bytes: 96 n (G1) / 160 n (G2).
#ifdef UG_MEASURE
#include <hipcub/hipcub.hpp>      // the library sort of rounds 1-2, kept for A/B runs (UG_SORT=cub) in -DUG_MEASURE builds only
#endif
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "dev_common.hpp"
#include "internal.hpp"
#include "segmap.hpp"

namespace ug {

namespace {

constexpr int LOG_SEG = 7;            // at most 2^7 entries per lane of the segmented accumulation (see pick_log_seg)
constexpr u32 FIX_MAX = 32;           // buckets cut into more pieces than this take the block-parallel path
constexpr u32 MEDIUM_MAX = 4096;      // up to this many pieces: one wave per bucket; above: two-level heavy path
constexpr u32 HEAVY_TASK = 1024;      // pieces summed by one workgroup of the heavy path
constexpr int CHUNK = 32;             // buckets per running-sum chunk, at most (reduce_chunk)

// ---- curve configurations -------------------------------------------------------------------------
struct G1Cfg {
    typedef Fq F;
    static constexpr int AFF_WORDS = 16, PT_WORDS = 36, BLOCK = 256, ACC_WAVES = 3;      // accumulate kernel: 168 VGPRs, 3 waves/SIMD
    static constexpr bool PREFETCH = true;  // (4 waves at 128 VGPRs without the prefetch registers: 18.1 vs 15.2 ms)
    static __device__ __forceinline__ bool load_affine(const u32* p, F& x, F& y) {
        u32 w[16];
        load8(w, p); load8(w + 8, p + 8);
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) o |= w[i];
        if (o == 0) return false;
        x = unpack256<FqParams>(w); y = unpack256<FqParams>(w + 8);
        return true;
    }
    // raw record words already in registers -> coordinates; false for the all-zero record (infinity)
    static __device__ __forceinline__ bool decode_affine(const u32* w, F& x, F& y) {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) o |= w[i];
        if (o == 0) return false;
        x = unpack256<FqParams>(w); y = unpack256<FqParams>(w + 8);
        return true;
    }
    static __device__ __forceinline__ void load_raw(u32* w, const u32* p) { load8(w, p); load8(w + 8, p + 8); }
    static __host__ __device__ __forceinline__ void store_affine_mont256(u32* o, const F& x, const F& y) {
        to_mont256(o, x); to_mont256(o + 8, y);
    }
    static __host__ __device__ __forceinline__ void store_affine_packed(u32* o, const F& x, const F& y) {
        pack256(o, cond_sub_q(mul(x, fp_one<FqParams>()))); pack256(o + 8, cond_sub_q(mul(y, fp_one<FqParams>())));
    }
    static __host__ __device__ __forceinline__ void to_words(u32* p, const XYZZ<F>& a, size_t stride) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            p[(size_t)i * stride] = a.x.l[i]; p[(size_t)(NL + i) * stride] = a.y.l[i];
            p[(size_t)(2 * NL + i) * stride] = a.zz.l[i]; p[(size_t)(3 * NL + i) * stride] = a.zzz.l[i];
        }
    }
    static __host__ __device__ __forceinline__ XYZZ<F> from_words(const u32* p, size_t stride) {
        XYZZ<F> a;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            a.x.l[i] = p[(size_t)i * stride]; a.y.l[i] = p[(size_t)(NL + i) * stride];
            a.zz.l[i] = p[(size_t)(2 * NL + i) * stride]; a.zzz.l[i] = p[(size_t)(3 * NL + i) * stride];
        }
        return a;
    }
};
struct G2Cfg {
    typedef Fq2 F;
    static constexpr int AFF_WORDS = 32, PT_WORDS = 72, BLOCK = 128, ACC_WAVES = 2;      // 256 VGPRs with ~30 spilled registers beat 1 wave/SIMD at 392 (41.5 vs 43.4 ms)
    static constexpr bool PREFETCH = false; // ... once the 32 prefetch registers are given up; the second wave hides the gather
    static __device__ __forceinline__ bool load_affine(const u32* p, F& x, F& y) {
        u32 w[32];
        load8(w, p); load8(w + 8, p + 8); load8(w + 16, p + 16); load8(w + 24, p + 24);
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) o |= w[i];
        if (o == 0) return false;
        x.a = unpack256<FqParams>(w); x.b = unpack256<FqParams>(w + 8);
        y.a = unpack256<FqParams>(w + 16); y.b = unpack256<FqParams>(w + 24);
        return true;
    }
    static __device__ __forceinline__ bool decode_affine(const u32* w, F& x, F& y) {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) o |= w[i];
        if (o == 0) return false;
        x.a = unpack256<FqParams>(w); x.b = unpack256<FqParams>(w + 8);
        y.a = unpack256<FqParams>(w + 16); y.b = unpack256<FqParams>(w + 24);
        return true;
    }
    static __device__ __forceinline__ void load_raw(u32* w, const u32* p) {
        load8(w, p); load8(w + 8, p + 8); load8(w + 16, p + 16); load8(w + 24, p + 24);
    }
    static __host__ __device__ __forceinline__ void store_affine_mont256(u32* o, const F& x, const F& y) {
        to_mont256(o, x.a); to_mont256(o + 8, x.b); to_mont256(o + 16, y.a); to_mont256(o + 24, y.b);
    }
    static __host__ __device__ __forceinline__ void store_affine_packed(u32* o, const F& x, const F& y) {
        const Fq* f[4] = {&x.a, &x.b, &y.a, &y.b};
        for (int k = 0; k < 4; k++) pack256(o + 8 * k, cond_sub_q(mul(*f[k], fp_one<FqParams>())));
    }
    static __host__ __device__ __forceinline__ void to_words(u32* p, const XYZZ<F>& a, size_t stride) {
        const Fq* f[8] = {&a.x.a, &a.x.b, &a.y.a, &a.y.b, &a.zz.a, &a.zz.b, &a.zzz.a, &a.zzz.b};
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int i = 0; i < NL; i++) p[(size_t)(k * NL + i) * stride] = f[k]->l[i];
    }
    static __host__ __device__ __forceinline__ XYZZ<F> from_words(const u32* p, size_t stride) {
        XYZZ<F> a;
        Fq* f[8] = {&a.x.a, &a.x.b, &a.y.a, &a.y.b, &a.zz.a, &a.zz.b, &a.zzz.a, &a.zzz.b};
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int i = 0; i < NL; i++) f[k]->l[i] = p[(size_t)(k * NL + i) * stride];
        return a;
    }
};

// ---- 1. digits: sort.hip (recode_scalar, digit_key; the pairs are made inside the first pass of the sort) ----------------------

// ---- 3. bucket bounds -----------------------------------------------------------------------------------
// meta[0] = number of heavy buckets, meta[1] = number of non-sentinel entries
// (four entries per lane, one 16-byte load: with one entry per lane the launch was bound by wave dispatch, 0.65 ms for
// 0.8 GB of keys at 2^24)
// (known != 0: the sort dropped the zero digits, the keys end after meta[1] entries -- what lies behind is not the sentinel but
// whatever the buffer held)
__global__ void bucket_bounds_kernel(const u32* keys, u64 total_bound, u32 sentinel, u32* start, u32* count, u32* meta, int known) {
    const u64 total = known ? (u64)meta[1] : total_bound;
    const u64 p0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p0 >= total) return;
    u32 k[6];                                   // k[0] = the entry before the four, k[5] = the one after (sentinel past the end)
    if (p0 + 4 <= total) {
        const uint4 v = *reinterpret_cast<const uint4*>(keys + p0);
        k[1] = v.x; k[2] = v.y; k[3] = v.z; k[4] = v.w;
    } else {
        for (int i = 0; i < 4; i++) k[1 + i] = p0 + i < total ? keys[p0 + i] : sentinel;
    }
    k[0] = p0 ? keys[p0 - 1] : sentinel;
    k[5] = p0 + 4 < total ? keys[p0 + 4] : sentinel;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 p = p0 + i;
        const u32 key = k[1 + i];
        if (p >= total || key == sentinel) continue;
        if (p == 0 || k[i] != key) start[key] = (u32)p;
        if (k[2 + i] != key) count[key] = (u32)p + 1;                   // end for now (past the end reads as the sentinel)
        if (k[2 + i] == sentinel) meta[1] = (u32)p + 1;
    }
}
// buckets whose entries span more than FIX_MAX lanes of the segmented accumulation are listed as heavy
// The list of buckets with 2 .. FIX_MAX pieces takes most buckets of a large MSM (one and a half million at 2^24): its
// slots are handed out per WORKGROUP -- lanes count themselves in LDS, one lane adds the workgroup's total to meta[4].
// (Same-address atomics are served one after the other by L2, about 11 ns each: one per wave, which is what the compiler
// makes of a per-lane atomicAdd, is 32 768 of them = 0.33 of this kernel's 0.38 ms; one per 1 024 lanes is 2 048.)
__global__ __launch_bounds__(1024) void bucket_counts_kernel(u32* start, u32* count, u32 nb, int log_a, int log_b, u32* meta, u32* heavy_list,
                                                           u32* medium_list, u32 heavy_cap, u32* small_list) {
    __shared__ u32 listed, list_base;
    if (threadIdx.x == 0) listed = 0;
    __syncthreads();
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0, first = 0, last = 0;
    if (b < nb) {
        const u32 e = count[b];
        c = e ? e - start[b] : 0;
        count[b] = c;
        if (c) { const SegMap map = SegMap::make(meta[1], log_a, log_b); first = map.seg_of(start[b]); last = map.seg_of(start[b] + c - 1); }
    }
    const u32 pieces = c ? last - first + 1 : 0;
    const bool small = pieces >= 2 && pieces <= FIX_MAX;      // at most one entry per bucket: nb slots
    u32 slot = 0;
    if (small) slot = atomicAdd(&listed, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && listed) list_base = atomicAdd(&meta[4], listed);
    __syncthreads();
    if (small) small_list[list_base + slot] = b;
    if (pieces > MEDIUM_MAX) {
        u32 pos = atomicAdd(&meta[0], 1u);
        if (pos < heavy_cap) { heavy_list[4 * pos] = b; heavy_list[4 * pos + 1] = first; heavy_list[4 * pos + 2] = last; heavy_list[4 * pos + 3] = 0; }
    } else if (pieces > FIX_MAX) {
        u32 pos = atomicAdd(&meta[3], 1u);
        if (pos < heavy_cap) { medium_list[4 * pos] = b; medium_list[4 * pos + 1] = first; medium_list[4 * pos + 2] = last; medium_list[4 * pos + 3] = 0; }
    }
}

// compile-time loop: the bodies below are far above the compiler's pragma-unroll budget, and a rolled loop would
// index the per-point arrays dynamically, i.e. keep them in scratch memory
template <int I, int N> struct StaticFor {
    template <class Fn> static __device__ __forceinline__ void up(Fn&& f) { f(std::integral_constant<int, I>()); StaticFor<I + 1, N>::up(f); }
    template <class Fn> static __device__ __forceinline__ void down(Fn&& f) { StaticFor<I + 1, N>::down(f); f(std::integral_constant<int, I>()); }
};
template <int N> struct StaticFor<N, N> {
    template <class Fn> static __device__ __forceinline__ void up(Fn&&) {}
    template <class Fn> static __device__ __forceinline__ void down(Fn&&) {}
};

// ---- 4. bucket accumulation -----------------------------------------------------------------------------
// Balanced, segmented form: the sorted entry list is cut into equal segments of 2^log_seg entries, one
// lane per segment, so every lane of a wave performs the same number of mixed additions whatever the
// bucket sizes are. A lane walks its segment run by run (a run = consecutive entries of one bucket):
//   - a run that begins and ends inside the segment is a whole bucket: written straight to bucket_pts;
//   - a run cut by the segment's start goes to slot 2t, a run cut only by its end to slot 2t + 1.
// bucket_fixup_kernel then adds the pieces of every bucket that straddles segments (normally two).
// Entries are stored "lane-transposed": the 2^log_seg entries of the 64 segments handled by one wave are
// interleaved so that step k of all 64 lanes reads 64 consecutive words (one fully coalesced 256-byte access).
// one workgroup per tile of 64 segments: coalesced reads of the tile's 64 * 2^log consecutive entries, transposition
// through LDS (row stride padded by one word), coalesced writes
__global__ __launch_bounds__(1024) void transpose_entries_kernel(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                                const u32* __restrict__ meta, u32 sentinel, int log_a, int log_b,
                                                                u32* __restrict__ tkeys, u32* __restrict__ tvals) {
    extern __shared__ u32 tile[];                                   // 2 arrays of 64 * (2^log_a + 1) words
    const u32 n_valid = meta[1];                                    // (the grid covers every entry: the host never reads the count)
    const SegMap map = SegMap::make(n_valid, log_a, log_b);
    const u32 seg = blockIdx.x << 6;                                // first segment of the tile
    const u64 base = map.first_entry(seg);
    if (base >= n_valid) return;
    const int log_seg = map.log_len(seg);
    const u32 S = 1u << log_seg, row = S + 1;
    u32* tk = tile;
    u32* tv = tile + 64 * row;
    for (u32 i = threadIdx.x; i < (64u << log_seg); i += blockDim.x) {
        u64 src = base + i;
        bool in = src < n_valid;
        u32 sg = i >> log_seg, k = i & (S - 1);
        tk[sg * row + k] = in ? keys[src] : sentinel;
        tv[sg * row + k] = in ? vals[src] : 0u;
    }
    __syncthreads();
    for (u32 o = threadIdx.x; o < (64u << log_seg); o += blockDim.x) {
        u32 k = o >> 6, lane = o & 63;
        tkeys[base + o] = tk[lane * row + k];
        tvals[base + o] = tv[lane * row + k];
    }
}

template <class Cfg>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(Cfg::ACC_WAVES, Cfg::ACC_WAVES))) void segment_accumulate_kernel(const u32* __restrict__ bases, u64 n_bases, int64_t delta,
                                                                 const u32* __restrict__ keys, const u32* __restrict__ tkeys,
                                                                 const u32* __restrict__ tvals, const u32* __restrict__ meta, int log_a, int log_b,
                                                                 u32* __restrict__ bucket_pts, u32* __restrict__ slot_pts) {
    typedef typename Cfg::F F;
    const u32 n_valid = meta[1];
    const SegMap map = SegMap::make(n_valid, log_a, log_b);
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 lo = map.first_entry(t);
    if (lo >= n_valid) return;
    u64 hi = lo + ((u64)1 << map.log_len(t));
    if (hi > n_valid) hi = n_valid;
    const u32 cnt = (u32)(hi - lo);
    const u64 tbase = map.transposed(t, 0);
    u32 cur = tkeys[tbase];
    bool first_run = true;
    const bool start_open = lo > 0 && keys[lo - 1] == cur;
    XYZZ<F> acc = xyzz_inf<F>();
    // entry = scalar index | table << 27 | sign << 31; table j of a base set starts j * n_bases records in
    const u32 IDX_MASK = (1u << TABLE_INDEX_BITS) - 1;
    auto close_run = [&](u32 key) {                // the previous run ended inside the segment
        u32* dst = (first_run && start_open) ? slot_pts + (size_t)(2 * t) * Cfg::PT_WORDS : bucket_pts + (size_t)cur * Cfg::PT_WORDS;
        Cfg::to_words(dst, acc, 1);
        acc = xyzz_inf<F>();
        cur = key;
        first_run = false;
    };
    if constexpr (Cfg::PREFETCH) {
        // software pipeline: the raw record of entry k + 1 is in flight while entry k is added
        u32 raw[Cfg::AFF_WORDS];
        u32 nkey = cur, nval = tvals[tbase];
        int64_t nidx = (int64_t)(nval & IDX_MASK) + delta;
        bool nin = nidx >= 0 && (u64)nidx < n_bases;
        if (nin) Cfg::load_raw(raw, bases + ((u64)((nval >> TABLE_INDEX_BITS) & 15u) * n_bases + (u64)nidx) * Cfg::AFF_WORDS);
        for (u32 k = 0; k < cnt; k++) {
            F x, y;
            bool valid = nin && Cfg::decode_affine(raw, x, y);
            if (valid && (nval >> 31)) y = neg<1>(y);
            u32 key = nkey;
            if (k + 1 < cnt) {
                nkey = tkeys[tbase + ((u64)(k + 1) << 6)];
                nval = tvals[tbase + ((u64)(k + 1) << 6)];
                nidx = (int64_t)(nval & IDX_MASK) + delta;
                nin = nidx >= 0 && (u64)nidx < n_bases;
                if (nin) Cfg::load_raw(raw, bases + ((u64)((nval >> TABLE_INDEX_BITS) & 15u) * n_bases + (u64)nidx) * Cfg::AFF_WORDS);
            }
            if (key != cur) close_run(key);
            if (valid) acc = xyzz_madd(acc, x, y);
        }
    } else {
        // no prefetch registers: a second resident wave hides the gather instead
        for (u32 k = 0; k < cnt; k++) {
            const u32 key = tkeys[tbase + ((u64)k << 6)], val = tvals[tbase + ((u64)k << 6)];
            const int64_t idx = (int64_t)(val & IDX_MASK) + delta;
            F x, y;
            bool valid = idx >= 0 && (u64)idx < n_bases &&
                         Cfg::load_affine(bases + ((u64)((val >> TABLE_INDEX_BITS) & 15u) * n_bases + (u64)idx) * Cfg::AFF_WORDS, x, y);
            if (valid && (val >> 31)) y = neg<1>(y);
            if (key != cur) close_run(key);
            if (valid) acc = xyzz_madd(acc, x, y);
        }
    }
    const bool end_open = hi < n_valid && keys[hi] == cur;
    u32* dst;
    if (first_run && start_open) dst = slot_pts + (size_t)(2 * t) * Cfg::PT_WORDS;
    else if (end_open) dst = slot_pts + (size_t)(2 * t + 1) * Cfg::PT_WORDS;
    else dst = bucket_pts + (size_t)cur * Cfg::PT_WORDS;
    Cfg::to_words(dst, acc, 1);
}

// The same walk for a GROUP of K G1 base sets that are multiplied by the same scalars (A, B1 and C of a Groth16 proof, A and
// B1 of an UltraGroth one: src/groth16.cpp:55,58,64 all read the witness), stored as ONE array of K-point records
// [set0_i | set1_i | ...] (K * 64 bytes; a slot without a point holds the all-zero record, infinity; window table j starts
// j * n_slots records in). One lane keeps K accumulators and adds the K points of every entry, so the entry list is read
// once instead of K times, run boundaries and loop control are paid once, and a gather touches K adjacent 64-byte pieces --
// a third (half) of the DRAM row activations that K separate tables cost (1.9 ms of a 15 ms launch are
// the row rate of random 64-byte reads). 3 * 36 accumulator registers put the kernel at two waves per SIMD, where the bare
// addition runs as fast as at three (profiles/r02_ubench_madd.txt). Only ONE 16-register record is in flight: the loads
// rotate through the members -- while member m is added, member m + 1 (or member 0 of the next entry) is on its way, a whole
// mixed addition (3 us) ahead of its use.
// Measured at 2^24 (round 3, profiles/r03_variants_ab.txt): the launch takes what the three single launches took (44.2 ms
// against 3 x 14.8): the kernel is bound by the instruction stream of the addition, so what the group saves is the entry
// list's two extra reads and two launches, not time. With every gather folded into cache (UG_GROUP_FOLD_LOG, a -DUG_MEASURE
// build) it takes 40.4 ms: the gathers' latency is worth 9 %, but neither a rolled loop over rotating accumulators (one
// addition's code instead of K copies; measured in round 3, removed) nor all K records of an entry fetched together by LDS-DMA
// (global_load_lds_dwordx4 into a per-wave region, built and verified bit-exact, 43.8 against 43.5 ms, not kept) gets any of
// it back: the limit is the latency of a random access into 36 GiB (translation + DRAM), one addition's time ahead is all a
// lane can look with its registers full, and adjacent pieces do not make that access shorter. Touching the records two or three
// turns ahead with one-dword LDS-DMA loads (no registers) makes the launch 9 % SLOWER (48.6 ms): vector-memory results retire in
// order, so the real load behind a touch that misses waits for that miss -- the latency moves, it does not shrink.
#ifdef UG_MEASURE
#define UG_FOLD_PARAM , u32 fold_mask
#else
#define UG_FOLD_PARAM
#endif
template <int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void segment_accumulate_group_kernel(
        const u32* __restrict__ bases, u64 n_slots, int64_t delta, const u32* __restrict__ keys, const u32* __restrict__ tkeys,
        const u32* __restrict__ tvals, const u32* __restrict__ meta, int log_a, int log_b, u32* __restrict__ bucket_pts,
        u32* __restrict__ slot_pts, size_t bucket_stride, size_t slot_stride UG_FOLD_PARAM) {
    typedef G1Cfg Cfg;
    typedef Fq F;
    const u32 n_valid = meta[1];
    const SegMap map = SegMap::make(n_valid, log_a, log_b);
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 lo = map.first_entry(t);
    if (lo >= n_valid) return;
    u64 hi = lo + ((u64)1 << map.log_len(t));
    if (hi > n_valid) hi = n_valid;
    const u32 cnt = (u32)(hi - lo);
    const u64 tbase = map.transposed(t, 0);
    u32 cur = tkeys[tbase];
    bool first_run = true;
    const bool start_open = lo > 0 && keys[lo - 1] == cur;
    XYZZ<F> acc[K];
    StaticFor<0, K>::up([&](auto m) { acc[m] = xyzz_inf<F>(); });
    const u32 IDX_MASK = (1u << TABLE_INDEX_BITS) - 1;
    auto record = [&](u32 val, int64_t idx, int m) -> const u32* {
#ifdef UG_MEASURE
        // measurement builds only -- UG_GROUP_FOLD_LOG folds every gather into the first 2^log records of table 0, i.e. into cache:
        // what the launch would take without DRAM misses; the sums are then WRONG (~0u leaves the addresses alone)
        if (fold_mask != ~0u) return bases + (((u64)idx & fold_mask) * K + (u64)m) * Cfg::AFF_WORDS;
#endif
        return bases + (((u64)((val >> TABLE_INDEX_BITS) & 15u) * n_slots + (u64)idx) * K + (u64)m) * Cfg::AFF_WORDS;
    };
    auto flush = [&](bool to_slot, size_t slot, u32 bucket) {
        StaticFor<0, K>::up([&](auto m) {
            u32* dst = to_slot ? slot_pts + (size_t)m * slot_stride + slot * Cfg::PT_WORDS
                               : bucket_pts + (size_t)m * bucket_stride + (size_t)bucket * Cfg::PT_WORDS;
            Cfg::to_words(dst, acc[m], 1);
            acc[m] = xyzz_inf<F>();
        });
    };
    u32 raw[Cfg::AFF_WORDS];
    u32 nkey = cur, nval = tvals[tbase];
    int64_t nidx = (int64_t)(nval & IDX_MASK) + delta;
    bool nin = nidx >= 0 && (u64)nidx < n_slots;
    if (nin) Cfg::load_raw(raw, record(nval, nidx, 0));
    for (u32 k = 0; k < cnt; k++) {
        const u32 key = nkey, val = nval;
        const int64_t idx = nidx;
        const bool in = nin;
        const bool more = k + 1 < cnt;
        if (more) {
            nkey = tkeys[tbase + ((u64)(k + 1) << 6)];
            nval = tvals[tbase + ((u64)(k + 1) << 6)];
            nidx = (int64_t)(nval & IDX_MASK) + delta;
            nin = nidx >= 0 && (u64)nidx < n_slots;
        }
        if (key != cur) {                          // the previous run ended inside the segment
            flush(first_run && start_open, (size_t)2 * t, cur);
            cur = key;
            first_run = false;
        }
        StaticFor<0, K>::up([&](auto m) {
            F x, y;
            const bool valid = in && Cfg::decode_affine(raw, x, y);
            if (m + 1 < K) { if (in) Cfg::load_raw(raw, record(val, idx, m + 1)); }
            else if (more && nin) Cfg::load_raw(raw, record(nval, nidx, 0));
            if (valid && (val >> 31)) y = neg<1>(y);
            if (valid) acc[m] = xyzz_madd(acc[m], x, y);
        });
    }
    const bool end_open = hi < n_valid && keys[hi] == cur;
    if (first_run && start_open) flush(true, (size_t)2 * t, 0);
    else if (end_open) flush(true, (size_t)2 * t + 1, 0);
    else flush(false, 0, cur);
}

// piece k of a bucket whose entries start in segment `first`: k = 0 is the end-cut run of `first`,
// k >= 1 the start-cut run of segment first + k
template <class Cfg>
__device__ __forceinline__ XYZZ<typename Cfg::F> load_piece(const u32* slot_pts, u32 first, u32 k) {
    size_t slot = k ? (size_t)2 * (first + k) : (size_t)2 * first + 1;
    return Cfg::from_words(slot_pts + slot * Cfg::PT_WORDS, 1);
}

// Buckets cut into 2 .. FIX_MAX pieces (listed by bucket_counts_kernel; the order of the list is arbitrary, the sums are
// not): one lane per listed bucket, so the lanes of a wave are all busy. (A lane per segment boundary left three quarters
// of the lanes idle when a bucket is cut several times -- the short segments of many-GPU shards: 2.4 ms of a 22 ms
// witness phase in kernels that run one wave per SIMD.) Buckets with more pieces go to the wave / workgroup paths.
template <class Cfg>
__global__ __launch_bounds__(128) void bucket_fixup_kernel(const u32* __restrict__ small_list, const u32* __restrict__ meta,
                                                           const u32* __restrict__ start, const u32* __restrict__ count, int log_a, int log_b,
                                                           const u32* __restrict__ slot_pts, u32* __restrict__ bucket_pts,
                                                           size_t slot_stride, size_t bucket_stride) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= meta[4]) return;
    slot_pts += blockIdx.y * slot_stride; bucket_pts += blockIdx.y * bucket_stride;      // blockIdx.y: which product of the batch
    typedef typename Cfg::F F;
    const u32 b = small_list[t];
    const u32 st = start[b];
    const SegMap map = SegMap::make(meta[1], log_a, log_b);
    const u32 first = map.seg_of(st), last = map.seg_of(st + count[b] - 1);
    const u32 pieces = last - first + 1;
    XYZZ<F> acc = load_piece<Cfg>(slot_pts, first, 0);
    for (u32 k = 1; k < pieces; k++) acc = xyzz_add(acc, load_piece<Cfg>(slot_pts, first, k));
    Cfg::to_words(bucket_pts + (size_t)b * Cfg::PT_WORDS, acc, 1);
}

// workgroup tree reduction of one XYZZ per thread through LDS ([word][thread] planes); result in thread 0
template <class Cfg>
__device__ __forceinline__ XYZZ<typename Cfg::F> block_reduce(XYZZ<typename Cfg::F> acc, u32* lds) {
    const int tid = threadIdx.x, nth = Cfg::BLOCK;
    for (int stride = nth / 2; stride > 0; stride >>= 1) {
        if (tid >= stride && tid < 2 * stride) Cfg::to_words(lds + (tid - stride), acc, stride);
        __syncthreads();
        if (tid < stride) acc = xyzz_add(acc, Cfg::from_words(lds + tid, stride));
        __syncthreads();
    }
    return acc;
}

// Medium buckets (FIX_MAX < pieces <= MEDIUM_MAX; the short top window of uniform scalars makes thousands of
// them): one wave per bucket, lanes stride over the pieces, then a 6-step cross-lane butterfly of full additions.
template <class Cfg>
__global__ __launch_bounds__(256) void medium_bucket_kernel(const HeavyBucket* mb, const u32* meta, u32 cap, const u32* slot_pts, u32* bucket_pts,
                                                            size_t slot_stride, size_t bucket_stride) {
    typedef typename Cfg::F F;
    slot_pts += blockIdx.y * slot_stride; bucket_pts += blockIdx.y * bucket_stride;
    const u32 lane = threadIdx.x & 63, n_waves = (gridDim.x * blockDim.x) >> 6;
    const u32 n_medium = meta[3] < cap ? meta[3] : cap;
    // the waves stride over the list: the host knows only an upper bound of its length (tens of thousands at 2^24, where
    // uniform scalars list none: a grid of that size took 0.16 ms to find nothing to do)
    for (u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < n_medium; wave += n_waves) {
        HeavyBucket h = mb[wave];
        u32 pieces = h.last_seg - h.first_seg + 1;
        XYZZ<F> acc = xyzz_inf<F>();
        for (u32 k = lane; k < pieces; k += 64) acc = xyzz_add(acc, load_piece<Cfg>(slot_pts, h.first_seg, k));
        u32 w[Cfg::PT_WORDS];
        for (int off = 32; off > 0; off >>= 1) {
            Cfg::to_words(w, acc, 1);
#pragma unroll
            for (int i = 0; i < Cfg::PT_WORDS; i++) w[i] = __shfl_xor(w[i], off, 64);
            acc = xyzz_add(acc, Cfg::from_words(w, 1));
        }
        if (lane == 0) Cfg::to_words(bucket_pts + (size_t)h.bucket * Cfg::PT_WORDS, acc, 1);
    }
}

// Heavy buckets (a 0/1-heavy witness, or the short top window of uniform scalars, puts up to millions of
// entries in one bucket) are reduced in two levels so that the whole chip works on them:
//   heavy_plan      one workgroup: tasks per heavy bucket = ceil(pieces / HEAVY_TASK), exclusive scan -> offsets
//   heavy_partial   one workgroup per task: sums up to HEAVY_TASK pieces (lanes stride, then LDS tree)
//   heavy_final     one workgroup per heavy bucket: sums its task partials, writes the bucket
__global__ __launch_bounds__(1024) void heavy_plan_kernel(const HeavyBucket* hb, u32 n_heavy_cap, u32* meta, u32* offsets) {
    __shared__ u32 part[1024];
    const u32 n = meta[0] < n_heavy_cap ? meta[0] : n_heavy_cap;
    const u32 per = (n + 1023) / 1024;
    u32 lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    u32 sum = 0;
    for (u32 i = lo; i < hi; i++) sum += (hb[i].last_seg - hb[i].first_seg + 1 + HEAVY_TASK - 1) / HEAVY_TASK;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int i = 0; i < 1024; i++) { u32 v = part[i]; part[i] = run; run += v; }
        meta[2] = run;
        offsets[n] = run;
    }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (u32 i = lo; i < hi; i++) {
        offsets[i] = run;
        run += (hb[i].last_seg - hb[i].first_seg + 1 + HEAVY_TASK - 1) / HEAVY_TASK;
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::BLOCK) void heavy_partial_kernel(const HeavyBucket* hb, const u32* offsets, const u32* meta, u32 cap,
                                                                   const u32* slot_pts, u32* task_pts, size_t slot_stride, size_t task_stride) {
    __shared__ u32 lds[Cfg::PT_WORDS * Cfg::BLOCK / 2];
    typedef typename Cfg::F F;
    slot_pts += blockIdx.y * slot_stride; task_pts += blockIdx.y * task_stride;
    const u32 task = blockIdx.x;
    const u32 n_heavy = meta[0] < cap ? meta[0] : cap;
    if (task >= meta[2]) return;                       // (uniform per workgroup: the grid is an upper bound)
    u32 lo = 0, hi = n_heavy;                          // last heavy bucket whose first task is <= task
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (offsets[mid] <= task) lo = mid; else hi = mid; }
    HeavyBucket h = hb[lo];
    u32 pieces = h.last_seg - h.first_seg + 1;
    u32 p0 = (task - offsets[lo]) * HEAVY_TASK, p1 = p0 + HEAVY_TASK < pieces ? p0 + HEAVY_TASK : pieces;
    XYZZ<F> acc = xyzz_inf<F>();
    for (u32 k = p0 + threadIdx.x; k < p1; k += Cfg::BLOCK) acc = xyzz_add(acc, load_piece<Cfg>(slot_pts, h.first_seg, k));
    acc = block_reduce<Cfg>(acc, lds);
    if (threadIdx.x == 0) Cfg::to_words(task_pts + (size_t)task * Cfg::PT_WORDS, acc, 1);
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::BLOCK) void heavy_final_kernel(const HeavyBucket* hb, const u32* offsets, const u32* meta, u32 cap,
                                                                 const u32* task_pts, u32* bucket_pts, size_t task_stride, size_t bucket_stride) {
    __shared__ u32 lds[Cfg::PT_WORDS * Cfg::BLOCK / 2];
    typedef typename Cfg::F F;
    task_pts += blockIdx.y * task_stride; bucket_pts += blockIdx.y * bucket_stride;
    if (blockIdx.x >= (meta[0] < cap ? meta[0] : cap)) return;
    HeavyBucket h = hb[blockIdx.x];
    u32 t0 = offsets[blockIdx.x], t1 = offsets[blockIdx.x + 1];
    XYZZ<F> acc = xyzz_inf<F>();
    for (u32 k = t0 + threadIdx.x; k < t1; k += Cfg::BLOCK) acc = xyzz_add(acc, Cfg::from_words(task_pts + (size_t)k * Cfg::PT_WORDS, 1));
    acc = block_reduce<Cfg>(acc, lds);
    if (threadIdx.x == 0) Cfg::to_words(bucket_pts + (size_t)h.bucket * Cfg::PT_WORDS, acc, 1);
}

// ---- 5. bucket reduction --------------------------------------------------------------------------------
// thread = (window, chunk of `chunk` buckets): P = sum_k (k_global + 1) * bucket_k over the chunk
template <class Cfg>
// (w counts the bucket sets of ALL products of a batch, one after the other in `buckets`; the entry counts belong to the
// schedule, which the products share: bucket set w of any product has the counts of set w mod windows)
// (out0, optional -- schedules with bucket classes: the plain sum of the chunk's buckets, S0 of internal.hpp's BucketClasses)
__global__ __launch_bounds__(128) void bucket_chunk_reduce_kernel(const u32* buckets, const u32* count, u32 per_window, int chunk,
                                                                  u32 nchunks_total, int scalar_bits, u32* out, u32 windows, u32* out0,
                                                                  size_t product_buckets) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks_total) return;
    typedef typename Cfg::F F;
    u32 chunks_per_window = per_window / chunk;
    u32 w = t / chunks_per_window, j = t % chunks_per_window;
    // (product_buckets: the buckets of one product -- its `windows` sets and, with bucket classes, the special buckets behind them)
    const size_t b0 = (size_t)(w / windows) * product_buckets + (size_t)(w % windows) * per_window + (size_t)j * chunk;
    const size_t c0 = (size_t)(w % windows) * per_window + (size_t)j * chunk;
    const u32* base = buckets + b0 * Cfg::PT_WORDS;
    XYZZ<F> run = xyzz_inf<F>(), acc = xyzz_inf<F>();
    for (int k = chunk - 1; k >= 0; k--) {
        if (count[c0 + k]) run = xyzz_add(run, Cfg::from_words(base + (size_t)k * Cfg::PT_WORDS, 1));   // empty bucket = infinity
        acc = xyzz_add(acc, run);
    }
    u32 off = j * (u32)chunk;                  // weight offset of the chunk
    if (off) acc = xyzz_add(acc, xyzz_mul_scalar(run, &off, scalar_bits));
    Cfg::to_words(out + (size_t)t * Cfg::PT_WORDS, acc, 1);
    if (out0) Cfg::to_words(out0 + (size_t)t * Cfg::PT_WORDS, run, 1);
}
// Schedules with bucket classes: the special buckets of window set w (ids special_base + w * S + b, b < S <= 64: the digits
// 1 .. S, kept by scalar range) -> their weighted sum  sum_b (b + 1) X_b. One wave per (window set, product of the batch).
template <class Cfg>
__global__ __launch_bounds__(64) void special_sum_kernel(const u32* bucket_pts, const u32* count, u32 special_base, u32 S, size_t bucket_stride,
                                                         u32* out, u32 window_sets) {
    typedef typename Cfg::F F;
    const u32 w = blockIdx.x, q = blockIdx.y, lane = threadIdx.x;
    XYZZ<F> x = xyzz_inf<F>();
    const u32 id = special_base + w * S + lane;
    if (lane < S && count[id]) x = Cfg::from_words(bucket_pts + (size_t)q * bucket_stride + (size_t)id * Cfg::PT_WORDS, 1);
    XYZZ<F> acc = xyzz_inf<F>();
    const u32 k = lane + 1;                                    // <= 64: seven bits
    for (int i = 6; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k >> i) & 1) acc = xyzz_add(acc, x);
    }
    u32 v[Cfg::PT_WORDS];
    for (int off = 32; off > 0; off >>= 1) {
        Cfg::to_words(v, acc, 1);
#pragma unroll
        for (int i = 0; i < Cfg::PT_WORDS; i++) v[i] = __shfl_xor(v[i], off, 64);
        acc = xyzz_add(acc, Cfg::from_words(v, 1));
    }
    if (lane == 0) Cfg::to_words(out + ((size_t)q * window_sets + w) * Cfg::PT_WORDS, acc, 1);
}
template <class Cfg>
__global__ __launch_bounds__(128) void ec_sum_groups_kernel(const u32* in, u32* out, u32 n_out, int group) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out) return;
    typedef typename Cfg::F F;
    XYZZ<F> acc = Cfg::from_words(in + (size_t)t * group * Cfg::PT_WORDS, 1);
    for (int g = 1; g < group; g++) acc = xyzz_add(acc, Cfg::from_words(in + ((size_t)t * group + g) * Cfg::PT_WORDS, 1));
    Cfg::to_words(out + (size_t)t * Cfg::PT_WORDS, acc, 1);
}

// The same sum with one WAVE per output: lanes stride over the group, then a cross-lane butterfly. A single lane's
// dependent EC additions run at ~5 us each, so the short last levels of the tree are latency-bound: 16 serial additions
// per level become group/64 + 6.
template <class Cfg>
__global__ __launch_bounds__(256) void ec_sum_wave_kernel(const u32* in, u32* out, u32 n_out, int group) {
    typedef typename Cfg::F F;
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_out) return;
    XYZZ<F> acc = xyzz_inf<F>();
    for (int g = (int)lane; g < group; g += 64) acc = xyzz_add(acc, Cfg::from_words(in + ((size_t)wave * group + g) * Cfg::PT_WORDS, 1));
    u32 w[Cfg::PT_WORDS];
    for (int off = 32; off > 0; off >>= 1) {
        Cfg::to_words(w, acc, 1);
#pragma unroll
        for (int i = 0; i < Cfg::PT_WORDS; i++) w[i] = __shfl_xor(w[i], off, 64);
        acc = xyzz_add(acc, Cfg::from_words(w, 1));
    }
    if (lane == 0) Cfg::to_words(out + (size_t)wave * Cfg::PT_WORDS, acc, 1);
}

// ---- zkey point conversion ------------------------------------------------------------------------------
__global__ void convert_coords_kernel(u32* pts, u64 n_coords_groups, int coords_per_point) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_coords_groups) return;
    u32* p = pts + i * 8 * coords_per_point;
    u32 o = 0;
    for (int k = 0; k < 8 * coords_per_point; k++) o |= p[k];
    if (o == 0) return;                                     // infinity stays (0,0)
    for (int k = 0; k < coords_per_point; k++) {
        u32 w[8];
        load8(w, p + 8 * k);
        Fq v = cond_sub_q(from_mont256<FqParams>(w));
        pack256(w, v);
        store8(p + 8 * k, w);
    }
}

// records of one member set -> their places in a group array: dst record (slot0 + i) * members + member  (64-byte G1 records)
__global__ void interleave_records_kernel(u32* __restrict__ dst, const u32* __restrict__ src, u64 n, int members, int member, u64 slot0) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte quarter of a record per lane
    if (g >= n * 4) return;
    const u64 i = g >> 2, part = g & 3;
    reinterpret_cast<uint4*>(dst + ((slot0 + i) * (u64)members + (u64)member) * 16)[part] = reinterpret_cast<const uint4*>(src + i * 16)[part];
}

// ---- fixed-base window tables ------------------------------------------------------------------------------
// pts holds `tables` tables of n affine records; table 0 is given, table j = 2^(c j) * table 0. A lane carries K
// consecutive points through c doublings per table and shares one field inversion among them for the conversion
// back to affine (Montgomery's trick); infinity stays (0,0) in every table.
// (the launch covers the points [first, end) of the n: a whole set at once, or one piece of a deferred build)
template <class Cfg, int K>
__global__ __launch_bounds__(128) void window_tables_kernel(u32* pts, u64 n_all, int c, int tables, u64 first, u64 end) {
    typedef typename Cfg::F F;
    const u64 i0 = first + ((u64)blockIdx.x * blockDim.x + threadIdx.x) * K;
    const u64 n = end;                     // lanes stop at the piece's end; the tables' stride is n_all
    if (i0 >= n) return;
    F x[K], y[K];
    bool live[K];
    StaticFor<0, K>::up([&](auto k) { live[k] = i0 + k < n && Cfg::load_affine(pts + (i0 + k) * Cfg::AFF_WORDS, x[k], y[k]); });
    for (int j = 1; j < tables; j++) {
        XYZZ<F> p[K];
        StaticFor<0, K>::up([&](auto k) { p[k] = live[k] ? xyzz_dbl_affine(x[k], y[k]) : xyzz_inf<F>(); });
        for (int d = 1; d < c; d++)
            StaticFor<0, K>::up([&](auto k) { p[k] = xyzz_dbl(p[k]); });     // infinity stays infinity
        F pre[K];
        F run = field_one((F*)0);
        StaticFor<0, K>::up([&](auto k) {
            pre[k] = run;
            if (live[k]) run = mulk<8>(run, p[k].zzz);
        });
        F irun = inv(run);                                               // 1 / prod zzz_k
        StaticFor<0, K>::down([&](auto k) {
            if (i0 + k >= n) return;
            u32* o = pts + ((u64)j * n_all + i0 + k) * Cfg::AFF_WORDS;
            if (live[k]) {
                F izzz = mulk<8>(irun, pre[k]);                          // 1 / zzz_k
                irun = mulk<8>(irun, p[k].zzz);
                F iz = mulk<8>(izzz, p[k].zz);                           // zz / zzz = 1 / z
                F izz = sqrk<8>(iz);
                x[k] = canon(mulk<8>(p[k].x, izz));
                y[k] = canon(mulk<8>(p[k].y, izzz));
                Cfg::store_affine_packed(o, x[k], y[k]);
            } else {
                for (int w = 0; w < Cfg::AFF_WORDS; w++) o[w] = 0;
            }
        });
    }
}

// ---- synthetic base points (bench / test tooling): record i = (seed + i) * G ------------------------------
// table: 64 affine records 2^j * G in device form; output: zkey-format records (Montgomery R = 2^256)
template <class Cfg>
__global__ __launch_bounds__(128) void synth_points_kernel(const u32* table, u64 seed, u64 n, u32* out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef typename Cfg::F F;
    u64 k = seed + i;
    XYZZ<F> acc = xyzz_inf<F>();
    for (int j = 0; j < 64; j++) {
        if (!((k >> j) & 1)) continue;
        F x, y;
        Cfg::load_affine(table + (size_t)j * Cfg::AFF_WORDS, x, y);
        acc = xyzz_madd(acc, x, y);
    }
    u32* o = out + i * Cfg::AFF_WORDS;
    if (is_inf(acc)) { for (int w = 0; w < Cfg::AFF_WORDS; w++) o[w] = 0; return; }
    F ax, ay;
    xyzz_to_affine(ax, ay, acc);
    Cfg::store_affine_mont256(o, ax, ay);
}

// buckets per thread of the reduction: a single lane's chain of dependent EC additions runs at ~5 us each, so the
// kernel is latency-bound below ~2 waves/SIMD: halve the chunk (down to 8) until there are 2^17 threads
// (sets: the products of a batch are reduced together, so their chunks add up: three products over 2^21 buckets keep 32-bucket
// chunks, i.e. half the per-chunk scalar multiples -- which are half of this kernel's work -- of a single product)
int reduce_chunk(const MsmGeometry& g, int sets = 1, bool g2 = false) {
    static const int g2_log = getenv("UG_REDUCE_G2_LOG") ? atoi(getenv("UG_REDUCE_G2_LOG")) : 16;      // tuning knobs
    static const int g1_log = getenv("UG_REDUCE_G1_LOG") ? atoi(getenv("UG_REDUCE_G1_LOG")) : 0;
    // G1: 2^17 lanes, but 2^16 when the whole batch has fewer than 2^21 buckets -- a rank of an eight-way shard (three products
    // over 2^19 buckets): there the chunks would fall to 8 buckets, and a chunk's scalar multiple is half of its work (round 5,
    // one rank of eight at 2^24: 22.8 -> 22.3-22.5 ms per step; full-size problems keep 2^17)
    const int g1_auto = (u64)sets * g.bucket_windows() * g.buckets < ((u64)1 << 21) ? 16 : 17;
    const u64 lanes = (u64)1 << (g2 ? g2_log : (g1_log ? g1_log : g1_auto));
    int chunk = g.buckets < (u32)CHUNK ? (int)g.buckets : CHUNK;
    while (chunk > 8 && (u64)sets * g.bucket_windows() * (g.buckets / chunk) < lanes) chunk >>= 1;
    return chunk;
}

template <class T> void dev_alloc(T*& p, size_t bytes) { alloc_epoch_bump(); if (p) hipFree(p); p = nullptr; UG_HIP(hipMalloc(&p, bytes ? bytes : 4)); }
template <class T> void dev_free(T*& p) { if (p) { alloc_epoch_bump(); hipFree(p); } p = nullptr; }
std::atomic<uint64_t> g_alloc_epoch{1};

}  // namespace

uint64_t alloc_epoch() { return g_alloc_epoch.load(std::memory_order_acquire); }
void alloc_epoch_bump() { g_alloc_epoch.fetch_add(1, std::memory_order_acq_rel); }

// ---- geometry ---------------------------------------------------------------------------------------------
namespace {
// entries per lane of the segmented accumulation: long segments cut fewer buckets into pieces, but keep about a
// million lanes in flight
int segment_log(u64 total_entries) {
    static const int lanes_log = getenv("UG_SEG_LANES_LOG") ? atoi(getenv("UG_SEG_LANES_LOG")) : 20;      // tuning knob
    int log_seg = 5;
    while (log_seg < LOG_SEG && (total_entries >> (log_seg + 1)) >= ((u64)1 << lanes_log)) log_seg++;
    return log_seg;
}
// Modelled cost of one MSM in mixed additions: one per entry, about four per bucket for the reduction, and a penalty
// when the average bucket is longer than a lane's segment -- such buckets are cut into pieces that a single lane adds
// up afterwards (measured at 2^20..2^24 with tools/run_tablec.sh: +4 % of the entry cost per segment length by which
// the average bucket exceeds one segment; e.g. 2^22, c = 20 -> 22: 48.2 -> 45.9 ms per proof).
double msm_cost(u64 n, int c, bool tables) {
    const int windows = (255 + c - 1) / c;
    const double entries = (double)windows * (double)n, buckets = (double)((u64)1 << (c - 1));
    const double per_bucket = (tables ? entries : (double)n) / buckets;
    const double over = per_bucket / (double)((u64)1 << segment_log((u64)entries)) - 1.0;
    return entries * (1.0 + (over > 0 ? 0.04 * over : 0.0)) + 4.0 * buckets * (tables ? 1.0 : (double)windows);
}
}  // namespace

MsmGeometry MsmGeometry::choose(u64 n, int force_c) {
    MsmGeometry g;
    g.n = n;
    int c = force_c;
    if (!c) {
        static const int env_c = getenv("UG_MSM_C") ? atoi(getenv("UG_MSM_C")) : 0;      // tuning knob
        c = env_c;
    }
    if (!c) {
        double best = 0;
        for (int k = 6; k <= 22; k++) {
            double cost = msm_cost(n, k, false);
            if (!c || cost < best) { best = cost; c = k; }
        }
    }
    if (c < 2) c = 2;
    if (c > 22) c = 22;
    g.c = c;
    g.windows = (255 + c - 1) / c;
    g.buckets = 1u << (c - 1);
    return g;
}

// Tables mode: W = ceil(255/c) digits per scalar, ONE bucket set of 2^(c-1).
int MsmGeometry::table_window(u64 n) {
    if (const char* e = getenv("UG_TABLE_C")) {                       // tuning knob
        int v = atoi(e);
        if (v >= TABLE_MIN_C && v <= TABLE_MAX_C) return v;
    }
    int c = 0;
    double best = 0;
    for (int k = TABLE_MIN_C; k <= TABLE_MAX_C; k++) {
        double cost = msm_cost(n, k, true);
        if (!c || cost < best) { best = cost; c = k; }
    }
    return c;
}
MsmGeometry MsmGeometry::choose_tables(u64 n, int c) {
    if (c < TABLE_MIN_C || c > TABLE_MAX_C) throw std::invalid_argument("msm: table window width outside [16, 24]");
    MsmGeometry g;
    g.n = n; g.c = c; g.tables = true;
    g.windows = (255 + c - 1) / c;
    g.buckets = 1u << (c - 1);
    return g;
}

void MsmGeometry::set_classes(const BucketClasses& k) {
    cls = BucketClasses();
    if (!k.on()) return;
    if (k.q_log > 8) throw std::invalid_argument("msm: more than 256 bucket classes");
    const u32 Q = 1u << k.q_log;
    if (k.cnt < 1 || k.r0 + k.cnt > Q) throw std::invalid_argument("msm: bucket residues outside [0, 2^q_log)");
    if (k.specials > MSM_MAX_SPECIALS) throw std::invalid_argument("msm: more than 64 special buckets");
    if (k.sp_lo > k.sp_hi || k.sp_hi > n) throw std::invalid_argument("msm: special-bucket scalar range outside the schedule");
    // every owned residue needs a bucket set of at least 8 buckets (the reduction's shortest chunk), and the specials must be
    // bucket ids the window has
    if (c - 1 < k.q_log + 3 || ((u64)1 << (c - 1)) < k.specials) throw std::invalid_argument("msm: window too narrow for this many bucket classes");
    cls = k;
    buckets = 1u << (c - 1 - k.q_log);
    if ((size_t)result_points() * 72 > MSM_PENDING_WORDS - 1) throw std::invalid_argument("msm: too many bucket sets for one result block");
}

// ---- schedule -----------------------------------------------------------------------------------------------
void MsmSchedule::reserve(const MsmGeometry& g) {
    u64 total = g.n * g.windows;
    if (total > capacity_n) {
        u64 padded = total + ((u64)64 << LOG_SEG);          // room for the lane-transposed copy's last tile
        dev_alloc(keys_a, padded * 4); dev_alloc(keys_b, padded * 4);
        dev_alloc(vals_a, padded * 4); dev_alloc(vals_b, padded * 4);
        dev_free(sort_tmp); sort_tmp_bytes = 0;                // (the library sort's scratch is sized on its first use)
        sorter.reserve(total, 12);
        capacity_n = total;
    }
    if (g.total_buckets() > capacity_buckets) {
        dev_alloc(bucket_start, g.total_buckets() * 4);
        dev_alloc(bucket_count, g.total_buckets() * 4);
        dev_alloc(small_list, g.total_buckets() * 4);
        capacity_buckets = g.total_buckets();
    }
    u64 hcap = (total >> 5) / (FIX_MAX - 1) + 2;         // a listed bucket covers at least FIX_MAX - 1 whole segments (>= 2^5 entries each)
    if (hcap > heavy_cap) {
        dev_alloc(heavy_list, 4 * hcap * 4); dev_alloc(medium_list, 4 * hcap * 4); dev_alloc(heavy_offsets, (hcap + 1) * 4);
        heavy_cap = (u32)hcap;
    }
    if (!meta) dev_alloc(meta, 32);
}

void MsmSchedule::build(const u32* scalars_dev, const MsmGeometry& g, hipStream_t stream) {
    // limits first: reserve() hands `total` to the sort as an int
    // (2^30: a look-back status word of the sort carries a 30-bit pair count, sort.hip -- one bin may hold every pair)
    if (g.n * (u64)g.windows > ((u64)1 << 30)) throw std::invalid_argument("msm: n * windows exceeds 2^30 entries");
    if (g.n > ((u64)1 << TABLE_INDEX_BITS)) throw std::invalid_argument("msm: more than 2^27 scalars in one schedule");
    geo = g;
    reserve(g);
    log_seg = segment_log(g.expected_entries());
    // short segments for the last eighth of a large schedule (SegMap): worth it once the accumulation takes several rounds
    // of workgroups (UG_SEG_TAPER=0 turns it off)
    static const bool taper = !(getenv("UG_SEG_TAPER") && atoi(getenv("UG_SEG_TAPER")) == 0);
    log_seg_tail = (taper && log_seg == LOG_SEG && (g.expected_entries() >> log_seg) >= ((u64)1 << 16)) ? log_seg - 2 : log_seg;
    if (g.n == 0) return;
    u64 total = g.n * g.windows;
    u32 nb = (u32)g.total_buckets();
    u32 sentinel = nb;
    int end_bit = 1;
    while (((u64)1 << end_bit) <= sentinel) end_bit++;
    UG_HIP(hipMemsetAsync(meta, 0, 32, stream));
    // the hand-written partition of sort.hip, whose first pass reads the scalars themselves (pairs in unsorted form are written
    // only when a scalar has more than 16 windows). -DUG_MEASURE builds: UG_SORT=cub takes the library sort of rounds 1-2 (A/B runs)
#ifdef UG_MEASURE
    static const bool use_cub = getenv("UG_SORT") && !strcmp(getenv("UG_SORT"), "cub");
#else
    constexpr bool use_cub = false;
#endif
    const bool pairs_first = use_cub || g.windows > 16;
    bool dropped = false;                 // the sort left out the zero digits and wrote the entry count to meta[1] itself
    if (pairs_first) digit_pairs(scalars_dev, g.digit_plan(), keys_a, vals_a, stream);
#ifdef UG_MEASURE
    if (use_cub) {
        hipcub::DoubleBuffer<u32> dk(keys_a, keys_b), dv(vals_a, vals_b);
        if (!sort_tmp_bytes) {
            size_t need = 0;
            UG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, dk, dv, (int)total, 0, 32));
            dev_alloc(sort_tmp, need); sort_tmp_bytes = need;
        }
        size_t tmp = sort_tmp_bytes;
        UG_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, tmp, dk, dv, (int)total, 0, end_bit, stream));
        keys = dk.Current();
        vals = dv.Current();
    } else
#endif
    {
        u32* const bk[2] = {keys_a, keys_b};
        u32* const bv[2] = {vals_a, vals_b};
        const int at = sorter.sort(pairs_first ? nullptr : scalars_dev, g, sentinel, total, end_bit, bk, bv, meta + 7, stream, meta + 1, &dropped);
        keys = bk[at];
        vals = bv[at];
    }
    UG_HIP(hipMemsetAsync(bucket_start, 0, (size_t)nb * 4, stream));
    UG_HIP(hipMemsetAsync(bucket_count, 0, (size_t)nb * 4, stream));
    hipLaunchKernelGGL(bucket_bounds_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, stream,
                       keys, total, sentinel, bucket_start, bucket_count, meta, dropped ? 1 : 0);
    UG_KERNEL_CHECK();
    hipLaunchKernelGGL(bucket_counts_kernel, dim3((nb + 1023) / 1024), dim3(1024), 0, stream, bucket_start, bucket_count, nb,
                       log_seg, log_seg_tail, meta, heavy_list, medium_list, heavy_cap, small_list);
    UG_KERNEL_CHECK();
    hipLaunchKernelGGL(heavy_plan_kernel, dim3(1), dim3(1024), 0, stream, (const HeavyBucket*)heavy_list, heavy_cap, meta, heavy_offsets);
    UG_KERNEL_CHECK();
    // No read-back of the counts: every later launch is sized by an upper bound derived from n * windows and reads the
    // true counts (meta) on the device, so a schedule costs the host no synchronisation.
    // lane-transposed copy of the valid entries into the sort's spare buffers
    u64 n_tiles = (SegMap::max_segments(total, log_seg, log_seg_tail) + 63) / 64;
    u32* tk = (keys == keys_a) ? keys_b : keys_a;
    u32* tv = (vals == vals_a) ? vals_b : vals_a;
    {
        size_t lds = (size_t)2 * 64 * (((size_t)1 << log_seg) + 1) * 4;
        hipLaunchKernelGGL(transpose_entries_kernel, dim3((unsigned)n_tiles), dim3(1024), lds, stream,
                           keys, vals, meta, sentinel, log_seg, log_seg_tail, tk, tv);
        UG_KERNEL_CHECK();
    }
    tkeys = tk; tvals = tv;
}

void MsmSchedule::release() {
    dev_free(keys_a); dev_free(keys_b); dev_free(vals_a); dev_free(vals_b); dev_free(sort_tmp);
    dev_free(bucket_start); dev_free(bucket_count); dev_free(small_list); dev_free(heavy_list); dev_free(medium_list); dev_free(heavy_offsets); dev_free(meta);
    sorter.release();
    capacity_n = capacity_buckets = 0; sort_tmp_bytes = 0; vals = nullptr; keys = nullptr; heavy_cap = 0;
}

// ---- workspace ------------------------------------------------------------------------------------------------
void MsmWorkspace::reserve(const MsmGeometry& g, bool g2, u64 n_segments, u32 n_heavy_tasks, int sets) {
    size_t ptw = (size_t)(g2 ? G2Cfg::PT_WORDS : G1Cfg::PT_WORDS) * (size_t)sets;      // the products of a batch lie one after the other
    size_t need = (size_t)g.total_buckets() * ptw * 4;
    if (need > bucket_bytes) { dev_alloc(bucket_pts, need); bucket_bytes = need; }
    const int chunk = reduce_chunk(g, sets, g2);           // the chunk the launch uses (msm_enqueue_multi): same arguments, same value
    // (bucket classes: the chunks' plain sums S0 beside their weighted sums, and the specials' sums behind both)
    size_t cneed = ((size_t)(g.cls.on() ? 2 : 1) * g.bucket_windows() * (g.buckets / chunk) + (size_t)g.window_sets()) * ptw * 4;
    if (cneed > chunk_bytes) { dev_alloc(chunk_pts, cneed); dev_alloc(chunk_pts2, cneed); chunk_bytes = cneed; }
    size_t sneed = (size_t)n_segments * 2 * ptw * 4;
    if (sneed > slot_bytes) { dev_alloc(slot_pts, sneed); slot_bytes = sneed; }
    size_t tneed = (size_t)n_heavy_tasks * ptw * 4;
    if (tneed > task_bytes) { dev_alloc(task_pts, tneed); task_bytes = tneed; }
}
void MsmWorkspace::release() {
    dev_free(bucket_pts); dev_free(chunk_pts); dev_free(chunk_pts2); dev_free(slot_pts); dev_free(task_pts);
    bucket_bytes = chunk_bytes = slot_bytes = task_bytes = 0;
}

// ---- driver -------------------------------------------------------------------------------------------------------
namespace {
// A batch of products over ONE schedule (A, B1 and C of a Groth16 proof share the witness schedule): the accumulation kernel is
// launched once per product, each into its own bucket and slot arrays, and everything after it -- fix-up of cut buckets,
// bucket reduction, tree sums: low-occupancy kernels that wait on dependent EC additions -- once for the whole batch, the
// products side by side in the grid (blockIdx.y, or extra bucket sets for the reduction). Measured at 2^24: the fix-up,
// medium-bucket and tree-sum launches of A, B1, C become one launch each (- 0.3 ms per proof); the bucket reduction itself
// is bound by its arithmetic (two full additions per bucket and a scalar multiple per chunk), not by latency, and gains
// only through the longer chunks a batch allows (reduce_chunk).
constexpr int MSM_MAX_BATCH = 4;
// group: 0 = `count` separate base arrays; K = bases[0] is ONE interleaved array of K-member records (n_bases[0] slots; count
// == K): a single launch of segment_accumulate_group_kernel accumulates all K products
// phase: MSM_PHASE_ALL, or the two halves of the same call made one after the other with the same arguments -- MSM_PHASE_ACCUMULATE
// (the accumulation launches) and MSM_PHASE_TAIL (everything behind them, possibly on another stream that has been ordered behind
// the accumulation): ug_msm_witness_enqueue runs the G1 and the G2 tails of a proof side by side
template <class Cfg>
void msm_enqueue_multi(const MsmSchedule& s, MsmWorkspace& ws, int count, const u32* const* bases, const u64* n_bases, const int64_t* delta,
                       hipStream_t stream, MsmStats* stats, u32* const* pinned_host, MsmPending* pend, int group = 0, int phase = MSM_PHASE_ALL) {
    const bool do_acc = phase != MSM_PHASE_TAIL, do_tail = phase != MSM_PHASE_ACCUMULATE;
    typedef typename Cfg::F F;
    const MsmGeometry& g = s.geo;
    if (count < 1 || count > MSM_MAX_BATCH) throw std::logic_error("msm: batch size");
    // products without points (or an empty schedule) are the point at infinity and take no part
    int live[MSM_MAX_BATCH], k = 0;
    for (int j = 0; j < count; j++) {
        pend[j] = MsmPending();
        pend[j].g2 = Cfg::PT_WORDS == G2Cfg::PT_WORDS;
        pend[j].c = g.c; pend[j].window_sets = g.window_sets(); pend[j].class_sets = g.class_sets(); pend[j].cls = g.cls;
        pend[j].host = pinned_host[j];
        if (g.n == 0 || n_bases[group ? 0 : j] == 0) continue;
        if ((size_t)g.result_points() * Cfg::PT_WORDS > MSM_PENDING_WORDS - 1) throw std::logic_error("msm: result block too large");
        pend[j].empty = false;
        live[k++] = j;
    }
    if (!k) return;
    // upper bounds of what the schedule holds (the true counts stay on the device, MsmSchedule::meta):
    // segments; heavy buckets (each spans more than MEDIUM_MAX segments); their 1024-piece tasks; medium buckets
    const u64 total = g.n * g.windows;
    const u64 nseg = SegMap::max_segments(total, s.log_seg, s.log_seg_tail);
    const u32 heavy_max = (u32)std::min<u64>(s.heavy_cap, nseg / MEDIUM_MAX + 1);
    const u32 tasks_max = (u32)((nseg + heavy_max) / HEAVY_TASK + heavy_max + 1);
    const u32 medium_max = (u32)std::min<u64>(s.heavy_cap, nseg / (FIX_MAX - 1) + 1);
    if (do_acc) ws.reserve(g, Cfg::PT_WORDS == G2Cfg::PT_WORDS, nseg, tasks_max, k);      // (the tail half finds what the first half reserved)
    // word strides between the arrays of consecutive products
    const size_t bucket_stride = (size_t)g.total_buckets() * Cfg::PT_WORDS, slot_stride = (size_t)nseg * 2 * Cfg::PT_WORDS,
                 task_stride = (size_t)tasks_max * Cfg::PT_WORDS;
    if constexpr (Cfg::PT_WORDS == G1Cfg::PT_WORDS) {
        if (group && do_acc) {                       // (all K products are live, or none: they share the slot count)
#ifdef UG_MEASURE
            const char* fold = getenv("UG_GROUP_FOLD_LOG");                   // measurement knob (WRONG sums): see the kernel
            const u32 fold_mask = fold && *fold ? ((1u << atoi(fold)) - 1) : ~0u;
#define UG_FOLD_ARG , fold_mask
#else
#define UG_FOLD_ARG
#endif
            int slot = stats ? stats->begin(stream, g.n * g.windows * (u64)group) : -1;
            if (nseg) {
                const dim3 grid((unsigned)((nseg + 255) / 256)), block(256);
#define UG_GROUP_LAUNCH(K_) hipLaunchKernelGGL((segment_accumulate_group_kernel<K_>), grid, block, 0, stream, bases[0], n_bases[0], \
                                                   delta[0], s.keys, s.tkeys, s.tvals, s.meta, s.log_seg, s.log_seg_tail, ws.bucket_pts, ws.slot_pts,   \
                                                   bucket_stride, slot_stride UG_FOLD_ARG)
                if (group == 3) UG_GROUP_LAUNCH(3);
                else if (group == 2) UG_GROUP_LAUNCH(2);
                else throw std::logic_error("msm: group size");
#undef UG_GROUP_LAUNCH
#undef UG_FOLD_ARG
                UG_KERNEL_CHECK();
            }
            if (stats) stats->end(slot, stream);
        }
    } else if (group) throw std::logic_error("msm: groups are G1 only");
    for (int q = 0; q < (group || !do_acc ? 0 : k); q++) {
        const int j = live[q];
        int slot = stats ? stats->begin(stream, g.n * g.windows) : -1;
        if (nseg) {
            hipLaunchKernelGGL(segment_accumulate_kernel<Cfg>, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, stream,
                               bases[j], n_bases[j], delta[j], s.keys, s.tkeys, s.tvals, s.meta, s.log_seg, s.log_seg_tail,
                               ws.bucket_pts + q * bucket_stride, ws.slot_pts + q * slot_stride);
            UG_KERNEL_CHECK();
        }
        if (stats) stats->end(slot, stream);
    }
    if (!do_tail) return;
    if (nseg > 1) {
        const u64 small_max = std::min<u64>(g.total_buckets(), nseg);       // a listed bucket crosses a segment boundary of its own
        hipLaunchKernelGGL(bucket_fixup_kernel<Cfg>, dim3((unsigned)((small_max + 127) / 128), k), dim3(128), 0, stream,
                           s.small_list, s.meta, s.bucket_start, s.bucket_count, s.log_seg, s.log_seg_tail, ws.slot_pts, ws.bucket_pts,
                           slot_stride, bucket_stride);
        UG_KERNEL_CHECK();
        if (nseg > FIX_MAX) {
            hipLaunchKernelGGL(medium_bucket_kernel<Cfg>, dim3(std::min<u32>((medium_max + 3) / 4, 2048), k), dim3(256), 0, stream,
                               (const HeavyBucket*)s.medium_list, s.meta, medium_max, ws.slot_pts, ws.bucket_pts, slot_stride, bucket_stride);
            UG_KERNEL_CHECK();
        }
        if (nseg > MEDIUM_MAX) {
            hipLaunchKernelGGL(heavy_partial_kernel<Cfg>, dim3(tasks_max, k), dim3(Cfg::BLOCK), 0, stream,
                               (const HeavyBucket*)s.heavy_list, s.heavy_offsets, s.meta, heavy_max, ws.slot_pts, ws.task_pts,
                               slot_stride, task_stride);
            UG_KERNEL_CHECK();
            hipLaunchKernelGGL(heavy_final_kernel<Cfg>, dim3(heavy_max, k), dim3(Cfg::BLOCK), 0, stream,
                               (const HeavyBucket*)s.heavy_list, s.heavy_offsets, s.meta, heavy_max, ws.task_pts, ws.bucket_pts,
                               task_stride, bucket_stride);
            UG_KERNEL_CHECK();
        }
    }
    int chunk = reduce_chunk(g, k, Cfg::PT_WORDS == G2Cfg::PT_WORDS);
    const int windows = g.bucket_windows();            // bucket sets per product: one per window, or one in all with window tables
                                                       // (times the owned residues of a schedule with bucket classes)
    const int bw = windows * k;                        // ... of the whole batch, product after product
    const bool classes = g.cls.on();
    u32 cpw = g.buckets / chunk;                       // chunks per bucket set
    u32 nchunks = cpw * bw;
    // with bucket classes the chunks' plain sums (S0) lie behind their weighted sums and go through the same tree: 2 bw sets
    hipLaunchKernelGGL(bucket_chunk_reduce_kernel<Cfg>, dim3((nchunks + 127) / 128), dim3(128), 0, stream,
                       ws.bucket_pts, s.bucket_count, g.buckets, chunk, nchunks, g.c, ws.chunk_pts, (u32)windows,
                       classes ? ws.chunk_pts + (size_t)nchunks * Cfg::PT_WORDS : (u32*)nullptr, (size_t)g.total_buckets());
    UG_KERNEL_CHECK();
    const int tree_sets = classes ? 2 * bw : bw;
    u32* cur = ws.chunk_pts; u32* nxt = ws.chunk_pts2;
    while (cpw > 1) {
        // a lane per output (16 serial additions) while that still fills the chip, then a wave per output
        if ((u64)(cpw / 16) * tree_sets >= 8192) {
            const int group = 16;
            u32 n_out = (cpw / group) * tree_sets;
            hipLaunchKernelGGL(ec_sum_groups_kernel<Cfg>, dim3((n_out + 127) / 128), dim3(128), 0, stream, cur, nxt, n_out, group);
            cpw /= group;
        } else {
            const int group = cpw >= 256 ? 256 : (int)cpw;
            u32 n_out = (cpw / group) * tree_sets;
            hipLaunchKernelGGL(ec_sum_wave_kernel<Cfg>, dim3((n_out + 3) / 4), dim3(256), 0, stream, cur, nxt, n_out, group);
            cpw /= group;
        }
        UG_KERNEL_CHECK();
        std::swap(cur, nxt);
    }
    // result block of a product: [S1 per set | S0 per set | weighted sum of the special buckets per window set]
    const int wsets = g.window_sets();
    const bool specials = classes && g.cls.specials;
    if (specials) {
        hipLaunchKernelGGL(special_sum_kernel<Cfg>, dim3((unsigned)wsets, (unsigned)k), dim3(64), 0, stream, ws.bucket_pts, s.bucket_count,
                           (u32)((u64)windows * g.buckets), g.cls.specials, bucket_stride, cur + (size_t)tree_sets * Cfg::PT_WORDS, (u32)wsets);
        UG_KERNEL_CHECK();
    }
    for (int q = 0; q < k; q++) {
        u32* host = pinned_host[live[q]];
        const size_t set_words = (size_t)windows * Cfg::PT_WORDS;
        UG_HIP(hipMemcpyAsync(host, cur + (size_t)q * set_words, set_words * 4, hipMemcpyDeviceToHost, stream));
        if (classes) UG_HIP(hipMemcpyAsync(host + set_words, cur + ((size_t)bw + (size_t)q * windows) * Cfg::PT_WORDS, set_words * 4, hipMemcpyDeviceToHost, stream));
        if (specials) UG_HIP(hipMemcpyAsync(host + 2 * set_words, cur + ((size_t)tree_sets + (size_t)q * wsets) * Cfg::PT_WORDS,
                                            (size_t)wsets * Cfg::PT_WORDS * 4, hipMemcpyDeviceToHost, stream));
        UG_HIP(hipMemcpyAsync(host + MSM_PENDING_WORDS - 1, s.meta + 7, 4, hipMemcpyDeviceToHost, stream));
    }
}
template <class Cfg>
MsmPending msm_enqueue(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta,
                       hipStream_t stream, MsmStats* stats, u32* pinned_host) {
    MsmPending pend;
    msm_enqueue_multi<Cfg>(s, ws, 1, &bases, &n_bases, &delta, stream, stats, &pinned_host, &pend);
    return pend;
}

// Horner over the window sets, top first (one set with window tables). With bucket classes (internal.hpp: BucketClasses) a
// window's sum is put together from its owned residues' sets: a digit in residue r = r0 + j, local bucket k, has the magnitude
// Q (k + 1) - m_j with m_j = Q - 1 - r0 - j, so the window's sum is  Q * sum_j S1_j  -  sum_j m_j S0_j  +  the specials' sum;
// m_j falls by one per set, so the middle term is m_last * (sum of all S0) plus the running sums of S0_0 .. S0_(cnt-2).
template <class Cfg>
XYZZ<typename Cfg::F> msm_collect(const MsmPending& p) {
    typedef typename Cfg::F F;
    XYZZ<F> acc = xyzz_inf<F>();
    if (p.empty) return acc;
    if (p.host[MSM_PENDING_WORDS - 1]) throw std::runtime_error("msm: the schedule's sort gave up waiting for a tile (look-back timeout)");
    const int sets = p.window_sets * p.class_sets;
    auto point = [&](int idx) { return Cfg::from_words(p.host + (size_t)idx * Cfg::PT_WORDS, 1); };
    for (int w = p.window_sets - 1; w >= 0; w--) {
        for (int k = 0; k < p.c; k++) acc = xyzz_dbl(acc);
        if (!p.cls.on()) { acc = xyzz_add(acc, point(w)); continue; }
        const int cnt = p.class_sets;
        XYZZ<F> s1 = xyzz_inf<F>(), s0_all = xyzz_inf<F>(), run = xyzz_inf<F>(), steps = xyzz_inf<F>();
        for (int j = 0; j < cnt; j++) {
            s1 = xyzz_add(s1, point(w * cnt + j));
            const XYZZ<F> s0 = point(sets + w * cnt + j);
            s0_all = xyzz_add(s0_all, s0);
            if (j + 1 < cnt) { run = xyzz_add(run, s0); steps = xyzz_add(steps, run); }      // sum_j (cnt - 1 - j) S0_j
        }
        for (int k = 0; k < p.cls.q_log; k++) s1 = xyzz_dbl(s1);                              // Q * sum S1
        const u32 m_last = (1u << p.cls.q_log) - p.cls.r0 - (u32)cnt;                          // m of the last owned residue
        XYZZ<F> minus = xyzz_add(steps, xyzz_mul_scalar(s0_all, &m_last, 9));
        XYZZ<F> wsum = xyzz_add(s1, xyzz_neg(minus));
        if (p.cls.specials) wsum = xyzz_add(wsum, point(2 * sets + w));
        acc = xyzz_add(acc, wsum);
    }
    return acc;
}

// the synchronous form: queue, wait, collect (result block in a caller-side buffer, not pinned: the copy then waits)
template <class Cfg>
XYZZ<typename Cfg::F> msm_run(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta,
                              hipStream_t stream, MsmStats* stats) {
    std::vector<u32> host(MSM_PENDING_WORDS);
    MsmPending p = msm_enqueue<Cfg>(s, ws, bases, n_bases, delta, stream, stats, host.data());
    UG_HIP(hipStreamSynchronize(stream));
    if (stats) stats->collect();
    return msm_collect<Cfg>(p);
}
}  // namespace

void MsmStats::create() { for (int i = 0; i < SLOTS; i++) { UG_HIP(hipEventCreate(&ev0[i])); UG_HIP(hipEventCreate(&ev1[i])); } }
void MsmStats::destroy() { for (int i = 0; i < SLOTS; i++) { if (ev0[i]) hipEventDestroy(ev0[i]); if (ev1[i]) hipEventDestroy(ev1[i]); ev0[i] = ev1[i] = nullptr; } }
// launches whose end event has already fired are accounted and their slots freed (no host wait)
void MsmStats::collect_ready() {
    int keep = 0;
    for (int i = 0; i < pending; i++) {
        if (hipEventQuery(ev1[i]) == hipSuccess) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev0[i], ev1[i]) == hipSuccess) { accumulate_ms += ms; launches++; entries += slot_entries[i]; }
        } else {
            if (keep != i) { std::swap(ev0[keep], ev0[i]); std::swap(ev1[keep], ev1[i]); slot_entries[keep] = slot_entries[i]; }
            keep++;
        }
    }
    (void)hipGetLastError();                       // hipEventQuery reports hipErrorNotReady through the error state
    pending = keep;
}
int MsmStats::begin(hipStream_t stream, u64 units) {
    if (capture) {                                 // the stream is being captured: an event pair of the graph's own
        CapturedSpan sp;
        sp.units = units;
        UG_HIP(hipEventCreate(&sp.e0));
        if (hipEventCreate(&sp.e1) != hipSuccess) { hipEventDestroy(sp.e0); throw HipError("HIP error: hipEventCreate"); }
        capture->push_back(sp);
        record_in_capture(sp.e0, stream);
        return SLOTS + (int)capture->size() - 1;
    }
    if (pending == SLOTS) collect_ready();
    if (pending == SLOTS) return -1;               // a caller that never waits: this launch goes untimed
    int slot = pending++;
    slot_entries[slot] = units;
    UG_HIP(hipEventRecord(ev0[slot], stream));
    return slot;
}
void MsmStats::end(int slot, hipStream_t stream) {
    if (slot >= SLOTS) {
        if (!capture || (size_t)(slot - SLOTS) >= capture->size()) throw std::logic_error("kernel statistics: captured slot without a capture");
        record_in_capture((*capture)[slot - SLOTS].e1, stream);
        return;
    }
    if (slot >= 0) UG_HIP(hipEventRecord(ev1[slot], stream));
}
// one completed launch of a graph that holds these pairs
void MsmStats::account(const std::vector<CapturedSpan>& spans) {
    for (const CapturedSpan& sp : spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.e0, sp.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        accumulate_ms += ms; launches++; entries += sp.units;
    }
}
void MsmStats::collect() {
    for (int i = 0; i < pending; i++) {
        float ms = 0;
        UG_HIP(hipEventElapsedTime(&ms, ev0[i], ev1[i]));
        accumulate_ms += ms; launches++; entries += slot_entries[i];
    }
    pending = 0;
}

MsmPending msm_enqueue_g1(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream,
                          MsmStats* stats, u32* pinned_host) { return msm_enqueue<G1Cfg>(s, ws, bases, n_bases, delta, stream, stats, pinned_host); }
MsmPending msm_enqueue_g2(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream,
                          MsmStats* stats, u32* pinned_host) { return msm_enqueue<G2Cfg>(s, ws, bases, n_bases, delta, stream, stats, pinned_host); }
void msm_enqueue_batch_g1(const MsmSchedule& s, MsmWorkspace& ws, int count, const u32* const* bases, const u64* n_bases, const int64_t* delta,
                          hipStream_t stream, MsmStats* stats, u32* const* pinned_host, MsmPending* pend) {
    msm_enqueue_multi<G1Cfg>(s, ws, count, bases, n_bases, delta, stream, stats, pinned_host, pend);
}
void msm_enqueue_batch_g2(const MsmSchedule& s, MsmWorkspace& ws, int count, const u32* const* bases, const u64* n_bases, const int64_t* delta,
                          hipStream_t stream, MsmStats* stats, u32* const* pinned_host, MsmPending* pend, int phase) {
    msm_enqueue_multi<G2Cfg>(s, ws, count, bases, n_bases, delta, stream, stats, pinned_host, pend, 0, phase);
}
void msm_enqueue_group_g1(const MsmSchedule& s, MsmWorkspace& ws, int members, const u32* bases, u64 n_slots, int64_t delta, hipStream_t stream,
                          MsmStats* stats, u32* const* pinned_host, MsmPending* pend, int phase) {
    if (members < 2 || members > 3) throw std::invalid_argument("msm: a base group has 2 or 3 members");
    const u32* b[MSM_MAX_BATCH] = {bases, bases, bases, bases};
    const u64 n[MSM_MAX_BATCH] = {n_slots, n_slots, n_slots, n_slots};
    const int64_t d[MSM_MAX_BATCH] = {delta, delta, delta, delta};
    msm_enqueue_multi<G1Cfg>(s, ws, members, b, n, d, stream, stats, pinned_host, pend, members, phase);
}
G1XYZZ msm_collect_g1(const MsmPending& p) { return msm_collect<G1Cfg>(p); }
G2XYZZ msm_collect_g2(const MsmPending& p) { return msm_collect<G2Cfg>(p); }

G1XYZZ msm_g1(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream, MsmStats* stats) {
    return msm_run<G1Cfg>(s, ws, bases, n_bases, delta, stream, stats);
}
G2XYZZ msm_g2(const MsmSchedule& s, MsmWorkspace& ws, const u32* bases, u64 n_bases, int64_t delta, hipStream_t stream, MsmStats* stats) {
    return msm_run<G2Cfg>(s, ws, bases, n_bases, delta, stream, stats);
}

namespace {
template <class Cfg>
void synth_points_run(u32* out_dev, const u32* gen_record_host, u64 seed, u64 n, hipStream_t stream) {
    typedef typename Cfg::F F;
    // 2^j * G for j < 64 on the host, packed device form
    std::vector<u32> table((size_t)64 * Cfg::AFF_WORDS);
    std::vector<u32> gen(gen_record_host, gen_record_host + Cfg::AFF_WORDS);
    XYZZ<F> p;
    {
        F x, y;
        const int nc = Cfg::AFF_WORDS / 8;
        Fq c[4];
        for (int k = 0; k < nc; k++) c[k] = from_mont256<FqParams>(gen.data() + 8 * k);
        if (nc == 2) { memcpy(&x, &c[0], sizeof(Fq)); memcpy(&y, &c[1], sizeof(Fq)); }
        else { memcpy(&x, &c[0], 2 * sizeof(Fq)); memcpy(&y, &c[2], 2 * sizeof(Fq)); }
        p = xyzz_from_affine(x, y);
    }
    for (int j = 0; j < 64; j++) {
        F ax, ay;
        xyzz_to_affine(ax, ay, p);
        Cfg::store_affine_packed(table.data() + (size_t)j * Cfg::AFF_WORDS, ax, ay);
        p = xyzz_dbl(p);
    }
    u32* d_table = nullptr;
    UG_HIP(hipMalloc(&d_table, table.size() * 4));
    UG_HIP(hipMemcpyAsync(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(synth_points_kernel<Cfg>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, d_table, seed, n, out_dev);
    UG_KERNEL_CHECK();
    UG_HIP(hipStreamSynchronize(stream));
    hipFree(d_table);
}
}  // namespace

void synth_points(bool g2, u32* out_dev, const u32* gen_record_host, u64 seed, u64 n, hipStream_t stream) {
    if (!n) return;
    if (g2) synth_points_run<G2Cfg>(out_dev, gen_record_host, seed, n, stream);
    else synth_points_run<G1Cfg>(out_dev, gen_record_host, seed, n, stream);
}

void build_window_tables(bool g2, u32* pts, u64 n, int c, int tables, hipStream_t stream, u64 first, u64 count) {
    if (!n || tables < 2) return;
    if (first > n) first = n;
    const u64 end = count > n - first ? n : first + count, m = end - first;
    if (!m) return;
    if (g2) {
        hipLaunchKernelGGL((window_tables_kernel<G2Cfg, 2>), dim3((unsigned)(((m + 1) / 2 + 127) / 128)), dim3(128), 0, stream, pts, n, c, tables, first, end);
    } else {
        hipLaunchKernelGGL((window_tables_kernel<G1Cfg, 4>), dim3((unsigned)(((m + 3) / 4 + 127) / 128)), dim3(128), 0, stream, pts, n, c, tables, first, end);
    }
    UG_KERNEL_CHECK();
}

void interleave_points_g1(u32* dst, const u32* src, u64 n, int members, int member, u64 slot0, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(interleave_records_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, stream, dst, src, n, members, member, slot0);
    UG_KERNEL_CHECK();
}
void convert_points_g1(u32* pts, u64 n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(convert_coords_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pts, n, 2);
    UG_KERNEL_CHECK();
}
void convert_points_g2(u32* pts, u64 n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(convert_coords_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pts, n, 4);
    UG_KERNEL_CHECK();
}

}  // namespace ug
