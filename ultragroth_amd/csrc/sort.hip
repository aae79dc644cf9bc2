// sort.hip -- grouping of an MSM schedule's (bucket, entry) pairs by bucket: a hand-written LSD radix partition for gfx950.
//
// The reference's multiMulByScalarMSM (call sites src/groth16.cpp:55-64,154; the routine itself lives in the un-vendored
// ffiasm submodule) walks its scalars window by window on the CPU and adds each base into a bucket array in cache. On the
// GPU the same grouping is a sort: every (scalar, window) digit becomes a pair (bucket id, entry = index | table | sign), and
// the accumulation kernel (msm.hip) wants the pairs of one bucket next to each other. Round 1 and 2 used the library sort
// for it; this file replaces it on the per-proof path:
//
//   radix_hist_kernel     one read of the SCALARS (not of a materialised pair array): recodes them into signed window digits
//                         and counts, for every pass of the sort, how many pairs fall into each digit bin
//   radix_scan_kernel     exclusive scan of those counts: where each bin of each pass starts
//   radix_pass_kernel     one pass = one stable partition by up to 8 key bits. A tile of 512 lanes x <= 16 pairs is ranked
//                         inside the workgroup (wave-level match on the digit bits, per-wave counters in LDS), reordered
//                         through LDS so that pairs of one bin leave in runs, and scattered; the offset of a tile within a
//                         bin comes from a decoupled look-back over the tiles before it (one 32-bit status word per tile and
//                         bin: 2 status bits + 30 count bits, so flag and value travel in ONE agent-scope atomic and need no
//                         fence). The FIRST pass reads the scalars and makes its pairs on the fly -- the pair arrays are
//                         never written in unsorted form (1.6 GB written and read again per 2^24-scalar schedule before).
//
// Keys are sorted on the low `bits` bits only (enough to tell the sentinel, = number of buckets, from every bucket id -- or, when
// the first pass drops the sentinel pairs, just the bucket ids: one bit less, 7 | 7 | 7 for 2^21 buckets), in ceil(bits / 8) passes
// of nearly equal width. A pass is launched as at most SORT_MAX_GRID workgroups that take tiles until none is left. Pairs that exist only to fill the last tile carry the key 0xffffffff: they sort
// behind everything and land beyond the `total` positions any consumer reads.
#include <atomic>
#include <cstdlib>
#include "dev_common.hpp"
#include "internal.hpp"

namespace ug {

namespace {

#ifndef UG_SORT_THREADS
#define UG_SORT_THREADS 512           // round 4 (profiles/r04_variants_ab.txt item 5): tiles of 8 192 pairs leave 256-byte runs per bin; 256 lanes:
                                      // 1.17 ms per pass over 201 M pairs, 512: 1.01, 1 024: 1.05
#endif
constexpr int SORT_THREADS = UG_SORT_THREADS;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_MAX_IPT = 16;              // most pairs a lane holds per tile (32 was measured: the longer unrolled bodies cost every shape 10 %)
constexpr int SORT_FUSED_MAX_WINDOWS = 16;    // the fused first pass takes whole scalars per lane: windows * scalars_per_lane <= SORT_MAX_IPT
constexpr int SORT_MAX_BINS = 256;
constexpr u32 LB_VALUE_MASK = (1u << 30) - 1, LB_AGGREGATE = 1u << 30, LB_PREFIX = 2u << 30;
constexpr u32 SORT_MAX_GRID = 4096;           // workgroups of a pass (each takes tiles until none is left): a few per CU slot
constexpr u32 SPIN_LIMIT = 1u << 20;          // a look-back that has not seen its predecessor by then gives up and flags the schedule

struct SortPassArgs {
    // source: either pair arrays ...
    const u32* keys_in; const u32* vals_in;
    // ... or the scalars themselves (first pass): plan.n scalars of 32 bytes, recoded by recode_scalar
    const u32* scalars; DigitPlan plan;
    u32* keys_out; u32* vals_out;
    u64 n_pairs;                      // pairs the source holds (the rest of the last tile is padding made on the fly)
    u32 n_moved;                      // pairs every pass moves (real ones + the first pass's padding): nothing is written beyond
    int ipt;                          // pairs per lane
    int spl;                          // fused first pass: scalars per lane (ipt = spl * windows)
    int shift, bins_log;              // this pass sorts on (key >> shift) & (2^bins_log - 1)
    const u32* bin_base;              // 2^bins_log exclusive bin starts of this pass
    const u32* n_valid;               // drop mode (see RadixSorter::sort): the pairs that exist after the first pass -- on the device
    int drop;                         // drop mode, first pass: pairs with key >= sentinel (zero digits, padding) are not written at all
    u32* lookback;                    // tiles x 2^bins_log status words, zeroed
    u32* tile_counter;                // zeroed: hands out tile numbers in the order they are taken
    u32 n_tiles;                      // fused first pass: its tile count (the other passes end at the pair count)
    u32* error_flag;
};

__device__ __forceinline__ bool scalar_geq_r(const u32* s) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (s[i] > FrParams::q32[i]) return true;
        if (s[i] < FrParams::q32[i]) return false;
    }
    return true;
}
// The bucket key of a digit of magnitude mag >= 1 in window w of scalar i, or the sentinel when this schedule does not keep it.
// Without classes: set w (one set in all with window tables), bucket mag - 1. With classes (internal.hpp: BucketClasses): the
// lowest bucket ids are kept by scalar range and get ids behind the regular sets, the others by the residue of the bucket id.
__device__ __forceinline__ u32 digit_key(const DigitPlan& d, int w, u32 mag, u64 i) {
    const u32 b = mag - 1;
    const u32 wset = d.tables ? 0u : (u32)w;
    if (d.q_log == 0) return wset * d.buckets + b;
    if (b < d.specials) return (i >= d.sp_lo && i < d.sp_hi) ? d.special_base + wset * d.specials + b : d.sentinel;
    const u32 j = (b & ((1u << d.q_log) - 1)) - d.r0;          // (unsigned: residues below r0 wrap to large values)
    return j < d.cnt ? (wset * d.cnt + j) * d.buckets + (b >> d.q_log) : d.sentinel;
}
// the signed-digit recoding of one scalar: digit in (-2^(c-1), 2^(c-1)], a carry into the next window; a zero digit (and a digit
// another rank owns) gets the sentinel key; f(window, key, val) is called for every window in order
// CW: the window width as a compile-time constant (0: taken from the plan). The widths the provers use at size (22: window
// tables of 2^24 points; 20: classic windows and many-GPU ranks) get an unrolled window loop with static limb indices -- the
// runtime form indexes the scalar's limbs dynamically, a chain of selects per window.
template <int CW, class Fn>
__device__ __forceinline__ void recode_scalar(const u32* scalars, u64 i, const DigitPlan& d, Fn&& f) {
    const int windows = CW ? (255 + CW - 1) / CW : d.windows;
    if (i >= d.n) {                                   // padding of the last tile
        for (int w = 0; w < windows; w++) f(w, 0xffffffffu, 0u);
        return;
    }
    u32 s[10];
    load8(s, scalars + i * 8);
    s[8] = 0; s[9] = 0;
    // the group has order r: scalars >= r (never produced by a well-formed witness) are reduced
    for (int it = 0; it < 6 && scalar_geq_r(s); it++) {
        u64 borrow = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            u64 dd = (u64)s[k] - FrParams::q32[k] - borrow;
            s[k] = (u32)dd; borrow = (dd >> 32) & 1;
        }
    }
    u32 carry = 0;
    const int c = CW ? CW : d.c;
    const u32 half = 1u << (c - 1), full = 1u << c, mask = full - 1;
    auto window = [&](int w) {
        const int bit = w * c, word = bit >> 5, sh = bit & 31;
        const u64 two = (u64)s[word] | ((u64)s[word + 1] << 32);
        const u32 raw = ((u32)(two >> sh) & mask) + carry;
        const u32 tag = d.tables ? (u32)w << TABLE_INDEX_BITS : 0u;
        u32 key, val;
        if (raw > half) {                             // negative digit raw - 2^c, carry into the next window
            const u32 mag = full - raw;               // 0 when the window was all ones and a carry came in
            carry = 1;
            key = mag ? digit_key(d, w, mag, i) : d.sentinel;
            val = (u32)i | tag | 0x80000000u;
        } else { carry = 0; key = raw ? digit_key(d, w, raw, i) : d.sentinel; val = (u32)i | tag; }
        f(w, key, val);
    };
    if constexpr (CW != 0) {
#pragma unroll
        for (int w = 0; w < (255 + CW - 1) / CW; w++) window(w);
    } else {
        for (int w = 0; w < windows; w++) window(w);
    }
}

__global__ void digit_pairs_kernel(const u32* scalars, DigitPlan d, u32* keys, u32* vals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    recode_scalar<0>(scalars, i, d, [&](int w, u32 key, u32 val) { keys[(u64)w * d.n + i] = key; vals[(u64)w * d.n + i] = val; });
}

struct SortHistArgs {
    const u32* keys_in;                // pair form (scalars == nullptr) ...
    const u32* scalars; DigitPlan plan;                                                 // ... or scalar form
    u32 sentinel;
    u64 n_pairs, n_padded;             // padded: whole tiles (the padding pairs are counted: they take part in every pass)
    int passes; int shift[4]; int bins_log[4];
    u32* hist;                         // passes x 256 counters, zeroed
    int drop;                          // keys >= sentinel are not counted at all (the first pass will not write them)
};

// counts per pass and bin; a grid-stride loop, counters in LDS, one global atomic per (workgroup, pass, bin)
template <int CW>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(SortHistArgs a) {
    __shared__ u32 h[4 * SORT_MAX_BINS];
    for (int i = threadIdx.x; i < 4 * SORT_MAX_BINS; i += SORT_THREADS) h[i] = 0;
    __syncthreads();
    // The sentinel key (a zero digit) is counted in a register and added once per lane: a circom-like witness is mostly zeros and
    // ones, i.e. nearly every pair of it carries the sentinel, and a workgroup's lanes adding to ONE LDS counter serialise (measured at
    // 2^24, circom-like mix: 1.3 ms for the witness schedule's histogram against 0.32 ms for uniform scalars).
    u32 zeros = 0;
    auto count = [&](u32 key) {
        if (a.drop && key >= a.sentinel) return;
        if (key == a.sentinel) { zeros++; return; }
        for (int p = 0; p < a.passes; p++) atomicAdd(&h[p * SORT_MAX_BINS + ((key >> a.shift[p]) & ((1u << a.bins_log[p]) - 1))], 1u);
    };
    const u64 stride = (u64)gridDim.x * SORT_THREADS;
    if (a.scalars) {
        const u64 lanes = a.n_padded / (u64)a.plan.windows;       // one scalar (all its windows) per lane turn
        for (u64 i = (u64)blockIdx.x * SORT_THREADS + threadIdx.x; i < lanes; i += stride)
            recode_scalar<CW>(a.scalars, i, a.plan, [&](int, u32 key, u32) { count(key); });
    } else {
        for (u64 i = (u64)blockIdx.x * SORT_THREADS + threadIdx.x; i < a.n_padded; i += stride) count(i < a.n_pairs ? a.keys_in[i] : 0xffffffffu);
    }
    if (zeros)
        for (int p = 0; p < a.passes; p++) atomicAdd(&h[p * SORT_MAX_BINS + ((a.sentinel >> a.shift[p]) & ((1u << a.bins_log[p]) - 1))], zeros);
    __syncthreads();
    for (int i = threadIdx.x; i < a.passes * SORT_MAX_BINS; i += SORT_THREADS)
        if (h[i]) atomicAdd(&a.hist[i], h[i]);
}
// hist[p][*] -> exclusive starts, in place (one workgroup, one wave per pass would do: a serial loop over 256 bins is nothing)
__global__ void radix_scan_kernel(u32* hist, int passes, u32* n_valid_out) {
    const int p = threadIdx.x;
    if (p >= passes) return;
    u32 run = 0;
    for (int b = 0; b < SORT_MAX_BINS; b++) { const u32 v = hist[p * SORT_MAX_BINS + b]; hist[p * SORT_MAX_BINS + b] = run; run += v; }
    if (p == 0 && n_valid_out) *n_valid_out = run;      // drop mode: every counted pair is a pair that stays
}

constexpr int LBW = 4;     // look-back window: the status words of four predecessors are read per turn (8 and 16 measured in round 3: no gain)
template <bool FROM_SCALARS, int CW>
__global__ __launch_bounds__(SORT_THREADS) void radix_pass_kernel(SortPassArgs a) {
    extern __shared__ u32 lds[];
    const int ipt = a.ipt;
    const u32 T = (u32)SORT_THREADS * (u32)ipt;
    u32* stage_k = lds;                                   // T keys
    u32* stage_v = lds + T;                               // T values
    u32* wcnt = lds + 2 * T;                              // SORT_WAVES x 256 per-wave counters, later per-wave starts
    u32* texcl = wcnt + SORT_WAVES * SORT_MAX_BINS;       // 256: start of each bin inside the tile
    u32* gbase = texcl + SORT_MAX_BINS;                   // 256: global position of the tile's first pair of a bin, minus its tile position
    u32* misc = gbase + SORT_MAX_BINS;                    // [0] tile number, [1 .. SORT_WAVES] wave totals of the scan
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u32 bins = 1u << a.bins_log, dmask = bins - 1;

    // drop mode, passes after the first: the pair count is the device's (what the first pass kept)
    const u64 n_pairs = (!FROM_SCALARS && a.n_valid) ? (u64)*a.n_valid : a.n_pairs;
    const u32 n_moved = (!FROM_SCALARS && a.n_valid) ? *a.n_valid : a.n_moved;
    // A workgroup takes tiles until none is left (the grid is at most SORT_MAX_GRID workgroups: the host knows only an upper bound
    // of the tile count when the first pass dropped pairs -- a circom-like witness, a schedule with bucket classes -- and a grid
    // of that bound cost one same-address atomic, ~11 ns each, one after the other, per tile that then had nothing to do: 0.46
    // of a 0.63 ms pass over an eighth of the pairs). Tiles are numbered in the order they are TAKEN, so a tile only ever waits
    // for tiles that running workgroups hold.
    for (;;) {
    __syncthreads();                                                  // (the previous turn's LDS is done with)
    if (tid == 0) misc[0] = atomicAdd(a.tile_counter, 1u);
    for (int i = tid; i < SORT_WAVES * SORT_MAX_BINS; i += SORT_THREADS) wcnt[i] = 0;
    __syncthreads();
    const u32 tile = misc[0];
    const u64 e0 = (u64)tile * T;
    if (FROM_SCALARS ? tile >= a.n_tiles : e0 >= n_pairs) return;

    // ---- load (or make) the tile's pairs: lane holds pairs j = 0 .. ipt-1; wave-striped, so that (wave, j, lane) is memory order
    u32 key[SORT_MAX_IPT], val[SORT_MAX_IPT];
    if constexpr (FROM_SCALARS) {
        // whole scalars per lane, pair j = window j of its first scalar, windows + j of its second (any order will do for a
        // first pass)
#pragma unroll
        for (int j = 0; j < SORT_MAX_IPT; j++) { key[j] = 0xffffffffu; val[j] = 0; }
        for (int sc = 0; sc < a.spl; sc++) {
            const u64 i = ((u64)tile * SORT_THREADS + tid) * (u64)a.spl + (u64)sc;
            const int j0 = sc * a.plan.windows;
            recode_scalar<CW>(a.scalars, i, a.plan, [&](int w, u32 k, u32 v) {
#pragma unroll
                for (int j = 0; j < SORT_MAX_IPT; j++) if (j == j0 + w) { key[j] = k; val[j] = v; }
            });
        }
    } else {
#pragma unroll
        for (int j = 0; j < SORT_MAX_IPT; j++) {
            const u64 idx = e0 + (u64)wave * (64u * (u32)ipt) + (u64)j * 64u + (u64)lane;
            const bool in = j < ipt && idx < n_pairs;
            key[j] = in ? a.keys_in[idx] : 0xffffffffu;
            val[j] = in ? a.vals_in[idx] : 0u;
        }
    }

    // ---- rank inside the wave, pair slot by pair slot: lanes with the same digit find each other by ballots over the digit bits
    u32 rank[SORT_MAX_IPT];                                // position of the pair among the wave's pairs of its bin
    u32* mycnt = wcnt + wave * SORT_MAX_BINS;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int j = 0; j < SORT_MAX_IPT; j++) {
        if (j < ipt) {
            const u32 d = (key[j] >> a.shift) & dmask;
            const bool keep = !(FROM_SCALARS && a.drop) || key[j] < a.plan.sentinel;      // (drop mode: zero digits and padding take no part)
            u64 peers = __ballot(keep);
            for (int b = 0; b < a.bins_log; b++) {
                const u64 vote = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? vote : ~vote;
            }
            if (keep) {
                const u32 before = (u32)__popcll(peers & lt_mask), same = (u32)__popcll(peers);
                const u32 old = mycnt[d];                  // every peer reads the counter ...
                if (before == 0) mycnt[d] = old + same;    // ... before its first lane moves it on (one wave, LDS accesses in program order)
                rank[j] = old + before;
            }
        }
    }
    __syncthreads();

    // ---- per bin: the waves' starts inside the bin, the tile's total, and the scan of the totals over the bins
    u32 total = 0;
    if ((u32)tid < bins) {
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) { const u32 cw = wcnt[w * SORT_MAX_BINS + tid]; wcnt[w * SORT_MAX_BINS + tid] = total; total += cw; }
    }
    // exclusive scan of `total` over tid (the first `bins` <= 256 of the tile's 512 lanes hold a count): wave scan by shuffles, wave totals through LDS
    u32 incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
    if (lane == 63) misc[1 + wave] = incl;
    __syncthreads();
    u32 wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += misc[1 + w];
    const u32 my_excl = wave_off + incl - total;
    if ((u32)tid < bins) texcl[tid] = my_excl;

    // ---- decoupled look-back, one lane per bin: pairs of this bin in all the tiles before this one
    if ((u32)tid < bins) {
        u32* mine = a.lookback + (size_t)tile * bins + tid;
        u32 prev = 0;
        if (tile) {
            __hip_atomic_store(mine, LB_AGGREGATE | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // LBW tiles back per turn: their status words are read together (one round trip to the memory side instead of LBW in
            // a row), then taken nearest first; a word that is still empty is waited for
            bool done = false;
            for (u32 t = tile; t > 0 && !done;) {
                u32 v[LBW];
#pragma unroll
                for (int q = 0; q < LBW; q++)
                    v[q] = t > (u32)q ? __hip_atomic_load(a.lookback + (size_t)(t - 1 - q) * bins + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PREFIX;
#pragma unroll
                for (int q = 0; q < LBW; q++) {
                    if (done) break;
                    u32 w = v[q], spins = 0;
                    while (!(w >> 30) && ++spins < SPIN_LIMIT) {
                        __builtin_amdgcn_s_sleep(1);
                        w = __hip_atomic_load(a.lookback + (size_t)(t - 1 - q) * bins + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (!(w >> 30)) { atomicOr(a.error_flag, 1u); done = true; break; }      // never seen: give up (the schedule is flagged), do not hang
                    prev += w & LB_VALUE_MASK;
                    if ((w >> 30) == 2) done = true;
                }
                t = t > (u32)LBW ? t - LBW : 0;
            }
        }
        __hip_atomic_store(mine, LB_PREFIX | (prev + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gbase[tid] = a.bin_base[tid] + prev - my_excl;
    }
    __syncthreads();

    // ---- reorder through LDS: the tile's pairs in bin order ...
#pragma unroll
    for (int j = 0; j < SORT_MAX_IPT; j++) {
        if (j < ipt && (!(FROM_SCALARS && a.drop) || key[j] < a.plan.sentinel)) {
            const u32 d = (key[j] >> a.shift) & dmask;
            const u32 lp = texcl[d] + wcnt[wave * SORT_MAX_BINS + d] + rank[j];
            stage_k[lp] = key[j]; stage_v[lp] = val[j];
        }
    }
    __syncthreads();
    // ... and out: neighbouring lanes hold neighbouring pairs of a bin, i.e. neighbouring addresses
    u32 staged = T;
    if (FROM_SCALARS && a.drop) {                          // drop mode: the pairs the tile kept
        staged = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) staged += misc[1 + w];
    }
    for (u32 lp = tid; lp < staged; lp += SORT_THREADS) {
        const u32 k = stage_k[lp];
        const u32 d = (k >> a.shift) & dmask;
        const u32 pos = gbase[d] + lp;                     // (mod 2^32: gbase holds a difference)
        if (pos < n_moved) {                               // (padding a later pass made for its own last tile sorts behind everything: dropped)
            a.keys_out[pos] = k;
            a.vals_out[pos] = stage_v[lp];
        }
    }
    }      // the next tile
}

template <class T> void sort_alloc(T*& p, size_t bytes) { alloc_epoch_bump(); if (p) hipFree(p); p = nullptr; UG_HIP(hipMalloc(&p, bytes ? bytes : 4)); }

}  // namespace

// pass plan for keys below 2^bits: ceil(bits / 8) passes of nearly equal width, low bits first, the narrower digits first (the
// fused first pass has the smallest tiles: windows pairs per lane)
int radix_plan(int bits, int* shift, int* bins_log) {
    if (bits < 1) bits = 1;
    const int passes = (bits + 7) / 8;
    int at = 0;
    for (int p = 0; p < passes; p++) {
        const int w = (bits - at) / (passes - p);
        shift[p] = at; bins_log[p] = w; at += w;
    }
    return passes;
}

void digit_pairs(const u32* scalars, const DigitPlan& plan, u32* keys, u32* vals, hipStream_t stream) {
    if (!plan.n) return;
    hipLaunchKernelGGL(digit_pairs_kernel, dim3((unsigned)((plan.n + 255) / 256)), dim3(256), 0, stream, scalars, plan, keys, vals);
    UG_KERNEL_CHECK();
}

void RadixSorter::reserve(u64 n_pairs, int ipt_min) {
    // (later passes move the pair count rounded up to the FIRST pass's tile, at most SORT_THREADS * 16 - 1 = 8 191 pairs more, in tiles of their own)
    const u64 tiles = (n_pairs + (u64)SORT_THREADS * SORT_MAX_IPT) / ((u64)SORT_THREADS * (u64)(ipt_min < 1 ? 1 : ipt_min)) + 2;
    if (tiles > tiles_cap) {
        sort_alloc(lookback, (size_t)tiles * SORT_MAX_BINS * 4);
        tiles_cap = tiles;
    }
    if (!small) sort_alloc(small, (4 * SORT_MAX_BINS + 16) * 4);
}
void RadixSorter::release() {
    if (lookback || small) alloc_epoch_bump();
    if (lookback) hipFree(lookback);
    if (small) hipFree(small);
    lookback = small = nullptr; tiles_cap = 0;
}

// Sorts the schedule's pairs by key. scalars != nullptr and windows <= 16: the pairs are made from the scalars inside the
// first pass; otherwise they are read from (buf_keys[0], buf_vals[0]). The passes ping-pong between the two buffer pairs
// (each with room for whole tiles: n_pairs + 8192 entries); returns the index of the pair that holds the result.
// n_valid_out (optional, device): DROP MODE -- when the pairs are made from the scalars, those with key >= sentinel (zero digits;
// the padding of the last tile) are never written: the first pass compacts them away, the later passes move only what is left,
// and the number of pairs that remain is stored at *n_valid_out (before any pass runs). A circom-like witness is mostly zeros
// and small values: more than half of its (scalar, window) pairs are zero digits that the other form carries through every pass
// only to cut them off at the end. *dropped_out tells whether the mode was taken (it needs the scalar form).
int RadixSorter::sort(const u32* scalars, const MsmGeometry& geo, u32 sentinel, u64 n_pairs, int bits,
                      u32* const buf_keys[2], u32* const buf_vals[2], u32* error_flag, hipStream_t stream, u32* n_valid_out, bool* dropped_out) {
    const int windows = geo.windows;
    const DigitPlan plan = geo.digit_plan();
    // measurement switches (-DUG_MEASURE builds): pairs per lane of the pair-form passes, scalars per lane of the fused first pass
#ifndef UG_SORT_IPT_DEFAULT
#define UG_SORT_IPT_DEFAULT 16
#endif
    static const int env_ipt = measure_env("UG_SORT_IPT") ? atoi(measure_env("UG_SORT_IPT")) : UG_SORT_IPT_DEFAULT;
    static const int env_spl = measure_env("UG_SORT_SPL") ? atoi(measure_env("UG_SORT_SPL")) : 1;
    const int ipt_pairs = env_ipt < 4 ? 4 : env_ipt > SORT_MAX_IPT ? SORT_MAX_IPT : env_ipt;
    const bool fused = scalars != nullptr && windows <= SORT_FUSED_MAX_WINDOWS;
    static const bool drop_ok = !(measure_env("UG_SORT_DROP") && atoi(measure_env("UG_SORT_DROP")) == 0);      // A/B switch
    const bool drop = fused && n_valid_out != nullptr && drop_ok;
    if (dropped_out) *dropped_out = drop;
    // drop mode sorts no sentinel: the keys are below it, and one bit less may do (2^21 buckets: 7 | 7 | 7 instead of 7 | 7 | 8 --
    // the last pass with 128 bins instead of 256, i.e. runs twice as long per bin and tile)
    if (drop) { bits = 1; while (((u64)1 << bits) < sentinel) bits++; }
    int shift[4], bins_log[4];
    const int passes = radix_plan(bits, shift, bins_log);
    if (passes > 4) throw std::logic_error("radix sort: key too wide");
    int spl = 1;
    if (fused) { spl = env_spl < 1 ? 1 : env_spl; while (spl > 1 && spl * windows > SORT_MAX_IPT) spl--; }
    const int ipt_first = fused ? windows * spl : ipt_pairs;
    reserve(n_pairs, fused ? std::min(windows * spl, ipt_pairs) : ipt_pairs);
    u32* hist = small;                         // passes x 256
    u32* counters = small + 4 * SORT_MAX_BINS; // one tile counter per pass
    UG_HIP(hipMemsetAsync(small, 0, (4 * SORT_MAX_BINS + 16) * 4, stream));
    // padded pair count: whole tiles of the FIRST pass (later passes see the same pairs: their last tile is padded on the fly)
    const u64 T0 = (u64)SORT_THREADS * (u64)ipt_first;
    const u64 tiles0 = (n_pairs + T0 - 1) / T0;
    const u64 n_padded = fused ? tiles0 * T0 : n_pairs;      // (the pair form pads inside the kernel and does not count the padding ...)
    {
        SortHistArgs h;
        h.keys_in = fused ? nullptr : buf_keys[0];
        h.scalars = fused ? scalars : nullptr; h.plan = plan; h.sentinel = sentinel;
        h.n_pairs = n_pairs; h.n_padded = fused ? n_padded : ((n_pairs + (u64)SORT_THREADS * ipt_pairs - 1) / ((u64)SORT_THREADS * ipt_pairs)) * ((u64)SORT_THREADS * ipt_pairs);
        h.passes = passes;
        for (int p = 0; p < 4; p++) { h.shift[p] = p < passes ? shift[p] : 0; h.bins_log[p] = p < passes ? bins_log[p] : 1; }
        h.hist = hist;
        h.drop = drop ? 1 : 0;
        const u64 lanes = fused ? h.n_padded / (u64)windows : h.n_padded;
        unsigned blocks = (unsigned)std::min<u64>((lanes + SORT_THREADS - 1) / SORT_THREADS, 2048);
        if (!blocks) blocks = 1;
        const int cw = fused ? (plan.c == 22 || plan.c == 20 ? plan.c : 0) : 0;
        if (cw == 22) hipLaunchKernelGGL(radix_hist_kernel<22>, dim3(blocks), dim3(SORT_THREADS), 0, stream, h);
        else if (cw == 20) hipLaunchKernelGGL(radix_hist_kernel<20>, dim3(blocks), dim3(SORT_THREADS), 0, stream, h);
        else hipLaunchKernelGGL(radix_hist_kernel<0>, dim3(blocks), dim3(SORT_THREADS), 0, stream, h);
        UG_KERNEL_CHECK();
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(64), 0, stream, hist, passes, drop ? n_valid_out : (u32*)nullptr);
        UG_KERNEL_CHECK();
    }
    // every pass moves the same multiset of pairs: the real ones plus the padding of the first pass's last tile
    const u64 moved = fused ? n_padded : (n_pairs + (u64)SORT_THREADS * ipt_pairs - 1) / ((u64)SORT_THREADS * ipt_pairs) * ((u64)SORT_THREADS * ipt_pairs);
    int cur = 0;                               // buffer pair that holds the current order (pair form: the input)
    for (int p = 0; p < passes; p++) {
        const bool first_fused = fused && p == 0;
        SortPassArgs a;
        a.keys_in = buf_keys[cur]; a.vals_in = buf_vals[cur];
        a.scalars = first_fused ? scalars : nullptr; a.plan = plan;
        const int dst = first_fused ? 0 : 1 - cur;
        a.keys_out = buf_keys[dst]; a.vals_out = buf_vals[dst];
        a.n_pairs = p == 0 ? n_pairs : moved;                // later passes read the padding of the first one as pairs
        a.n_moved = (u32)moved;
        a.ipt = first_fused ? ipt_first : ipt_pairs;
        a.spl = spl;
        a.shift = shift[p]; a.bins_log = bins_log[p];
        a.bin_base = hist + p * SORT_MAX_BINS;
        a.n_valid = drop ? n_valid_out : nullptr; a.drop = drop ? 1 : 0;
        a.lookback = lookback; a.tile_counter = counters + p; a.error_flag = error_flag;
        const u64 T = (u64)SORT_THREADS * (u64)a.ipt;
        const u64 tiles = (moved + T - 1) / T;
        a.n_tiles = (u32)tiles;
        if (tiles > tiles_cap) throw std::logic_error("radix sort: look-back table too small");
        UG_HIP(hipMemsetAsync(lookback, 0, (size_t)tiles * ((size_t)1 << bins_log[p]) * 4, stream));
        const size_t lds = (2 * T + SORT_WAVES * SORT_MAX_BINS + 2 * SORT_MAX_BINS + 16) * 4;
        const unsigned grid = (unsigned)std::min<u64>(tiles, SORT_MAX_GRID);
        // the fused first pass with the window width as a compile-time constant where the provers' sizes use it
        const int cw = first_fused ? (plan.c == 22 || plan.c == 20 ? plan.c : 0) : 0;
        if (lds > 64 * 1024) {                                 // (tiles of more than 64 KiB of LDS: the attribute is per device)
            static std::atomic<bool> allowed[64];
            int dev = 0;
            UG_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !allowed[dev].load(std::memory_order_acquire)) {
                // what the device really grants a workgroup (gfx950: 160 KiB); a tile that does not fit is refused with the reason
                int cap = 0;
                UG_HIP(hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
                if ((size_t)cap < lds)
                    throw std::runtime_error("radix sort: a tile of " + std::to_string(SORT_THREADS) + " lanes x " + std::to_string(a.ipt) + " pairs needs " +
                                             std::to_string((lds + 1023) / 1024) + " KiB of LDS per workgroup, the device grants " + std::to_string(cap / 1024) + " KiB");
                UG_HIP(hipFuncSetAttribute((const void*)radix_pass_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                UG_HIP(hipFuncSetAttribute((const void*)radix_pass_kernel<true, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                UG_HIP(hipFuncSetAttribute((const void*)radix_pass_kernel<true, 22>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                UG_HIP(hipFuncSetAttribute((const void*)radix_pass_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                if (dev >= 0 && dev < 64) allowed[dev].store(true, std::memory_order_release);
            }
        }
        if (!first_fused) hipLaunchKernelGGL((radix_pass_kernel<false, 0>), dim3(grid), dim3(SORT_THREADS), lds, stream, a);
        else if (cw == 22) hipLaunchKernelGGL((radix_pass_kernel<true, 22>), dim3(grid), dim3(SORT_THREADS), lds, stream, a);
        else if (cw == 20) hipLaunchKernelGGL((radix_pass_kernel<true, 20>), dim3(grid), dim3(SORT_THREADS), lds, stream, a);
        else hipLaunchKernelGGL((radix_pass_kernel<true, 0>), dim3(grid), dim3(SORT_THREADS), lds, stream, a);
        UG_KERNEL_CHECK();
        cur = dst;
    }
    return cur;
}

}  // namespace ug
