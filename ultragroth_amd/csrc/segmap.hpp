// segmap.hpp -- how the sorted entry list of an MSM schedule is cut into the segments of the accumulation kernel
// (msm.hip). Host- and device-compilable (the CPU test-suite checks the map's invariants through libug_hostmath_test.so).
#pragma once
#include "ff.hpp"

namespace ug {

// The sorted entry list is cut into segments, one lane of the accumulation kernel each: 2^log_a entries per segment up to
// entry `split`, 2^log_b (shorter) after it. Workgroups are dispatched in index order, so the short segments are the last
// to start and fill the end of the launch, where CUs that drifted apart over the 8 - 12 rounds of workgroups would
// otherwise wait for the slowest. Measured at 2^24 (A/B on one box, UG_SEG_TAPER): G1 launch 15.20 -> 15.07 ms, G2 40.98 ->
// 40.59, the MSM part of a proof 122.85 -> 122.55 ms (short segments everywhere would cost the fix-up more than that:
// four times as many bucket pieces). split is a whole number of wave tiles (64 long segments), so both kinds of tile stay
// contiguous in the lane-transposed copy. The map is made on the device from the
// number of valid entries (meta[1]), which the host never reads.
struct SegMap {
    u32 split, seg0;          // first entry / first segment of the short kind
    int log_a, log_b;
    UG_HD static SegMap make(u64 n_valid, int log_a, int log_b) {
        SegMap m;
        m.log_a = log_a; m.log_b = log_b;
        const u64 tile = (u64)64 << log_a;
        const u64 split = log_b < log_a ? ((n_valid - (n_valid >> 3)) / tile) * tile : ((n_valid + tile - 1) / tile) * tile;
        m.split = (u32)split; m.seg0 = (u32)(split >> log_a);
        return m;
    }
    UG_HD u32 seg_of(u32 pos) const { return pos < split ? pos >> log_a : seg0 + ((pos - split) >> log_b); }
    UG_HD u32 first_entry(u32 seg) const { return seg < seg0 ? seg << log_a : split + ((seg - seg0) << log_b); }
    UG_HD int log_len(u32 seg) const { return seg < seg0 ? log_a : log_b; }
    // position of entry k of segment seg in the lane-transposed copy (seg0 is a multiple of 64)
    UG_HD u64 transposed(u32 seg, u32 k) const {
        const u64 tile0 = seg < seg0 ? (u64)(seg >> 6) << (log_a + 6) : (u64)split + ((u64)((seg - seg0) >> 6) << (log_b + 6));
        return tile0 + ((u64)k << 6) + (seg & 63);
    }
    // upper bound of the number of segments of any n_valid <= total (what the host sizes grids and slot arrays with)
    static inline u64 max_segments(u64 total, int log_a, int log_b) {
        if (log_b >= log_a) return (total + ((u64)1 << log_a) - 1) >> log_a;
        return (total >> log_a) + (((total >> 3) + ((u64)64 << log_a)) >> log_b) + 64;
    }
};

}  // namespace ug
