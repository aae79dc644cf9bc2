// dev_common.hpp -- shared device-side helpers: error handling, packed 32-byte element I/O.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include "ec.hpp"

namespace ug {

struct HipError : public std::runtime_error {
    explicit HipError(const std::string& m) : std::runtime_error(m) {}
};

#define UG_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw ug::HipError(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr + \
                               " (" __FILE__ ":" + std::to_string(__LINE__) + ")");                \
    } while (0)

#define UG_KERNEL_CHECK() UG_HIP(hipGetLastError())

// An event recorded INSIDE a stream capture so that every launch of the resulting graph records it again (timed spans of a
// captured launch sequence): an explicit event-record node behind the stream's current capture dependencies, which then becomes
// the stream's dependency. (hipEventRecordWithFlags(.., hipEventRecordExternal) is the same thing in one call on ROCm 7.2, but
// the runtime a PyTorch wheel bundles -- ROCm 7.0, the one a Python process ends up with -- refuses it: tools/probe_graph_events.hip.)
inline void record_in_capture(hipEvent_t ev, hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t g = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t n = 0;
    UG_HIP(hipStreamGetCaptureInfo_v2(stream, &st, &id, &g, &deps, &n));
    if (st != hipStreamCaptureStatusActive || !g) throw HipError("HIP error: the stream is not being captured (record_in_capture)");
    hipGraphNode_t node = nullptr;
    UG_HIP(hipGraphAddEventRecordNode(&node, g, deps, n, ev));
    UG_HIP(hipStreamUpdateCaptureDependencies(stream, &node, 1, hipStreamSetCaptureDependencies));
}

// Environment switches come in two kinds (the table in include/ultragroth_hip.h lists both):
//   tuning knobs        getenv(): every setting gives correct results, the default is the measured best
//   measurement switches  measure_env(): A/B switches of finished experiments and settings that give WRONG results (folded
//                       gathers). They exist only in a -DUG_MEASURE build (`make MEASURE=1` -> libultragroth_hip_measure.so);
//                       the product library reads none of them.
#ifdef UG_MEASURE
inline const char* measure_env(const char* name) { return getenv(name); }
#else
inline const char* measure_env(const char*) { return nullptr; }
#endif

// ---- 32-byte packed elements in HBM: two 16-byte vector accesses -----------------------------------
__device__ __forceinline__ void load8(u32* w, const u32* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store8(u32* p, const u32* w) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// element stored as 32 bytes in device Montgomery form (value < 2^256, any representative)
template <class P> __device__ __forceinline__ Fp<P> ld_packed(const u32* p) {
    u32 w[8];
    load8(w, p);
    return unpack256<P>(w);
}
// a must be strict and < 2^256
template <class P> __device__ __forceinline__ void st_packed(u32* p, const Fp<P>& a) {
    u32 w[8];
    pack256(w, a);
    store8(p, w);
}

// Cheap range contraction: any weak a < 64 q  ->  strict, same residue, < 2.01 q.
// Quotient estimate from the top limb (bits 232..260) in fp32, chosen to never exceed the true
// quotient; 9 v_mad_u64_u32 for e*q, then one signed serial carry pass.
template <class P> __device__ __forceinline__ Fp<P> contract(const Fp<P>& a) {
    const float rcp = 1.0f / (float)(P::q[NL - 1] + 2);
    u32 e = (u32)((float)a.l[NL - 1] * rcp * 0.99999f);
    Fp<P> r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int64_t v = (int64_t)a.l[i] - (int64_t)((u64)e * P::q[i]) + carry;
        r.l[i] = (u32)v & MASK29;
        carry = v >> LB;
    }
    r.l[NL - 1] |= (u32)carry << LB;   // carry is 0 for in-range inputs; keeps the value if not
    return r;
}

__host__ __device__ __forceinline__ u32 bit_reverse(u32 x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}

}  // namespace ug
