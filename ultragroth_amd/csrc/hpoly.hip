// hpoly.hip -- the Fr-side kernels of the H-polynomial block around the NTTs.
//
// Replaces these loops of Groth16::Prover::prove (src/groth16.cpp):
//   :77-99   a = A.w, b = B.w   sparse scatter-add under 1024 striped mutexes
//   :100-108 c = a o b
//   :142-148 h = a o b - c, fromMontgomery
// The reference's scatter with locks becomes a gather: the coefficient records are sorted by
// (matrix, row) once at create time (the radix partition of sort.hip, on the device) into CSR form, and one lane
// sums one row -- deterministic and atomic-free. Rows are written at their bit-reversed position so
// that the first NTT pass reads contiguously.
#include "dev_common.hpp"
#include "internal.hpp"

namespace ug {

namespace {

__global__ void coef_keys_kernel(const uint8_t* raw, u64 ncoefs, u32 domain, u32 nvars, u32* keys, u32* idx, u32* bad) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncoefs) return;
    const u32* rec = reinterpret_cast<const u32*>(raw + i * 44);
    u32 m = rec[0], c = rec[1], s = rec[2];
    if (m > 1 || c >= domain || s >= nvars) { atomicOr(bad, 1u); m = 0; c = 0; }
    keys[i] = m * domain + c;
    idx[i] = (u32)i;
}
__global__ void coef_gather_kernel(const uint8_t* raw, u64 ncoefs, const u32* idx, u32* sig, u32* val) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ncoefs) return;
    const u32* rec = reinterpret_cast<const u32*>(raw + (u64)idx[p] * 44);
    sig[p] = rec[2];
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = rec[3 + k];
    // stored value is coef * 2^512; bring it to coef * 2^522 so that
    // mul(w_plain, val) = w * coef * 2^261
    Fr v = cond_sub_q(mul(unpack256<FrParams>(w), fp_from<FrParams>(FrParams::coef512)));
    st_packed(val + p * 8, v);
}
__global__ void row_ptr_kernel(const u32* keys, u64 ncoefs, u32 nrows, u32* row_ptr) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    u64 lo = 0, hi = ncoefs;                     // first position with key >= r
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (keys[mid] < r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = (u32)lo;
}

// the sum of one row of the coefficient matrix times the witness (rows r < domain: matrix A, the others: matrix B)
__device__ __forceinline__ Fr matvec_row(u32 r, const u32* row_ptr, const u32* sig, const u32* val, const u32* wtns) {
    u32 s = row_ptr[r], e = row_ptr[r + 1];
    Fr acc = fp_zero<FrParams>();
    u32 since = 0;
    // four entries at a time: their signal ids first, then the four 32-byte witness gathers and the four coefficients all in
    // flight together, then the products (one entry per turn left every gather's latency -- a DRAM row miss -- exposed: 69 %
    // of the kernel's wave cycles were waits)
    for (u32 p = s; p < e; p += 4) {
        const u32 cnt = e - p < 4 ? e - p : 4;
        u32 sg[4], wr[4][8], vr[4][8];
#pragma unroll
        for (int k = 0; k < 4; k++) sg[k] = (u32)k < cnt ? sig[p + k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((u32)k < cnt) { load8(wr[k], wtns + (size_t)sg[k] * 8); load8(vr[k], val + (size_t)(p + k) * 8); }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((u32)k < cnt) {
                acc = add(acc, mul(unpack256<FrParams>(wr[k]), unpack256<FrParams>(vr[k])));      // + < 2q  (w: plain integer, any value < 2^256)
                if (++since == 24) { acc = contract(acc); since = 0; }                              // keep below 64 q
            }
        }
    }
    return contract(acc);
}

// one lane per row, the result stored at the row's bit-reversed place (domains below 2^8)
__global__ __launch_bounds__(256) void matvec_kernel(u32* a_br, u32* b_br, const u32* row_ptr, const u32* sig,
                                                     const u32* val, const u32* wtns, u32 domain, int logn, int mask) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2 * domain) return;
    if (!((mask >> (r >= domain ? 1 : 0)) & 1)) return;        // bit 0: rows of A, bit 1: rows of B
    Fr acc = matvec_row(r, row_ptr, sig, val, wtns);
    u32 c = r >= domain ? r - domain : r;
    u32* dst = (r >= domain ? b_br : a_br) + (size_t)bit_reverse(c, logn) * 8;
    st_packed(dst, acc);
}

// The same with the bit reversal done in 16 x 16 tiles through LDS (round 3): the rows go to their bit-reversed places because
// the NTT that follows is decimation-in-time, and a lane that stores its own row's result writes 32 bytes 2^(logn-1) elements
// away from its neighbour's -- at 2^24, 33 M scattered 32-byte writes, as many partial lines as the witness gathers read.
// A workgroup takes the rows c = (i << (logn-4)) | (mid << 4) | j, i, j < 16: lane (i, j) reads row c (the 16 rows of a run are
// adjacent: 1 KB of coefficients), the results meet in LDS, and lane (j', l) stores the result of row (i = rev4(l), j') at
// rev(c) = (rev4(j') << (logn-4)) | (rev(mid) << 4) | l -- sixteen adjacent lanes, sixteen adjacent places: 512-byte runs.
__global__ __launch_bounds__(256) void matvec_tiled_kernel(u32* a_br, u32* b_br, const u32* row_ptr, const u32* sig,
                                                           const u32* val, const u32* wtns, u32 domain, int logn, int mask) {
    __shared__ u32 tile[16][16 * 8 + 4];                       // row stride padded: 132 words
    const u32 tiles_per_matrix = domain >> 8;
    const u32 m = blockIdx.x >= tiles_per_matrix ? 1u : 0u;    // which matrix
    if (!((mask >> m) & 1)) return;
    const u32 mid = blockIdx.x - m * tiles_per_matrix;
    const u32 i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const u32 c = (i << (logn - 4)) | (mid << 4) | j;
    const Fr acc = matvec_row(m * domain + c, row_ptr, sig, val, wtns);
    u32 w[8];
    pack256(w, acc);
#pragma unroll
    for (int k = 0; k < 8; k++) tile[i][j * 8 + k] = w[k];
    __syncthreads();
    const u32 j2 = threadIdx.x >> 4, l = threadIdx.x & 15, i2 = bit_reverse(l, 4);
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = tile[i2][j2 * 8 + k];
    const u32 pos = (bit_reverse(j2, 4) << (logn - 4)) | (bit_reverse(mid, logn - 8) << 4) | l;
    store8((m ? b_br : a_br) + (size_t)pos * 8, w);
}

__global__ void mul_pointwise_kernel(u32* out, const u32* x, const u32* y, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st_packed(out + i * 8, mul(ld_packed<FrParams>(x + i * 8), ld_packed<FrParams>(y + i * 8)));
}
__global__ void h_final_kernel(u32* h, const u32* a, const u32* b, const u32* c, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr t = mul(ld_packed<FrParams>(a + i * 8), ld_packed<FrParams>(b + i * 8));    // < 2q
    Fr u = sub<6>(t, ld_packed<FrParams>(c + i * 8));                               // c < 2^256 < 5.3 q
    u32 w[8];
    to_normal(w, u);
    store8(h + i * 8, w);
}
__global__ void from_mont256_kernel(u32* out, const u32* in, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    load8(w, in + i * 8);
    st_packed(out + i * 8, from_mont256<FrParams>(w));
}
__global__ void to_mont256_kernel(u32* out, const u32* in, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    to_mont256(w, ld_packed<FrParams>(in + i * 8));
    store8(out + i * 8, w);
}
template <class P>
__global__ void f_op_mont256_kernel(int op, u32* out, const u32* a, const u32* b, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 wa[8], wb[8], wo[8];
    load8(wa, a + i * 8); load8(wb, b + i * 8);
    Fp<P> x = from_mont256<P>(wa), y = from_mont256<P>(wb), r;
    switch (op) {
        case 0: r = mul(x, y); break;
        case 1: r = add(x, y); break;
        case 2: r = sub<2>(x, y); break;
        default: r = sqr(x); break;
    }
    to_mont256(wo, r);
    store8(out + i * 8, wo);
}

__global__ void gather_kernel(u32* out, const u32* src, const u32* index, u64 n, u64 src_n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 j = index[i];
    if (j < src_n) load8(w, src + j * 8);
    store8(out + i * 8, w);
}

__global__ void scatter_kernel(u32* dst, const u32* index, const u32* values, u64 n, u64 dst_n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 j = index[i];
    if (j >= dst_n) return;
    u32 w[8];
    load8(w, values + i * 8);
    store8(dst + j * 8, w);
}

// ---- UltraGroth lookup completion (src/ultra_groth.cpp:62-106) --------------------------------------------
// The reference runs  for i in order: wtns[w_idx[i]] = push[p_idx[i]]  on the host, with
// push = [rand | inv2[chunks[j]] for every chunk j | inv2[0..L) | prod[0..L)]. Here push is never materialised:
// `table` = [rand | inv2 | prod] (1 + 2L elements) and the chunk indirection is resolved per write. Writes to the
// same index keep the LAST one: pass 1 records the highest i per target, pass 2 lets only that i write, pass 3
// clears the scratch.
__global__ void lookup_mark_kernel(u32* last, const u32* w_idx, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(&last[w_idx[i]], (u32)i + 1);
}
__global__ void lookup_write_kernel(u32* dst, const u32* last, const u32* w_idx, const u32* p_idx, u64 n, const u32* chunks,
                                    u64 n_chunks, const u32* table) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 j = w_idx[i];
    if (last[j] != (u32)i + 1) return;
    const u64 p = p_idx[i];
    u64 t = p == 0 ? 0 : p <= n_chunks ? 1 + (u64)chunks[p - 1] : p - n_chunks;      // element of `table`
    u32 w[8];
    load8(w, table + t * 8);
    store8(dst + (u64)j * 8, w);
}
__global__ void lookup_clear_kernel(u32* last, const u32* w_idx, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) last[w_idx[i]] = 0;
}

// The lookup table itself (src/ultra_groth.cpp:71-80): row i -> inv2[i] = 1 / (i + rand), prod[i] = frequencies[i] * inv2[i],
// both as plain integers. One lane per row with its own Fermat inversion (2^16 rows: 25 M products, microseconds here;
// the reference makes one GMP inversion per row on one core). `i` and the uint32 frequency enter through RawFr::set(int)
// (build/fr.hpp:249-251, build/fr.cpp:209-223): values >= 2^31 mean value - 2^32.
__device__ __forceinline__ Fr fr_set_int(u32 v) {
    const bool negative = (v >> 31) != 0;
    u32 w[8] = {negative ? (u32)(0u - v) : v, 0, 0, 0, 0, 0, 0, 0};
    Fr m = from_normal<FrParams>(w);                           // < 2q, strict
    return negative ? neg<2>(m) : m;
}
__global__ void lookup_table_kernel(u32* table, const u32* freq, u64 L) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    u32 rw[8];
    load8(rw, table);                                          // table[0] = rand, plain integer
    Fr rand = from_normal<FrParams>(rw);
    Fr sum = cond_sub_q(mul(add(fr_set_int((u32)i), rand), fp_one<FrParams>()));
    Fr inv_i = limbs_all_zero(sum) ? fp_zero<FrParams>() : inv(sum);      // mpz_invert leaves 0 for 0
    u32 o[8];
    to_normal(o, inv_i);
    store8(table + (1 + i) * 8, o);
    to_normal(o, mul(fr_set_int(freq[i]), inv_i));
    store8(table + (1 + L + i) * 8, o);
}

template <class T> void dev_alloc(T*& p, size_t bytes) { if (p) hipFree(p); p = nullptr; UG_HIP(hipMalloc(&p, bytes ? bytes : 4)); }
inline unsigned grid_for(u64 n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

bool CoefMatrix::build(const uint8_t* raw44_dev, u64 ncoefs_, u32 domain_, u32 nvars, hipStream_t stream) {
    release();
    ncoefs = ncoefs_; domain = domain_;
    logn = 0;
    while ((1u << logn) < domain) logn++;
    if (ncoefs >= ((u64)1 << 31)) throw std::invalid_argument("coefficient count exceeds 2^31");
    u32 nrows = 2 * domain;
    u32 *keys_a = nullptr, *keys_b = nullptr, *idx_a = nullptr, *idx_b = nullptr, *bad = nullptr;
    const u64 padded = ncoefs + 8192;                  // the sort moves whole tiles
    dev_alloc(keys_a, padded * 4); dev_alloc(keys_b, padded * 4);
    dev_alloc(idx_a, padded * 4); dev_alloc(idx_b, padded * 4);
    u32* flag = nullptr;
    dev_alloc(flag, 4);
    RadixSorter sorter;
    dev_alloc(bad, 4);
    dev_alloc(row_ptr, ((size_t)nrows + 1) * 4);
    dev_alloc(sig, ncoefs * 4);
    dev_alloc(val, ncoefs * 32);
    UG_HIP(hipMemsetAsync(bad, 0, 4, stream));
    u32 bad_host = 0, sort_failed = 0;
    if (ncoefs) {
        hipLaunchKernelGGL(coef_keys_kernel, dim3(grid_for(ncoefs, 256)), dim3(256), 0, stream, raw44_dev, ncoefs, domain, nvars, keys_a, idx_a, bad);
        UG_KERNEL_CHECK();
        int end_bit = 1;
        while (((u64)1 << end_bit) < nrows) end_bit++;
        UG_HIP(hipMemsetAsync(flag, 0, 4, stream));
        u32* const bk[2] = {keys_a, keys_b};
        u32* const bv[2] = {idx_a, idx_b};
        const int at = sorter.sort(nullptr, MsmGeometry(), 0, ncoefs, end_bit, bk, bv, flag, stream);      // pair form: (row key, record index)
        hipLaunchKernelGGL(coef_gather_kernel, dim3(grid_for(ncoefs, 256)), dim3(256), 0, stream, raw44_dev, ncoefs, bv[at], sig, val);
        UG_KERNEL_CHECK();
        hipLaunchKernelGGL(row_ptr_kernel, dim3(grid_for((u64)nrows + 1, 256)), dim3(256), 0, stream, bk[at], ncoefs, nrows, row_ptr);
        UG_KERNEL_CHECK();
        UG_HIP(hipMemcpyAsync(&bad_host, bad, 4, hipMemcpyDeviceToHost, stream));
        UG_HIP(hipMemcpyAsync(&sort_failed, flag, 4, hipMemcpyDeviceToHost, stream));
    } else {
        UG_HIP(hipMemsetAsync(row_ptr, 0, ((size_t)nrows + 1) * 4, stream));
    }
    UG_HIP(hipStreamSynchronize(stream));
    hipFree(keys_a); hipFree(keys_b); hipFree(idx_a); hipFree(idx_b); hipFree(bad); hipFree(flag);
    sorter.release();
    if (sort_failed) throw std::runtime_error("coefficient matrix: the sort gave up waiting for a tile (look-back timeout)");
    return bad_host == 0;
}

void CoefMatrix::release() {
    if (row_ptr) hipFree(row_ptr);
    if (sig) hipFree(sig);
    if (val) hipFree(val);
    row_ptr = sig = val = nullptr;
}

void coef_matvec(u32* a_br, u32* b_br, const CoefMatrix& m, const u32* wtns_dev, int mask, hipStream_t stream) {
    static const bool tiled = !(measure_env("UG_MATVEC_TILED") && atoi(measure_env("UG_MATVEC_TILED")) == 0);      // A/B switch (-DUG_MEASURE)
    if (m.logn >= 8 && tiled)
        hipLaunchKernelGGL(matvec_tiled_kernel, dim3(2 * (m.domain >> 8)), dim3(256), 0, stream,
                           a_br, b_br, m.row_ptr, m.sig, m.val, wtns_dev, m.domain, m.logn, mask);
    else
        hipLaunchKernelGGL(matvec_kernel, dim3(grid_for((u64)2 * m.domain, 256)), dim3(256), 0, stream,
                           a_br, b_br, m.row_ptr, m.sig, m.val, wtns_dev, m.domain, m.logn, mask);
    UG_KERNEL_CHECK();
}
void fr_mul_pointwise(u32* out, const u32* x, const u32* y, u64 n, hipStream_t stream) {
    hipLaunchKernelGGL(mul_pointwise_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, out, x, y, n);
    UG_KERNEL_CHECK();
}
void fr_h_final(u32* h, const u32* a, const u32* b, const u32* c, u64 n, hipStream_t stream) {
    hipLaunchKernelGGL(h_final_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, h, a, b, c, n);
    UG_KERNEL_CHECK();
}
void fr_from_mont256(u32* out, const u32* in, u64 n, hipStream_t stream) {
    hipLaunchKernelGGL(from_mont256_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, out, in, n);
    UG_KERNEL_CHECK();
}
void fr_to_mont256(u32* out, const u32* in, u64 n, hipStream_t stream) {
    hipLaunchKernelGGL(to_mont256_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, out, in, n);
    UG_KERNEL_CHECK();
}
void gather_elements(u32* out, const u32* src, const u32* index_dev, u64 n, u64 src_n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, out, src, index_dev, n, src_n);
    UG_KERNEL_CHECK();
}
void scatter_elements(u32* dst, const u32* index_dev, const u32* values_dev, u64 n, u64 dst_n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, dst, index_dev, values_dev, n, dst_n);
    UG_KERNEL_CHECK();
}
void apply_lookup(u32* dst, u32* last_scratch, const u32* w_idx, const u32* p_idx, u64 n, const u32* chunks, u64 n_chunks,
                  const u32* table, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(lookup_mark_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, last_scratch, w_idx, n);
    UG_KERNEL_CHECK();
    hipLaunchKernelGGL(lookup_write_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, dst, last_scratch, w_idx, p_idx, n, chunks, n_chunks, table);
    UG_KERNEL_CHECK();
    hipLaunchKernelGGL(lookup_clear_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, last_scratch, w_idx, n);
    UG_KERNEL_CHECK();
}
void lookup_table(u32* table_dev, const u32* freq_dev, u64 L, hipStream_t stream) {
    if (!L) return;
    hipLaunchKernelGGL(lookup_table_kernel, dim3(grid_for(L, 128)), dim3(128), 0, stream, table_dev, freq_dev, L);
    UG_KERNEL_CHECK();
}
void f_op_mont256(int which, int op, u32* out, const u32* a, const u32* b, u64 n, hipStream_t stream) {
    if (which == 0) hipLaunchKernelGGL(f_op_mont256_kernel<FrParams>, dim3(grid_for(n, 256)), dim3(256), 0, stream, op, out, a, b, n);
    else hipLaunchKernelGGL(f_op_mont256_kernel<FqParams>, dim3(grid_for(n, 256)), dim3(256), 0, stream, op, out, a, b, n);
    UG_KERNEL_CHECK();
}

}  // namespace ug
