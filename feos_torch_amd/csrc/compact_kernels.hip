// K8 — device-side stream compaction of the solver outputs.
//
// The reference drops the rows its solver fails on inside the native call (`filter_map` + `status`,
// src/pcsaft.rs:93-101; `filter_binary`, :216-231) and then filters the model with the same mask
// (feos_torch/pcsaft_pure.py:235-243, pcsaft_mix.py:470-479, gc_pcsaft.py:514-528).  The solver kernels here write
// DENSE outputs + a status byte per row (static shapes: hipGraph capture, all-gather with fixed message sizes), so the
// drop is a separate, order-preserving compaction:
//
//   pcs_compact_plan   status[n] -> per-block counts of kept rows, their exclusive scan and the total (ONE int32 the
//                      caller reads back: the only host synchronisation of a property call);
//   pcs_compact_rows   dst[j, :] = src[i_j, :] for the j-th kept row (values, densities, parameter rows, gc row
//                      encodings viewed as doubles) and, optionally, the row index map j -> i_j;
//   pcs_expand_rows    the inverse for the backward pass, fused with the Jacobian product:
//                      dst[i, c] = g[j] * src[j, col0 + c] for kept rows, 0 for dropped ones.
//
// HBM-bound byte/word shuffling: one status byte per lane and load, ballot + popcount for the in-wave rank, LDS only for
// the four wave totals of a 256-row tile; a workgroup owns 2048 consecutive rows so the scan over workgroups is short
// (4,883 entries at 1e7 rows) and done by a single 1024-thread workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"

using namespace pcs_abi;

namespace {

constexpr int CTHREADS = 256;
constexpr int CTILES = 8;                       // 256-row tiles per workgroup
constexpr int CROWS = CTHREADS * CTILES;        // rows per workgroup
constexpr int CWS_HEAD = 2;                     // cws[0] = kept rows, cws[1] = number of workgroups, then the offsets

__device__ __forceinline__ unsigned long long lanes_below() {
    const int lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// kept rows of this workgroup's 2048-row slab
__global__ __launch_bounds__(CTHREADS) void k_compact_count(const uint8_t* __restrict__ status, int64_t n,
                                                            int32_t* __restrict__ cws) {
    __shared__ int wave_cnt[CTHREADS / 64];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * CROWS;
    int cnt = 0;  // lane 0 of each wave accumulates its wave's total
#pragma unroll
    for (int j = 0; j < CTILES; j++) {
        const int64_t i = row0 + j * CTHREADS + t;
        const bool keep = i < n && status[i] == 0;
        cnt += __popcll(__ballot(keep));
    }
    if ((t & 63) == 0) wave_cnt[t >> 6] = cnt;
    __syncthreads();
    if (t == 0) cws[CWS_HEAD + blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// exclusive scan of the per-workgroup counts (in place) + total; one workgroup
__global__ __launch_bounds__(1024) void k_compact_scan(int32_t* __restrict__ cws, int nb) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (nb + 1023) / 1024;
    const int lo = t * chunk, hi = min(lo + chunk, nb);
    int s = 0;
    for (int k = lo; k < hi; k++) s += cws[CWS_HEAD + k];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;  // exclusive prefix of this thread's chunk
    for (int k = lo; k < hi; k++) {
        const int c = cws[CWS_HEAD + k];
        cws[CWS_HEAD + k] = run;
        run += c;
    }
    if (t == 1023) {
        cws[0] = part[1023];
        cws[1] = nb;
    }
}

// rank of every kept row of the workgroup's slab, handed to `body(i, dest, keep)` tile by tile
template <class F>
__device__ __forceinline__ void for_each_row(const uint8_t* __restrict__ status, int64_t n, const int32_t* __restrict__ cws,
                                             F body) {
    __shared__ int wave_cnt[CTHREADS / 64];
    const int t = threadIdx.x, w = t >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * CROWS;
    int64_t base = status ? (int64_t)cws[CWS_HEAD + blockIdx.x] : row0;
    const unsigned long long below = lanes_below();
    for (int j = 0; j < CTILES; j++) {
        const int64_t i = row0 + j * CTHREADS + t;
        if (row0 + j * CTHREADS >= n) break;  // workgroup-uniform
        if (!status) {  // every row kept: the identity map
            if (i < n) body(i, i, true);
            continue;
        }
        const bool keep = i < n && status[i] == 0;
        const unsigned long long b = __ballot(keep);
        if ((t & 63) == 0) wave_cnt[w] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < CTHREADS / 64; k++) {
            const int c = wave_cnt[k];
            before += k < w ? c : 0;
            total += c;
        }
        if (i < n) body(i, base + before + __popcll(b & below), keep);
        base += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(CTHREADS) void k_compact_rows(const uint8_t* __restrict__ status, int64_t n,
                                                           const int32_t* __restrict__ cws, const double* __restrict__ src,
                                                           int width, double* __restrict__ dst, int32_t* __restrict__ index,
                                                           int vec16) {
    for_each_row(status, n, cws, [&](int64_t i, int64_t j, bool keep) {
        if (!keep) return;
        if (src) {
            const double* s = src + i * width;
            double* d = dst + j * width;
            if (vec16) {  // 16-byte moves: even number of doubles per row and 16-byte aligned bases (checked on the host)
                const double2* s2 = reinterpret_cast<const double2*>(s);
                double2* d2 = reinterpret_cast<double2*>(d);
                for (int k = 0; k < width / 2; k++) d2[k] = s2[k];
            } else {
                for (int k = 0; k < width; k++) d[k] = s[k];
            }
        }
        if (index) index[j] = (int32_t)i;
    });
}

__global__ __launch_bounds__(CTHREADS) void k_expand_rows(const uint8_t* __restrict__ status, int64_t n,
                                                          const int32_t* __restrict__ cws, const double* __restrict__ g,
                                                          const double* __restrict__ src, int src_stride, int col0, int ncol,
                                                          double* __restrict__ dst) {
    for_each_row(status, n, cws, [&](int64_t i, int64_t j, bool keep) {
        double* d = dst + i * ncol;
        if (!keep) {
            for (int k = 0; k < ncol; k++) d[k] = 0.0;
            return;
        }
        const double scale = g ? g[j] : 1.0;
        const double* s = src + j * src_stride + col0;
        for (int k = 0; k < ncol; k++) d[k] = scale * s[k];
    });
}

int compact_blocks(int64_t n) { return (int)((n + CROWS - 1) / CROWS); }

}  // namespace

extern "C" {

int64_t pcs_compact_workspace_bytes(int64_t n) { return (int64_t)sizeof(int32_t) * (CWS_HEAD + (n > 0 ? compact_blocks(n) : 0) + 2); }

int pcs_compact_plan(const uint8_t* status, int64_t n, void* cws, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (!cws || (n > 0 && !status)) return fail_msg("pcs_compact_plan: null required pointer");
    hipStream_t s = as_stream(stream);
    int32_t* w = static_cast<int32_t*>(cws);
    const int nb = n > 0 ? compact_blocks(n) : 0;
    if (nb) hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(CTHREADS), 0, s, status, n, w);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, s, w, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_compact_scan launch", e);
    return 0;
}

int pcs_compact_rows(const uint8_t* status, int64_t n, const void* cws, const double* src, int width, double* dst,
                     int32_t* index, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!status || !cws) return fail_msg("pcs_compact_rows: null required pointer");
    if ((src == nullptr) != (dst == nullptr)) return fail_msg("pcs_compact_rows: src and dst come together");
    if (!src && !index) return fail_msg("pcs_compact_rows: nothing to do");
    if (src && (width < 1 || width > 64)) return fail_msg("pcs_compact_rows: width must be 1..64 doubles");
    hipLaunchKernelGGL(k_compact_rows, dim3(compact_blocks(n)), dim3(CTHREADS), 0, as_stream(stream), status, n,
                       static_cast<const int32_t*>(cws), src, width, dst, index,
                       (int)(src && (width & 1) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_compact_rows launch", e);
    return 0;
}

int pcs_expand_rows(const uint8_t* status, int64_t n, const void* cws, const double* g, const double* src, int src_stride,
                    int col0, int ncol, double* dst, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!src || !dst || (status && !cws)) return fail_msg("pcs_expand_rows: null required pointer");
    if (ncol < 1 || col0 < 0 || col0 + ncol > src_stride || src_stride > 64) return fail_msg("pcs_expand_rows: bad column range");
    hipLaunchKernelGGL(k_expand_rows, dim3(compact_blocks(n)), dim3(CTHREADS), 0, as_stream(stream), status, n,
                       static_cast<const int32_t*>(cws), g, src, src_stride, col0, ncol, dst);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_expand_rows launch", e);
    return 0;
}

}  // extern "C"
