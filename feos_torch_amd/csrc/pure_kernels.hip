// gfx950 kernels + C ABI for the pure-component PC-SAFT path (see include/pcsaft_hip.h).
//
// Launch shape: one state point per lane, 256-thread workgroups (4 waves), grid = ceil(n/256)
// (>> 256 workgroups at the benchmark sizes, so all 8 XCDs fill; rows are independent, so the
// round-robin workgroup->XCD placement needs no remapping — there is no shared tile to keep
// in one L2).  HBM traffic is 81 B per state point against ~1e4 fp64 operations: the kernels
// are fp64-VALU bound; memory handling only has to be coalesced, which it is:
//   * the [n,8] AoS parameter rows of a workgroup (16 KB contiguous) are fetched with 16-byte
//     per-lane loads and staged through LDS (72-byte padded rows, conflict-free ds_read_b64);
//   * T, p and all outputs are SoA and accessed lane-contiguously.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"
#include "pure_solver.hpp"
#include "pure_jacobian.hpp"

// This file is compiled TWICE (feos_torch_amd/build.py), with the same relaxed floating-point flags but
//   PCS_PURE_PART = 1  with -fassociative-math -freciprocal-math: the pressure-only VLE kernel (0.783 -> 0.750 ms per 1e7 rows,
//                      x1.044), the Jacobian kernels (x1.13), fallback, derivatives, their VJP, and the whole C ABI;
//   PCS_PURE_PART = 2  without: the all-fp64 VLE kernel and the liquid-density kernel, which re-association makes SLOWER
//                      (x0.96; 0.895 -> 1.19 ms: the re-associated expressions cost k_pure_liquid_density its register budget),
//                      behind two launch functions that part 1 calls.
// Measured in one process with scripts/dev/ab_bench.py / ab_k2.py / ab_purejac.py (round 3).  A build-level switch like
// PCS_FAST_RCP, not an experiment: both parts are always built and linked.
#ifndef PCS_PURE_PART
#define PCS_PURE_PART 1
#endif

// stage 2 (k_pure_vle_robust) lives in pure_robust.hip, compiled with strict IEEE semantics
namespace pcs_abi {
int launch_pure_vle_retry(const double* params, const double* temp, double* p_sat, double* rho_eq, double* rho_vl,
                          uint8_t* status, int32_t* iters, const int32_t* retry, int64_t n, hipStream_t s);
// part 2 of this file
int launch_pure_vle_full(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq, double* rho_vl,
                         uint8_t* status, int32_t* iters, int32_t* retry, hipStream_t s);
int launch_pure_liquid_density(const double* params, const double* temp, const double* pressure, int64_t n, double* rho_out,
                               double* rho_root, uint8_t* status, hipStream_t s);
}
#define launch_vle_retry pcs_abi::launch_pure_vle_retry

using namespace pcs;
using namespace pcs_abi;

#if PCS_PURE_PART == 1
thread_local char pcs_abi::g_err[256] = "";
#endif

namespace {

constexpr int PCS_BLOCK = 256;
constexpr int BLOCK = PCS_BLOCK;
constexpr int ROW_PAD = 9;  // doubles per staged row (8 + 1 pad): bank-conflict-free per-lane reads

// Cooperative, coalesced load of the workgroup's parameter rows into LDS, then one row per lane.
// Rows past n are clamped to row n-1 (their results are never stored).
__device__ __forceinline__ void stage_params(const double* __restrict__ params, int64_t n, int64_t row0,
                                             double* lds, double par[8]) {
    const int t = threadIdx.x;
    const double2* src = reinterpret_cast<const double2*>(params);
    const int64_t last2 = n * 4 - 1;  // index of the last double2
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int idx2 = t + k * BLOCK;  // double2 index inside the block tile: row = idx2/4, col2 = idx2%4
        int64_t g = row0 * 4 + idx2;
        if (g > last2) g = last2 - 3 + (idx2 & 3);  // clamp to the same columns of row n-1
        double2 v = src[g];
        int r = idx2 >> 2, c2 = idx2 & 3;
        lds[r * ROW_PAD + 2 * c2] = v.x;
        lds[r * ROW_PAD + 2 * c2 + 1] = v.y;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) par[k] = lds[t * ROW_PAD + k];
}

// ------------------------------------------------------------------------------------------
// K1: pure VLE, fast path.  Rows that need the robust initialisation are appended to
// retry[1..]; retry[0] is the running count.
// ------------------------------------------------------------------------------------------
// waves per SIMD asked of the compiler for the all-fp64 instantiation (p_sat + densities): 210 VGPRs fit two; held to 128
// (152 spill instructions) four are resident and the kernel is 1.20 -> 0.96 ms per 1e7 rows (three: 1.00 ms)
constexpr int K1_WAVES = 4;
constexpr int K1_WAVES_LITE = 4;  // pressure-only instantiation: 122 VGPR, 4 waves per SIMD (without -fno-slp-vectorize: 168 VGPR, 3 waves)
constexpr int K1_BINS = 4;
// (Round 2 measured a block-level straggler exchange of the fp32 coupled iteration -- lanes not converged after two iterations
// parked in LDS and finished together by the first wave: 0.815 ms against 0.797 ms without it at 256 threads per workgroup,
// 0.834 ms at 512, 1.014 ms at 1024; the two barriers and the serial straggler wave cost more than the 1.4 iterations per
// wave they save.  Removed.)

// Bucket key of a staged row: model class (which branches of the Helmholtz energy the row needs) in
// Gray order none, polar, polar+assoc, assoc -- neighbouring buckets share a branch.  Lanes of one
// wave then mostly run the same branches.  (A second key, a coarse reduced temperature as predictor of
// the iteration count, measured 0.5-1.5 % slower on a 256-row domain: not used.)
__device__ __forceinline__ int k1_bucket(const double* row) {
    const bool polar = row[3] != 0.0;
    const bool assoc = (row[4] != 0.0) && (row[6] != 0.0 || row[7] != 0.0);
    return polar ? (assoc ? 2 : 1) : (assoc ? 3 : 0);
}

// The 256 staged rows of the workgroup are bucketed (LDS counting sort; the order inside a bucket is
// irrelevant) and lane t solves the row at sorted position t.  A larger bucketing domain (4 tiles per
// workgroup) measured only 2-3 % faster and costs either LDS occupancy or a non-inlined call per
// tile whose callee-saved registers go through scratch (3.8 GB of HBM writes per 1e7 rows): not used.
// LITE (pressure-only output, fp32 pre-solve available): vle_fast_lite; rows without a usable fp32 result are
// appended with bit 31 set (k_pure_vle_fallback takes them), rows for the robust pass without.
// POLISH (with LITE): the densities are handed out (rho_eq / rho_vl) and take the exact Newton update of vle_lite_finish<true>
template <bool LITE, bool POLISH = false>
__global__ __launch_bounds__(BLOCK, LITE ? K1_WAVES_LITE : K1_WAVES) void k_pure_vle(const double* __restrict__ params,
                                                    const double* __restrict__ temp, int64_t n,
                                                    double* __restrict__ p_sat, double* __restrict__ rho_eq,
                                                    double* __restrict__ rho_vl, uint8_t* __restrict__ status,
                                                    int32_t* __restrict__ iters, int32_t* __restrict__ retry) {
    __shared__ double lds[BLOCK * ROW_PAD];  // 8 parameters + T per row
    __shared__ int perm[BLOCK];
    __shared__ int bins[K1_BINS];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * BLOCK;
    // cooperative, coalesced staging (rows past n clamp to row n-1; never stored)
    {
        const double2* src = reinterpret_cast<const double2*>(params);
        const int64_t last2 = n * 4 - 1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int idx2 = t + k * BLOCK;
            int64_t g = row0 * 4 + idx2;
            if (g > last2) g = last2 - 3 + (idx2 & 3);
            double2 v = src[g];
            int r = idx2 >> 2, c2 = idx2 & 3;
            lds[r * ROW_PAD + 2 * c2] = v.x;
            lds[r * ROW_PAD + 2 * c2 + 1] = v.y;
        }
        const int64_t g = row0 + t;
        lds[t * ROW_PAD + 8] = temp[g < n ? g : n - 1];
        if (t < K1_BINS) bins[t] = 0;
    }
    __syncthreads();
    const int key = k1_bucket(&lds[t * ROW_PAD]);
    atomicAdd(&bins[key], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
#pragma unroll
        for (int b = 0; b < K1_BINS; b++) {
            int c = bins[b];
            bins[b] = acc;
            acc += c;
        }
    }
    __syncthreads();
    perm[atomicAdd(&bins[key], 1)] = t;
    __syncthreads();

    const int r = perm[t];
    const int64_t i = row0 + r;
    const bool live = i < n;
    double par[8];
#pragma unroll
    for (int q = 0; q < 8; q++) par[q] = lds[r * ROW_PAD + q];
    const double T = lds[r * ROW_PAD + 8];

    VleResult res;
    int st;  // wave-uniform calls
#ifdef PCS_F32_PRESOLVE
    if (LITE) st = vle_fast_lite<POLISH>(&lds[r * ROW_PAD], lds[r * ROW_PAD + 8], res);  // the row is re-read from LDS for the fp64 coefficients
    else st = rho_eq ? vle_fast<true>(par, T, res, 1e-8, TOL_STEP) : vle_fast<true>(par, T, res);
#else
    st = rho_eq ? vle_fast<false>(par, T, res, 1e-8, TOL_STEP) : vle_fast<false>(par, T, res);
#endif

    if (!live) return;
    if (st == ST_OK) {
        if (p_sat) p_sat[i] = res.p_star * T * P_UNIT;
        if (rho_eq) rho_eq[i] = res.rho_l * (1.0 / RHO_UNIT);
        if (rho_vl) {
            rho_vl[2 * i] = res.rho_v;
            rho_vl[2 * i + 1] = res.rho_l;
        }
        if (iters) iters[i] = res.iters;
        status[i] = 0;
    } else {
        status[i] = 1;  // provisional; a later pass overwrites it
        const int slot = atomicAdd(&retry[0], 1);
        // bounded append: a counter that was not reset (stale / foreign workspace) must not send the store out of the list
        if (slot >= 0 && slot < n) retry[1 + slot] = (int32_t)((uint32_t)i | (st == ST_FALLBACK ? 0x80000000u : 0u));  // n < 2^31 checked on the host
    }
}

// Rows of the list with bit 31 set (no usable fp32 pre-solve): the all-fp64 fast path.  Solved rows keep the
// bit (the robust pass skips them), rows that need the robust pass get it cleared.
#if PCS_PURE_PART == 1
constexpr int FALLBACK_GRID = 1024;
__global__ __launch_bounds__(64) void k_pure_vle_fallback(const double* __restrict__ params, const double* __restrict__ temp,
                                                          double* __restrict__ p_sat, double* __restrict__ rho_eq,
                                                          double* __restrict__ rho_vl,
                                                          uint8_t* __restrict__ status, int32_t* __restrict__ iters,
                                                          int32_t* __restrict__ retry, int64_t n) {
    // count and entries are bounded by n: a list that was not produced by the main kernel of THIS launch (stale or
    // uninitialised workspace handed to pcs_pure_vle_retry, a runtime that reorders the launches) must not fault
    const int count = (int)min((int64_t)max(retry[0], 0), n);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t entry = (uint32_t)retry[1 + k];
        if (!(entry & 0x80000000u)) continue;
        const int64_t i = (int64_t)(entry & 0x7fffffffu);
        if (i >= n) continue;
        double par[8];
#pragma unroll
        for (int j = 0; j < 8; j++) par[j] = params[8 * i + j];
        const double T = temp[i];
        VleResult res;
        int st = rho_eq ? vle_fast<false>(par, T, res, 1e-8, TOL_STEP) : vle_fast<false>(par, T, res);
        if (st == ST_OK) {
            if (p_sat) p_sat[i] = res.p_star * T * P_UNIT;
            if (rho_eq) rho_eq[i] = res.rho_l * (1.0 / RHO_UNIT);
            if (rho_vl) {
                rho_vl[2 * i] = res.rho_v;
                rho_vl[2 * i + 1] = res.rho_l;
            }
            if (iters) iters[i] = res.iters;
            status[i] = 0;
        } else {
            retry[1 + k] = (int32_t)i;  // bit 31 cleared: robust pass
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: liquid density at (T, p)
// ------------------------------------------------------------------------------------------
// liquid-density kernel: 204 VGPRs fit two waves per SIMD; held to 168 (52 spill instructions) three: 1.02 -> 0.89 ms per 1e7
// rows (four, 128 VGPRs: 1.02 ms)
#endif  // part 1

constexpr int K4_WAVES = 2;  // Jacobian kernels: two (256 VGPRs); held to 168 for three they spill inside the DN<9> pass: 0.89 -> 2.24 ms
constexpr int K2_WAVES = 3;
#if PCS_PURE_PART == 2
__global__ __launch_bounds__(BLOCK, K2_WAVES) void k_pure_liquid_density(const double* __restrict__ params,
                                                               const double* __restrict__ temp,
                                                               const double* __restrict__ pressure, int64_t n,
                                                               double* __restrict__ rho_out,
                                                               double* __restrict__ rho_root,
                                                               uint8_t* __restrict__ status) {
    __shared__ double lds[BLOCK * ROW_PAD];
    const int64_t row0 = (int64_t)blockIdx.x * BLOCK;
    double par[8];
    // rows of the workgroup bucketed by class as in k_pure_vle (LDS counting sort): a wave mostly runs one set of
    // branches of the Helmholtz energy
    __shared__ int perm[BLOCK];
    __shared__ int bins[K1_BINS];
    const int t = threadIdx.x;
    if (t < K1_BINS) bins[t] = 0;
    stage_params(params, n, row0, lds, par);
    const int key = k1_bucket(&lds[t * ROW_PAD]);
    atomicAdd(&bins[key], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
#pragma unroll
        for (int b = 0; b < K1_BINS; b++) {
            int c = bins[b];
            bins[b] = acc;
            acc += c;
        }
    }
    __syncthreads();
    perm[atomicAdd(&bins[key], 1)] = t;
    __syncthreads();
    const int src = perm[t];
#pragma unroll
    for (int q = 0; q < 8; q++) par[q] = lds[src * ROW_PAD + q];
    const int64_t i = row0 + src;
    const bool live = i < n;
    const int64_t ii = live ? i : n - 1;
    const double T = temp[ii];
    const double p_red = pressure[ii] / (T * P_UNIT);  // pcsaft_pure.py:196

    PureCoef<double> c;
    pure_coef<double>(c, par, T, false);
    double rho;
    Eval last;
    int st = liquid_density_solve(c, p_red, TOL_STEP, rho, last);
    if (!live) return;
    // `rho` already carries the final Newton update rho - (p - p_spec)/dp (pcsaft_pure.py:198) taken
    // at a point whose relative step was <= TOL_STEP, i.e. it is converged to ~1e-11; it is both
    // the returned property and the density reported as the root.
    bool ok = (st == ST_OK);
    if (rho_out) rho_out[i] = ok ? rho * (1.0 / RHO_UNIT) : 0.0;
    if (rho_root) rho_root[i] = ok ? rho : 0.0;
    status[i] = ok ? 0 : 1;
}
#endif  // part 2

#if PCS_PURE_PART == 1
// ------------------------------------------------------------------------------------------
// derivatives (a, p, dp) at given (T, rho)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_pure_derivatives(const double* __restrict__ params,
                                                            const double* __restrict__ temp,
                                                            const double* __restrict__ rho_in, int64_t n,
                                                            double* __restrict__ a, double* __restrict__ p,
                                                            double* __restrict__ dp) {
    __shared__ double lds[BLOCK * ROW_PAD];
    const int64_t row0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = row0 + threadIdx.x;
    const bool live = i < n;
    double par[8];
    stage_params(params, n, row0, lds, par);
    const int64_t ii = live ? i : n - 1;
    PureCoef<double> c;
    pure_coef<double>(c, par, temp[ii], false);
    Eval e = pure_eval(c, rho_in[ii]);
    if (!live) return;
    if (a) a[i] = e.a;
    if (p) p[i] = e.p;
    if (dp) dp[i] = e.dp;
}

// ------------------------------------------------------------------------------------------
// K4: Jacobian of a property w.r.t. (8 parameters, T, p) at fixed densities
// ------------------------------------------------------------------------------------------
template <int WHICH>
__global__ __launch_bounds__(BLOCK, K4_WAVES) void k_pure_jacobian(const double* __restrict__ params,
                                                         const double* __restrict__ temp,
                                                         const double* __restrict__ pressure,
                                                         const double* __restrict__ rho_vl, int64_t n,
                                                         double* __restrict__ jac, const double* __restrict__ gout,
                                                         double* __restrict__ grad_params, double* __restrict__ grad_temp,
                                                         double* __restrict__ grad_press, int polish) {
    __shared__ double lds[BLOCK * ROW_PAD];
    const int64_t row0 = (int64_t)blockIdx.x * BLOCK;
    double par[8];
    // class bucketing as in k_pure_vle for the two liquid-density properties (A/B on 1e7 rows: 4.0 -> 3.9 ms and
    // 12.0 -> 8.4 ms; the vapour-pressure Jacobian gets slightly slower with it, 3.5 -> 3.7 ms, and keeps the row order)
    constexpr bool BUCKET = WHICH != 0;
    __shared__ int perm[BUCKET ? BLOCK : 1];
    __shared__ int bins[K1_BINS];
    const int t = threadIdx.x;
    if (BUCKET && t < K1_BINS) bins[t] = 0;
    stage_params(params, n, row0, lds, par);
    int src = t;
    if (BUCKET) {
        const int key = k1_bucket(&lds[t * ROW_PAD]);
        atomicAdd(&bins[key], 1);
        __syncthreads();
        if (t == 0) {
            int acc = 0;
#pragma unroll
            for (int b = 0; b < K1_BINS; b++) {
                int c = bins[b];
                bins[b] = acc;
                acc += c;
            }
        }
        __syncthreads();
        perm[atomicAdd(&bins[key], 1)] = t;
        __syncthreads();
        src = perm[t];
#pragma unroll
        for (int q = 0; q < 8; q++) par[q] = lds[src * ROW_PAD + q];
    }
    const int64_t i = row0 + src;
    const bool live = i < n;
    const int64_t ii = live ? i : n - 1;
    const double T = temp[ii];
    const double p_pa = (WHICH == 1) ? pressure[ii] : 0.0;
    const double rv = rho_vl[2 * ii], rl = rho_vl[2 * ii + 1];
    double g[10];
    pure_jacobian<WHICH>(par, T, p_pa, rv, rl, g, polish != 0);
    if (!live) return;
    // a failed row carries zero densities -> NaNs; the caller masks by status
    if (jac) {
#pragma unroll
        for (int k = 0; k < 10; k++) jac[10 * i + k] = g[k];
    }
    if (gout) {  // vector-Jacobian form (the backward pass of a property): upstream gradient folded in, no [n,10] round trip
        const double w = gout[i];
        if (grad_params) {
            double2* dst = reinterpret_cast<double2*>(grad_params + 8 * i);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = make_double2(w * g[2 * k], w * g[2 * k + 1]);
        }
        if (grad_temp) grad_temp[i] = w * g[8];
        if (grad_press) grad_press[i] = w * g[9];
    }
}

// ------------------------------------------------------------------------------------------
// vector-Jacobian product of (a, p, dp) w.r.t. (8 parameters, T, rho): autograd of PcSaftPure.derivatives / helmholtz_energy
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_pure_derivatives_vjp(const double* __restrict__ params,
                                                                const double* __restrict__ temp,
                                                                const double* __restrict__ rho_in, int64_t n,
                                                                const double* __restrict__ g_a, const double* __restrict__ g_p,
                                                                const double* __restrict__ g_dp,
                                                                double* __restrict__ grad_params, double* __restrict__ grad_temp,
                                                                double* __restrict__ grad_rho) {
    __shared__ double lds[BLOCK * ROW_PAD];
    const int64_t row0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = row0 + threadIdx.x;
    const bool live = i < n;
    double par[8];
    stage_params(params, n, row0, lds, par);
    const int64_t ii = live ? i : n - 1;
    double g[VJP_DIRS];
    pure_derivatives_vjp(par, temp[ii], rho_in[ii], g_a ? g_a[ii] : 0.0, g_p ? g_p[ii] : 0.0, g_dp ? g_dp[ii] : 0.0, g);
    if (!live) return;
    if (grad_params) {
#pragma unroll
        for (int k = 0; k < 8; k++) grad_params[8 * i + k] = g[k];
    }
    if (grad_temp) grad_temp[i] = g[8];
    if (grad_rho) grad_rho[i] = g[9];
}

#endif  // part 1

}  // namespace

#if PCS_PURE_PART == 2
// launch functions of the kernels compiled without re-association (called from part 1)
namespace pcs_abi {
int launch_pure_vle_full(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq, double* rho_vl,
                         uint8_t* status, int32_t* iters, int32_t* retry, hipStream_t s) {
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_pure_vle<false>, dim3(grid), dim3(BLOCK), 0, s, params, temp, n, p_sat, rho_eq, rho_vl, status, iters,
                       retry);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_vle launch", e);
    return 0;
}
int launch_pure_liquid_density(const double* params, const double* temp, const double* pressure, int64_t n, double* rho_out,
                               double* rho_root, uint8_t* status, hipStream_t s) {
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_pure_liquid_density, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, n, rho_out, rho_root, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_liquid_density launch", e);
    return 0;
}
}  // namespace pcs_abi
#endif

#if PCS_PURE_PART == 1
extern "C" {

int pcs_abi_version(void) { return 104; }  // 103: pcs_pure_vapor_pressure, pcs_compact_* / pcs_expand_rows; 104: pcs_pure_vle_fp64

const char* pcs_last_error(void) { return g_err; }

// retry list of the pure / gc two-pass schedules (n + 1 ints) or row order + control block of the mixture work queue (n + 64)
int64_t pcs_workspace_bytes(int64_t n) { return (int64_t)sizeof(int32_t) * (n + 64); }

// stage 1: zero the retry counter and run the fast kernel over all rows
static int launch_vle_fast(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                           double* rho_vl, uint8_t* status, int32_t* iters, int32_t* retry, hipStream_t s,
                           bool force_lite = false, bool all_fp64 = false) {
    if (int ez = zero_ints(retry, 1, s)) return ez;
    hipError_t e;
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    // main kernel (lean: rows without an fp32 pre-solve go to the list with bit 31 set) + all-fp64 fallback kernel.
    //  * pressure only (and pcs_pure_vapor_pressure, whatever it hands out): k_pure_vle<true>, densities converged to ~1e-9
    //    (enough for p*, whose error is of second order in them);
    //  * rho_vl without rho_eq (the Jacobian kernels' input), and everything under all_fp64 (pcs_pure_vle_fp64):
    //    k_pure_vle<false>, the fp64 D2 iteration from the fp32 pre-solve's start -- one iteration at the pressure
    //    tolerances, densities ~1e-11 (0.945 ms per 1e7 rows against 0.990 for the next form);
    //  * rho_eq requested (tolerance 1e-8 on the liquid step: two D2 iterations of the all-fp64 kernel, 1.55-1.67 ms): the
    //    pressure-only kernel + the exact Newton update of vle_lite_finish<true> instead, 1.22 ms (round 3).
    if (all_fp64 || (!rho_eq && rho_vl && !force_lite)) {
        if (int ef = launch_pure_vle_full(params, temp, n, p_sat, rho_eq, rho_vl, status, iters, retry, s)) return ef;
    } else if (!rho_eq) {
        hipLaunchKernelGGL((k_pure_vle<true, false>), dim3(grid), dim3(BLOCK), 0, s, params, temp, n, p_sat, rho_eq, rho_vl, status,
                           iters, retry);
    } else {
        hipLaunchKernelGGL((k_pure_vle<true, true>), dim3(grid), dim3(BLOCK), 0, s, params, temp, n, p_sat, rho_eq, rho_vl, status,
                           iters, retry);
    }
    hipLaunchKernelGGL(k_pure_vle_fallback, dim3(FALLBACK_GRID), dim3(64), 0, s, params, temp, p_sat, rho_eq, rho_vl, status,
                       iters, retry, n);
    e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_vle launch", e);
    return 0;
}

static int vle_args_ok(const double* params, const double* temp, int64_t n, const uint8_t* status,
                       const void* workspace) {
    if (int e = check_n(n)) return e;
    if (n > 0 && (!params || !temp || !status || !workspace)) return fail_msg("pcs_pure_vle: null required pointer");
    return 0;
}

int pcs_pure_vle(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                 double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = vle_args_ok(params, temp, n, status, workspace)) return e;
    if (n == 0) return 0;
    int32_t* retry = static_cast<int32_t*>(workspace);
    if (int e = launch_vle_fast(params, temp, n, p_sat, rho_eq, rho_vl, status, iters, retry, as_stream(stream))) return e;
    return launch_vle_retry(params, temp, p_sat, rho_eq, rho_vl, status, iters, retry, n, as_stream(stream));
}

int pcs_pure_vle_fp64(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                      double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = vle_args_ok(params, temp, n, status, workspace)) return e;
    if (n == 0) return 0;
    int32_t* retry = static_cast<int32_t*>(workspace);
    if (int e = launch_vle_fast(params, temp, n, p_sat, rho_eq, rho_vl, status, iters, retry, as_stream(stream), false, true)) return e;
    return launch_vle_retry(params, temp, p_sat, rho_eq, rho_vl, status, iters, retry, n, as_stream(stream));
}

int pcs_pure_vapor_pressure(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_vl,
                            uint8_t* status, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = vle_args_ok(params, temp, n, status, workspace)) return e;
    if (n == 0) return 0;
    int32_t* retry = static_cast<int32_t*>(workspace);
    if (int e = launch_vle_fast(params, temp, n, p_sat, nullptr, rho_vl, status, nullptr, retry, as_stream(stream), true)) return e;
    return launch_vle_retry(params, temp, p_sat, nullptr, rho_vl, status, nullptr, retry, n, as_stream(stream));
}

int pcs_pure_vle_fast(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                      double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = vle_args_ok(params, temp, n, status, workspace)) return e;
    if (n == 0) return 0;
    return launch_vle_fast(params, temp, n, p_sat, rho_eq, rho_vl, status, iters, static_cast<int32_t*>(workspace),
                           as_stream(stream));
}

int pcs_pure_vle_retry(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                       double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = vle_args_ok(params, temp, n, status, workspace)) return e;
    if (n == 0) return 0;
    return launch_vle_retry(params, temp, p_sat, rho_eq, rho_vl, status, iters,
                            static_cast<const int32_t*>(workspace), n, as_stream(stream));
}

int pcs_pure_liquid_density(const double* params, const double* temp, const double* pressure, int64_t n,
                            double* rho_out, double* rho_root, uint8_t* status, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !temp || !pressure || !status) return fail_msg("pcs_pure_liquid_density: null required pointer");
    return launch_pure_liquid_density(params, temp, pressure, n, rho_out, rho_root, status, as_stream(stream));
}

int pcs_pure_derivatives(const double* params, const double* temp, const double* rho, int64_t n, double* a,
                         double* p, double* dp, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !temp || !rho) return fail_msg("pcs_pure_derivatives: null required pointer");
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_pure_derivatives, dim3(grid), dim3(BLOCK), 0, as_stream(stream), params, temp, rho, n, a, p,
                       dp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_derivatives launch", e);
    return 0;
}

int pcs_pure_derivatives_vjp(const double* params, const double* temp, const double* rho, int64_t n, const double* g_a,
                             const double* g_p, const double* g_dp, double* grad_params, double* grad_temp, double* grad_rho,
                             void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !temp || !rho) return fail_msg("pcs_pure_derivatives_vjp: null required pointer");
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_pure_derivatives_vjp, dim3(grid), dim3(BLOCK), 0, as_stream(stream), params, temp, rho, n, g_a, g_p,
                       g_dp, grad_params, grad_temp, grad_rho);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_derivatives_vjp launch", e);
    return 0;
}

int pcs_pure_jacobian(int which, const double* params, const double* temp, const double* pressure,
                      const double* rho_vl, int64_t n, double* jac, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !temp || !rho_vl || !jac) return fail_msg("pcs_pure_jacobian: null required pointer");
    const int polish = (which & PCS_JAC_POLISH) ? 1 : 0;
    which &= ~PCS_JAC_POLISH;
    if (which == 1 && !pressure) return fail_msg("pcs_pure_jacobian: pressure required for liquid_density");
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipStream_t s = as_stream(stream);
    switch (which) {
        case 0: hipLaunchKernelGGL(k_pure_jacobian<0>, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, rho_vl, n, jac, nullptr, nullptr, nullptr, nullptr, polish); break;
        case 1: hipLaunchKernelGGL(k_pure_jacobian<1>, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, rho_vl, n, jac, nullptr, nullptr, nullptr, nullptr, polish); break;
        case 2: hipLaunchKernelGGL(k_pure_jacobian<2>, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, rho_vl, n, jac, nullptr, nullptr, nullptr, nullptr, polish); break;
        default: return fail_msg("pcs_pure_jacobian: which must be 0, 1 or 2");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_jacobian launch", e);
    return 0;
}

int pcs_pure_jacobian_vjp(int which, const double* params, const double* temp, const double* pressure, const double* rho_vl,
                          const double* gout, int64_t n, double* grad_params, double* grad_temp, double* grad_pressure,
                          void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !temp || !rho_vl || !gout) return fail_msg("pcs_pure_jacobian_vjp: null required pointer");
    const int polish = (which & PCS_JAC_POLISH) ? 1 : 0;
    which &= ~PCS_JAC_POLISH;
    if (which == 1 && !pressure) return fail_msg("pcs_pure_jacobian_vjp: pressure required for liquid_density");
    if ((reinterpret_cast<uintptr_t>(grad_params) & 15) != 0) return fail_msg("pcs_pure_jacobian_vjp: grad_params must be 16-byte aligned");
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipStream_t s = as_stream(stream);
    double* none = nullptr;
    switch (which) {
        case 0: hipLaunchKernelGGL(k_pure_jacobian<0>, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, rho_vl, n, none, gout, grad_params, grad_temp, grad_pressure, polish); break;
        case 1: hipLaunchKernelGGL(k_pure_jacobian<1>, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, rho_vl, n, none, gout, grad_params, grad_temp, grad_pressure, polish); break;
        case 2: hipLaunchKernelGGL(k_pure_jacobian<2>, dim3(grid), dim3(BLOCK), 0, s, params, temp, pressure, rho_vl, n, none, gout, grad_params, grad_temp, grad_pressure, polish); break;
        default: return fail_msg("pcs_pure_jacobian_vjp: which must be 0, 1 or 2");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_jacobian launch", e);
    return 0;
}

}  // extern "C"
#endif  // part 1
