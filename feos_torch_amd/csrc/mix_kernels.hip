// gfx950 kernels + C ABI for the binary-mixture PC-SAFT path (bubble / dew points,
// PcSaftMix.derivatives).  One state point per lane; see pure_kernels.hip for the launch-shape
// rationale.  Per-row inputs: parameters [n,2,8] (128 B AoS, read with four 16-byte loads per
// component row), kij [n,2], T, z, p_init — 168 B read, 8 B + 32 B + 1 B written per row against
// ~1e5 fp64 operations: compute bound by three orders of magnitude.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"
#include "mix_model.hpp"
#include <cstddef>
#include "mix_solver.hpp"
#include "mix_solver_sm.hpp"
#ifndef PCS_MIX_SM
#define PCS_MIX_SM 1  // 1: state-machine form of the solver (mix_solver_sm.hpp), 0: sequential form
#endif
#if PCS_MIX_SM
#define PCS_BD_SOLVE bubble_dew_solve_sm
#else
#define PCS_BD_SOLVE bubble_dew_solve
#endif
#include "mix_jacobian.hpp"

using namespace pcs;
using namespace pcs_abi;

namespace {

#ifndef PCS_MBLOCK
#define PCS_MBLOCK 128
#endif
constexpr int MBLOCK = PCS_MBLOCK;

// PCS_MIX_PRELOAD = 1 (experiment, off): all coefficient loads of the evaluation function issued up front.  The ten
// s_waitcnt of the function become one; SQ_WAIT_ANY drops 8.6e8 -> 6.9e8 quad-cycles per dew launch and SQ_WAIT_INST_ANY
// rises by as much -- the kernel time does not move (2.84 / 6.44 ms vs 2.81 / 6.43 ms).  The single resident wave is
// stalled by instruction dependencies, not by these loads.
#ifndef PCS_MIX_PRELOAD
#define PCS_MIX_PRELOAD 0
#endif
struct MixModel {
    MixCoef<double> c;
#if defined(PCS_MIX_DIAG) && PCS_MIX_DIAG == 2  // diagnostics builds: evaluation counters
    mutable int n_line = 0, n_phase = 0;
    template <class R> PCS_DEV R a(const R& r0, const R& r1) const {
        if (sizeof(R) == sizeof(T2<double>)) n_phase++; else n_line++;
        return mix_a<double, R>(c, r0, r1);
    }
#else
    template <class R> PCS_DEV R a(const R& r0, const R& r1) const {
#if PCS_MIX_PRELOAD
        // The solvers' evaluation function receives the model by reference (private memory).  Left to the compiler its
        // ~37 loads sit in ten groups next to their uses, each followed by s_waitcnt: ten exposed memory round trips per
        // call with one wave per SIMD (SQ_WAIT_ANY = 23 % of the wave cycles).  Here every coefficient the row needs is
        // read up front -- the empty asm statements keep the loads above them and their results in registers -- so one
        // round trip is exposed instead.  (Volatile loads do not do it: the backend waits after each of them.)
        if (sizeof(R) == sizeof(T2<double>)) {
            MixCoef<double> l;
            const double* src = reinterpret_cast<const double*>(&c);
            double* dst = reinterpret_cast<double*>(&l);
            constexpr int N = sizeof(MixCoef<double>) / 8;
            static_assert(offsetof(MixCoef<double>, polar) % 8 == 0 && offsetof(MixCoef<double>, acls) % 8 == 0, "flag slots");
            // (pj, tj of a non-polar row hold whatever mix_coef left there: loaded all the same, never used)
#pragma unroll
            for (int k = 0; k < N; k++) dst[k] = src[k];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < N; k++) asm volatile("" : "+v"(dst[k]));
            return mix_a<double, R>(l, r0, r1);
        }
#endif
        return mix_a<double, R>(c, r0, r1);
    }
#endif
    PCS_DEV double packing(double x0, double x1) const { return x0 * c.zk[3][0] + x1 * c.zk[3][1]; }
};

__device__ __forceinline__ void load_mix_row(const double* __restrict__ params, const double* __restrict__ kij,
                                             int64_t i, double par[16], double& k0, double& k1) {
    const double2* src = reinterpret_cast<const double2*>(params + 16 * i);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        double2 v = src[k];
        par[2 * k] = v.x;
        par[2 * k + 1] = v.y;
    }
    double2 kk = reinterpret_cast<const double2*>(kij)[i];
    k0 = kk.x;
    k1 = kk.y;
}

#ifndef PCS_MIX_QUEUE
#define PCS_MIX_QUEUE 1  // 1: class-ordered work queue with persistent waves, 0: fast pass + retry pass
#endif
#ifndef PCS_INPLACE_ROBUST
#define PCS_INPLACE_ROBUST 1  // 0 (A/B builds): the work-queue kernel does not restart failed rows
#endif
#ifndef PCS_REFILL_MIN
#define PCS_REFILL_MIN 8  // lanes that must be idle before the wave refills (A/B dew 1e6 rows: 1: 7.9 ms, 4: 7.7, 8: 7.5, 16: 7.8)
#endif
#ifndef PCS_QUEUE_WAVES_PER_SIMD
#define PCS_QUEUE_WAVES_PER_SIMD 1
#endif
#ifndef PCS_FAST_SS
#define PCS_FAST_SS 12
#define PCS_FAST_NEWTON 12
#endif
constexpr int FAST_SS = PCS_FAST_SS, FAST_NEWTON = PCS_FAST_NEWTON;  // iteration caps of the fast pass (see mix_solver.hpp)
constexpr int MIX_RETRY_GRID = 2048;             // 64-thread workgroups of the robust pass

constexpr int MIX_BINS = 8;
// association class (none, self, induced, cross: mix_model.hpp) x polarity of a parameter row [2][8]
__device__ __forceinline__ int mix_bucket(const double* __restrict__ row) {
    const double na0 = row[6], nb0 = row[7], na1 = row[14], nb1 = row[15];
    const int associating = (na0 + nb0 != 0.0) + (na1 + nb1 != 0.0);
    const int self_assoc = (na0 * nb0 != 0.0) + (na1 * nb1 != 0.0);
    int cls = 0;
    if (associating == 1 && self_assoc == 1) cls = 1;
    if (associating == 2 && self_assoc == 1) cls = 2;
    if (associating == 2 && self_assoc == 2) cls = 3;
    const int polar = (row[3] != 0.0) || (row[11] != 0.0);
    return 2 * cls + polar;
}

template <bool DEW>
__device__ __forceinline__ void mix_store(int64_t i, int rc, const MixResult& r, double T, double* __restrict__ p_out,
                                          double* __restrict__ rho4, uint8_t* __restrict__ status,
                                          int32_t* __restrict__ iters) {
    const bool ok = rc == BD_OK;
    if (p_out) p_out[i] = ok ? r.p * T * P_UNIT : 0.0;
    if (rho4) {
        // reference layout (src/pcsaft.rs:225-228): [rhoV_1, rhoV_2, rhoL_1, rhoL_2]
        double v0 = DEW ? r.spec0 : r.inc0, v1 = DEW ? r.spec1 : r.inc1;
        double l0 = DEW ? r.inc0 : r.spec0, l1 = DEW ? r.inc1 : r.spec1;
        reinterpret_cast<double4*>(rho4)[i] = ok ? make_double4(v0, v1, l0, l1) : make_double4(0.0, 0.0, 0.0, 0.0);
    }
#ifdef PCS_MIX_DIAG
    if (iters) iters[i] = r.iters;
#else
    if (iters) iters[i] = ok ? r.iters : -1;
#endif
    status[i] = ok ? 0 : 1;
}

// K5 fast pass: small iteration caps; rows that hit a cap go to the retry list (retry[0] = count).
// retry == nullptr: single pass with the full caps.
#ifndef MIX_WAVES
#define MIX_WAVES 1  // the non-inlined evaluation needs the whole register file
#endif
template <bool DEW>
__global__ __launch_bounds__(MBLOCK, MIX_WAVES) void k_mix_bubble_dew(const double* __restrict__ params,
                                                           const double* __restrict__ kij,
                                                           const double* __restrict__ temp,
                                                           const double* __restrict__ z,
                                                           const double* __restrict__ p_init, int64_t n,
                                                           double* __restrict__ p_out, double* __restrict__ rho4,
                                                           uint8_t* __restrict__ status, int32_t* __restrict__ iters,
                                                           int32_t* __restrict__ retry) {
    // bucket the rows of the workgroup by association class and polarity (LDS counting sort) so the lanes of
    // a wave mostly run the same branches of the Helmholtz energy: the cross-association site-fraction solve
    // costs ~5x the rest of an evaluation and would otherwise be paid by every wave
    __shared__ int perm[MBLOCK];
    __shared__ int bins[MIX_BINS + 1];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * MBLOCK;
    if (t <= MIX_BINS) bins[t] = 0;
    __syncthreads();
    int key = MIX_BINS;  // rows past n sort last
    if (row0 + t < n) key = mix_bucket(params + 16 * (row0 + t));
    atomicAdd(&bins[key], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
#pragma unroll
        for (int b = 0; b <= MIX_BINS; b++) {
            int c = bins[b];
            bins[b] = acc;
            acc += c;
        }
    }
    __syncthreads();
    perm[atomicAdd(&bins[key], 1)] = t;
    __syncthreads();
    const int64_t i = row0 + perm[t];
    if (i >= n) return;
    double par[16], k0, k1;
    load_mix_row(params, kij, i, par, k0, k1);
    const double T = temp[i];
    MixModel m;
    mix_coef<double>(m.c, par, k0, k1, T);
    MixResult r;
    const double p_red = p_init[i] / (T * P_UNIT);
#ifdef PCS_MIX_DIAG
    long long t0 = clock64();
#endif
#if PCS_MIX_SM
    bool root_failed = false;
    int rc = bubble_dew_solve_sm<DEW>(m, z[i], p_red, r, retry ? FAST_SS : SS_MAX_IT, retry ? FAST_NEWTON : NEWTON_MAX_IT, false, &root_failed);
    // single pass without a work list (workspace == NULL): the robust second attempt runs in place
    if (!retry && rc != BD_OK && root_failed) rc = bubble_dew_solve_sm<DEW>(m, z[i], p_red, r, SS_MAX_IT, NEWTON_MAX_IT, true);
#else
    int rc = PCS_BD_SOLVE<DEW>(m, z[i], p_red, r, retry ? FAST_SS : SS_MAX_IT, retry ? FAST_NEWTON : NEWTON_MAX_IT);
#endif
#ifdef PCS_MIX_DIAG
#if PCS_MIX_DIAG == 3
    // r.iters already holds the evaluation count
#elif PCS_MIX_DIAG == 2
    r.iters = m.n_phase | (m.n_line << 12);
#else
    r.iters = (int)((clock64() - t0) >> 10);  // diagnostics builds: per-row solve time in 1024-cycle units
#endif
    if (rc == BD_CAP && iters) iters[i] = r.iters;
#endif
    if (rc == BD_CAP) {
        status[i] = 1;  // provisional
        const int slot = atomicAdd(&retry[0], 1);
        if (slot >= 0 && slot < n) retry[1 + slot] = (int32_t)i;  // bounded append (see pure_kernels.hip)
        return;
    }
    mix_store<DEW>(i, rc, r, T, p_out, rho4, status, iters);
}

// K5 second pass over a compacted list (count read on the device; PCS_MIX_QUEUE = 0 builds): the rows that hit a cap of the
// fast pass, same arithmetic with the full caps, then -- as everywhere -- the robust attempt for rows that still fail
template <bool DEW>
__global__ __launch_bounds__(64, MIX_WAVES) void k_mix_bubble_dew_retry(const double* __restrict__ params,
                                                             const double* __restrict__ kij,
                                                             const double* __restrict__ temp,
                                                             const double* __restrict__ z,
                                                             const double* __restrict__ p_init,
                                                             double* __restrict__ p_out, double* __restrict__ rho4,
                                                             uint8_t* __restrict__ status, int32_t* __restrict__ iters,
                                                             const int32_t* __restrict__ retry, int64_t n) {
    const int count = (int)min((int64_t)max(retry[0], 0), n);  // count and entries bounded by n: a foreign list must not fault
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const int64_t i = retry[1 + k];
        if (i < 0 || i >= n) continue;
        double par[16], k0, k1;
        load_mix_row(params, kij, i, par, k0, k1);
        const double T = temp[i];
        MixModel m;
        mix_coef<double>(m.c, par, k0, k1, T);
        MixResult r;
#ifdef PCS_MIX_DIAG
        long long t0 = clock64();
#endif
#if PCS_MIX_SM
        bool root_failed = false;
        int rc = bubble_dew_solve_sm<DEW>(m, z[i], p_init[i] / (T * P_UNIT), r, SS_MAX_IT, NEWTON_MAX_IT, false, &root_failed);
        if (rc != BD_OK && root_failed) rc = bubble_dew_solve_sm<DEW>(m, z[i], p_init[i] / (T * P_UNIT), r, SS_MAX_IT, NEWTON_MAX_IT, true);
#else
        int rc = PCS_BD_SOLVE<DEW>(m, z[i], p_init[i] / (T * P_UNIT), r);
#endif
#ifdef PCS_MIX_DIAG
#if PCS_MIX_DIAG == 3
        r.iters |= (1 << 30);
#elif PCS_MIX_DIAG == 2
        r.iters = m.n_phase | (m.n_line << 12) | (1 << 30);
#else
        r.iters = (int)((clock64() - t0) >> 10) | (1 << 30);
#endif
#endif
        mix_store<DEW>(i, rc, r, T, p_out, rho4, status, iters);
    }
}

// ------------------------------------------------------------------------------------------
// K5 as a work queue.  The iteration counts of the rows differ by an order of magnitude (5 ... 400
// evaluations), so with one row per lane a wave idles most of its lanes most of the time.  Here the
// rows are first ordered by class (device counting sort -> perm, expensive classes first so their long
// rows overlap with the bulk), then a fixed set of resident waves works through that order: a lane that
// finishes its row stores it and takes the next one, every pass of the wave's loop is one evaluation
// for all its lanes.  No second pass: slow rows just keep their lane longer.
// control block (int32, after perm[n] in the workspace): [0..8] class counts -> offsets, [16] queue head
// ------------------------------------------------------------------------------------------
constexpr int QCTRL_HEAD = 16, QCTRL_INTS = 64;
constexpr int QCHUNK = 64;  // rows a wave reserves per atomic on the queue head

__device__ __forceinline__ int mix_queue_bin(const double* __restrict__ row) { return MIX_BINS - 1 - mix_bucket(row); }

__global__ __launch_bounds__(256) void k_mix_class_count(const double* __restrict__ params, int64_t n,
                                                         int32_t* __restrict__ ctrl) {
    __shared__ int cnt[MIX_BINS];
    if (threadIdx.x < MIX_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[mix_queue_bin(params + 16 * i)], 1);
    __syncthreads();
    if (threadIdx.x < MIX_BINS && cnt[threadIdx.x]) atomicAdd(&ctrl[threadIdx.x], cnt[threadIdx.x]);
}

__global__ void k_mix_class_scan(int32_t* __restrict__ ctrl) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < MIX_BINS; b++) {
            int c = ctrl[b];
            ctrl[b] = acc;
            acc += c;
        }
        ctrl[QCTRL_HEAD] = 0;
    }
}

__global__ __launch_bounds__(256) void k_mix_class_scatter(const double* __restrict__ params, int64_t n,
                                                           int32_t* __restrict__ ctrl, int32_t* __restrict__ perm) {
    __shared__ int cnt[MIX_BINS], base[MIX_BINS];
    if (threadIdx.x < MIX_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int bin = 0, rank = 0;
    if (i < n) {
        bin = mix_queue_bin(params + 16 * i);
        rank = atomicAdd(&cnt[bin], 1);
    }
    __syncthreads();
    if (threadIdx.x < MIX_BINS && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&ctrl[threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
    if (i < n) perm[base[bin] + rank] = (int32_t)i;
}

// PCS_QUEUE_LDS_COEF = 1: the work-queue kernel keeps each lane's model coefficients in LDS (520 B per lane, 32.5 KB per
// wave, one wave per SIMD: 130 of the CU's 160 KB).  The evaluation is a non-inlined call (it needs the whole VGPR file), so the coefficients
// cannot stay in registers across it; handed over by reference they live in private memory and the callee reads them back
// with ~37 flat loads per call (a memory round trip the single resident wave cannot hide: 24 % of the wave's cycles were
// SQ_WAIT_ANY).  With the model object in LDS the same flat loads resolve in the LDS aperture.
// Measured (round 2, 1e6 rows): SQ_WAIT_ANY 8.7e8 -> 6.1e8 quad-cycles per dew launch, but the build then saves the
// callee-saved VGPRs of the evaluation function through scratch (79 stores + 79 loads per call) and the kernel time does
// not move (2.92 / 6.64 ms against 2.90 / 6.70 ms): off.
#ifndef PCS_QUEUE_LDS_COEF
#define PCS_QUEUE_LDS_COEF 0
#endif
// PCS_QUEUE_LDS_STATE = 1: the per-lane solver state (BdLane, 47 doubles) of the work-queue kernel lives in LDS (24 KB per
// wave) instead of AGPRs: 1,613 -> 453 AGPR moves and 4,427 -> 3,155 VALU instructions in the kernel body, bubble 2.87 -> 2.71 ms,
// dew 6.43 -> 6.25 ms per 1e6 rows, identical results.  With the state out of the registers two waves per SIMD were tried again
// (PCS_QUEUE_WAVES_PER_SIMD = 2): the model coefficients and the loop's own values then spill (344 scratch instructions) and LDS
// holds six waves per CU: 3.56 / 9.86 ms.
#ifndef PCS_QUEUE_LDS_STATE
#define PCS_QUEUE_LDS_STATE 1
#endif
struct alignas(16) MixModelSlot {
    MixModel m;
    double pad[(520 - sizeof(MixModel) % 520) / 8];  // lane stride 520 B = 130 dwords: consecutive lanes two banks apart
};

template <bool DEW>
__global__ __launch_bounds__(64, PCS_QUEUE_WAVES_PER_SIMD) void k_mix_bubble_dew_queue(const double* __restrict__ params,
                                                             const double* __restrict__ kij,
                                                             const double* __restrict__ temp,
                                                             const double* __restrict__ z,
                                                             const double* __restrict__ p_init, int64_t n,
                                                             const int32_t* __restrict__ perm, int32_t* __restrict__ ctrl,
                                                             double* __restrict__ p_out, double* __restrict__ rho4,
                                                             uint8_t* __restrict__ status, int32_t* __restrict__ iters) {
#if PCS_QUEUE_LDS_COEF
    __shared__ MixModelSlot slots[64];
#endif
    const int lane = threadIdx.x;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int total = (int)n;
    int next = 0, end = 0;  // this wave's reserved slice [next, end) of the queue (wave-uniform)
    bool drained = false;   // queue head passed n (wave-uniform)
#if PCS_QUEUE_LDS_STATE
    // the solver state of every lane in LDS (padded to an odd number of doubles per lane): nothing of it is live in
    // registers across the evaluation call, so the kernel fits the 256 registers of two waves per SIMD
    struct alignas(8) LaneSlot {
        BdLane<DEW> L;
        double pad[(sizeof(BdLane<DEW>) / 8) % 2 == 0 ? 1 : 2];
    };
    __shared__ LaneSlot lane_slots[64];
    BdLane<DEW>& L = lane_slots[threadIdx.x].L;
#else
    BdLane<DEW> L;
#endif
    L.idle();
#if PCS_QUEUE_LDS_COEF
    MixModel& m = slots[threadIdx.x].m;
#else
    MixModel m;
#endif
    int64_t row = -1;
    double T = 0.0;
    int evals = 0;  // evaluations of this lane's current row (bounded by BD_EVAL_GUARD: the loop below cannot hang)
#if defined(PCS_MIX_DIAG) && PCS_MIX_DIAG == 2
    int nev = 0, t_start = 0;  // diagnostics: evaluations of this lane's row, kernel-relative start time
    const long long t_kernel = clock64();
#endif
    for (;;) {
        // hand rows to the lanes that have none
        unsigned long long need = __ballot(L.done());
#if PCS_REFILL_MIN > 1
        if (__popcll(need) < PCS_REFILL_MIN) need = 0ull;  // wait until a few lanes are free: the refill code runs for the whole wave
#endif
        while (need != 0ull && !(drained && next >= end)) {
            if (next >= end) {
                int head = 0;
                if (lane == 0) head = atomicAdd(&ctrl[QCTRL_HEAD], QCHUNK);
                head = __builtin_amdgcn_readfirstlane(head);
                next = head;
                end = head + QCHUNK < total ? head + QCHUNK : total;
                if (head >= total) { drained = true; next = end = 0; break; }
            }
            const int avail = end - next;
            const int rank = __popcll(need & below);
            const bool take = L.done() && ((need >> lane) & 1ull) && rank < avail;
            if (take) {
                row = perm[next + rank];
                double par[16], k0, k1;
                load_mix_row(params, kij, row, par, k0, k1);
                T = temp[row];
                mix_coef<double>(m.c, par, k0, k1, T);
                L.start(m, z[row], p_init[row] / (T * P_UNIT));
                evals = 0;
#if defined(PCS_MIX_DIAG) && PCS_MIX_DIAG == 2
                nev = 0;
                t_start = (int)((clock64() - t_kernel) >> 14);
#endif
            }
            const int wanted = __popcll(need);
            next += wanted < avail ? wanted : avail;
            need = __ballot(L.done());
        }
        if (__ballot(!L.done()) == 0ull) break;  // nothing in flight and nothing left to take
        if (!L.done()) {
            double e0, e1;
            L.point(e0, e1);
            PhaseEval e = phase_eval(m, e0, e1);  // the only evaluation site
            L.consume(m, e);
            // evaluation budget: BD_EVAL_GUARD bounds the plain form, robust_eval_budget the second attempt (mix_solver_sm.hpp)
            if (++evals >= (L.robust ? robust_eval_budget<DEW>() : BD_EVAL_GUARD) && !L.done()) L.idle();  // rc = BD_FAILED
#if defined(PCS_MIX_DIAG) && PCS_MIX_DIAG == 2
            nev++;
            if (L.done()) L.out.iters = (nev & 4095) | ((t_start & 0xffff) << 12);
#endif
            if (PCS_INPLACE_ROBUST && L.done() && L.rc != BD_OK && !L.robust && L.root_failed) {
                // the plain form gave the row up at a liquid root: second attempt with bracketed liquid roots, in place (the
                // lane keeps the row and its coefficients; the other lanes of the wave go on with theirs).  Rows that fail in
                // the Newton with sound roots (98 % of the failing bubble rows: unstable liquids) are not repeated: the
                // robust form differs in the roots only, and every repeated row is a potential tail of the kernel
                L.start(m, z[row], p_init[row] / (T * P_UNIT), SS_MAX_IT, NEWTON_MAX_IT, true);
                evals = 0;
            }
            if (L.done()) mix_store<DEW>(row, L.rc, L.out, T, p_out, rho4, status, iters);
        }
    }
}

// PcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420): a, p, mu_i, v_i at given partial densities
__global__ __launch_bounds__(MBLOCK, 2) void k_mix_derivatives(const double* __restrict__ params,
                                                            const double* __restrict__ kij,
                                                            const double* __restrict__ temp,
                                                            const double* __restrict__ rho, int64_t n,
                                                            double* __restrict__ a, double* __restrict__ p,
                                                            double* __restrict__ mu, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * MBLOCK + threadIdx.x;
    if (i >= n) return;
    double par[16], k0, k1;
    load_mix_row(params, kij, i, par, k0, k1);
    MixModel m;
    mix_coef<double>(m.c, par, k0, k1, temp[i]);
    PhaseEval e = phase_eval(m, rho[2 * i], rho[2 * i + 1]);
    if (a) a[i] = e.a;
    if (p) p[i] = e.p();
    if (mu) { mu[2 * i] = e.g0; mu[2 * i + 1] = e.g1; }
    if (v) {
        double d0 = e.dp0(), d1 = e.dp1();
        double den = 1.0 / (e.r0 * d0 + e.r1 * d1);
        v[2 * i] = d0 * den;
        v[2 * i + 1] = d1 * den;
    }
}

// K6: gradient of the bubble / dew pressure at the converged densities
__global__ __launch_bounds__(MBLOCK) void k_mix_jacobian(int dew, const double* __restrict__ params,
                                                         const double* __restrict__ kij,
                                                         const double* __restrict__ temp,
                                                         const double* __restrict__ rho4, int64_t n,
                                                         double* __restrict__ jac, const int32_t* __restrict__ order) {
    // waves of non-associating / non-polar rows skip the structurally-zero directions altogether, so rows are taken
    // in class order: from the batch-wide permutation of k_mix_class_* when the caller provides a workspace, else
    // bucketed inside the workgroup as in k_mix_bubble_dew
    __shared__ int perm[MBLOCK];
    __shared__ int bins[MIX_BINS + 1];
#if PCS_MIX_ADJOINT
    __shared__ double adj_lds[ADJ_SLOTS * MBLOCK];  // coefficient adjoints of this lane's row: adj_lds[k * MBLOCK + t]
#endif
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * MBLOCK;
    int64_t i;
    if (order) {
        if (row0 + t >= n) return;
        i = order[row0 + t];
        if (i < 0 || i >= n) return;
    } else {
        if (t <= MIX_BINS) bins[t] = 0;
        __syncthreads();
        int key = MIX_BINS;
        if (row0 + t < n) key = mix_bucket(params + 16 * (row0 + t));
        atomicAdd(&bins[key], 1);
        __syncthreads();
        if (t == 0) {
            int acc = 0;
#pragma unroll
            for (int b = 0; b <= MIX_BINS; b++) {
                int c = bins[b];
                bins[b] = acc;
                acc += c;
            }
        }
        __syncthreads();
        perm[atomicAdd(&bins[key], 1)] = t;
        __syncthreads();
        i = row0 + perm[t];
        if (i >= n) return;
    }
    double par[16], k0, k1;
    load_mix_row(params, kij, i, par, k0, k1);
    double4 r = reinterpret_cast<const double4*>(rho4)[i];  // (V0, V1, L0, L1)
    double* g = jac + MIX_DIRS * i;
#if PCS_MIX_ADJOINT
    double* adj = adj_lds + t;
#else
    double* adj = nullptr;
#endif
    if (dew) mix_jacobian(par, k0, k1, temp[i], r.x, r.y, r.z, r.w, true, g, adj, MBLOCK);
    else mix_jacobian(par, k0, k1, temp[i], r.z, r.w, r.x, r.y, false, g, adj, MBLOCK);
}

// vector-Jacobian product of PcSaftMix.derivatives (autograd of derivatives / helmholtz_energy_density)
__global__ __launch_bounds__(64) void k_mix_derivatives_vjp(const double* __restrict__ params, const double* __restrict__ kij,
                                                            const double* __restrict__ temp, const double* __restrict__ rho,
                                                            int64_t n, const double* __restrict__ g_a,
                                                            const double* __restrict__ g_p, const double* __restrict__ g_mu,
                                                            const double* __restrict__ g_v, double* __restrict__ grad,
                                                            const int32_t* __restrict__ order) {
    int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (order) {  // class order (k_mix_class_*): class-uniform waves skip the structurally-zero directions
        i = order[i];
        if (i < 0 || i >= n) return;
    }
    double par[16], k0, k1;
    load_mix_row(params, kij, i, par, k0, k1);
    mix_derivatives_vjp(par, k0, k1, temp[i], rho[2 * i], rho[2 * i + 1], g_a ? g_a[i] : 0.0, g_p ? g_p[i] : 0.0,
                        g_mu ? g_mu[2 * i] : 0.0, g_mu ? g_mu[2 * i + 1] : 0.0, g_v ? g_v[2 * i] : 0.0, g_v ? g_v[2 * i + 1] : 0.0,
                        grad + MIX_VJP_DIRS * i);
}

// resident waves of the queue kernel: one per SIMD (the evaluation needs the whole register file)
int queue_waves() {
    static int waves = 0;
    if (!waves) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        waves = cus * 4 * PCS_QUEUE_WAVES_PER_SIMD;
    }
    return waves;
}

}  // namespace

extern "C" {

int pcs_mix_bubble_dew(int dew, const double* params, const double* kij, const double* temp, const double* z,
                       const double* p_init, int64_t n, double* p_out, double* rho4, uint8_t* status, int32_t* iters,
                       void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !kij || !temp || !z || !p_init || !status) return fail_msg("pcs_mix_bubble_dew: null required pointer");
    const unsigned grid = (unsigned)((n + MBLOCK - 1) / MBLOCK);
    hipStream_t s = as_stream(stream);
#if PCS_MIX_QUEUE
    if (workspace) {
        // work-queue schedule: perm[n] + control block in the workspace
        int32_t* perm = static_cast<int32_t*>(workspace);
        int32_t* ctrl = perm + n;
        if (int ez = zero_ints(ctrl, QCTRL_INTS, s)) return ez;
        const unsigned g256 = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_mix_class_count, dim3(g256), dim3(256), 0, s, params, n, ctrl);
        hipLaunchKernelGGL(k_mix_class_scan, dim3(1), dim3(64), 0, s, ctrl);
        hipLaunchKernelGGL(k_mix_class_scatter, dim3(g256), dim3(256), 0, s, params, n, ctrl, perm);
        hipError_t e;
        unsigned waves = (unsigned)queue_waves();
        const unsigned needed = (unsigned)((n + 63) / 64);
        if (waves > needed) waves = needed;
        if (dew)
            hipLaunchKernelGGL(k_mix_bubble_dew_queue<true>, dim3(waves), dim3(64), 0, s, params, kij, temp, z, p_init, n,
                               (const int32_t*)perm, ctrl, p_out, rho4, status, iters);
        else
            hipLaunchKernelGGL(k_mix_bubble_dew_queue<false>, dim3(waves), dim3(64), 0, s, params, kij, temp, z, p_init, n,
                               (const int32_t*)perm, ctrl, p_out, rho4, status, iters);
        e = hipGetLastError();
        if (e != hipSuccess) return fail("k_mix_bubble_dew_queue launch", e);
        return 0;
    }
#endif
    int32_t* retry = static_cast<int32_t*>(workspace);
    if (retry) {
        if (int ez = zero_ints(retry, 1, s)) return ez;
    }
    if (dew) {
        hipLaunchKernelGGL(k_mix_bubble_dew<true>, dim3(grid), dim3(MBLOCK), 0, s, params, kij, temp, z, p_init, n, p_out,
                           rho4, status, iters, retry);
        if (retry)
            hipLaunchKernelGGL(k_mix_bubble_dew_retry<true>, dim3(MIX_RETRY_GRID), dim3(64), 0, s, params, kij, temp, z, p_init,
                               p_out, rho4, status, iters, (const int32_t*)retry, n);
    } else {
        hipLaunchKernelGGL(k_mix_bubble_dew<false>, dim3(grid), dim3(MBLOCK), 0, s, params, kij, temp, z, p_init, n, p_out,
                           rho4, status, iters, retry);
        if (retry)
            hipLaunchKernelGGL(k_mix_bubble_dew_retry<false>, dim3(MIX_RETRY_GRID), dim3(64), 0, s, params, kij, temp, z, p_init,
                               p_out, rho4, status, iters, (const int32_t*)retry, n);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_mix_bubble_dew launch", e);
    return 0;
}

int pcs_mix_derivatives(const double* params, const double* kij, const double* temp, const double* rho, int64_t n,
                        double* a, double* p, double* mu, double* v, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !kij || !temp || !rho) return fail_msg("pcs_mix_derivatives: null required pointer");
    const unsigned grid = (unsigned)((n + MBLOCK - 1) / MBLOCK);
    hipLaunchKernelGGL(k_mix_derivatives, dim3(grid), dim3(MBLOCK), 0, as_stream(stream), params, kij,
                       temp, rho, n, a, p, mu, v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_mix_derivatives launch", e);
    return 0;
}

int pcs_mix_derivatives_vjp(const double* params, const double* kij, const double* temp, const double* rho, int64_t n,
                            const double* g_a, const double* g_p, const double* g_mu, const double* g_v, double* grad,
                            void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !kij || !temp || !rho || !grad) return fail_msg("pcs_mix_derivatives_vjp: null required pointer");
    hipStream_t s = as_stream(stream);
    const int32_t* order = nullptr;
    if (workspace) {
        int32_t* perm = static_cast<int32_t*>(workspace);
        int32_t* ctrl = perm + n;
        if (int ez = zero_ints(ctrl, QCTRL_INTS, s)) return ez;
        const unsigned g256 = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_mix_class_count, dim3(g256), dim3(256), 0, s, params, n, ctrl);
        hipLaunchKernelGGL(k_mix_class_scan, dim3(1), dim3(64), 0, s, ctrl);
        hipLaunchKernelGGL(k_mix_class_scatter, dim3(g256), dim3(256), 0, s, params, n, ctrl, perm);
        order = perm;
    }
    hipLaunchKernelGGL(k_mix_derivatives_vjp, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, params, kij, temp, rho, n, g_a,
                       g_p, g_mu, g_v, grad, order);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_mix_derivatives_vjp launch", e);
    return 0;
}

int pcs_mix_jacobian(int dew, const double* params, const double* kij, const double* temp, const double* rho4,
                     int64_t n, double* jac, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !kij || !temp || !rho4 || !jac) return fail_msg("pcs_mix_jacobian: null required pointer");
    const unsigned grid = (unsigned)((n + MBLOCK - 1) / MBLOCK);
    hipStream_t s = as_stream(stream);
    const int32_t* order = nullptr;
    if (workspace) {  // batch-wide class order (the permutation of the work-queue schedule)
        int32_t* perm = static_cast<int32_t*>(workspace);
        int32_t* ctrl = perm + n;
        if (int ez = zero_ints(ctrl, QCTRL_INTS, s)) return ez;
        const unsigned g256 = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_mix_class_count, dim3(g256), dim3(256), 0, s, params, n, ctrl);
        hipLaunchKernelGGL(k_mix_class_scan, dim3(1), dim3(64), 0, s, ctrl);
        hipLaunchKernelGGL(k_mix_class_scatter, dim3(g256), dim3(256), 0, s, params, n, ctrl, perm);
        order = perm;
    }
    hipLaunchKernelGGL(k_mix_jacobian, dim3(grid), dim3(MBLOCK), 0, s, dew, params, kij, temp, rho4, n, jac, order);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_mix_jacobian launch", e);
    return 0;
}

}  // extern "C"
