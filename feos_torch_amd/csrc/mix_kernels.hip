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
#include "mix_jacobian.hpp"

using namespace pcs;
using namespace pcs_abi;

namespace {

constexpr int MBLOCK = 128;

struct MixModel {
    MixCoef<double> c;
    template <class R> PCS_DEV R a(const R& r0, const R& r1) const { return mix_a<double, R>(c, r0, r1); }
    template <class R, class Z> PCS_DEV R a_z(const R& r0, const R& r1, const Z& zeta3) const { return mix_a_z<double, R, Z>(c, r0, r1, zeta3); }
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

// lanes that must be idle before a queue wave refills: the refill code (row load, coefficient set-up) runs for the whole
// wave (A/B dew 1e6 rows, round 1, one kernel: 1: 7.9 ms, 4: 7.7, 8: 7.5, 16: 7.8; round 3, two kernels: 8: 4.93, 12: 4.84, 16: 4.83, 24: 4.89)
constexpr int PCS_REFILL_MIN = 16;
constexpr int REFILL_MIN = PCS_REFILL_MIN;

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
    if (iters) iters[i] = ok ? r.iters : -1;
    status[i] = ok ? 0 : 1;
}

// K5 without a workspace: one row per lane in a single pass with the full iteration caps, the robust second attempt in
// place.  Same arithmetic as the work-queue schedule below (tests/test_mix_missed_gpu.py), several times slower: a wave
// pays its slowest row.
template <bool DEW>
__global__ __launch_bounds__(MBLOCK, 1) void k_mix_bubble_dew(const double* __restrict__ params,
                                                           const double* __restrict__ kij,
                                                           const double* __restrict__ temp,
                                                           const double* __restrict__ z,
                                                           const double* __restrict__ p_init, int64_t n,
                                                           double* __restrict__ p_out, double* __restrict__ rho4,
                                                           uint8_t* __restrict__ status, int32_t* __restrict__ iters) {
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
    bool root_failed = false;
    int rc = bubble_dew_solve_sm<DEW>(m, z[i], p_red, r, SS_MAX_IT, NEWTON_MAX_IT, false, &root_failed);
    if (rc != BD_OK && root_failed) rc = bubble_dew_solve_sm<DEW>(m, z[i], p_red, r, SS_MAX_IT, NEWTON_MAX_IT, true);
    mix_store<DEW>(i, rc, r, T, p_out, rho4, status, iters);
}

// ------------------------------------------------------------------------------------------
// K5 as a work queue.  The iteration counts of the rows differ by an order of magnitude (5 ... 400
// evaluations), so with one row per lane a wave idles most of its lanes most of the time.  Here the
// rows are first ordered by class (device counting sort -> perm, expensive classes first so their long
// rows overlap with the bulk), then a fixed set of resident waves works through that order: a lane that
// finishes its row stores it and takes the next one, every pass of the wave's loop is one evaluation
// for all its lanes.
//
// Round 3: the solve is cut in two queue kernels at the point where the Newton iteration starts.  Every pass of a
// queue wave executes the evaluation (1,600-3,500 instructions by class) AND the union of the solver code of all the
// stages its lanes are in -- liquid-root logic, the substitution sweep (4 exp, 3 log, 8 divisions), the Newton step
// (3x3 solve, 3 exp, 4 log): with all stages in one kernel that union was nearly as long as the evaluation itself.
//   k_mix_init_queue   the plain form's initialisation only (liquid roots, Raoult + successive substitution; BdLane in
//                      INIT mode): ends a row where the Newton would start and writes (rho_spec, rho_inc_1, rho_inc_2)
//                      + a code to the workspace; rows whose plain initialisation fails at a liquid root are appended to
//                      the `robust` list, rows that fail otherwise are final;
//   k_mix_bubble_dew_queue  first the robust list (second attempt from scratch, bracketed liquid roots: the long rows
//                      start first and overlap with the bulk), then every other row straight into the Newton stage.
// The arithmetic of a row is the state machine's, stage by stage, as before: same results.
// control block (int32, after perm[n] in the workspace): [0..8] class counts -> offsets, [16] / [17] queue heads of the
// two kernels, [18] length of the robust list
// ------------------------------------------------------------------------------------------
constexpr int QCTRL_HEAD = 16, QCTRL_HEAD2 = 17, QCTRL_NROB = 18, QCTRL_INTS = 64;
constexpr int QCHUNK = 64;  // rows a wave reserves per atomic on the queue head
enum : int { INIT_OK = 0, INIT_OK_ROOT_FAILED = 1, INIT_FAILED = 2, INIT_ROBUST = 3 };  // code of a row's init record

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

// ------------------------------------------------------------------------------------------
// K5 pre-pass (dew points): Raoult's law needs the zero-pressure liquid fugacity of both PURE components.  The solver's
// state machine finds them with the full mixture evaluation at x = (1, 0) and (0, 1) -- 8-9 of a dew row's ~20 T2
// evaluations, on the one-wave-per-SIMD queue kernel.  At those compositions the mixture model IS the pure-component
// model (BMCSL -> Carnahan-Starling, g_ii -> (1 - eta/2)/(1 - eta)^3, one-fluid dispersion -> the pure sums, the dipole
// quotient's pure limit, association of the component's own sites only), so the same Newton iteration -- same start,
// same scaled function, same acceptance rule, same first-order carry of the chemical potential to the root as
// BdLane::consume (S_ROOT, plain form) -- runs here on the closed-form pure evaluation (pure_a<double, D2>, ~350
// instructions instead of 1,600-3,500), one (row, component) per lane, four waves per SIMD.  Results agree with the
// state machine's to rounding; a lane the plain form would give up on writes NaN and the row takes the state machine's own
// route (including its robust second attempt), so no decision of the solver changes.
// fug[4 k + c] = rho_L exp(mu_res) of component c of the row at queue position k at p = 0, or NaN; fug[4 k + 2 + c] = rho_L.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mix_pure_fugacity(const double* __restrict__ params, const double* __restrict__ temp,
                                                           int64_t n, const int32_t* __restrict__ perm,
                                                           double* __restrict__ fug) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < 2 * n;
    const int64_t pos = live ? (t >> 1) : 0;
    const int comp = (int)(t & 1);
    int64_t row = perm ? perm[pos] : pos;
    if (row < 0 || row >= n) row = 0;
    double par[8], oth[8];
    {
        const double2* src = reinterpret_cast<const double2*>(params + 16 * row + 8 * comp);
        const double2* so = reinterpret_cast<const double2*>(params + 16 * row + 8 * (1 - comp));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double2 v = src[k], w = so[k];
            par[2 * k] = v.x; par[2 * k + 1] = v.y;
            oth[2 * k] = w.x; oth[2 * k + 1] = w.y;
        }
    }
    const double T = temp[row];
    // the component's own association in the pure limit: only a self-associating component associates with itself.  In the
    // class with ONE associating component the reference sums kappa and eps over both components (pcsaft_mix.py:211-218):
    // the pure model is that limit only if the site-less partner carries none
    const bool self_i = par[6] * par[7] != 0.0;
    const bool other_sites = oth[6] + oth[7] != 0.0;
    bool usable = !(self_i && !other_sites && (oth[4] != 0.0 || oth[5] != 0.0));
    if (!self_i) { par[6] = 0.0; par[7] = 0.0; }
    PureCoef<double> c;
    pure_coef<double>(c, par, T, false);
    const double pk = c.ceta;
    double rho = 0.5 / pk, err_prev = 1.0, f = __longlong_as_double(0x7ff8000000000000LL), rho_root = 0.0;
    bool dense = false, active = live && usable;
    int it = 0;
    for (int guard = 0; guard < LIQ_ROOT_MAX_IT + 2; guard++) {
        if (active) {
            const D2<double> a = pure_a<double, D2<double>>(c, D2<double>(rho, 1.0, 0.0));
            const double p = rho - a.v + rho * a.d1, dp = 1.0 + rho * a.d2;
            if (it == 0 && !dense && !(p > 0.0)) {
                rho = 0.62 / pk;  // very cold / dense: restart on the dense side (BdLane::consume)
                dense = true;
            } else {
                const double den = dense ? dp : dp - 4.0 * p * pk / (1.0 - rho * pk);
                bool bad = !(dp > 0.0) || !(den > 0.0) || !is_finite_bits(p);
                const double step = p / den, rho_new = rho - step;
                bad = bad || !(rho_new > 0.0) || !is_finite_bits(rho_new);
                bad = bad || (dense && !(rho_new * pk > 0.5 && rho_new * pk < 0.62));
                bool done = false;
                if (!bad) {
                    const double err = fabs(step) / rho;
                    done = err <= LIQ_ROOT_TOL || (it >= 3 && err < 1e-7 && err >= 0.25 * err_prev);
                    err_prev = err;
                    it++;
                    if (!done && it >= LIQ_ROOT_MAX_IT) bad = true;
                }
                if (bad) {
                    active = false;  // NaN: the state machine decides (robust second attempt)
                } else if (done) {
                    f = rho_new * exp(a.d1 - a.d2 * step);  // chemical potential carried to the root to first order
                    rho_root = rho_new;
                    active = false;
                } else {
                    rho = rho_new;
                }
            }
        }
        if (__ballot(active) == 0ull) break;
    }
    if (live) {
        fug[4 * pos + comp] = f;
        fug[4 * pos + 2 + comp] = rho_root;
    }
}

// One wave's slice of a queue [0, total): hands `rank`-th idle lane the queue position next + rank.  Wave-uniform state.
struct QueueSlice {
    int next = 0, end = 0;
    bool drained = false;
    // reserves a new chunk when the slice is empty; false when the queue is exhausted
    __device__ __forceinline__ bool refill(int32_t* __restrict__ head, int total) {
        if (next < end) return true;
        if (drained) return false;
        int h = 0;
        if ((threadIdx.x & 63) == 0) h = atomicAdd(head, QCHUNK);
        h = __builtin_amdgcn_readfirstlane(h);
        if (h >= total) { drained = true; next = end = 0; return false; }
        next = h;
        end = h + QCHUNK < total ? h + QCHUNK : total;
        return true;
    }
};

// the per-lane solver state of a queue kernel lives in LDS (24 KB per wave, padded to an odd number of doubles per lane):
// 1,613 -> 453 AGPR moves in the kernel body, -5 % / -3 % kernel time (round 2)
template <class Lane>
struct alignas(8) LaneSlot {
    Lane L;
    double pad[(sizeof(Lane) / 8) % 2 == 0 ? 1 : 2];
};

// What a queue lane loads when it takes a row.  (Round 3 also tried to set the coefficients up ONCE per row in a separate
// kernel and to hand the queue lanes a 544-byte record: the refill then waits for 34 scattered 16-byte loads with nothing to
// hide the latency behind -- one wave per SIMD -- and the solve got slower, bubble 2.6 -> 3.1 ms, dew 5.2 -> 5.5 ms per 1e6
// rows; with the evaluation reading the record in place, uncoalesced, 3.6 / 5.9 ms.  The set-up stays in the refill.)
struct RowScalars {
    int64_t row;
    double T, z, p_red;
};
__device__ __forceinline__ RowScalars take_row(const double* __restrict__ params, const double* __restrict__ kij,
                                               const double* __restrict__ temp, const double* __restrict__ z,
                                               const double* __restrict__ p_init, int64_t row, MixModel& m) {
    double par[16], k0, k1;
    load_mix_row(params, kij, row, par, k0, k1);
    RowScalars r;
    r.row = row;
    r.T = temp[row];
    r.z = z[row];
    r.p_red = p_init[row] / (r.T * P_UNIT);
    mix_coef<double>(m.c, par, k0, k1, r.T);
    return r;
}

// K5a: initialisation of the plain form (see above).  init[k] = (rho_spec, rho_inc_1, rho_inc_2, code) of queue position k
template <bool DEW>
__global__ __launch_bounds__(64, 1) void k_mix_init_queue(const double* __restrict__ params, const double* __restrict__ kij,
                                                       const double* __restrict__ temp, const double* __restrict__ z,
                                                       const double* __restrict__ p_init, int64_t n,
                                                       const int32_t* __restrict__ perm, int32_t* __restrict__ ctrl,
                                                       const double* __restrict__ fug, double4* __restrict__ init,
                                                       int32_t* __restrict__ robust_list) {
    typedef BdLane<DEW, BD_MODE_INIT> Lane;
    __shared__ LaneSlot<Lane> lane_slots[64];
    Lane& L = lane_slots[threadIdx.x].L;
    const int lane = threadIdx.x;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int total = (int)n;
    QueueSlice q;
    L.idle();
    MixModel m;
    int pos = 0;
    int evals = 0;  // evaluations of this lane's current row (bounded by BD_EVAL_GUARD: the loop below cannot hang)
    for (;;) {
        unsigned long long need = __ballot(L.done());
        if (__popcll(need) < REFILL_MIN) need = 0ull;
        while (need != 0ull && q.refill(&ctrl[QCTRL_HEAD], total)) {
            const int avail = q.end - q.next;
            const int rank = __popcll(need & below);
            if (L.done() && ((need >> lane) & 1ull) && rank < avail) {
                pos = q.next + rank;
                const RowScalars r = take_row(params, kij, temp, z, p_init, perm[pos], m);
                double4 ff = make_double4(-1.0, -1.0, 0.0, 0.0);
                if (DEW && fug) ff = reinterpret_cast<const double4*>(fug)[pos];  // k_mix_pure_fugacity; NaN = not pre-solved
                L.start(m, r.z, r.p_red, SS_MAX_IT, NEWTON_MAX_IT, false, ff.x, ff.y, ff.z, ff.w);
                evals = 0;
            }
            const int wanted = __popcll(need);
            q.next += wanted < avail ? wanted : avail;
            need = __ballot(L.done());
        }
        if (__ballot(!L.done()) == 0ull) break;  // nothing in flight and nothing left to take
        if (!L.done()) {
            double e0, e1;
            L.point(e0, e1);
            PhaseEval e = phase_eval_inline(m, e0, e1);  // the only evaluation site of the kernel, inlined (see mix_solver.hpp)
            L.consume(m, e);
            if (++evals >= BD_EVAL_GUARD && !L.done()) L.idle();  // rc = BD_FAILED
            if (L.done()) {
                int code = INIT_FAILED;
                if (L.rc == BD_HANDOVER) code = L.root_failed ? INIT_OK_ROOT_FAILED : INIT_OK;
                else if (L.root_failed) {
                    // the plain form gave the row up at a liquid root: second attempt with bracketed liquid roots, from
                    // scratch, at the head of the second kernel's queue.  Rows that fail with sound roots are not repeated
                    // (the robust form differs in the roots only)
                    code = INIT_ROBUST;
                    const int slot = atomicAdd(&ctrl[QCTRL_NROB], 1);
                    if (slot >= 0 && slot < total) robust_list[slot] = pos;
                }
                init[pos] = make_double4(L.out.spec0, L.out.inc0, L.out.inc1, (double)code);
            }
        }
    }
}

// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]; every lane of the wave must execute it
__device__ __forceinline__ int nb_int(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ double nb_double(double x) { return __hiloint2double(nb_int(__double2hiint(x)), nb_int(__double2loint(x))); }
// the neighbour's coefficient set, taken over by a lane that has no row of its own any more (tail of the Newton queue, below)
__device__ __forceinline__ void adopt_neighbour_model(MixModel& m, bool take) {
    MixCoef<double>& c = m.c;
#define PCS_ADOPT(x) { const double t_ = nb_double(x); if (take) x = t_; }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        PCS_ADOPT(c.m[i]) PCS_ADOPT(c.mm1[i]) PCS_ADOPT(c.d[i]) PCS_ADOPT(c.na[i]) PCS_ADOPT(c.nb[i])
#pragma unroll
        for (int k = 0; k < 4; k++) PCS_ADOPT(c.zk[k][i])
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        PCS_ADOPT(c.A[k]) PCS_ADOPT(c.B[k]) PCS_ADOPT(c.dij[k]) PCS_ADOPT(c.S[k])
#pragma unroll
        for (int n = 0; n < 5; n++) PCS_ADOPT(c.pj[k][n])
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int n = 0; n < 4; n++) PCS_ADOPT(c.tj[k][n])
#undef PCS_ADOPT
    const int pol = nb_int(c.polar ? 1 : 0), acl = nb_int(c.acls);
    if (take) { c.polar = pol != 0; c.acls = acl; }
}

// K5b: robust list first, then the Newton iteration of every initialised row.
// Tail: once the queue is drained the kernel ends with its longest rows -- a failing dew row runs ~20 plain + up to 24 damped
// Newton iterations of two evaluations each, ~1 ms, while most lanes have nothing left to do (measured with all caps cut short:
// the tails are 20 % of the bubble and 30 % of the dew kernel).  From then on a lane without a row takes over the
// incipient-phase evaluation of its neighbour (lane ^ 1) whenever that one starts a Newton iteration: it adopts the neighbour's
// coefficient set, evaluates at (rho_inc_1, rho_inc_2) in the same pass in which the neighbour evaluates the specified phase,
// and hands the result over -- one pass per Newton iteration instead of two.  Same function, same inputs: identical results.
template <bool DEW>
__global__ __launch_bounds__(64, 1) void k_mix_bubble_dew_queue(const double* __restrict__ params,
                                                             const double* __restrict__ kij,
                                                             const double* __restrict__ temp,
                                                             const double* __restrict__ z,
                                                             const double* __restrict__ p_init, int64_t n,
                                                             const int32_t* __restrict__ perm, int32_t* __restrict__ ctrl,
                                                             const double4* __restrict__ init,
                                                             const int32_t* __restrict__ robust_list,
                                                             double* __restrict__ p_out, double* __restrict__ rho4,
                                                             uint8_t* __restrict__ status, int32_t* __restrict__ iters) {
    typedef BdLane<DEW, BD_MODE_FULL> Lane;
    __shared__ LaneSlot<Lane> lane_slots[64];
    Lane& L = lane_slots[threadIdx.x].L;
    const int lane = threadIdx.x;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int n_rob = min(max(ctrl[QCTRL_NROB], 0), (int)n);
    const int total = (int)n + n_rob;
    QueueSlice q;
    L.idle();
    MixModel m;
    RowScalars r;
    r.row = 0; r.T = r.z = r.p_red = 0.0;
    int evals = 0;
    int64_t adopted_row = -1;  // the row whose coefficient set this lane has taken over from its neighbour (tail)
    for (;;) {
        // hand rows to the lanes that have none
        unsigned long long need = __ballot(L.done());
        if (__popcll(need) < REFILL_MIN) need = 0ull;  // wait until a few lanes are free: the refill code runs for the whole wave
        while (need != 0ull && q.refill(&ctrl[QCTRL_HEAD2], total)) {
            const int avail = q.end - q.next;
            const int rank = __popcll(need & below);
            if (L.done() && ((need >> lane) & 1ull) && rank < avail) {
                const int qpos = q.next + rank;
                int pos, code;
                if (qpos < n_rob) {
                    pos = robust_list[qpos];
                    code = (pos >= 0 && pos < (int)n) ? INIT_ROBUST : -1;
                } else {
                    pos = qpos - n_rob;
                    code = (int)init[pos].w;
                    if (code == INIT_ROBUST) code = -1;  // taken from the robust list
                }
                if (code == INIT_FAILED) {
                    const int64_t row = perm[pos];
                    L.idle();  // rc = BD_FAILED, final
                    mix_store<DEW>(row, BD_FAILED, L.out, temp[row], p_out, rho4, status, iters);
                } else if (code >= 0) {
                    r = take_row(params, kij, temp, z, p_init, perm[pos], m);
                    if (code == INIT_ROBUST) {
                        L.start(m, r.z, r.p_red, SS_MAX_IT, NEWTON_MAX_IT, true);
                    } else {
                        const double4 s0 = init[pos];
                        L.start_newton(r.z, r.p_red, s0.x, s0.y, s0.z, code == INIT_OK_ROOT_FAILED);
                    }
                    evals = 0;
                }
            }
            const int wanted = __popcll(need);
            q.next += wanted < avail ? wanted : avail;
            need = __ballot(L.done());
        }
        if (__ballot(!L.done()) == 0ull) break;  // nothing in flight and nothing left to take
        const bool act = !L.done();
        double e0 = 0.0, e1 = 0.0;
        if (act) L.point(e0, e1);
        bool help = false, helped = false;
        if (q.drained) {  // wave-uniform: the tail (see above)
            const bool starts_newton = act && L.stage == Lane::S_NEWTON_S;
            // (the exchanges are executed by every lane: no short-circuit in front of them)
            const int nb_starts = nb_int(starts_newton ? 1 : 0), nb_act = nb_int(act ? 1 : 0);
            help = !act && nb_starts != 0;        // my neighbour starts a Newton iteration and I have no row
            helped = starts_newton && nb_act == 0;  // ... and the other way round
            const double h0 = nb_double(L.ri0), h1 = nb_double(L.ri1);
            const int64_t nb_row = ((int64_t)nb_int((int)(r.row >> 32)) << 32) | (uint32_t)nb_int((int)r.row);
            const bool adopt = help && nb_row != adopted_row;
            if (__ballot(adopt) != 0ull) {
                adopt_neighbour_model(m, adopt);
                if (adopt) adopted_row = nb_row;
            }
            if (help) { e0 = h0; e1 = h1; }
        }
        PhaseEval e;
        e.r0 = e.r1 = e.a = e.g0 = e.g1 = e.h00 = e.h01 = e.h11 = 0.0;
        if (act || help) e = phase_eval_inline(m, e0, e1);  // the only evaluation site of the kernel, inlined (see mix_solver.hpp)
        PhaseEval en = e;
        if (q.drained) {
            en.r0 = nb_double(e.r0); en.r1 = nb_double(e.r1); en.a = nb_double(e.a); en.g0 = nb_double(e.g0); en.g1 = nb_double(e.g1);
            en.h00 = nb_double(e.h00); en.h01 = nb_double(e.h01); en.h11 = nb_double(e.h11);
        }
        if (act) {
            L.consume(m, e);
            if (helped && !L.done()) {  // (stage is S_NEWTON_N now) the neighbour's evaluation of the incipient phase completes the iteration
                L.consume(m, en);
                evals++;
            }
            // evaluation budget: BD_EVAL_GUARD bounds the plain form, robust_eval_budget the second attempt (mix_solver_sm.hpp)
            if (++evals >= (L.robust ? robust_eval_budget<DEW>() : BD_EVAL_GUARD) && !L.done()) L.idle();  // rc = BD_FAILED
            if (L.done() && L.rc != BD_OK && !L.robust && L.root_failed) {
                // a liquid root of the plain form had failed (the row went on with its second choice of the pressure) and the
                // Newton gave the row up: second attempt with bracketed liquid roots, in place (the lane keeps the row and its
                // coefficients; the other lanes of the wave go on with theirs)
                L.start(m, r.z, r.p_red, SS_MAX_IT, NEWTON_MAX_IT, true);
                evals = 0;
            }
            if (L.done()) mix_store<DEW>(r.row, L.rc, L.out, r.T, p_out, rho4, status, iters);
        }
    }
}

// PcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420): a, p, mu_i, v_i at given partial densities
__global__ __launch_bounds__(MBLOCK, 1) void k_mix_derivatives(const double* __restrict__ params,
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
    __shared__ double adj_lds[ADJ_SLOTS * MBLOCK];  // coefficient adjoints of this lane's row: adj_lds[k * MBLOCK + t]
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
    double* adj = adj_lds + t;
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

// byte offset / pointer of the pre-pass area behind perm[n] + control block in the mixture workspace
int64_t mix_ws_offset(int64_t n) { return (((int64_t)sizeof(int32_t) * ((n > 0 ? n : 0) + QCTRL_INTS)) + 15) & ~(int64_t)15; }

// resident waves of the queue kernel: one per SIMD (the evaluation needs the whole register file)
int queue_waves() {
    static int waves = 0;
    if (!waves) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        waves = cus * 4;
    }
    return waves;
}

}  // namespace

extern "C" {

// workspace of pcs_mix_bubble_dew: row order perm[n] + control block (as pcs_workspace_bytes), then -- 16-byte aligned --
// the pre-pass fugacities and densities (4 doubles per row), the init records (4 doubles per row) and the robust list (1 int
// per row)
int64_t pcs_mix_workspace_bytes(int64_t n) {
    const int64_t m = n > 0 ? n : 0;
    return mix_ws_offset(n) + (int64_t)sizeof(double) * 8 * m + (int64_t)sizeof(int32_t) * m;
}

int pcs_mix_bubble_dew(int dew, const double* params, const double* kij, const double* temp, const double* z,
                       const double* p_init, int64_t n, double* p_out, double* rho4, uint8_t* status, int32_t* iters,
                       void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (n == 0) return 0;
    if (!params || !kij || !temp || !z || !p_init || !status) return fail_msg("pcs_mix_bubble_dew: null required pointer");
    hipStream_t s = as_stream(stream);
    if (workspace) {
        // work-queue schedule: perm[n] + control block, pre-pass fugacities, init records and the robust list in the workspace
        int32_t* perm = static_cast<int32_t*>(workspace);
        int32_t* ctrl = perm + n;
        double* fug = reinterpret_cast<double*>(static_cast<char*>(workspace) + mix_ws_offset(n));
        double4* init = reinterpret_cast<double4*>(fug + 4 * n);
        int32_t* robust_list = reinterpret_cast<int32_t*>(fug + 8 * n);
        if (int ez = zero_ints(ctrl, QCTRL_INTS, s)) return ez;
        const unsigned g256 = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_mix_class_count, dim3(g256), dim3(256), 0, s, params, n, ctrl);
        hipLaunchKernelGGL(k_mix_class_scan, dim3(1), dim3(64), 0, s, ctrl);
        hipLaunchKernelGGL(k_mix_class_scatter, dim3(g256), dim3(256), 0, s, params, n, ctrl, perm);
        if (dew)
            hipLaunchKernelGGL(k_mix_pure_fugacity, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, s, params, temp, n,
                               (const int32_t*)perm, fug);
        unsigned waves = (unsigned)queue_waves();
        const unsigned needed = (unsigned)((n + 63) / 64);
        if (waves > needed) waves = needed;
        if (dew) {
            hipLaunchKernelGGL(k_mix_init_queue<true>, dim3(waves), dim3(64), 0, s, params, kij, temp, z, p_init, n,
                               (const int32_t*)perm, ctrl, (const double*)fug, init, robust_list);
            hipLaunchKernelGGL(k_mix_bubble_dew_queue<true>, dim3(waves), dim3(64), 0, s, params, kij, temp, z, p_init, n,
                               (const int32_t*)perm, ctrl, (const double4*)init, (const int32_t*)robust_list, p_out, rho4, status,
                               iters);
        } else {
            hipLaunchKernelGGL(k_mix_init_queue<false>, dim3(waves), dim3(64), 0, s, params, kij, temp, z, p_init, n,
                               (const int32_t*)perm, ctrl, (const double*)nullptr, init, robust_list);
            hipLaunchKernelGGL(k_mix_bubble_dew_queue<false>, dim3(waves), dim3(64), 0, s, params, kij, temp, z, p_init, n,
                               (const int32_t*)perm, ctrl, (const double4*)init, (const int32_t*)robust_list, p_out, rho4, status,
                               iters);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail("k_mix_bubble_dew_queue launch", e);
        return 0;
    }
    const unsigned grid = (unsigned)((n + MBLOCK - 1) / MBLOCK);
    if (dew)
        hipLaunchKernelGGL(k_mix_bubble_dew<true>, dim3(grid), dim3(MBLOCK), 0, s, params, kij, temp, z, p_init, n, p_out, rho4,
                           status, iters);
    else
        hipLaunchKernelGGL(k_mix_bubble_dew<false>, dim3(grid), dim3(MBLOCK), 0, s, params, kij, temp, z, p_init, n, p_out, rho4,
                           status, iters);
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
