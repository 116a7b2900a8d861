// gfx950 kernels + C ABI for the heterosegmented gc-PC-SAFT path (GcPcSaftMix bubble / dew
// points and derivatives).  One state point per lane.  The per-batch segment / pair tables are
// staged into LDS by every workgroup; each lane keeps its bond list (d_ab, count) in a
// lane-strided LDS scratch area (conflict-free) so the hard-chain loop needs no registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"
#include "gc_model.hpp"
#include "gc_kernel_common.hpp"
#include "mix_solver.hpp"
#include "mix_solver_sm.hpp"

using namespace pcs;
using namespace pcs_abi;

namespace {

constexpr int PCS_GBLOCK = 128;
constexpr int GBLOCK = PCS_GBLOCK;
// Jacobian kernel: 59 doubles of LDS per thread (bond diameters of the double and the dual model, row bytes); a 256-thread workgroup
// shares ONE copy of the table among four waves (S = 22: 131 KB, S = 32: 144 KB of the CU's 160 KB) -- 64-thread workgroups fit three
// times (one wave each): the kernel holds a full register file per wave, so resident waves per CU are what counts
constexpr int GJBLOCK = 256;

constexpr int PCS_GC_FAST_SS = 6;  // A/B on the synthetic dew batch (scripts/dev/ab_gc.py): 12/12: 9.0 ms, 6/8: 7.9, 4/8: 8.4, 7/7: 8.6
constexpr int PCS_GC_FAST_NEWTON = 8;
constexpr int GC_FAST_SS = PCS_GC_FAST_SS, GC_FAST_NEWTON = PCS_GC_FAST_NEWTON;  // fast-pass caps (mix_solver.hpp)
constexpr int GC_RETRY_BLOCKS = 1024;
// no-progress leash of the dew-point Newton here (mixture kernels: 20, followed by their damped run): the second pass ends with its
// slowest row; 10 loses no row of the synthetic batch and shortens it: dew 3.70 -> 3.60 ms per 1e6 rows
constexpr int GC_NO_PROGRESS_DEW = 10;

// association class x polarity of a row (the evaluation's branches): key of the workgroup bucketing in the fast pass
constexpr int GC_BINS = 8;
__device__ __forceinline__ int gc_bucket(const unsigned char* __restrict__ row, const GcTable& tb) {
    int associating = 0, self_assoc = 0, polar = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        double ka = 0.0, eab = 0.0, na = 0.0, nb = 0.0, mu2 = 0.0;
#pragma unroll 1
        for (int e = 0; e < GC_MAXE; e++) {
            const int n = row[16 + i * GC_MAXE + e];
            if (n == 0) continue;
            const double* p = tb.seg + 8 * row[i * GC_MAXE + e];
            mu2 += n * p[3] * p[3];
            ka += n * p[4];
            eab += n * p[5];
            na += n * p[6];
            nb += n * p[7];
        }
        associating += (ka * eab != 0.0);
        self_assoc += (na * nb != 0.0);
        polar |= (mu2 > 0.0);
    }
    int cls = 0;
    if (associating == 1 && self_assoc == 1) cls = 1;
    if (associating == 2 && self_assoc == 1) cls = 2;
    if (associating == 2 && self_assoc == 2) cls = 3;
    return 2 * cls + polar;
}

template <bool DEW>
__device__ __forceinline__ void gc_store(int64_t i, int rc, const MixResult& r, double T, double* __restrict__ p_out,
                                         double* __restrict__ rho4, uint8_t* __restrict__ status,
                                         int32_t* __restrict__ iters) {
    const bool ok = rc == BD_OK;
    if (p_out) p_out[i] = ok ? r.p * T * P_UNIT : 0.0;
    if (rho4) {
        double v0 = DEW ? r.spec0 : r.inc0, v1 = DEW ? r.spec1 : r.inc1;
        double l0 = DEW ? r.inc0 : r.spec0, l1 = DEW ? r.inc1 : r.spec1;
        reinterpret_cast<double4*>(rho4)[i] = ok ? make_double4(v0, v1, l0, l1) : make_double4(0.0, 0.0, 0.0, 0.0);
    }
    if (iters) iters[i] = ok ? r.iters : -1;
    status[i] = ok ? 0 : 1;
}

// (A work-queue schedule as in mix_kernels.hip was measured here too: the gc rows need nearly the same number of
// evaluations each, so it only adds the refill overhead: bubble 4.1 -> 4.7 ms, dew 12.4 -> 15.3 ms per 1e6 rows.)
// K7.  RETRY = false: fast pass over all rows (small caps, cap hits appended to the retry list;
// retry == nullptr -> full caps, single pass).  RETRY = true: robust pass, grid-stride over the list.
template <bool DEW, bool RETRY>
__global__ __launch_bounds__(GBLOCK) void k_gc_bubble_dew(const double* __restrict__ table, int S,
                                                          const unsigned char* __restrict__ rows,
                                                          const double* __restrict__ phi,
                                                          const double* __restrict__ temp, const double* __restrict__ z,
                                                          const double* __restrict__ p_init, int64_t n,
                                                          double* __restrict__ p_out, double* __restrict__ rho4,
                                                          uint8_t* __restrict__ status, int32_t* __restrict__ iters,
                                                          int32_t* __restrict__ retry, const int32_t* __restrict__ order) {
    extern __shared__ double lds[];
    if (RETRY && (int64_t)blockIdx.x * GBLOCK >= (int64_t)retry[0]) return;  // whole workgroup idle: skip the staging
    GcTable tb = stage_table(table, S, lds);
    double* bonds = lds + gc_table_doubles(S);  // [2*MAXE][GBLOCK]: the bond diameters d_ab
    double* row_area = bonds + 2 * GC_MAXE * GBLOCK;  // the lanes' row bytes (stage_row)
    GcModelT<double> m;
    m.c.bond_dab = bonds + threadIdx.x;
    m.c.stride = GBLOCK;
    int64_t first = (int64_t)blockIdx.x * GBLOCK + threadIdx.x;
    if (!RETRY && order) {
        // batch-wide class order from the caller (computed once per model: the rows are fixed): class-uniform waves.
        // Measured with host-sorted rows: bubble 3.0 -> 2.1 ms, dew 7.8 -> 5.6 ms per 1e6 rows
        const int64_t o = first < n ? (int64_t)order[first] : n;
        first = (o >= 0 && o < n) ? o : n;  // a foreign order array must not fault
    }
    if (!RETRY && !order) {
        // rows of the workgroup bucketed by class (LDS counting sort): lane t takes the row at sorted position t, so a
        // wave mostly runs one set of branches of the evaluation
        __shared__ int bins[GC_BINS + 1];
        __shared__ int perm[GBLOCK];
        const int t = threadIdx.x;
        if (t <= GC_BINS) bins[t] = 0;
        __syncthreads();
        int key = GC_BINS;  // rows past n sort last
        if (first < n) key = gc_bucket(rows + (size_t)first * GC_ROW_BYTES, tb);
        atomicAdd(&bins[key], 1);
        __syncthreads();
        if (t == 0) {
            int acc = 0;
#pragma unroll
            for (int b = 0; b <= GC_BINS; b++) {
                int c = bins[b];
                bins[b] = acc;
                acc += c;
            }
        }
        __syncthreads();
        perm[atomicAdd(&bins[key], 1)] = t;
        __syncthreads();
        first = (int64_t)blockIdx.x * GBLOCK + perm[t];
    }
    const int64_t total = RETRY ? min((int64_t)max(retry[0], 0), n) : n;  // count and entries bounded by n: a foreign list must not fault
    const int64_t stride = RETRY ? (int64_t)gridDim.x * GBLOCK : total;  // fast pass: one row per lane
    for (int64_t k = first; k < total; k += stride) {
        const int64_t i = RETRY ? (int64_t)retry[1 + k] : k;
        if (RETRY && (i < 0 || i >= n)) continue;
        const double T = temp[i];
        gc_coef<double>(m.c, stage_row(rows + (size_t)i * GC_ROW_BYTES, row_area), tb, phi[2 * i], phi[2 * i + 1], T);
        MixResult r;
        const double p_red = p_init[i] / (T * P_UNIT);
        const bool fast = !RETRY && retry;
        bool root_failed = false;
        // dew: the pure-liquid fugacities of Raoult's law on the one-variable evaluation (5 of a row's ~20 two-variable
        // evaluations otherwise; round 3: dew 4.25 -> 4.02 ms per 1e6 rows)
        double fug[2], rho_pure[2];
        if (DEW) pure_fugacities_on_the_line(m, fug, rho_pure);
        // (may_damp = false: no damped second run of a failed Newton here -- mix_solver_sm.hpp::newton_failed.  The second pass
        // ends with its slowest row, and on the synthetic batch the damped run recovers 1 row per 1e6 for +0.2 ms.)
        // A row that fails at a liquid root with the full caps gets the robust second attempt (bracketed liquid roots,
        // mix_solver_sm.hpp): in the second pass, or in place when there is no work list.
        int rc;
        if constexpr (DEW) {
            rc = bubble_dew_solve_sm<DEW>(m, z[i], p_red, r, fast ? GC_FAST_SS : SS_MAX_IT, fast ? GC_FAST_NEWTON : NEWTON_MAX_IT, false,
                                          &root_failed, fug, rho_pure, false, GC_NO_PROGRESS_DEW);
            if (!fast && rc != BD_OK && root_failed)
                rc = bubble_dew_solve_sm<DEW>(m, z[i], p_red, r, SS_MAX_IT, NEWTON_MAX_IT, true, nullptr, nullptr, nullptr, false, GC_NO_PROGRESS_DEW);
        } else {
            (void)root_failed;
            rc = bubble_dew_solve_sm_both<DEW>(m, z[i], p_red, r, fast ? GC_FAST_SS : SS_MAX_IT, fast ? GC_FAST_NEWTON : NEWTON_MAX_IT, !fast,
                                               nullptr, nullptr, false);
        }
        if (fast && rc != BD_OK) {  // cap hit or failed: the second pass decides
            status[i] = 1;  // provisional
            const int slot = atomicAdd(&retry[0], 1);
            if (slot >= 0 && slot < n) retry[1 + slot] = (int32_t)i;  // bounded append (see pure_kernels.hip)
        } else {
            gc_store<DEW>(i, rc, r, T, p_out, rho4, status, iters);
        }
        if (!RETRY) break;
    }
}

// GcPcSaftMix.derivatives (feos_torch/gc_pcsaft.py:443-468)
__global__ __launch_bounds__(GBLOCK) void k_gc_derivatives(const double* __restrict__ table, int S,
                                                           const unsigned char* __restrict__ rows,
                                                           const double* __restrict__ phi,
                                                           const double* __restrict__ temp,
                                                           const double* __restrict__ rho, int64_t n,
                                                           double* __restrict__ a, double* __restrict__ p,
                                                           double* __restrict__ mu, double* __restrict__ v) {
    extern __shared__ double lds[];
    GcTable tb = stage_table(table, S, lds);
    double* bonds = lds + gc_table_doubles(S);
    const int64_t i = (int64_t)blockIdx.x * GBLOCK + threadIdx.x;
    if (i >= n) return;
    GcModelT<double> m;
    m.c.bond_dab = bonds + threadIdx.x;
    m.c.stride = GBLOCK;
    gc_coef<double>(m.c, stage_row(rows + (size_t)i * GC_ROW_BYTES, bonds + 2 * GC_MAXE * GBLOCK), tb, phi[2 * i], phi[2 * i + 1], temp[i]);
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

// gradient of the bubble / dew pressure w.r.t. the six dispersion aggregates and T (implicit-
// function form, see mix_jacobian.hpp); k_ab and phi enter the model only through the
// aggregates (feos_torch/gc_pcsaft.py:181-194), the host chains them (gc_pcsaft.py in this package).
constexpr int GC_DIRS = 7;   // A00, A01, A11, B00, B01, B11, T
constexpr int GC_CHUNK = 1;  // the only dual-number direction left is T

__global__ __launch_bounds__(GJBLOCK) void k_gc_jacobian(int dew, const double* __restrict__ table, int S,
                                                         const unsigned char* __restrict__ rows,
                                                         const double* __restrict__ phi,
                                                         const double* __restrict__ temp,
                                                         const double* __restrict__ rho4, int64_t n,
                                                         double* __restrict__ jac, double* __restrict__ agg,
                                                         const int32_t* __restrict__ order) {
    typedef DN<double, GC_CHUNK> G;
    typedef T1<G> R;
    extern __shared__ double lds[];
    GcTable tb = stage_table(table, S, lds);
    double* bonds = lds + gc_table_doubles(S);                       // double model: [2*MAXE dab] x block
    G* gbonds = reinterpret_cast<G*>(bonds + 2 * GC_MAXE * GJBLOCK);  // dual model dab
    int64_t i = (int64_t)blockIdx.x * GJBLOCK + threadIdx.x;
    if (i >= n) return;
    if (order) {  // class order of the rows (see pcs_gc_bubble_dew)
        i = order[i];
        if (i < 0 || i >= n) return;
    }
    // (row bytes staged behind the dual model's bond area)
    const unsigned char* row = stage_row(rows + (size_t)i * GC_ROW_BYTES, bonds + (2 * GC_MAXE + 2 * GC_MAXE * (1 + GC_CHUNK)) * GJBLOCK);
    const double T = temp[i], ph0 = phi[2 * i], ph1 = phi[2 * i + 1];
    const double4 r4 = reinterpret_cast<const double4*>(rho4)[i];  // (V0, V1, L0, L1)
    const double s0 = dew ? r4.x : r4.z, s1 = dew ? r4.y : r4.w, i0 = dew ? r4.z : r4.x, i1 = dew ? r4.w : r4.y;
    GcModelT<double> m;
    m.c.bond_dab = bonds + threadIdx.x;
    m.c.stride = GJBLOCK;
    gc_coef<double>(m.c, row, tb, ph0, ph1, T);
    if (agg) {
#pragma unroll
        for (int k = 0; k < 3; k++) { agg[6 * i + k] = m.c.A[k]; agg[6 * i + 3 + k] = m.c.B[k]; }
    }
    PhaseEval s = phase_eval(m, s0, s1);
    PhaseEval nn = phase_eval(m, i0, i1);
    const double rs = s0 + s1, z0 = s0 / rs, z1 = s1 / rs;
    double J[3][3];
    J[0][0] = rs * (z0 * (1.0 / s.r0 + s.h00) + z1 * s.h01);
    J[1][0] = rs * (z0 * s.h01 + z1 * (1.0 / s.r1 + s.h11));
    J[2][0] = rs * (z0 * s.dp0() + z1 * s.dp1());
    J[0][1] = -i0 * (1.0 / i0 + nn.h00);
    J[1][1] = -i0 * nn.h01;
    J[2][1] = -i0 * nn.dp0();
    J[0][2] = -i1 * nn.h01;
    J[1][2] = -i1 * (1.0 / i1 + nn.h11);
    J[2][2] = -i1 * nn.dp1();
    double A[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) A[r][cc] = J[cc][r];
        A[r][3] = 0.0;
    }
    if (dew) {
        A[0][3] = J[2][0];
    } else {
        A[1][3] = -J[2][1];
        A[2][3] = -J[2][2];
    }
    double w[3];
    const bool ok = solve3(A, w);
    const double p_red = dew ? s.p() : nn.p();
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    double* g = jac + GC_DIRS * i;

    // (1) the six aggregates: the dispersion term is linear in them, a_disp = F1 sum_k A_k q_k + F2 sum_k B_k q_k with
    //     q = (r0^2, r0 r1, r1^2): d(a, da/dr0, da/dr1)/dA_k = (F1 q_k, dF1/dr0 q_k + F1 dq_k/dr0, ...) -- no dual pass
    {
        typedef T1<double> Q;
        Q F1s, F2s, F1i, F2i;
        dispersion_factors(m.c, Q(s0, 1.0, 0.0), Q(s1, 0.0, 1.0), F1s, F2s);
        dispersion_factors(m.c, Q(i0, 1.0, 0.0), Q(i1, 0.0, 1.0), F1i, F2i);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // q_k and its gradient in both phases
            const double qs = (k == 0) ? s0 * s0 : (k == 1 ? s0 * s1 : s1 * s1);
            const double qs0 = (k == 0) ? 2.0 * s0 : (k == 1 ? s1 : 0.0), qs1 = (k == 0) ? 0.0 : (k == 1 ? s0 : 2.0 * s1);
            const double qi = (k == 0) ? i0 * i0 : (k == 1 ? i0 * i1 : i1 * i1);
            const double qi0 = (k == 0) ? 2.0 * i0 : (k == 1 ? i1 : 0.0), qi1 = (k == 0) ? 0.0 : (k == 1 ? i0 : 2.0 * i1);
#pragma unroll
            for (int ab = 0; ab < 2; ab++) {
                const Q& Fs = ab == 0 ? F1s : F2s;
                const Q& Fi = ab == 0 ? F1i : F2i;
                const double aS = Fs.v * qs, gS0 = Fs.g0 * qs + Fs.v * qs0, gS1 = Fs.g1 * qs + Fs.v * qs1;
                const double aI = Fi.v * qi, gI0 = Fi.g0 * qi + Fi.v * qi0, gI1 = Fi.g1 * qi + Fi.v * qi1;
                const double dF0 = gS0 - gI0, dF1 = gS1 - gI1;
                const double dpS = -aS + s0 * gS0 + s1 * gS1;
                const double dpI = -aI + i0 * gI0 + i1 * gI1;
                const double dp = (dew ? dpS : dpI) - (w[0] * dF0 + w[1] * dF1 + w[2] * (dpS - dpI));
                g[3 * ab + k] = ok ? dp * T * P_UNIT : nanv;
            }
        }
    }

    // (2) T: one dual-number direction, both phases through one evaluation site
    {
        G gT;
        gT.v = T;
        gT.e[0] = 1.0;
        GcCoef<G> c;
        c.bond_dab = gbonds + threadIdx.x;
        c.bond_cnt = m.c.bond_cnt;  // counts are shared (written identically)
        c.stride = GJBLOCK;
        gc_coef<G>(c, row, tb, ph0, ph1, gT);
        double av[2], ag0[2], ag1[2];
#pragma unroll 1
        for (int ph = 0; ph < 2; ph++) {
            const double q0 = ph == 0 ? s0 : i0, q1 = ph == 0 ? s1 : i1;
            R a = gc_a_tangent<G, R>(c, R(G(q0), G(1.0), G(0.0)), R(G(q1), G(0.0), G(1.0)));
            if (ph == 0) { av[0] = a.v.e[0]; ag0[0] = a.g0.e[0]; ag1[0] = a.g1.e[0]; }
            else { av[1] = a.v.e[0]; ag0[1] = a.g0.e[0]; ag1[1] = a.g1.e[0]; }
        }
        const double dF0 = ag0[0] - ag0[1], dF1 = ag1[0] - ag1[1];
        const double dpS = -av[0] + s0 * ag0[0] + s1 * ag1[0];
        const double dpI = -av[1] + i0 * ag0[1] + i1 * ag1[1];
        const double dp = (dew ? dpS : dpI) - (w[0] * dF0 + w[1] * dF1 + w[2] * (dpS - dpI));
        g[6] = ok ? dp * T * P_UNIT + p_red * P_UNIT : nanv;
    }
}

}  // namespace

extern "C" {

int64_t pcs_gc_table_doubles(int S) { return (int64_t)S * 8 + 3 * (int64_t)S * S; }

int pcs_gc_bubble_dew(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                      const double* z, const double* p_init, int64_t n, double* p_out, double* rho4, uint8_t* status,
                      int32_t* iters, const int32_t* order, void* workspace, void* stream) {
    g_err[0] = 0;
    if (int e = gc_check(S, n)) return e;
    if (n == 0) return 0;
    if (!table || !rows || !phi || !temp || !z || !p_init || !status) return fail_msg("pcs_gc_bubble_dew: null required pointer");
    if (reinterpret_cast<uintptr_t>(rows) & 15) return fail_msg("pcs_gc_bubble_dew: rows must be 16-byte aligned");
    const unsigned grid = (unsigned)((n + GBLOCK - 1) / GBLOCK);
    const size_t lds = gc_lds_bytes(S, GBLOCK, 2 * GC_MAXE + GC_ROW_LDS_DOUBLES);
    hipStream_t s = as_stream(stream);
    int32_t* retry = static_cast<int32_t*>(workspace);
    if (retry) {
        if (int e = zero_ints(retry, 1, s)) return e;
    }
    if (dew) {
        hipLaunchKernelGGL((k_gc_bubble_dew<true, false>), dim3(grid), dim3(GBLOCK), lds, s, table, S, rows, phi, temp, z, p_init,
                           n, p_out, rho4, status, iters, retry, order);
        if (retry)
            hipLaunchKernelGGL((k_gc_bubble_dew<true, true>), dim3(GC_RETRY_BLOCKS), dim3(GBLOCK), lds, s, table, S, rows, phi,
                               temp, z, p_init, n, p_out, rho4, status, iters, retry, order);
    } else {
        hipLaunchKernelGGL((k_gc_bubble_dew<false, false>), dim3(grid), dim3(GBLOCK), lds, s, table, S, rows, phi, temp, z,
                           p_init, n, p_out, rho4, status, iters, retry, order);
        if (retry)
            hipLaunchKernelGGL((k_gc_bubble_dew<false, true>), dim3(GC_RETRY_BLOCKS), dim3(GBLOCK), lds, s, table, S, rows, phi,
                               temp, z, p_init, n, p_out, rho4, status, iters, retry, order);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_gc_bubble_dew launch", e);
    return 0;
}

int pcs_gc_derivatives(const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                       const double* rho, int64_t n, double* a, double* p, double* mu, double* v, void* stream) {
    g_err[0] = 0;
    if (int e = gc_check(S, n)) return e;
    if (n == 0) return 0;
    if (!table || !rows || !phi || !temp || !rho) return fail_msg("pcs_gc_derivatives: null required pointer");
    if (reinterpret_cast<uintptr_t>(rows) & 15) return fail_msg("pcs_gc_derivatives: rows must be 16-byte aligned");
    const unsigned grid = (unsigned)((n + GBLOCK - 1) / GBLOCK);
    hipLaunchKernelGGL(k_gc_derivatives, dim3(grid), dim3(GBLOCK), gc_lds_bytes(S, GBLOCK, 2 * GC_MAXE + GC_ROW_LDS_DOUBLES), as_stream(stream),
                       table, S, rows, phi, temp, rho, n, a, p, mu, v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_gc_derivatives launch", e);
    return 0;
}

int pcs_gc_jacobian(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                    const double* rho4, int64_t n, double* jac, double* agg, const int32_t* order, void* stream) {
    g_err[0] = 0;
    if (int e = gc_check(S, n)) return e;
    if (n == 0) return 0;
    if (!table || !rows || !phi || !temp || !rho4 || !jac) return fail_msg("pcs_gc_jacobian: null required pointer");
    if (reinterpret_cast<uintptr_t>(rows) & 15) return fail_msg("pcs_gc_jacobian: rows must be 16-byte aligned");
    const unsigned grid = (unsigned)((n + GJBLOCK - 1) / GJBLOCK);
    // double model: 2*MAXE doubles per thread; dual model dab: 2*MAXE * (1 + GC_CHUNK) doubles per thread
    const size_t lds = gc_lds_bytes(S, GJBLOCK, 2 * GC_MAXE + 2 * GC_MAXE * (1 + GC_CHUNK) + GC_ROW_LDS_DOUBLES);
    if (lds > 64 * 1024) {  // above the default dynamic-LDS limit
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gc_jacobian), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return fail("k_gc_jacobian attribute", ea);
    }
    hipLaunchKernelGGL(k_gc_jacobian, dim3(grid), dim3(GJBLOCK), lds, as_stream(stream), dew, table, S, rows, phi, temp,
                       rho4, n, jac, agg, order);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_gc_jacobian launch", e);
    return 0;
}

}  // extern "C"
