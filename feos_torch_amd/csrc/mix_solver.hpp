// Per-lane bubble- / dew-point solver for binary mixtures (device only).
//
// Stands where the reference calls feos' PhaseEquilibrium::bubble_point / dew_point per row
// (src/pcsaft.rs:150-214, src/gc_pcsaft.rs:103-171).  Problem: at fixed T and composition z of
// the SPECIFIED phase (liquid for bubble, vapour for dew) find rho^spec (total) and the partial
// densities rho^inc_1,2 of the incipient phase with equal chemical potentials and pressure:
//     mu_i = ln rho_i + da/drho_i,     p = sum rho - a + sum rho_k da/drho_k        (reduced)
// Newton in the logarithms of the three unknowns with the full analytic Jacobian from one
// T2<double> evaluation per phase.  Initialisation: bubble — liquid root at the caller's initial
// pressure, ideal vapour at the liquid's fugacities; dew — Raoult's law from zero-pressure pure-
// liquid fugacities (the caller's pressure and the vapour composition where a pure-component limit
// of the model is not finite), refined by ideal-vapour successive substitution: a scalar fixed-point
// map in ln(x_1/x_2), iterated with secant steps, the liquid density carried along by the Newton step
// each sweep's evaluation provides (re-solved, from the tracked density, only when that step is large).
// This file holds the evaluation entry point, the constants and the pieces shared by every driver; the iteration itself
// is the per-lane state machine of mix_solver_sm.hpp (restated sequentially by the CPU oracle, oracle/mix_solver.hpp).
// The returned pressure is the reference's final explicit Newton step (feos_torch/
// pcsaft_mix.py:435-444 / :459-468) evaluated at the converged densities.
//
// `Model` = a coefficient struct with  template<class R> R a(const R& r0, const R& r1) const  and
// packing(x0, x1) = zeta3 / rho_total.  Shared by PcSaftMix and GcPcSaftMix kernels.
#pragma once
#include "dual.hpp"
#include "pcsaft_consts.hpp"

namespace pcs {

struct PhaseEval {
    double r0, r1;
    double a, g0, g1, h00, h01, h11;
    PCS_DEV double mu0() const { return d_log(r0) + g0; }
    PCS_DEV double mu1() const { return d_log(r1) + g1; }
    PCS_DEV double p() const { return r0 + r1 - a + r0 * g0 + r1 * g1; }
    PCS_DEV double dp0() const { return 1.0 + r0 * h00 + r1 * h01; }  // dp/drho_0
    PCS_DEV double dp1() const { return 1.0 + r0 * h01 + r1 * h11; }
};

// The evaluation entry points phase_eval / line_eval are NOT inlined: round 1's sequential solver called them from ~10 sites,
// and inlined the kernels spilled ~2,700 VGPRs (3.4 KB scratch per lane); as calls 300 (A/B: dew 110 -> 76 ms, bubble 23 ->
// 18.5 ms per 1e6 rows).  The two work-queue kernels of mix_kernels.hip -- ONE evaluation site each, solver state in LDS -- call
// phase_eval_inline since round 3: the model coefficients then stay in registers / AGPRs instead of being read back from the
// caller's stack frame by ~37 loads per evaluation (512 -> 104-160 B of scratch): bubble 2.34 -> 2.20 ms, dew 4.94 -> 4.61 ms
// per 1e6 rows, results within 5e-13.  (The gc solver kernel, whose state machine is instantiated for two attempts and keeps its
// state in registers, gets slower inlined: dew 3.66 -> 4.2 ms.)
#ifndef PCS_EVAL_ATTR
#define PCS_EVAL_ATTR __device__ __attribute__((noinline))
#endif
// Value, gradient and Hessian of a in the partial densities.  Evaluated in the coordinates (u, w) = (zeta_3, rho_2) -- the
// packing fraction zeta_3 = c0 rho_1 + c1 rho_2 (c_i = packing(e_i) > 0) is a linear, invertible change of the first
// coordinate -- so that every function of the packing fraction alone is a one-variable Taylor series (dual.hpp, "D2 (x)
// T2"), then mapped back:  d/drho_1 = c0 d/du,  d/drho_2 = c1 d/du + d/dw.
template <class Model>
PCS_DEV PhaseEval phase_eval_inline(const Model& m, double r0, double r1) {
    typedef T2<double> R;
    typedef D2<double> Z;
    const double c0 = m.packing(1.0, 0.0), c1 = m.packing(0.0, 1.0);
    const double rc0 = 1.0 / c0;
    R a = m.template a_z<R, Z>(R(r0, rc0, -(c1 * rc0), 0.0, 0.0, 0.0), R(r1, 0.0, 1.0, 0.0, 0.0, 0.0), Z(r0 * c0 + r1 * c1, 1.0, 0.0));
    PhaseEval e;
    e.r0 = r0; e.r1 = r1;
    e.a = a.v;
    e.g0 = c0 * a.g0;
    e.g1 = c1 * a.g0 + a.g1;
    e.h00 = (c0 * c0) * a.h00;
    e.h01 = c0 * (c1 * a.h00 + a.h01);
    e.h11 = c1 * (c1 * a.h00 + 2.0 * a.h01) + a.h11;
    return e;
}
template <class Model>
PCS_EVAL_ATTR PhaseEval phase_eval(const Model& m, double r0, double r1) { return phase_eval_inline(m, r0, r1); }

// a and its first two derivatives along the line rho_i = x_i rho (one-variable series: ~1/3 of the multiply-adds of the
// two-variable evaluation above).  Not inlined either: the second evaluation site of the kernels that use it.
template <class Model>
PCS_EVAL_ATTR D2<double> line_eval(const Model& m, double x0, double x1, double rho) {
    typedef D2<double> R;
    return m.template a<R>(R(x0 * rho, x0, 0.0), R(x1 * rho, x1, 0.0));
}

// relative step at which a liquid root is accepted.  The roots only initialise the substitution / the Newton, and the
// state machine carries the chemical potentials to the root to first order: 1e-3 leaves a 1e-6 error
constexpr double LIQ_ROOT_TOL = 1e-3;
constexpr int LIQ_ROOT_MAX_IT = 30;  // Newton from the dense side needs ~5-10; a row that needs more fails

// 3x3 linear solve, Gaussian elimination with partial pivoting; all indices static so the
// augmented matrix stays in registers (runtime-indexed local arrays would go to scratch).
PCS_DEV void swap_rows(double* a, double* b) {
#pragma unroll
    for (int j = 0; j < 4; j++) { double t = a[j]; a[j] = b[j]; b[j] = t; }
}
PCS_DEV bool solve3(double A[3][4], double* x) {
    if (fabs(A[1][0]) > fabs(A[0][0])) swap_rows(A[0], A[1]);
    if (fabs(A[2][0]) > fabs(A[0][0])) swap_rows(A[0], A[2]);
    if (A[0][0] == 0.0) return false;
    double inv = d_recip(A[0][0]);
    double f1 = A[1][0] * inv, f2 = A[2][0] * inv;
#pragma unroll
    for (int j = 1; j < 4; j++) { A[1][j] -= f1 * A[0][j]; A[2][j] -= f2 * A[0][j]; }
    if (fabs(A[2][1]) > fabs(A[1][1])) swap_rows(A[1], A[2]);
    if (A[1][1] == 0.0) return false;
    const double inv1 = d_recip(A[1][1]);
    double f = A[2][1] * inv1;
    A[2][2] -= f * A[1][2];
    A[2][3] -= f * A[1][3];
    if (A[2][2] == 0.0) return false;
    x[2] = A[2][3] * d_recip(A[2][2]);
    x[1] = (A[1][3] - A[1][2] * x[2]) * inv1;
    x[0] = (A[0][3] - A[0][1] * x[1] - A[0][2] * x[2]) * inv;
    return true;
}

// Iteration caps.  The kernels are latency-bound by their slowest lane, so the caps matter: converged
// rows need <= 31 (bubble) / <= 25 at the 99.9th percentile (dew) Newton iterations on the synthetic
// workload; rows that need more are reported as failed (status 1), as are successive-substitution
// runs that have not settled after SS_MAX_IT sweeps.  The CPU oracle uses the same caps.
constexpr int SS_MAX_IT = 40;
// ... and, for a trace component, no longer in relative ones: |d ln(x_1/x_2)| of the sweep below SS_RES_TOL (with x_2 ~ 1e-6
// falling by the damped factor 5 per sweep the absolute change of x_1 is below SS_TOL long before the trace component has
// found its level; the Newton then starts in the dimerisation regime of a strongly associating trace component, where
// d mu / d ln rho changes sign, and runs away)
constexpr double SS_RES_TOL = 1e-2;
// the substitution's secant step is taken where the residual of the map decreases along xi, however slowly (slope below this;
// -0.05 until round 3: on a nearly flat map -- the liquid close to a liquid-liquid critical point -- the plain sweeps then
// crept at dx ~ 2e-4 per sweep into the cap and the Newton started far from the solution); its length stays capped at ln 5
// and the bracket of the fixed point stays in force
constexpr double SS_SECANT_SLOPE = -1e-5;
constexpr double SS_TOL = 1e-5;        // composition change at which the dew-point successive substitution hands over to Newton
// A Newton iteration whose largest step has not shrunk by NEWTON_PROGRESS (relative to the smallest one so far) within
// NEWTON_NO_PROGRESS iterations is given up: it cycles (typically a 2-cycle whose amplitude creeps down by 1e-3 per
// round: no phase equilibrium at this state).  Bubble points start next to the solution (liquid root + ideal vapour)
// and get the shorter leash; measured on the synthetic batches no converging row is lost by either.
constexpr int NEWTON_NO_PROGRESS = 20, NEWTON_NO_PROGRESS_BUBBLE = 15;  // dew: 30 until round 3 (now followed by the damped second run)
constexpr double NEWTON_PROGRESS = 0.9;
constexpr double NEWTON_TRACE = 1e-4, NEWTON_TRACE_MAX = 100.0;  // see the Newton step below
constexpr int NEWTON_MAX_IT = 60;
constexpr int NEWTON_DAMPED_MAX_IT = 24, NEWTON_DAMPED_HALVINGS = 4;  // second, damped run of a failed dew-point Newton (mix_solver_sm.hpp)
// Largest step (in the logarithms of the three densities) at which the Newton iteration is accepted.  The state the two
// evaluations were taken at is then within that of the solution, the densities handed out carry the step, and the
// reference's final formula is second order in it: measured against 1e-9 (the value until round 3) on the 1e6-row batches
// the pressures move by <= 1.0e-12 relative with 1e-7 (5e-11 with 1e-6: too much) and no failure mask changes; a wave of
// the gc kernels no longer waits for the lanes that only needed the confirming iteration (gc bubble 1.76 -> 1.64 ms).
constexpr double NEWTON_ACCEPT = 1e-7;
// An iteration whose step has stopped shrinking below NEWTON_FLOOR (ratio >= 0.25) sits on the rounding floor of its residuals
// (a trace component: the liquid's chemical potential of a component at x ~ 1e-13 carries ~1e-7 of noise) and is accepted;
// the reference's final formula is second order in the remaining step (<= 1e-12).  1e-7 until round 3: rows cycling at
// 4e-7 ... 9e-7 ran into the no-progress exit and failed.
constexpr double NEWTON_FLOOR = 1e-6;

struct MixResult {
    double spec0, spec1, inc0, inc1;  // converged partial densities
    double p;                         // reduced pressure from the reference's final formula
    int iters;
};

// the reference's final formula (pcsaft_mix.py:435-444 with spec = liquid, inc = vapour; :459-468
// with the roles swapped): p = -(a_inc/rho_inc + p_spec v + g - 1) / (1/rho_inc - v)
PCS_DEV double bubble_dew_formula(const PhaseEval& s, const PhaseEval& n) {
    double rho_i = n.r0 + n.r1;
    double y0 = n.r0 / rho_i, y1 = n.r1 / rho_i;
    // partial molar volumes of the specified phase: v_i = (dp/drho_i) / sum_j rho_j dp/drho_j  (:416-418)
    double d0 = s.dp0(), d1 = s.dp1();
    double den = 1.0 / (s.r0 * d0 + s.r1 * d1);
    double v = (y0 * d0 + y1 * d1) * den;
    double g = y0 * (d_log(n.r0 / s.r0) - s.g0) + y1 * (d_log(n.r1 / s.r1) - s.g1);
    return -(n.a / rho_i + s.p() * v + g - 1.0) / (1.0 / rho_i - v);
}

// Return codes of bubble_dew_solve
enum : int { BD_OK = 0, BD_FAILED = 1, BD_CAP = 2 };

// (The sequential, readable form of the solver -- nested loops instead of a state machine -- is the CPU oracle's
// oracle/mix_solver.hpp, which restates mix_solver_sm.hpp decision by decision; the device-side copy of it that rounds 1-2
// kept here for reference was unused since the state machine and is gone.)

}  // namespace pcs
