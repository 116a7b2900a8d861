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
// This file is the sequential form (readable, restated 1:1 by oracle/mix_solver.hpp, used by the
// single-pass kernels); mix_solver_sm.hpp is the same algorithm as a per-lane state machine.
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

// The two evaluation entry points are NOT inlined: the solver calls them from ~10 sites, and inlined the
// kernels spill ~2,700 VGPRs (3.4 KB scratch per lane); as calls 300 (A/B: dew 110 -> 76 ms, bubble 23 -> 18.5 ms
// per 1e6 rows).
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
#ifdef PCS_EVAL_PLAIN  // A/B builds: both partial densities as coordinates
    {
        R a = m.template a<R>(R(r0, 1.0, 0.0, 0.0, 0.0, 0.0), R(r1, 0.0, 1.0, 0.0, 0.0, 0.0));
        PhaseEval e;
        e.r0 = r0; e.r1 = r1;
        e.a = a.v; e.g0 = a.g0; e.g1 = a.g1; e.h00 = a.h00; e.h01 = a.h01; e.h11 = a.h11;
        return e;
    }
#endif
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

// p and dp/drho along a fixed composition (x0, x1): one D2 evaluation
template <class Model>
PCS_EVAL_ATTR void line_eval(const Model& m, double x0, double x1, double rho, double& p, double& dp, double& a) {
    typedef D2<double> R;
    R r = m.template a<R>(R(x0 * rho, x0, 0.0), R(x1 * rho, x1, 0.0));
    a = r.v;
    p = rho - r.v + rho * r.d1;
    dp = 1.0 + rho * r.d2;
}

// relative step at which a liquid root is accepted.  The roots only initialise the substitution / the Newton, and the
// state machine carries the chemical potentials to the root to first order: 1e-3 leaves a 1e-6 error
constexpr double LIQ_ROOT_TOL = 1e-3;
constexpr int LIQ_ROOT_MAX_IT = 30;  // Newton from the dense side needs ~5-10; a row that needs more fails

// liquid-like root of p(rho) = p_spec at composition x.  Cold start: Newton from the dense side
// (eta = 0.5, monotone on the convex branch).  rho_start > 0: warm start from a previous root at a
// nearby composition (successive substitution), falling back to the cold start if it misbehaves.
template <class Model>
PCS_DEV bool liquid_root(const Model& m, double x0, double x1, double p_spec, double& rho_out, double rho_start = 0.0) {
    double pk = m.packing(x0, x1);
    bool warm = rho_start > 0.0, dense = false;
    double rho = warm ? rho_start : 0.5 / pk;
    double err_prev = 1.0;
    for (int it = 0; it < LIQ_ROOT_MAX_IT; it++) {
        double p, dp, a;
        line_eval(m, x0, x1, rho, p, dp, a);
        if (!warm && it == 0 && !(p > p_spec)) {
            rho = 0.62 / pk;
            dense = true;
            line_eval(m, x0, x1, rho, p, dp, a);
        }
        // Newton on (p - p_spec)(1 - eta)^4 = 0 (same root, nearly linear: the hard-sphere pole is scaled out);
        // plain Newton for the dense restart, which is monotone from above
        double den = dense ? dp : dp - 4.0 * (p - p_spec) * pk / (1.0 - rho * pk);
        bool bad = !(dp > 0.0) || !(den > 0.0) || !is_finite_bits(p);
        double step = (p - p_spec) / den;
        double rho_new = rho - step;
        bad = bad || !(rho_new > 0.0) || !is_finite_bits(rho_new) || (warm && !(rho_new * pk < 0.7));
        if (bad) {
            if (!warm) return false;
            warm = false;  // restart cold
            rho = 0.5 / pk;
            err_prev = 1.0;
            it = -1;
            continue;
        }
        double err = fabs(step) / rho;
        bool done = err <= LIQ_ROOT_TOL || (it >= 3 && err < 1e-7 && err >= 0.25 * err_prev);
        err_prev = err;
        rho = rho_new;
        if (done) {
            rho_out = rho;
            return true;
        }
    }
    return false;
}

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
    double inv = 1.0 / A[0][0];
    double f1 = A[1][0] * inv, f2 = A[2][0] * inv;
#pragma unroll
    for (int j = 1; j < 4; j++) { A[1][j] -= f1 * A[0][j]; A[2][j] -= f2 * A[0][j]; }
    if (fabs(A[2][1]) > fabs(A[1][1])) swap_rows(A[1], A[2]);
    if (A[1][1] == 0.0) return false;
    double f = A[2][1] / A[1][1];
    A[2][2] -= f * A[1][2];
    A[2][3] -= f * A[1][3];
    if (A[2][2] == 0.0) return false;
    x[2] = A[2][3] / A[2][2];
    x[1] = (A[1][3] - A[1][2] * x[2]) / A[1][1];
    x[0] = (A[0][3] - A[0][1] * x[1] - A[0][2] * x[2]) * inv;
    return true;
}

// Iteration caps.  The kernels are latency-bound by their slowest lane, so the caps matter: converged
// rows need <= 31 (bubble) / <= 25 at the 99.9th percentile (dew) Newton iterations on the synthetic
// workload; rows that need more are reported as failed (status 1), as are successive-substitution
// runs that have not settled after SS_MAX_IT sweeps.  The CPU oracle uses the same caps.
constexpr int SS_MAX_IT = 40;
constexpr double SS_TOL = 1e-5;        // composition change at which the dew-point successive substitution hands over to Newton
// A Newton iteration whose largest step has not shrunk by NEWTON_PROGRESS (relative to the smallest one so far) within
// NEWTON_NO_PROGRESS iterations is given up: it cycles (typically a 2-cycle whose amplitude creeps down by 1e-3 per
// round: no phase equilibrium at this state).  Bubble points start next to the solution (liquid root + ideal vapour)
// and get the shorter leash; measured on the synthetic batches no converging row is lost by either.
constexpr int NEWTON_NO_PROGRESS = 30, NEWTON_NO_PROGRESS_BUBBLE = 15;
constexpr double NEWTON_PROGRESS = 0.9;
constexpr double NEWTON_TRACE = 1e-4, NEWTON_TRACE_MAX = 100.0;  // see the Newton step below
constexpr int NEWTON_MAX_IT = 60;

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

// ss_max / newton_max: iteration caps of this call.  BD_CAP = a cap was hit before the iteration
// settled: with the full caps that is a failure; the fast pass of the kernels uses small caps and
// hands BD_CAP rows to the robust pass, which repeats the identical arithmetic with the full caps.
template <bool DEW, class Model>
PCS_DEV int bubble_dew_solve(const Model& m, double z0, double p_init, MixResult& out, int ss_max = SS_MAX_IT,
                             int newton_max = NEWTON_MAX_IT) {
    const double z1 = 1.0 - z0;
    double rs, ri0, ri1;
    out.iters = 0;
    if (!DEW) {
        if (!liquid_root(m, z0, z1, p_init, rs) && !liquid_root(m, z0, z1, 0.0, rs)) return BD_FAILED;
        PhaseEval e = phase_eval(m, z0 * rs, z1 * rs);
        ri0 = e.r0 * exp(e.g0);
        ri1 = e.r1 * exp(e.g1);
    } else {
        double f[2];
        bool ok = true;
#pragma unroll 1
        for (int i = 0; i < 2; i++) {
            double x0 = (i == 0) ? 1.0 : 0.0, x1 = 1.0 - x0, rho0;
            if (!liquid_root(m, x0, x1, 0.0, rho0)) { ok = false; break; }
            PhaseEval e = phase_eval(m, x0 * rho0, x1 * rho0);
            f[i] = rho0 * exp(i == 0 ? e.g0 : e.g1);
        }
        double x0, x1, p0;
        if (ok) {
            p0 = 1.0 / (z0 / f[0] + z1 / f[1]);
            x0 = z0 * p0 / f[0];
            x1 = z1 * p0 / f[1];
        } else {
            p0 = p_init;
            x0 = z0;
            x1 = z1;
        }
        double rl = 0.0, xi_prev = 0.0, res_prev = 0.0, xi_lo = -1e300, xi_hi = 1e300;
        bool settled = false, have = false;
        for (int ss = 0; ss < ss_max; ss++) {
            // The liquid density is not re-solved in every sweep: the evaluation at (x, rl) gives p and dp/drho along x,
            // i.e. the Newton step drho to the zero-pressure root, and the chemical potentials are carried to that
            // root to first order with the Hessian.  A full root solve is done at the start and whenever the step is
            // not small (composition moved a lot) or the linearisation is unusable.
            PhaseEval e;
            double drho = 0.0;
            bool fine_prev = false;
#pragma unroll 1
            for (int attempt = 0; attempt < 2; attempt++) {
                if (!have || attempt == 1) {
                    // a re-solve starts from the tracked density when the evaluation there was usable
                    const double warm = (have && attempt == 1 && fine_prev) ? rl : 0.0;
                    if (!liquid_root(m, x0, x1, 0.0, rl, warm) && !liquid_root(m, x0, x1, p0, rl)) return BD_FAILED;
                    have = true;
                }
                e = phase_eval(m, x0 * rl, x1 * rl);
                double p = e.p(), dp = x0 * e.dp0() + x1 * e.dp1();
                drho = -p / dp;
                bool fine = (dp > 0.0) && is_finite_bits(p);
                fine_prev = fine;
                if (fine && fabs(drho) <= 0.05 * rl) break;
                if (attempt == 1) {
                    if (!fine) return BD_FAILED;
                    if (!(fabs(drho) <= 0.05 * rl)) drho = 0.0;
                }
            }
            double rlc = rl + drho;
            // w_i = z_i / (rho exp(G_i)) with the smaller exponent factored out: far from the solution G_i exceeds the range
            // of exp (both weights 0, composition 0/0) although only their ratio and the pressure estimate are needed
            double G0 = e.g0 + (x0 * e.h00 + x1 * e.h01) * drho, G1 = e.g1 + (x0 * e.h01 + x1 * e.h11) * drho;
            double Gm = fmin(G0, G1);
            double w0 = z0 * exp(Gm - G0), w1 = z1 * exp(Gm - G1);
            rl = rlc;
            double sum = (w0 + w1) / (rlc * exp(Gm));
            double n0 = w0 / (w0 + w1), n1 = w1 / (w0 + w1);
            double dx = fabs(n0 - x0);
            // The sweep is a scalar fixed-point map xi -> G(xi) in xi = ln(x_1/x_2); its plain iteration converges
            // linearly (slowly for strongly non-ideal liquids), so from the second sweep on the secant step on
            // r(xi) = G(xi) - xi is taken when it is well defined (r decreasing, step at most ln 5).
            double xi = d_log(x0 / x1);
            double res = d_log(n0 / n1) - xi;
            bool secant = false;
            // Bracket of the fixed point: r > 0 at xi_lo, r < 0 at xi_hi (r decreases through a stable fixed point).  For
            // strongly non-ideal liquids the map cycles around a steep or discontinuous stretch of r (the liquid root
            // changes branch); an iterate that leaves the bracket is then replaced by its midpoint, and a bracket
            // narrower than the tolerance ends the substitution.
            if (res > 0.0 && xi > xi_lo) xi_lo = xi;
            if (res < 0.0 && xi < xi_hi) xi_hi = xi;
            if (ss > 0 && xi != xi_prev) {
                double slope = (res - res_prev) / (xi - xi_prev);
                if (slope < -0.05) {
                    double dxi = fmin(fmax(-res / slope, -1.6), 1.6);
                    double e = exp(xi + dxi);
                    x0 = e / (1.0 + e);
                    x1 = 1.0 / (1.0 + e);
                    secant = true;
                }
            }
            xi_prev = xi;
            res_prev = res;
            if (!secant) {
                // damp when a component would change by more than a factor 5 in one sweep
                n0 = fmin(fmax(n0, 0.2 * x0), 5.0 * x0);
                n1 = fmin(fmax(n1, 0.2 * x1), 5.0 * x1);
                double s2 = n0 + n1;
                x0 = n0 / s2;
                x1 = n1 / s2;
            }
            bool narrow = false;
            if (xi_lo < xi_hi && xi_lo > -1e299 && xi_hi < 1e299) {
                double xin = d_log(x0 / x1);
                if (!(xin > xi_lo && xin < xi_hi)) {
                    double e = exp(0.5 * (xi_lo + xi_hi));
                    x0 = e / (1.0 + e);
                    x1 = 1.0 / (1.0 + e);
                }
                narrow = xi_hi - xi_lo < SS_TOL;
            }
            p0 = 1.0 / sum;
            if (dx < SS_TOL || narrow) { settled = true; break; }
        }
        if (!settled && ss_max < SS_MAX_IT) return BD_CAP;
        ri0 = x0 * rl;
        ri1 = x1 * rl;
        rs = p0;
    }
    double err_prev = 1.0, err_best = 1e300;
    int it_best = 0;
    for (int it = 0; it < newton_max; it++) {
        PhaseEval s = phase_eval(m, z0 * rs, z1 * rs);
        PhaseEval n = phase_eval(m, ri0, ri1);
        double A[3][4];
        // rows: mu_0, mu_1, p;  columns: ln rho_spec, ln rho_inc_0, ln rho_inc_1
        A[0][0] = rs * (z0 * (1.0 / s.r0 + s.h00) + z1 * s.h01);
        A[1][0] = rs * (z0 * s.h01 + z1 * (1.0 / s.r1 + s.h11));
        A[2][0] = rs * (z0 * s.dp0() + z1 * s.dp1());
        A[0][1] = -ri0 * (1.0 / ri0 + n.h00);
        A[1][1] = -ri0 * n.h01;
        A[2][1] = -ri0 * n.dp0();
        A[0][2] = -ri1 * n.h01;
        A[1][2] = -ri1 * (1.0 / ri1 + n.h11);
        A[2][2] = -ri1 * n.dp1();
        A[0][3] = -(s.mu0() - n.mu0());
        A[1][3] = -(s.mu1() - n.mu1());
        A[2][3] = -(s.p() - n.p());
        double du[3];
        if (!solve3(A, du)) return BD_FAILED;
        double mx = fmax(fabs(du[0]), fmax(fabs(du[1]), fabs(du[2])));
        if (!is_finite_bits(mx)) return BD_FAILED;
        // no new smallest Newton step for NEWTON_NO_PROGRESS iterations: the iteration cycles / wanders (no phase
        // equilibrium at this state, or the EOS is ill-behaved there) -> fail now, not at the cap
        if (mx < NEWTON_PROGRESS * err_best) { err_best = mx; it_best = it; }
        else if (it - it_best >= (DEW ? NEWTON_NO_PROGRESS : NEWTON_NO_PROGRESS_BUBBLE)) return BD_FAILED;
        // at most a factor e per iteration -- except for a trace component of the incipient phase (mole fraction below
        // NEWTON_TRACE): its chemical potential is linear in ln rho_i there (ideal dilution), so the Newton step lands on
        // the solution however long it is and limiting it only makes the iteration march (rows with p ~ 1e-10 Pa and
        // x_i ~ 1e-30 needed 35 ... 90 iterations of unit steps, the longer ones ran into the cap)
        const double rtot = ri0 + ri1;
        const bool tr0 = ri0 < NEWTON_TRACE * rtot, tr1 = ri1 < NEWTON_TRACE * rtot;
        const double mxl = fmax(fabs(du[0]), fmax(tr0 ? 0.0 : fabs(du[1]), tr1 ? 0.0 : fabs(du[2])));
        const double scale = mxl > 1.0 ? 1.0 / mxl : 1.0;
        rs *= exp(scale * du[0]);
        ri0 *= exp(tr0 ? fmin(fmax(du[1], -NEWTON_TRACE_MAX), NEWTON_TRACE_MAX) : scale * du[1]);
        ri1 *= exp(tr1 ? fmin(fmax(du[2], -NEWTON_TRACE_MAX), NEWTON_TRACE_MAX) : scale * du[2]);
        out.iters = it + 1;
        // collapsed onto the trivial solution (both phases identical): the Jacobian is singular there and the steps wander
        // along its null direction until a cap stops them -> give the row up now
        if (fabs(ri0 + ri1 - rs) <= 1e-6 * rs && fabs(ri0 - z0 * rs) <= 1e-6 * rs) return BD_FAILED;
        bool stagnated = it >= 3 && mx < 1e-7 && mx >= 0.25 * err_prev;
        err_prev = mx;
        if (mx <= 1e-9 || stagnated) {
            double dens_i = ri0 + ri1;
            double lo = DEW ? rs : dens_i, hi = DEW ? dens_i : rs;
            if (!(lo < hi * (1.0 - 1e-6))) return BD_FAILED;  // trivial solution
            // converged: the state of this iteration's evaluations is within mx of the solution and the reference's
            // final formula is second order in that error, so it is applied to them directly; the densities handed
            // out carry the last step
            out.spec0 = z0 * rs; out.spec1 = z1 * rs; out.inc0 = ri0; out.inc1 = ri1;
            out.p = bubble_dew_formula(s, n);
            return is_finite_bits(out.p) ? BD_OK : BD_FAILED;
        }
    }
    return (newton_max < NEWTON_MAX_IT) ? BD_CAP : BD_FAILED;
}

}  // namespace pcs
