// Per-lane solvers for the pure-component path (device only).
//
// What the reference delegates to the third-party feos crate (src/pcsaft.rs:91
// PhaseEquilibrium::pure, :116-122 State::new_npt(.., Liquid)) is implemented here as
// fixed-cap batched Newton iterations, one state point per lane:
//
//   * liquid density at given (T, p): Newton in rho from a liquid-like packing fraction,
//     monotone from the high-density side where p(rho) is convex;
//   * pure VLE at given T: both phase densities are driven towards the equal-area pressure
//     p* = -(f_V - f_L)/(v_V - v_L),  f = a/rho + ln rho, by one Newton step each per
//     iteration.  p* is exactly the reference's final formula (feos_torch/pcsaft_pure.py:214)
//     and the liquid step is exactly its equilibrium-density formula (:232), so the values the
//     reference computes AFTER the solve fall out of the last iteration for free.
//
// Loops are wave-uniform: every lane keeps a `done` flag and the loop exits on
// __ballot(!done) == 0 (or the iteration cap).
#pragma once
#include "pure_model.hpp"
#include "pure_f32.hpp"

namespace pcs {

enum : int { ST_OK = 0, ST_FAILED = 1, ST_RETRY = 2, ST_FALLBACK = 3 };

struct Eval {
    double a, p, dp;  // a, p = rho - a + rho a', dp/drho = 1 + rho a''   (pcsaft_pure.py:182)
};

PCS_DEV Eval pure_eval(const PureCoef<double>& c, double rho) {
    D2<double> r = pure_a<double, D2<double>>(c, D2<double>(rho, 1.0, 0.0));
    Eval e;
    e.a = r.v;
    e.p = rho - r.v + rho * r.d1;
    e.dp = 1.0 + rho * r.d2;
    return e;
}

// a'(rho) as well (residual chemical potential), used for the ideal-gas vapour estimate
PCS_DEV Eval pure_eval_mu(const PureCoef<double>& c, double rho, double& mu_res) {
    D2<double> r = pure_a<double, D2<double>>(c, D2<double>(rho, 1.0, 0.0));
    Eval e;
    e.a = r.v;
    e.p = rho - r.v + rho * r.d1;
    e.dp = 1.0 + rho * r.d2;
    mu_res = r.d1;
    return e;
}

constexpr int LIQ_MAX_IT = 40;
constexpr int LITE_MAX_IT = 3;
constexpr int VLE_MAX_IT = 40;
constexpr double ETA_START = 0.5;
constexpr double TOL_STEP = 1e-6;  // relative Newton step at which a lane is converged (see vle_step)
// Pressure-only output: with the second-order corrected p* (vle_step) the error is O(step^3), so the
// lanes may stop at much larger steps (measured: same max error vs the long-double oracle, x1.11).
constexpr double TOL_L_P = 1e-5, TOL_V_P = 1e-4;

// Newton for p(rho) = p_spec from the dense side.  Returns ST_OK with the converged density
// (rho) and the LAST Newton update already applied (so rho is also the reference's final
// formula rho - (p - p_spec)/dp of pcsaft_pure.py:198), ST_FAILED when the liquid branch has
// no root at this pressure (iterate crossed the spinodal), or when the cap is hit.
// `tol` is loose (1e-6) when used as an initialiser.
// `skip` lanes idle through the (wave-uniform) loop and return ST_OK untouched.
// `warm` lanes start from `rho` as passed in (fp32 pre-solve) instead of eta = 0.5.
PCS_DEV int liquid_newton(const PureCoef<double>& c, double p_spec, double tol, double& rho, Eval& last, bool skip = false,
                          bool warm = false) {
    if (!warm) rho = ETA_START / c.ceta;
    bool done = skip, fail = false;
    for (int it = 0; it < LIQ_MAX_IT; it++) {
        if (!done && !fail) {
            Eval e = pure_eval(c, rho);
            if (it == 0 && !warm && !(e.p > p_spec)) {
                // very cold / very dense state: start further right (still eta < 0.74)
                rho = 0.62 / c.ceta;
                e = pure_eval(c, rho);
            }
            if (!is_finite_bits(e.dp) || !(e.dp > 0.0) || !is_finite_bits(e.p)) {
                fail = true;
            } else {
                double step = (e.p - p_spec) / e.dp;
                last = e;
                double rho_new = rho - step;
                if (!is_finite_bits(rho_new) || !(rho_new > 0.0)) {
                    fail = true;
                } else {
                    done = fabs(step) <= tol * rho;
                    rho = rho_new;
                }
            }
        }
        if (__ballot(!done && !fail) == 0ull) break;
    }
    return (done && !fail) ? ST_OK : ST_FAILED;
}

constexpr float PCS_K2_F32_TOL = 1e-4f;
// Liquid density at (T, p): fp32 root (pure_f32.hpp) to its noise floor, then the fp64 Newton of
// pcsaft_pure.py:196-199 from there (typically one evaluation).  Lanes whose fp32 pass misbehaves
// start from eta = 0.5 in fp64.
PCS_DEV int liquid_density_solve(const PureCoef<double>& c, double p_spec, double tol, double& rho, Eval& last) {
    bool warm = false;
#ifdef PCS_F32_PRESOLVE
    {
        PureCoefF f;
        to_f32(c, f);
        float rl;
        int n_eval = 0;
        warm = liquid_root_f32(f, (float)p_spec, PCS_K2_F32_TOL, PCS_K2_F32_TOL, 12, rl, n_eval);
        rho = (double)rl;
    }
#endif
    int st;
    if (__ballot(!warm) == 0ull) {
        // every lane of the wave has its fp32 root: ONE fp64 evaluation in straight-line code (it is the last one on all
        // but 0.04 % of the rows; as the first trip of liquid_newton's loop the same evaluation cost 0.3 ms per 1e7 rows
        // more: the loop's live ranges spill), then the general loop for the waves in which a lane has to go on
        const Eval e = pure_eval(c, rho);
        bool fail = !is_finite_bits(e.dp) || !(e.dp > 0.0) || !is_finite_bits(e.p);
        const double step = (e.p - p_spec) / e.dp;
        const double rho_new = rho - step;
        fail = fail || !is_finite_bits(rho_new) || !(rho_new > 0.0);
        bool done = false;
        if (!fail) {
            last = e;
            done = fabs(step) <= tol * rho;
            rho = rho_new;
        }
        st = fail ? ST_FAILED : ST_OK;
        if (__ballot(!done && !fail) != 0ull) {
            const int st2 = liquid_newton(c, p_spec, tol, rho, last, done || fail, true);
            if (!done && !fail) st = st2;
        }
    } else {
        st = liquid_newton(c, p_spec, tol, rho, last, false, warm);
    }
#ifdef PCS_F32_PRESOLVE
    if (__ballot(warm && st != ST_OK) != 0ull) {
        // the fp32 root was not on the liquid branch after all: redo those lanes from the dense side
        double rho2;
        Eval last2;
        int st2 = liquid_newton(c, p_spec, tol, rho2, last2, !(warm && st != ST_OK), false);
        if (warm && st != ST_OK) {
            st = st2;
            rho = rho2;
            last = last2;
        }
    }
#endif
    return st;
}

struct VleResult {
    double rho_v, rho_l;   // phase densities after the last Newton update (the converged state)
    double p_star;         // equal-area pressure at that state, reduced   (pcsaft_pure.py:214 / :231)
    int iters;
};

// x > 0 that is false for NaN even when the compiler may assume there are no NaNs
PCS_DEV bool gt0(double x) { return is_finite_bits(x) && x > 0.0; }

// A saturated vapour has 0 < Z = p/(rho kT) <= 1 (attraction dominates below the critical temperature).  Strongly
// polar parameter sets give the EOS a second van-der-Waals loop at liquid-like densities; an iteration that lands on
// it returns an "equilibrium" between two dense states with a negative or enormous pressure, which this rejects.
PCS_DEV bool vapour_is_physical(double p_star, double rho_v) { return gt0(p_star) && p_star <= 1.0001 * rho_v; }

// One coupled Newton update from evaluations l, v at (rl, rv):
//   p*  = -(f_V - f_L)/(v_V - v_L),  f = a/rho + ln rho          (== pcsaft_pure.py:214)
//   rho_k <- rho_k - (p_k - p*)/p'_k   for both phases           (liquid: == pcsaft_pure.py:232)
// p* is stationary w.r.t. both densities at the solution, so its value at the UPDATED densities
// is p* + 1/2 [ (p_V-p*)^2/(rho_V^2 p'_V) - (p_L-p*)^2/(rho_L^2 p'_L) ] / (v_V - v_L) + O(step^3):
// the pressure is taken with this second-order term, which makes it exact to ~1e-17 once the
// relative steps are below TOL_STEP = 1e-6 and saves the confirming iteration.
struct VleStep {
    double p_star, p_corr, dl, dv;
};
PCS_DEV VleStep vle_step(const Eval& l, const Eval& v, double rl, double rv) {
    VleStep s;
    // d_recip / d_log: the refined hardware reciprocal and the short logarithm where the unit is built with
    // PCS_FAST_RCP / PCS_FAST_LOG (dual.hpp), the IEEE division and library log otherwise
    double inv_v = d_recip(rv), inv_l = d_recip(rl);
    double inv_dv = d_recip(inv_v - inv_l);
    s.p_star = -(v.a * inv_v - l.a * inv_l + d_log(rv * inv_l)) * inv_dv;
    double rl_res = l.p - s.p_star, rv_res = v.p - s.p_star;
    double il = d_recip(l.dp), iv = d_recip(v.dp);
    s.dl = -rl_res * il;
    s.dv = -rv_res * iv;
    s.p_corr = s.p_star + 0.5 * ((rv_res * rv_res) * (inv_v * inv_v) * iv - (rl_res * rl_res) * (inv_l * inv_l) * il) * inv_dv;
    return s;
}

// Fast path of the pure VLE: zero-pressure liquid + ideal-gas vapour initialisation, then the
// coupled Newton.  ST_RETRY = this initialisation does not apply (near-critical temperature);
// the robust kernel takes those rows.
// tol_l: relative liquid step at which a lane stops.  TOL_STEP suffices for the pressure (second-order
// corrected); the saturated liquid density itself is only as good as the last step squared, so the
// caller passes a tighter value when that output is requested.
// LEAN: lanes without a usable fp32 pre-solve are not solved here (ST_FALLBACK, see vle_fast_lite): the main
// kernels instantiate this, k_pure_vle_fallback the complete form.
template <bool LEAN>
PCS_DEV int vle_fast(const double* par, double T, VleResult& out, double tol_l = TOL_L_P, double tol_v = TOL_V_P) {
    double rl = 0.0, rv = 0.0;
    Eval l;
    bool warm = false;
    float dpl32 = 1.0f, dpv32 = 1.0f;
#ifdef PCS_F32_PRESOLVE
    PureCoefF cf;
    pure_coef_f32(cf, par, T);
#endif
#ifdef PCS_F32_PRESOLVE
    warm = vle_presolve_f32(cf, rl, rv, dpl32, dpv32);  // fp32 initialiser + first iterations (pure_f32.hpp)
#endif
    PureCoef<double> c;
    pure_coef<double>(c, par, T, false);
    bool active = warm;
    if (!LEAN && __ballot(!warm) != 0ull) {
        // lanes without a usable fp32 result: fp64 zero-pressure liquid (the others idle through it)
        double rl0;
        int st = liquid_newton(c, 0.0, 1e-3, rl0, l, warm);
        if (!warm) {
            rl = rl0;
            active = (st == ST_OK);
        }
    }
    if (LEAN && !warm) { rl = 0.4 / c.ceta; rv = 1e-3 * rl; }  // harmless evaluation point for the idle lanes
    {
        double mu;
        l = pure_eval_mu(c, rl, mu);  // all lanes (wave-uniform call); also the first liquid evaluation
        if (!is_finite_bits(l.dp) || !(l.dp > 0.0) || !is_finite_bits(mu)) active = false;
        if (!warm) rv = rl * exp(mu);  // ln rho_V = ln rho_L + a'(rho_L): ideal vapour at the liquid's fugacity
        // strongly non-ideal vapour estimates are left to the robust path
        // (a warm lane carries a converged fp32 vapour density instead: only sanity-checked)
        if (!is_finite_bits(rv) || !(rv < (warm ? 0.7 : 0.05) * rl)) active = false;
    }
    bool done = false;
    out.iters = 0;
    for (int it = 0; it < VLE_MAX_IT; it++) {
        if (active && !done) {
            Eval v = pure_eval(c, rv);
            VleStep s = vle_step(l, v, rl, rv);
            bool ok = is_finite_bits(l.dp) && is_finite_bits(v.dp) && (l.dp > 0.0) && (v.dp > 0.0) && is_finite_bits(s.p_star) && is_finite_bits(s.dl) && is_finite_bits(s.dv);
            double rl_new = rl + s.dl, rv_new = rv + s.dv;
            ok = ok && (rl_new > 0.0) && (rv_new > 0.0) && (rv_new < rl_new);
            if (!ok) {
                active = false;
            } else {
                done = (fabs(s.dl) <= tol_l * rl) && (fabs(s.dv) <= tol_v * rv);
                rl = rl_new;
                rv = rv_new;
                out.rho_v = rv;
                out.rho_l = rl;
                out.p_star = s.p_corr;
                out.iters = it + 1;
            }
        }
        if (__ballot(active && !done) == 0ull) break;
        if (active && !done) l = pure_eval(c, rl);
    }
    if (LEAN && !warm) return ST_FALLBACK;
    if (done && out.rho_v < 0.7 * out.rho_l && vapour_is_physical(out.p_star, out.rho_v)) return ST_OK;
    return ST_RETRY;  // includes cap hit and near-critical states: let the robust path decide
}

#ifdef PCS_F32_PRESOLVE
// Pressure-only fast path: fp32 pre-solve, then an fp64 finish that evaluates only a and a' in fp64 (D1s) and
// takes dp/drho for its Newton steps and for the second-order term of p* from the fp32 pass -- the Jacobian only
// steers the step (error ~1e-3 of a ~1e-6 step); the residuals and p* are fp64.  Lanes without a usable fp32
// result return ST_FALLBACK (the all-fp64 path runs on them in a separate small kernel, which keeps this one at
// 122 VGPRs / four waves per SIMD), lanes that fail afterwards ST_RETRY (robust pass).
// vle_fast_lite = fp32 pre-solve (pure_f32.hpp) + vle_lite_finish; k_pure_vle<true> runs the two parts itself with the
// block-level straggler exchange in between.
// POLISH (densities requested: equilibrium_liquid_density, rho_vl for the Jacobians): one more update of both densities
// with the exact dp/drho of an fp64 D2 evaluation at the converged state -- the iteration above leaves them at ~1e-9
// (linear convergence with the fp32 slope), the exact Newton step squares that.  p* then carries the exact second-order term.
template <bool POLISH = false>
PCS_DEV int vle_lite_finish(const double* par, double T, bool warm, double rl, double rv, float dpl32, float dpv32,
                            VleResult& out, double tol_l = TOL_L_P, double tol_v = TOL_V_P) {
    PureCoef<double> c;
    pure_coef<double>(c, par, T, false);
    bool active = warm && is_finite_bits(rv) && (rv < 0.7 * rl) && (dpl32 > 0.0f) && (dpv32 > 0.0f);
    bool done = false;
    out.iters = 0;
    Eval le, ve;
    le.dp = (double)dpl32;
    ve.dp = (double)dpv32;
    // at most LITE_MAX_IT iterations: with the fp32 dp/drho the iteration converges linearly, fast (ratio ~1e-3) on
    // ordinary rows but slowly close to the critical point where dp/drho -> 0 -- those rows go to the all-fp64 path
    for (int it = 0; it < LITE_MAX_IT; it++) {
        if (active && !done) {
            D1s al = pure_a<double, D1s>(c, D1s(rl, 1.0));
            D1s av = pure_a<double, D1s>(c, D1s(rv, 1.0));
            le.a = al.v; le.p = rl - al.v + rl * al.d1;
            ve.a = av.v; ve.p = rv - av.v + rv * av.d1;
            VleStep s = vle_step(le, ve, rl, rv);
            bool ok = is_finite_bits(s.p_star) && is_finite_bits(s.dl) && is_finite_bits(s.dv);
            double rl_new = rl + s.dl, rv_new = rv + s.dv;
            ok = ok && (rl_new > 0.0) && (rv_new > 0.0) && (rv_new < rl_new);
            if (!ok) {
                active = false;
            } else {
                // (Round 2 A/B, tests/tools/lite_accuracy_ab.py: additionally requiring the second-order term of p* to be below
                // 1e-7 ... 1e-9 p* changes neither the maximum nor the 99.99 % quantile of the error against the long-double
                // oracle -- the remaining 2e-10 on a handful of rows per 1e6 is the fp64 conditioning of the association
                // term on strongly associating rows far below the triple point, not the stop criterion.)
                done = (fabs(s.dl) <= tol_l * rl) && (fabs(s.dv) <= tol_v * rv);
                rl = rl_new;
                rv = rv_new;
                out.rho_v = rv;
                out.rho_l = rl;
                out.p_star = s.p_corr;
                out.iters = it + 1;
            }
        }
        if (__ballot(active && !done) == 0ull) break;
    }
    if (!warm || (active && !done)) return ST_FALLBACK;
    if (POLISH) {
        // accepted when the exact step is below 1e-7 (liquid) / 1e-6 (vapour): the updated densities are then converged to
        // the square of that.  A lane whose linear iteration stopped further out (slow ratio close to the critical point)
        // takes the all-fp64 iteration instead (ST_FALLBACK: k_pure_vle_fallback): no second step for the whole wave
        if (done) {
            const Eval pl = pure_eval(c, rl), pv = pure_eval(c, rv);
            const VleStep s = vle_step(pl, pv, rl, rv);
            rl += s.dl;
            rv += s.dv;
            if (!(is_finite_bits(s.p_corr) && fabs(s.dl) <= 1e-7 * rl && fabs(s.dv) <= 1e-6 * rv)) return ST_FALLBACK;
            out.rho_l = rl;
            out.rho_v = rv;
            out.p_star = s.p_corr;
            out.iters++;
        }
    }
    if (done && out.rho_v < 0.7 * out.rho_l && vapour_is_physical(out.p_star, out.rho_v)) return ST_OK;
    return ST_RETRY;
}

template <bool POLISH = false>
PCS_DEV int vle_fast_lite(const double* par, double T, VleResult& out, double tol_l = TOL_L_P, double tol_v = TOL_V_P) {
    double rl = 0.0, rv = 0.0;
    float dpl32 = 1.0f, dpv32 = 1.0f;
    PureCoefF cf;
    pure_coef_f32(cf, par, T);
    const bool warm = vle_presolve_f32(cf, rl, rv, dpl32, dpv32);
    return vle_lite_finish<POLISH>(par, T, warm, rl, rv, dpl32, dpv32, out, tol_l, tol_v);
}
#endif

// ---------------------------------------------------------------------------------------------
// Robust path (rare rows: near-critical temperatures, strongly non-ideal vapour).  Lane-serial
// bisections; no wave-uniform tricks needed because it runs on a compacted list of few rows.
// ---------------------------------------------------------------------------------------------

PCS_DEV double branch_solve(const PureCoef<double>& c, double p_spec, double lo, double hi, double rho) {
    for (int it = 0; it < 60; it++) {
        Eval e = pure_eval(c, rho);
        if (e.p > p_spec) hi = rho; else lo = rho;
        double rho_new = gt0(e.dp) ? rho - (e.p - p_spec) / e.dp : -1.0;
        if (!(rho_new > lo && rho_new < hi)) rho_new = 0.5 * (lo + hi);
        double diff = fabs(rho_new - rho);
        rho = rho_new;
        if (diff <= 1e-13 * rho) break;
    }
    return rho;
}

PCS_DEV double spinodal_bisect(const PureCoef<double>& c, double lo, double hi, bool dp_positive_at_lo) {
    for (int it = 0; it < 26; it++) {  // 2^-26 of the bracket: the spinodals only seed the branch solves
        double mid = 0.5 * (lo + hi);
        Eval e = pure_eval(c, mid);
        if (gt0(e.dp) == dp_positive_at_lo) lo = mid; else hi = mid;
    }
    return 0.5 * (lo + hi);
}

PCS_DEV int vle_robust(const PureCoef<double>& c, VleResult& out, double tol_l = TOL_STEP) {
    double rl, rv;
    Eval el;
    // 1. try the zero-pressure liquid with a tight tolerance and pull the vapour guess back
    //    onto the stable vapour branch by halving
    bool have_init = false;
    {
        double rho = ETA_START / c.ceta;
        bool ok = true;
        for (int it = 0; it < 40; it++) {
            Eval e = pure_eval(c, rho);
            if (!gt0(e.dp) || !is_finite_bits(e.p)) { ok = false; break; }
            double step = e.p / e.dp;
            double rho_new = rho - step;
            if (!gt0(rho_new)) { ok = false; break; }
            bool conv = fabs(step) <= 1e-8 * rho;
            rho = rho_new;
            if (conv) break;
        }
        if (ok) {
            double mu;
            Eval e = pure_eval_mu(c, rho, mu);
            if (gt0(e.dp)) {
                rl = rho;
                rv = rl * exp(mu);
                for (int k = 0; k < 60; k++) {
                    Eval v = pure_eval(c, rv);
                    if (gt0(v.dp) && gt0(v.p) && rv < 0.5 * rl) { have_init = true; break; }
                    rv *= 0.5;
                }
            }
        }
    }
    // 2. spinodal initialisation
    if (!have_init) {
        double rho = ETA_START / c.ceta, rho_stable = rho;
        bool found = false;
        // geometric scans (this pass is latency-bound by its slowest row: keep the evaluation count low)
        for (int k = 0; k < 60; k++) {
            Eval e = pure_eval(c, rho);
            if (!gt0(e.dp)) { found = true; break; }
            rho_stable = rho;
            rho *= 0.9;
        }
        if (!found) return ST_FAILED;  // super-critical
        double rho_sl = spinodal_bisect(c, rho, rho_stable, false);
        double rho_unstable = rho;
        found = false;
        for (int k = 0; k < 120; k++) {
            rho *= 0.8;
            Eval e = pure_eval(c, rho);
            if (gt0(e.dp)) { found = true; break; }
            rho_unstable = rho;
        }
        if (!found) return ST_FAILED;
        double rho_sv = spinodal_bisect(c, rho, rho_unstable, true);
        double p_sl = pure_eval(c, rho_sl).p, p_sv = pure_eval(c, rho_sv).p;
        if (!gt0(p_sv)) return ST_FAILED;
        double p0 = 0.5 * ((p_sl > 0.0 ? p_sl : 0.0) + p_sv);
        double hi = 0.6 / c.ceta;
        rl = branch_solve(c, p0, rho_sl, hi, 0.5 * (rho_sl + hi));
        rv = branch_solve(c, p0, 0.0, rho_sv, 0.5 * rho_sv);
    }
    // 3. coupled Newton with backtracking onto the stable branches
    double err_prev = 1.0;
    for (int it = 0; it < 60; it++) {
        Eval l = pure_eval(c, rl);
        Eval v = pure_eval(c, rv);
        VleStep s = vle_step(l, v, rl, rv);
        double dl = s.dl, dv = s.dv;
        double rl_new = rl + dl, rv_new = rv + dv;
        bool damped = false;
        for (int k = 0; k < 12; k++) {
            if (gt0(rl_new) && gt0(pure_eval(c, rl_new).dp)) break;
            dl *= 0.5;
            rl_new = rl + dl;
            damped = true;
        }
        for (int k = 0; k < 12; k++) {
            if (gt0(rv_new) && gt0(pure_eval(c, rv_new).dp)) break;
            dv *= 0.5;
            rv_new = rv + dv;
            damped = true;
        }
        double err = fmax(fabs(dl) / rl * (TOL_STEP / tol_l), fabs(dv) / rv);
        rl = rl_new;
        rv = rv_new;
        out.rho_v = rv;
        out.rho_l = rl;
        out.p_star = s.p_corr;
        out.iters = it + 1;
        if (!is_finite_bits(rl) || !is_finite_bits(rv)) return ST_FAILED;
        bool stagnated = it >= 3 && err < 1e-7 && err >= 0.25 * err_prev;
        err_prev = err;
        if (!damped && (err <= TOL_STEP || stagnated)) {
            if (!(rv < rl * (1.0 - 1e-6))) return ST_FAILED;  // trivial solution
            if (!vapour_is_physical(out.p_star, rv)) return ST_FAILED;
            // the vapour root must lie on the branch that starts at zero density: no mechanically unstable state below it
            double probe = rv;
            for (int k = 0; k < 8; k++) {
                probe *= 0.5;
                if (!gt0(pure_eval(c, probe).dp)) return ST_FAILED;
            }
            return ST_OK;
        }
    }
    return ST_FAILED;
}

}  // namespace pcs
