// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
//
// Bubble- / dew-point iteration for binary mixtures.  In the reference this is feos'
// PhaseEquilibrium::bubble_point / dew_point (src/pcsaft.rs:170-178, :203-211; src/gc_pcsaft.rs:
// 124-133, :159-168) — third-party crate feos = "0.6" (Cargo.toml:18), absent from
// /root/reference.  Restated from the published problem statement: at fixed T and fixed
// composition z of the SPECIFIED phase (liquid for bubble, vapour for dew) find the total
// density of that phase and the two partial densities of the INCIPIENT phase such that
//     mu_i^spec = mu_i^inc (i = 1, 2),   p^spec = p^inc,
// mu_i = ln rho_i + da/drho_i,  p = sum rho - a + sum rho_k da/drho_k   (all reduced).
// Newton in the logarithms of the three unknowns; initialisation from ideal-gas / ideal-
// solution estimates (bubble: liquid at the initial pressure, vapour partial densities from
// the liquid fugacities; dew: Raoult's law with zero-pressure pure-liquid fugacities).
// The result the reference returns is one further explicit Newton step
// (feos_torch/pcsaft_mix.py:443, :467) whose density derivatives vanish at the solution, so it
// depends on the Helmholtz model only.
//
// Two attempts per row, as in the kernels (csrc/mix_solver_sm.hpp): the plain form first; a row it gives up on is solved
// again with `robust` = bracketed liquid roots (liquid_root_bracketed).  Which failed rows have a solution at all is judged
// by the independent second solver, mix_continuation.hpp.
//
// Model-agnostic: `Model` provides  template<class S> S a(const S& T, const S* rho) const
// (reduced residual Helmholtz energy density) and  F packing(F T, const F* x)  =
// zeta3 / rho_total at composition x.  Used for PcSaftMix and GcPcSaftMix.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "dual.hpp"

namespace oracle {

template <class F>
struct PhaseEval {
    F a, g[2], h[2][2];  // a, da/drho_i, d2a/drho_i drho_j
    F rho[2];
    F mu(int i) const { return log(rho[i]) + g[i]; }
    F mu_res(int i) const { return g[i]; }
    F p() const { return rho[0] + rho[1] - a + rho[0] * g[0] + rho[1] * g[1]; }
    F dmu(int i, int j) const { return (i == j ? F(1) / rho[i] : F(0)) + h[i][j]; }
    F dp(int j) const { return F(1) + rho[0] * h[0][j] + rho[1] * h[1][j]; }
};

// full gradient + Hessian from two hyper-dual passes (eps2 along rho_1, then along rho_2)
template <class F, class Model>
PhaseEval<F> eval_phase(const Model& model, F T, const F* rho) {
    typedef HyperDual<F, 2> H;
    PhaseEval<F> e;
    e.rho[0] = rho[0];
    e.rho[1] = rho[1];
    for (int k = 0; k < 2; k++) {
        H r[2];
        for (int i = 0; i < 2; i++) {
            r[i].re = rho[i];
            r[i].eps1[i] = F(1);
            if (i == k) r[i].eps2 = F(1);
        }
        H Th;
        Th.re = T;
        H A = model.template a<H>(Th, r);
        e.a = A.re;
        e.g[0] = A.eps1[0];
        e.g[1] = A.eps1[1];
        e.h[0][k] = A.eps1eps2[0];
        e.h[1][k] = A.eps1eps2[1];
    }
    return e;
}

// dense-side Newton for p(rho_total) = p_spec at fixed composition x (liquid-like root).
// The iteration runs on (p - p_spec)(1 - eta)^4 = 0 (same root; the hard-sphere pole makes p steep,
// the scaled function is nearly linear).  Only used to initialise the phase-equilibrium Newton, so
// a relative step of LIQ_ROOT_TOL suffices.  Same logic and caps as csrc/mix_solver.hpp.
constexpr double LIQ_ROOT_TOL = 1e-3;  // as csrc/mix_solver.hpp
constexpr int NEWTON_NO_PROGRESS = 20, NEWTON_NO_PROGRESS_BUBBLE = 15;  // as csrc/mix_solver.hpp
constexpr double NEWTON_PROGRESS = 0.9;
constexpr double NEWTON_TRACE = 1e-4, NEWTON_TRACE_MAX = 100.0;
constexpr double SS_RES_TOL = 1e-2;     // as csrc/mix_solver.hpp
constexpr double SS_SECANT_SLOPE = -1e-5;  // as csrc/mix_solver.hpp: the secant step wherever the map's residual decreases along xi
constexpr int NEWTON_DAMPED_MAX_IT = 24, NEWTON_DAMPED_HALVINGS = 4;  // as csrc/mix_solver.hpp
constexpr double NEWTON_FLOOR = 1e-6;  // as csrc/mix_solver.hpp: a step that has stopped shrinking below it sits on the rounding floor and is accepted
constexpr double SS_TOL = 1e-5;  // composition change at which the dew-point successive substitution hands over to Newton
// Robust form of the liquid root (second pass of the solvers, csrc/mix_solver.hpp: ROBUST): a bracket [lo, hi] with
// p(lo) < p_spec < p(hi), dp(hi) > 0 is established first -- hi = the first of eta = 0.5, 0.62, 0.70, 0.78, 0.86 at which p
// exceeds p_spec, lo = the candidate below it (or, for eta = 0.5, found by halving the density until p < p_spec or dp < 0) --
// then Newton on the scaled function, safeguarded by bisection.  Needed for very cold heavy components (T/Tc < 0.25): there
// the PC-SAFT pressure is not monotone-convex above eta = 0.5 and the plain Newton of the fast form jumps over the root.
template <class F, class Model>
bool liquid_root_bracketed(const Model& model, F T, const F* x, F p_spec, F& rho_out) {
    // restated 1:1 from the `robust` branch of BdLane::consume, stage S_ROOT (csrc/mix_solver_sm.hpp): same candidates, same
    // bracket updates, same order of evaluations -- on rows whose pressure is not monotone inside the bracket the root that
    // is found depends on them
    const F pk = model.packing(T, x);
    auto ev = [&](F rho, F& p, F& dp) {
        F r[2] = {x[0] * rho, x[1] * rho};
        PhaseEval<F> e = eval_phase<F>(model, T, r);
        p = e.p();
        dp = x[0] * e.dp(0) + x[1] * e.dp(1);
    };
    const double cand[5] = {0.5, 0.62, 0.70, 0.78, 0.86};
    F lo = F(0), hi = F(0), p, dp, rho = F(cand[0]) / pk;
    int phase = 0, k = 0, it = 0;
    for (int guard = 0; guard < 200; guard++) {
        ev(rho, p, dp);
        if (!(p == p)) return false;
        const bool above = (p > p_spec) && (dp > 0);
        if (phase == 0) {
            if (above) {
                hi = rho;
                if (k == 0) { phase = 1; lo = rho; it = 0; rho = F(0.8) * rho; continue; }
                phase = 2;
            } else {
                lo = rho;
                if (++k >= 5) return false;
                rho = F(cand[k]) / pk;
                continue;
            }
        } else if (phase == 1) {
            if (!above) { lo = rho; phase = 2; it = 0; rho = F(0.5) * (lo + hi); continue; }
            hi = rho;
            if (++it >= 12) return false;
            rho = F(0.8) * rho;
            continue;
        }
        if (above) hi = rho; else lo = rho;
        F den = dp - F(4) * (p - p_spec) * pk / (F(1) - rho * pk);
        F rho_new = (dp > 0 && den > 0) ? rho - (p - p_spec) / den : F(-1);
        const bool newton = rho_new > lo && rho_new < hi;
        if (!newton) rho_new = F(0.5) * (lo + hi);
        F step = rho - rho_new;
        const bool done = (newton && (step < 0 ? -step : step) <= F(LIQ_ROOT_TOL) * rho) || (hi - lo) <= F(1e-12) * hi;
        it++;
        rho = rho_new;
        if (done) { rho_out = rho; return true; }
        if (it >= 60) return false;
    }
    return false;
}

template <class F, class Model>
bool liquid_root(const Model& model, F T, const F* x, F p_spec, F& rho_out, F rho_start = F(0), bool robust = false) {
    if (robust) return liquid_root_bracketed<F>(model, T, x, p_spec, rho_out);  // second pass: no warm starts
    static const bool plain = getenv("ORC_LIQ_PLAIN") != nullptr;
    static const double tol = getenv("ORC_LIQ_TOL") ? atof(getenv("ORC_LIQ_TOL")) : LIQ_ROOT_TOL;
    F pk = model.packing(T, x);
    // warm start (re-solves during the dew-point substitution); a warm start that misbehaves falls back to the cold one
    const bool warm = rho_start > F(0) && rho_start * pk < F(0.7);
    F rho = warm ? rho_start : F(0.5) / pk;
    F err_prev = F(1);
    bool dense = false;
    for (int it = 0; it < 30; it++) {  // same cap as the kernels
        F r[2] = {x[0] * rho, x[1] * rho};
        PhaseEval<F> e = eval_phase<F>(model, T, r);
        F p = e.p(), dp = x[0] * e.dp(0) + x[1] * e.dp(1);
        if (it == 0 && !warm && !(p > p_spec)) { rho = F(0.62) / pk; dense = true; continue; }
        if (getenv("ORC_TRACE_LIQ")) fprintf(stderr, "  liq it %d eta %.5f p %.4e dp %.4e p_spec %.3e\n", it, (double)(rho * pk), (double)p, (double)dp, (double)p_spec);
        F den = (dense || plain) ? dp : dp - F(4) * (p - p_spec) * pk / (F(1) - rho * pk);
        F step = (p - p_spec) / den;
        F rho_new = rho - step;
        // dense restart: an iterate that leaves (0.5, 0.62) is on its way to another liquid-like root -> robust form
        // (as csrc/mix_solver_sm.hpp)
        const bool left = dense && !(rho_new * pk > F(0.5) && rho_new * pk < F(0.62));
        if (left || !(dp > 0) || !(p == p) || !(den > 0) || !(rho_new > 0) || !(rho_new == rho_new)) {
            if (warm) return liquid_root<F>(model, T, x, p_spec, rho_out);
            return false;
        }
        F err = (step < 0 ? -step : step) / rho;
        bool done = err <= F(tol) || (it >= 3 && err < F(1e-7) && err >= F(0.25) * err_prev);
        err_prev = err;
        rho = rho_new;
        if (done) { rho_out = rho; return true; }
    }
    if (warm) return liquid_root<F>(model, T, x, p_spec, rho_out);
    return false;
}

template <class F>
bool solve3(F J[3][3], const F* b, F* x) {  // Gaussian elimination with partial pivoting
    F A[3][4];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = J[i][j]; A[i][3] = b[i]; }
    for (int c = 0; c < 3; c++) {
        int piv = c;
        for (int r = c + 1; r < 3; r++) if (fabsl((long double)A[r][c]) > fabsl((long double)A[piv][c])) piv = r;
        if (A[piv][c] == 0) return false;
        for (int j = 0; j < 4; j++) { F t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        for (int r = c + 1; r < 3; r++) {
            F f = A[r][c] / A[c][c];
            for (int j = c; j < 4; j++) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 2; i >= 0; i--) {
        F s = A[i][3];
        for (int j = i + 1; j < 3; j++) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    return true;
}

struct MixSolveInfo {
    int iters;
    bool root_failed = false;  // a cold liquid root of this attempt failed (the only failure the robust attempt repairs)
};

// z = mole fraction of component 1 in the specified phase; p_init [reduced] = caller's initial
// pressure (src/pcsaft.rs:174 passes it to feos as Some(p)).  Outputs partial densities.
// The Newton stage of bubble_dew from (rs, ri): in (ln rho_spec, ln rho_inc_1, ln rho_inc_2).  damped: with the natural
// monotonicity test (second attempt of a row whose plain iteration failed, see bubble_dew)
template <class F, class Model>
bool newton_stage(const Model& model, F T, const F* z, bool dew, F rs, F ri0, F ri1, F* rho_spec, F* rho_inc, MixSolveInfo& info,
                  F tol, bool damped, int np_limit = 0) {
    F ri[2] = {ri0, ri1};
    const bool bt_on = damped;
    // Newton in (ln rho_spec, ln rho_inc_1, ln rho_inc_2)
    F err_prev = F(1), err_best = F(1e300);
    int it_best = 0;
    const int no_progress = getenv("ORC_NP") ? atoi(getenv("ORC_NP")) : (np_limit > 0 ? np_limit : (dew ? NEWTON_NO_PROGRESS : NEWTON_NO_PROGRESS_BUBBLE));
    F m_prev = F(1e300), st_prev[3] = {F(0), F(0), F(0)};
    int n_bt = 0;
    for (int it = 0; it < (damped ? NEWTON_DAMPED_MAX_IT : 60); it++) {
        F r_s[2] = {z[0] * rs, z[1] * rs};
        PhaseEval<F> s = eval_phase<F>(model, T, r_s);
        PhaseEval<F> n = eval_phase<F>(model, T, ri);
        F Fv[3] = {s.mu(0) - n.mu(0), s.mu(1) - n.mu(1), s.p() - n.p()};
        F J[3][3];
        for (int i = 0; i < 2; i++) {
            J[i][0] = rs * (z[0] * s.dmu(i, 0) + z[1] * s.dmu(i, 1));
            J[i][1] = -ri[0] * n.dmu(i, 0);
            J[i][2] = -ri[1] * n.dmu(i, 1);
        }
        J[2][0] = rs * (z[0] * s.dp(0) + z[1] * s.dp(1));
        J[2][1] = -ri[0] * n.dp(0);
        J[2][2] = -ri[1] * n.dp(1);
        F rhs[3] = {-Fv[0], -Fv[1], -Fv[2]}, du[3];
        if (!solve3<F>(J, rhs, du)) return false;
        F mx = 0;
        for (int k = 0; k < 3; k++) { F a = du[k] < 0 ? -du[k] : du[k]; if (a > mx) mx = a; }
        if (!(mx == mx)) return false;
        if (bt_on) {
            const double bt_theta = 1.0;
            const int bt_max = NEWTON_DAMPED_HALVINGS;
            // natural monotonicity test: the Newton correction at the new point is larger than the one that led there --
            // the step overshot: take half of it back and look again
            if (it > 0 && mx > F(bt_theta) * m_prev && n_bt < bt_max && m_prev > F(1e-3)) {
                for (int k = 0; k < 3; k++) st_prev[k] = F(0.5) * st_prev[k];
                rs = rs * exp(-st_prev[0]);
                ri[0] = ri[0] * exp(-st_prev[1]);
                ri[1] = ri[1] * exp(-st_prev[2]);
                n_bt++;
                if (getenv("ORC_TRACE")) fprintf(stderr, "it %d backtrack %d mx %.3e > %.3e\n", it, n_bt, (double)mx, (double)m_prev);
                continue;
            }
            m_prev = mx;
            n_bt = 0;
        }
        // no new smallest Newton step for NEWTON_NO_PROGRESS iterations: the iteration cycles / wanders
        // (no phase equilibrium at this state, or the EOS is ill-behaved there) -> fail now, not at the cap
        static const double np_factor = getenv("ORC_NP_FACTOR") ? atof(getenv("ORC_NP_FACTOR")) : NEWTON_PROGRESS;
        if (mx < F(np_factor) * err_best) { err_best = mx; it_best = it; }
        else if (it - it_best >= no_progress) return false;
        // at most a factor e per iteration -- except for a trace component of the incipient phase (mole fraction
        // below NEWTON_TRACE): its chemical potential is linear in ln rho_i there (ideal dilution), the Newton step
        // lands on the solution however long it is, and limiting it would only make the iteration march
        static const bool trace_rule = getenv("ORC_NO_TRACE") == nullptr;
        F rtot = ri[0] + ri[1];
        bool tr0 = trace_rule && ri[0] < F(NEWTON_TRACE) * rtot, tr1 = trace_rule && ri[1] < F(NEWTON_TRACE) * rtot;
        F mxl = du[0] < 0 ? -du[0] : du[0];
        if (!tr0) { F a = du[1] < 0 ? -du[1] : du[1]; if (a > mxl) mxl = a; }
        if (!tr1) { F a = du[2] < 0 ? -du[2] : du[2]; if (a > mxl) mxl = a; }
        F scale = mxl > F(1) ? F(1) / mxl : F(1);
        F s0 = scale * du[1], s1 = scale * du[2];
        if (tr0) s0 = du[1] > F(NEWTON_TRACE_MAX) ? F(NEWTON_TRACE_MAX) : (du[1] < F(-NEWTON_TRACE_MAX) ? F(-NEWTON_TRACE_MAX) : du[1]);
        if (tr1) s1 = du[2] > F(NEWTON_TRACE_MAX) ? F(NEWTON_TRACE_MAX) : (du[2] < F(-NEWTON_TRACE_MAX) ? F(-NEWTON_TRACE_MAX) : du[2]);
        rs = rs * exp(scale * du[0]);
        ri[0] = ri[0] * exp(s0);
        ri[1] = ri[1] * exp(s1);
        st_prev[0] = scale * du[0]; st_prev[1] = s0; st_prev[2] = s1;
        if (getenv("ORC_TRACE")) fprintf(stderr, "it %d mx %.3e du %.3e %.3e %.3e rs %.6e ri %.6e %.6e F %.3e %.3e %.3e\n", it, (double)mx, (double)du[0], (double)du[1], (double)du[2], (double)rs, (double)ri[0], (double)ri[1], (double)Fv[0], (double)Fv[1], (double)Fv[2]);
        info.iters = it + 1;
        {   // collapsed onto the trivial solution (both phases identical, singular Jacobian): give up (as csrc/mix_solver.hpp)
            F dtot = ri[0] + ri[1] - rs, d0 = ri[0] - z[0] * rs;
            if ((dtot < 0 ? -dtot : dtot) <= F(1e-6) * rs && (d0 < 0 ? -d0 : d0) <= F(1e-6) * rs) return false;
        }
        bool stagnated = it >= 3 && mx < F(NEWTON_FLOOR) && mx >= F(0.25) * err_prev;
        err_prev = mx;
        if (mx <= tol || stagnated) {
            F dens_s = rs, dens_i = ri[0] + ri[1];
            F lo = dew ? dens_s : dens_i, hi = dew ? dens_i : dens_s;  // vapour, liquid
            if (!(lo < hi * (F(1) - F(1e-6)))) return false;           // trivial solution
            rho_spec[0] = z[0] * rs;
            rho_spec[1] = z[1] * rs;
            rho_inc[0] = ri[0];
            rho_inc[1] = ri[1];
            return true;
        }
    }
    return false;
}

template <class F, class Model>
bool bubble_dew(const Model& model, F T, F z1, F p_init, bool dew, F* rho_spec, F* rho_inc, MixSolveInfo& info, F tol,
                bool robust = false, bool may_damp = true, int np_limit = 0) {
    F z[2] = {z1, F(1) - z1};
    F rs, ri[2];  // total density of the specified phase, partial densities of the incipient one
    info.iters = 0;
    if (!dew) {
        // liquid at the initial pressure (fallback: zero pressure), ideal vapour at its fugacities
        if (!liquid_root<F>(model, T, z, p_init, rs, F(0), robust)) {
            info.root_failed = true;
            if (!liquid_root<F>(model, T, z, F(0), rs, F(0), robust)) return false;
        }
        F r[2] = {z[0] * rs, z[1] * rs};
        PhaseEval<F> e = eval_phase<F>(model, T, r);
        // (no stability test of the specified liquid any more: as csrc/mix_solver_sm.hpp, round 3)
        for (int i = 0; i < 2; i++) ri[i] = r[i] * exp(e.g[i]);
    } else {
        // Raoult: zero-pressure pure-liquid fugacities f_i, p = 1/sum(y_i/f_i), x_i = y_i p/f_i
        F f[2], rho_pure[2] = {F(0), F(0)};
        bool ok = true;
        for (int i = 0; i < 2; i++) {
            F xi[2] = {i == 0 ? F(1) : F(0), i == 1 ? F(1) : F(0)};
            F rho0;
            if (!liquid_root<F>(model, T, xi, F(0), rho0, F(0), robust)) {
                info.root_failed = true;
                if (!robust) return false;  // plain form: straight to the robust attempt (as csrc/mix_solver_sm.hpp)
                ok = false;
                break;
            }
            F r[2] = {xi[0] * rho0, xi[1] * rho0};
            PhaseEval<F> e = eval_phase<F>(model, T, r);
            f[i] = rho0 * exp(e.g[i]);
            rho_pure[i] = rho0;
        }
        F x[2], p0;
        if (ok) {
            p0 = F(1) / (z[0] / f[0] + z[1] / f[1]);
            x[0] = z[0] * p0 / f[0];
            x[1] = z[1] * p0 / f[1];
        } else {
            p0 = p_init;
            x[0] = z[0];
            x[1] = z[1];
        }
        // successive substitution with an ideal vapour: the liquid sits at (nearly) zero pressure,
        // f_i = rho^L_i exp(da/drho_i) are its fugacities, y_i p = f_i fixes the next composition:
        //   x_i <- (y_i x_i / f_i) / sum_j (y_j x_j / f_j),   p = 1 / sum_j (y_j x_j / f_j)
        F rl = 0;
        bool have = false;
        if (ok && !robust) {
            // plain form: the liquid at the Raoult composition starts at the ideal-mixing (Amagat) density of the two pure
            // liquids and goes straight into the first sweep (as csrc/mix_solver_sm.hpp::raoult): the sweep's own Newton step
            // carries it to the zero-pressure root, or asks for the root from there
            F am = F(1) / (x[0] / rho_pure[0] + x[1] / rho_pure[1]);
            if (am > 0 && am == am && am * F(2) != am) { rl = am; have = true; }
        }
        static const bool ss_secant = getenv("ORC_SS_PLAIN") == nullptr;
        static const bool ss_track = getenv("ORC_SS_NOTRACK") == nullptr;
        F xi_prev = 0, res_prev = 0;
        // bracket of the fixed point in xi: r > 0 at xi_lo, r < 0 at xi_hi (r decreases through a stable fixed point)
        static const bool ss_bracket = getenv("ORC_SS_NOBRACKET") == nullptr;
        F xi_lo = F(-1e300), xi_hi = F(1e300);
        static const int ss_cap = getenv("ORC_SS_CAP") ? atoi(getenv("ORC_SS_CAP")) : 40;
        static const double ss_tol = getenv("ORC_SS_TOL") ? atof(getenv("ORC_SS_TOL")) : SS_TOL;
        for (int ss = 0; ss < ss_cap; ss++) {  // same caps as the kernels (csrc/mix_solver.hpp)
            // The liquid density is not re-solved in every sweep: the evaluation at (x, rl) gives p and dp/drho along x,
            // i.e. the Newton step drho to the zero-pressure root, and the chemical potentials are carried to that
            // root to first order with the Hessian.  A full root solve is done at the start and whenever the step is
            // not small (composition moved a lot) or the linearisation is unusable.
            PhaseEval<F> e;
            F drho = F(0);
            bool fine_prev = false;
            for (int attempt = 0; attempt < 2; attempt++) {
                if (!have || attempt == 1) {
                    // a re-solve starts from the tracked density when the evaluation there was usable
                    const F warm_rho = (have && attempt == 1 && fine_prev) ? rl : F(0);
                    bool root_ok = liquid_root<F>(model, T, x, F(0), rl, warm_rho, robust);
                    if (!root_ok) { info.root_failed = true; root_ok = liquid_root<F>(model, T, x, p0, rl, F(0), robust); }
                    if (!root_ok) {
                        if (getenv("ORC_TRACE")) fprintf(stderr, "FAIL ss-liquid-root ss %d x %.6e %.6e p0 %.6e\n", ss, (double)x[0], (double)x[1], (double)p0);
                        return false;
                    }
                    have = true;
                }
                F r[2] = {x[0] * rl, x[1] * rl};
                e = eval_phase<F>(model, T, r);
                F p = e.p(), dp = x[0] * e.dp(0) + x[1] * e.dp(1);
                drho = -p / dp;
                F ad = drho < 0 ? -drho : drho;
                fine_prev = dp > 0 && p == p;
                if (ss_track && dp > 0 && p == p && ad <= F(0.05) * rl) break;
                if (attempt == 1) { if (!(dp > 0) || !(p == p)) return false; if (!(ad <= F(0.05) * rl)) drho = F(0); }
            }
            F rlc = rl + drho;
            // w_i = z_i / (rho exp(G_i)) with the smaller exponent factored out (as csrc/mix_solver.hpp): far from the
            // solution G_i exceeds the range of exp although only the ratio of the weights and 1/sum are needed
            F G[2], w[2];
            for (int i = 0; i < 2; i++) G[i] = e.g[i] + (x[0] * e.h[i][0] + x[1] * e.h[i][1]) * drho;
            F Gm = G[0] < G[1] ? G[0] : G[1];
            for (int i = 0; i < 2; i++) w[i] = z[i] * exp(Gm - G[i]);
            rl = rlc;
            F sum = (w[0] + w[1]) / (rlc * exp(Gm));
            F xn[2] = {w[0] / (w[0] + w[1]), w[1] / (w[0] + w[1])};
            F dx = xn[0] - x[0];
            if (dx < 0) dx = -dx;
            // The sweep is a scalar fixed-point map xi -> G(xi) in xi = ln(x_1/x_2); its plain iteration converges
            // linearly (slowly for strongly non-ideal liquids), so from the second sweep on the secant step on
            // r(xi) = G(xi) - xi is taken when it is well defined (r decreasing, step at most ln 5).
            F xi = log(x[0] / x[1]);
            F res = log(xn[0] / xn[1]) - xi;
            bool secant = false;
            if (ss_bracket) {
                if (res > F(0) && xi > xi_lo) xi_lo = xi;
                if (res < F(0) && xi < xi_hi) xi_hi = xi;
            }
            if (ss_secant && ss > 0 && xi != xi_prev) {
                F slope = (res - res_prev) / (xi - xi_prev);
                static const double ss_slope = getenv("ORC_SS_SLOPE") ? atof(getenv("ORC_SS_SLOPE")) : SS_SECANT_SLOPE;
                if (slope < F(ss_slope)) {
                    F dxi = -res / slope;
                    if (dxi > F(1.6)) dxi = F(1.6);
                    if (dxi < F(-1.6)) dxi = F(-1.6);
                    F e = exp(xi + dxi);
                    xi_prev = xi;
                    res_prev = res;
                    x[0] = e / (F(1) + e);
                    x[1] = F(1) / (F(1) + e);
                    secant = true;
                }
            }
            if (!secant) {
                xi_prev = xi;
                res_prev = res;
                // damp when a component would change by more than a factor 5 in one sweep
                for (int i = 0; i < 2; i++) {
                    if (xn[i] > F(5) * x[i]) xn[i] = F(5) * x[i];
                    if (xn[i] < F(0.2) * x[i]) xn[i] = F(0.2) * x[i];
                }
                F s2 = xn[0] + xn[1];
                x[0] = xn[0] / s2;
                x[1] = xn[1] / s2;
            }
            bool narrow = false;
            if (ss_bracket && xi_lo < xi_hi && xi_lo > F(-1e299) && xi_hi < F(1e299)) {
                // the map cycles around a steep (or discontinuous: the liquid root changes branch) stretch of r(xi):
                // an iterate outside the bracket is replaced by its midpoint
                F xin = log(x[0] / x[1]);
                if (!(xin > xi_lo && xin < xi_hi)) {
                    F e = exp(F(0.5) * (xi_lo + xi_hi));
                    x[0] = e / (F(1) + e);
                    x[1] = F(1) / (F(1) + e);
                }
                narrow = xi_hi - xi_lo < F(ss_tol);
            }
            p0 = F(1) / sum;
            if (getenv("ORC_TRACE")) fprintf(stderr, "ss %d x %.6e %.6e p0 %.6e rl %.6e dx %.3e\n", ss, (double)x[0], (double)x[1], (double)p0, (double)rl, (double)dx);
            // settled: the composition no longer moves -- in absolute terms AND, for a trace component, in relative ones
            // (|res| = |d ln(x_1/x_2)| of the sweep; with x_2 ~ 1e-6 falling by the damped factor 5 per sweep the absolute
            // change of x_1 drops below the tolerance long before the trace component has found its level)
            F ares = res < 0 ? -res : res;
            if ((dx < F(ss_tol) && ares < F(SS_RES_TOL)) || narrow) break;
        }
        if (!have) return false;
        if (!ss_track && !liquid_root<F>(model, T, x, p0, rl, F(0), robust) && !liquid_root<F>(model, T, x, F(0), rl, F(0), robust)) return false;
        ri[0] = x[0] * rl;
        ri[1] = x[1] * rl;
        rs = p0;  // ideal vapour
    }
    // Newton from there; a row whose iteration fails (it diverges or cycles from the ideal-vapour fixed point of the
    // substitution: a nearly critical liquid) gets a second, damped run from the same start (dew points)
    const F rs0 = rs, ri00 = ri[0], ri10 = ri[1];
    if (newton_stage<F>(model, T, z, dew, rs0, ri00, ri10, rho_spec, rho_inc, info, tol, false, np_limit)) return true;
    static const bool damped_retry = getenv("ORC_NO_DAMPED") == nullptr;
    if (damped_retry && may_damp && dew && !robust) return newton_stage<F>(model, T, z, dew, rs0, ri00, ri10, rho_spec, rho_inc, info, tol, true, np_limit);
    return false;
}

// feos_torch/pcsaft_mix.py:395-420 == feos_torch/gc_pcsaft.py:443-468 for any `Model`:
// hyper-dual pass over A(N, V) = V a(N/V): a, p = sum rho - A_V, mu_i = A_Ni,
// v_i = -(1 - A_VNi)/(-sum rho - A_VV).
template <class F, class Model>
void derivatives_generic(const Model& model, F T, const F* rho, F& a, F& p, F* mu, F* v) {
    typedef HyperDual<F, 3> H;
    auto lift = [](F x) { H h; h.re = x; return h; };
    H volume = lift(F(1));
    volume.eps1[2] = F(1);
    volume.eps2 = F(1);
    H dens[2];
    for (int i = 0; i < 2; i++) {
        H moles = lift(rho[i]);
        moles.eps1[i] = F(1);
        dens[i] = moles / volume;
    }
    H A = model.template a<H>(lift(T), dens) * volume;
    F rs = rho[0] + rho[1];
    p = rs - A.eps2;
    for (int i = 0; i < 2; i++) {
        mu[i] = A.eps1[i];
        v[i] = -(F(1) - A.eps1eps2[i]) / (-rs - A.eps1eps2[2]);
    }
    a = A.re;
}

// bubble (spec = liquid) / dew (spec = vapour) tail, pcsaft_mix.py:435-444 / :459-468 ==
// gc_pcsaft.py:481-490 / :503-512, reduced pressure
template <class F, class Model>
F bubble_dew_formula_generic(const Model& model, F T, const F* rho_spec, const F* rho_inc) {
    F rho_i = rho_inc[0] + rho_inc[1];
    F y[2] = {rho_inc[0] / rho_i, rho_inc[1] / rho_i};
    F a_s, p_s, mu_s[2], v_s[2];
    derivatives_generic<F>(model, T, rho_spec, a_s, p_s, mu_s, v_s);
    F a_i = model.template a<F>(T, rho_inc) / rho_i;
    F v = y[0] * v_s[0] + y[1] * v_s[1];
    F g = y[0] * (log(rho_inc[0] / rho_spec[0]) - mu_s[0]) + y[1] * (log(rho_inc[1] / rho_spec[1]) - mu_s[1]);
    return -(a_i + p_s * v + g - F(1)) / (F(1) / rho_i - v);
}

}  // namespace oracle
