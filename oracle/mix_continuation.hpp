// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
//
// A SECOND, algorithmically different bubble- / dew-point solver for binary mixtures, used to judge the first one
// (mix_solver.hpp, which the kernels' solver restates) on the rows it gives up on: which of them have a solution that
// the first algorithm merely misses, and which have none that a careful method can reach either.  It stands where the
// reference calls feos (src/pcsaft.rs:170-178, :203-211), like mix_solver.hpp, and shares nothing with it but the model
// evaluation (eval_phase) and the 3x3 linear solve:
//
//   * no Raoult / ideal-vapour / successive-substitution initialisation at the target composition;
//   * instead CONTINUATION IN COMPOSITION along the bubble (dew) curve at fixed T, started next to a pure-component
//     end where the problem degenerates to a pure-fluid VLE, and marched to the target composition in the variable
//     xi = ln(z1/z2) with adaptive steps (secant predictor, damped Newton corrector in the logarithms of the three
//     unknown densities, step halving on any failure);
//   * the start point comes from a fully BRACKETED pure-fluid solve along the composition line (spinodals by scanning,
//     Maxwell construction by bisection in ln p) -- no Newton from a guess;
//   * both ends are followed (from component 1 and from component 2) and, where they arrive at different solutions, the stable
//     one is returned (dew: lower pressure, bubble: higher); the curve may end before the target in a critical
//     point (phases become identical) or a liquid-liquid region (the liquid phase becomes mechanically or diffusionally
//     unstable), which is reported.
//
// Result codes: CONT_OK (solution at the target composition), CONT_NO_START (no pure-fluid VLE next to either end: T above
// both pseudo-critical temperatures), CONT_CRITICAL (every route ended in a critical point before the target),
// CONT_STALLED (a route stalled: step size underflow without a critical point; typically a limit of stability of one
// phase, i.e. a three-phase / liquid-liquid situation).
#pragma once
#include "mix_solver.hpp"

namespace oracle {

enum : int { CONT_OK = 0, CONT_NO_START = 1, CONT_CRITICAL = 2, CONT_STALLED = 3 };

struct ContInfo {
    int code = CONT_NO_START;
    int steps = 0, newton = 0;  // continuation steps / Newton iterations used
    int route = -1;             // 0: from component 1 (z1 -> 1), 1: from component 2
};

namespace cont_detail {

template <class F> F absf(F x) { return x < 0 ? -x : x; }

// p, dp/drho and the line chemical potential g = d(a_line)/drho + ln rho along the composition x (total density rho)
template <class F, class Model>
void line_point(const Model& model, F T, const F* x, F rho, F& p, F& dp, F& g) {
    F r[2] = {x[0] * rho, x[1] * rho};
    PhaseEval<F> e = eval_phase<F>(model, T, r);
    p = e.p();
    dp = x[0] * e.dp(0) + x[1] * e.dp(1);
    // Gibbs energy per particle of the line fluid (up to composition-only terms): ln rho + sum x_i da/drho_i
    g = log(rho) + x[0] * e.g[0] + x[1] * e.g[1];
}

// root of p(rho) = p_spec inside a bracket [lo, hi] on which p is increasing: safeguarded Newton
template <class F, class Model>
F branch_root(const Model& model, F T, const F* x, F p_spec, F lo, F hi) {
    F rho = F(0.5) * (lo + hi);
    for (int it = 0; it < 200; it++) {
        F p, dp, g;
        line_point<F>(model, T, x, rho, p, dp, g);
        if (p > p_spec) hi = rho; else lo = rho;
        F nr = dp > 0 ? rho - (p - p_spec) / dp : F(-1);
        if (!(nr > lo && nr < hi)) nr = F(0.5) * (lo + hi);
        F d = absf<F>(nr - rho);
        rho = nr;
        if (d <= F(1e-15) * rho) break;
    }
    return rho;
}

// Pure-fluid (fixed-composition) VLE along the line x: spinodals by scanning from the dense side, Maxwell construction
// by bisection in ln p.  Returns false if the line fluid has no van-der-Waals loop at T (supercritical).
template <class F, class Model>
bool line_vle(const Model& model, F T, const F* x, F& rho_v, F& rho_l) {
    const F pk = model.packing(T, x);
    F p, dp, g;
    // scan eta from 0.60 down to 1e-9 (geometric below 0.01): find the liquid spinodal (dp changes + -> -) and the
    // vapour spinodal (- -> +)
    F rho_hi = F(0.60) / pk;
    line_point<F>(model, T, x, rho_hi, p, dp, g);
    if (!(dp > 0)) return false;
    F rho = rho_hi, prev = rho_hi, sl_lo = 0, sl_hi = 0, sv_lo = 0, sv_hi = 0;
    bool in_loop = false, done = false;
    for (int k = 0; k < 4000 && !done; k++) {
        prev = rho;
        rho = rho * (rho * pk > F(0.02) ? F(0.985) : F(0.9));
        if (rho * pk < F(1e-12)) break;
        line_point<F>(model, T, x, rho, p, dp, g);
        if (!(p == p)) return false;
        if (!in_loop && !(dp > 0)) { in_loop = true; sl_lo = rho; sl_hi = prev; }
        else if (in_loop && dp > 0) { sv_lo = rho; sv_hi = prev; done = true; }
    }
    if (!done) return false;
    auto bisect_dp = [&](F lo, F hi, bool pos_at_lo) {  // dp changes sign between lo and hi
        for (int it = 0; it < 100; it++) {
            F mid = F(0.5) * (lo + hi), pp, dd, gg;
            line_point<F>(model, T, x, mid, pp, dd, gg);
            if ((dd > 0) == pos_at_lo) lo = mid; else hi = mid;
        }
        return F(0.5) * (lo + hi);
    };
    const F rho_sl = bisect_dp(sl_lo, sl_hi, false);  // dp <= 0 at sl_lo, > 0 at sl_hi
    const F rho_sv = bisect_dp(sv_lo, sv_hi, true);   // dp > 0 at sv_lo, <= 0 at sv_hi
    F p_sl, p_sv;
    line_point<F>(model, T, x, rho_sl, p_sl, dp, g);
    line_point<F>(model, T, x, rho_sv, p_sv, dp, g);
    if (!(p_sv > 0)) return false;
    // Maxwell: d(g_L - g_V)/dp = v_L - v_V < 0, so g_L - g_V decreases through zero at the saturation pressure; bracket
    // ln p between the liquid spinodal pressure (or, if that is negative, a pressure low enough that the vapour is the
    // stable phase) and the vapour spinodal pressure
    auto maxwell = [&](F lp, F& rl, F& rv) {
        F ps = exp(lp);
        rl = branch_root<F>(model, T, x, ps, rho_sl, rho_hi);
        rv = branch_root<F>(model, T, x, ps, F(0), rho_sv);
        F pl, dl, gl, pv, dv, gv;
        line_point<F>(model, T, x, rl, pl, dl, gl);
        line_point<F>(model, T, x, rv, pv, dv, gv);
        return gl - gv;
    };
    F lp_hi = log(p_sv), lp_lo;
    F rl, rv;
    if (p_sl > 0) {
        lp_lo = log(p_sl);
    } else {
        lp_lo = lp_hi;
        bool found = false;
        for (int k = 0; k < 900 && !found; k++) {
            lp_lo -= F(1);
            if (maxwell(lp_lo, rl, rv) > 0) found = true;
        }
        if (!found) return false;
    }
    for (int it = 0; it < 200; it++) {
        F lp = F(0.5) * (lp_lo + lp_hi);
        if (maxwell(lp, rl, rv) > 0) lp_lo = lp; else lp_hi = lp;
        rho_l = rl;
        rho_v = rv;
        if (lp_hi - lp_lo < F(1e-14)) break;
    }
    return rho_v < rho_l;
}

// Newton corrector at spec-phase composition z: unknowns (ln rs, ln ri0, ln ri1), steps limited to `cap` in the
// logarithms.  Returns the number of iterations used, or -1 on failure.
template <class F, class Model>
int corrector(const Model& model, F T, const F* z, F& rs, F* ri, F tol, F cap, int max_it) {
    F mx_prev = F(1);
    for (int it = 0; it < max_it; it++) {
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
        if (!solve3<F>(J, rhs, du)) return -1;
        F mx = 0;
        for (int k = 0; k < 3; k++) { if (!(du[k] == du[k])) return -1; if (absf<F>(du[k]) > mx) mx = absf<F>(du[k]); }
        F scale = mx > cap ? cap / mx : F(1);
        rs = rs * exp(scale * du[0]);
        ri[0] = ri[0] * exp(scale * du[1]);
        ri[1] = ri[1] * exp(scale * du[2]);
        // converged, or stalled at the rounding-noise floor of the model evaluation
        if (mx <= tol || (it >= 3 && mx < F(1e-8) && mx >= F(0.25) * mx_prev)) return it + 1;
        mx_prev = mx;
    }
    return -1;
}

}  // namespace cont_detail

// z1 = mole fraction of component 1 in the specified phase (liquid for bubble, vapour for dew).  On CONT_OK rho_spec /
// rho_inc hold the partial densities of the specified and the incipient phase.
template <class F, class Model>
ContInfo bubble_dew_continuation(const Model& model, F T, F z1, bool dew, F* rho_spec, F* rho_inc, F tol = F(1e-12)) {
    using namespace cont_detail;
    ContInfo best, found;
    F p_found = F(0);
    const F xi_target = log(z1 / (F(1) - z1));
    bool any_start = false, any_critical = false;
    for (int route = 0; route < 2; route++) {
        // ---- start next to the pure end of this route -----------------------------------------------------------
        const int major = route;  // component that is nearly pure at the start
        F xp[2] = {major == 0 ? F(1) - F(1e-6) : F(1e-6), major == 0 ? F(1e-6) : F(1) - F(1e-6)};
        F rv0, rl0;
        if (!line_vle<F>(model, T, xp, rv0, rl0)) continue;
        any_start = true;
        // K-factor of the minor component at infinite dilution from the two line states
        F rL[2] = {xp[0] * rl0, xp[1] * rl0}, rV[2] = {xp[0] * rv0, xp[1] * rv0};
        PhaseEval<F> eL = eval_phase<F>(model, T, rL), eV = eval_phase<F>(model, T, rV);
        const int minor = 1 - major;
        // y_i / x_i = (rho_L / rho_V) exp(g_i^L - g_i^V)
        F lnK = log(rl0 / rv0) + eL.g[minor] - eV.g[minor];
        // start composition: minor mole fraction <= 1e-3 in BOTH phases
        F ln_delta = log(F(1e-3));
        if (!dew && lnK > 0) ln_delta -= lnK;  // bubble: y_minor = K x_minor
        if (dew && lnK < 0) ln_delta += lnK;   // dew:    x_minor = y_minor / K
        F xi = (major == 0) ? -ln_delta : ln_delta;  // xi = ln(z1/z2): z_minor = delta
        if ((major == 0 && xi < xi_target) || (major == 1 && xi > xi_target)) xi = xi_target;  // target is even closer to the end
        auto comp = [](F xi_, F* z) { F e = exp(-absf<F>(xi_)); F big = F(1) / (F(1) + e), small = e / (F(1) + e); z[0] = xi_ >= 0 ? big : small; z[1] = xi_ >= 0 ? small : big; };
        F z[2];
        comp(xi, z);
        // initial state: specified phase = line state of its kind, incipient partial densities from fugacity equality with
        // the other line state's residual potentials
        F rs = dew ? rv0 : rl0, ri[2];
        {
            const PhaseEval<F>& es = dew ? eV : eL;
            const PhaseEval<F>& ei = dew ? eL : eV;
            for (int i = 0; i < 2; i++) ri[i] = z[i] * rs * exp(es.g[i] - ei.g[i]);
        }
        ContInfo info;
        info.route = route;
        int used = corrector<F>(model, T, z, rs, ri, tol, F(1), 60);
        if (used < 0) continue;
        info.newton += used;
        // ---- march to the target ---------------------------------------------------------------------------------
        F u_prev[3] = {log(rs), log(ri[0]), log(ri[1])}, xi_prev = xi;
        F u_prev2[3] = {u_prev[0], u_prev[1], u_prev[2]}, xi_prev2 = xi;
        bool have2 = false;
        F h = F(0.5);
        const F dir = xi_target >= xi ? F(1) : F(-1);
        bool reached = (xi == xi_target), critical = false;
        while (!reached && info.steps < 4000) {
            F step = h;
            if (absf<F>(xi_target - xi) <= step) step = absf<F>(xi_target - xi);
            const F xi_new = xi + dir * step;
            comp(xi_new, z);
            // secant predictor
            F u[3];
            for (int k = 0; k < 3; k++) {
                F slope = have2 ? (u_prev[k] - u_prev2[k]) / (xi_prev - xi_prev2) : F(0);
                u[k] = u_prev[k] + slope * (xi_new - xi_prev);
            }
            F rs_t = exp(u[0]), ri_t[2] = {exp(u[1]), exp(u[2])};
            used = corrector<F>(model, T, z, rs_t, ri_t, tol, F(0.7), 25);
            bool ok = used > 0;
            if (ok) {
                // stayed on the two-phase branch? (phases distinct, the liquid denser than the vapour)
                F dens_i = ri_t[0] + ri_t[1];
                F lo = dew ? rs_t : dens_i, hi = dew ? dens_i : rs_t;
                if (!(lo < hi)) ok = false;
                else if ((hi - lo) < F(2e-2) * hi) { critical = true; break; }  // approaching a critical point
                // no jump to another solution: the corrected state is close to the prediction
                for (int k = 0; k < 3 && ok; k++) {
                    F v = k == 0 ? log(rs_t) : log(ri_t[k - 1]);
                    if (absf<F>(v - u[k]) > F(2.5)) ok = false;
                }
            }
            info.steps++;
            if (!ok) {
                h *= F(0.5);
                if (h < F(1e-7)) break;
                continue;
            }
            info.newton += used;
            for (int k = 0; k < 3; k++) u_prev2[k] = u_prev[k];
            xi_prev2 = xi_prev;
            u_prev[0] = log(rs_t); u_prev[1] = log(ri_t[0]); u_prev[2] = log(ri_t[1]);
            xi_prev = xi_new;
            have2 = true;
            xi = xi_new;
            rs = rs_t; ri[0] = ri_t[0]; ri[1] = ri_t[1];
            if (used <= 4 && h < F(1.0)) h *= F(1.5);
            if (xi == xi_target) reached = true;
        }
        if (reached) {
            comp(xi_target, z);
            z[0] = z1; z[1] = F(1) - z1;
            // final polish exactly at the requested composition
            used = corrector<F>(model, T, z, rs, ri, tol, F(0.5), 30);
            if (used > 0) {
                F dens_i = ri[0] + ri[1];
                F lo = dew ? rs : dens_i, hi = dew ? dens_i : rs;
                if (lo < hi * (F(1) - F(1e-6))) {
                    // a solution at the target.  BOTH routes are followed: where they arrive at different solutions (a
                    // liquid-liquid split: two incipient liquids satisfy the dew equations, two vapours never do) the
                    // STABLE one is kept -- at fixed vapour composition the dew point is the lowest pressure at which a liquid
                    // can form, at fixed liquid composition the bubble point the highest at which a vapour can
                    F r_s[2] = {z[0] * rs, z[1] * rs};
                    const F p_here = eval_phase<F>(model, T, r_s).p();
                    if (found.code != CONT_OK || (dew ? p_here < p_found : p_here > p_found)) {
                        rho_spec[0] = r_s[0]; rho_spec[1] = r_s[1];
                        rho_inc[0] = ri[0]; rho_inc[1] = ri[1];
                        info.code = CONT_OK;
                        found = info;
                        p_found = p_here;
                    }
                    continue;
                }
            }
        }
        any_critical = any_critical || critical;
        if (info.steps > best.steps) { best.steps = info.steps; best.newton = info.newton; best.route = route; }
    }
    if (found.code == CONT_OK) return found;
    best.code = !any_start ? CONT_NO_START : (any_critical ? CONT_CRITICAL : CONT_STALLED);
    return best;
}

}  // namespace oracle
