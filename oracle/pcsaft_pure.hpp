// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
//
// CPU restatement of the reference's pure-component PC-SAFT path:
//   helmholtz_energy      <- feos_torch/pcsaft_pure.py:106-178  (statement order kept)
//   derivatives           <- feos_torch/pcsaft_pure.py:180-182
//   final Newton steps    <- feos_torch/pcsaft_pure.py:196-199, :212-215, :228-233
//   parameter conventions <- feos_torch/pcsaft_pure.py:90-104, src/pcsaft.rs:131-148
// The iteration that produces the converged densities lives in the third-party crate
// feos = "0.6" (Cargo.toml:18; call sites src/pcsaft.rs:91 and :116-122), which is NOT in
// /root/reference.  It is restated here from its published algorithm (Rehner et al.,
// feos `PhaseEquilibrium::pure`: Newton on both phase densities towards the equal-area
// pressure p* = -(f_V - f_L)/(v_V - v_L); `State::new_npt(.., Liquid)`: Newton in density
// from a liquid-like start) and anchored on the reference's stored answers README.md:26-29.
// Because every returned property is one explicit Newton step at the converged densities
// whose density-derivative vanishes at the root, the result is defined by the Helmholtz
// model alone, not by the iteration history.
#pragma once
#include <cstdint>
#include "constants.hpp"
#include "dual.hpp"

namespace oracle {

// Parameter row layout (README.md:12): m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb
template <class S>
struct PureParams {
    S m, sigma, epsilon_k, mu2, kappa_ab, epsilon_k_ab, na, nb;
};

// feos_torch/pcsaft_pure.py:90-104
template <class S>
PureParams<S> make_pure_params(const S* p) {
    PureParams<S> q;
    q.m = p[0];
    q.sigma = p[1];
    q.epsilon_k = p[2];
    q.mu2 = p[3] * p[3] / (q.m * (q.sigma * q.sigma * q.sigma) * q.epsilon_k) * 1e-19 * (1.0 / KB);
    q.kappa_ab = p[4];
    q.epsilon_k_ab = p[5];
    q.na = p[6];
    q.nb = p[7];
    return q;
}

// feos_torch/pcsaft_pure.py:106-178.  S is double, long double or any dual type;
// m_re is the plain value of m (needed for the clamp at :146).
template <class S>
S helmholtz_energy(const PureParams<S>& q, const S& temperature, const S& density) {
    // temperature dependent segment diameter (:108)
    S d = q.sigma * (1.0 - 0.12 * exp(-3.0 * q.epsilon_k / temperature));

    S eta = PI / 6.0 * q.m * density * (d * d * d);  // :110
    S eta2 = eta * eta;
    S eta3 = eta2 * eta;
    S eta_m1 = 1.0 / (1.0 - eta);
    S eta_m2 = eta_m1 * eta_m1;
    S etas[7] = {S(1.0), eta, eta2, eta3, eta2 * eta2, eta2 * eta3, eta3 * eta3};  // :115

    // hard sphere (:118)
    S hs = q.m * density * (4.0 * eta - 3.0 * eta2) * eta_m2;

    // hard chain (:121-122)
    S g = (1.0 - eta / 2.0) * eta_m1 * eta_m2;
    S hc = -density * (q.m - 1.0) * log(g);

    // dispersion (:125-142)
    S e = q.epsilon_k / temperature;
    S s3 = q.sigma * q.sigma * q.sigma;
    S I1(0.0), I2(0.0);
    S m1 = (q.m - 1.0) / q.m;
    S m2 = (q.m - 2.0) / q.m;
    for (int i = 0; i < 7; i++) {
        I1 = I1 + (m1 * (m2 * A2[i] + A1[i]) + A0[i]) * etas[i];
        I2 = I2 + (m1 * (m2 * B2[i] + B1[i]) + B0[i]) * etas[i];
    }
    S C1 = 1.0 / (1.0 + q.m * (8.0 * eta - 2.0 * eta2) * eta_m2 * eta_m2 +
                  (1.0 - q.m) * (20.0 * eta - 27.0 * eta2 + 12.0 * eta2 * eta - 2.0 * eta2 * eta2) /
                      ((1.0 - eta) * (1.0 - eta) * (2.0 - eta) * (2.0 - eta)));
    S I = 2.0 * I1 + C1 * I2 * q.m * e;
    S disp = (-PI * density * density * (q.m * q.m) * e * s3) * I;

    // dipoles (:145-160); m clamped to <= 2 (:146)
    S mu2 = q.mu2 * e * s3;
    S mc = (re(q.m) > 2.0) ? S(2.0) : q.m;
    S md1 = (mc - 1.0) / mc;
    S md2 = md1 * (mc - 2.0) / mc;
    S J1(0.0), J2(0.0);
    for (int i = 0; i < 5; i++) {
        S a = AD[i][0] + md1 * AD[i][1] + md2 * AD[i][2];
        S b = BD[i][0] + md1 * BD[i][1] + md2 * BD[i][2];
        J1 = J1 + (a + b * e) * etas[i];
    }
    for (int i = 0; i < 4; i++) J2 = J2 + (CD[i][0] + md1 * CD[i][1] + md2 * CD[i][2]) * etas[i];
    const double PI_SQ_43 = 4.0 / 3.0 * PI * PI;
    // mu is factored out of these expressions to deal with the case where mu=0 (:157)
    S phi2 = -density * density * J1 / s3 * PI;
    S phi3 = -density * density * density * J2 / s3 * PI_SQ_43;
    S dipole = phi2 * phi2 * mu2 * mu2 / (phi2 - phi3 * mu2);

    // association (:163-176)
    S delta_assoc = (exp(q.epsilon_k_ab / temperature) - 1.0) * (q.sigma * q.sigma * q.sigma) * q.kappa_ab;
    S k = eta * eta_m1;
    S delta = (1.0 + k * (1.5 + 0.5 * k)) * eta_m1 * delta_assoc;
    S rhoa = q.na * density;
    S rhob = q.nb * density;
    S xa, xb;
    site_fractions_two_types(rhoa, rhob, delta, xa, xb);  // :172-175 (literal in double, conjugate forms in long double)
    S assoc = rhoa * (log(xa) - 0.5 * xa + 0.5) + rhob * (log(xb) - 0.5 * xb + 0.5);

    return hs + hc + disp + dipole + assoc;
}

// feos_torch/pcsaft_pure.py:180-182: (a, p = rho - a + rho a', dp = 1 + rho a'')
template <class F>
void derivatives(const PureParams<F>& q, F T, F rho, F& a, F& p, F& dp) {
    typedef Dual3<F> D;
    PureParams<D> qd = {D(q.m, 0, 0), D(q.sigma, 0, 0), D(q.epsilon_k, 0, 0), D(q.mu2, 0, 0),
                        D(q.kappa_ab, 0, 0), D(q.epsilon_k_ab, 0, 0), D(q.na, 0, 0), D(q.nb, 0, 0)};
    D r = helmholtz_energy(qd, D(T, 0, 0), D::diff(rho));
    a = r.re;
    p = rho - r.re + rho * r.v1;
    dp = F(1.0) + rho * r.v2;
}

// ---------------------------------------------------------------------------------------
// Solver (restated feos algorithm; see file header).  F = double or long double.
// ---------------------------------------------------------------------------------------
struct SolveInfo {
    int iters;  // outer iterations used
    int path;   // 0 = zero-pressure-liquid initialisation, 1 = spinodal initialisation
};

template <class F>
F eta_to_rho(const PureParams<F>& q, F T, F eta) {
    F d = q.sigma * (F(1.0) - F(0.12) * exp(F(-3.0) * q.epsilon_k / T));
    return eta / (F(PI) / F(6.0) * q.m * d * d * d);
}

// Rightmost root of p(rho) = p_spec reached from a liquid-like start while staying on the
// mechanically stable branch (dp > 0).  Restates feos State::new_npt(.., Liquid) as called at
// src/pcsaft.rs:116-122.  Returns false when the liquid branch has no such root.
template <class F>
bool liquid_density_at_p(const PureParams<F>& q, F T, F p_spec, F& rho_out, int& iters, F tol) {
    F rho = eta_to_rho(q, T, F(0.5));
    F a, p, dp;
    for (int tries = 0; tries < 6; tries++) {  // make sure we start right of the root
        derivatives(q, T, rho, a, p, dp);
        if (p > p_spec && dp > 0) break;
        rho = rho * F(1.1);
    }
    F err_prev = F(1);
    for (int it = 0; it < 200; it++) {
        derivatives(q, T, rho, a, p, dp);
        if (!(dp > 0) || !(p == p)) return false;
        F step = (p - p_spec) / dp;
        F rho_new = rho - step;
        if (!(rho_new > 0)) return false;
        iters = it + 1;
        F err = (step < 0 ? -step : step) / rho;
        // below tol, or stalled at the rounding-noise floor of the model (see vle_pure)
        bool done = err <= tol || (it >= 3 && err < F(1e-7) && err >= F(0.25) * err_prev);
        err_prev = err;
        rho = rho_new;
        if (done) {
            derivatives(q, T, rho, a, p, dp);
            if (!(dp > 0)) return false;
            rho_out = rho;
            return true;
        }
    }
    return false;
}

// Safeguarded Newton for p(rho) = p_spec inside a bracket [lo, hi] on which p is increasing.
template <class F>
F solve_on_branch(const PureParams<F>& q, F T, F p_spec, F lo, F hi, F rho) {
    F a, p, dp;
    for (int it = 0; it < 200; it++) {
        derivatives(q, T, rho, a, p, dp);
        if (p > p_spec) hi = rho; else lo = rho;
        F rho_new = (dp > 0) ? rho - (p - p_spec) / dp : F(-1);
        if (!(rho_new > lo && rho_new < hi)) rho_new = F(0.5) * (lo + hi);
        F diff = rho_new - rho;
        rho = rho_new;
        if ((diff < 0 ? -diff : diff) <= F(1e-13) * rho) break;
    }
    return rho;
}

// Root of dp/drho = 0 in [lo, hi] by bisection (dp changes sign in the bracket).
template <class F>
F spinodal_bisect(const PureParams<F>& q, F T, F lo, F hi, bool dp_positive_at_lo) {
    F a, p, dp;
    for (int it = 0; it < 60; it++) {
        F mid = F(0.5) * (lo + hi);
        derivatives(q, T, mid, a, p, dp);
        if ((dp > 0) == dp_positive_at_lo) lo = mid; else hi = mid;
    }
    return F(0.5) * (lo + hi);
}

// Pure-component VLE at fixed T.  Returns false = failed (status True in the reference's
// convention, src/pcsaft.rs:93).  On success rho_v < rho_l are the converged densities.
template <class F>
bool vle_pure(const PureParams<F>& q, F T, F& rho_v, F& rho_l, SolveInfo& info, F tol) {
    F a, p, dp;
    info.iters = 0;
    info.path = 0;
    // --- initialisation 1: liquid at zero pressure, vapour from ideal-gas fugacity equality
    int it0 = 0;
    F rl = 0, rv = 0;
    bool ok = liquid_density_at_p(q, T, F(0), rl, it0, F(1e-8));
    if (ok) {
        Dual3<F> r;
        {
            typedef Dual3<F> D;
            PureParams<D> qd = {D(q.m, 0, 0), D(q.sigma, 0, 0), D(q.epsilon_k, 0, 0), D(q.mu2, 0, 0),
                                D(q.kappa_ab, 0, 0), D(q.epsilon_k_ab, 0, 0), D(q.na, 0, 0), D(q.nb, 0, 0)};
            r = helmholtz_energy(qd, D(T, 0, 0), D::diff(rl));
        }
        rv = rl * exp(r.v1);  // ln rho_V = ln rho_L + a'(rho_L), ideal vapour
        // pull the vapour guess back onto the stable vapour branch if necessary
        for (int k = 0; k < 60; k++) {
            derivatives(q, T, rv, a, p, dp);
            if (dp > 0 && p > 0 && rv < F(0.5) * rl) break;
            rv = rv * F(0.5);
        }
    } else {
        // --- initialisation 2: spinodals (near-critical temperatures)
        info.path = 1;
        F rho = eta_to_rho(q, T, F(0.5));
        F rho_stable = rho;
        bool found = false;
        for (int k = 0; k < 400; k++) {  // walk down until dp < 0
            derivatives(q, T, rho, a, p, dp);
            if (!(dp > 0)) { found = true; break; }
            rho_stable = rho;
            rho = rho * F(0.97);
            if (rho < F(1e-3) * rho_stable && k > 300) break;
        }
        if (!found) return false;  // supercritical: no spinodal
        F rho_sl = spinodal_bisect(q, T, rho, rho_stable, false);
        // vapour spinodal: walk further down until dp > 0 again
        F rho_unstable = rho;
        found = false;
        for (int k = 0; k < 2000; k++) {
            rho = rho * F(0.97);
            derivatives(q, T, rho, a, p, dp);
            if (dp > 0) { found = true; break; }
            rho_unstable = rho;
        }
        if (!found) return false;
        F rho_sv = spinodal_bisect(q, T, rho, rho_unstable, true);
        F p_sl, p_sv;
        derivatives(q, T, rho_sl, a, p_sl, dp);
        derivatives(q, T, rho_sv, a, p_sv, dp);
        if (!(p_sv > 0)) return false;
        F p0 = F(0.5) * ((p_sl > 0 ? p_sl : F(0)) + p_sv);
        F hi = eta_to_rho(q, T, F(0.6));
        rl = solve_on_branch(q, T, p0, rho_sl, hi, F(0.5) * (rho_sl + hi));
        rv = solve_on_branch(q, T, p0, F(0), rho_sv, F(0.5) * rho_sv);
    }
    // --- Newton on both densities towards the equal-area pressure
    F err_prev = F(1);
    for (int it = 0; it < 100; it++) {
        F a_l, p_l, dp_l, a_v, p_v, dp_v;
        derivatives(q, T, rl, a_l, p_l, dp_l);
        derivatives(q, T, rv, a_v, p_v, dp_v);
        F pstar = -(a_v / rv - a_l / rl + log(rv / rl)) / (F(1) / rv - F(1) / rl);
        F dl = -(p_l - pstar) / dp_l;
        F dv = -(p_v - pstar) / dp_v;
        // keep both phases on their stable branches: backtrack while the trial point is unstable
        F rl_new = rl + dl, rv_new = rv + dv;
        for (int k = 0; k < 40; k++) {
            F aa, pp, dd;
            derivatives(q, T, rl_new, aa, pp, dd);
            if (rl_new > 0 && dd > 0) break;
            dl = dl * F(0.5);
            rl_new = rl + dl;
        }
        for (int k = 0; k < 40; k++) {
            F aa, pp, dd;
            if (rv_new > 0) {
                derivatives(q, T, rv_new, aa, pp, dd);
                if (dd > 0) break;
            }
            dv = dv * F(0.5);
            rv_new = rv + dv;
        }
        info.iters = it + 1;
        F el = (dl < 0 ? -dl : dl) / rl, ev = (dv < 0 ? -dv : dv) / rv;
        F err = el > ev ? el : ev;
        rl = rl_new;
        rv = rv_new;
        if (!(rl == rl) || !(rv == rv)) return false;
        // Converged when the relative Newton steps are below tol, or when they have stopped
        // shrinking at the rounding-noise floor of the model itself (the association term
        // xb = 2/(sqrt + aux), feos_torch/pcsaft_pure.py:175, cancels catastrophically for
        // strongly associating fluids at low T, so the floor can sit far above machine epsilon).
        bool stagnated = it >= 3 && err < F(1e-7) && err >= F(0.25) * err_prev;
        err_prev = err;
        if (err <= tol || stagnated) {
            if (!(rv < rl * (F(1) - F(1e-6)))) return false;  // trivial solution
            rho_v = rv;
            rho_l = rl;
            return true;
        }
    }
    return false;
}

// feos_torch/pcsaft_pure.py:212-215 evaluated at given densities -> Pa
template <class F>
F vapor_pressure_formula(const PureParams<F>& q, F T, F rho_v, F rho_l) {
    F a_l = helmholtz_energy(q, T, rho_l) / rho_l;
    F a_v = helmholtz_energy(q, T, rho_v) / rho_v;
    F p = -(a_v - a_l + log(rho_v / rho_l)) / (F(1) / rho_v - F(1) / rho_l);
    return p * T * F(P_UNIT);
}

// feos_torch/pcsaft_pure.py:196-199 -> kmol/m3
template <class F>
F liquid_density_formula(const PureParams<F>& q, F T, F p_pa, F rho) {
    F pressure = p_pa / T * F(1.0 / P_UNIT);
    F a, p, dp;
    derivatives(q, T, rho, a, p, dp);
    F density = rho - (p - pressure) / dp;
    return density / F(RHO_UNIT);
}

// feos_torch/pcsaft_pure.py:226-233 -> kmol/m3
template <class F>
F equilibrium_liquid_density_formula(const PureParams<F>& q, F T, F rho_v, F rho_l) {
    F a_l, p_l, dp_l;
    derivatives(q, T, rho_l, a_l, p_l, dp_l);
    a_l = a_l / rho_l;
    F a_v = helmholtz_energy(q, T, rho_v) / rho_v;
    F p = -(a_v - a_l + log(rho_v / rho_l)) / (F(1) / rho_v - F(1) / rho_l);
    F liquid_density = rho_l - (p_l - p) / dp_l;
    return liquid_density / F(RHO_UNIT);
}

}  // namespace oracle
