// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
//
// CPU restatement of the reference's binary-mixture PC-SAFT path:
//   helmholtz_energy_density <- feos_torch/pcsaft_mix.py:31-154
//   phi_dipole               <- :156-208, pair_integral :482-490, triplet_integral :493-497
//   phi_self_assoc           <- :210-239
//   phi_cross_assoc          <- :241-321   (2x2 Newton carried by Dual2, dual_torch.py:165-208)
//   phi_induced_assoc        <- :324-393
//   association_strength     <- :500-522
//   derivatives              <- :395-420   (hyper-dual pass: a, p, mu_i, v_i)
//   bubble / dew tails       <- :435-444, :459-468
//   parameter conventions    <- :13-29, src/pcsaft.rs:163-168 (kij[0] = k_ij, kij[1] = eps_AiBj or 0)
// n = 2 components throughout (the reference's association code is binary-only, :250, :336).
// The bubble/dew ITERATION lives in the absent feos crate (src/pcsaft.rs:170, :203); it is
// restated in mix_solver.hpp.
#pragma once
#include "constants.hpp"
#include "dual.hpp"

namespace oracle {

template <class S>
struct MixParams {
    S m[2], sigma[2], epsilon_k[2], mu2[2], kappa_ab[2], epsilon_k_ab[2], na[2], nb[2];
    S kij;        // kij[:,0]
    S eps_aibj;   // kij[:,1]; 0 = use the arithmetic mean (:509-516)
    // false: association sub-iterations exactly as the reference runs them (start 0.2, <= 50 steps,
    //        step-back x <- 0.2 x_old, :270-311, :361-385) — used to pin against the reference Python.
    // true : same start and Newton steps, but safeguarded (bracketing / successive-substitution
    //        fallback) so that it also converges where the reference's iteration runs away
    //        (e.g. A-site excess with X_A > 0.2, where the reference collapses to X_A -> 0 -> NaN).
    //        Identical results wherever the reference converges.  Used by the solvers.
    bool robust = false;
};

// feos_torch/pcsaft_mix.py:13-29.  par = [2][8] rows (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb)
template <class S>
MixParams<S> make_mix_params(const S* par, const S& kij0, const S& kij1) {
    MixParams<S> q;
    for (int i = 0; i < 2; i++) {
        const S* p = par + 8 * i;
        q.m[i] = p[0];
        q.sigma[i] = p[1];
        q.epsilon_k[i] = p[2];
        q.mu2[i] = p[3] * p[3] / (q.m[i] * (q.sigma[i] * q.sigma[i] * q.sigma[i]) * q.epsilon_k[i]) * 1e-19 * (1.0 / KB);
        q.kappa_ab[i] = p[4];
        q.epsilon_k_ab[i] = p[5];
        q.na[i] = p[6];
        q.nb[i] = p[7];
    }
    q.kij = kij0;
    q.eps_aibj = kij1;
    return q;
}

template <class S>
S clamp2(const S& m) { return (re(m) > 2.0) ? S(2.0) : m; }  // torch .clamp(max=2)

// :482-490
template <class S>
S pair_integral(const S& mij1, const S& mij2, const S* etas, const S& eps_ij_t) {
    S r(0.0);
    for (int i = 0; i < 5; i++)
        r = r + etas[i] * ((eps_ij_t * (BD[i][0] + mij1 * BD[i][1] + mij2 * BD[i][2])) + (AD[i][0] + mij1 * AD[i][1] + mij2 * AD[i][2]));
    return r;
}
// :493-497
template <class S>
S triplet_integral(const S& mijk1, const S& mijk2, const S* etas) {
    S r(0.0);
    for (int i = 0; i < 4; i++) r = r + etas[i] * (CD[i][0] + mijk1 * CD[i][1] + mijk2 * CD[i][2]);
    return r;
}

// :500-522.  use_aibj: the cross-association call passes kij[:,1] (:141), the others None.
template <class S>
S association_strength(int i, int j, const S& T, const S* sigma, const S* kappa_ab, const S* epsilon_k_ab,
                       bool use_aibj, const S& eps_aibj, const S* d, const S& zeta2, const S& zeta3_m1) {
    S di = d[i], dj = d[j];
    S k = di * dj / (di + dj) * zeta2 * zeta3_m1;
    S ss = sigma[i] * sigma[j];
    S sigma3_kappa = ss * sqrt(ss) * sqrt(kappa_ab[i] * kappa_ab[j]);  // (..)**1.5 * sqrt(..)
    S e;
    if (use_aibj && i != j && re(eps_aibj) != 0.0) e = eps_aibj;
    else e = 0.5 * (epsilon_k_ab[i] + epsilon_k_ab[j]);
    return zeta3_m1 * (k * (2.0 * k + 3.0) + 1.0) * sigma3_kappa * (exp(e / T) - 1.0);
}

template <class S>
S site_f(const S& x) { return log(x) - 0.5 * x + 0.5; }

// :156-208
template <class S>
S phi_dipole(const MixParams<S>& q, const S& T, const S* rho, const S* etas) {
    S mu2_term[2];
    for (int i = 0; i < 2; i++) mu2_term[i] = (q.sigma[i] * q.sigma[i] * q.sigma[i]) * q.epsilon_k[i] * q.mu2[i] / T;
    S phi2(0.0), phi3(0.0);
    for (int i = 0; i < 2; i++) {
        for (int j = i; j < 2; j++) {
            S s_ij = 0.5 * (q.sigma[i] + q.sigma[j]);
            S sigma_ij_3 = s_ij * s_ij * s_ij;
            S mij = sqrt(clamp2(q.m[i]) * clamp2(q.m[j]));
            S mij1 = (mij - 1.0) / mij;
            S mij2 = mij1 * (mij - 2.0) / mij;
            S eps_ij_t = sqrt(q.epsilon_k[i] * q.epsilon_k[j]) / T;
            double c = (i == j) ? 1.0 : 2.0;
            phi2 = phi2 - rho[i] * rho[j] * mu2_term[i] * mu2_term[j] * pair_integral(mij1, mij2, etas, eps_ij_t) / sigma_ij_3 * c;
            for (int k = j; k < 2; k++) {
                S sigma_ij = 0.5 * (q.sigma[i] + q.sigma[j]);
                S sigma_ik = 0.5 * (q.sigma[i] + q.sigma[k]);
                S sigma_jk = 0.5 * (q.sigma[j] + q.sigma[k]);
                S mijk = cbrt(clamp2(q.m[i]) * clamp2(q.m[j]) * clamp2(q.m[k]));
                S mijk1 = (mijk - 1.0) / mijk;
                S mijk2 = mijk1 * (mijk - 2.0) / mijk;
                int distinct = 1 + (j != i) + (k != j);  // len({i,j,k}) for i <= j <= k
                double c3 = (distinct == 1) ? 1.0 : (distinct == 2 ? 3.0 : 6.0);
                phi3 = phi3 - rho[i] * rho[j] * rho[k] * mu2_term[i] * mu2_term[j] * mu2_term[k] *
                                  triplet_integral(mijk1, mijk2, etas) / (sigma_ij * sigma_ik * sigma_jk) * c3;
            }
        }
    }
    phi2 = phi2 * PI;
    phi3 = phi3 * (4.0 / 3.0 * PI * PI);
    // phi2 and phi3 vanish together where no polar component is present (a pure-component limit of a
    // mixture with one polar partner): the quotient is 0/0 in the reference's Python; its limit is
    // phi2 + O(rho_polar^3), which is what feos' dipole term (the solver's own model) returns there.
    // ... and the same limit is taken for a TRACE polar component (|phi2| < 1e-90, i.e. partial densities below ~1e-45 of the
    // liquid's): the second derivatives of the quotient carry 1/phi2^3, which overflows fp64 there (NaN Newton steps on dew
    // rows whose incipient liquid holds 1e-50 of the polar component); the neglected phi3 term is O(rho_polar^3) < 1e-135
    if (fabsl((long double)re(phi2)) < 1e-90L) return phi2;
    return phi2 * phi2 / (phi2 - phi3);
}

// :210-239
template <class S>
S phi_self_assoc(const MixParams<S>& q, const S& T, const S* rho, const S* d, const S& zeta2, const S& zeta3_m1) {
    S kappa_ab = q.kappa_ab[0] + q.kappa_ab[1];
    S epsilon_k_ab = q.epsilon_k_ab[0] + q.epsilon_k_ab[1];
    S na_sum = q.na[0] + q.na[1];
    S sigma = (q.na[0] * q.sigma[0] + q.na[1] * q.sigma[1]) / na_sum;
    S dd = (q.na[0] * d[0] + q.na[1] * d[1]) / na_sum;
    S s1[1] = {sigma}, k1[1] = {kappa_ab}, e1[1] = {epsilon_k_ab}, d1[1] = {dd};
    S delta = association_strength(0, 0, T, s1, k1, e1, false, S(0.0), d1, zeta2, zeta3_m1);
    S rhoa = q.na[0] * rho[0] + q.na[1] * rho[1];
    S rhob = q.nb[0] * rho[0] + q.nb[1] * rho[1];
    S xa, xb;
    site_fractions_two_types(rhoa, rhob, delta, xa, xb);  // :235-238 (literal in double, conjugate forms in long double)
    return rhoa * site_f(xa) + rhob * site_f(xb);
}

// ---- association sub-iterations -------------------------------------------------------------
// One Newton update of the cross-association unknowns (X_A0, X_A1), residuals and Jacobian
// carried by Dual2 exactly as at feos_torch/pcsaft_mix.py:272-303.
template <class S>
void cross_newton_step(const S& xa0, const S& xa1, const S* rhoa, const S* rhob, const S& d00, const S& d01,
                       const S& d10, const S& d11, S& g0, S& g1, S& dx0, S& dx1) {
    Dual2<S> X0(xa0, S(1.0), S(0.0)), X1(xa1, S(0.0), S(1.0));
    Dual2<S> xb0_i = S(1.0) + X0 * (rhoa[0] * d00) + X1 * (rhoa[1] * d01);
    Dual2<S> xb1_i = S(1.0) + X0 * (rhoa[0] * d10) + X1 * (rhoa[1] * d11);
    Dual2<S> f0 = X0 - S(1.0) + X0 / xb0_i * (rhob[0] * d00) + X0 / xb1_i * (rhob[1] * d01);
    Dual2<S> f1 = (X1 - S(1.0)) + X1 / xb0_i * (rhob[0] * d10) + X1 / xb1_i * (rhob[1] * d11);
    g0 = f0.re;
    g1 = f1.re;
    S j00 = f0.eps1, j01 = f0.eps2, j10 = f1.eps1, j11 = f1.eps2;
    S det = j00 * j11 - j01 * j10;
    dx0 = (j11 * g0 - j01 * g1) / det;
    dx1 = (-1.0 * j10 * g0 + j00 * g1) / det;
}

// :241-321.  Literal mode: the Newton iteration runs on S-valued unknowns exactly as the
// reference runs it on DualTensors; the stop test looks at real parts only (:310, per row here
// instead of the reference's batch-global norm) and `extra` further iterations
// are taken after it first holds so the dual parts are converged as well.
// Robust mode: the real parts are first converged in plain arithmetic with a successive-
// substitution fallback whenever Newton leaves (0, 1.5], then three Newton updates in S
// arithmetic from that point deliver the dual parts (implicit differentiation).
template <class S>
S phi_cross_assoc(const MixParams<S>& q, const S& T, const S* rho, const S* d, const S& zeta2, const S& zeta3_m1,
                  int extra) {
    S rhoa[2] = {rho[0] * q.na[0], rho[1] * q.na[1]};
    S rhob[2] = {rho[0] * q.nb[0], rho[1] * q.nb[1]};
    auto delta = [&](int i, int j) {
        return association_strength(i, j, T, q.sigma, q.kappa_ab, q.epsilon_k_ab, true, q.eps_aibj, d, zeta2, zeta3_m1);
    };
    S d00 = delta(0, 0), d01 = delta(0, 1), d10 = delta(1, 0), d11 = delta(1, 1);
    S xa0 = 0.2 * d00 / d00, xa1 = 0.2 * d00 / d00;  // :270
    S g0, g1, dx0, dx1;
    if (!q.robust) {
        int after = -1;
        for (int it = 0; it < 50; it++) {
            cross_newton_step(xa0, xa1, rhoa, rhob, d00, d01, d10, d11, g0, g1, dx0, dx1);
            S xa0_old = xa0, xa1_old = xa1;
            xa0 = xa0_old - dx0;
            xa1 = xa1_old - dx1;
            if (re(xa0) < 0.0) xa0 = 0.2 * xa0_old;  // :304-308
            if (re(xa1) < 0.0) xa1 = 0.2 * xa1_old;
            if (after < 0 && std::fabs((double)re(g0)) < 1e-10 && std::fabs((double)re(g1)) < 1e-10) after = 0;  // :310
            else if (after >= 0) after++;
            if (after >= extra) break;
        }
    } else {
        typedef decltype(re(d00)) R;
        R ra[2] = {re(rhoa[0]), re(rhoa[1])}, rb[2] = {re(rhob[0]), re(rhob[1])};
        R r00 = re(d00), r01 = re(d01), r10 = re(d10), r11 = re(d11);
        R x0 = R(0.2), x1 = R(0.2);
        for (int it = 0; it < 500; it++) {
            R h0, h1, e0, e1;
            cross_newton_step<R>(x0, x1, ra, rb, r00, r01, r10, r11, h0, h1, e0, e1);
            R n0 = x0 - e0, n1 = x1 - e1;
            if (!(n0 > 0 && n0 <= R(1.5) && n1 > 0 && n1 <= R(1.5))) {
                if (it < 60 && e0 == e0 && e1 == e1) {
                    // the Newton step leaves (0, 1.5]: take it in ln X instead (X <- X exp(-dX/X), at most a factor
                    // e^3 per component, capped at 1).  Strong association puts the root at X ~ 1e-5, which the
                    // successive substitution below approaches only sub-linearly.
                    R q0 = -e0 / x0, q1 = -e1 / x1;
                    q0 = q0 > R(3) ? R(3) : (q0 < R(-3) ? R(-3) : q0);
                    q1 = q1 > R(3) ? R(3) : (q1 < R(-3) ? R(-3) : q1);
                    n0 = x0 * exp(q0);
                    n1 = x1 * exp(q1);
                    if (n0 > R(1)) n0 = R(1);
                    if (n1 > R(1)) n1 = R(1);
                } else {
                // successive substitution X_Ai = 1/(1 + sum_j X_Bj rhob_j Delta_ij): lands in (0, 1]
                R xb0 = R(1) / (R(1) + x0 * ra[0] * r00 + x1 * ra[1] * r01);
                R xb1 = R(1) / (R(1) + x0 * ra[0] * r10 + x1 * ra[1] * r11);
                n0 = R(1) / (R(1) + xb0 * rb[0] * r00 + xb1 * rb[1] * r01);
                n1 = R(1) / (R(1) + xb0 * rb[0] * r10 + xb1 * rb[1] * r11);
                }
            }
            R c0 = (n0 - x0) / x0, c1 = (n1 - x1) / x1;
            x0 = n0;
            x1 = n1;
            if ((c0 < 0 ? -c0 : c0) < R(1e-15) && (c1 < 0 ? -c1 : c1) < R(1e-15)) break;
        }
        xa0 = S(0.0) + x0 * 1.0;
        xa1 = S(0.0) + x1 * 1.0;
        for (int k = 0; k < 3; k++) {
            cross_newton_step(xa0, xa1, rhoa, rhob, d00, d01, d10, d11, g0, g1, dx0, dx1);
            xa0 = xa0 - dx0;
            xa1 = xa1 - dx1;
        }
    }
    S xb0 = 1.0 / (1.0 + xa0 * rhoa[0] * d00 + xa1 * rhoa[1] * d01);
    S xb1 = 1.0 / (1.0 + xa0 * rhoa[0] * d10 + xa1 * rhoa[1] * d11);
    return rhoa[0] * site_f(xa0) + rhoa[1] * site_f(xa1) + rhob[0] * site_f(xb0) + rhob[1] * site_f(xb1);
}

// residual and Newton step of the induced-association unknown X_A (:364-377)
template <class S>
void induced_newton_step(const S& xa, const S& na0, const S& na1, const S& nb0, const S& nb1, const S& d00,
                         const S& d01, const S& d10, const S& d11, S& fval, S& dx) {
    Dual2<S> X(xa, S(1.0), S(0.0));
    Dual2<S> xb0_i = S(1.0) + X * (na0 * d00 + na1 * d01);
    Dual2<S> xb1_i = S(1.0) + X * (na0 * d10 + na1 * d11);
    Dual2<S> f0 = X * (xb0_i * xb1_i + xb1_i * (nb0 * d00) + xb0_i * (nb1 * d01)) - xb0_i * xb1_i;
    Dual2<S> f1 = X * (xb0_i * xb1_i + xb1_i * (nb0 * d10) + xb0_i * (nb1 * d11)) - xb0_i * xb1_i;
    Dual2<S> f = f0 * na0 + f1 * na1;
    fval = f.re;
    dx = f.re / f.eps1;
}

// :324-393 ("hard-coded for nA = 0" on the induced component, :323).  Literal / robust as above;
// robust real-part phase: f(0) = -(na0 + na1) < 0 < f(1), so a bracket is kept and bisected
// whenever Newton leaves it.
template <class S>
S phi_induced_assoc(const MixParams<S>& q, const S& T, const S* rho, const S* d, const S& zeta2, const S& zeta3_m1,
                    int extra) {
    const S &na0 = q.na[0], &na1 = q.na[1], &nb0 = q.nb[0], &nb1 = q.nb[1];
    auto delta_rho = [&](int i, int j) {
        return association_strength(i, j, T, q.sigma, q.kappa_ab, q.epsilon_k_ab, false, S(0.0), d, zeta2, zeta3_m1) * rho[j];
    };
    S d00 = delta_rho(0, 0), d01 = delta_rho(0, 1), d10 = delta_rho(1, 0), d11 = delta_rho(1, 1);
    S xa = 0.2 * d00 / d00;  // :361
    S fval, dx;
    if (!q.robust) {
        int after = -1;
        for (int it = 0; it < 50; it++) {
            induced_newton_step(xa, na0, na1, nb0, nb1, d00, d01, d10, d11, fval, dx);
            S xa_old = xa;
            xa = xa_old - dx;
            if (re(xa) < 0.0) xa = 0.2 * xa_old;  // :380-382
            if (after < 0 && std::fabs((double)re(fval)) < 1e-10) after = 0;  // :384
            else if (after >= 0) after++;
            if (after >= extra) break;
        }
    } else {
        typedef decltype(re(d00)) R;
        R a0 = re(na0), a1 = re(na1), b0 = re(nb0), b1 = re(nb1);
        R r00 = re(d00), r01 = re(d01), r10 = re(d10), r11 = re(d11);
        R x = R(0.2), lo = R(0), hi = R(2);
        for (int it = 0; it < 500; it++) {
            R fv, e;
            induced_newton_step<R>(x, a0, a1, b0, b1, r00, r01, r10, r11, fv, e);
            if (fv == 0) break;
            if (fv < 0) lo = x; else hi = x;
            R n = x - e;
            if (!(n >= lo && n <= hi && n > 0)) n = R(0.5) * (lo + hi);
            R c = (n - x) / x;
            x = n;
            if ((c < 0 ? -c : c) < R(1e-15)) break;
        }
        xa = S(0.0) + x * 1.0;
        for (int k = 0; k < 3; k++) {
            induced_newton_step(xa, na0, na1, nb0, nb1, d00, d01, d10, d11, fval, dx);
            xa = xa - dx;
        }
    }
    S xb0 = 1.0 / (1.0 + xa * (na0 * d00 + na1 * d01));
    S xb1 = 1.0 / (1.0 + xa * (na0 * d10 + na1 * d11));
    return rho[0] * (site_f(xa) * na0 + site_f(xb0) * nb0) + rho[1] * (site_f(xa) * na1 + site_f(xb1) * nb1);
}

// :31-154.  assoc_extra = Newton iterations after the real-part stop test first holds.
template <class S>
S helmholtz_energy_density_mix(const MixParams<S>& q, const S& T, const S* rho, int assoc_extra = 2) {
    S d[2];
    for (int i = 0; i < 2; i++) d[i] = q.sigma[i] * (1.0 - 0.12 * exp(-3.0 * q.epsilon_k[i] / T));  // :33

    S zeta0 = PI / 6.0 * (q.m[0] * rho[0] + q.m[1] * rho[1]);
    S zeta1 = PI / 6.0 * (q.m[0] * rho[0] * d[0] + q.m[1] * rho[1] * d[1]);
    S zeta2 = PI / 6.0 * (q.m[0] * rho[0] * d[0] * d[0] + q.m[1] * rho[1] * d[1] * d[1]);
    S zeta3 = PI / 6.0 * (q.m[0] * rho[0] * d[0] * d[0] * d[0] + q.m[1] * rho[1] * d[1] * d[1] * d[1]);

    S zeta23 = zeta2 / zeta3;
    S zeta3_2 = zeta3 * zeta3;
    S zeta3_3 = zeta3_2 * zeta3;
    S zeta3_m1 = 1.0 / (1.0 - zeta3);
    S zeta3_m2 = zeta3_m1 * zeta3_m1;
    S etas[7] = {S(1.0), zeta3, zeta3_2, zeta3_3, zeta3_2 * zeta3_2, zeta3_2 * zeta3_3, zeta3_3 * zeta3_3};

    // hard sphere (:56-60)
    S hs = (6.0 / PI) * (zeta1 * zeta2 * zeta3_m1 * 3.0 + zeta2 * zeta2 * zeta3_m2 * zeta23 +
                         (zeta2 * zeta23 * zeta23 - zeta0) * log(1.0 - zeta3));

    // hard chain (:63-65)
    S c = zeta2 * zeta3_m2;
    S hc(0.0);
    for (int i = 0; i < 2; i++) {
        S g = zeta3_m1 + d[i] * c * 1.5 - d[i] * d[i] * c * c * (zeta3 - 1.0) * 0.5;
        hc = hc + (-1.0 * rho[i]) * (q.m[i] - 1.0) * log(g);
    }

    // dispersion (:69-106)
    S rho_sum = rho[0] + rho[1];
    S m = (rho[0] / rho_sum) * q.m[0] + (rho[1] / rho_sum) * q.m[1];
    S rho1mix(0.0), rho2mix(0.0);
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 2; j++) {
            S eps_ij = sqrt(q.epsilon_k[i] * q.epsilon_k[j]) / T;
            if (i != j) eps_ij = eps_ij * (1.0 - q.kij);
            S s = 0.5 * (q.sigma[i] + q.sigma[j]);
            S sigma_ij = s * s * s;
            S m_ij = q.m[i] * q.m[j];
            S rhoij = rho[i] * rho[j] * (m_ij * eps_ij * sigma_ij);
            rho1mix = rho1mix + rhoij;
            rho2mix = rho2mix + rhoij * eps_ij;
        }
    }
    S I1(0.0), I2(0.0);
    S m1 = (m - 1.0) / m;
    S m2 = m1 * (m - 2.0) / m;
    for (int i = 0; i < 7; i++) {
        I1 = I1 + (m2 * A2[i] + m1 * A1[i] + A0[i]) * etas[i];
        I2 = I2 + (m2 * B2[i] + m1 * B1[i] + B0[i]) * etas[i];
    }
    S C1 = 1.0 / (1.0 + m * (8.0 * zeta3 - 2.0 * zeta3_2) * zeta3_m2 * zeta3_m2 +
                  (1.0 - m) * (20.0 * zeta3 - 27.0 * zeta3_2 + 12.0 * zeta3_2 * zeta3 - 2.0 * zeta3_2 * zeta3_2) /
                      ((1.0 - zeta3) * (1.0 - zeta3) * (2.0 - zeta3) * (2.0 - zeta3)));
    S disp = (-1.0 * rho1mix * 2.0 * I1 - rho2mix * C1 * I2 * m) * PI;

    S phi = hs + hc + disp;

    // dipoles (:111-115)
    if (re(q.mu2[0]) > 0.0 || re(q.mu2[1]) > 0.0) phi = phi + phi_dipole(q, T, rho, etas);

    // association (:118-152)
    int associating = (re(q.na[0] + q.nb[0]) != 0.0) + (re(q.na[1] + q.nb[1]) != 0.0);
    int self_assoc = (re(q.na[0] * q.nb[0]) != 0.0) + (re(q.na[1] * q.nb[1]) != 0.0);
    if (associating == 1 && self_assoc == 1) phi = phi + phi_self_assoc(q, T, rho, d, zeta2, zeta3_m1);
    if (associating == 2 && self_assoc == 2) phi = phi + phi_cross_assoc(q, T, rho, d, zeta2, zeta3_m1, assoc_extra);
    if (associating == 2 && self_assoc == 1) phi = phi + phi_induced_assoc(q, T, rho, d, zeta2, zeta3_m1, assoc_extra);
    return phi;
}

// :395-420.  F = double or long double.  Outputs: a, p, mu[2], v[2] (all reduced).
template <class F>
void derivatives_mix(const MixParams<F>& q, F T, const F* rho, F& a, F& p, F* mu, F* v) {
    typedef HyperDual<F, 3> H;
    auto lift = [](F x) { H h; h.re = x; return h; };
    MixParams<H> qh;
    for (int i = 0; i < 2; i++) {
        qh.m[i] = lift(q.m[i]); qh.sigma[i] = lift(q.sigma[i]); qh.epsilon_k[i] = lift(q.epsilon_k[i]);
        qh.mu2[i] = lift(q.mu2[i]); qh.kappa_ab[i] = lift(q.kappa_ab[i]); qh.epsilon_k_ab[i] = lift(q.epsilon_k_ab[i]);
        qh.na[i] = lift(q.na[i]); qh.nb[i] = lift(q.nb[i]);
    }
    qh.kij = lift(q.kij);
    qh.eps_aibj = lift(q.eps_aibj);
    qh.robust = q.robust;
    H volume = lift(F(1));          // :397-403
    volume.eps1[2] = F(1);
    volume.eps2 = F(1);
    H moles[2], dens[2];            // :404-411
    for (int i = 0; i < 2; i++) {
        moles[i] = lift(rho[i]);
        moles[i].eps1[i] = F(1);
        dens[i] = moles[i] / volume;
    }
    H A = helmholtz_energy_density_mix(qh, lift(T), dens) * volume;  // :413
    F rs = rho[0] + rho[1];
    p = rs - A.eps2;                                                  // :414
    for (int i = 0; i < 2; i++) {
        mu[i] = A.eps1[i];                                            // :415
        v[i] = -(F(1) - A.eps1eps2[i]) / (-rs - A.eps1eps2[2]);       // :416-418
    }
    a = A.re;
}

// :435-444 (bubble) and :459-468 (dew) written once: `spec` is the phase whose composition is
// specified (liquid for bubble, vapour for dew), `inc` the incipient phase.  -> Pa
template <class F>
F bubble_dew_formula(const MixParams<F>& q, F T, const F* rho_spec, const F* rho_inc) {
    F rho_i = rho_inc[0] + rho_inc[1];
    F y[2] = {rho_inc[0] / rho_i, rho_inc[1] / rho_i};
    F a_s, p_s, mu_s[2], v_s[2];
    derivatives_mix(q, T, rho_spec, a_s, p_s, mu_s, v_s);
    F a_i = helmholtz_energy_density_mix(q, T, rho_inc) / rho_i;
    F v = y[0] * v_s[0] + y[1] * v_s[1];
    F g = y[0] * (log(rho_inc[0] / rho_spec[0]) - mu_s[0]) + y[1] * (log(rho_inc[1] / rho_spec[1]) - mu_s[1]);
    F p = -(a_i + p_s * v + g - F(1)) / (F(1) / rho_i - v);
    return p * T * F(P_UNIT);
}

}  // namespace oracle
