// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
//
// n-component PC-SAFT mixtures: CPU restatement of the parts of the reference's PcSaftMix that are written
// for any number of components --
//   helmholtz_energy_density  <- feos_torch/pcsaft_mix.py:31-154   (hs :56-60, hc :63-65, dispersion :69-106 without k_ij:
//                                "kij can only be used for binary mixtures!" :75-76, dipoles :156-208, self association of
//                                ONE associating component :210-239; two associating components are binary-only :250, :336)
//   derivatives               <- :395-420  (a, p, mu_i, v_i from one hyper-dual pass over A(N, V) = V a(N/V))
// Plain loops over the components; S is double, long double or a HyperDual.  NCMAX components at most.
#pragma once
#include "constants.hpp"
#include "dual.hpp"
#include "pcsaft_mix.hpp"  // clamp2, pair_integral, triplet_integral, site_f

namespace oracle {

constexpr int MIXN_MAX = 6;

// par [nc][8] rows (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb).  Returns false through `ok` if the rows ask for
// an association class the reference only implements for binary mixtures.
template <class S>
S helmholtz_energy_density_mixn(int nc, const double* par, const S& T, const S* rho, bool& ok) {
    ok = true;
    S d[MIXN_MAX];
    double m[MIXN_MAX], sigma[MIXN_MAX], eps[MIXN_MAX], mu2[MIXN_MAX];
    for (int i = 0; i < nc; i++) {
        const double* p = par + 8 * i;
        m[i] = p[0]; sigma[i] = p[1]; eps[i] = p[2];
        mu2[i] = p[3] * p[3] / (m[i] * sigma[i] * sigma[i] * sigma[i] * eps[i]) * 1e-19 * (1.0 / KB);  // :17-22
        d[i] = sigma[i] * (1.0 - 0.12 * exp(-3.0 * S(eps[i]) / T));                                     // :33
    }
    S zeta0(0.0), zeta1(0.0), zeta2(0.0), zeta3(0.0), rho_sum(0.0);
    for (int i = 0; i < nc; i++) {
        S mr = rho[i] * m[i];
        zeta0 = zeta0 + mr; zeta1 = zeta1 + mr * d[i]; zeta2 = zeta2 + mr * d[i] * d[i]; zeta3 = zeta3 + mr * d[i] * d[i] * d[i];
        rho_sum = rho_sum + rho[i];
    }
    zeta0 = zeta0 * (PI / 6.0); zeta1 = zeta1 * (PI / 6.0); zeta2 = zeta2 * (PI / 6.0); zeta3 = zeta3 * (PI / 6.0);  // :35-38
    S zeta23 = zeta2 / zeta3;
    S zeta3_2 = zeta3 * zeta3, zeta3_3 = zeta3_2 * zeta3;
    S zeta3_m1 = 1.0 / (1.0 - zeta3);
    S zeta3_m2 = zeta3_m1 * zeta3_m1;
    S etas[7] = {S(1.0), zeta3, zeta3_2, zeta3_3, zeta3_2 * zeta3_2, zeta3_2 * zeta3_3, zeta3_3 * zeta3_3};
    // hard sphere (:56-60)
    S phi = (6.0 / PI) * (zeta1 * zeta2 * zeta3_m1 * 3.0 + zeta2 * zeta2 * zeta3_m2 * zeta23 + (zeta2 * zeta23 * zeta23 - zeta0) * log(1.0 - zeta3));
    // hard chain (:63-65)
    S c = zeta2 * zeta3_m2;
    for (int i = 0; i < nc; i++) {
        S g = zeta3_m1 + d[i] * c * 1.5 - d[i] * d[i] * c * c * (zeta3 - 1.0) * 0.5;
        phi = phi - rho[i] * (m[i] - 1.0) * log(g);
    }
    // dispersion (:69-106), no k_ij for nc != 2
    S mbar(0.0);
    for (int i = 0; i < nc; i++) mbar = mbar + (rho[i] / rho_sum) * m[i];
    S rho1mix(0.0), rho2mix(0.0);
    for (int i = 0; i < nc; i++)
        for (int j = 0; j < nc; j++) {
            S eps_ij = S(std::sqrt(eps[i] * eps[j])) / T;
            double s = 0.5 * (sigma[i] + sigma[j]);
            S rhoij = rho[i] * rho[j] * (m[i] * m[j] * (s * s * s)) * eps_ij;
            rho1mix = rho1mix + rhoij;
            rho2mix = rho2mix + rhoij * eps_ij;
        }
    S I1(0.0), I2(0.0);
    S m1 = (mbar - 1.0) / mbar;
    S m2 = m1 * (mbar - 2.0) / mbar;
    for (int i = 0; i < 7; i++) {
        I1 = I1 + (m2 * A2[i] + m1 * A1[i] + A0[i]) * etas[i];
        I2 = I2 + (m2 * B2[i] + m1 * B1[i] + B0[i]) * etas[i];
    }
    S C1 = 1.0 / (1.0 + mbar * (8.0 * zeta3 - 2.0 * zeta3_2) * zeta3_m2 * zeta3_m2 +
                  (1.0 - mbar) * (20.0 * zeta3 - 27.0 * zeta3_2 + 12.0 * zeta3_2 * zeta3 - 2.0 * zeta3_2 * zeta3_2) /
                      ((1.0 - zeta3) * (1.0 - zeta3) * (2.0 - zeta3) * (2.0 - zeta3)));
    phi = phi + (-1.0 * rho1mix * 2.0 * I1 - rho2mix * C1 * I2 * mbar) * PI;
    // dipoles (:156-208)
    bool dipolar = false;
    for (int i = 0; i < nc; i++) dipolar = dipolar || mu2[i] > 0.0;
    if (dipolar) {
        S mu2_term[MIXN_MAX];
        for (int i = 0; i < nc; i++) mu2_term[i] = S(sigma[i] * sigma[i] * sigma[i] * eps[i] * mu2[i]) / T;  // :163
        S phi2(0.0), phi3(0.0);
        for (int i = 0; i < nc; i++)
            for (int j = i; j < nc; j++) {
                double s_ij = 0.5 * (sigma[i] + sigma[j]);
                double mij = std::sqrt(std::fmin(m[i], 2.0) * std::fmin(m[j], 2.0));
                S mij1((mij - 1.0) / mij);
                S mij2 = mij1 * ((mij - 2.0) / mij);
                S eps_ij_t = S(std::sqrt(eps[i] * eps[j])) / T;
                double cc = (i == j) ? 1.0 : 2.0;
                phi2 = phi2 - rho[i] * rho[j] * mu2_term[i] * mu2_term[j] * pair_integral(mij1, mij2, etas, eps_ij_t) / (s_ij * s_ij * s_ij) * cc;
                for (int k = j; k < nc; k++) {
                    double sij = 0.5 * (sigma[i] + sigma[j]), sik = 0.5 * (sigma[i] + sigma[k]), sjk = 0.5 * (sigma[j] + sigma[k]);
                    double mijk = std::cbrt(std::fmin(m[i], 2.0) * std::fmin(m[j], 2.0) * std::fmin(m[k], 2.0));
                    S mijk1((mijk - 1.0) / mijk);
                    S mijk2 = mijk1 * ((mijk - 2.0) / mijk);
                    int distinct = 1 + (j != i) + (k != j);
                    double c3 = (distinct == 1) ? 1.0 : (distinct == 2 ? 3.0 : 6.0);
                    phi3 = phi3 - rho[i] * rho[j] * rho[k] * mu2_term[i] * mu2_term[j] * mu2_term[k] * triplet_integral(mijk1, mijk2, etas) / (sij * sik * sjk) * c3;
                }
            }
        phi2 = phi2 * PI;
        phi3 = phi3 * (4.0 / 3.0 * PI * PI);
        if (re(phi2) == 0.0) phi = phi + phi2;  // 0/0 where no polar component is present (see pcsaft_mix.hpp)
        else phi = phi + phi2 * phi2 / (phi2 - phi3);
    }
    // association (:118-152): one associating component -> phi_self_assoc (:210-239); more -> binary only
    int associating = 0, self_assoc = 0;
    for (int i = 0; i < nc; i++) {
        const double* p = par + 8 * i;
        associating += (p[6] + p[7] != 0.0);
        self_assoc += (p[6] * p[7] != 0.0);
    }
    if (associating > 1) { ok = false; return phi; }
    if (associating == 1 && self_assoc == 1) {
        double kap = 0, eab = 0, na_sum = 0, sg = 0;
        S dd(0.0), rhoa(0.0), rhob(0.0);
        for (int i = 0; i < nc; i++) {
            const double* p = par + 8 * i;
            kap += p[4]; eab += p[5]; na_sum += p[6]; sg += p[6] * sigma[i];
            dd = dd + d[i] * p[6];
            rhoa = rhoa + rho[i] * p[6];
            rhob = rhob + rho[i] * p[7];
        }
        sg /= na_sum;
        dd = dd * (1.0 / na_sum);
        // association_strength(0, 0, ...) (:500-522) with the site-weighted sigma and d
        S k = dd * dd / (dd + dd) * zeta2 * zeta3_m1;
        S delta = zeta3_m1 * (k * (2.0 * k + 3.0) + 1.0) * (sg * sg * sg * kap) * (exp(S(eab) / T) - 1.0);
        S xa, xb;
        site_fractions_two_types(rhoa, rhob, delta, xa, xb);  // :235-238
        phi = phi + rhoa * site_f(xa) + rhob * site_f(xb);
    }
    return phi;
}

// :395-420 for nc components: a, p, mu[nc], v[nc]
template <class F>
bool derivatives_mixn(int nc, const double* par, F T, const F* rho, F& a, F& p, F* mu, F* v) {
    typedef HyperDual<F, MIXN_MAX + 1> H;
    auto lift = [](F x) { H h; h.re = x; return h; };
    H volume = lift(F(1));
    volume.eps1[nc] = F(1);
    volume.eps2 = F(1);
    H dens[MIXN_MAX];
    F rs = 0;
    for (int i = 0; i < nc; i++) {
        H moles = lift(rho[i]);
        moles.eps1[i] = F(1);
        dens[i] = moles / volume;
        rs += rho[i];
    }
    bool ok;
    H A = helmholtz_energy_density_mixn<H>(nc, par, lift(T), dens, ok) * volume;
    p = rs - A.eps2;
    for (int i = 0; i < nc; i++) {
        mu[i] = A.eps1[i];
        v[i] = -(F(1) - A.eps1eps2[i]) / (-rs - A.eps1eps2[nc]);
    }
    a = A.re;
    return ok;
}

}  // namespace oracle
