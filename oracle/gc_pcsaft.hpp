// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
//
// CPU restatement of the reference's heterosegmented gc-PC-SAFT path for binary mixtures:
//   __init__ (segment / bond counting, molecule-level dipole and association parameters)
//                             <- feos_torch/gc_pcsaft.py:14-86
//   helmholtz_energy_density  <- :116-253
//   phi_dipole                <- :255-307   (pair/triplet integrals :531-546)
//   phi_assoc                 <- :309-330   (closed form, nA = nB = 1)
//   phi_cross_assoc           <- :333-380   (2x2 Newton, start 0.2, no step-back)
//   phi_induced_assoc         <- :383-441
//   association_strength      <- :549-564
//   derivatives / bubble / dew tails are the mixture ones (:443-512 == pcsaft_mix.py:395-468)
// Data layout follows the reference's dense tensors: counts[2][S], bonds[2][S][S] (lower
// triangle, :32-52), kab[S][S] symmetric (:60-63), phi[2] (:58).
#pragma once
#include <vector>
#include "constants.hpp"
#include "dual.hpp"
#include "pcsaft_mix.hpp"  // pair_integral, triplet_integral, clamp2, site_f, Dual2 helpers

namespace oracle {

// per-row structure + per-batch segment table, plain doubles (T-independent preprocessing of :55-86)
struct GcRow {
    int S;
    const double* seg;     // [S][8]  m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb
    const double* kab;     // [S][S]
    const double* counts;  // [2][S]
    const double* bonds;   // [2][S][S]
    double phi[2];
    // molecule-level parameters (:66-86)
    double m_mix[2], sigma_mix[2], epsilon_k_mix[2], mu2[2];
    double sigma_assoc[2], epsilon_k_assoc[2], kappa_ab[2], epsilon_k_ab[2], na[2], nb[2];
    bool robust = false;  // see MixParams::robust
};

inline double sgn(double x) { return (x > 0) - (x < 0); }

inline bool gc_prepare(GcRow& r) {
    for (int i = 0; i < 2; i++) {
        double mm = 0, s3 = 0, ek = 0, mu2 = 0, isum = 0, sa = 0, ea = 0, ka = 0, eab = 0, na = 0, nb = 0;
        for (int a = 0; a < r.S; a++) {
            const double* p = r.seg + 8 * a;
            double n = r.counts[i * r.S + a];
            mm += n * p[0];
            s3 += n * p[0] * p[1] * p[1] * p[1];
            ek += n * p[0] * p[2];
            mu2 += n * p[3] * p[3];
            double ia = n * sgn(p[4] * p[5]);
            isum += ia;
            sa += ia * p[1];
            ea += ia * p[2];
            ka += n * p[4];
            eab += n * p[5];
            na += n * p[6];
            nb += n * p[7];
        }
        if (isum > 1) return false;  // "Only up to one associating segment per component is allowed!" (:77-80)
        r.m_mix[i] = mm;
        r.sigma_mix[i] = std::cbrt(s3 / mm);
        r.epsilon_k_mix[i] = ek / mm;
        r.mu2[i] = mu2 / mm * (1e-19 * (1.0 / KB));
        r.sigma_assoc[i] = sa;
        r.epsilon_k_assoc[i] = ea;
        r.kappa_ab[i] = ka;
        r.epsilon_k_ab[i] = eab;
        r.na[i] = na;
        r.nb[i] = nb;
    }
    return true;
}

// :549-564
template <class S>
S gc_association_strength(int i, int j, const S& T, const double* sigma, const double* kappa_ab,
                          const double* epsilon_k_ab, const S* d, const S& zeta2, const S& zeta3_m1) {
    S k = d[i] * d[j] / (d[i] + d[j]) * zeta2 * zeta3_m1;
    double ss = sigma[i] * sigma[j];
    double sigma3_kappa = ss * std::sqrt(ss) * std::sqrt(kappa_ab[i] * kappa_ab[j]);
    double e = 0.5 * (epsilon_k_ab[i] + epsilon_k_ab[j]);
    return zeta3_m1 * (k * (2.0 * k + 3.0) + 1.0) * sigma3_kappa * (exp(S(e) / T) - 1.0);
}

template <class S>
void gc_cross_step(const S& xa0, const S& xa1, const S& d00, const S& d01, const S& d10, const S& d11, S& g0, S& g1,
                   S& dx0, S& dx1) {  // :361-374
    Dual2<S> X0(xa0, S(1.0), S(0.0)), X1(xa1, S(0.0), S(1.0));
    Dual2<S> f0 = X0 + X0 * X0 * d00 + X0 * X1 * d01 - S(1.0);
    Dual2<S> f1 = X1 + X1 * X0 * d10 + X1 * X1 * d11 - S(1.0);
    g0 = f0.re;
    g1 = f1.re;
    S det = f0.eps1 * f1.eps2 - f0.eps2 * f1.eps1;
    dx0 = (f1.eps2 * g0 - f0.eps2 * g1) / det;
    dx1 = (-1.0 * f1.eps1 * g0 + f0.eps1 * g1) / det;
}

// :116-253.  T may carry dual parts (temperature gradient); kab_ov / phi_ov let a caller pass
// S-valued copies of kab (one entry) and phi for gradients; NULL = use the plain tables.
template <class S>
S gc_helmholtz_energy_density(const GcRow& r, const S& T, const S* rho, const S* phi_s = nullptr, int ka = -1,
                              int kb = -1, const S* kab_s = nullptr) {
    const int Sg = r.S;
    std::vector<S> d(Sg);
    for (int a = 0; a < Sg; a++) {
        const double* p = r.seg + 8 * a;
        d[a] = p[1] * (1.0 - 0.12 * exp(-3.0 * S(p[2]) / T));  // :118-120
    }
    S zeta[4];
    for (int k = 0; k < 4; k++) {
        S z(0.0);
        for (int i = 0; i < 2; i++) {
            S mi(0.0);
            for (int a = 0; a < Sg; a++) {
                double m = r.counts[i * Sg + a] * r.seg[8 * a];
                if (m == 0.0) continue;
                S t(m);
                for (int q = 0; q < k; q++) t = t * d[a];
                mi = mi + t;
            }
            z = z + mi * rho[i];
        }
        zeta[k] = PI / 6.0 * z;  // :122-131
    }
    const S &zeta0 = zeta[0], &zeta1 = zeta[1], &zeta2 = zeta[2], &zeta3 = zeta[3];
    S zeta23 = zeta2 / zeta3;
    S zeta3_2 = zeta3 * zeta3;
    S zeta3_3 = zeta3_2 * zeta3;
    S zeta3_m1 = 1.0 / (1.0 - zeta3);
    S zeta3_m2 = zeta3_m1 * zeta3_m1;
    S etas[7] = {S(1.0), zeta3, zeta3_2, zeta3_3, zeta3_2 * zeta3_2, zeta3_2 * zeta3_3, zeta3_3 * zeta3_3};

    // hard sphere (:149-153)
    S hs = (6.0 / PI) * (zeta1 * zeta2 * zeta3_m1 * 3.0 + zeta2 * zeta2 * zeta3_m2 * zeta23 +
                         (zeta2 * zeta23 * zeta23 - zeta0) * log(1.0 - zeta3));

    // hard chain over bond types (:156-165)
    S c = zeta2 * zeta3_m2;
    S hc(0.0);
    for (int a = 0; a < Sg; a++) {
        for (int b = 0; b <= a; b++) {
            double n0 = r.bonds[(0 * Sg + a) * Sg + b], n1 = r.bonds[(1 * Sg + a) * Sg + b];
            if (n0 == 0.0 && n1 == 0.0) continue;
            S cdab = c * ((d[a] * d[b]) / (d[a] + d[b]));
            S g = zeta3_m1 + cdab * 3.0 - cdab * cdab * (zeta3 - 1.0) * 2.0;
            hc = hc - (rho[0] * n0 + rho[1] * n1) * log(g);
        }
    }

    // dispersion (:169-210)
    S rho_sum = rho[0] + rho[1];
    S m = (rho[0] / rho_sum) * r.m_mix[0] + (rho[1] / rho_sum) * r.m_mix[1];
    S rho1mix(0.0), rho2mix(0.0);
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 2; j++) {
            S phiij = phi_s ? phi_s[i] * phi_s[j] : S(r.phi[i] * r.phi[j]);
            for (int a = 0; a < Sg; a++) {
                double ma = r.counts[i * Sg + a] * r.seg[8 * a];
                if (ma == 0.0) continue;
                for (int b = 0; b < Sg; b++) {
                    double mb = r.counts[j * Sg + b] * r.seg[8 * b];
                    if (mb == 0.0) continue;
                    // sqrt(eps_a eps_b phi_i phi_j)/T (:181-186); the segment part stays a plain number
                    // (epsilon_k = 0 for '>C<' would make d sqrt/dx = inf in a dual)
                    S eps_ab = sqrt(phiij) * std::sqrt(r.seg[8 * a + 2] * r.seg[8 * b + 2]) / T;
                    if (i != j) {
                        bool ov = kab_s && ((a == ka && b == kb) || (a == kb && b == ka));
                        S one_minus_k = ov ? (1.0 - *kab_s) : S(1.0 - r.kab[a * Sg + b]);
                        eps_ab = eps_ab * one_minus_k;  // :187-188
                    }
                    double s = 0.5 * (r.seg[8 * a + 1] + r.seg[8 * b + 1]);
                    S rhoij = rho[i] * rho[j] * (ma * mb * (s * s * s)) * eps_ab;
                    rho1mix = rho1mix + rhoij;
                    rho2mix = rho2mix + rhoij * eps_ab;
                }
            }
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
    S phi = hs + hc + (-1.0 * rho1mix * 2.0 * I1 - rho2mix * C1 * I2 * m) * PI;

    // dipoles (:214-218, :255-307): molecule-level parameters, mu2_term = mu2 / T
    if (r.mu2[0] > 0.0 || r.mu2[1] > 0.0) {
        S mu2_term[2] = {S(r.mu2[0]) / T, S(r.mu2[1]) / T};
        S phi2(0.0), phi3(0.0);
        for (int i = 0; i < 2; i++) {
            for (int j = i; j < 2; j++) {
                double s_ij = 0.5 * (r.sigma_mix[i] + r.sigma_mix[j]);
                double mi = std::fmin(r.m_mix[i], 2.0), mj = std::fmin(r.m_mix[j], 2.0);
                double mij = std::sqrt(mi * mj);
                S mij1((mij - 1.0) / mij);
                S mij2 = mij1 * ((mij - 2.0) / mij);
                S eps_ij_t = S(std::sqrt(r.epsilon_k_mix[i] * r.epsilon_k_mix[j])) / T;
                double cc = (i == j) ? 1.0 : 2.0;
                phi2 = phi2 - rho[i] * rho[j] * mu2_term[i] * mu2_term[j] * pair_integral(mij1, mij2, etas, eps_ij_t) /
                                  (s_ij * s_ij * s_ij) * cc;
                for (int k = j; k < 2; k++) {
                    double sij = 0.5 * (r.sigma_mix[i] + r.sigma_mix[j]), sik = 0.5 * (r.sigma_mix[i] + r.sigma_mix[k]),
                           sjk = 0.5 * (r.sigma_mix[j] + r.sigma_mix[k]);
                    double mk = std::fmin(r.m_mix[k], 2.0);
                    double mijk = std::cbrt(mi * mj * mk);
                    S mijk1((mijk - 1.0) / mijk);
                    S mijk2 = mijk1 * ((mijk - 2.0) / mijk);
                    int distinct = 1 + (j != i) + (k != j);
                    double c3 = (distinct == 1) ? 1.0 : (distinct == 2 ? 3.0 : 6.0);
                    phi3 = phi3 - rho[i] * rho[j] * rho[k] * mu2_term[i] * mu2_term[j] * mu2_term[k] *
                                      triplet_integral(mijk1, mijk2, etas) / (sij * sik * sjk) * c3;
                }
            }
        }
        phi2 = phi2 * PI;
        phi3 = phi3 * (4.0 / 3.0 * PI * PI);
        // 0/0 where no polar component is present: limit phi2 + O(rho_polar^3) (see pcsaft_mix.hpp)
        // also for a trace polar component (|phi2| < 1e-90): 1/phi2^3 in the second derivatives overflows fp64 (pcsaft_mix.hpp)
        if (fabsl((long double)re(phi2)) < 1e-90L) phi = phi + phi2;
        else phi = phi + phi2 * phi2 / (phi2 - phi3);
    }

    // association (:221-251)
    int associating = (r.kappa_ab[0] * r.epsilon_k_ab[0] != 0.0) + (r.kappa_ab[1] * r.epsilon_k_ab[1] != 0.0);
    int self_assoc = (r.na[0] * r.nb[0] != 0.0) + (r.na[1] * r.nb[1] != 0.0);
    if (associating == 1 && self_assoc == 1) {
        // phi_assoc (:309-330)
        double sigma[1] = {r.sigma_assoc[0] + r.sigma_assoc[1]}, eps_k = r.epsilon_k_assoc[0] + r.epsilon_k_assoc[1];
        double kap[1] = {r.kappa_ab[0] + r.kappa_ab[1]}, eab[1] = {r.epsilon_k_ab[0] + r.epsilon_k_ab[1]};
        S dd[1] = {sigma[0] * (1.0 - 0.12 * exp(-3.0 * S(eps_k) / T))};
        S rho_a = rho[0] * sgn(r.kappa_ab[0] * r.epsilon_k_ab[0]) + rho[1] * sgn(r.kappa_ab[1] * r.epsilon_k_ab[1]);
        S deltarho = gc_association_strength(0, 0, T, sigma, kap, eab, dd, zeta2, zeta3_m1) * rho_a;
        S xa = 2.0 / (sqrt(1.0 + 4.0 * deltarho) + 1.0);
        phi = phi + rho_a * (2.0 * log(xa) - xa + 1.0);
    } else if (associating == 2) {
        S dd[2];
        for (int i = 0; i < 2; i++) dd[i] = r.sigma_assoc[i] * (1.0 - 0.12 * exp(-3.0 * S(r.epsilon_k_assoc[i]) / T));
        auto delta_rho = [&](int i, int j) {
            return gc_association_strength(i, j, T, r.sigma_assoc, r.kappa_ab, r.epsilon_k_ab, dd, zeta2, zeta3_m1) * rho[j];
        };
        S d00 = delta_rho(0, 0), d01 = delta_rho(0, 1), d10 = delta_rho(1, 0), d11 = delta_rho(1, 1);
        if (self_assoc == 2) {
            // phi_cross_assoc (:333-380), hard-coded nA = nB = 1
            S xa0(0.2), xa1(0.2), g0, g1, dx0, dx1;
            if (!r.robust) {
                int after = -1;
                for (int it = 0; it < 50; it++) {
                    gc_cross_step(xa0, xa1, d00, d01, d10, d11, g0, g1, dx0, dx1);
                    xa0 = xa0 - dx0;
                    xa1 = xa1 - dx1;
                    if (after < 0 && std::fabs((double)re(g0)) < 1e-10 && std::fabs((double)re(g1)) < 1e-10) after = 0;  // :376
                    else if (after >= 0) after++;
                    if (after >= 2) break;
                }
            } else {
                typedef decltype(re(d00)) R;
                R r00 = re(d00), r01 = re(d01), r10 = re(d10), r11 = re(d11), x0 = R(0.2), x1 = R(0.2);
                for (int it = 0; it < 500; it++) {
                    R h0, h1, e0, e1;
                    gc_cross_step<R>(x0, x1, r00, r01, r10, r11, h0, h1, e0, e1);
                    R n0 = x0 - e0, n1 = x1 - e1;
                    if (!(n0 > 0 && n0 <= R(1.5) && n1 > 0 && n1 <= R(1.5))) {
                        if (it < 60 && e0 == e0 && e1 == e1) {
                            // Newton step in ln X (see oracle/pcsaft_mix.hpp, phi_cross_assoc)
                            R q0 = -e0 / x0, q1 = -e1 / x1;
                            q0 = q0 > R(3) ? R(3) : (q0 < R(-3) ? R(-3) : q0);
                            q1 = q1 > R(3) ? R(3) : (q1 < R(-3) ? R(-3) : q1);
                            n0 = x0 * exp(q0);
                            n1 = x1 * exp(q1);
                            if (n0 > R(1)) n0 = R(1);
                            if (n1 > R(1)) n1 = R(1);
                        } else {
                            n0 = R(1) / (R(1) + x0 * r00 + x1 * r01);  // successive substitution, lands in (0, 1]
                            n1 = R(1) / (R(1) + x0 * r10 + x1 * r11);
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
                    gc_cross_step(xa0, xa1, d00, d01, d10, d11, g0, g1, dx0, dx1);
                    xa0 = xa0 - dx0;
                    xa1 = xa1 - dx1;
                }
            }
            phi = phi + rho[0] * (2.0 * log(xa0) - xa0 + 1.0) + rho[1] * (2.0 * log(xa1) - xa1 + 1.0);
        } else if (self_assoc == 1) {
            // phi_induced_assoc (:383-441), hard-coded nA = 0 on the induced component
            S na0(r.na[0]), na1(r.na[1]), nb0(r.nb[0]), nb1(r.nb[1]);
            S xa(0.2), fval, dx;
            if (!r.robust) {
                int after = -1;
                for (int it = 0; it < 50; it++) {
                    induced_newton_step(xa, na0, na1, nb0, nb1, d00, d01, d10, d11, fval, dx);
                    xa = xa - dx;  // :430 (no step-back in the gc version)
                    if (after < 0 && std::fabs((double)re(fval)) < 1e-10) after = 0;  // :432
                    else if (after >= 0) after++;
                    if (after >= 2) break;
                }
            } else {
                typedef decltype(re(d00)) R;
                R a0 = r.na[0], a1 = r.na[1], b0 = r.nb[0], b1 = r.nb[1];
                R r00 = re(d00), r01 = re(d01), r10 = re(d10), r11 = re(d11);
                R x = R(0.2), lo = R(0), hi = R(2);
                for (int it = 0; it < 500; it++) {
                    R fv, e;
                    induced_newton_step<R>(x, a0, a1, b0, b1, r00, r01, r10, r11, fv, e);
                    if (fv == 0) break;
                    if (fv < 0) lo = x; else hi = x;
                    R n = x - e;
                    if (!(n >= lo && n <= hi && n > 0)) n = R(0.5) * (lo + hi);
                    R cc = (n - x) / x;
                    x = n;
                    if ((cc < 0 ? -cc : cc) < R(1e-15)) break;
                }
                xa = S(0.0) + x * 1.0;
                for (int k = 0; k < 3; k++) {
                    induced_newton_step(xa, na0, na1, nb0, nb1, d00, d01, d10, d11, fval, dx);
                    xa = xa - dx;
                }
            }
            S xb0 = 1.0 / (1.0 + xa * (na0 * d00 + na1 * d01));
            S xb1 = 1.0 / (1.0 + xa * (na0 * d10 + na1 * d11));
            phi = phi + rho[0] * (site_f(xa) * na0 + site_f(xb0) * nb0) + rho[1] * (site_f(xa) * na1 + site_f(xb1) * nb1);
        }
    }
    return phi;
}

}  // namespace oracle
