// Binary-mixture PC-SAFT reduced residual Helmholtz energy density a(T, rho_1, rho_2) [A^-3]
// for the gfx950 kernels — the model of the reference's PcSaftMix.helmholtz_energy_density
// (feos_torch/pcsaft_mix.py:31-154) with phi_dipole (:156-208), phi_self_assoc (:210-239),
// phi_cross_assoc (:241-321), phi_induced_assoc (:324-393), association_strength (:500-522).
//
// Same two-stage split as the pure model: mix_coef() hoists everything that depends on
// (parameters, k_ij, T) only; mix_a() is the density-dependent part evaluated inside the
// bubble/dew Newton with R = T2<double> (value, gradient, Hessian w.r.t. the partial densities).
//
// Association sub-problems are solved per lane in registers: the site fractions' real parts are
// converged in plain fp64 (Newton from the reference's start value 0.2, safeguarded by a bracket
// / successive-substitution fallback where the reference's step-back rule x <- 0.2 x_old can run
// away), then two Newton updates in R arithmetic deliver their first and second density
// derivatives (implicit differentiation) — instead of the reference's <= 50 Newton sweeps over
// full DualTensors with per-element Python loops (:271-311, :363-385).
#pragma once
#include "dual.hpp"
#include "pcsaft_consts.hpp"
#include "pure_model.hpp"  // horner, site_term

namespace pcs {

enum : int { ASSOC_NONE = 0, ASSOC_SELF = 1, ASSOC_CROSS = 2, ASSOC_INDUCED = 3 };

template <class P>
struct MixCoef {
    P m[2], mm1[2], d[2];
    P zk[4][2];   // zeta_k = zk[k][0] rho_0 + zk[k][1] rho_1       (:35-38)
    P A[3], B[3]; // rho1mix = A0 r0^2 + A1 r0 r1 + A2 r1^2 (A1 holds both ij and ji), same for rho2mix (:78-90)
    bool polar;
    P pj[3][5];   // phi2 = sum_pairs r_i r_j pj[pair](eta)   pairs 00, 01, 11; all constants folded (:166-182)
    P tj[4][4];   // phi3 = sum_triplets r_i r_j r_k tj[t](eta)  triplets 000, 001, 011, 111 (:183-205)
    int acls;
    P na[2], nb[2];
    P dij[3];     // d_i d_j / (d_i + d_j) for 00, 01, 11   (self: the site-weighted d / 2 in dij[0])
    P S[3];       // (sigma_i sigma_j)^1.5 sqrt(kappa_i kappa_j) (exp(eps_ij/T) - 1)   (:506-522)
};

// Pair / triplet polynomial coefficients of the Gross-Vrabec dipole term for a binary mixture
// (feos_torch/pcsaft_mix.py:156-208 == feos_torch/gc_pcsaft.py:255-307) with every constant folded:
//   phi2 = sum_pairs r_i r_j pj[pair](eta),  phi3 = sum_triplets r_i r_j r_k tj[t](eta),
//   dipole = phi2^2 / (phi2 - phi3).   m, sig, eps = (molecule-level) m, sigma, epsilon_k;
//   mu2t_i = the reference's `mu2_term` of component i.
template <class P>
PCS_DEV void dipole_coefficients(P pj[3][5], P tj[4][4], const P* m, const P* sig, const P* eps, const P* mu2t, const P& rT) {
    P mc[2], s3[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        mc[i] = (re(m[i]) > 2.0) ? P(2.0) : m[i];
        s3[i] = sig[i] * sig[i] * sig[i];
    }
    P s01 = 0.5 * (sig[0] + sig[1]);
    P s01_3 = s01 * s01 * s01;
    P e00 = eps[0] * rT, e11 = eps[1] * rT;
    P et01 = d_sqrt(eps[0] * eps[1]) * rT;  // no k_ij here (:172)
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
        const int i = (pr == 2) ? 1 : 0, j = (pr == 0) ? 0 : 1;
        P mij = (i == j) ? mc[i] : d_sqrt(mc[0] * mc[1]);
        P rm = d_recip(mij);
        P m1 = (mij - 1.0) * rm;
        P m2 = m1 * ((mij - 2.0) * rm);
        P et = (pr == 0) ? e00 : (pr == 2 ? e11 : et01);
        P sij3 = (pr == 0) ? s3[0] : (pr == 2 ? s3[1] : s01_3);
        P pref = (mu2t[i] * mu2t[j]) * d_recip(sij3) * (-PI * ((i == j) ? 1.0 : 2.0));
#pragma unroll
        for (int n = 0; n < 5; n++) {
            P a = AD[n][0] + m1 * AD[n][1] + m2 * AD[n][2];
            if (n < 3) a = a + (BD[n][0] + m1 * BD[n][1] + m2 * BD[n][2]) * et;
            pj[pr][n] = a * pref;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {  // triplets 000, 001, 011, 111
        const int n1 = t;
        P prod = (n1 == 0) ? mc[0] * mc[0] * mc[0] : (n1 == 1) ? mc[0] * mc[0] * mc[1] : (n1 == 2) ? mc[0] * mc[1] * mc[1] : mc[1] * mc[1] * mc[1];
        P mijk = (n1 == 0) ? mc[0] : (n1 == 3) ? mc[1] : d_cbrt(prod);
        P rm = d_recip(mijk);
        P m1 = (mijk - 1.0) * rm;
        P m2 = m1 * ((mijk - 2.0) * rm);
        P sprod = (n1 == 0) ? s3[0] : (n1 == 3) ? s3[1] : (n1 == 1) ? sig[0] * (s01 * s01) : (s01 * s01) * sig[1];
        P muprod = (n1 == 0) ? mu2t[0] * mu2t[0] * mu2t[0] : (n1 == 1) ? mu2t[0] * mu2t[0] * mu2t[1] : (n1 == 2) ? mu2t[0] * mu2t[1] * mu2t[1] : mu2t[1] * mu2t[1] * mu2t[1];
        P pref = muprod * d_recip(sprod) * (-PI_SQ_43 * ((n1 == 0 || n1 == 3) ? 1.0 : 3.0));
#pragma unroll
        for (int n = 0; n < 4; n++) tj[t][n] = (CD[n][0] + m1 * CD[n][1] + m2 * CD[n][2]) * pref;
    }
}

// The coefficient set in blocks, each a function of its own few inputs (mix_coef below calls them in turn; the gradient
// kernels differentiate them one by one with just those inputs seeded, mix_jacobian.hpp).
// segment diameter (:33)
template <class P>
PCS_DEV P mix_diameter(const P& sig, const P& eps, const P& rT) { return sig * (1.0 - 0.12 * d_exp(-3.0 * (eps * rT))); }
// component i: m, mm1, d, zk
template <class C, class P>
PCS_DEV void mix_coef_component(C& c, int i, const P& m, const P& sig, const P& eps, const P& rT) {
    c.m[i] = m;
    c.mm1[i] = m - 1.0;
    c.d[i] = mix_diameter(sig, eps, rT);
    P md = c.m[i] * FRAC_PI_6;
    c.zk[0][i] = md;
    c.zk[1][i] = md * c.d[i];
    c.zk[2][i] = md * (c.d[i] * c.d[i]);
    c.zk[3][i] = md * (c.d[i] * c.d[i] * c.d[i]);
}
// dispersion aggregates (:78-90)
template <class C, class P>
PCS_DEV void mix_coef_dispersion(C& c, const P* m, const P* sig, const P* eps, const P& kij0, const P& rT) {
    P s3[2] = {sig[0] * sig[0] * sig[0], sig[1] * sig[1] * sig[1]};
    P e01 = d_sqrt(eps[0] * eps[1]) * rT * (1.0 - kij0);
    P s01 = 0.5 * (sig[0] + sig[1]);
    P s01_3 = s01 * s01 * s01;
    P e00 = eps[0] * rT, e11 = eps[1] * rT;
    c.A[0] = (m[0] * m[0]) * (e00 * s3[0]);
    c.A[1] = 2.0 * ((m[0] * m[1]) * (e01 * s01_3));
    c.A[2] = (m[1] * m[1]) * (e11 * s3[1]);
    c.B[0] = c.A[0] * e00;
    c.B[1] = c.A[1] * e01;
    c.B[2] = c.A[2] * e11;
}
// sigma^3 eps mu2 / T with mu2 = mu^2/(m sigma^3 eps) * MU2_UNIT  (:17-22, :163)
template <class P>
PCS_DEV P mix_mu2t(const P& mu, const P& m, const P& rT) { return (mu * mu) * (d_recip(m) * rT) * MU2_UNIT; }
// association class from the site counts (:118-152)
PCS_DEV int mix_assoc_class(double na0, double nb0, double na1, double nb1) {
    const int associating = (na0 + nb0 != 0.0) + (na1 + nb1 != 0.0);
    const int self_assoc = (na0 * nb0 != 0.0) + (na1 * nb1 != 0.0);
    int acls = ASSOC_NONE;
    if (associating == 1 && self_assoc == 1) acls = ASSOC_SELF;
    if (associating == 2 && self_assoc == 2) acls = ASSOC_CROSS;
    if (associating == 2 && self_assoc == 1) acls = ASSOC_INDUCED;
    return acls;
}
// association strengths and contact distances for class c.acls; c.na, c.nb, c.d must be set
template <class C, class P>
PCS_DEV void mix_coef_assoc(C& c, const P* sig, const P* kap, const P* eab, const P& kij1, const P& rT) {
    if (c.acls == ASSOC_SELF) {
        // site-weighted sigma and d, summed kappa and eps (:211-218)
        P kp = kap[0] + kap[1];
        P ea = eab[0] + eab[1];
        P rna = d_recip(c.na[0] + c.na[1]);
        P sg = (c.na[0] * sig[0] + c.na[1] * sig[1]) * rna;
        P dd = (c.na[0] * c.d[0] + c.na[1] * c.d[1]) * rna;
        c.dij[0] = 0.5 * dd;
        c.S[0] = (sg * sg * sg) * kp * (d_exp(ea * rT) - 1.0);
    } else if (c.acls == ASSOC_CROSS || c.acls == ASSOC_INDUCED) {
        c.dij[0] = 0.5 * c.d[0];
        c.dij[1] = (c.d[0] * c.d[1]) * d_recip(c.d[0] + c.d[1]);
        c.dij[2] = 0.5 * c.d[1];
        P e_cross = 0.5 * (eab[0] + eab[1]);
        if (c.acls == ASSOC_CROSS && re(kij1) != 0.0) e_cross = kij1;  // :509-514
        P ss = sig[0] * sig[1];
        c.S[0] = (sig[0] * sig[0] * sig[0]) * kap[0] * (d_exp(eab[0] * rT) - 1.0);
        c.S[1] = (ss * d_sqrt(ss)) * d_sqrt(kap[0] * kap[1]) * (d_exp(e_cross * rT) - 1.0);
        c.S[2] = (sig[1] * sig[1] * sig[1]) * kap[1] * (d_exp(eab[1] * rT) - 1.0);
    }
}

// par = [2][8] rows (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb); kij0 = k_ij,
// kij1 = explicit eps_AiBj/k or 0 (src/pcsaft.rs:163).
template <class P>
PCS_DEV void mix_coef(MixCoef<P>& c, const P* par, const P& kij0, const P& kij1, const P& T) {
    P rT = d_recip(T);
    P sig[2], eps[2], mu2t[2], kap[2], eab[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const P* p = par + 8 * i;
        sig[i] = p[1];
        eps[i] = p[2];
        mix_coef_component(c, i, p[0], sig[i], eps[i], rT);
        mu2t[i] = mix_mu2t(p[3], p[0], rT);
        kap[i] = p[4];
        eab[i] = p[5];
        c.na[i] = p[6];
        c.nb[i] = p[7];
    }
    mix_coef_dispersion(c, c.m, sig, eps, kij0, rT);

    // dipoles (:156-208)
    c.polar = (re(par[3]) != 0.0) || (re(par[8 + 3]) != 0.0);
    if (c.polar) dipole_coefficients<P>(c.pj, c.tj, c.m, sig, eps, mu2t, rT);

    c.acls = mix_assoc_class(re(c.na[0]), re(c.nb[0]), re(c.na[1]), re(c.nb[1]));
    mix_coef_assoc(c, sig, kap, eab, kij1, rT);
}

// ---- association pieces --------------------------------------------------------------------
// cross association residuals f_i = X_Ai (1 + sum_j X_Bj rhob_j D_ij) - 1 and Newton step
template <class X>
PCS_DEV void cross_step(const X& xa0, const X& xa1, const X& A0, const X& A1, const X& B0, const X& B1, const X& d00,
                        const X& d01, const X& d11, X& dx0, X& dx1) {
    X u0 = d_recip(1.0 + xa0 * (A0 * d00) + xa1 * (A1 * d01));  // X_B0
    X u1 = d_recip(1.0 + xa0 * (A0 * d01) + xa1 * (A1 * d11));  // X_B1
    X b00 = B0 * d00, b01 = B1 * d01, b10 = B0 * d01, b11 = B1 * d11;
    X s0 = b00 * u0 + b01 * u1;
    X s1 = b10 * u0 + b11 * u1;
    X f0 = xa0 * (1.0 + s0) - 1.0;
    X f1 = xa1 * (1.0 + s1) - 1.0;
    X u0s = u0 * u0, u1s = u1 * u1;
    X du0_0 = -(u0s * (A0 * d00)), du0_1 = -(u0s * (A1 * d01));
    X du1_0 = -(u1s * (A0 * d01)), du1_1 = -(u1s * (A1 * d11));
    X j00 = 1.0 + s0 + xa0 * (b00 * du0_0 + b01 * du1_0);
    X j01 = xa0 * (b00 * du0_1 + b01 * du1_1);
    X j10 = xa1 * (b10 * du0_0 + b11 * du1_0);
    X j11 = 1.0 + s1 + xa1 * (b10 * du0_1 + b11 * du1_1);
    X rdet = d_recip(j00 * j11 - j01 * j10);
    dx0 = (j11 * f0 - j01 * f1) * rdet;
    dx1 = (j00 * f1 - j10 * f0) * rdet;
}

// induced association: residual f = na0 f0 + na1 f1 (polynomial form of :364-375), f' analytic
template <class X>
PCS_DEV void induced_step(const X& xa, const X& na0, const X& na1, const X& nb0, const X& nb1, const X& d00, const X& d01,
                          const X& d10, const X& d11, X& f, X& dx) {
    X c0 = na0 * d00 + na1 * d01, c1 = na0 * d10 + na1 * d11;
    X xb0 = 1.0 + xa * c0, xb1 = 1.0 + xa * c1;
    X q = xb0 * xb1, dq = c0 * xb1 + c1 * xb0;
    X t0 = q + xb1 * (nb0 * d00) + xb0 * (nb1 * d01);
    X t1 = q + xb1 * (nb0 * d10) + xb0 * (nb1 * d11);
    X dt0 = dq + c1 * (nb0 * d00) + c0 * (nb1 * d01);
    X dt1 = dq + c1 * (nb0 * d10) + c0 * (nb1 * d11);
    X f0 = xa * t0 - q, f1 = xa * t1 - q;
    X df0 = t0 + xa * dt0 - dq, df1 = t1 + xa * dt1 - dq;
    f = na0 * f0 + na1 * f1;
    dx = f * d_recip(na0 * df0 + na1 * df1);
}

template <class R>
PCS_DEV R lift_real(double x) { return R(x); }

// Derivatives of the site fractions: Newton updates in R arithmetic starting from the converged real parts.  Update k
// makes the k-th derivatives exact, so first-order types would need one and second-order types two; the generic form
// runs two.  For R = T2 the first update only has to produce the gradient and runs in T1 arithmetic (a third of the
// multiply-adds of a T2 update; the cross-association step is ~45 products).
template <class R>
PCS_DEV void cross_refine(R& xa0, R& xa1, const R& A0, const R& A1, const R& B0, const R& B1, const R& d00, const R& d01, const R& d11) {
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        R dx0, dx1;
        cross_step<R>(xa0, xa1, A0, A1, B0, B1, d00, d01, d11, dx0, dx1);
        xa0 = xa0 - dx0;
        xa1 = xa1 - dx1;
    }
}
template <class T>
PCS_DEV void cross_refine(T2<T>& xa0, T2<T>& xa1, const T2<T>& A0, const T2<T>& A1, const T2<T>& B0, const T2<T>& B1, const T2<T>& d00,
                          const T2<T>& d01, const T2<T>& d11) {
    T1<T> y0 = lower(xa0), y1 = lower(xa1), s0, s1;
    cross_step<T1<T>>(y0, y1, lower(A0), lower(A1), lower(B0), lower(B1), lower(d00), lower(d01), lower(d11), s0, s1);
    xa0 = raise(y0 - s0);
    xa1 = raise(y1 - s1);
    T2<T> dx0, dx1;
    cross_step<T2<T>>(xa0, xa1, A0, A1, B0, B1, d00, d01, d11, dx0, dx1);
    xa0 = xa0 - dx0;
    xa1 = xa1 - dx1;
}
template <class R>
PCS_DEV void induced_refine(R& xa, const R& na0, const R& na1, const R& nb0, const R& nb1, const R& d00, const R& d01, const R& d10, const R& d11) {
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        R f, dx;
        induced_step<R>(xa, na0, na1, nb0, nb1, d00, d01, d10, d11, f, dx);
        xa = xa - dx;
    }
}
template <class T>
PCS_DEV void induced_refine(T2<T>& xa, const T2<T>& na0, const T2<T>& na1, const T2<T>& nb0, const T2<T>& nb1, const T2<T>& d00,
                            const T2<T>& d01, const T2<T>& d10, const T2<T>& d11) {
    T1<T> y = lower(xa), f1, s;
    induced_step<T1<T>>(y, lower(na0), lower(na1), lower(nb0), lower(nb1), lower(d00), lower(d01), lower(d10), lower(d11), f1, s);
    xa = raise(y - s);
    T2<T> f, dx;
    induced_step<T2<T>>(xa, na0, na1, nb0, nb1, d00, d01, d10, d11, f, dx);
    xa = xa - dx;
}

// Packing-fraction quantities shared by all contributions.  Z: type of zeta_3 and of everything that depends on it alone --
// R in general; D2<double> where the caller evaluates in the coordinates (zeta_3, rho_2) (dual.hpp, "D2 (x) T2")
template <class R, class Z = R>
struct Packing {
    R zeta2;
    Z zeta3, z3m1, z3m2, omz;
};

// hard sphere (:56-60) + dispersion (:69-106) + dipoles (:156-208) for any coefficient struct C that
// provides m[2], zk[4][2], A[3], B[3], polar, pj, tj (MixCoef, GcCoef).  Fills `pk`.
template <class C, class R, class Z>
PCS_DEV R core_terms_z(const C& c, const R& r0, const R& r1, const Z& zeta3, Packing<R, Z>& pk) {
    R zeta0 = r0 * c.zk[0][0] + r1 * c.zk[0][1];
    R zeta1 = r0 * c.zk[1][0] + r1 * c.zk[1][1];
    R zeta2 = r0 * c.zk[2][0] + r1 * c.zk[2][1];
    Z omz = 1.0 - zeta3;
    Z z3m1 = d_recip(omz);
    Z z3m2 = z3m1 * z3m1;
    R zeta23 = zeta2 * d_recip(zeta3);
    Z l13 = d_log(omz);
    pk.zeta2 = zeta2; pk.zeta3 = zeta3; pk.z3m1 = z3m1; pk.z3m2 = z3m2; pk.omz = omz;

    // hard sphere
    R a = (6.0 / PI) * (3.0 * (zeta1 * zeta2) * z3m1 + (zeta2 * zeta2) * z3m2 * zeta23 + (zeta2 * (zeta23 * zeta23) - zeta0) * l13);

    // dispersion
    R r00 = r0 * r0, r01 = r0 * r1, r11 = r1 * r1;
    R rs = r0 + r1;
    R mbar = (r0 * c.m[0] + r1 * c.m[1]) * d_recip(rs);
    R rmb = d_recip(mbar);
    R m1 = (mbar - 1.0) * rmb;
    R m2 = m1 * ((mbar - 2.0) * rmb);
    R I1 = horner_z<7>(A0, zeta3) + m1 * horner_z<7>(A1, zeta3) + m2 * horner_z<7>(A2, zeta3);
    R I2 = horner_z<7>(B0, zeta3) + m1 * horner_z<7>(B1, zeta3) + m2 * horner_z<7>(B2, zeta3);
    Z z3m4 = z3m2 * z3m2;
    Z t2 = z3m1 * d_recip(2.0 - zeta3);
    Z poly = zeta3 * (20.0 + zeta3 * (-27.0 + zeta3 * (12.0 - 2.0 * zeta3)));
    R C1 = d_recip(1.0 + mbar * ((zeta3 * (8.0 - 2.0 * zeta3)) * z3m4) + (1.0 - mbar) * (poly * (t2 * t2)));
    R rho1mix = r00 * c.A[0] + r01 * c.A[1] + r11 * c.A[2];
    R rho2mix = r00 * c.B[0] + r01 * c.B[1] + r11 * c.B[2];
    a = a - PI * (2.0 * (rho1mix * I1) + (rho2mix * (C1 * I2)) * mbar);

    // dipoles
    if (c.polar) {
        R phi2 = r00 * horner_z<5>(c.pj[0], zeta3) + r01 * horner_z<5>(c.pj[1], zeta3) + r11 * horner_z<5>(c.pj[2], zeta3);
        R phi3 = (r00 * r0) * horner_z<4>(c.tj[0], zeta3) + (r00 * r1) * horner_z<4>(c.tj[1], zeta3) +
                 (r0 * r11) * horner_z<4>(c.tj[2], zeta3) + (r11 * r1) * horner_z<4>(c.tj[3], zeta3);
        // phi2 = phi3 = 0 where no polar component is present (pure-component limit next to a polar partner): the
        // quotient's limit is phi2 + O(rho_polar^3) (value and gradient 0, Hessian that of phi2).  The same limit for a TRACE
        // polar component (|phi2| < 1e-90: partial densities below ~1e-45 of the liquid's): the quotient's second
        // derivatives carry 1/phi2^3, which overflows fp64 there and turned the Newton steps of dew rows whose incipient
        // liquid holds ~1e-50 of the polar component into NaN; the neglected phi3 term is O(rho_polar^3) < 1e-135
        if (fabs(re(phi2)) < 1e-90) a = a + phi2;
        else a = a + (phi2 * phi2) * d_recip(phi2 - phi3);
    }
    return a;
}
template <class C, class R>
PCS_DEV R core_terms(const C& c, const R& r0, const R& r1, Packing<R>& pk) {
    return core_terms_z<C, R, R>(c, r0, r1, r0 * c.zk[3][0] + r1 * c.zk[3][1], pk);
}

// The dispersion term is linear in the aggregates: a_disp = F1 (r00 A0 + r01 A1 + r11 A2) + F2 (r00 B0 + r01 B1 + r11 B2)
// with F1 = -2 pi I1, F2 = -pi m_bar C1 I2 (same expressions as in core_terms).  Used by the gc Jacobian kernel
// for d/dA_k, d/dB_k without a dual-number pass.
template <class C, class R>
PCS_DEV void dispersion_factors(const C& c, const R& r0, const R& r1, R& F1, R& F2) {
    R zeta3 = r0 * c.zk[3][0] + r1 * c.zk[3][1];
    R z3m1 = d_recip(1.0 - zeta3);
    R z3m2 = z3m1 * z3m1;
    R rs = r0 + r1;
    R mbar = (r0 * c.m[0] + r1 * c.m[1]) * d_recip(rs);
    R rmb = d_recip(mbar);
    R m1 = (mbar - 1.0) * rmb;
    R m2 = m1 * ((mbar - 2.0) * rmb);
    R I1 = horner_zeta<7>(A0, zeta3) + m1 * horner_zeta<7>(A1, zeta3) + m2 * horner_zeta<7>(A2, zeta3);
    R I2 = horner_zeta<7>(B0, zeta3) + m1 * horner_zeta<7>(B1, zeta3) + m2 * horner_zeta<7>(B2, zeta3);
    R z3m4 = z3m2 * z3m2;
    R t2 = z3m1 * d_recip(2.0 - zeta3);
    R poly = zeta3 * (20.0 + zeta3 * (-27.0 + zeta3 * (12.0 - 2.0 * zeta3)));
    R C1 = d_recip(1.0 + mbar * ((zeta3 * (8.0 - 2.0 * zeta3)) * z3m4) + (1.0 - mbar) * (poly * (t2 * t2)));
    F1 = (-2.0 * PI) * I1;
    F2 = (-PI) * ((C1 * I2) * mbar);
}

// stage 2: a(rho_0, rho_1) at fixed coefficients; zeta3 = the packing fraction as Z (see Packing)
template <class P, class R, class Z>
PCS_DEV R mix_a_z(const MixCoef<P>& c, const R& r0, const R& r1, const Z& zeta3) {
    Packing<R, Z> pk;
    R a = core_terms_z<MixCoef<P>, R, Z>(c, r0, r1, zeta3, pk);
    const R& zeta2 = pk.zeta2;
    const Z& z3m1 = pk.z3m1;

    // hard chain (:63-65):  g_i = 1/(1-z3) + 1.5 d_i c + 0.5 d_i^2 c^2 (1 - z3),  c = z2/(1-z3)^2
    R cc = zeta2 * pk.z3m2;
    {
        R cd = cc * c.d[0];
        R g = z3m1 + 1.5 * cd + 0.5 * ((cd * cd) * pk.omz);
        a = a - (r0 * c.mm1[0]) * d_log(g);
        cd = cc * c.d[1];
        g = z3m1 + 1.5 * cd + 0.5 * ((cd * cd) * pk.omz);
        a = a - (r1 * c.mm1[1]) * d_log(g);
    }

    // association (:118-152)
    if (c.acls == ASSOC_SELF) {
        R k = (zeta2 * z3m1) * c.dij[0];
        R delta = (z3m1 * (k * (2.0 * k + 3.0) + 1.0)) * c.S[0];
        R rhoa = r0 * c.na[0] + r1 * c.na[1];
        R rhob = r0 * c.nb[0] + r1 * c.nb[1];
        R t = (rhob - rhoa) * delta;
        R aux = 1.0 - t;
        R sq = d_sqrt(aux * aux + 4.0 * (rhob * delta));
        R xa, xb;  // cancellation-free site fractions, see pure_model.hpp
        double tr = re(t);
        if (tr > 0.5) {
            xa = 2.0 * d_recip(sq + 1.0 + t);
            xb = (sq - 1.0 + t) * d_recip(2.0 * (rhob * delta));
        } else if (tr < -0.5) {
            xa = (sq - 1.0 - t) * d_recip(2.0 * (rhoa * delta));
            xb = 2.0 * d_recip(sq + 1.0 - t);
        } else {
            xa = 2.0 * d_recip(sq + 1.0 + t);
            xb = 2.0 * d_recip(sq + 1.0 - t);
        }
        a = a + rhoa * site_term(xa) + rhob * site_term(xb);
    } else if (c.acls == ASSOC_CROSS || c.acls == ASSOC_INDUCED) {
        R zz = zeta2 * z3m1;
        R D[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            R k = zz * c.dij[q];
            D[q] = (z3m1 * (k * (2.0 * k + 3.0) + 1.0)) * c.S[q];
        }
        if (c.acls == ASSOC_CROSS) {
            R A0 = r0 * c.na[0], A1 = r1 * c.na[1], B0 = r0 * c.nb[0], B1 = r1 * c.nb[1];
            // real parts: Newton from 0.2 (:270) with a successive-substitution fallback
            double a0 = re(A0), a1 = re(A1), b0 = re(B0), b1 = re(B1), e00 = re(D[0]), e01 = re(D[1]), e11 = re(D[2]);
            double x0 = 0.2, x1 = 0.2;
            for (int it = 0; it < 200; it++) {
                double s0, s1;
                cross_step<double>(x0, x1, a0, a1, b0, b1, e00, e01, e11, s0, s1);
                double n0 = x0 - s0, n1 = x1 - s1;
                if (!(n0 > 0.0 && n0 <= 1.5 && n1 > 0.0 && n1 <= 1.5)) {
                    if (it < 60 && is_finite_bits(s0) && is_finite_bits(s1)) {
                        // the Newton step leaves (0, 1.5]: take it in ln X instead (X <- X exp(-dX/X), at most a
                        // factor e^3 per component, capped at 1).  Strong association puts the root at X ~ 1e-5,
                        // which the successive substitution below approaches only sub-linearly.
                        n0 = fmin(x0 * exp(fmin(fmax(-s0 / x0, -3.0), 3.0)), 1.0);
                        n1 = fmin(x1 * exp(fmin(fmax(-s1 / x1, -3.0), 3.0)), 1.0);
                    } else {
                        double u0 = 1.0 / (1.0 + x0 * a0 * e00 + x1 * a1 * e01);
                        double u1 = 1.0 / (1.0 + x0 * a0 * e01 + x1 * a1 * e11);
                        n0 = 1.0 / (1.0 + u0 * b0 * e00 + u1 * b1 * e01);
                        n1 = 1.0 / (1.0 + u0 * b0 * e01 + u1 * b1 * e11);
                    }
                }
                // 1e-12 is enough: the two Newton updates in R arithmetic below square the remaining error
                bool conv = fabs(n0 - x0) <= 1e-12 * x0 && fabs(n1 - x1) <= 1e-12 * x1;
                x0 = n0;
                x1 = n1;
                if (conv) break;
            }
            R xa0 = lift_real<R>(x0), xa1 = lift_real<R>(x1);
            cross_refine(xa0, xa1, A0, A1, B0, B1, D[0], D[1], D[2]);  // 1st and 2nd derivatives
            R xb0 = d_recip(1.0 + xa0 * (A0 * D[0]) + xa1 * (A1 * D[1]));
            R xb1 = d_recip(1.0 + xa0 * (A0 * D[1]) + xa1 * (A1 * D[2]));
            a = a + A0 * site_term(xa0) + A1 * site_term(xa1) + B0 * site_term(xb0) + B1 * site_term(xb1);
        } else {
            // delta_rho(i, j) = Delta_ij rho_j  (:341-359)
            R d00 = D[0] * r0, d01 = D[1] * r1, d10 = D[1] * r0, d11 = D[2] * r1;
            double n0 = re(c.na[0]), n1 = re(c.na[1]), m0 = re(c.nb[0]), m1n = re(c.nb[1]);
            double e00 = re(d00), e01 = re(d01), e10 = re(d10), e11 = re(d11);
            double x = 0.2, lo = 0.0, hi = 2.0;  // f(0) = -(na0+na1) < 0 < f(1): bracketed Newton
            for (int it = 0; it < 200; it++) {
                double f, s;
                induced_step<double>(x, n0, n1, m0, m1n, e00, e01, e10, e11, f, s);
                if (f == 0.0) break;
                if (f < 0.0) lo = x; else hi = x;
                double n = x - s;
                if (!(n >= lo && n <= hi && n > 0.0)) n = 0.5 * (lo + hi);
                bool conv = fabs(n - x) <= 1e-12 * x;
                x = n;
                if (conv) break;
            }
            R xa = lift_real<R>(x);
            R na0 = Lift<R, P>::go(c.na[0]), na1 = Lift<R, P>::go(c.na[1]), nb0 = Lift<R, P>::go(c.nb[0]), nb1 = Lift<R, P>::go(c.nb[1]);
            induced_refine(xa, na0, na1, nb0, nb1, d00, d01, d10, d11);
            R xb0 = d_recip(1.0 + xa * (na0 * d00 + na1 * d01));
            R xb1 = d_recip(1.0 + xa * (na0 * d10 + na1 * d11));
            R sa = site_term(xa);
            a = a + r0 * (sa * na0 + site_term(xb0) * nb0) + r1 * (sa * na1 + site_term(xb1) * nb1);
        }
    }
    return a;
}
template <class P, class R>
PCS_DEV R mix_a(const MixCoef<P>& c, const R& r0, const R& r1) {
    return mix_a_z<P, R, R>(c, r0, r1, r0 * c.zk[3][0] + r1 * c.zk[3][1]);
}

}  // namespace pcs
