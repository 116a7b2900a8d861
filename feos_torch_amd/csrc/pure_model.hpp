// Pure-component PC-SAFT reduced residual Helmholtz energy density a(T, rho) [A^-3] for the
// gfx950 kernels — the model of the reference's PcSaftPure.helmholtz_energy
// (feos_torch/pcsaft_pure.py:106-178), re-derived for a one-state-point-per-lane kernel:
//
//   stage 1  pure_coef():  everything that depends on (parameters, T) only — segment diameter
//            (:108), packing-fraction factor (:110), the 7+7 dispersion polynomial coefficients
//            (:129-133), the 5+4 dipole coefficients (:146-154), prefactors and the
//            association strength prefactor (:163-167).  Done ONCE per state point.
//   stage 2  pure_a():     the density-dependent part, called inside the Newton loops.  Horner
//            polynomials instead of the reference's power table (:115), one log for the chain
//            term, and a cancellation-free form of the association site fractions (see below).
//
// P = type of parameters/coefficients (double in solvers, a dual in gradient kernels),
// R = type of the density (D2<double> in solvers).  Either P is double or P == R.
#pragma once
#include "dual.hpp"
#include "pcsaft_consts.hpp"

namespace pcs {

template <class P>
struct PureCoef {
    P m, mm1, ceta;    // m, m-1, eta = ceta * rho
    P ai[7], bi[7];    // I1, I2 polynomial coefficients in eta
    P kd1, kd2;        // disp = rho^2 (kd1 I1 + kd2 C1 I2)
    P j1[5], j2[4];    // dipole pair / triplet polynomials with -pi/s3 resp. -4/3 pi^2 mu2t/s3 folded in
    P qm;              // mu2t^2:  dipole = rho^2 qm J1^2 / (J1 - rho J2)
    P da;              // (exp(eps_AB/T) - 1) sigma^3 kappa_AB
    P na, nb;
    bool polar, assoc;
};

// stage 1.  par = (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb)  (README.md:12)
// for_gradient keeps terms whose VALUE vanishes but whose parameter derivative does not
// (kappa_ab = 0 or epsilon_k_ab = 0 with sites present).
// The coefficient set in blocks, each a function of its own few inputs (pure_coef calls them in turn; the Jacobian kernel
// differentiates them one by one with just those inputs seeded, pure_jacobian.hpp).  C: any struct with the members written.
// core (m, sigma, eps, 1/T): m, mm1, ceta, ai, bi, kd1, kd2
template <class C, class P>
PCS_DEV void pure_coef_core(C& c, const P& m, const P& sigma, const P& eps, const P& rT) {
    P s3 = sigma * sigma * sigma;
    P e = eps * rT;
    P d = sigma * (1.0 - 0.12 * d_exp(-3.0 * e));  // :108
    c.m = m;
    c.mm1 = m - 1.0;
    c.ceta = FRAC_PI_6 * (m * (d * d * d));  // :110
    P rm = d_recip(m);
    P m1 = c.mm1 * rm;
    P m2 = (m - 2.0) * rm;
#pragma unroll
    for (int i = 0; i < 7; i++) {  // :131-133
        c.ai[i] = m1 * (m2 * A2[i] + A1[i]) + A0[i];
        c.bi[i] = m1 * (m2 * B2[i] + B1[i]) + B0[i];
    }
    P pref = (-PI) * ((m * m) * (e * s3));  // :142
    c.kd1 = 2.0 * pref;
    c.kd2 = pref * (m * e);  // :141
}
// dipole (m, sigma, eps, mu, 1/T): j1, j2, qm (:145-160).  mu2 e s3 = mu^2 / (m s3 eps) * MU2_UNIT * (eps/T) * s3
template <class C, class P>
PCS_DEV void pure_coef_dipole(C& c, const P& m, const P& sigma, const P& eps, const P& mu, const P& rT) {
    P s3 = sigma * sigma * sigma;
    P e = eps * rT;
    P rm = d_recip(m);
    P m1 = (m - 1.0) * rm;
    P m2 = (m - 2.0) * rm;
    P mu2t = (mu * mu) * (rm * rT) * MU2_UNIT;
    bool clamp = re(m) > 2.0;  // :146
    P md1 = clamp ? P(0.5) : m1;
    P md2 = clamp ? P(0.0) : md1 * m2;
    // phi2 = rho^2 (-pi/s3) J1, phi3 = rho^3 (-4/3 pi^2/s3) J2 (:158-159);
    // dipole = phi2^2 mu2t^2 / (phi2 - phi3 mu2t) = rho^2 mu2t^2 J1'^2 / (J1' - rho J2')
    // with J1' = (-pi/s3) J1 and J2' = (-4/3 pi^2 mu2t/s3) J2.
    P rs3 = d_recip(s3);
    P f2 = (-PI) * rs3;
    P f3 = (-PI_SQ_43) * (rs3 * mu2t);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        P a = AD[i][0] + md1 * AD[i][1] + md2 * AD[i][2];
        if (i < 3) a = a + (BD[i][0] + md1 * BD[i][1] + md2 * BD[i][2]) * e;
        c.j1[i] = a * f2;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) c.j2[i] = (CD[i][0] + md1 * CD[i][1] + md2 * CD[i][2]) * f3;
    c.qm = mu2t * mu2t;
}
// association strength prefactor (sigma, kappa_ab, eps_ab, 1/T) (:163-167)
template <class P>
PCS_DEV P pure_coef_da(const P& sigma, const P& kap, const P& eab, const P& rT) {
    P s3 = sigma * sigma * sigma;
    return (d_exp(eab * rT) - 1.0) * s3 * kap;
}

template <class P>
PCS_DEV void pure_coef(PureCoef<P>& c, const P* par, const P& T, bool for_gradient) {
    P rT = d_recip(T);
    pure_coef_core(c, par[0], par[1], par[2], rT);
    c.polar = re(par[3]) != 0.0;
    if (c.polar) pure_coef_dipole(c, par[0], par[1], par[2], par[3], rT);
    c.na = par[6];
    c.nb = par[7];
    bool sites = (re(par[6]) != 0.0) || (re(par[7]) != 0.0);
    c.da = pure_coef_da(par[1], par[4], par[5], rT);
    c.assoc = sites && (for_gradient || re(c.da) != 0.0);
}

template <int N, class P, class R>
PCS_DEV R horner(const P* coef, const R& x) {
    R acc = x * coef[N - 1] + coef[N - 2];
#pragma unroll
    for (int i = N - 3; i >= 0; i--) acc = acc * x + coef[i];
    return acc;
}

// Horner in a two-variable Taylor type (mixture / gc evaluations): P, P', P''/2 by the three-term recurrence in the
// value type T (3 FMA per coefficient, one dependent FMA per step) and ONE chain rule at the end, instead of a full
// T2 (T1) product per coefficient (~15 (5) multiply-adds, three dependent).  N >= 3.
template <int N, class P, class T>
PCS_DEV T2<T> horner_zeta(const P* coef, const T2<T>& x) {
    T d2 = T(coef[N - 1]);
    T d1 = d2 * x.v + coef[N - 2];
    T p = d1 * x.v + coef[N - 3];
    d1 = d2 * x.v + d1;
#pragma unroll
    for (int i = N - 4; i >= 0; i--) {
        d2 = d2 * x.v + d1;
        d1 = d1 * x.v + p;
        p = p * x.v + coef[i];
    }
    return x.chain(p, d1, 2.0 * d2);
}
template <int N, class P, class T>
PCS_DEV T1<T> horner_zeta(const P* coef, const T1<T>& x) {
    T d1 = T(coef[N - 1]);
    T p = d1 * x.v + coef[N - 2];
#pragma unroll
    for (int i = N - 3; i >= 0; i--) {
        d1 = d1 * x.v + p;
        p = p * x.v + coef[i];
    }
    return x.chain(p, d1);
}
template <int N, class P, class T, PCS_IFDUAL(T)>
PCS_DEV D1<T> horner_zeta(const P* coef, const D1<T>& x) {
    T d1 = T(coef[N - 1]);
    T p = d1 * x.v + coef[N - 2];
#pragma unroll
    for (int i = N - 3; i >= 0; i--) {
        d1 = d1 * x.v + p;
        p = p * x.v + coef[i];
    }
    return x.chain(p, d1);
}
template <int N, class P, class R>
PCS_DEV R horner_zeta(const P* coef, const R& x) { return horner<N>(coef, x); }

// Horner for the solver's R = D2<double> when x.d2 == 0 structurally (x = eta = ceta * rho):
// P, P', P'' by the three-term recurrence (3 FMA per coefficient instead of a full D2 product).
template <int N>
PCS_DEV D2<double> horner_lin(const double* coef, const D2<double>& x) {
    double p = coef[N - 1], d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int i = N - 2; i >= 0; i--) {
        d2 = __builtin_fma(d2, x.v, d1);
        d1 = __builtin_fma(d1, x.v, p);
        p = __builtin_fma(p, x.v, coef[i]);
    }
    return D2<double>(p, d1 * x.d1, 2.0 * d2 * (x.d1 * x.d1));
}
// polynomial in the packing fraction zeta_3 given as Z: the two-variable Taylor types take horner_zeta (recurrence + one chain
// rule); where zeta_3 is itself the first coordinate (Z = D2<double> with d1 = 1, d2 = 0) the recurrence IS the result
template <int N, class P, class Z>
PCS_DEV Z horner_z(const P* coef, const Z& x) { return horner_zeta<N>(coef, x); }
template <int N>
PCS_DEV D2<double> horner_z(const double* coef, const D2<double>& x) { return horner_lin<N>(coef, x); }

// generic fall-back (gradient kernels): plain Horner in R arithmetic
template <int N, class P, class R>
PCS_DEV R horner_eta(const P* coef, const R& x) { return horner<N>(coef, x); }
template <int N>
PCS_DEV D2<double> horner_eta(const double* coef, const D2<double>& x) { return horner_lin<N>(coef, x); }
// value + first derivative: 2 FMA per coefficient
template <int N>
PCS_DEV D1s horner_eta(const double* coef, const D1s& x) {
    double p = coef[N - 1], d1 = 0.0;
#pragma unroll
    for (int i = N - 2; i >= 0; i--) {
        d1 = __builtin_fma(d1, x.v, p);
        p = __builtin_fma(p, x.v, coef[i]);
    }
    return D1s(p, d1 * x.d1);
}

// sum coef[k] x^k and its x-derivative in R arithmetic (the coefficient adjoints of the Jacobian kernels)
template <int N, class R>
PCS_DEV void poly_and_derivative(const double* coef, const R& x, R& p, R& dp) {  // sum coef[k] x^k and its x-derivative
    p = x * coef[N - 1] + coef[N - 2];
    dp = x * ((N - 1) * coef[N - 1]) + (N - 2) * coef[N - 2];
#pragma unroll
    for (int k = N - 3; k >= 0; k--) {
        p = p * x + coef[k];
        if (k >= 1) dp = dp * x + k * coef[k];
    }
}
template <class R>
PCS_DEV R site_term(const R& x) {  // ln x - x/2 + 1/2   (:176)
    return d_log(x) - 0.5 * x + 0.5;
}

// stage 2: a(rho) at fixed coefficients
template <class P, class R>
PCS_DEV R pure_a(const PureCoef<P>& c, const R& rho) {
    R eta = rho * c.ceta;
    R eta2 = eta * eta;
    R eta_m1 = d_recip(1.0 - eta);
    R eta_m2 = eta_m1 * eta_m1;

    // hard sphere (:118) + hard chain (:121-122)
    R mrho = rho * c.m;
    R hs = mrho * ((4.0 * eta - 3.0 * eta2) * eta_m2);
    R g = (1.0 - 0.5 * eta) * (eta_m1 * eta_m2);
    R hc = (rho * c.mm1) * d_log(g);

    // dispersion (:125-142)
    R I1 = horner_eta<7>(c.ai, eta);
    R I2 = horner_eta<7>(c.bi, eta);
    R eta_m4 = eta_m2 * eta_m2;
    R t2 = eta_m1 * d_recip(2.0 - eta);
    R poly = eta * (20.0 + eta * (-27.0 + eta * (12.0 - 2.0 * eta)));
    R C1 = d_recip(1.0 + (eta * (8.0 - 2.0 * eta)) * eta_m4 * c.m - (poly * (t2 * t2)) * c.mm1);
    R rho2 = rho * rho;
    R a = hs - hc + rho2 * (I1 * c.kd1 + (C1 * I2) * c.kd2);

    // dipoles (:145-160)
    if (c.polar) {
        R J1 = horner_eta<5>(c.j1, eta);
        R J2 = horner_eta<4>(c.j2, eta);
        a = a + (rho2 * c.qm) * ((J1 * J1) * d_recip(J1 - rho * J2));
    }

    // association (:163-176)
    if (c.assoc) {
        R k = eta * eta_m1;
        R delta = ((1.0 + k * (1.5 + 0.5 * k)) * eta_m1) * c.da;
        R rhoa = rho * c.na;
        R rhob = rho * c.nb;
        R t = (rhob - rhoa) * delta;
        R aux = 1.0 - t;
        R sq = d_sqrt(aux * aux + 4.0 * (rhob * delta));
        // Site fractions.  As written in the reference, xa = 2/(sq+1+t), xb = 2/(sq+1-t); one of
        // the two denominators cancels catastrophically when |t| >> 1 (strong association, low
        // T: up to 8 digits lost in fp64).  (sq+1+t)(sq-1-t) = 4 rhoa delta and
        // (sq+1-t)(sq-1+t) = 4 rhob delta give the algebraically identical conjugate forms.
        R xa, xb;
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
    }
    return a;
}

// The pressure-only fp64 finish (R = D1s: a and a') with hard sphere, chain and dispersion in closed form, as the fp32
// pre-solve does (pure_f32.hpp): a = rho F(eta) + rho^2 G(eta),
//   F = m HS - (m-1) ln g,  HS = (4 eta - 3 eta^2) u^2,  HS' = (4 - 2 eta) u^3,  (ln g)' = 3u - w,  u = 1/(1-eta), w = 1/(2-eta)
//   G = kd1 I1 + kd2 C I2,  C = 1/D,  D = 1 + m A - (m-1) B,  A' = (8 + 20 eta - 4 eta^2) u^5,  B' = q (poly' + 2 poly (u + w)), q = u^2 w^2
template <>
PCS_DEV D1s pure_a<double, D1s>(const PureCoef<double>& c, const D1s& rho) {
    const double r = rho.v, eta = r * c.ceta;
    const double u = d_recip(1.0 - eta), w = d_recip(2.0 - eta);
    const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
    const double HS = eta * (4.0 - 3.0 * eta) * u2, HS1 = (4.0 - 2.0 * eta) * u3;
    const double LG = d_log((1.0 - 0.5 * eta) * u3), LG1 = 3.0 * u - w;
    const double F = c.m * HS - c.mm1 * LG, F1 = c.m * HS1 - c.mm1 * LG1;
    const D1s e1(eta, 1.0);
    const D1s I1 = horner_eta<7>(c.ai, e1), I2 = horner_eta<7>(c.bi, e1);  // value and eta-derivative
    const double A = eta * (8.0 - 2.0 * eta) * u4, A1 = (8.0 + eta * (20.0 - 4.0 * eta)) * (u4 * u);
    const double poly = eta * (20.0 + eta * (-27.0 + eta * (12.0 - 2.0 * eta)));
    const double poly1 = 20.0 + eta * (-54.0 + eta * (36.0 - 8.0 * eta));
    const double q = u2 * (w * w);
    const double B = poly * q, B1 = q * (poly1 + 2.0 * poly * (u + w));
    const double D = 1.0 + c.m * A - c.mm1 * B, D1_ = c.m * A1 - c.mm1 * B1;
    const double C = d_recip(D), C1 = -D1_ * (C * C);
    const double G = c.kd1 * I1.v + c.kd2 * (C * I2.v);
    const double G1 = c.kd1 * I1.d1 + c.kd2 * (C1 * I2.v + C * I2.d1);
    const double rc = r * c.ceta;
    D1s a(r * (F + r * G), (F + rc * F1 + r * (2.0 * G + rc * G1)) * rho.d1);
    if (c.polar || c.assoc) {
        const D1s eta_d = rho * c.ceta;
        if (c.polar) {
            D1s J1 = horner_eta<5>(c.j1, eta_d);
            D1s J2 = horner_eta<4>(c.j2, eta_d);
            a = a + ((rho * rho) * c.qm) * ((J1 * J1) * d_recip(J1 - rho * J2));
        }
        if (c.assoc) {
            // closed form, value and first derivative (see assoc_closed_f32, pure_f32.hpp): a_assoc = rho q(S),
            // a' = q + rho q_S S',  q_S = -na nb XA XB (the energy is stationary in the site fractions), S = rho da h(eta)
            const double u_ = d_recip(1.0 - eta), eu = eta * u_;
            const double h = u_ * (1.0 + eu * (1.5 + 0.5 * eu));
            const double h1 = (u_ * u_) * (2.5 + eu * (4.0 + 1.5 * eu));
            const double S = r * c.da * h, S1 = c.da * (h + eta * h1);
            const double sa = c.na * S, sb = c.nb * S, t = sb - sa, aux = 1.0 - t;
            const double sq = d_sqrt(aux * aux + 4.0 * sb);
            double xa, xb;
            if (t > 0.5) {
                xa = 2.0 * d_recip(sq + 1.0 + t);
                xb = (sq - 1.0 + t) * d_recip(2.0 * sb);
            } else if (t < -0.5) {
                xa = (sq - 1.0 - t) * d_recip(2.0 * sa);
                xb = 2.0 * d_recip(sq + 1.0 - t);
            } else {
                xa = 2.0 * d_recip(sq + 1.0 + t);
                xb = 2.0 * d_recip(sq + 1.0 - t);
            }
            const double q = c.na * (d_log(xa) - 0.5 * xa + 0.5) + c.nb * (d_log(xb) - 0.5 * xb + 0.5);
            const double q1 = -(c.na * c.nb) * (xa * xb);
            a = a + D1s(r * q, (q + r * q1 * S1) * rho.d1);
        }
    }
    return a;
}

// The same for R = D2<double> (a, a', a'': the solvers' Newton evaluations): second derivatives
// HS'' = (10 - 4 eta) u^4, (ln g)'' = 3u^2 - w^2, A'' = (60 + 72 eta - 12 eta^2) u^6,
// B'' = 2 q s (poly' + 2 poly s) + q (poly'' + 2 poly' s + 2 poly (u^2 + w^2)), s = u + w, C'' = (2 D'^2 C - D'') C^2.
template <>
PCS_DEV D2<double> pure_a<double, D2<double>>(const PureCoef<double>& c, const D2<double>& rho) {
    const double r = rho.v, eta = r * c.ceta;
    const double u = d_recip(1.0 - eta), w = d_recip(2.0 - eta);
    const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2, w2 = w * w;
    const double HS = eta * (4.0 - 3.0 * eta) * u2, HS1 = (4.0 - 2.0 * eta) * u3, HS2 = (10.0 - 4.0 * eta) * u4;
    const double LG = d_log((1.0 - 0.5 * eta) * u3), LG1 = 3.0 * u - w, LG2 = 3.0 * u2 - w2;
    const double F = c.m * HS - c.mm1 * LG, F1 = c.m * HS1 - c.mm1 * LG1, F2 = c.m * HS2 - c.mm1 * LG2;
    const D2<double> e2(eta, 1.0, 0.0);
    const D2<double> I1 = horner_eta<7>(c.ai, e2), I2 = horner_eta<7>(c.bi, e2);  // value, eta-derivatives
    const double A = eta * (8.0 - 2.0 * eta) * u4, A1 = (8.0 + eta * (20.0 - 4.0 * eta)) * (u4 * u),
                 A2 = (60.0 + eta * (72.0 - 12.0 * eta)) * (u4 * u2);
    const double poly = eta * (20.0 + eta * (-27.0 + eta * (12.0 - 2.0 * eta)));
    const double poly1 = 20.0 + eta * (-54.0 + eta * (36.0 - 8.0 * eta)), poly2 = -54.0 + eta * (72.0 - 24.0 * eta);
    const double q = u2 * w2, sm = u + w;
    const double tq = poly1 + 2.0 * poly * sm;
    const double B = poly * q, B1 = q * tq, B2 = q * (2.0 * sm * tq + poly2 + 2.0 * poly1 * sm + 2.0 * poly * (u2 + w2));
    const double D = 1.0 + c.m * A - c.mm1 * B, D1_ = c.m * A1 - c.mm1 * B1, D2_ = c.m * A2 - c.mm1 * B2;
    const double C = d_recip(D), Csq = C * C;
    const double C1 = -D1_ * Csq, C2 = (2.0 * D1_ * D1_ * C - D2_) * Csq;
    const double G = c.kd1 * I1.v + c.kd2 * (C * I2.v);
    const double G1 = c.kd1 * I1.d1 + c.kd2 * (C1 * I2.v + C * I2.d1);
    const double G2 = c.kd1 * I1.d2 + c.kd2 * (C2 * I2.v + 2.0 * C1 * I2.d1 + C * I2.d2);
    const double ce = c.ceta, rc = r * ce;
    const double a0 = r * (F + r * G);
    const double a1 = F + rc * F1 + r * (2.0 * G + rc * G1);
    const double a2 = ce * (2.0 * F1 + rc * F2) + 2.0 * G + rc * (4.0 * G1 + rc * G2);
    D2<double> a(a0, a1 * rho.d1, a2 * (rho.d1 * rho.d1) + a1 * rho.d2);
    if (c.polar || c.assoc) {
        typedef D2<double> R;
        const R eta_d = rho * c.ceta;
        if (c.polar) {
            R J1 = horner_eta<5>(c.j1, eta_d);
            R J2 = horner_eta<4>(c.j2, eta_d);
            a = a + ((rho * rho) * c.qm) * ((J1 * J1) * d_recip(J1 - rho * J2));
        }
        if (c.assoc) {
            R eta_m1 = d_recip(1.0 - eta_d);
            R k = eta_d * eta_m1;
            R delta = ((1.0 + k * (1.5 + 0.5 * k)) * eta_m1) * c.da;
            R rhoa = rho * c.na;
            R rhob = rho * c.nb;
            R t = (rhob - rhoa) * delta;
            R aux = 1.0 - t;
            R sq = d_sqrt(aux * aux + 4.0 * (rhob * delta));
            R xa, xb;  // conjugate forms, see pure_a above
            const double tr = re(t);
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
        }
    }
    return a;
}

}  // namespace pcs
