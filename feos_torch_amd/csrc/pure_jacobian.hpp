// Jacobians of the pure-component properties w.r.t. (m, sigma, epsilon_k, mu, kappa_ab,
// epsilon_k_ab, na, nb, T, p) with the phase densities held fixed (device only).
//
// The reference obtains these by torch reverse mode through its Python tail
// (feos_torch/pcsaft_pure.py:196-199, :212-215, :228-233): the densities arrive detached from
// the Rust solver, so only the explicit dependence on parameters / T / p is differentiated.
// Here the same partial derivatives are propagated FORWARD with DN<double,C> tangents, C
// directions per pass, so that the whole state stays in registers (a single 10-direction pass
// would need > 512 VGPRs for the coefficient set alone).
#pragma once
#include "pure_model.hpp"

namespace pcs {

constexpr int JAC_DIRS = 10;  // 8 parameters, T, p
constexpr int PCS_JAC_CHUNK = 2;
constexpr int JAC_CHUNK = PCS_JAC_CHUNK;  // directions per pass

// Adjoint of a(rho; c) with respect to the coefficient set at fixed density, in closed form (the expressions of the
// D1s evaluation, pure_model.hpp):  g_k = da/dc_k.  Returns a.
//   a = rho (m HS - mm1 LG) + rho^2 (kd1 I1 + kd2 C I2) + rho^2 qm J1^2/(J1 - rho J2) + rho q(S; na, nb),  S = rho da h(eta)
//   d/dm = rho HS - rho^2 kd2 I2 C^2 A,  d/dmm1 = -rho LG + rho^2 kd2 I2 C^2 B   (C = 1/D, D = 1 + m A - mm1 B)
//   d/dai[k] = rho^2 kd1 eta^k,  d/dbi[k] = rho^2 kd2 C eta^k,  d/dkd1 = rho^2 I1,  d/dkd2 = rho^2 C I2
//   d/dj1[i] = rho^2 qm eta^i (2 J1/Dn - N/Dn^2),  d/dj2[i] = rho^3 qm eta^i N/Dn^2,  d/dqm = rho^2 N/Dn   (N = J1^2, Dn = J1 - rho J2)
//   d/dna = rho ln XA,  d/dnb = rho ln XB,  d/dda = rho^2 h q_S,  q_S = -na nb XA XB   (the association energy is stationary in
//   the site fractions: only the explicit dependences count)
//   d/dceta = rho da/deta with every eta-dependence above.
// R = double: the adjoints themselves; R = D1s seeded with d rho = 1: .d1 = adjoints of a' = da/drho (the pressure
// p = rho - a + rho a' of the liquid-density Jacobians).
template <class R>
struct PureCoefAdj {
    R m, mm1, ceta, ai[7], bi[7], kd1, kd2, j1[5], j2[4], qm, da, na, nb;
};
template <class R>
PCS_DEV R pure_a_adjoint(const PureCoef<double>& c, const R& r, PureCoefAdj<R>& g) {
    const R eta = r * c.ceta, r2 = r * r;
    const R u = d_recip(1.0 - eta), w2 = d_recip(2.0 - eta);
    const R u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
    const R HS = (eta * (4.0 - 3.0 * eta)) * u2, HS1 = (4.0 - 2.0 * eta) * u3;
    const R LG = d_log((1.0 - 0.5 * eta) * u3), LG1 = 3.0 * u - w2;
    R I1, I1d, I2, I2d;
    poly_and_derivative<7>(c.ai, eta, I1, I1d);
    poly_and_derivative<7>(c.bi, eta, I2, I2d);
    const R A = (eta * (8.0 - 2.0 * eta)) * u4, A1 = (8.0 + eta * (20.0 - 4.0 * eta)) * (u4 * u);
    const R poly = eta * (20.0 + eta * (-27.0 + eta * (12.0 - 2.0 * eta)));
    const R poly1 = 20.0 + eta * (-54.0 + eta * (36.0 - 8.0 * eta));
    const R q = u2 * (w2 * w2);
    const R B = poly * q, B1 = q * (poly1 + 2.0 * (poly * (u + w2)));
    const R D = 1.0 + c.m * A - c.mm1 * B, D1_ = c.m * A1 - c.mm1 * B1;
    const R C = d_recip(D), C1 = -(D1_ * (C * C));
    R a = r * (c.m * HS - c.mm1 * LG) + r2 * (c.kd1 * I1 + c.kd2 * (C * I2));
    R a_eta = r * (c.m * HS1 - c.mm1 * LG1) + r2 * (c.kd1 * I1d + c.kd2 * (C1 * I2 + C * I2d));
    const R k2 = (r2 * c.kd2) * (I2 * (C * C));
    g.m = r * HS - k2 * A;
    g.mm1 = k2 * B - r * LG;
    g.kd1 = r2 * I1;
    g.kd2 = r2 * (C * I2);
    {
        R pa = r2 * c.kd1, pb = (r2 * c.kd2) * C;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            g.ai[k] = pa;
            g.bi[k] = pb;
            pa = pa * eta;
            pb = pb * eta;
        }
    }
    if (c.polar) {
        R J1, J1d, J2, J2d;
        poly_and_derivative<5>(c.j1, eta, J1, J1d);
        poly_and_derivative<4>(c.j2, eta, J2, J2d);
        const R Dn = J1 - r * J2, rD = d_recip(Dn), N = J1 * J1, NrD2 = N * (rD * rD);
        a = a + (r2 * c.qm) * (N * rD);
        g.qm = r2 * (N * rD);
        R p1 = (r2 * c.qm) * (2.0 * (J1 * rD) - NrD2), p2 = (r2 * c.qm) * (r * NrD2);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            g.j1[i] = p1;
            if (i < 4) g.j2[i] = p2;
            p1 = p1 * eta;
            p2 = p2 * eta;
        }
        a_eta = a_eta + (r2 * c.qm) * (2.0 * ((J1 * J1d) * rD) - NrD2 * (J1d - r * J2d));
    }
    if (c.assoc) {
        const R eu = eta * u;
        const R h = u * (1.0 + eu * (1.5 + 0.5 * eu));
        const R h1 = u2 * (2.5 + eu * (4.0 + 1.5 * eu));
        const R S = (r * c.da) * h;
        const R sa = c.na * S, sb = c.nb * S, t = sb - sa, aux = 1.0 - t;
        const R sq = d_sqrt(aux * aux + 4.0 * sb);
        R xa, xb;  // cancellation-free forms, see pure_a
        const double tr = re(t);
        if (tr > 0.5) {
            xa = 2.0 * d_recip(sq + 1.0 + t);
            xb = (sq - 1.0 + t) * d_recip(2.0 * sb);
        } else if (tr < -0.5) {
            xa = (sq - 1.0 - t) * d_recip(2.0 * sa);
            xb = 2.0 * d_recip(sq + 1.0 - t);
        } else {
            xa = 2.0 * d_recip(sq + 1.0 + t);
            xb = 2.0 * d_recip(sq + 1.0 - t);
        }
        const R la = d_log(xa), lb = d_log(xb);
        a = a + r * (c.na * (la - 0.5 * xa + 0.5) + c.nb * (lb - 0.5 * xb + 0.5));
        const R qS = (-(c.na * c.nb)) * (xa * xb);
        g.na = r * la;
        g.nb = r * lb;
        g.da = (r2 * h) * qS;
        a_eta = a_eta + ((r2 * c.da) * h1) * qS;
    }
    g.ceta = r * a_eta;
    return a;
}

// adj += w * value(g)      (V: functor picking the double to use from an R-valued adjoint)
template <class R, class V>
PCS_DEV void adjoint_axpy(const PureCoef<double>& c, PureCoefAdj<double>& adj, const PureCoefAdj<R>& g, double w, V&& pick) {
    adj.m += w * pick(g.m); adj.mm1 += w * pick(g.mm1); adj.ceta += w * pick(g.ceta);
    adj.kd1 += w * pick(g.kd1); adj.kd2 += w * pick(g.kd2);
#pragma unroll
    for (int k = 0; k < 7; k++) { adj.ai[k] += w * pick(g.ai[k]); adj.bi[k] += w * pick(g.bi[k]); }
    if (c.polar) {
        adj.qm += w * pick(g.qm);
#pragma unroll
        for (int k = 0; k < 5; k++) adj.j1[k] += w * pick(g.j1[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) adj.j2[k] += w * pick(g.j2[k]);
    }
    if (c.assoc) { adj.da += w * pick(g.da); adj.na += w * pick(g.na); adj.nb += w * pick(g.nb); }
}

// WHICH: 0 vapor_pressure [Pa], 1 liquid_density [kmol/m3], 2 equilibrium_liquid_density [kmol/m3]
template <int WHICH>
PCS_DEV void pure_jacobian(const double par[8], double T, double p_pa, double rv, double rl, double g[JAC_DIRS], bool polish = false) {
    typedef DN<double, JAC_CHUNK> G;
    constexpr int NPASS = (JAC_DIRS + JAC_CHUNK - 1) / JAC_CHUNK;
    {
        // Every property is  val = F(a_c(rho_V), a_c(rho_L), a'_c(rho_L); T, p)  with the densities fixed, so its parameter
        // derivative is  sum_k abar_k dc_k/dtheta + explicit T / p terms  with the coefficient adjoints abar from closed-form
        // evaluations in plain doubles (pure_a_adjoint); only the coefficient set itself is then differentiated forward, all
        // nine directions (8 parameters, T) in ONE DN<9> pass over pure_coef -- the products abar_k c_k are accumulated as
        // the coefficients appear, so the 33 x 9 tangents never sit in registers together.
        PureCoef<double> c0;
        pure_coef<double>(c0, par, T, true);
        if (WHICH == 0 && polish) {
            // densities from the pressure-only kernel (pcs_pure_vapor_pressure: ~1e-9 from the root on ordinary rows, 1e-5 close
            // to the critical point where dp/drho -> 0): one fp64 Newton step of the coupled iteration (vle_step, quadratic)
            // before the derivatives are taken; the pressure itself is not touched
            const Eval l = pure_eval(c0, rl), v = pure_eval(c0, rv);
            const VleStep s = vle_step(l, v, rl, rv);
            if (is_finite_bits(s.dl) && is_finite_bits(s.dv) && fabs(s.dl) < 0.1 * rl && fabs(s.dv) < 0.5 * rv) {
                rl += s.dl;
                rv += s.dv;
            }
        }
        PureCoefAdj<double> adj;
        adj.m = adj.mm1 = adj.ceta = adj.kd1 = adj.kd2 = adj.qm = adj.da = adj.na = adj.nb = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) adj.ai[k] = adj.bi[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) adj.j1[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) adj.j2[k] = 0.0;
        const double inv_v = 1.0 / rv, inv_l = 1.0 / rl;
        double gT_explicit = 0.0, gP_explicit = 0.0;
        auto value = [](double x) { return x; };
        if (WHICH == 0) {
            // p_sat = K (a_V/rho_V - a_L/rho_L + ln(rho_V/rho_L)) T kB/A^3,  K = -1/(1/rho_V - 1/rho_L)   (:212-215)
            const double K = -1.0 / (inv_v - inv_l) * (T * P_UNIT);
            PureCoefAdj<double> gg;
            const double a_v = pure_a_adjoint<double>(c0, rv, gg);
            adjoint_axpy(c0, adj, gg, K * inv_v, value);
            const double a_l = pure_a_adjoint<double>(c0, rl, gg);
            adjoint_axpy(c0, adj, gg, -K * inv_l, value);
            gT_explicit = K * (a_v * inv_v - a_l * inv_l + log(rv * inv_l)) / T;
        } else {
            // liquid_density: rho - (p(rho) - p_spec)/dp (:196-199); equilibrium_liquid_density: the same with the
            // equal-area pressure pp in place of p_spec (:228-233).  At the root the tangent of the quotient is
            // -(dp_tan - dp_spec_tan)/dp up to the term (p - p_spec) dp_tan / dp^2, proportional to the residual the solve
            // left at the returned density.  Measured against the exact long-double gradient (tests/test_large_parity_gpu.py,
            // 2e5 rows): 6.6e-11 of the row's largest component for liquid_density, 1.5e-9 for equilibrium_liquid_density
            // (asserted at 1e-8; the reference's own tests compare gradients at 1e-4).  The reference's autograd carries the term.
            D2<double> a0 = pure_a<double, D2<double>>(c0, D2<double>(rl, 1.0, 0.0));
            const double dp_plain = 1.0 + rl * a0.d2;
            const double wq = -1.0 / (dp_plain * RHO_UNIT);
            PureCoefAdj<D1s> gl;
            pure_a_adjoint<D1s>(c0, D1s(rl, 1.0), gl);
            // dp/dc_k = -da/dc_k + rho da'/dc_k
            adjoint_axpy(c0, adj, gl, wq, [rl](const D1s& x) { return rl * x.d1 - x.v; });
            if (WHICH == 1) {
                const double p_spec = p_pa / (T * P_UNIT);
                gP_explicit = -wq / (T * P_UNIT);
                gT_explicit = wq * p_spec / T;
            } else {
                const double Kp = -1.0 / (inv_v - inv_l);
                PureCoefAdj<double> gv;
                pure_a_adjoint<double>(c0, rv, gv);
                adjoint_axpy(c0, adj, gv, -wq * Kp * inv_v, value);
                adjoint_axpy(c0, adj, gl, wq * Kp * inv_l, [](const D1s& x) { return x.v; });
            }
        }
        // the coefficient set block by block, each with just its own inputs seeded (pure_model.hpp): core (m, sigma, eps, T)
        // DN<4>, dipole polynomials (m, sigma, eps, mu, T) DN<5>, association prefactor (sigma, kappa_ab, eps_ab, T) DN<4>;
        // the site counts enter directly
#pragma unroll
        for (int d = 0; d < 9; d++) g[d] = 0.0;
        {
            typedef DN<double, 4> G;
            G x[4];
            const double v[4] = {par[0], par[1], par[2], T};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x[k].v = v[k];
#pragma unroll
                for (int j = 0; j < 4; j++) x[k].e[j] = (j == k) ? 1.0 : 0.0;
            }
            PureCoef<G> c;
            pure_coef_core(c, x[0], x[1], x[2], d_recip(x[3]));
            G S = c.m * adj.m + c.mm1 * adj.mm1 + c.ceta * adj.ceta + c.kd1 * adj.kd1 + c.kd2 * adj.kd2;
#pragma unroll
            for (int k = 0; k < 7; k++) S = S + c.ai[k] * adj.ai[k] + c.bi[k] * adj.bi[k];
            g[0] += S.e[0]; g[1] += S.e[1]; g[2] += S.e[2]; g[8] += S.e[3];
        }
        if (c0.polar) {
            typedef DN<double, 5> G;
            G x[5];
            const double v[5] = {par[0], par[1], par[2], par[3], T};
#pragma unroll
            for (int k = 0; k < 5; k++) {
                x[k].v = v[k];
#pragma unroll
                for (int j = 0; j < 5; j++) x[k].e[j] = (j == k) ? 1.0 : 0.0;
            }
            PureCoef<G> c;
            pure_coef_dipole(c, x[0], x[1], x[2], x[3], d_recip(x[4]));
            G S = c.qm * adj.qm;
#pragma unroll
            for (int k = 0; k < 5; k++) S = S + c.j1[k] * adj.j1[k];
#pragma unroll
            for (int k = 0; k < 4; k++) S = S + c.j2[k] * adj.j2[k];
            g[0] += S.e[0]; g[1] += S.e[1]; g[2] += S.e[2]; g[3] += S.e[3]; g[8] += S.e[4];
        }
        if (c0.assoc) {
            typedef DN<double, 4> G;
            G x[4];
            const double v[4] = {par[1], par[4], par[5], T};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x[k].v = v[k];
#pragma unroll
                for (int j = 0; j < 4; j++) x[k].e[j] = (j == k) ? 1.0 : 0.0;
            }
            const G S = pure_coef_da(x[0], x[1], x[2], d_recip(x[3])) * adj.da;
            g[1] += S.e[0]; g[4] += S.e[1]; g[5] += S.e[2]; g[8] += S.e[3];
            g[6] += adj.na;
            g[7] += adj.nb;
        }
        g[8] += gT_explicit;
        g[9] = gP_explicit;
        return;
    }
    double dp_plain = 1.0;
    if (WHICH != 0) {
        PureCoef<double> c0;
        pure_coef<double>(c0, par, T, true);
        D2<double> a0 = pure_a<double, D2<double>>(c0, D2<double>(rl, 1.0, 0.0));
        dp_plain = 1.0 + rl * a0.d2;
    }
#pragma unroll 1
    for (int pass = 0; pass < NPASS; pass++) {
        const int d0 = pass * JAC_CHUNK;
        G gp[8], gT, gP;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            gp[k].v = par[k];
#pragma unroll
            for (int j = 0; j < JAC_CHUNK; j++) gp[k].e[j] = (d0 + j == k) ? 1.0 : 0.0;
        }
        gT.v = T;
        gP.v = p_pa;
#pragma unroll
        for (int j = 0; j < JAC_CHUNK; j++) {
            gT.e[j] = (d0 + j == 8) ? 1.0 : 0.0;
            gP.e[j] = (d0 + j == 9) ? 1.0 : 0.0;
        }
        PureCoef<G> c;
        pure_coef<G>(c, gp, gT, true);
        G val;
        if (WHICH == 0) {
            // p = -(a_V/rho_V - a_L/rho_L + ln(rho_V/rho_L)) / (1/rho_V - 1/rho_L) * T * kB/A^3   (:212-215)
            G a_l = pure_a<G, G>(c, G(rl));
            G a_v = pure_a<G, G>(c, G(rv));
            double inv_v = 1.0 / rv, inv_l = 1.0 / rl;
            G num = a_v * inv_v - a_l * inv_l + log(rv * inv_l);
            val = (num * (-1.0 / (inv_v - inv_l))) * gT * P_UNIT;
        } else if (WHICH == 1) {
            // rho - (p(rho) - p_spec)/dp  (:196-199).  At the root p = p_spec, so the tangent of the quotient is
            // -(dp_tan - dp_spec_tan)/dp up to a term proportional to the residual left by the solve (measured bound above): only a
            // and a' need parameter tangents (D1<G>); dp/drho is a plain number from a D2<double> evaluation.
            typedef D1<G> R1;
            R1 r = pure_a<G, R1>(c, R1(G(rl), G(1.0)));
            G p = rl - r.v + rl * r.d1;
            G p_spec = gP / gT * (1.0 / P_UNIT);
            val = (rl - (p - p_spec) * (1.0 / dp_plain)) * (1.0 / RHO_UNIT);
        } else {
            // (:228-233), same argument: the equal-area pressure pp needs the values of a in both phases, the liquid
            // pressure a and a'
            typedef D1<G> R1;
            R1 r = pure_a<G, R1>(c, R1(G(rl), G(1.0)));
            G p_l = rl - r.v + rl * r.d1;
            double inv_v = 1.0 / rv, inv_l = 1.0 / rl;
            G a_l = r.v * inv_l;
            G a_v = pure_a<G, G>(c, G(rv)) * inv_v;
            G pp = (a_v - a_l + log(rv * inv_l)) * (-1.0 / (inv_v - inv_l));
            val = (rl - (p_l - pp) * (1.0 / dp_plain)) * (1.0 / RHO_UNIT);
        }
#pragma unroll
        for (int d = 0; d < JAC_DIRS; d++) {
#pragma unroll
            for (int j = 0; j < JAC_CHUNK; j++)
                if (d == d0 + j && d < JAC_DIRS) g[d] = val.e[j];
        }
    }
}

// Vector-Jacobian product of PcSaftPure.derivatives (feos_torch/pcsaft_pure.py:180-182): the outputs
//   a,  p = rho - a + rho a1,  dp = 1 + rho a2        (a1, a2, a3: density derivatives of a)
// are plain torch graphs in the reference, so a loss built on them back-propagates to parameters, temperature AND
// density.  Given the upstream gradients (ga, gp, gdp) this returns dL/d(8 parameters, T, rho) with
// L = ga a + gp p + gdp dp: D2<DN> evaluations (a, a1, a2 with tangents), JAC_CHUNK directions per pass; the density is
// the tenth direction (its tangent of a2 is a3) plus the explicit terms gp (1 + a1) + gdp a2.
constexpr int VJP_DIRS = 10;  // 8 parameters, T, rho
PCS_DEV void pure_derivatives_vjp(const double par[8], double T, double rho, double ga, double gp, double gdp, double g[VJP_DIRS]) {
    typedef DN<double, JAC_CHUNK> G;
    typedef D2<G> R;
    constexpr int NPASS = (VJP_DIRS + JAC_CHUNK - 1) / JAC_CHUNK;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; pass++) {
        const int d0 = pass * JAC_CHUNK;
        G gp_[8], gT, gR;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            gp_[k].v = par[k];
#pragma unroll
            for (int j = 0; j < JAC_CHUNK; j++) gp_[k].e[j] = (d0 + j == k) ? 1.0 : 0.0;
        }
        gT.v = T;
        gR.v = rho;
#pragma unroll
        for (int j = 0; j < JAC_CHUNK; j++) {
            gT.e[j] = (d0 + j == 8) ? 1.0 : 0.0;
            gR.e[j] = (d0 + j == 9) ? 1.0 : 0.0;
        }
        PureCoef<G> c;
        pure_coef<G>(c, gp_, gT, true);
        R a = pure_a<G, R>(c, R(gR, G(1.0), G(0.0)));
#pragma unroll
        for (int j = 0; j < JAC_CHUNK; j++) {
            double val = (ga - gp) * a.v.e[j] + (gp * rho) * a.d1.e[j] + (gdp * rho) * a.d2.e[j];
            if (d0 + j == 9) val += gp * (1.0 + a.d1.v) + gdp * a.d2.v;  // explicit density dependence of p and dp
#pragma unroll
            for (int d = 0; d < VJP_DIRS; d++)
                if (d == d0 + j) g[d] = val;
        }
    }
}

}  // namespace pcs
