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
#ifndef PCS_JAC_CHUNK
#define PCS_JAC_CHUNK 2
#endif
constexpr int JAC_CHUNK = PCS_JAC_CHUNK;  // directions per pass

// WHICH: 0 vapor_pressure [Pa], 1 liquid_density [kmol/m3], 2 equilibrium_liquid_density [kmol/m3]
template <int WHICH>
PCS_DEV void pure_jacobian(const double par[8], double T, double p_pa, double rv, double rl, double g[JAC_DIRS]) {
    typedef DN<double, JAC_CHUNK> G;
    constexpr int NPASS = (JAC_DIRS + JAC_CHUNK - 1) / JAC_CHUNK;
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
            // -(dp_tan - dp_spec_tan)/dp up to a term proportional to the last Newton step (~1e-12 relative): only a
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
