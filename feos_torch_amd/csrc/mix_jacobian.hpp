// Gradient of the bubble / dew pressure w.r.t. (16 component parameters, k_ij, eps_AiBj, T) for
// binary mixtures (device only).
//
// The reference gets it by torch reverse mode through its explicit final Newton step
// (feos_torch/pcsaft_mix.py:435-444 / :459-468), constructed so that the result equals the
// implicit-function derivative of the converged pressure.  Here the implicit-function theorem is
// applied directly at the converged state u = (ln rho_spec, ln rho_inc_0, ln rho_inc_1):
//     F(u, theta) = (mu_0^S - mu_0^I, mu_1^S - mu_1^I, p^S - p^I) = 0,   p = p^S(u, theta)
//     dp/dtheta = dp^vap/dtheta|_u - w . dF/dtheta|_u,      J^T w = dp^vap/du   (vap = vapour phase)
// J comes from one T2<double> evaluation per phase (as in the solver).  The explicit parameter derivatives enter only
// through one scalar per phase, alpha a + beta . grad_rho a with row constants (alpha, beta) made of w and the
// densities, so each phase needs d/dtheta of a and of ONE directional derivative of a: D1<DN<double,C>> evaluations
// along beta (4 numbers per quantity for C = 1 instead of the 6 of a full gradient), C directions per pass.
#pragma once
#include "mix_adjoint.hpp"
#include "mix_model.hpp"
#include "mix_solver.hpp"

namespace pcs {

constexpr int MIX_DIRS = 19;  // 16 parameters (component 0 then 1), k_ij, eps_AiBj, T
// directions per pass; A/B on 1e6 rows: 1: 13.3 ms, 2: 9.8 ms, 3: 10.4 ms.  The kernel's stack frame (coefficient struct
// with tangents + spills) must stay small: at 2.5-4 KB per lane the runtime throttles the resident waves (measured:
// 21 ms, and bimodal with the launch history); with the coefficient set-up out of line it is 2.2 KB for 2 directions
constexpr int MIX_CHUNK = 2;
// Direction handled by slot j of a pass (-1: none).  With two directions per pass they are paired so that the
// structurally-zero ones share passes (both dipole moments; the association parameters of one component; eps_AiBj alone):
// a non-polar non-associating row runs 4 passes instead of 6, an associating one skips the eps_AiBj pass unless it is used.
PCS_DEV int mix_direction(int pass, int j) {
    if (MIX_CHUNK == 2) {
        // (m0, s0) (e0, m1) (s1, e1) (kij, T) (mu0, mu1) (kap0, eab0) (na0, nb0) (kap1, eab1) (na1, nb1) (epsAB, -)
        switch (2 * pass + j) {
            case 0: return 0;   case 1: return 1;
            case 2: return 2;   case 3: return 8;
            case 4: return 9;   case 5: return 10;
            case 6: return 16;  case 7: return 18;
            case 8: return 3;   case 9: return 11;
            case 10: return 4;  case 11: return 5;
            case 12: return 6;  case 13: return 7;
            case 14: return 12; case 15: return 13;
            case 16: return 14; case 17: return 15;
            case 18: return 17;
            default: return -1;
        }
    }
    const int d = pass * MIX_CHUNK + j;
    return d < MIX_DIRS ? d : -1;
}

struct MixModelD {
    MixCoef<double> c;
    template <class R> PCS_DEV R a(const R& r0, const R& r1) const { return mix_a<double, R>(c, r0, r1); }
    template <class R, class Z> PCS_DEV R a_z(const R& r0, const R& r1, const Z& zeta3) const { return mix_a_z<double, R, Z>(c, r0, r1, zeta3); }
    PCS_DEV double packing(double x0, double x1) const { return x0 * c.zk[3][0] + x1 * c.zk[3][1]; }
};

template <class G, class R>
__device__ __attribute__((noinline)) R mix_a_tangent(const MixCoef<G>& c, const R& r0, const R& r1) {
    return mix_a<G, R>(c, r0, r1);
}

// coefficients with their parameter tangents for direction(s) d0 .. d0 + MIX_CHUNK - 1; out of line so that its
// register spills live in its own stack frame (which the evaluation's frame then reuses) instead of the kernel's
template <class G>
__device__ __attribute__((noinline)) void mix_coef_tangent(MixCoef<G>& c, const double* __restrict__ par, double k0, double k1, double T, int pass) {
    G gp[16], gk0, gk1, gT;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        gp[k].v = par[k];
#pragma unroll
        for (int j = 0; j < MIX_CHUNK; j++) gp[k].e[j] = (mix_direction(pass, j) == k) ? 1.0 : 0.0;
    }
    gk0.v = k0; gk1.v = k1; gT.v = T;
#pragma unroll
    for (int j = 0; j < MIX_CHUNK; j++) {
        const int d = mix_direction(pass, j);
        gk0.e[j] = (d == 16) ? 1.0 : 0.0;
        gk1.e[j] = (d == 17) ? 1.0 : 0.0;
        gT.e[j] = (d == 18) ? 1.0 : 0.0;
    }
    mix_coef<G>(c, gp, gk0, gk1, gT);
}

// spec = (rho_spec_0, rho_spec_1), inc = (rho_inc_0, rho_inc_1); out[19] in Pa per unit of theta
// spec_is_vapor: the pressure functional is always taken on the VAPOUR phase (p^S for dew, p^I for
// bubble): the liquid-phase pressure is a difference of O(0.1) terms, so its explicit parameter
// derivative would have to cancel against w . dF/dtheta to the size of p itself.
constexpr int PCS_MIX_ADJ_CHUNK = 3;  // A/B 1e6 rows: 2: 2.6 ms, 3: 2.5, 4: 2.5, 5: 3.0, 7: 3.8, 10: 5.1 (spills of the pass function)
constexpr int MIX_ADJ_CHUNK = PCS_MIX_ADJ_CHUNK;  // parameter directions per pass over mix_coef in the adjoint form

// one pass of the adjoint form: e[j] = d/dtheta_{d0+j} sum_k adj[k] c_k(theta), the coefficient set with MIX_ADJ_CHUNK tangents
__device__ __attribute__((noinline)) void mix_adjoint_pass(const double* __restrict__ par, double k0, double k1, double T, int d0,
                                                           const double* adj, int adj_stride, double* __restrict__ e) {
    typedef DN<double, MIX_ADJ_CHUNK> GA;
    GA gp[16], gk0, gk1, gT;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        gp[k].v = par[k];
#pragma unroll
        for (int j = 0; j < MIX_ADJ_CHUNK; j++) gp[k].e[j] = (d0 + j == k) ? 1.0 : 0.0;
    }
    gk0.v = k0; gk1.v = k1; gT.v = T;
#pragma unroll
    for (int j = 0; j < MIX_ADJ_CHUNK; j++) {
        gk0.e[j] = (d0 + j == 16) ? 1.0 : 0.0;
        gk1.e[j] = (d0 + j == 17) ? 1.0 : 0.0;
        gT.e[j] = (d0 + j == 18) ? 1.0 : 0.0;
    }
    MixCoef<GA> cg;
    mix_coef<GA>(cg, gp, gk0, gk1, gT);
    const GA S = mix_adjoint_contract(cg, adj, adj_stride);
#pragma unroll
    for (int j = 0; j < MIX_ADJ_CHUNK; j++) e[j] = S.e[j];
}

// Gradient of S(theta) = sum_k adj[k] c_k(theta) w.r.t. the 19 inputs, block by block: every block of mix_coef (mix_model.hpp)
// depends on a few inputs only -- a component's (m, sigma, eps, T); the dispersion aggregates on 8; the dipole polynomials
// on 9; the association strengths on 14 -- so each is differentiated forward with just those seeded (DN<4>, DN<8>, DN<9>,
// DN<14> over a small function) instead of 19 directions through the whole coefficient set (7 DN<3> passes: 19 k
// instructions per row; this form: 1.5 k for a non-polar non-associating row, ~6 k with both).
template <class G> struct MixCompOut { G m[2], mm1[2], d[2], zk[4][2]; };
template <class G> struct MixDispOut { G A[3], B[3]; };
template <class G> struct MixAssocIO { int acls; G na[2], nb[2], d[2], dij[3], S[3]; };
template <int N>
PCS_DEV void seed_inputs(DN<double, N>* x, const double* val) {
#pragma unroll
    for (int k = 0; k < N; k++) {
        x[k].v = val[k];
#pragma unroll
        for (int j = 0; j < N; j++) x[k].e[j] = (j == k) ? 1.0 : 0.0;
    }
}
__device__ __attribute__((noinline)) void mix_coef_gradient(const double* __restrict__ par, double k0, double k1, double T,
                                                            const double* adj, int st, double* __restrict__ grad) {
#define PCS_AD(slot) adj[(slot) * st]
#pragma unroll
    for (int d = 0; d < MIX_DIRS; d++) grad[d] = 0.0;
    // components
#pragma unroll 1
    for (int i = 0; i < 2; i++) {
        typedef DN<double, 4> G;
        const double val[4] = {par[8 * i], par[8 * i + 1], par[8 * i + 2], T};
        G x[4];
        seed_inputs<4>(x, val);
        const G rT = d_recip(x[3]);
        MixCompOut<G> o;
        mix_coef_component(o, i, x[0], x[1], x[2], rT);
        G S = o.m[i] * PCS_AD(ADJ_M + i) + o.mm1[i] * PCS_AD(ADJ_MM1 + i) + o.d[i] * PCS_AD(ADJ_D + i);
#pragma unroll
        for (int k = 0; k < 4; k++) S = S + o.zk[k][i] * PCS_AD(ADJ_ZK + 2 * k + i);
        grad[8 * i] += S.e[0];
        grad[8 * i + 1] += S.e[1];
        grad[8 * i + 2] += S.e[2];
        grad[18] += S.e[3];
    }
    // dispersion aggregates: m0, m1, sigma0, sigma1, eps0, eps1, k_ij, T
    {
        typedef DN<double, 8> G;
        const double val[8] = {par[0], par[8], par[1], par[9], par[2], par[10], k0, T};
        G x[8];
        seed_inputs<8>(x, val);
        const G rT = d_recip(x[7]);
        MixDispOut<G> o;
        mix_coef_dispersion(o, &x[0], &x[2], &x[4], x[6], rT);
        G S = o.A[0] * PCS_AD(ADJ_A) + o.B[0] * PCS_AD(ADJ_B);
#pragma unroll
        for (int q = 1; q < 3; q++) S = S + o.A[q] * PCS_AD(ADJ_A + q) + o.B[q] * PCS_AD(ADJ_B + q);
        const int idx[8] = {0, 8, 1, 9, 2, 10, 16, 18};
#pragma unroll
        for (int k = 0; k < 8; k++) grad[idx[k]] += S.e[k];
    }
    // dipole polynomials: m0, m1, sigma0, sigma1, eps0, eps1, mu0, mu1, T
    if (par[3] != 0.0 || par[11] != 0.0) {
        typedef DN<double, 9> G;
        const double val[9] = {par[0], par[8], par[1], par[9], par[2], par[10], par[3], par[11], T};
        G x[9];
        seed_inputs<9>(x, val);
        const G rT = d_recip(x[8]);
        G mu2t[2] = {mix_mu2t(x[6], x[0], rT), mix_mu2t(x[7], x[1], rT)};
        G pj[3][5], tj[4][4];
        dipole_coefficients<G>(pj, tj, &x[0], &x[2], &x[4], mu2t, rT);
        G S(0.0);
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
#pragma unroll
            for (int k = 0; k < 5; k++) S = S + pj[pr][k] * PCS_AD(ADJ_PJ + 5 * pr + k);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < 4; k++) S = S + tj[t][k] * PCS_AD(ADJ_TJ + 4 * t + k);
        const int idx[9] = {0, 8, 1, 9, 2, 10, 3, 11, 18};
#pragma unroll
        for (int k = 0; k < 9; k++) grad[idx[k]] += S.e[k];
    }
    // association: sigma0, sigma1, eps0, eps1, kappa0, kappa1, eps_ab0, eps_ab1, na0, na1, nb0, nb1, eps_AiBj, T
    const int acls = mix_assoc_class(par[6], par[7], par[14], par[15]);
    if (acls != ASSOC_NONE) {
        typedef DN<double, 14> G;
        const double val[14] = {par[1], par[9], par[2], par[10], par[4], par[12], par[5], par[13], par[6], par[14], par[7], par[15], k1, T};
        G x[14];
        seed_inputs<14>(x, val);
        const G rT = d_recip(x[13]);
        MixAssocIO<G> o;
        o.acls = acls;
        o.na[0] = x[8]; o.na[1] = x[9]; o.nb[0] = x[10]; o.nb[1] = x[11];
        o.d[0] = mix_diameter(x[0], x[2], rT);
        o.d[1] = mix_diameter(x[1], x[3], rT);
        mix_coef_assoc(o, &x[0], &x[4], &x[6], x[12], rT);
        G S = o.na[0] * PCS_AD(ADJ_NA) + o.na[1] * PCS_AD(ADJ_NA + 1) + o.nb[0] * PCS_AD(ADJ_NB) + o.nb[1] * PCS_AD(ADJ_NB + 1);
        const int nq = acls == ASSOC_SELF ? 1 : 3;
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (q < nq) S = S + o.dij[q] * PCS_AD(ADJ_DIJ + q) + o.S[q] * PCS_AD(ADJ_S + q);
        const int idx[14] = {1, 9, 2, 10, 4, 12, 5, 13, 6, 14, 7, 15, 17, 18};
#pragma unroll
        for (int k = 0; k < 14; k++) grad[idx[k]] += S.e[k];
    }
#undef PCS_AD
}

// adj: lane-strided scratch of ADJ_SLOTS doubles (adj[k * adj_stride]), LDS in k_mix_jacobian (unused by the tangent form)
PCS_DEV void mix_jacobian(const double par[16], double k0, double k1, double T, double s0, double s1, double i0,
                          double i1, bool spec_is_vapor, double* __restrict__ g, double* adj, int adj_stride) {
    MixModelD m;
    mix_coef<double>(m.c, par, k0, k1, T);
    PhaseEval s = phase_eval(m, s0, s1);
    PhaseEval n = phase_eval(m, i0, i1);
    const double rs = s0 + s1, z0 = s0 / rs, z1 = s1 / rs;
    // transposed Jacobian of F w.r.t. u, right-hand side dp^S/du = (rs (z0 dp0 + z1 dp1), 0, 0)
    double J[3][3];
    J[0][0] = rs * (z0 * (1.0 / s.r0 + s.h00) + z1 * s.h01);
    J[1][0] = rs * (z0 * s.h01 + z1 * (1.0 / s.r1 + s.h11));
    J[2][0] = rs * (z0 * s.dp0() + z1 * s.dp1());
    J[0][1] = -i0 * (1.0 / i0 + n.h00);
    J[1][1] = -i0 * n.h01;
    J[2][1] = -i0 * n.dp0();
    J[0][2] = -i1 * n.h01;
    J[1][2] = -i1 * (1.0 / i1 + n.h11);
    J[2][2] = -i1 * n.dp1();
    double A[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) A[r][cc] = J[cc][r];  // J^T
        A[r][3] = 0.0;
    }
    if (spec_is_vapor) {
        A[0][3] = J[2][0];  // dp^S/du = (rs (z0 dp0 + z1 dp1), 0, 0)
    } else {
        A[1][3] = -J[2][1];  // dp^I/du = (0, i0 dp0^I, i1 dp1^I)
        A[2][3] = -J[2][2];
    }
    double w[3];
    bool ok = solve3(A, w);
    const double p_red = spec_is_vapor ? s.p() : n.p();
    typedef DN<double, MIX_CHUNK> G;
    typedef D1<G> R;
    // dp/dtheta = sum over the two phases of d/dtheta [alpha a + beta . grad a]:
    //   p^X = -a^X + rho^X . grad a^X,  F_i = grad_i a^S - grad_i a^I,  dp = dp^vap - (w0 F0 + w1 F1 + w2 (p^S - p^I))
    double alpha[2], beta0[2], beta1[2];
    if (spec_is_vapor) {
        const double u = 1.0 - w[2];
        alpha[0] = -u;     beta0[0] = u * s0 - w[0];     beta1[0] = u * s1 - w[1];
        alpha[1] = -w[2];  beta0[1] = w[2] * i0 + w[0];  beta1[1] = w[2] * i1 + w[1];
    } else {
        const double u = 1.0 + w[2];
        alpha[0] = w[2];   beta0[0] = -w[2] * s0 - w[0]; beta1[0] = -w[2] * s1 - w[1];
        alpha[1] = -u;     beta0[1] = u * i0 + w[0];     beta1[1] = u * i1 + w[1];
    }
    {
        // dp/dtheta = T kB/A^3 sum_k abar_k dc_k/dtheta (+ p/T for theta = T) with the coefficient adjoints
        // abar = sum over the phases of d/dc [alpha a + beta . grad a], closed form (mix_adjoint.hpp)
#pragma unroll
        for (int k = 0; k < ADJ_SLOTS; k++) adj[k * adj_stride] = 0.0;
#pragma unroll 1
        for (int ph = 0; ph < 2; ph++) {
            const double q0 = ph == 0 ? s0 : i0, q1 = ph == 0 ? s1 : i1;
            const double al = ph == 0 ? alpha[0] : alpha[1], b0 = ph == 0 ? beta0[0] : beta0[1], b1 = ph == 0 ? beta1[0] : beta1[1];
            mix_a_adjoint(m.c, q0, q1, b0, b1, al, adj, adj_stride);
        }
        double e[MIX_DIRS];
        mix_coef_gradient(par, k0, k1, T, adj, adj_stride, e);
#pragma unroll
        for (int d = 0; d < MIX_DIRS; d++) {
            double val = e[d] * T * P_UNIT;
            if (d == 18) val += p_red * P_UNIT;  // p [Pa] = p_red T kB/A^3
            if (!ok) val = __longlong_as_double(0x7ff8000000000000LL);
            g[d] = val;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Vector-Jacobian product of PcSaftMix.derivatives / GcPcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420):
//   a,  p = r0 + r1 - a + r0 a_0 + r1 a_1,  mu_i = a_i,  v_i = d_i / (r0 d0 + r1 d1),  d_i = 1 + r0 a_i0 + r1 a_i1
// (a_i, a_ij: density derivatives of a).  In the reference these are torch graphs, so a loss built on them reaches
// parameters, temperature and the densities.  With upstream gradients (ga, gp, gmu[2], gv[2]) the loss is, to first
// order, linear in the six Taylor coefficients of a: dL = sum_k c_k d(coef_k); DerivWeights holds the c_k, the explicit
// density terms and v.  Any forward-mode evaluation R = T2<DN> then gives dL/dtheta = sum_k c_k coef_k.e.
// ------------------------------------------------------------------------------------------------------------------
struct DerivWeights {
    double cv, cg0, cg1, ch00, ch01, ch11;  // dL/d(a, a_0, a_1, a_00, a_01, a_11)
    double x0, x1;                          // explicit dL/dr0, dL/dr1 at fixed Taylor coefficients
};
PCS_DEV DerivWeights deriv_weights(const PhaseEval& e, double ga, double gp, double gm0, double gm1, double gv0, double gv1) {
    DerivWeights w;
    const double d0 = e.dp0(), d1 = e.dp1();
    const double rD = 1.0 / (e.r0 * d0 + e.r1 * d1);
    const double Sv = (gv0 * d0 + gv1 * d1) * rD;  // sum gv_i v_i
    const double e0 = (gv0 - Sv * e.r0) * rD, e1 = (gv1 - Sv * e.r1) * rD;  // dL/dd_j
    w.cv = ga - gp;
    w.cg0 = gp * e.r0 + gm0;
    w.cg1 = gp * e.r1 + gm1;
    w.ch00 = e0 * e.r0;
    w.ch01 = e0 * e.r1 + e1 * e.r0;
    w.ch11 = e1 * e.r1;
    w.x0 = gp * (1.0 + e.g0) + e0 * e.h00 + e1 * e.h01 - Sv * d0 * rD;
    w.x1 = gp * (1.0 + e.g1) + e0 * e.h01 + e1 * e.h11 - Sv * d1 * rD;
    return w;
}
template <class G>
PCS_DEV double deriv_contract(const DerivWeights& w, const T2<G>& a, int j) {
    return w.cv * a.v.e[j] + w.cg0 * a.g0.e[j] + w.cg1 * a.g1.e[j] + w.ch00 * a.h00.e[j] + w.ch01 * a.h01.e[j] + w.ch11 * a.h11.e[j];
}

constexpr int MIX_VJP_DIRS = 21;  // 16 parameters, k_ij, eps_AiBj, T, rho_0, rho_1

// coefficients with the tangent of ONE direction d (0..18 as MIX_DIRS); out of line like mix_coef_tangent
template <class G>
__device__ __attribute__((noinline)) void mix_coef_direction(MixCoef<G>& c, const double* __restrict__ par, double k0, double k1, double T, int d) {
    G gp[16], gk0, gk1, gT;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        gp[k].v = par[k];
        gp[k].e[0] = (d == k) ? 1.0 : 0.0;
    }
    gk0.v = k0; gk1.v = k1; gT.v = T;
    gk0.e[0] = (d == 16) ? 1.0 : 0.0;
    gk1.e[0] = (d == 17) ? 1.0 : 0.0;
    gT.e[0] = (d == 18) ? 1.0 : 0.0;
    mix_coef<G>(c, gp, gk0, gk1, gT);
}

// out[21] = dL/d(params[0,0..7], params[1,0..7], kij0, kij1, T, rho_0, rho_1)
PCS_DEV void mix_derivatives_vjp(const double par[16], double k0, double k1, double T, double r0, double r1, double ga,
                                 double gp, double gm0, double gm1, double gv0, double gv1, double* __restrict__ g) {
    typedef DN<double, 1> G;
    typedef T2<G> R;
    MixModelD m;
    mix_coef<double>(m.c, par, k0, k1, T);
    const PhaseEval e = phase_eval(m, r0, r1);
    const DerivWeights w = deriv_weights(e, ga, gp, gm0, gm1, gv0, gv1);
    const bool no_assoc = m.c.acls == ASSOC_NONE;
    const bool eab_used = m.c.acls == ASSOC_CROSS && k1 != 0.0;
#pragma unroll 1
    for (int d = 0; d < MIX_VJP_DIRS; d++) {
        bool zero = false;  // structurally-zero directions (see mix_jacobian)
        const int kk = d & 7;
        if (d < 16) zero = (kk == 3 && par[d] == 0.0) || (kk >= 4 && no_assoc);
        else if (d == 17) zero = !eab_used;
        if (__ballot(!zero) == 0ull) {
            g[d] = 0.0;
            continue;
        }
        MixCoef<G> c;
        mix_coef_direction<G>(c, par, k0, k1, T, d < 19 ? d : -1);
        G q0(r0), q1(r1);
        if (d == 19) q0.e[0] = 1.0;
        if (d == 20) q1.e[0] = 1.0;
        const G one(1.0), nul(0.0);
        R a = mix_a_tangent<G, R>(c, R(q0, one, nul, nul, nul, nul), R(q1, nul, one, nul, nul, nul));
        double val = deriv_contract(w, a, 0);
        if (d == 19) val += w.x0;
        if (d == 20) val += w.x1;
        g[d] = zero ? 0.0 : val;
    }
}

}  // namespace pcs
