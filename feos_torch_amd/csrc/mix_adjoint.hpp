// Coefficient adjoints of the binary-mixture Helmholtz energy (device only): for a phase at partial densities q and a
// direction b in density space,
//     out[k] += alpha da/dc_k (q) + d/ds [da/dc_k (q + s b)]_{s=0}          for every field c_k of MixCoef
// i.e. the derivative of  alpha a + b . grad_rho a  (the scalar the bubble / dew pressure gradient needs from each phase,
// mix_jacobian.hpp) with respect to the coefficient set, in closed form and in D1s arithmetic along b -- instead of pushing
// parameter tangents through the whole evaluation once per direction.  The terms are those of core_terms / mix_a
// (mix_model.hpp; feos_torch/pcsaft_mix.py:31-154):
//   hard sphere and chain through the packing sums zeta_k = zk[k][0] r0 + zk[k][1] r1:  d/dzk[k][i] = (da/dzeta_k) r_i
//   dispersion: linear in A[], B[]; through m_bar = (r0 m0 + r1 m1)/(r0 + r1) in m[]
//   dipoles: a = phi2^2/(phi2 - phi3), phi2 / phi3 linear in pj / tj
//   association (self, cross): the association energy is stationary in the site fractions at the mass-action solution, so
//   only the explicit dependences count:  d/d(site density) = ln X,  d/dDelta_ij = -rho_Ai rho_Bj X_Ai X_Bj.
// Induced association (the reference solves ONE A-site fraction for both components from a weighted residual,
// pcsaft_mix.py:341-375, which is not a stationary point of the energy): that block alone is differentiated forward in its
// seven inputs.
#pragma once
#include "mix_model.hpp"

namespace pcs {

// slot of each MixCoef field in the adjoint array
enum : int {
    ADJ_M = 0, ADJ_MM1 = 2, ADJ_D = 4, ADJ_ZK = 6 /* [k][i] -> 6 + 2k + i */, ADJ_A = 14, ADJ_B = 17, ADJ_PJ = 20 /* [pr][k] -> 20 + 5pr + k */,
    ADJ_TJ = 35 /* [t][k] -> 35 + 4t + k */, ADJ_NA = 51, ADJ_NB = 53, ADJ_DIJ = 55, ADJ_S = 58, ADJ_SLOTS = 61
};

// out: lane-strided array (out[k * stride])
// Induced association block of mix_a differentiated forward in its seven inputs (na0, na1, nb0, nb1, Delta_00, Delta_01,
// Delta_11), first-order tangents over D1s: dn[4] = d a_assoc / d(na0, na1, nb0, nb1), dD[3] = d a_assoc / d Delta_q.  Out of
// line: its DN<D1s,7> arithmetic needs the register file to itself.
__device__ __attribute__((noinline)) void induced_assoc_adjoint(double cna0, double cna1, double cnb0, double cnb1, D1s r0, D1s r1, D1s D0,
                                                                D1s D1_, D1s D2, D1s* dn, D1s* dD) {
    typedef D1s R;
    const R D[3] = {D0, D1_, D2};
    typedef DN<R, 7> L;
    L in[7], rl0(0.0), rl1(0.0);
    rl0.v = r0;
    rl1.v = r1;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        in[k] = L(0.0);
        in[k].v = k == 0 ? R(cna0) : k == 1 ? R(cna1) : k == 2 ? R(cnb0) : k == 3 ? R(cnb1) : D[k - 4];
        in[k].e[k] = R(1.0);
    }
    const L &na0 = in[0], &na1 = in[1], &nb0 = in[2], &nb1 = in[3];
    const L d00 = in[4] * rl0, d01 = in[5] * rl1, d10 = in[5] * rl0, d11 = in[6] * rl1;
    // real part exactly as mix_a: bracketed Newton from 0.2
    const double n0 = cna0, n1 = cna1, m0 = cnb0, m1n = cnb1;
    const double e00 = re(d00), e01 = re(d01), e10 = re(d10), e11 = re(d11);
    double x = 0.2, lo = 0.0, hi = 2.0;
    for (int it = 0; it < 200; it++) {
        double f, st;
        induced_step<double>(x, n0, n1, m0, m1n, e00, e01, e10, e11, f, st);
        if (f == 0.0) break;
        if (f < 0.0) lo = x; else hi = x;
        double nx = x - st;
        if (!(nx >= lo && nx <= hi && nx > 0.0)) nx = 0.5 * (lo + hi);
        bool conv = fabs(nx - x) <= 1e-12 * x;
        x = nx;
        if (conv) break;
    }
    L xa(x);
    induced_refine(xa, na0, na1, nb0, nb1, d00, d01, d10, d11);
    const L xb0 = d_recip(1.0 + xa * (na0 * d00 + na1 * d01));
    const L xb1 = d_recip(1.0 + xa * (na0 * d10 + na1 * d11));
    const L sa = site_term(xa);
    const L aas = rl0 * (sa * na0 + site_term(xb0) * nb0) + rl1 * (sa * na1 + site_term(xb1) * nb1);
    dn[0] = aas.e[0];
    dn[1] = aas.e[1];
    dn[2] = aas.e[2];
    dn[3] = aas.e[3];
    dD[0] = aas.e[4];
    dD[1] = aas.e[5];
    dD[2] = aas.e[6];
}

// state shared by the parts of an adjoint evaluation: the density point along its direction, the packing quantities and
// the running derivatives of a with respect to the packing sums
struct AdjCtx {
    D1s r0, r1, r00, r01, r11, zeta2, zeta3, omz, z3m1, z3m2, dz0, dz1, dz2, dz3;
};
#define PCS_ADJ(slot, expr)                          \
    {                                                \
        const R g_ = (expr);                         \
        out[(slot) * stride] += alpha * g_.v + g_.d1; \
    }
// packing sums, hard sphere, dispersion, dipoles: any coefficient struct with m, zk, A, B, polar, pj, tj (MixCoef, GcCoef)
template <class C>
PCS_DEV void adjoint_core(const C& c, double q0, double q1, double b0, double b1, double alpha, double* out, int stride, AdjCtx& x) {
    typedef D1s R;
    const R r0(q0, b0), r1(q1, b1);
    const R zeta0 = r0 * c.zk[0][0] + r1 * c.zk[0][1];
    const R zeta1 = r0 * c.zk[1][0] + r1 * c.zk[1][1];
    const R zeta2 = r0 * c.zk[2][0] + r1 * c.zk[2][1];
    const R zeta3 = r0 * c.zk[3][0] + r1 * c.zk[3][1];
    const R omz = 1.0 - zeta3;
    const R z3m1 = d_recip(omz), z3m2 = z3m1 * z3m1;
    const R rz3 = d_recip(zeta3), rz3sq = rz3 * rz3;
    const R zeta23 = zeta2 * rz3;
    const R l13 = d_log(omz);
    // hard sphere (:56-60): a = K [3 z1 z2/(1-z3) + z2^3/(z3 (1-z3)^2) + (z2^3/z3^2 - z0) ln(1-z3)]
    const double K = 6.0 / PI;
    const R t3 = (zeta2 * zeta2) * zeta2;
    R dz0 = (-K) * l13;
    R dz1 = (3.0 * K) * (zeta2 * z3m1);
    R dz2 = (3.0 * K) * (zeta1 * z3m1 + (zeta2 * zeta23) * z3m2 + (zeta23 * zeta23) * l13);
    R dz3 = K * (3.0 * ((zeta1 * zeta2) * z3m2) + t3 * (2.0 * ((z3m1 * z3m2) * rz3) - z3m2 * rz3sq) - 2.0 * ((t3 * (rz3sq * rz3)) * l13) -
                 (t3 * rz3sq - zeta0) * z3m1);
    // dispersion (:69-106): a = -pi [2 rho1mix I1 + rho2mix C1 I2 m_bar]
    const R r00 = r0 * r0, r01 = r0 * r1, r11 = r1 * r1;
    {
        const R rrs = d_recip(r0 + r1);
        const R mbar = (r0 * c.m[0] + r1 * c.m[1]) * rrs;
        const R rmb = d_recip(mbar), rmb2 = rmb * rmb;
        const R m1 = (mbar - 1.0) * rmb;
        const R m2 = m1 * ((mbar - 2.0) * rmb);
        const R dm1 = rmb2, dm2 = 3.0 * rmb2 - 4.0 * (rmb2 * rmb);  // d m1 / d m_bar, d m2 / d m_bar
        R P0, P0d, P1, P1d, P2, P2d;
        poly_and_derivative<7>(A0, zeta3, P0, P0d);
        poly_and_derivative<7>(A1, zeta3, P1, P1d);
        poly_and_derivative<7>(A2, zeta3, P2, P2d);
        const R I1 = P0 + m1 * P1 + m2 * P2, I1d = P0d + m1 * P1d + m2 * P2d, I1m = P1 * dm1 + P2 * dm2;
        poly_and_derivative<7>(B0, zeta3, P0, P0d);
        poly_and_derivative<7>(B1, zeta3, P1, P1d);
        poly_and_derivative<7>(B2, zeta3, P2, P2d);
        const R I2 = P0 + m1 * P1 + m2 * P2, I2d = P0d + m1 * P1d + m2 * P2d, I2m = P1 * dm1 + P2 * dm2;
        const R z3m4 = z3m2 * z3m2;
        const R w = d_recip(2.0 - zeta3), t2 = z3m1 * w, qq = t2 * t2;
        const R poly = zeta3 * (20.0 + zeta3 * (-27.0 + zeta3 * (12.0 - 2.0 * zeta3)));
        const R poly1 = 20.0 + zeta3 * (-54.0 + zeta3 * (36.0 - 8.0 * zeta3));
        const R Ca = (zeta3 * (8.0 - 2.0 * zeta3)) * z3m4, Ca1 = (8.0 + zeta3 * (20.0 - 4.0 * zeta3)) * (z3m4 * z3m1);
        const R Cb = poly * qq, Cb1 = qq * (poly1 + 2.0 * (poly * (z3m1 + w)));
        const R C1 = d_recip(1.0 + mbar * Ca + (1.0 - mbar) * Cb), C1sq = C1 * C1;
        const R C1m = -(C1sq * (Ca - Cb)), C1d = -(C1sq * (mbar * Ca1 + (1.0 - mbar) * Cb1));
        const R rho1 = r00 * c.A[0] + r01 * c.A[1] + r11 * c.A[2];
        const R rho2 = r00 * c.B[0] + r01 * c.B[1] + r11 * c.B[2];
        const R fA = (-2.0 * PI) * I1, fB = (-PI) * ((C1 * I2) * mbar);
        PCS_ADJ(ADJ_A + 0, fA * r00);
        PCS_ADJ(ADJ_A + 1, fA * r01);
        PCS_ADJ(ADJ_A + 2, fA * r11);
        PCS_ADJ(ADJ_B + 0, fB * r00);
        PCS_ADJ(ADJ_B + 1, fB * r01);
        PCS_ADJ(ADJ_B + 2, fB * r11);
        const R dmb = (-PI) * (2.0 * (rho1 * I1m) + rho2 * (((C1m * I2) + (C1 * I2m)) * mbar + C1 * I2));
        PCS_ADJ(ADJ_M + 0, dmb * (r0 * rrs));
        PCS_ADJ(ADJ_M + 1, dmb * (r1 * rrs));
        dz3 = dz3 + (-PI) * (2.0 * (rho1 * I1d) + (rho2 * mbar) * (C1d * I2 + C1 * I2d));
    }
    // dipoles (:156-208): a = phi2^2/(phi2 - phi3)
    if (c.polar) {
        const R rr[3] = {r00, r01, r11};
        const R rrr[4] = {r00 * r0, r00 * r1, r0 * r11, r11 * r1};
        R phi2(0.0), phi3(0.0), phi2d(0.0), phi3d(0.0);
#pragma unroll
        for (int pr = 0; pr < 3; pr++) {
            R P, Pd;
            poly_and_derivative<5>(c.pj[pr], zeta3, P, Pd);
            phi2 = phi2 + rr[pr] * P;
            phi2d = phi2d + rr[pr] * Pd;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            R P, Pd;
            poly_and_derivative<4>(c.tj[t], zeta3, P, Pd);
            phi3 = phi3 + rrr[t] * P;
            phi3d = phi3d + rrr[t] * Pd;
        }
        R f2(1.0), f3(0.0);  // da/dphi2, da/dphi3 (phi2 = phi3 = 0: a = phi2, see core_terms)
        if (!(fabs(re(phi2)) < 1e-90)) {  // (trace polar component: the limit a = phi2, as core_terms)
            const R rd = d_recip(phi2 - phi3), rd2 = rd * rd;
            f2 = (phi2 * (phi2 - 2.0 * phi3)) * rd2;
            f3 = (phi2 * phi2) * rd2;
        }
#pragma unroll
        for (int pr = 0; pr < 3; pr++) {
            R pw = f2 * rr[pr];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                PCS_ADJ(ADJ_PJ + 5 * pr + k, pw);
                pw = pw * zeta3;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            R pw = f3 * rrr[t];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                PCS_ADJ(ADJ_TJ + 4 * t + k, pw);
                pw = pw * zeta3;
            }
        }
        dz3 = dz3 + f2 * phi2d + f3 * phi3d;
    }
    x.r0 = r0; x.r1 = r1; x.r00 = r00; x.r01 = r01; x.r11 = r11;
    x.zeta2 = zeta2; x.zeta3 = zeta3; x.omz = omz; x.z3m1 = z3m1; x.z3m2 = z3m2;
    x.dz0 = dz0; x.dz1 = dz1; x.dz2 = dz2; x.dz3 = dz3;
}

// association strengths Delta_q = S_q (1 + k_q (2 k_q + 3))/(1 - z3), k_q = dij_q z2/(1 - z3), q < nq
struct AdjDelta {
    D1s kq[3], pk[3], D[3];
};
template <class C>
PCS_DEV void adjoint_delta(const C& c, const AdjCtx& x, int nq, AdjDelta& dl) {
    typedef D1s R;
    const R zz = x.zeta2 * x.z3m1;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        if (q < nq) {
            dl.kq[q] = zz * c.dij[q];
            dl.pk[q] = dl.kq[q] * (2.0 * dl.kq[q] + 3.0) + 1.0;
            dl.D[q] = (x.z3m1 * dl.pk[q]) * c.S[q];
        }
    }
}
// from dD[q] = da/dDelta_q to the adjoints of S, dij and the packing sums
template <class C>
PCS_DEV void adjoint_delta_chain(const C& c, AdjCtx& x, int nq, const AdjDelta& dl, const D1s* dD, double alpha, double* out, int stride) {
    typedef D1s R;
    const R zz = x.zeta2 * x.z3m1;
    R dzz(0.0);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        if (q < nq) {
            PCS_ADJ(ADJ_S + q, dD[q] * (x.z3m1 * dl.pk[q]));
            const R dS = dD[q] * c.S[q];
            const R dk = dS * (x.z3m1 * (4.0 * dl.kq[q] + 3.0));
            PCS_ADJ(ADJ_DIJ + q, dk * zz);
            dzz = dzz + dk * c.dij[q];
            x.dz3 = x.dz3 + dS * (x.z3m2 * dl.pk[q]);
        }
    }
    x.dz2 = x.dz2 + dzz * x.z3m1;
    x.dz3 = x.dz3 + dzz * (x.zeta2 * x.z3m2);
}
// d/dzk[k][i] = (da/dzeta_k) r_i
PCS_DEV void adjoint_flush(const AdjCtx& x, double alpha, double* out, int stride) {
    typedef D1s R;
    PCS_ADJ(ADJ_ZK + 0, x.dz0 * x.r0);
    PCS_ADJ(ADJ_ZK + 1, x.dz0 * x.r1);
    PCS_ADJ(ADJ_ZK + 2, x.dz1 * x.r0);
    PCS_ADJ(ADJ_ZK + 3, x.dz1 * x.r1);
    PCS_ADJ(ADJ_ZK + 4, x.dz2 * x.r0);
    PCS_ADJ(ADJ_ZK + 5, x.dz2 * x.r1);
    PCS_ADJ(ADJ_ZK + 6, x.dz3 * x.r0);
    PCS_ADJ(ADJ_ZK + 7, x.dz3 * x.r1);
}

// (out of line: the evaluation wants the whole register file to itself, like the solvers' phase_eval)
__device__ __attribute__((noinline)) void mix_a_adjoint(const MixCoef<double>& c, double q0, double q1, double b0, double b1, double alpha,
                                                        double* out, int stride) {
    typedef D1s R;
    AdjCtx x;
    adjoint_core(c, q0, q1, b0, b1, alpha, out, stride, x);
    const R &r0 = x.r0, &r1 = x.r1, &zeta2 = x.zeta2, &omz = x.omz, &z3m1 = x.z3m1, &z3m2 = x.z3m2;
    R &dz2 = x.dz2, &dz3 = x.dz3;
    // hard chain (:63-65): a = -sum r_i mm1_i ln g_i,  g_i = 1/(1-z3) + 1.5 d_i cc + 0.5 d_i^2 cc^2 (1-z3),  cc = z2/(1-z3)^2
    const R cc = zeta2 * z3m2;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const R& ri = i == 0 ? r0 : r1;
        const R cd = cc * c.d[i];
        const R g = z3m1 + 1.5 * cd + 0.5 * ((cd * cd) * omz);
        const R pre = (ri * c.mm1[i]) * d_recip(g);
        PCS_ADJ(ADJ_MM1 + i, -(ri * d_log(g)));
        PCS_ADJ(ADJ_D + i, -(pre * (1.5 * cc + (cd * cc) * omz)));
        dz2 = dz2 - pre * ((1.5 * c.d[i] + c.d[i] * (cd * omz)) * z3m2);
        dz3 = dz3 - pre * (z3m2 + 3.0 * (cd * z3m1) + 1.5 * (cd * cd));
    }
    // association (:118-152)
    if (c.acls != ASSOC_NONE) {
        const int nq = c.acls == ASSOC_SELF ? 1 : 3;
        AdjDelta dl;
        adjoint_delta(c, x, nq, dl);
        const R* D = dl.D;
        R dD[3];
        if (c.acls == ASSOC_SELF) {
            const R rhoa = r0 * c.na[0] + r1 * c.na[1], rhob = r0 * c.nb[0] + r1 * c.nb[1];
            const R sa = rhoa * D[0], sb = rhob * D[0];
            const R t = sb - sa, aux = 1.0 - t;
            const R sq = d_sqrt(aux * aux + 4.0 * sb);
            R xa, xb;  // cancellation-free site fractions, see pure_model.hpp
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
            PCS_ADJ(ADJ_NA + 0, r0 * la);
            PCS_ADJ(ADJ_NA + 1, r1 * la);
            PCS_ADJ(ADJ_NB + 0, r0 * lb);
            PCS_ADJ(ADJ_NB + 1, r1 * lb);
            dD[0] = -((rhoa * rhob) * (xa * xb));
        } else if (c.acls == ASSOC_INDUCED) {
            // The reference solves ONE A-site fraction for both components from the weighted residual na0 f0 + na1 f1
            // (:341-375): not a stationary point of the energy, so the block is differentiated forward in its seven inputs
            // (na, nb, Delta_00, Delta_01, Delta_11) -- first-order tangents over the D1s arithmetic of this function
            R dn[4];
            induced_assoc_adjoint(c.na[0], c.na[1], c.nb[0], c.nb[1], r0, r1, D[0], D[1], D[2], dn, dD);
            PCS_ADJ(ADJ_NA + 0, dn[0]);
            PCS_ADJ(ADJ_NA + 1, dn[1]);
            PCS_ADJ(ADJ_NB + 0, dn[2]);
            PCS_ADJ(ADJ_NB + 1, dn[3]);
        } else {
            const R A0_ = r0 * c.na[0], A1_ = r1 * c.na[1], B0_ = r0 * c.nb[0], B1_ = r1 * c.nb[1];
            // real parts exactly as mix_a: Newton from 0.2 (:270) with the ln X / successive-substitution fallbacks
            const double a0 = re(A0_), a1 = re(A1_), bb0 = re(B0_), bb1 = re(B1_), e00 = re(D[0]), e01 = re(D[1]), e11 = re(D[2]);
            double x0 = 0.2, x1 = 0.2;
            for (int it = 0; it < 200; it++) {
                double s0, s1;
                cross_step<double>(x0, x1, a0, a1, bb0, bb1, e00, e01, e11, s0, s1);
                double n0 = x0 - s0, n1 = x1 - s1;
                if (!(n0 > 0.0 && n0 <= 1.5 && n1 > 0.0 && n1 <= 1.5)) {
                    if (it < 60 && is_finite_bits(s0) && is_finite_bits(s1)) {
                        n0 = fmin(x0 * exp(fmin(fmax(-s0 / x0, -3.0), 3.0)), 1.0);
                        n1 = fmin(x1 * exp(fmin(fmax(-s1 / x1, -3.0), 3.0)), 1.0);
                    } else {
                        double u0 = 1.0 / (1.0 + x0 * a0 * e00 + x1 * a1 * e01);
                        double u1 = 1.0 / (1.0 + x0 * a0 * e01 + x1 * a1 * e11);
                        n0 = 1.0 / (1.0 + u0 * bb0 * e00 + u1 * bb1 * e01);
                        n1 = 1.0 / (1.0 + u0 * bb0 * e01 + u1 * bb1 * e11);
                    }
                }
                bool conv = fabs(n0 - x0) <= 1e-12 * x0 && fabs(n1 - x1) <= 1e-12 * x1;
                x0 = n0;
                x1 = n1;
                if (conv) break;
            }
            R xa0(x0), xa1(x1);
            cross_refine(xa0, xa1, A0_, A1_, B0_, B1_, D[0], D[1], D[2]);
            const R xb0 = d_recip(1.0 + xa0 * (A0_ * D[0]) + xa1 * (A1_ * D[1]));
            const R xb1 = d_recip(1.0 + xa0 * (A0_ * D[1]) + xa1 * (A1_ * D[2]));
            PCS_ADJ(ADJ_NA + 0, r0 * d_log(xa0));
            PCS_ADJ(ADJ_NA + 1, r1 * d_log(xa1));
            PCS_ADJ(ADJ_NB + 0, r0 * d_log(xb0));
            PCS_ADJ(ADJ_NB + 1, r1 * d_log(xb1));
            dD[0] = -((A0_ * B0_) * (xa0 * xb0));
            dD[1] = -((A0_ * B1_) * (xa0 * xb1) + (A1_ * B0_) * (xa1 * xb0));
            dD[2] = -((A1_ * B1_) * (xa1 * xb1));
        }
        adjoint_delta_chain(c, x, nq, dl, dD, alpha, out, stride);
    }
    adjoint_flush(x, alpha, out, stride);
}

// S = sum_k adj[k] c_k for a coefficient set with tangents: its tangent part is the parameter derivative
template <class G, class Ptr>
PCS_DEV G mix_adjoint_contract(const MixCoef<G>& c, Ptr adj, int stride) {
#define PCS_AD(slot) adj[(slot) * stride]
    G S = c.m[0] * PCS_AD(ADJ_M) + c.m[1] * PCS_AD(ADJ_M + 1) + c.mm1[0] * PCS_AD(ADJ_MM1) + c.mm1[1] * PCS_AD(ADJ_MM1 + 1) +
          c.d[0] * PCS_AD(ADJ_D) + c.d[1] * PCS_AD(ADJ_D + 1);
#pragma unroll
    for (int k = 0; k < 4; k++) S = S + c.zk[k][0] * PCS_AD(ADJ_ZK + 2 * k) + c.zk[k][1] * PCS_AD(ADJ_ZK + 2 * k + 1);
#pragma unroll
    for (int q = 0; q < 3; q++) S = S + c.A[q] * PCS_AD(ADJ_A + q) + c.B[q] * PCS_AD(ADJ_B + q);
    if (c.polar) {
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
#pragma unroll
            for (int k = 0; k < 5; k++) S = S + c.pj[pr][k] * PCS_AD(ADJ_PJ + 5 * pr + k);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < 4; k++) S = S + c.tj[t][k] * PCS_AD(ADJ_TJ + 4 * t + k);
    }
    if (c.acls != ASSOC_NONE) {
        S = S + c.na[0] * PCS_AD(ADJ_NA) + c.na[1] * PCS_AD(ADJ_NA + 1) + c.nb[0] * PCS_AD(ADJ_NB) + c.nb[1] * PCS_AD(ADJ_NB + 1);
        const int nq = c.acls == ASSOC_SELF ? 1 : 3;
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (q < nq) S = S + c.dij[q] * PCS_AD(ADJ_DIJ + q) + c.S[q] * PCS_AD(ADJ_S + q);
    }
#undef PCS_AD
    return S;
}

#undef PCS_ADJ

}  // namespace pcs
