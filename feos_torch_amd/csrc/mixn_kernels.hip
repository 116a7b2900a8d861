// n-component PC-SAFT mixtures on gfx950: PcSaftMix.derivatives / helmholtz_energy_density for parameters
// [n, nc, 8] without k_ij -- the parts of the reference's model that are written for any number of components
// (feos_torch/pcsaft_mix.py:31-154: hard sphere :56-60, hard chain :63-65, dispersion :69-106, dipoles :156-208, self
// association of ONE associating component :210-239; two associating components are binary-only there, :250 / :336, and stay
// with pcs_mix_derivatives).  One state point per lane, one evaluation per row: no coefficient hoisting, plain loops over
// the components, everything in registers / per-lane stack.
//
// Derivative type: HV<NC+1> = the reference's DualTensor for one row (dual_torch.py:4-158) expressed in densities instead of
// (N, V): first-order directions e_1 .. e_nc and r = (rho_1 .. rho_nc), crossed with the first-order direction r:
//   re = a,  e1[i] = a_i,  e1[nc] = e2 = r . grad a,  e12[i] = sum_j rho_j a_ij,  e12[nc] = r . H . r
// from which (pcsaft_mix.py:395-420)  p = sum rho - a + e2,  mu_i = e1[i],  v_i = (1 + e12[i]) / (sum rho + e12[nc]).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"
#include "dual.hpp"
#include "pcsaft_consts.hpp"

using namespace pcs;
using namespace pcs_abi;

namespace {

#define HVD __device__ __forceinline__
// scalar helpers: S = double in the forward kernel, DN<double, K> (value + K parameter / temperature / density tangents,
// dual.hpp) in the backward kernel; the model below is written once over S
HVD double s_recip(double x) { return 1.0 / x; }
HVD double s_log(double x) { return log(x); }
HVD double s_sqrt(double x) { return sqrt(x); }
HVD double s_exp(double x) { return exp(x); }
HVD double s_cbrt(double x) { return cbrt(x); }
template <int K> HVD DN<double, K> s_recip(const DN<double, K>& x) { return d_recip(x); }
template <int K> HVD DN<double, K> s_log(const DN<double, K>& x) { return d_log(x); }
template <int K> HVD DN<double, K> s_sqrt(const DN<double, K>& x) { return d_sqrt(x); }
template <int K> HVD DN<double, K> s_exp(const DN<double, K>& x) { return d_exp(x); }
template <int K> HVD DN<double, K> s_cbrt(const DN<double, K>& x) { return d_cbrt(x); }
template <class S> HVD S s_min2(const S& m) { return re(m) > 2.0 ? S(2.0) : m; }  // min(m, 2) (:169)

template <int M, class S>
struct HV {
    S re, e2;
    S e1[M], e12[M];
    HVD HV() {}
    HVD explicit HV(const S& x) : re(x), e2(0.0) {
#pragma unroll
        for (int i = 0; i < M; i++) { e1[i] = S(0.0); e12[i] = S(0.0); }
    }
    // f(u): f0, f' = f1, f'' = f2   (dual_torch.py:109-117)
    HVD HV chain(const S& f0, const S& f1, const S& f2) const {
        HV r;
        r.re = f0;
        r.e2 = f1 * e2;
#pragma unroll
        for (int i = 0; i < M; i++) {
            r.e1[i] = f1 * e1[i];
            r.e12[i] = f1 * e12[i] + f2 * (e1[i] * e2);
        }
        return r;
    }
};
template <int M, class S> HVD HV<M, S> operator+(const HV<M, S>& a, const HV<M, S>& b) {
    HV<M, S> r; r.re = a.re + b.re; r.e2 = a.e2 + b.e2;
#pragma unroll
    for (int i = 0; i < M; i++) { r.e1[i] = a.e1[i] + b.e1[i]; r.e12[i] = a.e12[i] + b.e12[i]; }
    return r;
}
template <int M, class S> HVD HV<M, S> operator-(const HV<M, S>& a, const HV<M, S>& b) {
    HV<M, S> r; r.re = a.re - b.re; r.e2 = a.e2 - b.e2;
#pragma unroll
    for (int i = 0; i < M; i++) { r.e1[i] = a.e1[i] - b.e1[i]; r.e12[i] = a.e12[i] - b.e12[i]; }
    return r;
}
template <int M, class S> HVD HV<M, S> operator*(const HV<M, S>& a, const HV<M, S>& b) {  // dual_torch.py:80-107
    HV<M, S> r; r.re = a.re * b.re; r.e2 = a.re * b.e2 + b.re * a.e2;
#pragma unroll
    for (int i = 0; i < M; i++) {
        r.e1[i] = a.re * b.e1[i] + b.re * a.e1[i];
        r.e12[i] = a.re * b.e12[i] + a.e1[i] * b.e2 + a.e2 * b.e1[i] + a.e12[i] * b.re;
    }
    return r;
}
// scalar (S or double) on either side
template <int M, class S, class B> HVD HV<M, S> hv_scale(const HV<M, S>& a, const B& b) {
    HV<M, S> r; r.re = a.re * b; r.e2 = a.e2 * b;
#pragma unroll
    for (int i = 0; i < M; i++) { r.e1[i] = a.e1[i] * b; r.e12[i] = a.e12[i] * b; }
    return r;
}
template <int M, class S> HVD HV<M, S> operator*(const HV<M, S>& a, double b) { return hv_scale(a, b); }
template <int M, class S> HVD HV<M, S> operator*(double b, const HV<M, S>& a) { return hv_scale(a, b); }
template <int M, class S, PCS_IFDUAL(S)> HVD HV<M, S> operator*(const HV<M, S>& a, const S& b) { return hv_scale(a, b); }
template <int M, class S, PCS_IFDUAL(S)> HVD HV<M, S> operator*(const S& b, const HV<M, S>& a) { return hv_scale(a, b); }
template <int M, class S> HVD HV<M, S> operator+(const HV<M, S>& a, double b) { HV<M, S> r = a; r.re = a.re + b; return r; }
template <int M, class S> HVD HV<M, S> operator+(double b, const HV<M, S>& a) { return a + b; }
template <int M, class S> HVD HV<M, S> operator-(const HV<M, S>& a, double b) { HV<M, S> r = a; r.re = a.re - b; return r; }
template <int M, class S> HVD HV<M, S> operator-(double b, const HV<M, S>& a) { return (a * -1.0) + b; }
template <int M, class S> HVD HV<M, S> hv_recip(const HV<M, S>& a) { S r = s_recip(a.re), r2 = r * r; return a.chain(r, -r2, 2.0 * r2 * r); }
template <int M, class S> HVD HV<M, S> hv_log(const HV<M, S>& a) { S r = s_recip(a.re); return a.chain(s_log(a.re), r, -(r * r)); }
template <int M, class S> HVD HV<M, S> hv_sqrt(const HV<M, S>& a) { S s = s_sqrt(a.re), h = 0.5 * s_recip(s); return a.chain(s, h, -0.5 * h * s_recip(a.re)); }
template <int M, class S> HVD HV<M, S> operator/(const HV<M, S>& a, const HV<M, S>& b) { return a * hv_recip(b); }

template <int N, int M, class S, class C>
HVD HV<M, S> hv_poly(const C* coef, const HV<M, S>& x) {  // sum coef[i] x^i by Horner in HV arithmetic
    HV<M, S> acc = x * coef[N - 1] + coef[N - 2];
#pragma unroll
    for (int i = N - 3; i >= 0; i--) acc = acc * x + coef[i];
    return acc;
}
template <int M, class S, PCS_IFDUAL(S)> HVD HV<M, S> operator+(const HV<M, S>& a, const S& b) { HV<M, S> r = a; r.re = a.re + b; return r; }

constexpr int NBLOCK = 64;

// The model: a as HV<NC+1, S> from par [NC][8] (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb), T and the partial
// densities x, all of type S.  `bad`: more than one associating component (the reference raises).
template <int NC, class S>
HVD HV<NC + 1, S> mixn_a(const S* par, const S& T, const S* x, S& rs, bool& bad) {
    constexpr int M = NC + 1;
    typedef HV<M, S> R;
    const S rT = s_recip(T);
    S m[NC], sig[NC], eps[NC], d[NC], mu2t[NC];
    R r[NC];
    rs = S(0.0);
#pragma unroll
    for (int i = 0; i < NC; i++) {
        m[i] = par[8 * i]; sig[i] = par[8 * i + 1]; eps[i] = par[8 * i + 2];
        d[i] = sig[i] * (1.0 - 0.12 * s_exp(-3.0 * eps[i] * rT));  // :33
        const S mu = par[8 * i + 3];
        // sigma^3 eps mu2 / T with mu2 = mu^2 / (m sigma^3 eps) * MU2_UNIT  (:17-22, :163)
        mu2t[i] = mu * mu * s_recip(m[i]) * rT * MU2_UNIT;
        rs = rs + x[i];
        r[i] = R(x[i]);
        r[i].e1[i] = S(1.0);
        r[i].e1[NC] = x[i];
        r[i].e2 = x[i];
    }
    // packing sums (:35-38)
    R z0(S(0.0)), z1(S(0.0)), z2(S(0.0)), z3(S(0.0)), rsum(S(0.0)), mb(S(0.0));
#pragma unroll
    for (int i = 0; i < NC; i++) {
        const S md = m[i] * FRAC_PI_6;
        z0 = z0 + r[i] * md; z1 = z1 + r[i] * (md * d[i]); z2 = z2 + r[i] * (md * d[i] * d[i]); z3 = z3 + r[i] * (md * d[i] * d[i] * d[i]);
        rsum = rsum + r[i];
        mb = mb + r[i] * m[i];
    }
    const R omz = 1.0 - z3;
    const R z3m1 = hv_recip(omz);
    const R z3m2 = z3m1 * z3m1;
    const R z23 = z2 * hv_recip(z3);
    const R l13 = hv_log(omz);
    // hard sphere (:56-60)
    R a = (6.0 / PI) * ((z1 * z2) * z3m1 * 3.0 + (z2 * z2) * z3m2 * z23 + (z2 * (z23 * z23) - z0) * l13);
    // hard chain (:63-65): g_i = 1/(1-z3) + 1.5 d_i c + 0.5 d_i^2 c^2 (1 - z3), c = z2/(1-z3)^2
    const R cc = z2 * z3m2;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        const R cd = cc * d[i];
        const R g = z3m1 + cd * 1.5 + ((cd * cd) * omz) * 0.5;
        a = a - (r[i] * (m[i] - 1.0)) * hv_log(g);
    }
    // dispersion (:69-106), no k_ij
    {
        const R mbar = mb * hv_recip(rsum);
        const R rmb = hv_recip(mbar);
        const R m1 = (mbar - 1.0) * rmb;
        const R m2 = m1 * ((mbar - 2.0) * rmb);
        const R I1 = hv_poly<7>(A0, z3) + m1 * hv_poly<7>(A1, z3) + m2 * hv_poly<7>(A2, z3);
        const R I2 = hv_poly<7>(B0, z3) + m1 * hv_poly<7>(B1, z3) + m2 * hv_poly<7>(B2, z3);
        const R z3m4 = z3m2 * z3m2;
        const R t2 = z3m1 * hv_recip(2.0 - z3);
        const R poly = z3 * (20.0 + z3 * (-27.0 + z3 * (12.0 - 2.0 * z3)));
        const R C1 = hv_recip(1.0 + mbar * ((z3 * (8.0 - 2.0 * z3)) * z3m4) + (1.0 - mbar) * (poly * (t2 * t2)));
        R rho1mix(S(0.0)), rho2mix(S(0.0));
#pragma unroll
        for (int i = 0; i < NC; i++) {
#pragma unroll
            for (int j = i; j < NC; j++) {
                const S e = s_sqrt(eps[i] * eps[j]) * rT, s = 0.5 * (sig[i] + sig[j]);
                const S w = (i == j ? 1.0 : 2.0) * m[i] * m[j] * (s * s * s) * e;
                const R rij = r[i] * r[j];
                rho1mix = rho1mix + rij * w;
                rho2mix = rho2mix + rij * (w * e);
            }
        }
        a = a - PI * (2.0 * (rho1mix * I1) + (rho2mix * (C1 * I2)) * mbar);
    }
    // dipoles (:156-208)
    bool polar = false;
#pragma unroll
    for (int i = 0; i < NC; i++) polar = polar || pcs::re(mu2t[i]) != 0.0;
    if (polar) {
        R phi2(S(0.0)), phi3(S(0.0));
        for (int i = 0; i < NC; i++) {  // (left to the optimizer: with the triplet loop below it is not unrolled for every NC)
            if (pcs::re(mu2t[i]) == 0.0) continue;
#pragma unroll
            for (int j = i; j < NC; j++) {
                if (pcs::re(mu2t[j]) == 0.0) continue;
                const S sij = 0.5 * (sig[i] + sig[j]);
                const S mij = s_sqrt(s_min2(m[i]) * s_min2(m[j]));
                const S rmij = s_recip(mij);
                const S q1 = (mij - 1.0) * rmij, q2 = q1 * (mij - 2.0) * rmij;
                const S et = s_sqrt(eps[i] * eps[j]) * rT;
                S cf[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    cf[k] = AD[k][0] + q1 * AD[k][1] + q2 * AD[k][2];
                    if (k < 3) cf[k] = cf[k] + (BD[k][0] + q1 * BD[k][1] + q2 * BD[k][2]) * et;
                }
                const S pref = -(i == j ? 1.0 : 2.0) * mu2t[i] * mu2t[j] * s_recip(sij * sij * sij);
                phi2 = phi2 + (r[i] * r[j]) * hv_poly<5>(cf, z3) * pref;
#pragma unroll
                for (int k = j; k < NC; k++) {
                    if (pcs::re(mu2t[k]) == 0.0) continue;
                    const S sik = 0.5 * (sig[i] + sig[k]), sjk = 0.5 * (sig[j] + sig[k]);
                    const S mijk = s_cbrt(s_min2(m[i]) * s_min2(m[j]) * s_min2(m[k]));
                    const S rmijk = s_recip(mijk);
                    const S t1 = (mijk - 1.0) * rmijk, t2_ = t1 * (mijk - 2.0) * rmijk;
                    S cg[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) cg[q] = CD[q][0] + t1 * CD[q][1] + t2_ * CD[q][2];
                    const int distinct = 1 + (j != i) + (k != j);
                    const double c3 = distinct == 1 ? 1.0 : (distinct == 2 ? 3.0 : 6.0);
                    const S pre3 = -c3 * mu2t[i] * mu2t[j] * mu2t[k] * s_recip(sij * sik * sjk);
                    phi3 = phi3 + ((r[i] * r[j]) * r[k]) * hv_poly<4>(cg, z3) * pre3;
                }
            }
        }
        phi2 = phi2 * PI;
        phi3 = phi3 * PI_SQ_43;
        if (pcs::re(phi2.re) == 0.0) a = a + phi2;  // no polar component present at this state: limit of the quotient (mix_model.hpp)
        else a = a + (phi2 * phi2) * hv_recip(phi2 - phi3);
    }
    // association (:118-152): exactly one associating component -> phi_self_assoc (:210-239)
    int associating = 0, self_assoc = 0;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        associating += (pcs::re(par[8 * i + 6]) + pcs::re(par[8 * i + 7]) != 0.0);
        self_assoc += (pcs::re(par[8 * i + 6]) * pcs::re(par[8 * i + 7]) != 0.0);
    }
    bad = associating > 1;  // "Only up to two associating components are allowed!" and two only for binary mixtures
    if (associating == 1 && self_assoc == 1) {
        S kap(0.0), eab(0.0), nas(0.0), sg(0.0), dd(0.0);
        R rhoa(S(0.0)), rhob(S(0.0));
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const S na = par[8 * i + 6], nb = par[8 * i + 7];
            kap = kap + par[8 * i + 4]; eab = eab + par[8 * i + 5]; nas = nas + na; sg = sg + na * sig[i]; dd = dd + na * d[i];
            rhoa = rhoa + r[i] * na;
            rhob = rhob + r[i] * nb;
        }
        const S rnas = s_recip(nas);
        sg = sg * rnas;
        dd = dd * rnas;
        const R k = (z2 * z3m1) * (0.5 * dd);
        const R delta = (z3m1 * (k * (2.0 * k + 3.0) + 1.0)) * ((sg * sg * sg) * kap * (s_exp(eab * rT) - 1.0));
        const R t = (rhob - rhoa) * delta;
        const R aux = 1.0 - t;
        const R sq = hv_sqrt(aux * aux + 4.0 * (rhob * delta));
        R xa, xb;  // cancellation-free site fractions (pure_model.hpp)
        if (pcs::re(t.re) > 0.5) {
            xa = 2.0 * hv_recip(sq + 1.0 + t);
            xb = (sq - 1.0 + t) * hv_recip(2.0 * (rhob * delta));
        } else if (pcs::re(t.re) < -0.5) {
            xa = (sq - 1.0 - t) * hv_recip(2.0 * (rhoa * delta));
            xb = 2.0 * hv_recip(sq + 1.0 - t);
        } else {
            xa = 2.0 * hv_recip(sq + 1.0 + t);
            xb = 2.0 * hv_recip(sq + 1.0 - t);
        }
        a = a + rhoa * (hv_log(xa) - 0.5 * xa + 0.5) + rhob * (hv_log(xb) - 0.5 * xb + 0.5);
    }
    return a;
}

template <int NC>
__global__ __launch_bounds__(NBLOCK) void k_mixn_derivatives(const double* __restrict__ params, const double* __restrict__ temp,
                                                            const double* __restrict__ rho_in, int64_t n, double* __restrict__ a_out,
                                                            double* __restrict__ p_out, double* __restrict__ mu_out,
                                                            double* __restrict__ v_out) {
    const int64_t row = (int64_t)blockIdx.x * NBLOCK + threadIdx.x;
    if (row >= n) return;
    double par[NC * 8], x[NC];
#pragma unroll
    for (int k = 0; k < NC * 8; k++) par[k] = params[(size_t)row * NC * 8 + k];
#pragma unroll
    for (int i = 0; i < NC; i++) x[i] = rho_in[(size_t)row * NC + i];
    double rs;
    bool bad;
    const HV<NC + 1, double> a = mixn_a<NC, double>(par, temp[row], x, rs, bad);
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    if (a_out) a_out[row] = bad ? nanv : a.re;
    if (p_out) p_out[row] = bad ? nanv : rs - a.re + a.e2;
    const double den = 1.0 / (rs + a.e12[NC]);
#pragma unroll
    for (int i = 0; i < NC; i++) {
        if (mu_out) mu_out[(size_t)row * NC + i] = bad ? nanv : a.e1[i];
        if (v_out) v_out[(size_t)row * NC + i] = bad ? nanv : (1.0 + a.e12[i]) * den;
    }
}

// Backward pass of pcs_mixn_derivatives (the reference's n-component model is an ordinary torch graph,
// feos_torch/pcsaft_mix.py:31-154, :395-420): grad[row, d] = d/d(input d) of
//   L = g_a a + g_p p + sum_i g_mu_i mu_i + sum_i g_v_i v_i,     inputs d = (parameters [NC][8], T, rho [NC]),
// one input direction per pass as the tangent of S = DN<double, 1> through the same model code (the density directions pick up
// the third density derivatives that d v / d rho needs).  Correct first: 9 NC + 1 evaluations per row.
template <int NC>
__global__ __launch_bounds__(NBLOCK) void k_mixn_derivatives_vjp(const double* __restrict__ params, const double* __restrict__ temp,
                                                                const double* __restrict__ rho_in, int64_t n,
                                                                const double* __restrict__ g_a, const double* __restrict__ g_p,
                                                                const double* __restrict__ g_mu, const double* __restrict__ g_v,
                                                                double* __restrict__ grad) {
    typedef DN<double, 1> S;
    constexpr int D = 9 * NC + 1;
    const int64_t row = (int64_t)blockIdx.x * NBLOCK + threadIdx.x;
    if (row >= n) return;
    const double ga = g_a ? g_a[row] : 0.0, gp = g_p ? g_p[row] : 0.0;
    double gm[NC], gv[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        gm[i] = g_mu ? g_mu[(size_t)row * NC + i] : 0.0;
        gv[i] = g_v ? g_v[(size_t)row * NC + i] : 0.0;
    }
    const double Tv = temp[row];
#pragma unroll 1
    for (int dir = 0; dir < D; dir++) {
        S par[NC * 8], x[NC], T;
#pragma unroll
        for (int k = 0; k < NC * 8; k++) { par[k].v = params[(size_t)row * NC * 8 + k]; par[k].e[0] = (k == dir) ? 1.0 : 0.0; }
        T.v = Tv; T.e[0] = (dir == NC * 8) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NC; i++) { x[i].v = rho_in[(size_t)row * NC + i]; x[i].e[0] = (dir == NC * 8 + 1 + i) ? 1.0 : 0.0; }
        S rs;
        bool bad;
        const HV<NC + 1, S> a = mixn_a<NC, S>(par, T, x, rs, bad);
        S L = a.re * ga + (rs - a.re + a.e2) * gp;
        const S den = s_recip(rs + a.e12[NC]);
#pragma unroll
        for (int i = 0; i < NC; i++) L = L + a.e1[i] * gm[i] + ((1.0 + a.e12[i]) * den) * gv[i];
        grad[(size_t)row * D + dir] = bad ? __longlong_as_double(0x7ff8000000000000LL) : L.e[0];
    }
}

}  // namespace

extern "C" {

int pcs_mixn_derivatives(const double* params, const double* temp, const double* rho, int ncomp, int64_t n, double* a, double* p,
                         double* mu, double* v, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (ncomp < 1 || ncomp > 6) return fail_msg("pcs_mixn_derivatives: ncomp must be in [1, 6]");
    if (n == 0) return 0;
    if (!params || !temp || !rho) return fail_msg("pcs_mixn_derivatives: null required pointer");
    const dim3 grid((unsigned)((n + NBLOCK - 1) / NBLOCK)), block(NBLOCK);
    hipStream_t s = as_stream(stream);
    switch (ncomp) {
        case 1: hipLaunchKernelGGL(k_mixn_derivatives<1>, grid, block, 0, s, params, temp, rho, n, a, p, mu, v); break;
        case 2: hipLaunchKernelGGL(k_mixn_derivatives<2>, grid, block, 0, s, params, temp, rho, n, a, p, mu, v); break;
        case 3: hipLaunchKernelGGL(k_mixn_derivatives<3>, grid, block, 0, s, params, temp, rho, n, a, p, mu, v); break;
        case 4: hipLaunchKernelGGL(k_mixn_derivatives<4>, grid, block, 0, s, params, temp, rho, n, a, p, mu, v); break;
        case 5: hipLaunchKernelGGL(k_mixn_derivatives<5>, grid, block, 0, s, params, temp, rho, n, a, p, mu, v); break;
        default: hipLaunchKernelGGL(k_mixn_derivatives<6>, grid, block, 0, s, params, temp, rho, n, a, p, mu, v); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_mixn_derivatives launch", e);
    return 0;
}

int pcs_mixn_derivatives_vjp(const double* params, const double* temp, const double* rho, int ncomp, int64_t n, const double* g_a,
                             const double* g_p, const double* g_mu, const double* g_v, double* grad, void* stream) {
    g_err[0] = 0;
    if (int e = check_n(n)) return e;
    if (ncomp < 1 || ncomp > 6) return fail_msg("pcs_mixn_derivatives_vjp: ncomp must be in [1, 6]");
    if (n == 0) return 0;
    if (!params || !temp || !rho || !grad) return fail_msg("pcs_mixn_derivatives_vjp: null required pointer");
    const dim3 grid((unsigned)((n + NBLOCK - 1) / NBLOCK)), block(NBLOCK);
    hipStream_t s = as_stream(stream);
    switch (ncomp) {
        case 1: hipLaunchKernelGGL(k_mixn_derivatives_vjp<1>, grid, block, 0, s, params, temp, rho, n, g_a, g_p, g_mu, g_v, grad); break;
        case 2: hipLaunchKernelGGL(k_mixn_derivatives_vjp<2>, grid, block, 0, s, params, temp, rho, n, g_a, g_p, g_mu, g_v, grad); break;
        case 3: hipLaunchKernelGGL(k_mixn_derivatives_vjp<3>, grid, block, 0, s, params, temp, rho, n, g_a, g_p, g_mu, g_v, grad); break;
        case 4: hipLaunchKernelGGL(k_mixn_derivatives_vjp<4>, grid, block, 0, s, params, temp, rho, n, g_a, g_p, g_mu, g_v, grad); break;
        case 5: hipLaunchKernelGGL(k_mixn_derivatives_vjp<5>, grid, block, 0, s, params, temp, rho, n, g_a, g_p, g_mu, g_v, grad); break;
        default: hipLaunchKernelGGL(k_mixn_derivatives_vjp<6>, grid, block, 0, s, params, temp, rho, n, g_a, g_p, g_mu, g_v, grad); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_mixn_derivatives_vjp launch", e);
    return 0;
}

}  // extern "C"
