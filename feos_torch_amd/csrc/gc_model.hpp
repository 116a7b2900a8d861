// Heterosegmented gc-PC-SAFT for binary mixtures on gfx950 — the model of the reference's
// GcPcSaftMix (feos_torch/gc_pcsaft.py:14-86 constructor preprocessing, :116-253
// helmholtz_energy_density, :255-441 dipole / association, :549-564 association_strength).
//
// Data layout (instead of the reference's dense [N,2,S] segment and [N,2,S,S] bond tensors,
// 8.5 GB at N = 1e6):
//   per row   80 bytes: for each of the 2 molecules up to 8 (segment type, count) and up to 8
//             (bond type pair, count) entries, uint8 each:
//               [ 0:16] seg_id[2][8]   [16:32] seg_cnt[2][8]
//               [32:48] bond_a[2][8]   [48:64] bond_b[2][8]   [64:80] bond_cnt[2][8]
//             + phi[2], T, z, p_init (fp64)
//   per batch one table staged in LDS by every workgroup: seg[S][8] (m, sigma, epsilon_k, mu,
//             kappa_ab, epsilon_k_ab, na, nb), E1[S][S] = sqrt(eps_a eps_b) sigma_ab^3,
//             E2[S][S] = eps_a eps_b sigma_ab^3, K[S][S] = 1 - k_ab.
// gc_coef() turns a row into molecule-level coefficients once (T-only work: segment diameters,
// packing sums, dispersion double sums, bond diameters); gc_a() is the density-dependent part.
#pragma once
#include "mix_model.hpp"

namespace pcs {

constexpr int GC_MAXE = 8;    // entries per molecule (segments / bonds)
constexpr int GC_ROW_BYTES = 80;
constexpr int GC_MAXS = 32;   // segment types per table

struct GcTable {  // views into LDS
    int S;
    const double* seg;  // [S][8]
    const double* E1;   // [S][S]
    const double* E2;
    const double* K;
};

// doubles needed in LDS for a table of S segment types
PCS_DEV int gc_table_doubles(int S) { return S * 8 + 3 * S * S; }

template <class P>
struct GcCoef {
    P m[2];        // molecule m (sum of segment m)
    P zk[4][2];
    P A[3], B[3];
    bool polar;
    P pj[3][5], tj[4][4];
    int acls;
    P na[2], nb[2];
    double isa[2];  // sign(kappa_ab eps_ab) per molecule (:317-320)
    P dij[3], S[3];
    // bonds: per thread GC_MAXE*2 diameters d_ab in LDS, strided by the workgroup size; the counts are the row's own bytes
    // [64:80] (the kernels stage the row in LDS, gc_kernel_common.hpp::stage_row)
    P* bond_dab;
    const unsigned char* bond_cnt;
    int stride;
};

PCS_DEV double sgn_d(double x) { return (x > 0.0) - (x < 0.0); }

// Molecule-level sums of a row: everything the model needs from the segment table besides the bond diameters.
// P = type of the temperature-dependent sums, Q = type of the others (double in the solvers; the segment-parameter
// gradient kernel seeds tangents on all of them: P = Q = DN).
template <class P, class Q>
struct GcMol {
    Q M[2];                                       // sum n m                               (:55, :66)
    P Zk[3][2];                                   // sum n m d^k, k = 1..3                 (:122-131)
    Q S3[2], EK[2], MU[2];                        // sum n m sigma^3, sum n m eps, sum n mu^2   (:66-73)
    Q sa[2], ea[2], ka[2], eab[2], na[2], nb[2];  // association picks / sums              (:76-86)
    Q s1[3], s2[3];                               // dispersion double sums, pairs 00, 01, 11 (:177-194) without phi and T
};

// segment diameter (:118-120)
template <class P>
PCS_DEV P gc_diameter(const double* seg, const P& rT) { return seg[1] * (1.0 - 0.12 * d_exp((-3.0 * seg[2]) * rT)); }

// row = 80 bytes of structure (see header) -> molecule-level sums; also fills the lane's bond list (d_ab, count)
template <class P>
PCS_DEV void gc_mol(GcMol<P, double>& ml, P* bond_dab, int stride, const unsigned char* row, const GcTable& tb,
                    const P& rT) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        double mtot = 0.0, s3sum = 0.0, eksum = 0.0, mu2sum = 0.0, sa = 0.0, ea = 0.0, ka = 0.0, eabs = 0.0, nas = 0.0, nbs = 0.0;
        P z1(0.0), z2(0.0), z3(0.0);
#pragma unroll 1
        for (int e = 0; e < GC_MAXE; e++) {
            const int n = row[16 + i * GC_MAXE + e];
            if (n == 0) continue;
            const double* p = tb.seg + 8 * row[i * GC_MAXE + e];
            const double ma = n * p[0];
            P d = gc_diameter<P>(p, rT);
            z1 = z1 + ma * d;
            z2 = z2 + ma * (d * d);
            z3 = z3 + ma * (d * d * d);
            mtot += ma;
            s3sum += ma * p[1] * p[1] * p[1];
            eksum += ma * p[2];
            mu2sum += n * p[3] * p[3];
            const double ia = n * sgn_d(p[4] * p[5]);
            sa += ia * p[1];
            ea += ia * p[2];
            ka += n * p[4];
            eabs += n * p[5];
            nas += n * p[6];
            nbs += n * p[7];
        }
        ml.M[i] = mtot; ml.Zk[0][i] = z1; ml.Zk[1][i] = z2; ml.Zk[2][i] = z3;
        ml.S3[i] = s3sum; ml.EK[i] = eksum; ml.MU[i] = mu2sum;
        ml.sa[i] = sa; ml.ea[i] = ea; ml.ka[i] = ka; ml.eab[i] = eabs; ml.na[i] = nas; ml.nb[i] = nbs;
    }
    // dispersion double sums (:177-194): sum m_ia m_jb E1_ab (1-k_ab)[i!=j], likewise E2 with (1-k_ab)^2
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
        const int i = (pr == 2) ? 1 : 0, j = (pr == 0) ? 0 : 1;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 1
        for (int e = 0; e < GC_MAXE; e++) {
            const int na_ = row[16 + i * GC_MAXE + e];
            if (na_ == 0) continue;
            const int ia = row[i * GC_MAXE + e];
            const double ma = na_ * tb.seg[8 * ia];
#pragma unroll 1
            for (int f = 0; f < GC_MAXE; f++) {
                const int nb_ = row[16 + j * GC_MAXE + f];
                if (nb_ == 0) continue;
                const int ib = row[j * GC_MAXE + f];
                const double mm = ma * (nb_ * tb.seg[8 * ib]);
                double t1 = tb.E1[ia * tb.S + ib], t2 = tb.E2[ia * tb.S + ib];
                if (i != j) {
                    const double k = tb.K[ia * tb.S + ib];
                    t1 *= k;
                    t2 *= k * k;
                }
                s1 += mm * t1;
                s2 += mm * t2;
            }
        }
        ml.s1[pr] = s1;
        ml.s2[pr] = s2;
    }
    // bonds (:156-165): d_ab = d_a d_b / (d_a + d_b) per bond-type entry
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll 1
        for (int e = 0; e < GC_MAXE; e++) {
            const int n = row[64 + i * GC_MAXE + e];
            const int slot = (i * GC_MAXE + e) * stride;
            if (n == 0) continue;
            P da = gc_diameter<P>(tb.seg + 8 * row[32 + i * GC_MAXE + e], rT);
            P db = gc_diameter<P>(tb.seg + 8 * row[48 + i * GC_MAXE + e], rT);
            bond_dab[slot] = (da * db) * d_recip(da + db);
        }
    }
}

// Blocks of gc_finish as functions of their own inputs (the segment-gradient kernel differentiates them one by one).
// dipoles: molecule-level averages (:66-73), mu2_term = mu2/T (:262)
template <class P, class Q>
PCS_DEV void gc_dipole_block(P pj[3][5], P tj[4][4], const Q* M, const Q* S3, const Q* EK, const Q* MU, const P& rT) {
    P mm[2], sg[2], ek[2], mu2t[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        mm[i] = P(M[i]);
        sg[i] = P(d_cbrt(S3[i] / M[i]));
        ek[i] = P(EK[i] / M[i]);
        mu2t[i] = rT * (MU[i] / M[i] * MU2_UNIT);
    }
    dipole_coefficients<P>(pj, tj, mm, sg, ek, mu2t, rT);
}
// association contact distances and strengths of class acls (:310-327, :334-356, :384-412)
template <class P, class Q>
PCS_DEV void gc_assoc_block(int acls, P* dij, P* S, const Q* sa, const Q* ea, const Q* ka, const Q* eab_, const P& rT) {
    if (acls == ASSOC_SELF) {
        Q sg = sa[0] + sa[1], ek = ea[0] + ea[1], kap = ka[0] + ka[1], eab = eab_[0] + eab_[1];
        P d = sg * (1.0 - 0.12 * d_exp((-3.0 * ek) * rT));
        dij[0] = 0.5 * d;
        S[0] = (sg * sg * sg * kap) * (d_exp(eab * rT) - 1.0);
    } else if (acls != ASSOC_NONE) {
        P d0 = sa[0] * (1.0 - 0.12 * d_exp((-3.0 * ea[0]) * rT));
        P d1 = sa[1] * (1.0 - 0.12 * d_exp((-3.0 * ea[1]) * rT));
        dij[0] = 0.5 * d0;
        dij[1] = (d0 * d1) * d_recip(d0 + d1);
        dij[2] = 0.5 * d1;
        Q ss = sa[0] * sa[1];
        S[0] = (sa[0] * sa[0] * sa[0] * ka[0]) * (d_exp(eab_[0] * rT) - 1.0);
        S[1] = (ss * d_sqrt(ss) * d_sqrt(ka[0] * ka[1])) * (d_exp((0.5 * (eab_[0] + eab_[1])) * rT) - 1.0);
        S[2] = (sa[1] * sa[1] * sa[1] * ka[1]) * (d_exp(eab_[1] * rT) - 1.0);
    }
}

// molecule-level sums -> coefficients of gc_a (bond list aside).  Either Q = double or Q = P.
template <class P, class Q>
PCS_DEV void gc_finish(GcCoef<P>& c, const GcMol<P, Q>& ml, double phi0, double phi1, const P& rT) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        c.m[i] = P(ml.M[i]);
        c.zk[0][i] = P(ml.M[i] * FRAC_PI_6);  // :122-131
#pragma unroll
        for (int k = 1; k < 4; k++) c.zk[k][i] = ml.Zk[k - 1][i] * FRAC_PI_6;
    }
    // dispersion aggregates (:177-194): A_ij = sqrt(phi_i phi_j)/T sum m_ia m_jb E1_ab (1-k_ab)[i!=j]
    {
        const double p00 = phi0, p01 = sqrt(phi0 * phi1), p11 = phi1;
        P rT2 = rT * rT;
        c.A[0] = rT * (p00 * ml.s1[0]);
        c.A[1] = rT * (2.0 * p01 * ml.s1[1]);
        c.A[2] = rT * (p11 * ml.s1[2]);
        c.B[0] = rT2 * (p00 * p00 * ml.s2[0]);
        c.B[1] = rT2 * (2.0 * phi0 * phi1 * ml.s2[1]);
        c.B[2] = rT2 * (p11 * p11 * ml.s2[2]);
    }
    // dipoles: molecule-level averages (:66-73), mu2_term = mu2/T (:262)
    c.polar = (re(ml.MU[0]) > 0.0) || (re(ml.MU[1]) > 0.0);
    if (c.polar) gc_dipole_block<P, Q>(c.pj, c.tj, ml.M, ml.S3, ml.EK, ml.MU, rT);
    // association (:76-86, :221-251)
    const int associating = (re(ml.ka[0]) * re(ml.eab[0]) != 0.0) + (re(ml.ka[1]) * re(ml.eab[1]) != 0.0);
    const int self_assoc = (re(ml.na[0]) * re(ml.nb[0]) != 0.0) + (re(ml.na[1]) * re(ml.nb[1]) != 0.0);
    c.acls = ASSOC_NONE;
    if (associating == 1 && self_assoc == 1) c.acls = ASSOC_SELF;
    if (associating == 2 && self_assoc == 2) c.acls = ASSOC_CROSS;
    if (associating == 2 && self_assoc == 1) c.acls = ASSOC_INDUCED;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        c.na[i] = P(ml.na[i]);
        c.nb[i] = P(ml.nb[i]);
        c.isa[i] = sgn_d(re(ml.ka[i]) * re(ml.eab[i]));
    }
    gc_assoc_block<P, Q>(c.acls, c.dij, c.S, ml.sa, ml.ea, ml.ka, ml.eab, rT);
}

// row -> coefficients: T-only work done once per state point (segment diameters, packing sums, dispersion double
// sums, bond diameters, dipole / association coefficients)
template <class P>
PCS_DEV void gc_coef(GcCoef<P>& c, const unsigned char* row, const GcTable& tb, double phi0, double phi1, const P& T) {
    P rT = d_recip(T);
    GcMol<P, double> ml;
    c.bond_cnt = row + 64;
    gc_mol<P>(ml, c.bond_dab, c.stride, row, tb, rT);
    gc_finish<P, double>(c, ml, phi0, phi1, rT);
}

// gc cross association, hard-coded nA = nB = 1 (:361-374): f_i = X_i + X_i sum_j X_j D_ij - 1
template <class X>
PCS_DEV void gc_cross_step(const X& x0, const X& x1, const X& d00, const X& d01, const X& d10, const X& d11, X& dx0, X& dx1) {
    X f0 = x0 * (1.0 + x0 * d00 + x1 * d01) - 1.0;
    X f1 = x1 * (1.0 + x0 * d10 + x1 * d11) - 1.0;
    X j00 = 1.0 + 2.0 * (x0 * d00) + x1 * d01, j01 = x0 * d01;
    X j10 = x1 * d10, j11 = 1.0 + x0 * d10 + 2.0 * (x1 * d11);
    X rdet = d_recip(j00 * j11 - j01 * j10);
    dx0 = (j11 * f0 - j01 * f1) * rdet;
    dx1 = (j00 * f1 - j10 * f0) * rdet;
}

// see cross_refine (mix_model.hpp)
template <class R>
PCS_DEV void gc_cross_refine(R& xa0, R& xa1, const R& d00, const R& d01, const R& d10, const R& d11) {
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        R dx0, dx1;
        gc_cross_step<R>(xa0, xa1, d00, d01, d10, d11, dx0, dx1);
        xa0 = xa0 - dx0;
        xa1 = xa1 - dx1;
    }
}
template <class T>
PCS_DEV void gc_cross_refine(T2<T>& xa0, T2<T>& xa1, const T2<T>& d00, const T2<T>& d01, const T2<T>& d10, const T2<T>& d11) {
    T1<T> y0 = lower(xa0), y1 = lower(xa1), s0, s1;
    gc_cross_step<T1<T>>(y0, y1, lower(d00), lower(d01), lower(d10), lower(d11), s0, s1);
    xa0 = raise(y0 - s0);
    xa1 = raise(y1 - s1);
    T2<T> dx0, dx1;
    gc_cross_step<T2<T>>(xa0, xa1, d00, d01, d10, d11, dx0, dx1);
    xa0 = xa0 - dx0;
    xa1 = xa1 - dx1;
}

// zeta3: the packing fraction as Z (R in general, D2<double> in the solvers' (zeta_3, rho_2) coordinates: mix_model.hpp Packing)
template <class P, class R, class Z>
PCS_DEV R gc_a_z(const GcCoef<P>& c, const R& r0, const R& r1, const Z& zeta3) {
    Packing<R, Z> pk;
    R a = core_terms_z<GcCoef<P>, R, Z>(c, r0, r1, zeta3, pk);
    const R& zeta2 = pk.zeta2;
    const Z& z3m1 = pk.z3m1;

    // hard chain over bond types (:156-165): g = 1/(1-z3) + 3 c d_ab + 2 (c d_ab)^2 (1 - z3)
    R cc = zeta2 * pk.z3m2;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const R& ri = (i == 0) ? r0 : r1;
#pragma unroll 1
        for (int e = 0; e < GC_MAXE; e++) {
            const int slot = (i * GC_MAXE + e) * c.stride;
            const double n = (double)c.bond_cnt[i * GC_MAXE + e];
            if (n == 0.0) continue;
            R cd = cc * c.bond_dab[slot];
            R g = z3m1 + 3.0 * cd + 2.0 * ((cd * cd) * pk.omz);
            a = a - (ri * n) * d_log(g);
        }
    }

    if (c.acls == ASSOC_SELF) {  // phi_assoc (:309-330), nA = nB = 1
        R k = (zeta2 * z3m1) * c.dij[0];
        R rho_a = r0 * c.isa[0] + r1 * c.isa[1];
        R deltarho = ((z3m1 * (k * (2.0 * k + 3.0) + 1.0)) * c.S[0]) * rho_a;
        R xa = 2.0 * d_recip(d_sqrt(1.0 + 4.0 * deltarho) + 1.0);
        a = a + rho_a * (2.0 * d_log(xa) - xa + 1.0);
    } else if (c.acls == ASSOC_CROSS || c.acls == ASSOC_INDUCED) {
        R zz = zeta2 * z3m1;
        R D[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            R k = zz * c.dij[q];
            D[q] = (z3m1 * (k * (2.0 * k + 3.0) + 1.0)) * c.S[q];
        }
        R d00 = D[0] * r0, d01 = D[1] * r1, d10 = D[1] * r0, d11 = D[2] * r1;  // delta_rho(i, j) = Delta_ij rho_j
        double e00 = re(d00), e01 = re(d01), e10 = re(d10), e11 = re(d11);
        if (c.acls == ASSOC_CROSS) {
            double x0 = 0.2, x1 = 0.2;  // :358
            for (int it = 0; it < 200; it++) {
                double s0, s1;
                gc_cross_step<double>(x0, x1, e00, e01, e10, e11, s0, s1);
                double n0 = x0 - s0, n1 = x1 - s1;
                if (!(n0 > 0.0 && n0 <= 1.5 && n1 > 0.0 && n1 <= 1.5)) {
                    if (it < 60 && is_finite_bits(s0) && is_finite_bits(s1)) {  // Newton step in ln X (see mix_model.hpp)
                        n0 = fmin(x0 * exp(fmin(fmax(-s0 / x0, -3.0), 3.0)), 1.0);
                        n1 = fmin(x1 * exp(fmin(fmax(-s1 / x1, -3.0), 3.0)), 1.0);
                    } else {  // successive substitution fallback
                        n0 = 1.0 / (1.0 + x0 * e00 + x1 * e01);
                        n1 = 1.0 / (1.0 + x0 * e10 + x1 * e11);
                    }
                }
                // 1e-12 is enough: the two Newton updates in R arithmetic below square the remaining error
                bool conv = fabs(n0 - x0) <= 1e-12 * x0 && fabs(n1 - x1) <= 1e-12 * x1;
                x0 = n0;
                x1 = n1;
                if (conv) break;
            }
            R xa0 = lift_real<R>(x0), xa1 = lift_real<R>(x1);
            gc_cross_refine(xa0, xa1, d00, d01, d10, d11);
            a = a + r0 * (2.0 * d_log(xa0) - xa0 + 1.0) + r1 * (2.0 * d_log(xa1) - xa1 + 1.0);  // :379-380
        } else {
            double n0 = re(c.na[0]), n1 = re(c.na[1]), m0 = re(c.nb[0]), m1n = re(c.nb[1]);
            double x = 0.2, lo = 0.0, hi = 2.0;
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
            R sa_ = site_term(xa);
            a = a + r0 * (sa_ * na0 + site_term(xb0) * nb0) + r1 * (sa_ * na1 + site_term(xb1) * nb1);  // :438-441
        }
    }
    return a;
}
template <class P, class R>
PCS_DEV R gc_a(const GcCoef<P>& c, const R& r0, const R& r1) {
    return gc_a_z<P, R, R>(c, r0, r1, r0 * c.zk[3][0] + r1 * c.zk[3][1]);
}

}  // namespace pcs
