// Gradient of the gc-PC-SAFT bubble / dew pressure w.r.t. the SEGMENT parameter table [S,8]
// (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb) on gfx950.
//
// In the reference the eight segment-parameter vectors are torch tensors that enter every derived quantity
// of GcPcSaftMix.__init__ and helmholtz_energy_density (feos_torch/gc_pcsaft.py:14-22, :54-86, :116-253), so
// reverse mode through bubble_point / dew_point (:470-512) reaches them; parameter fitting is what the library
// is for (README.md:21-23).  Here:
//   1. implicit-function form of the pressure derivative at the converged densities, exactly as mix_jacobian.hpp:
//      dp/dtheta = sum over the two phases of d/dtheta [alpha a + beta . grad_rho a] with row constants (alpha, beta);
//   2. a row depends on the table only through 13 molecule-level sums per molecule (GcMol: M, Z1..3, S3, EK, MU and
//      the association picks), the six dispersion double sums and the bond diameters.  The 26 sums are dual-number
//      directions (one quantity of both molecules per pass, D1<DN<double,2>> along beta, passes skipped per wave for
//      non-polar / non-associating rows); the dispersion sums and the bond diameters enter a linearly resp. through a
//      three-line closed form and are done analytically in D1<double>;
//   3. the chain from the sums to the parameters of the <= 8 segment types of each molecule is a few multiply-adds per
//      entry; g_i * dp_i/dtheta is accumulated over the rows of a workgroup in an LDS copy of the [S,8] table
//      (ds_add_f64) and flushed once per workgroup with global fp64 atomics (persistent grid: 2048 flushes per call).
// For the bubble / dew pressures only the table gradient is produced here; k_ab, phi and T are pcs_gc_jacobian's.
//
// The same machinery serves the vector-Jacobian product of GcPcSaftMix.derivatives / helmholtz_energy_density
// (feos_torch/gc_pcsaft.py:443-468; MODE 1): there the functional is L = ga a + gp p + gmu . mu + gv . v at ONE density
// point, linear in the six Taylor coefficients of a (DerivWeights, mix_jacobian.hpp), so the probes are T2<DN> / T2<double>
// instead of D1 along beta, and the kernel also returns the per-row dL/d(6 aggregates, T, rho_0, rho_1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"
#include "gc_model.hpp"
#include "gc_kernel_common.hpp"
#include "mix_solver.hpp"
#include "mix_jacobian.hpp"  // DerivWeights

namespace {

constexpr int GSBLOCK = 64;
constexpr int GS_GRID = 2048;
// dual-number directions per pass.  MODE 0 (D1 probes): one molecule-level quantity of BOTH molecules (slot j = molecule j);
// MODE 1 (T2 probes, three times the state per value): one molecule per pass -- with two the stack frame is 4.4 KB per lane
template <int MODE> struct GsChunk { static constexpr int value = MODE == 0 ? 2 : 1; };
// LDS doubles per thread behind the table and the gradient accumulator: the double model's bond list (4 * MAXE) + the dual model's
// bond diameters, which double as the lane-strided coefficient-adjoint array of MODE 0 (ADJ_SLOTS); the row bytes follow (stage_row)
// (MODE 0: the coefficient adjoints, the dual model is not instantiated; MODE 1: the dual model with CH = 1 direction)
template <int MODE> struct GsPerThread { static constexpr int value = 2 * GC_MAXE + (MODE == 0 ? ADJ_SLOTS : 2 * GC_MAXE * (1 + GsChunk<1>::value)); };

enum : int { Q_M, Q_Z1, Q_Z2, Q_Z3, Q_S3, Q_EK, Q_MU, Q_SA, Q_EA, Q_KA, Q_EAB, Q_NA, Q_NB, Q_COUNT };

// Accumulation into the workgroup's [S,8] gradient table in LDS: ds_add_f64 per contributing lane (0 = 0).
// 0 = 1 (experiment, off): the workgroup is ONE wave and with class-ordered rows many of its lanes add to the
// same table entry at the same time; there the lanes that target the same entry are summed with a DPP butterfly and ONE
// lane does a plain read-modify-write, once per distinct entry in the wave.  Measured on 1e6 rows: 6.55 ms against 4.71 ms
// with the atomics -- the waits of this kernel (PMC: 71 % of the wave cycles) are not the LDS atomics.  Either way every
// lane of the wave reaches the call (on = this lane contributes).
__device__ __forceinline__ double wave_sum(double x) {
#define PCS_DPP_STEP(ctrl, rmask)                                                                \
    {                                                                                            \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, rmask, 0xf, false); \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, rmask, 0xf, false); \
        x += __hiloint2double(hi, lo);                                                           \
    }
    PCS_DPP_STEP(0xB1, 0xf)   // quad_perm [1,0,3,2]
    PCS_DPP_STEP(0x4E, 0xf)   // quad_perm [2,3,0,1]
    PCS_DPP_STEP(0x141, 0xf)  // row_half_mirror
    PCS_DPP_STEP(0x140, 0xf)  // row_mirror: every lane holds its row's sum
    PCS_DPP_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    PCS_DPP_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
#undef PCS_DPP_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
__device__ __forceinline__ void lds_add(double* p, double v, bool on) {
    if (on) unsafeAtomicAdd(p, v);
}

// d and its derivatives w.r.t. sigma and epsilon_k of the segment (:118-120)
__device__ __forceinline__ void diameter_grad(const double* seg, double rT, double& d, double& d_sig, double& d_eps) {
    const double ex = exp(-3.0 * seg[2] * rT);
    const double f = 1.0 - 0.12 * ex;
    d = seg[1] * f;
    d_sig = f;
    d_eps = seg[1] * (0.36 * rT) * ex;
}

// coefficients with the tangents of molecule-level quantity q; out of line as mix_coef_tangent.  CH = 2: slot j = molecule
// j; CH = 1: slot 0 = molecule jm
template <class G, int CH>
__device__ __attribute__((noinline)) void gc_finish_tangent(GcCoef<G>& c, const GcMol<double, double>& ml, int q, int jm,
                                                            double phi0, double phi1, double rT) {
    GcMol<G, G> g;
#define PCS_SEED(field, code)                                   \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {             \
        g.field[i] = G(ml.field[i]);                            \
        if (q == code && (CH == 2 || i == jm)) g.field[i].e[CH == 2 ? i : 0] = 1.0; \
    }
    PCS_SEED(M, Q_M)
    PCS_SEED(S3, Q_S3)
    PCS_SEED(EK, Q_EK)
    PCS_SEED(MU, Q_MU)
    PCS_SEED(sa, Q_SA)
    PCS_SEED(ea, Q_EA)
    PCS_SEED(ka, Q_KA)
    PCS_SEED(eab, Q_EAB)
    PCS_SEED(na, Q_NA)
    PCS_SEED(nb, Q_NB)
#undef PCS_SEED
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            g.Zk[k][i] = G(ml.Zk[k][i]);
            if (q == Q_Z1 + k && (CH == 2 || i == jm)) g.Zk[k][i].e[CH == 2 ? i : 0] = 1.0;
        }
        g.s1[k] = G(ml.s1[k]);
        g.s2[k] = G(ml.s2[k]);
    }
    gc_finish<G, G>(c, g, phi0, phi1, G(rT));
}


// Coefficient adjoints of the gc Helmholtz energy along (q, b), as mix_a_adjoint (mix_adjoint.hpp): shared packing / hard
// sphere / dispersion / dipole part, the bond-based chain term (only its packing-sum dependence here: the bond diameters
// are part (3) of the kernel) and gc's association forms -- one site of each kind per molecule (:309-330, :358-380), whose
// energy is stationary in the site fractions:  d/dDelta = -rho_a^2 X^2 (self),  d/dDelta_ij = -rho_i rho_j X_i X_j (cross);
// induced association as in the binary model.
__device__ __attribute__((noinline)) void gc_a_adjoint(const GcCoef<double>& c, double q0, double q1, double b0, double b1, double alpha,
                                                       double* out, int stride) {
    typedef D1s R;
    AdjCtx x;
    adjoint_core(c, q0, q1, b0, b1, alpha, out, stride, x);
    const R &r0 = x.r0, &r1 = x.r1;
    {
        const R cc = x.zeta2 * x.z3m2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const R& ri = i == 0 ? r0 : r1;
#pragma unroll 1
            for (int e = 0; e < GC_MAXE; e++) {
                const int slot = (i * GC_MAXE + e) * c.stride;
                const double n = (double)c.bond_cnt[i * GC_MAXE + e];
                if (n == 0.0) continue;
                const double dab = c.bond_dab[slot];
                const R cd = cc * dab;
                const R g = x.z3m1 + 3.0 * cd + 2.0 * ((cd * cd) * x.omz);
                const R pre = (ri * n) * d_recip(g);
                x.dz2 = x.dz2 - pre * ((3.0 * dab + (4.0 * dab) * (cd * x.omz)) * x.z3m2);
                x.dz3 = x.dz3 - pre * (x.z3m2 + 6.0 * (cd * x.z3m1) + 6.0 * (cd * cd));
            }
        }
    }
    if (c.acls != ASSOC_NONE) {
        const int nq = c.acls == ASSOC_SELF ? 1 : 3;
        AdjDelta dl;
        adjoint_delta(c, x, nq, dl);
        R dD[3];
        if (c.acls == ASSOC_SELF) {
            const R rho_a = r0 * c.isa[0] + r1 * c.isa[1];
            const R xa = 2.0 * d_recip(d_sqrt(1.0 + 4.0 * (dl.D[0] * rho_a)) + 1.0);
            dD[0] = -((rho_a * rho_a) * (xa * xa));
        } else if (c.acls == ASSOC_CROSS) {
            const R d00 = dl.D[0] * r0, d01 = dl.D[1] * r1, d10 = dl.D[1] * r0, d11 = dl.D[2] * r1;
            const double e00 = re(d00), e01 = re(d01), e10 = re(d10), e11 = re(d11);
            double x0 = 0.2, x1 = 0.2;  // real parts exactly as gc_a
            for (int it = 0; it < 200; it++) {
                double s0, s1;
                gc_cross_step<double>(x0, x1, e00, e01, e10, e11, s0, s1);
                double n0 = x0 - s0, n1 = x1 - s1;
                if (!(n0 > 0.0 && n0 <= 1.5 && n1 > 0.0 && n1 <= 1.5)) {
                    if (it < 60 && is_finite_bits(s0) && is_finite_bits(s1)) {
                        n0 = fmin(x0 * exp(fmin(fmax(-s0 / x0, -3.0), 3.0)), 1.0);
                        n1 = fmin(x1 * exp(fmin(fmax(-s1 / x1, -3.0), 3.0)), 1.0);
                    } else {
                        n0 = 1.0 / (1.0 + x0 * e00 + x1 * e01);
                        n1 = 1.0 / (1.0 + x0 * e10 + x1 * e11);
                    }
                }
                bool conv = fabs(n0 - x0) <= 1e-12 * x0 && fabs(n1 - x1) <= 1e-12 * x1;
                x0 = n0;
                x1 = n1;
                if (conv) break;
            }
            R xa0(x0), xa1(x1);
            gc_cross_refine(xa0, xa1, d00, d01, d10, d11);
            dD[0] = -((r0 * r0) * (xa0 * xa0));
            dD[1] = -(2.0 * ((r0 * r1) * (xa0 * xa1)));
            dD[2] = -((r1 * r1) * (xa1 * xa1));
        } else {
            R dn[4];
            induced_assoc_adjoint(c.na[0], c.na[1], c.nb[0], c.nb[1], r0, r1, dl.D[0], dl.D[1], dl.D[2], dn, dD);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                out[(ADJ_NA + k) * stride] += alpha * dn[k].v + dn[k].d1;
                out[(ADJ_NB + k) * stride] += alpha * dn[2 + k].v + dn[2 + k].d1;
            }
        }
        adjoint_delta_chain(c, x, nq, dl, dD, alpha, out, stride);
    }
    adjoint_flush(x, alpha, out, stride);
}

// From the coefficient adjoints to the 13 molecule-level sums of both molecules: val[q][j] = d/d(sum q of molecule j) of
// sum_k adj[k] c_k.  The packing sums are linear (gc_finish); the dipole polynomials depend on (M, S3, EK, MU) and the
// association strengths on (sa, ea, ka, eab, na, nb) of both molecules: those two blocks are differentiated forward with
// just their inputs seeded.
__device__ __attribute__((noinline)) void gc_finish_gradient(const GcMol<double, double>& ml, int acls, bool polar, double rT, const double* adj,
                                                             int st, double* __restrict__ val /* [Q_COUNT][2] */) {
#define PCS_AD(slot) adj[(slot) * st]
#pragma unroll
    for (int k = 0; k < Q_COUNT * 2; k++) val[k] = 0.0;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        val[Q_M * 2 + j] = PCS_AD(ADJ_M + j) + FRAC_PI_6 * PCS_AD(ADJ_ZK + j);
#pragma unroll
        for (int k = 1; k < 4; k++) val[(Q_Z1 + k - 1) * 2 + j] = FRAC_PI_6 * PCS_AD(ADJ_ZK + 2 * k + j);
    }
    if (polar) {
        typedef DN<double, 8> G;  // M0, M1, S3_0, S3_1, EK0, EK1, MU0, MU1
        const double v[8] = {ml.M[0], ml.M[1], ml.S3[0], ml.S3[1], ml.EK[0], ml.EK[1], ml.MU[0], ml.MU[1]};
        G x[8];
        seed_inputs<8>(x, v);
        G pj[3][5], tj[4][4];
        gc_dipole_block<G, G>(pj, tj, &x[0], &x[2], &x[4], &x[6], G(rT));
        G S(0.0);
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
#pragma unroll
            for (int k = 0; k < 5; k++) S = S + pj[pr][k] * PCS_AD(ADJ_PJ + 5 * pr + k);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < 4; k++) S = S + tj[t][k] * PCS_AD(ADJ_TJ + 4 * t + k);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            val[Q_M * 2 + j] += S.e[j];
            val[Q_S3 * 2 + j] += S.e[2 + j];
            val[Q_EK * 2 + j] += S.e[4 + j];
            val[Q_MU * 2 + j] += S.e[6 + j];
        }
    }
    if (acls != ASSOC_NONE) {
        typedef DN<double, 8> G;  // sa0, sa1, ea0, ea1, ka0, ka1, eab0, eab1
        const double v[8] = {ml.sa[0], ml.sa[1], ml.ea[0], ml.ea[1], ml.ka[0], ml.ka[1], ml.eab[0], ml.eab[1]};
        G x[8];
        seed_inputs<8>(x, v);
        G dij[3], Sx[3];
        gc_assoc_block<G, G>(acls, dij, Sx, &x[0], &x[2], &x[4], &x[6], G(rT));
        const int nq = acls == ASSOC_SELF ? 1 : 3;
        G S(0.0);
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (q < nq) S = S + dij[q] * PCS_AD(ADJ_DIJ + q) + Sx[q] * PCS_AD(ADJ_S + q);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            val[Q_SA * 2 + j] = S.e[j];
            val[Q_EA * 2 + j] = S.e[2 + j];
            val[Q_KA * 2 + j] = S.e[4 + j];
            val[Q_EAB * 2 + j] = S.e[6 + j];
            val[Q_NA * 2 + j] = PCS_AD(ADJ_NA + j);  // the site counts enter gc_a directly (induced association only)
            val[Q_NB * 2 + j] = PCS_AD(ADJ_NB + j);
        }
    }
#undef PCS_AD
}

// the functional whose table gradient is taken: MODE 0 = bubble / dew pressure (two phases, D1 probes along beta),
// MODE 1 = L = ga a + gp p + gmu . mu + gv . v of `derivatives` (one point, T2 probes)
template <int MODE, class X> struct ProbeT { typedef D1<X> type; };
template <class X> struct ProbeT<1, X> { typedef T2<X> type; };

struct GcGradArgs {
    int dew;
    const double* table; int S; const unsigned char* rows; const double* phi; const double* temp;
    const double* rho;   // MODE 0: rho4 [n,4], MODE 1: rho [n,2]
    int64_t n;
    const double* gout;  // MODE 0: upstream dL/dp [n] or NULL
    const double *g_a, *g_p, *g_mu, *g_v;  // MODE 1: upstream gradients (each may be NULL)
    double* grad_seg;    // [S,8] accumulated
    double* jac9;        // MODE 1: [n,9] dL/d(A00, A01, A11, B00, B01, B11, T, rho_0, rho_1)
    double* agg;         // MODE 1: [n,6] aggregate values (optional)
    const int32_t* order;
};

// BLOCK: 64, or 192 where the LDS of three waves fits one CU next to ONE copy of the table (launch_gc_gradient)
template <int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_gc_segment_gradient(const GcGradArgs A_) {
    const int dew = A_.dew, S = A_.S;
    const double* __restrict__ table = A_.table;
    const unsigned char* __restrict__ rows = A_.rows;
    const double* __restrict__ phi = A_.phi;
    const double* __restrict__ temp = A_.temp;
    const int64_t n = A_.n;
    const double* __restrict__ gout = A_.gout;
    double* __restrict__ grad_seg = A_.grad_seg;
    const int32_t* __restrict__ order = A_.order;
    constexpr int NPT = MODE == 0 ? 2 : 1;
    constexpr int CH = GsChunk<MODE>::value;
    typedef DN<double, CH> G;
    typedef typename ProbeT<MODE, G>::type R;
    typedef typename ProbeT<MODE, double>::type Q1;
    extern __shared__ double lds[];
    GcTable tb = stage_table(table, S, lds);
    double* acc = lds + gc_table_doubles(S);                          // [S][8] gradient of this workgroup
    double* bonds = acc + S * 8;                                      // double model: [2*MAXE dab] x block
    G* gbonds = reinterpret_cast<G*>(bonds + 2 * GC_MAXE * BLOCK);  // dual model dab (zero tangents)
    double* row_area = bonds + GsPerThread<MODE>::value * BLOCK;     // the lanes' row bytes (stage_row)
    for (int k = threadIdx.x; k < S * 8; k += BLOCK) acc[k] = 0.0;
    __syncthreads();

    GcModelT<double> m;
    m.c.bond_dab = bonds + threadIdx.x;
    m.c.stride = BLOCK;
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);

    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        // every lane runs the body (the accumulation below is a wave-level reduction); lanes past the end repeat a valid
        // row with weight zero
        int64_t i = tile * BLOCK + threadIdx.x;
        bool live = i < n;
        if (!live) i = n - 1;
        if (order) {
            i = order[i];
            if (i < 0 || i >= n) { live = false; i = 0; }
        }
        const unsigned char* row = stage_row(rows + (size_t)i * GC_ROW_BYTES, row_area);  // read byte by byte in the loops below
        const double T = temp[i], ph0 = phi[2 * i], ph1 = phi[2 * i + 1];
        const double rT = 1.0 / T;
        // density points of the functional: pt 0 = specified phase (MODE 0) or the state point (MODE 1), pt 1 = incipient phase
        double s0, s1, i0, i1;
        if (MODE == 0) {
            const double4 r4 = reinterpret_cast<const double4*>(A_.rho)[i];  // (V0, V1, L0, L1)
            s0 = dew ? r4.x : r4.z; s1 = dew ? r4.y : r4.w; i0 = dew ? r4.z : r4.x; i1 = dew ? r4.w : r4.y;
        } else {
            s0 = i0 = A_.rho[2 * i]; s1 = i1 = A_.rho[2 * i + 1];
        }

        GcMol<double, double> ml;
        m.c.bond_cnt = row + 64;
        gc_mol<double>(ml, m.c.bond_dab, m.c.stride, row, tb, rT);
        gc_finish<double, double>(m.c, ml, ph0, ph1, rT);

        double alpha[2], beta0[2], beta1[2];
        DerivWeights dw;
        bool ok = true;
        if (MODE == 0) {
            // adjoint of the converged state (mix_jacobian.hpp): J^T w = dp^vap/du
            PhaseEval s = phase_eval(m, s0, s1);
            PhaseEval nn = phase_eval(m, i0, i1);
            const double rs = s0 + s1, z0 = s0 / rs, z1 = s1 / rs;
            double J[3][3];
            J[0][0] = rs * (z0 * (1.0 / s.r0 + s.h00) + z1 * s.h01);
            J[1][0] = rs * (z0 * s.h01 + z1 * (1.0 / s.r1 + s.h11));
            J[2][0] = rs * (z0 * s.dp0() + z1 * s.dp1());
            J[0][1] = -i0 * (1.0 / i0 + nn.h00);
            J[1][1] = -i0 * nn.h01;
            J[2][1] = -i0 * nn.dp0();
            J[0][2] = -i1 * nn.h01;
            J[1][2] = -i1 * (1.0 / i1 + nn.h11);
            J[2][2] = -i1 * nn.dp1();
            double A[3][4];
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
                for (int cc = 0; cc < 3; cc++) A[r][cc] = J[cc][r];
                A[r][3] = 0.0;
            }
            if (dew) {
                A[0][3] = J[2][0];
            } else {
                A[1][3] = -J[2][1];
                A[2][3] = -J[2][2];
            }
            double w[3];
            ok = solve3(A, w);
            if (dew) {  // the specified phase is the vapour
                const double u = 1.0 - w[2];
                alpha[0] = -u;     beta0[0] = u * s0 - w[0];     beta1[0] = u * s1 - w[1];
                alpha[1] = -w[2];  beta0[1] = w[2] * i0 + w[0];  beta1[1] = w[2] * i1 + w[1];
            } else {
                const double u = 1.0 + w[2];
                alpha[0] = w[2];   beta0[0] = -w[2] * s0 - w[0]; beta1[0] = -w[2] * s1 - w[1];
                alpha[1] = -u;     beta0[1] = u * i0 + w[0];     beta1[1] = u * i1 + w[1];
            }
        } else {
            const PhaseEval e = phase_eval(m, s0, s1);
            dw = deriv_weights(e, A_.g_a ? A_.g_a[i] : 0.0, A_.g_p ? A_.g_p[i] : 0.0, A_.g_mu ? A_.g_mu[2 * i] : 0.0,
                               A_.g_mu ? A_.g_mu[2 * i + 1] : 0.0, A_.g_v ? A_.g_v[2 * i] : 0.0, A_.g_v ? A_.g_v[2 * i + 1] : 0.0);
            alpha[0] = alpha[1] = beta0[0] = beta0[1] = beta1[0] = beta1[1] = 0.0;
        }
        // the two probe flavours: seeds of the density arguments and the contraction of a result with the row's weights
        auto seed0 = [&](int pt, auto zero) {
            typedef decltype(zero) X;
            if constexpr (MODE == 0) return D1<X>(X(pt == 0 ? s0 : i0), X(pt == 0 ? beta0[0] : beta0[1]));
            else return T2<X>(X(s0), X(1.0), X(0.0), X(0.0), X(0.0), X(0.0));
        };
        auto seed1 = [&](int pt, auto zero) {
            typedef decltype(zero) X;
            if constexpr (MODE == 0) return D1<X>(X(pt == 0 ? s1 : i1), X(pt == 0 ? beta1[0] : beta1[1]));
            else return T2<X>(X(s1), X(0.0), X(1.0), X(0.0), X(0.0), X(0.0));
        };
        auto dot_g = [&](const R& a, int pt, int j) -> double {
            if constexpr (MODE == 0) return (pt == 0 ? alpha[0] : alpha[1]) * a.v.e[j] + a.d1.e[j];
            else return deriv_contract(dw, a, j);
        };
        auto dot_q = [&](const Q1& a, int pt) -> double {
            if constexpr (MODE == 0) return (pt == 0 ? alpha[0] : alpha[1]) * a.v + a.d1;
            else return dw.cv * a.v + dw.cg0 * a.g0 + dw.cg1 * a.g1 + dw.ch00 * a.h00 + dw.ch01 * a.h01 + dw.ch11 * a.h11;
        };
        // weight of this row: upstream gradient x (reduced -> Pa); a singular adjoint poisons the result like the
        // per-row NaN of pcs_gc_jacobian
        const double wrow = !live ? 0.0 : MODE == 0 ? (ok ? (gout ? gout[i] : 1.0) * (T * P_UNIT) : nanv) : 1.0;

        // chain from a molecule-level sum (quantity q of molecule j, gq = weight x d functional / d sum) to the segment
        // parameters of that molecule's entries
        auto chain = [&](int q, int j, double gq, bool on_q) {
#pragma unroll 1
            for (int e = 0; e < GC_MAXE; e++) {
                const int cnt = row[16 + j * GC_MAXE + e];
                const bool on = on_q && cnt != 0;
                if (__ballot(on) == 0ull) continue;
                const int a = row[j * GC_MAXE + e];
                const double* p = tb.seg + 8 * a;
                double* ga = acc + 8 * a;
                const double gn = gq * cnt;
                if (q == Q_M) {
                    lds_add(ga + 0, gn, on);
                } else if (q <= Q_Z3) {
                    double d, ds, de;
                    diameter_grad(p, rT, d, ds, de);
                    const int k = q - Q_Z1 + 1;                                   // Z_k = sum n m d^k
                    const double dk1 = (k == 1) ? 1.0 : (k == 2 ? d : d * d);     // d^(k-1)
                    lds_add(ga + 0, gn * dk1 * d, on);
                    const double t = gn * p[0] * (k * dk1);
                    lds_add(ga + 1, t * ds, on);
                    lds_add(ga + 2, t * de, on);
                } else if (q == Q_S3) {
                    lds_add(ga + 0, gn * p[1] * p[1] * p[1], on);
                    lds_add(ga + 1, gn * p[0] * 3.0 * p[1] * p[1], on);
                } else if (q == Q_EK) {
                    lds_add(ga + 0, gn * p[2], on);
                    lds_add(ga + 2, gn * p[0], on);
                } else if (q == Q_MU) {
                    lds_add(ga + 3, gn * 2.0 * p[3], on);
                } else if (q == Q_SA) {
                    lds_add(ga + 1, gn * sgn_d(p[4] * p[5]), on);
                } else if (q == Q_EA) {
                    lds_add(ga + 2, gn * sgn_d(p[4] * p[5]), on);
                } else {
                    lds_add(ga + (q - Q_KA + 4), gn, on);  // kappa_ab, epsilon_k_ab, na, nb: plain sums
                }
            }
        };
        // ---- (1) molecule-level sums -------------------------------------------------------------------------------
        if constexpr (MODE == 0 && 1) {
            // coefficient adjoints of both phases (closed form), then their chain to the 26 sums
            double* adj = reinterpret_cast<double*>(gbonds) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < ADJ_SLOTS; k++) adj[k * BLOCK] = 0.0;
#pragma unroll 1
            for (int pt = 0; pt < 2; pt++)
                gc_a_adjoint(m.c, pt == 0 ? s0 : i0, pt == 0 ? s1 : i1, pt == 0 ? beta0[0] : beta0[1], pt == 0 ? beta1[0] : beta1[1],
                             pt == 0 ? alpha[0] : alpha[1], adj, BLOCK);
            double val[Q_COUNT * 2];
            gc_finish_gradient(ml, m.c.acls, m.c.polar, rT, adj, BLOCK, val);
            const bool polar = m.c.polar, assoc = m.c.acls != ASSOC_NONE;
#pragma unroll 1
            for (int q = 0; q < Q_COUNT; q++) {
                const bool need = live && ((q <= Q_Z3) || (q <= Q_MU ? polar : assoc));
                if (__ballot(need) == 0ull) continue;
                chain(q, 0, wrow * val[2 * q], need);
                chain(q, 1, wrow * val[2 * q + 1], need);
            }
        } else {
            // zero-tangent dual copy of the bond diameters (their derivative is part (3))
#pragma unroll 1
            for (int e = 0; e < 2 * GC_MAXE; e++) gbonds[threadIdx.x + e * BLOCK] = G(m.c.bond_dab[e * BLOCK]);
            const bool polar = m.c.polar, assoc = m.c.acls != ASSOC_NONE;
#pragma unroll 1
            for (int qq = 0; qq < Q_COUNT * (3 - CH); qq++) {
                const int q = CH == 2 ? qq : qq >> 1, jm = CH == 2 ? 0 : qq & 1;
                const bool need = (q <= Q_Z3) || (q <= Q_MU ? polar : assoc);
                if (__ballot(need) == 0ull) continue;  // the whole wave skips the pass
                GcCoef<G> c;
                c.bond_dab = gbonds + threadIdx.x;
                c.bond_cnt = m.c.bond_cnt;
                c.stride = BLOCK;
                gc_finish_tangent<G, CH>(c, ml, q, jm, ph0, ph1, rT);
                double val[2] = {0.0, 0.0};
#pragma unroll 1
                for (int pt = 0; pt < NPT; pt++) {
                    R a = gc_a_tangent<G, R>(c, seed0(pt, G(0.0)), seed1(pt, G(0.0)));
                    if (CH == 2) {
#pragma unroll
                        for (int j = 0; j < 2; j++) val[j] += dot_g(a, pt, j < CH ? j : 0);
                    } else {
                        const double v = dot_g(a, pt, 0);
                        if (jm == 0) val[0] += v; else val[1] += v;
                    }
                }
                // chain to the segment parameters of molecule j's entries
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (CH == 1 && j != jm) continue;
                    chain(q, j, wrow * val[j], need && live);
                }
            }
        }

        // ---- MODE 1 only: the per-row directions T, rho_0, rho_1 (T through the full coefficient set-up) -------------
        if constexpr (MODE == 1) {
            double* g9 = A_.jac9 + 9 * i;
            GcCoef<G> c;
            c.bond_dab = gbonds + threadIdx.x;
            c.bond_cnt = m.c.bond_cnt;  // counts are rewritten identically
            c.stride = BLOCK;
#pragma unroll 1
            for (int pass = 0; pass < 3; pass++) {  // T (through the full coefficient set-up), rho_0, rho_1
                if (pass < 2) {
                    G gT(T);
                    if (pass == 0) gT.e[0] = 1.0;
                    gc_coef<G>(c, row, tb, ph0, ph1, gT);
                }
                G q0(s0), q1(s1);
                if (pass == 1) q0.e[0] = 1.0;
                if (pass == 2) q1.e[0] = 1.0;
                const G one(1.0), nul(0.0);
                R a = gc_a_tangent<G, R>(c, R(q0, one, nul, nul, nul, nul), R(q1, nul, one, nul, nul, nul));
                const double v = deriv_contract(dw, a, 0);
                if (!live) continue;
                if (pass == 0) g9[6] = v;
                else if (pass == 1) g9[7] = v + dw.x0;
                else g9[8] = v + dw.x1;
            }
        }

        // ---- (2) dispersion double sums and (3) bond diameters: closed forms in the probe type over double ----------
        double gA[3] = {0.0, 0.0, 0.0}, gB[3] = {0.0, 0.0, 0.0};
        Q1 pcc[NPT], pz3m1[NPT], pomz[NPT], pr[NPT][2];
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) {
            const Q1 r0 = seed0(pt, 0.0), r1 = seed1(pt, 0.0);
            Q1 F1, F2;
            dispersion_factors(m.c, r0, r1, F1, F2);
            const Q1 q[3] = {r0 * r0, r0 * r1, r1 * r1};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                gA[k] += dot_q(F1 * q[k], pt);
                gB[k] += dot_q(F2 * q[k], pt);
            }
            const Q1 zeta2 = r0 * m.c.zk[2][0] + r1 * m.c.zk[2][1];
            const Q1 zeta3 = r0 * m.c.zk[3][0] + r1 * m.c.zk[3][1];
            pomz[pt] = 1.0 - zeta3;
            pz3m1[pt] = d_recip(pomz[pt]);
            pcc[pt] = zeta2 * (pz3m1[pt] * pz3m1[pt]);
            pr[pt][0] = r0;
            pr[pt][1] = r1;
        }
        if (MODE == 1 && live) {
            double* g9 = A_.jac9 + 9 * i;
#pragma unroll
            for (int k = 0; k < 3; k++) { g9[k] = gA[k]; g9[3 + k] = gB[k]; }
            if (A_.agg) {
#pragma unroll
                for (int k = 0; k < 3; k++) { A_.agg[6 * i + k] = m.c.A[k]; A_.agg[6 * i + 3 + k] = m.c.B[k]; }
            }
        }
        {
            // A[pr] = rT (phi-factor) s1[pr], B[pr] = rT^2 (phi-factor)^2 s2[pr]  (gc_finish)
            const double p01 = sqrt(ph0 * ph1);
            const double cA[3] = {rT * ph0, rT * 2.0 * p01, rT * ph1};
            const double cB[3] = {rT * rT * ph0 * ph0, rT * rT * 2.0 * ph0 * ph1, rT * rT * ph1 * ph1};
#pragma unroll
            for (int pq = 0; pq < 3; pq++) {
                const int mi = (pq == 2) ? 1 : 0, mj = (pq == 0) ? 0 : 1;
                const double gs1 = wrow * gA[pq] * cA[pq], gs2 = wrow * gB[pq] * cB[pq];
                // two sweeps so that each table entry of the row receives ONE accumulated term per parameter (the sums over
                // the partner entries stay in registers): side 0 = entries of molecule mi (partner mj), side 1 the reverse
#pragma unroll 1
                for (int side = 0; side < 2; side++) {
                    const int mo = side == 0 ? mi : mj, mp = side == 0 ? mj : mi;  // own / partner molecule
#pragma unroll 1
                    for (int e = 0; e < GC_MAXE; e++) {
                        const int ne = row[16 + mo * GC_MAXE + e];
                        const bool on = live && ne != 0;
                        if (__ballot(on) == 0ull) continue;
                        const int a = row[mo * GC_MAXE + e];
                        const double* pa = tb.seg + 8 * a;
                        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll 1
                        for (int f = 0; f < GC_MAXE; f++) {
                            const int nf = row[16 + mp * GC_MAXE + f];
                            if (nf == 0) continue;
                            const int b = row[mp * GC_MAXE + f];
                            const double* pb = tb.seg + 8 * b;
                            const double mb = nf * pb[0];
                            double k1 = 1.0;
                            if (mi != mj) k1 = tb.K[a * tb.S + b];  // E1, E2, K are symmetric tables
                            const double t1 = tb.E1[a * tb.S + b] * k1, t2 = tb.E2[a * tb.S + b] * (k1 * k1);
                            const double common = gs1 * t1 + gs2 * t2;
                            g0 += mb * common;
                            const double sab = 0.5 * (pa[1] + pb[1]);
                            g1 += mb * common * (1.5 / sab);  // d sigma_ab^3 / d sigma_a = 1.5 sigma_ab^2
                            // E1 = sqrt(eps_a eps_b) sigma_ab^3: d/d eps_a = E1 / (2 eps_a); the square root is not
                            // differentiable at eps_a = 0 (segment '>C<'): that term is left out (the reference's autograd
                            // returns NaN for every epsilon_k there).  E2 = eps_a eps_b sigma_ab^3.
                            const double s3k2 = sab * sab * sab * (k1 * k1);
                            g2 += mb * ((pa[2] != 0.0 ? gs1 * t1 * (0.5 / pa[2]) : 0.0) + gs2 * pb[2] * s3k2);
                        }
                        const double ma = ne * pa[0];
                        lds_add(acc + 8 * a + 0, ne * g0, on);
                        lds_add(acc + 8 * a + 1, ma * g1, on);
                        lds_add(acc + 8 * a + 2, ma * g2, on);
                    }
                }
            }
        }
        // bonds (:156-165): a_hc = -sum rho_i n ln g(c d_ab), g = 1/(1-z3) + 3 c d_ab + 2 (c d_ab)^2 (1 - z3)
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
#pragma unroll 1
            for (int e = 0; e < GC_MAXE; e++) {
                const int slot = (mi * GC_MAXE + e) * BLOCK;
                const double cnt = (double)m.c.bond_cnt[mi * GC_MAXE + e];
                const bool on = live && cnt != 0.0;
                if (__ballot(on) == 0ull) continue;
                const double dab = m.c.bond_dab[slot];
                double gd = 0.0;
#pragma unroll
                for (int pt = 0; pt < NPT; pt++) {
                    const Q1 cd = pcc[pt] * dab;
                    const Q1 g = pz3m1[pt] + 3.0 * cd + 2.0 * ((cd * cd) * pomz[pt]);
                    const Q1 x = (pr[pt][mi] * (-cnt)) * ((pcc[pt] * (3.0 + 4.0 * (cd * pomz[pt]))) * d_recip(g));
                    gd += dot_q(x, pt);
                }
                gd *= wrow;
                const int a = row[32 + mi * GC_MAXE + e], b = row[48 + mi * GC_MAXE + e];
                double da, das, dae, db, dbs, dbe;
                diameter_grad(tb.seg + 8 * a, rT, da, das, dae);
                diameter_grad(tb.seg + 8 * b, rT, db, dbs, dbe);
                const double rs = 1.0 / (da + db);
                const double wa = gd * (db * rs) * (db * rs), wb = gd * (da * rs) * (da * rs);  // d d_ab / d d_a, d d_b
                lds_add(acc + 8 * a + 1, wa * das, on);
                lds_add(acc + 8 * a + 2, wa * dae, on);
                lds_add(acc + 8 * b + 1, wb * dbs, on);
                lds_add(acc + 8 * b + 2, wb * dbe, on);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < S * 8; k += BLOCK) {
        const double v = acc[k];
        if (v != 0.0) unsafeAtomicAdd(grad_seg + k, v);
    }
}

template <int MODE, int BLOCK>
static int launch_gc_gradient_block(const GcGradArgs& a, size_t lds, void* stream, const char* what) {
    const int64_t tiles = (a.n + BLOCK - 1) / BLOCK;
    const int64_t cap = (int64_t)GS_GRID * GSBLOCK / BLOCK;
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    if (lds > 64 * 1024) {  // above the default dynamic-LDS limit (the CU has 160 KB)
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gc_segment_gradient<MODE, BLOCK>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return fail(what, ea);
    }
    hipLaunchKernelGGL((k_gc_segment_gradient<MODE, BLOCK>), dim3(grid), dim3(BLOCK), lds, as_stream(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(what, e);
    return 0;
}

template <int MODE>
static int launch_gc_gradient_mode(const GcGradArgs& a, void* stream, const char* what) {
    auto lds_of = [&](int block) { return gc_lds_bytes(a.S, block, GsPerThread<MODE>::value + GC_ROW_LDS_DOUBLES) + sizeof(double) * a.S * 8; };
    constexpr size_t LDS_CU = 160 * 1024;
    if (lds_of(256) <= LDS_CU) return launch_gc_gradient_block<MODE, 256>(a, lds_of(256), stream, what);
    if (lds_of(192) <= LDS_CU) return launch_gc_gradient_block<MODE, 192>(a, lds_of(192), stream, what);
    return launch_gc_gradient_block<MODE, GSBLOCK>(a, lds_of(GSBLOCK), stream, what);
}
}  // namespace

extern "C" {

// LDS of a workgroup: table + gradient accumulator + per thread: bond diameters 2*MAXE doubles, the dual model's diameters
// 2*MAXE*(1+CHUNK) / the coefficient adjoints, the row bytes.  The kernel holds a full register file per wave and waits mostly
// for its own stack frame, so the number of resident waves is what counts (measured: one wave per CU 6.6 ms, two 3.8 ms per 1e6
// rows): 64-thread workgroups carry one copy of the table each and fit twice (S = 22: 59 KB); a 192-thread workgroup shares one
// copy among three waves (S = 22: 149 KB; the VJP of `derivatives`, MODE 1, needs less per thread: four waves, 132 KB) -- the
// largest workgroup that fits the CU's 160 KB is taken.
static int launch_gc_gradient(int mode, const GcGradArgs& a, void* stream, const char* what) {
    return mode == 0 ? launch_gc_gradient_mode<0>(a, stream, what) : launch_gc_gradient_mode<1>(a, stream, what);
}

int pcs_gc_segment_gradient(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                            const double* rho4, int64_t n, const double* gout, double* grad_seg, const int32_t* order,
                            void* stream) {
    g_err[0] = 0;
    if (int e = gc_check(S, n)) return e;
    if (n == 0) return 0;
    if (!table || !rows || !phi || !temp || !rho4 || !grad_seg) return fail_msg("pcs_gc_segment_gradient: null required pointer");
    if (reinterpret_cast<uintptr_t>(rows) & 15) return fail_msg("pcs_gc_segment_gradient: rows must be 16-byte aligned");
    GcGradArgs a{};
    a.dew = dew; a.table = table; a.S = S; a.rows = rows; a.phi = phi; a.temp = temp; a.rho = rho4; a.n = n;
    a.gout = gout; a.grad_seg = grad_seg; a.order = order;
    return launch_gc_gradient(0, a, stream, "k_gc_segment_gradient launch");
}

int pcs_gc_derivatives_vjp(const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                           const double* rho, int64_t n, const double* g_a, const double* g_p, const double* g_mu,
                           const double* g_v, double* grad_seg, double* jac9, double* agg, const int32_t* order, void* stream) {
    g_err[0] = 0;
    if (int e = gc_check(S, n)) return e;
    if (n == 0) return 0;
    if (!table || !rows || !phi || !temp || !rho || !grad_seg || !jac9) return fail_msg("pcs_gc_derivatives_vjp: null required pointer");
    if (reinterpret_cast<uintptr_t>(rows) & 15) return fail_msg("pcs_gc_derivatives_vjp: rows must be 16-byte aligned");
    GcGradArgs a{};
    a.table = table; a.S = S; a.rows = rows; a.phi = phi; a.temp = temp; a.rho = rho; a.n = n;
    a.g_a = g_a; a.g_p = g_p; a.g_mu = g_mu; a.g_v = g_v; a.grad_seg = grad_seg; a.jac9 = jac9; a.agg = agg; a.order = order;
    return launch_gc_gradient(1, a, stream, "k_gc_derivatives_vjp launch");
}

}  // extern "C"
