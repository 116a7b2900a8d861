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
// Only the table gradient is produced here; k_ab, phi and T are pcs_gc_jacobian's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcsaft_hip.h"
#include "abi_common.hpp"
#include "gc_model.hpp"
#include "gc_kernel_common.hpp"
#include "mix_solver.hpp"

namespace {

constexpr int GSBLOCK = 64;
constexpr int GS_GRID = 2048;
constexpr int GS_CHUNK = 2;  // slot j of a pass = molecule j

enum : int { Q_M, Q_Z1, Q_Z2, Q_Z3, Q_S3, Q_EK, Q_MU, Q_SA, Q_EA, Q_KA, Q_EAB, Q_NA, Q_NB, Q_COUNT };

__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// d and its derivatives w.r.t. sigma and epsilon_k of the segment (:118-120)
__device__ __forceinline__ void diameter_grad(const double* seg, double rT, double& d, double& d_sig, double& d_eps) {
    const double ex = exp(-3.0 * seg[2] * rT);
    const double f = 1.0 - 0.12 * ex;
    d = seg[1] * f;
    d_sig = f;
    d_eps = seg[1] * (0.36 * rT) * ex;
}

// coefficients with the tangents of molecule-level quantity q (slot j: molecule j); out of line as mix_coef_tangent
template <class G>
__device__ __attribute__((noinline)) void gc_finish_tangent(GcCoef<G>& c, const GcMol<double, double>& ml, int q, double phi0,
                                                            double phi1, double rT) {
    GcMol<G, G> g;
#define PCS_SEED(field, code)                                   \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {             \
        g.field[i] = G(ml.field[i]);                            \
        if (q == code) g.field[i].e[i] = 1.0;                   \
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
            if (q == Q_Z1 + k) g.Zk[k][i].e[i] = 1.0;
        }
        g.s1[k] = G(ml.s1[k]);
        g.s2[k] = G(ml.s2[k]);
    }
    gc_finish<G, G>(c, g, phi0, phi1, G(rT));
}

__global__ __launch_bounds__(GSBLOCK) void k_gc_segment_gradient(int dew, const double* __restrict__ table, int S,
                                                                 const unsigned char* __restrict__ rows,
                                                                 const double* __restrict__ phi,
                                                                 const double* __restrict__ temp,
                                                                 const double* __restrict__ rho4, int64_t n,
                                                                 const double* __restrict__ gout,
                                                                 double* __restrict__ grad_seg,
                                                                 const int32_t* __restrict__ order) {
    typedef DN<double, GS_CHUNK> G;
    typedef D1<G> R;
    typedef D1<double> Q1;
    extern __shared__ double lds[];
    GcTable tb = stage_table(table, S, lds);
    double* acc = lds + gc_table_doubles(S);                          // [S][8] gradient of this workgroup
    double* bonds = acc + S * 8;                                      // double model: [2*MAXE dab][2*MAXE cnt] x block
    G* gbonds = reinterpret_cast<G*>(bonds + 4 * GC_MAXE * GSBLOCK);  // dual model dab (zero tangents)
    for (int k = threadIdx.x; k < S * 8; k += GSBLOCK) acc[k] = 0.0;
    __syncthreads();

    GcModelT<double> m;
    m.c.bond_dab = bonds + threadIdx.x;
    m.c.bond_cnt = bonds + 2 * GC_MAXE * GSBLOCK + threadIdx.x;
    m.c.stride = GSBLOCK;
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);

    const int64_t tiles = (n + GSBLOCK - 1) / GSBLOCK;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        int64_t i = tile * GSBLOCK + threadIdx.x;
        if (i >= n) continue;
        if (order) {
            i = order[i];
            if (i < 0 || i >= n) continue;
        }
        const unsigned char* row = rows + (size_t)i * GC_ROW_BYTES;
        const double T = temp[i], ph0 = phi[2 * i], ph1 = phi[2 * i + 1];
        const double rT = 1.0 / T;
        const double4 r4 = reinterpret_cast<const double4*>(rho4)[i];  // (V0, V1, L0, L1)
        const double s0 = dew ? r4.x : r4.z, s1 = dew ? r4.y : r4.w, i0 = dew ? r4.z : r4.x, i1 = dew ? r4.w : r4.y;

        GcMol<double, double> ml;
        gc_mol<double>(ml, m.c.bond_dab, m.c.bond_cnt, m.c.stride, row, tb, rT);
        gc_finish<double, double>(m.c, ml, ph0, ph1, rT);

        // adjoint of the converged state (mix_jacobian.hpp): J^T w = dp^vap/du
        double alpha[2], beta0[2], beta1[2];
        bool ok;
        {
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
        }
        // weight of this row: upstream gradient x (reduced -> Pa); a singular adjoint poisons the result like the
        // per-row NaN of pcs_gc_jacobian
        const double wrow = ok ? (gout ? gout[i] : 1.0) * (T * P_UNIT) : nanv;

        // ---- (1) molecule-level sums: dual-number passes -------------------------------------------------------
        {
            // zero-tangent dual copy of the bond diameters (their derivative is part (3))
#pragma unroll 1
            for (int e = 0; e < 2 * GC_MAXE; e++) gbonds[threadIdx.x + e * GSBLOCK] = G(m.c.bond_dab[e * GSBLOCK]);
            const bool polar = m.c.polar, assoc = m.c.acls != ASSOC_NONE;
#pragma unroll 1
            for (int q = 0; q < Q_COUNT; q++) {
                const bool need = (q <= Q_Z3) || (q <= Q_MU ? polar : assoc);
                if (__ballot(need) == 0ull) continue;  // the whole wave skips the pass
                GcCoef<G> c;
                c.bond_dab = gbonds + threadIdx.x;
                c.bond_cnt = m.c.bond_cnt;
                c.stride = GSBLOCK;
                gc_finish_tangent<G>(c, ml, q, ph0, ph1, rT);
                double val[2] = {0.0, 0.0};
#pragma unroll 1
                for (int ph = 0; ph < 2; ph++) {
                    const double q0 = ph == 0 ? s0 : i0, q1 = ph == 0 ? s1 : i1;
                    const double al = ph == 0 ? alpha[0] : alpha[1], b0 = ph == 0 ? beta0[0] : beta0[1], b1 = ph == 0 ? beta1[0] : beta1[1];
                    R a = gc_a_tangent<G, R>(c, R(G(q0), G(b0)), R(G(q1), G(b1)));
#pragma unroll
                    for (int j = 0; j < 2; j++) val[j] += al * a.v.e[j] + a.d1.e[j];
                }
                if (!need) continue;
                // chain to the segment parameters of molecule j's entries
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const double gq = wrow * val[j];
#pragma unroll 1
                    for (int e = 0; e < GC_MAXE; e++) {
                        const int cnt = row[16 + j * GC_MAXE + e];
                        if (cnt == 0) continue;
                        const int a = row[j * GC_MAXE + e];
                        const double* p = tb.seg + 8 * a;
                        double* ga = acc + 8 * a;
                        const double gn = gq * cnt;
                        if (q == Q_M) {
                            lds_add(ga + 0, gn);
                        } else if (q <= Q_Z3) {
                            double d, ds, de;
                            diameter_grad(p, rT, d, ds, de);
                            const int k = q - Q_Z1 + 1;                                   // Z_k = sum n m d^k
                            const double dk1 = (k == 1) ? 1.0 : (k == 2 ? d : d * d);     // d^(k-1)
                            lds_add(ga + 0, gn * dk1 * d);
                            const double t = gn * p[0] * (k * dk1);
                            lds_add(ga + 1, t * ds);
                            lds_add(ga + 2, t * de);
                        } else if (q == Q_S3) {
                            lds_add(ga + 0, gn * p[1] * p[1] * p[1]);
                            lds_add(ga + 1, gn * p[0] * 3.0 * p[1] * p[1]);
                        } else if (q == Q_EK) {
                            lds_add(ga + 0, gn * p[2]);
                            lds_add(ga + 2, gn * p[0]);
                        } else if (q == Q_MU) {
                            lds_add(ga + 3, gn * 2.0 * p[3]);
                        } else if (q == Q_SA) {
                            lds_add(ga + 1, gn * sgn_d(p[4] * p[5]));
                        } else if (q == Q_EA) {
                            lds_add(ga + 2, gn * sgn_d(p[4] * p[5]));
                        } else {
                            lds_add(ga + (q - Q_KA + 4), gn);  // kappa_ab, epsilon_k_ab, na, nb: plain sums
                        }
                    }
                }
            }
        }

        // ---- (2) dispersion double sums and (3) bond diameters: closed forms in D1<double> along beta ------------
        double gA[3] = {0.0, 0.0, 0.0}, gB[3] = {0.0, 0.0, 0.0};
        Q1 pcc[2], pz3m1[2], pomz[2], pr[2][2];
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {
            const double al = alpha[ph];
            const Q1 r0(ph == 0 ? s0 : i0, beta0[ph]), r1(ph == 0 ? s1 : i1, beta1[ph]);
            Q1 F1, F2;
            dispersion_factors(m.c, r0, r1, F1, F2);
            const Q1 q[3] = {r0 * r0, r0 * r1, r1 * r1};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                Q1 t = F1 * q[k];
                gA[k] += al * t.v + t.d1;
                t = F2 * q[k];
                gB[k] += al * t.v + t.d1;
            }
            const Q1 zeta2 = r0 * m.c.zk[2][0] + r1 * m.c.zk[2][1];
            const Q1 zeta3 = r0 * m.c.zk[3][0] + r1 * m.c.zk[3][1];
            pomz[ph] = 1.0 - zeta3;
            pz3m1[ph] = d_recip(pomz[ph]);
            pcc[ph] = zeta2 * (pz3m1[ph] * pz3m1[ph]);
            pr[ph][0] = r0;
            pr[ph][1] = r1;
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
#pragma unroll 1
                for (int e = 0; e < GC_MAXE; e++) {
                    const int ne = row[16 + mi * GC_MAXE + e];
                    if (ne == 0) continue;
                    const int a = row[mi * GC_MAXE + e];
                    const double* pa = tb.seg + 8 * a;
                    const double ma = ne * pa[0];
#pragma unroll 1
                    for (int f = 0; f < GC_MAXE; f++) {
                        const int nf = row[16 + mj * GC_MAXE + f];
                        if (nf == 0) continue;
                        const int b = row[mj * GC_MAXE + f];
                        const double* pb = tb.seg + 8 * b;
                        const double mb = nf * pb[0];
                        double k1 = 1.0;
                        if (mi != mj) k1 = tb.K[a * tb.S + b];
                        const double t1 = tb.E1[a * tb.S + b] * k1, t2 = tb.E2[a * tb.S + b] * (k1 * k1);
                        const double common = gs1 * t1 + gs2 * t2;
                        lds_add(acc + 8 * a + 0, (ne * mb) * common);
                        lds_add(acc + 8 * b + 0, (ma * nf) * common);
                        const double mm = ma * mb;
                        const double sab = 0.5 * (pa[1] + pb[1]);
                        const double dsg = mm * common * (1.5 / sab);  // d sigma_ab^3 / d sigma_a = 1.5 sigma_ab^2
                        lds_add(acc + 8 * a + 1, dsg);
                        lds_add(acc + 8 * b + 1, dsg);
                        // E1 = sqrt(eps_a eps_b) sigma_ab^3: d/d eps_a = E1 / (2 eps_a); the square root is not
                        // differentiable at eps_a = 0 (segment '>C<'): that term is left out (the reference's autograd
                        // returns NaN for every epsilon_k there).  E2 = eps_a eps_b sigma_ab^3.
                        const double s3k2 = sab * sab * sab * (k1 * k1);
                        lds_add(acc + 8 * a + 2, mm * ((pa[2] != 0.0 ? gs1 * t1 * (0.5 / pa[2]) : 0.0) + gs2 * pb[2] * s3k2));
                        lds_add(acc + 8 * b + 2, mm * ((pb[2] != 0.0 ? gs1 * t1 * (0.5 / pb[2]) : 0.0) + gs2 * pa[2] * s3k2));
                    }
                }
            }
        }
        // bonds (:156-165): a_hc = -sum rho_i n ln g(c d_ab), g = 1/(1-z3) + 3 c d_ab + 2 (c d_ab)^2 (1 - z3)
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
#pragma unroll 1
            for (int e = 0; e < GC_MAXE; e++) {
                const int slot = (mi * GC_MAXE + e) * GSBLOCK;
                const double cnt = m.c.bond_cnt[slot];
                if (cnt == 0.0) continue;
                const double dab = m.c.bond_dab[slot];
                double gd = 0.0;
#pragma unroll
                for (int ph = 0; ph < 2; ph++) {
                    const Q1 cd = pcc[ph] * dab;
                    const Q1 g = pz3m1[ph] + 3.0 * cd + 2.0 * ((cd * cd) * pomz[ph]);
                    const Q1 x = (pr[ph][mi] * (-cnt)) * ((pcc[ph] * (3.0 + 4.0 * (cd * pomz[ph]))) * d_recip(g));
                    gd += alpha[ph] * x.v + x.d1;
                }
                gd *= wrow;
                const int a = row[32 + mi * GC_MAXE + e], b = row[48 + mi * GC_MAXE + e];
                double da, das, dae, db, dbs, dbe;
                diameter_grad(tb.seg + 8 * a, rT, da, das, dae);
                diameter_grad(tb.seg + 8 * b, rT, db, dbs, dbe);
                const double rs = 1.0 / (da + db);
                const double wa = gd * (db * rs) * (db * rs), wb = gd * (da * rs) * (da * rs);  // d d_ab / d d_a, d d_b
                lds_add(acc + 8 * a + 1, wa * das);
                lds_add(acc + 8 * a + 2, wa * dae);
                lds_add(acc + 8 * b + 1, wb * dbs);
                lds_add(acc + 8 * b + 2, wb * dbe);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < S * 8; k += GSBLOCK) {
        const double v = acc[k];
        if (v != 0.0) unsafeAtomicAdd(grad_seg + k, v);
    }
}

}  // namespace

extern "C" {

int pcs_gc_segment_gradient(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                            const double* rho4, int64_t n, const double* gout, double* grad_seg, const int32_t* order,
                            void* stream) {
    g_err[0] = 0;
    if (int e = gc_check(S, n)) return e;
    if (n == 0) return 0;
    if (!table || !rows || !phi || !temp || !rho4 || !grad_seg) return fail_msg("pcs_gc_segment_gradient: null required pointer");
    const int64_t tiles = (n + GSBLOCK - 1) / GSBLOCK;
    const unsigned grid = (unsigned)(tiles < GS_GRID ? tiles : GS_GRID);
    // table + gradient accumulator + per thread: double model 4*MAXE doubles, dual model dab 2*MAXE*(1+CHUNK)
    const size_t lds = gc_lds_bytes(S, GSBLOCK, 4 * GC_MAXE + 2 * GC_MAXE * (1 + GS_CHUNK)) + sizeof(double) * S * 8;
    hipLaunchKernelGGL(k_gc_segment_gradient, dim3(grid), dim3(GSBLOCK), lds, as_stream(stream), dew, table, S, rows, phi,
                       temp, rho4, n, gout, grad_seg, order);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_gc_segment_gradient launch", e);
    return 0;
}

}  // extern "C"
