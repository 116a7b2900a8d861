/*
 * pcsaft_hip.h — C ABI of libpcsaft_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the native half of feos-torch.  Each entry point replaces one method
 * of the reference's PyO3 classes (module `feos_torch.feos_torch`, src/lib.rs:10-16) together
 * with the per-row feos solve behind it, and additionally returns the property the reference's
 * Python tail computes from the converged densities (the kernels produce it in the same pass).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch tensor .data_ptr()), fp64 unless
 *     noted, C-contiguous; the caller owns all memory; nothing is allocated or freed here;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     the call returns without synchronising;
 *   - outputs are DENSE (length n): status[i] = 1 marks a row the solver could not converge
 *     (the reference drops such rows, src/pcsaft.rs:93-95; compaction is left to the caller);
 *   - optional outputs may be NULL;
 *   - return value 0 = enqueued, non-zero = API error, message via pcs_last_error();
 *   - parameter row layout (README.md:12): m, sigma[A], epsilon_k[K], mu[D], kappa_ab,
 *     epsilon_k_ab[K], na, nb.
 */
#ifndef PCSAFT_HIP_H
#define PCSAFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library / ABI version (major*100 + minor). */
int pcs_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* pcs_last_error(void);

/* Bytes of device scratch a call on n rows needs (retry list of the pure / gc robust pass, or row order +
 * control block of the mixture work queue). */
int64_t pcs_workspace_bytes(int64_t n);

/*
 * Pure-component VLE at fixed temperature.
 * Replaces PcSaft.vapor_pressure(parameters[N,8], temperature[N]) -> (rho[n_ok,4], status[N])
 * (src/pcsaft.rs:18-26, :82-103) and the Python tails of PcSaftPure.vapor_pressure
 * (feos_torch/pcsaft_pure.py:201-215) and .equilibrium_liquid_density (:217-233).
 *   params   [n,8]   in
 *   temp     [n]     in   K
 *   p_sat    [n]     out  Pa            (optional)  pcsaft_pure.py:214-215
 *   rho_eq   [n]     out  kmol/m3       (optional)  pcsaft_pure.py:232-233
 *   rho_vl   [n,2]   out  A^-3: (rho_V, rho_L) at which the formulas were evaluated (optional;
 *                         the reference's rho[:,0:2] of src/pcsaft.rs:99-100)
 *   status   [n]     out  uint8, 1 = failed
 *   iters    [n]     out  int32, Newton iterations used (optional, diagnostics)
 *   workspace        device scratch of pcs_workspace_bytes(n)
 * Kernels: pressure only -- fp32 pre-solve + fp64 finish with the pre-solve's dp/drho; rho_vl -- the all-fp64 iteration from
 * the pre-solve's start (densities ~1e-11); rho_eq -- the pressure-only path + one exact fp64 Newton update of both densities
 * (~1e-14; p_sat then carries that step's second-order term).
 */
int pcs_pure_vle(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                 double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream);

/*
 * The two stages of pcs_pure_vle as separate launches, same arguments (for per-kernel timing and
 * for callers that want to overlap the rare-row robust pass with other work):
 *   pcs_pure_vle_fast   zeroes the retry counter, runs the fast kernel on all n rows and appends
 *                       rows that need the robust initialisation to the workspace list;
 *   pcs_pure_vle_retry  runs the robust kernel over that list (count read on the device).
 * pcs_pure_vle == pcs_pure_vle_fast followed by pcs_pure_vle_retry on the same stream.
 */
int pcs_pure_vle_fast(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                      double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream);
int pcs_pure_vle_retry(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                       double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream);

/*
 * pcs_pure_vle with the all-fp64 main kernel (fp64 value / first / second derivative in every Newton iteration after the
 * fp32 start) whatever the outputs -- the validation twin of pcs_pure_vle: same arguments, same results to ~1e-11.
 */
int pcs_pure_vle_fp64(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_eq,
                      double* rho_vl, uint8_t* status, int32_t* iters, void* workspace, void* stream);

/*
 * PcSaftPure.vapor_pressure as ONE call (feos_torch/pcsaft_pure.py:201-215 incl. its Rust solve, src/pcsaft.rs:82-103):
 * always the pressure-only kernel of pcs_pure_vle (fp32 pre-solve, fp64 finish), so p_sat carries the same bits whether
 * or not the caller also asks for the densities.
 *   rho_vl [n,2] out (optional) the densities the last fp64 Newton step of that kernel ended on: ~1e-9 (relative) from the
 *                root -- what the Jacobian kernel needs; p_sat is second order in that error.  For converged densities
 *                (the reference's rho output) call pcs_pure_vle with rho_vl.
 */
int pcs_pure_vapor_pressure(const double* params, const double* temp, int64_t n, double* p_sat, double* rho_vl,
                            uint8_t* status, void* workspace, void* stream);

/*
 * Liquid density at given (T, p).
 * Replaces PcSaft.liquid_density(parameters[N,8], temperature[N], pressure[N]) ->
 * (rho[n_ok], status[N]) (src/pcsaft.rs:28-41, :105-129) and the tail of
 * PcSaftPure.liquid_density (feos_torch/pcsaft_pure.py:184-199).
 *   pressure [n]  in   Pa
 *   rho_out  [n]  out  kmol/m3  (optional)  pcsaft_pure.py:198-199
 *   rho_root [n]  out  A^-3 converged density (the reference's rho of src/pcsaft.rs:127) (optional)
 */
int pcs_pure_liquid_density(const double* params, const double* temp, const double* pressure, int64_t n,
                            double* rho_out, double* rho_root, uint8_t* status, void* stream);

/*
 * (a, p, dp/drho) at given (T, rho) — PcSaftPure.derivatives (feos_torch/pcsaft_pure.py:180-182).
 * All reduced (A^-3).  Any output may be NULL.
 */
int pcs_pure_derivatives(const double* params, const double* temp, const double* rho, int64_t n, double* a,
                         double* p, double* dp, void* stream);

/*
 * Jacobian of a pure-component property w.r.t. its inputs with the phase densities held
 * fixed — what torch reverse mode through the reference's Python tail yields
 * (feos_torch/pcsaft_pure.py:196-199 / :212-215 / :228-233).
 *   which    0 = vapor_pressure, 1 = liquid_density, 2 = equilibrium_liquid_density; 0 | PCS_JAC_POLISH: rho_vl comes
 *            from pcs_pure_vapor_pressure (~1e-9 from the root) and takes one fp64 Newton step before the derivatives
 *   pressure [n]    in  Pa (which = 1 only, else NULL)
 *   rho_vl   [n,2]  in  A^-3 (rho_V, rho_L) from pcs_pure_vle; for which = 1 column 1 holds
 *                       rho_root from pcs_pure_liquid_density, column 0 is ignored
 *   jac      [n,10] out d value / d (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb, T, p)
 */
#define PCS_JAC_POLISH 0x100
int pcs_pure_jacobian(int which, const double* params, const double* temp, const double* pressure,
                      const double* rho_vl, int64_t n, double* jac, void* stream);

/*
 * The same Jacobian in vector-Jacobian form -- the backward pass of a property call: grad_x[i] = gout[i] * d value_i / d x_i
 * written straight into the dense gradient arrays (no [n,10] round trip through memory).
 *   gout [n] in; grad_params [n,8] (16-byte aligned), grad_temp [n], grad_pressure [n] out, each optional.
 */
int pcs_pure_jacobian_vjp(int which, const double* params, const double* temp, const double* pressure, const double* rho_vl,
                          const double* gout, int64_t n, double* grad_params, double* grad_temp, double* grad_pressure,
                          void* stream);

/*
 * Binary-mixture bubble point (dew = 0: z = liquid mole fraction of component 1) or dew point
 * (dew = 1: z = vapour mole fraction of component 1) at fixed temperature.
 * Replaces PcSaft.bubble_point / PcSaft.dew_point(parameters[N,2,8], kij[N,2], temperature[N],
 * molefracs[N], pressure[N]) -> (rho[n_ok,4], status[N])  (src/pcsaft.rs:43-79, :150-231) and
 * the Python tails PcSaftMix.bubble_point / dew_point (feos_torch/pcsaft_mix.py:422-468).
 *   params  [n,2,8] in    component rows as for the pure model
 *   kij     [n,2]   in    kij[:,0] = k_ij, kij[:,1] = explicit eps_AiBj/k, 0 = combining rule
 *                         (src/pcsaft.rs:163, pcsaft_mix.py:509-516)
 *   temp, z, p_init [n] in    K, -, Pa (p_init = the caller's starting pressure, src/pcsaft.rs:174)
 *   p_out   [n]     out   Pa   pcsaft_mix.py:443-444 / :467-468                 (optional)
 *   rho4    [n,4]   out   A^-3 (rhoV_1, rhoV_2, rhoL_1, rhoL_2), src/pcsaft.rs:225-228 (optional)
 *   status  [n]     out   uint8, 1 = failed
 *   iters   [n]     out   int32 Newton iterations (optional)
 *   workspace       device scratch of pcs_mix_workspace_bytes(n) for the work-queue schedule (rows ordered by
 *                   class on the device, a pre-pass with the pure-component fugacities of Raoult's law, persistent
 *                   waves, a lane that finishes a row takes the next one);
 *                   NULL = one row per lane in a single pass (same results, several times slower)
 */
int64_t pcs_mix_workspace_bytes(int64_t n);
int pcs_mix_bubble_dew(int dew, const double* params, const double* kij, const double* temp, const double* z,
                       const double* p_init, int64_t n, double* p_out, double* rho4, uint8_t* status, int32_t* iters,
                       void* workspace, void* stream);

/*
 * PcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420) at given partial densities rho [n,2]:
 * a [n], p [n] (reduced), residual chemical potentials mu [n,2], partial molar volumes v [n,2].
 * Any output may be NULL.
 */
int pcs_mix_derivatives(const double* params, const double* kij, const double* temp, const double* rho, int64_t n,
                        double* a, double* p, double* mu, double* v, void* stream);

/*
 * n-component mixtures: PcSaftMix.derivatives for parameters [n, ncomp, 8] WITHOUT k_ij -- the parts of the
 * reference's model that are written for any number of components (feos_torch/pcsaft_mix.py:31-154: hard sphere, hard chain,
 * dispersion, dipoles, self association of ONE associating component; "kij can only be used for binary mixtures", :75-76, and two
 * associating components are binary-only, :250 / :336 -- both stay with pcs_mix_derivatives).
 *   params [n,ncomp,8], temp [n], rho [n,ncomp] in;  a [n], p [n], mu [n,ncomp], v [n,ncomp] out (each optional); 1 <= ncomp <= 6.
 * A row with more than one associating component gets NaN in every output (the reference raises).
 */
int pcs_mixn_derivatives(const double* params, const double* temp, const double* rho, int ncomp, int64_t n, double* a, double* p,
                         double* mu, double* v, void* stream);

/*
 * Backward pass of pcs_mixn_derivatives -- the reference's n-component model is an ordinary torch graph
 * (feos_torch/pcsaft_mix.py:31-154, :395-420), so its autograd reaches parameters, temperature and densities:
 *   grad [n, 9 ncomp + 1] = d L / d (params [ncomp][8], T, rho [ncomp]),  L = g_a a + g_p p + g_mu . mu + g_v . v
 * with the upstream gradients g_a [n], g_p [n], g_mu [n,ncomp], g_v [n,ncomp] (each optional, NULL = 0).  One forward-mode
 * pass per input direction through the same model code (correct first, not tuned: 9 ncomp + 1 evaluations per row).
 */
int pcs_mixn_derivatives_vjp(const double* params, const double* temp, const double* rho, int ncomp, int64_t n, const double* g_a,
                             const double* g_p, const double* g_mu, const double* g_v, double* grad, void* stream);

/*
 * Gradient of the bubble (dew = 0) / dew (dew = 1) pressure [Pa] at the converged densities rho4
 * (from pcs_mix_bubble_dew) — what torch reverse mode through feos_torch/pcsaft_mix.py:435-444 /
 * :459-468 yields.  jac [n,19] = d p / d (params[0,0..7], params[1,0..7], kij[0], kij[1], T).
 * workspace: device scratch of pcs_workspace_bytes(n) or NULL.  With it the rows are processed in batch-wide class
 * order (waves of non-polar / non-associating rows skip the structurally-zero directions); results are identical.
 */
int pcs_mix_jacobian(int dew, const double* params, const double* kij, const double* temp, const double* rho4,
                     int64_t n, double* jac, void* workspace, void* stream);

/*
 * ---- heterosegmented gc-PC-SAFT (binary mixtures) -----------------------------------------
 * Replaces the reference's stateful GcPcSaft class (src/gc_pcsaft.rs:15-99: segment records,
 * per-row chemical records, binary segment records, phi) and the Python tails of
 * GcPcSaftMix.bubble_point / dew_point (feos_torch/gc_pcsaft.py:470-512).
 *
 *   table  [S*8 + 3*S*S] in   per-batch: seg[S][8] (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab,
 *                             na, nb), E1[S][S] = sqrt(eps_a eps_b) sigma_ab^3, E2[S][S] = eps_a eps_b
 *                             sigma_ab^3, K[S][S] = 1 - k_ab, sigma_ab = (sigma_a + sigma_b)/2; S <= 32
 *   rows   [n,80] uint8  in   (16-byte aligned: the kernels take a row with five 16-byte loads)
 *                             molecule structures: for each of the 2 molecules up to 8 entries of
 *                             seg_id [0:16], seg_cnt [16:32], bond_a [32:48], bond_b [48:64],
 *                             bond_cnt [64:80] (count 0 = unused entry)
 *   phi    [n,2]         in   src/gc_pcsaft.rs:30, feos_torch/gc_pcsaft.py:182-185
 *   temp, z, p_init [n]  in   as for pcs_mix_bubble_dew
 *   outputs, workspace        as for pcs_mix_bubble_dew
 *   order  [n] int32     in   optional (NULL: rows are bucketed by class inside each workgroup): a permutation of the rows,
 *                             position -> row, normally sorted by model class (association class x polarity) so that a
 *                             wavefront runs one set of branches.  The rows of a model are fixed, so the caller computes
 *                             it once.  Only the schedule changes: results are identical.
 */
int64_t pcs_gc_table_doubles(int S);
int pcs_gc_bubble_dew(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                      const double* z, const double* p_init, int64_t n, double* p_out, double* rho4, uint8_t* status,
                      int32_t* iters, const int32_t* order, void* workspace, void* stream);

/* GcPcSaftMix.derivatives (feos_torch/gc_pcsaft.py:443-468): a, p, mu [n,2], v [n,2] at rho [n,2]. */
int pcs_gc_derivatives(const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                       const double* rho, int64_t n, double* a, double* p, double* mu, double* v, void* stream);

/*
 * Gradient of the gc bubble / dew pressure [Pa] at the converged densities rho4:
 *   jac [n,7] = d p / d (A00, A01, A11, B00, B01, B11, T), the dispersion aggregates
 *               rho1mix = A00 r0^2 + A01 r0 r1 + A11 r1^2, rho2mix likewise with B
 *               (feos_torch/gc_pcsaft.py:177-194) — k_ab and phi enter the model only through them;
 *   agg [n,6] = the aggregate values (optional);
 *   order [n]   = optional class order of the rows as for pcs_gc_bubble_dew (schedule only).
 */
int pcs_gc_jacobian(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                    const double* rho4, int64_t n, double* jac, double* agg, const int32_t* order, void* stream);

/*
 * Gradient of the gc bubble / dew pressures [Pa] w.r.t. the SEGMENT PARAMETER TABLE, contracted with an upstream
 * gradient over the rows — what torch reverse mode through GcPcSaftMix.bubble_point / dew_point delivers to the eight
 * segment parameter vectors passed to the constructor (feos_torch/gc_pcsaft.py:14-22, :54-86, :470-512):
 *   grad_seg [S,8]  +=  sum_i gout[i] * d p_i / d seg[S,8]      (m, sigma, epsilon_k, mu, kappa_ab, epsilon_k_ab, na, nb)
 *   gout     [n]    in   upstream gradient dL/dp_i (NULL = 1 for every row)
 *   grad_seg [S*8]  inout ACCUMULATED with fp64 atomics: the caller zeroes it (or keeps accumulating over shards)
 *   rho4, order     as for pcs_gc_jacobian (converged densities of the same rows; optional class order)
 * d sqrt(eps_a eps_b) / d eps_a is taken as 0 where eps_a = 0 (the square root is not differentiable there; the
 * reference's autograd returns NaN for every epsilon_k of such a table).
 */
int pcs_gc_segment_gradient(int dew, const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                            const double* rho4, int64_t n, const double* gout, double* grad_seg, const int32_t* order,
                            void* stream);

/*
 * ---- vector-Jacobian products of the state functions -------------------------------------------------------------
 * In the reference helmholtz_energy(_density) and derivatives are ordinary torch graphs (feos_torch/pcsaft_pure.py:106-182,
 * pcsaft_mix.py:31-154 / :395-420, gc_pcsaft.py:116-253 / :443-468): a loss built on their outputs back-propagates to the
 * parameters, the temperature and the densities.  These entry points are that backward pass: upstream gradients of the
 * outputs in (NULL = that output does not enter the loss), gradients of the loss w.r.t. the inputs out.  Forward mode
 * inside (dual-number directions per pass), contracted with the upstream gradients in the kernel.
 */

/* PcSaftPure.derivatives -> (a, p, dp).  g_a, g_p, g_dp [n] in; grad_params [n,8], grad_temp [n], grad_rho [n] out
 * (each optional). */
int pcs_pure_derivatives_vjp(const double* params, const double* temp, const double* rho, int64_t n, const double* g_a,
                             const double* g_p, const double* g_dp, double* grad_params, double* grad_temp, double* grad_rho,
                             void* stream);

/* PcSaftMix.derivatives -> (a, p, mu [n,2], v [n,2]).  g_a, g_p [n], g_mu, g_v [n,2] in;
 * grad [n,21] = dL/d(params[0,0..7], params[1,0..7], kij[0], kij[1], T, rho_0, rho_1) out.
 * workspace: optional device scratch of pcs_workspace_bytes(n) (batch-wide class order, as pcs_mix_jacobian). */
int pcs_mix_derivatives_vjp(const double* params, const double* kij, const double* temp, const double* rho, int64_t n,
                            const double* g_a, const double* g_p, const double* g_mu, const double* g_v, double* grad,
                            void* workspace, void* stream);

/* GcPcSaftMix.derivatives -> (a, p, mu, v).  Upstream gradients as for pcs_mix_derivatives_vjp;
 *   grad_seg [S*8] inout  += sum_i dL_i / d seg[S,8]  (accumulated, caller zeroes; see pcs_gc_segment_gradient)
 *   jac9     [n,9]  out   dL_i / d (A00, A01, A11, B00, B01, B11, T, rho_0, rho_1): k_ab and phi enter through the
 *                         six dispersion aggregates (see pcs_gc_jacobian), the caller chains them
 *   agg      [n,6]  out   the aggregate values (optional) */
int pcs_gc_derivatives_vjp(const double* table, int S, const uint8_t* rows, const double* phi, const double* temp,
                           const double* rho, int64_t n, const double* g_a, const double* g_p, const double* g_mu,
                           const double* g_v, double* grad_seg, double* jac9, double* agg, const int32_t* order, void* stream);

/*
 * Order-preserving stream compaction of dense solver outputs.  The reference drops the rows its solver
 * fails on inside the native call (src/pcsaft.rs:93-101, :216-231: `rho[n_ok, ..]` + `status[N]`) and filters the model
 * with the same mask (feos_torch/pcsaft_pure.py:235-243, pcsaft_mix.py:470-479, gc_pcsaft.py:514-528).
 *   pcs_compact_workspace_bytes(n)  bytes of the plan buffer `cws`
 *   pcs_compact_plan    status [n] (uint8, 0 = keep) -> cws; afterwards ((int32_t*)cws)[0] = number of kept rows (the one
 *                       value a caller has to read back to size its outputs)
 *   pcs_compact_rows    dst[j, 0..width) = src[i_j, 0..width) for the j-th kept row, rows of `width` doubles (1..64);
 *                       index [n_ok] (int32, optional) = i_j.  src / dst may be NULL together (index only).
 *   pcs_expand_rows     inverse, fused with the chain rule of the backward pass: dst[i, c] = g[j] * src[j, col0 + c]
 *                       (c < ncol) for kept rows, 0 for dropped ones; src rows have src_stride doubles; g NULL = 1;
 *                       status NULL = every row kept (j = i, cws unused).
 */
int64_t pcs_compact_workspace_bytes(int64_t n);
int pcs_compact_plan(const uint8_t* status, int64_t n, void* cws, void* stream);
int pcs_compact_rows(const uint8_t* status, int64_t n, const void* cws, const double* src, int width, double* dst,
                     int32_t* index, void* stream);
int pcs_expand_rows(const uint8_t* status, int64_t n, const void* cws, const double* g, const double* src, int src_stride,
                    int col0, int ncol, double* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCSAFT_HIP_H */
