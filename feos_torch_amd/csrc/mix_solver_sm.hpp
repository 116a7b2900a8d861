// The bubble / dew solve as a per-lane state machine around ONE evaluation site (device only).
//
// The algorithm and its constants are documented in mix_solver.hpp; the CPU oracle (oracle/mix_solver.hpp) restates it
// sequentially, decision by decision.  A sequential form
// nests loops (pure-liquid roots, successive substitution with occasional root re-solves, Newton)
// whose trip counts differ per lane, so a wave executes the union of all lanes' paths one after
// the other.  Here every lane carries its stage in registers and each pass of the single wave-level
// loop performs exactly one T2 evaluation for every unfinished lane, whatever stage it is in: the
// wave's cost is the largest per-lane evaluation count, not the sum over code paths.
//
// Differences in arithmetic (all below the solver tolerances, parity tests unchanged):
//  * liquid roots take p and dp/drho along the composition from the T2 evaluation
//    (dp = x0 dp/drho_0 + x1 dp/drho_1) instead of a separate D2 line evaluation;
//  * where the sequential form re-evaluates at a freshly converged root (pure-liquid fugacities,
//    the bubble-point liquid, the first sweep), the last root evaluation is carried to the root to
//    first order with the Hessian (the step is <= 1e-6 relative).
#pragma once
#include "mix_solver.hpp"


namespace pcs {

// evaluations a robust second attempt may use (all drivers).  Rows it recovers need a bracketed root (~8) + a few Newton
// iterations (bubble) resp. two pure roots + ~5 sweeps + ~8 Newton iterations (dew); rows without a solution would run
// 200-350 evaluations each and, being few and scattered, make up the tail of the work-queue kernel
constexpr int PCS_ROBUST_BUDGET_BUBBLE = 48;
constexpr int PCS_ROBUST_BUDGET_DEW = 96;
template <bool DEW> constexpr int robust_eval_budget() { return DEW ? PCS_ROBUST_BUDGET_DEW : PCS_ROBUST_BUDGET_BUBBLE; }

// Per-lane solver state.  start() -> { point(); e = phase_eval(...); consume(e); } until done().
// `robust` (second attempt on a row the plain form gives up on; a run-time flag of the lane so that a persistent wave can
// restart a failed row in place): every liquid root is BRACKETED before it is refined --
// hi = the first of eta = 0.5, 0.62, 0.70, 0.78, 0.86 with p > p_spec and dp > 0, lo = the candidate below it (for 0.5: the
// density is lowered by factors 0.8 until p < p_spec or dp < 0) -- then Newton on the scaled function safeguarded by
// bisection.  Very cold heavy components (T/Tc < 0.25) need it: above eta = 0.5 their PC-SAFT pressure is neither monotone
// nor convex and the plain Newton of the first pass jumps over the root into the unstable region.  Judged with the oracle's
// independent continuation solver (oracle/mix_continuation.hpp, tests/test_mix_missed_gpu.py): of the rows the first pass
// fails on, the ones that do have a solution are recovered by this pass.  A bubble-point row whose specified liquid is
// diffusionally unstable at its root (inside a liquid-liquid spinodal: 98 % of the rows that fail) is given up at once.
// MODE: BD_MODE_FULL -- the whole solve; BD_MODE_INIT -- the plain form's initialisation only: where the Newton iteration
// would start the lane is done with rc = BD_HANDOVER and (rho_spec, rho_inc_1, rho_inc_2) in out.spec0 / inc0 / inc1, to be
// continued by a FULL lane through start_newton() (the two work-queue kernels of mix_kernels.hip).  The robust branches and
// the Newton stages are compiled out of an INIT lane.
enum : int { BD_MODE_FULL = 0, BD_MODE_INIT = 1 };
constexpr int BD_HANDOVER = 3;  // return code of an INIT lane (next to BD_OK / BD_FAILED / BD_CAP)

template <bool DEW, int MODE = BD_MODE_FULL>
struct BdLane {
    bool robust;
    bool root_failed;  // a cold liquid root of this attempt failed: the only kind of failure the robust form can repair
    enum : int { S_ROOT, S_SS, S_NEWTON_S, S_NEWTON_N, S_DONE };
    enum : int { R_PURE0, R_PURE1, R_SS, R_BUBBLE };  // who asked for the liquid root
    int stage, rc;
    double z0, z1, p_init;
    int ss_max, newton_max;
    // liquid-root sub-machine
    int r_for, r_it;
    bool r_dense, r_has_alt, r_warm;
    double r_x0, r_x1, r_pk, r_rho, r_pspec, r_palt, r_errprev;
    int r_phase, r_k;     // robust: 0 = looking for hi among the candidates, 1 = walking down for lo, 2 = safeguarded Newton
    double r_lo, r_hi;    // robust: bracket of the root
    // dew initialisation
    double f0, rl0, x0, x1, p0, rl, xi_prev, res_prev, xi_lo, xi_hi;  // rl0: zero-pressure liquid density of pure component 0
    int ss;
    bool resolved;
    // Newton
    double rs, ri0, ri1, err_prev, err_best;
    int it, it_best;
    // damped second run of the Newton stage (see newton_failed): the start of the stage, the largest correction of the last
    // accepted iteration, the step that led to the current point, halvings of that step
    bool damped, may_damp;  // may_damp: the driver allows the damped run (set by start / start_newton; the gc kernels clear it)
    int np_limit;           // iterations without a new smallest correction after which the Newton stage gives up (drivers may shorten it)
    int n_bt;
    double rs0, ri00, ri10, m_prev, st0, st1, st2;
    PhaseEval sv;
    MixResult out;

    PCS_DEV void idle() { stage = S_DONE; rc = BD_FAILED; }
    PCS_DEV bool done() const { return stage == S_DONE; }

    template <class Model>
    PCS_DEV void start_root(const Model& m, int who, double xa, double xb, double pspec, bool has_alt, double palt,
                            double rho_start = 0.0) {
        r_for = who;
        r_x0 = xa;
        r_x1 = xb;
        r_pk = m.packing(xa, xb);
        // warm start from the liquid density tracked at the previous composition (re-solves during the substitution);
        // a warm start that misbehaves falls back to the cold one
        r_warm = !robust && rho_start > 0.0 && rho_start * r_pk < 0.7;  // warm starts are an optimisation of the first attempt
        r_rho = r_warm ? rho_start : 0.5 / r_pk;
        r_phase = 0; r_k = 0; r_lo = 0.0; r_hi = 0.0;
        r_pspec = pspec;
        r_has_alt = has_alt;
        r_palt = palt;
        r_it = 0;
        r_dense = false;
        r_errprev = 1.0;
        stage = S_ROOT;
    }

    template <class Model>
    PCS_DEV void start(const Model& m, double z0_, double p_init_, int ss_max_ = SS_MAX_IT, int newton_max_ = NEWTON_MAX_IT,
                       bool robust_ = false, double fug0 = -1.0, double fug1 = -1.0, double rho0_ = 0.0, double rho1_ = 0.0) {
        robust = robust_;
        root_failed = false;
        z0 = z0_; z1 = 1.0 - z0_; p_init = p_init_;
        ss_max = ss_max_; newton_max = newton_max_;
        rc = BD_FAILED;
        f0 = 0.0; rl0 = 0.0; x0 = z0; x1 = z1; p0 = p_init; rl = 0.0; xi_prev = 0.0; res_prev = 0.0; xi_lo = -1e300; xi_hi = 1e300;
        ss = 0;
        resolved = false;
        rs = 0.0; ri0 = 0.0; ri1 = 0.0;
        damped = false; may_damp = true; rs0 = ri00 = ri10 = 0.0;
        np_limit = DEW ? NEWTON_NO_PROGRESS : NEWTON_NO_PROGRESS_BUBBLE;
        reset_newton();
        sv.r0 = sv.r1 = sv.a = sv.g0 = sv.g1 = sv.h00 = sv.h01 = sv.h11 = 0.0;
        out.spec0 = out.spec1 = out.inc0 = out.inc1 = out.p = 0.0;
        out.iters = 0;
        if (DEW && !robust_ && fug0 > 0.0 && fug1 > 0.0 && is_finite_bits(fug0) && is_finite_bits(fug1)) {
            // pure-liquid fugacities from the pre-pass (k_mix_pure_fugacity: the same two roots on the pure-component
            // evaluation): straight to Raoult's law, as after R_PURE1 below
            f0 = fug0;
            rl0 = rho0_;
            raoult(m, fug1, rho1_);
        } else if (DEW) start_root(m, R_PURE0, 1.0, 0.0, 0.0, false, 0.0);
        else start_root(m, R_BUBBLE, z0, z1, p_init, true, 0.0);
    }

    PCS_DEV void reset_newton() {
        err_prev = 1.0; err_best = 1e300;
        it = 0; it_best = 0;
        n_bt = 0; m_prev = 1e300; st0 = st1 = st2 = 0.0;
    }
    // the Newton stage starts at (rs, ri0, ri1)
    PCS_DEV void enter_newton() {
        rs0 = rs; ri00 = ri0; ri10 = ri1;
        stage = S_NEWTON_S;
    }
    // The Newton stage has failed (singular step, no progress, collapse onto the trivial solution, iteration cap).  Dew
    // points, plain form, full caps: one second run from the start of the stage with the natural monotonicity test -- where
    // the correction at the new point is larger than the one that led there, half of the step is taken back (at most
    // NEWTON_DAMPED_HALVINGS times in a row) -- and at most NEWTON_DAMPED_MAX_IT iterations.  It recovers rows whose
    // substitution settles on the ideal-vapour fixed point of a nearly critical liquid, from where the plain iteration
    // diverges or cycles (round 3: 24 of the 83 missed dew rows per 1e6; the first run is untouched, so no row is lost).
    PCS_DEV void newton_failed() {
        if (DEW && may_damp && !robust && !damped && newton_max >= NEWTON_MAX_IT) {
            damped = true;
            rs = rs0; ri0 = ri00; ri1 = ri10;
            reset_newton();
            stage = S_NEWTON_S;
        } else {
            stage = S_DONE;  // rc = BD_FAILED
        }
    }

    // Raoult's law from the pure-liquid fugacities (f0, f1) and the start of the successive substitution.  The liquid at the
    // Raoult composition starts at the ideal-mixing (Amagat) density of the two pure liquids, 1/rho = x_0/rho_0 + x_1/rho_1,
    // and goes straight into the first sweep: its evaluation carries the density to the zero-pressure root by the Newton step
    // it provides (as every later sweep does), or, if that step is not small, asks for the root from there -- instead of a
    // cold root solve from eta = 0.5 before the first sweep (2-3 evaluations per dew row; round 3).  Plain form only: the
    // robust second attempt keeps its bracketed root.
    template <class Model>
    PCS_DEV void raoult(const Model& m, double f1, double rl1) {
        p0 = 1.0 / (z0 / f0 + z1 / f1);
        x0 = z0 * p0 / f0;
        x1 = z1 * p0 / f1;
        const double amagat = 1.0 / (x0 / rl0 + x1 / rl1);
        if (!robust && rl0 > 0.0 && rl1 > 0.0 && is_finite_bits(amagat) && amagat > 0.0) {
            rl = amagat;
            resolved = false;
            stage = S_SS;
        } else {
            start_root(m, R_SS, x0, x1, 0.0, true, p0);
        }
    }

    // continue a row an INIT lane has initialised: the Newton iteration from (rho_spec, rho_inc_1, rho_inc_2);
    // root_failed_ = a liquid root of that initialisation had failed (decides whether a failing Newton is followed by the
    // robust second attempt)
    PCS_DEV void start_newton(double z0_, double p_init_, double rs_, double ri0_, double ri1_, bool root_failed_) {
        robust = false;
        root_failed = root_failed_;
        z0 = z0_; z1 = 1.0 - z0_; p_init = p_init_;
        ss_max = SS_MAX_IT; newton_max = NEWTON_MAX_IT;
        rc = BD_FAILED;
        rs = rs_; ri0 = ri0_; ri1 = ri1_;
        damped = false; may_damp = true;
        np_limit = DEW ? NEWTON_NO_PROGRESS : NEWTON_NO_PROGRESS_BUBBLE;
        reset_newton();
        sv.r0 = sv.r1 = sv.a = sv.g0 = sv.g1 = sv.h00 = sv.h01 = sv.h11 = 0.0;
        out.spec0 = out.spec1 = out.inc0 = out.inc1 = out.p = 0.0;
        out.iters = 0;
        enter_newton();
    }

    // INIT lanes: done, to be continued at the Newton iteration
    PCS_DEV void handover() {
        out.spec0 = rs; out.inc0 = ri0; out.inc1 = ri1;
        rc = BD_HANDOVER;
        stage = S_DONE;
    }

    // partial densities of this lane's next evaluation
    PCS_DEV void point(double& e0, double& e1) const {
        if (stage == S_ROOT) { e0 = r_x0 * r_rho; e1 = r_x1 * r_rho; }
        else if (stage == S_SS) { e0 = x0 * rl; e1 = x1 * rl; }
        else if (stage == S_NEWTON_S) { e0 = z0 * rs; e1 = z1 * rs; }
        else { e0 = ri0; e1 = ri1; }
    }

    template <class Model>
    PCS_DEV void consume(const Model& m, const PhaseEval& e) {
#define PCS_SM_START_ROOT(who, xa, xb, pspec, has_alt, palt) start_root(m, who, xa, xb, pspec, has_alt, palt)
#define PCS_SM_START_ROOT_WARM(who, xa, xb, pspec, has_alt, palt, rho0) start_root(m, who, xa, xb, pspec, has_alt, palt, rho0)
        if (stage == S_ROOT) {
            double p = e.p(), dp = r_x0 * e.dp0() + r_x1 * e.dp1();
            bool bad = false, done = false;
            double step = 0.0, rho_new = r_rho;
            if (MODE != BD_MODE_INIT && robust) {
                bad = !is_finite_bits(p);
                const bool above = (p > r_pspec) && (dp > 0.0);  // on the liquid branch above the root
                if (!bad && r_phase == 0) {
                    if (above) {
                        r_hi = r_rho;
                        if (r_k == 0) { r_phase = 1; r_lo = r_rho; r_it = 0; r_rho = 0.8 * r_rho; return; }
                        r_phase = 2;  // r_lo = the candidate below; this evaluation (at hi) starts the Newton
                    } else {
                        r_lo = r_rho;
                        r_k++;
                        if (r_k >= 5) bad = true;
                        else { r_rho = (r_k == 1 ? 0.62 : (r_k == 2 ? 0.70 : (r_k == 3 ? 0.78 : 0.86))) / r_pk; return; }
                    }
                } else if (!bad && r_phase == 1) {
                    if (!above) { r_lo = r_rho; r_phase = 2; r_it = 0; r_rho = 0.5 * (r_lo + r_hi); return; }
                    r_hi = r_rho;  // still above the root: a tighter hi
                    if (++r_it >= 12) bad = true;
                    else { r_rho = 0.8 * r_rho; return; }
                }
                if (!bad) {  // phase 2: bracket update, scaled Newton, bisection where Newton leaves the bracket
                    if (above) r_hi = r_rho; else r_lo = r_rho;
                    const double den = dp - 4.0 * (p - r_pspec) * r_pk * d_recip(1.0 - r_rho * r_pk);
                    rho_new = (dp > 0.0 && den > 0.0) ? r_rho - (p - r_pspec) * d_recip(den) : -1.0;
                    const bool newton = rho_new > r_lo && rho_new < r_hi;
                    if (!newton) rho_new = 0.5 * (r_lo + r_hi);
                    step = r_rho - rho_new;
                    done = (newton && fabs(step) <= LIQ_ROOT_TOL * r_rho) || (r_hi - r_lo) <= 1e-12 * r_hi;
                    r_it++;
                    if (!done && r_it >= 2 * LIQ_ROOT_MAX_IT) bad = true;
                }
            } else {
            if (r_it == 0 && !r_dense && !r_warm && !(p > r_pspec)) {
                r_rho = 0.62 / r_pk;  // very cold / dense: restart on the dense side (plain Newton from there)
                r_dense = true;
                return;
            }
            double den = r_dense ? dp : dp - 4.0 * (p - r_pspec) * r_pk * d_recip(1.0 - r_rho * r_pk);
            bad = !(dp > 0.0) || !(den > 0.0) || !is_finite_bits(p);
            step = (p - r_pspec) * d_recip(den);
            rho_new = r_rho - step;
            bad = bad || !(rho_new > 0.0) || !is_finite_bits(rho_new);
            // dense restart: the root is bracketed by eta = 0.5 (p < p_spec) and 0.62 (p > p_spec).  An iterate that leaves
            // that interval is on its way to ANOTHER liquid-like root of a very cold fluid (the pressure is not monotone up
            // there); which one the plain Newton would land on is a matter of luck, so the row goes to the robust form, whose
            // systematic search always takes the first bracket above eta = 0.5
            bad = bad || (r_dense && !(rho_new * r_pk > 0.5 && rho_new * r_pk < 0.62));
            if (!bad) {
                double err = fabs(step) * d_recip(r_rho);
                done = err <= LIQ_ROOT_TOL || (r_it >= 3 && err < 1e-7 && err >= 0.25 * r_errprev);
                r_errprev = err;
                r_it++;
                if (!done && r_it >= LIQ_ROOT_MAX_IT) bad = true;
            }
            }
            if (bad && r_warm) {  // the warm start left the liquid branch: same root from the cold start
                PCS_SM_START_ROOT(r_for, r_x0, r_x1, r_pspec, r_has_alt, r_palt);
                return;
            }
            if (bad) {
                root_failed = true;
                if (r_has_alt) {  // second choice of the specified pressure
                    PCS_SM_START_ROOT(r_for, r_x0, r_x1, r_palt, false, 0.0);
                } else if (!robust && (r_for == R_PURE0 || r_for == R_PURE1)) {
                    // plain form: a pure-liquid root that the plain Newton cannot find (very cold component) -- give the row
                    // to the robust form at once instead of iterating from an uninformed start
                    stage = S_DONE;  // rc = BD_FAILED
                } else if (r_for == R_PURE0 || r_for == R_PURE1) {
                    // no Raoult estimate: start the substitution from the vapour composition at the caller's pressure
                    p0 = p_init; x0 = z0; x1 = z1;
                    PCS_SM_START_ROOT(R_SS, x0, x1, 0.0, true, p0);
                } else {
                    stage = S_DONE;  // rc = BD_FAILED
                }
                return;
            }
            if (!done) { r_rho = rho_new; return; }
            // converged: chemical potentials carried to the root to first order
            const double g0c = e.g0 - (r_x0 * e.h00 + r_x1 * e.h01) * step;
            const double g1c = e.g1 - (r_x0 * e.h01 + r_x1 * e.h11) * step;
            if (r_for == R_PURE0) {
                f0 = rho_new * exp(g0c);
                rl0 = rho_new;
                PCS_SM_START_ROOT(R_PURE1, 0.0, 1.0, 0.0, false, 0.0);
                return;
            }
            if (r_for == R_PURE1) {
                raoult(m, rho_new * exp(g1c), rho_new);
                return;
            }
            if (r_for == R_BUBBLE) {
                // (Rounds 1-2 gave a row of the robust attempt up here when its liquid lay deep inside a liquid-liquid spinodal,
                // det M <= -0.5 |M00 M11| with M = d2(a + ideal)/drho_i drho_j: 98 % of the failing bubble rows, each a
                // potential tail of the queue.  The continuation solver finds a -- metastable -- bubble point on 36 of those rows
                // per 2e5, and since round 3 the robust rows start at the HEAD of the second queue, where their 48-evaluation
                // budget overlaps with the bulk: the test is gone; missed bubble rows 42 -> 6 per 2e5.)
                rs = rho_new;
                ri0 = (z0 * rs) * exp(g0c);  // ideal vapour at the liquid's fugacities
                ri1 = (z1 * rs) * exp(g1c);
                if (MODE == BD_MODE_INIT) handover(); else enter_newton();
                return;
            }
            // R_SS: this evaluation (one tiny step away from the root) serves as the sweep's evaluation
            rl = r_rho;
            stage = S_SS;
        }

        if (stage == S_SS) {
            double p = e.p(), dp = x0 * e.dp0() + x1 * e.dp1();
            double drho = -p * d_recip(dp);
            const bool fine = (dp > 0.0) && is_finite_bits(p);
            if (!(fine && fabs(drho) <= 0.05 * rl)) {
                if (!resolved) {  // composition moved a lot: re-solve the liquid root here, then redo the sweep
                    resolved = true;
                    PCS_SM_START_ROOT_WARM(R_SS, x0, x1, 0.0, true, p0, fine ? rl : 0.0);
                    return;
                }
                if (!fine) { stage = S_DONE; return; }
                drho = 0.0;
            }
            resolved = false;
            const double rlc = rl + drho;
            // w_i = z_i / (rho exp(G_i)) with the smaller exponent factored out: far from the solution G_i exceeds the range
            // of exp (both weights 0, composition 0/0) although only their ratio and the pressure estimate are needed
            const double G0 = e.g0 + (x0 * e.h00 + x1 * e.h01) * drho, G1 = e.g1 + (x0 * e.h01 + x1 * e.h11) * drho;
            const double Gm = fmin(G0, G1);
            const double w0 = z0 * exp(Gm - G0), w1 = z1 * exp(Gm - G1);
            rl = rlc;
            const double rw = d_recip(w0 + w1);
            const double sum = (w0 + w1) * d_recip(rlc * exp(Gm));
            double n0 = w0 * rw, n1 = w1 * rw;
            const double dx = fabs(n0 - x0);
            const double xi = d_log(x0 * d_recip(x1));
            const double res = d_log(n0 * d_recip(n1)) - xi;
            bool secant = false;
            // bracket of the fixed point: r > 0 at xi_lo, r < 0 at xi_hi (see mix_solver.hpp)
            if (res > 0.0 && xi > xi_lo) xi_lo = xi;
            if (res < 0.0 && xi < xi_hi) xi_hi = xi;
            if (ss > 0 && xi != xi_prev) {
                const double slope = (res - res_prev) * d_recip(xi - xi_prev);
                if (slope < SS_SECANT_SLOPE) {
                    const double dxi = fmin(fmax(-res * d_recip(slope), -1.6), 1.6);
                    const double ee = exp(xi + dxi);
                    x1 = d_recip(1.0 + ee);
                    x0 = ee * x1;
                    secant = true;
                }
            }
            xi_prev = xi;
            res_prev = res;
            if (!secant) {
                n0 = fmin(fmax(n0, 0.2 * x0), 5.0 * x0);
                n1 = fmin(fmax(n1, 0.2 * x1), 5.0 * x1);
                const double s2 = n0 + n1;
                const double rs2 = d_recip(s2);
                x0 = n0 * rs2;
                x1 = n1 * rs2;
            }
            bool narrow = false;
            if (xi_lo < xi_hi && xi_lo > -1e299 && xi_hi < 1e299) {
                const double xin = d_log(x0 * d_recip(x1));
                if (!(xin > xi_lo && xin < xi_hi)) {
                    const double ee = exp(0.5 * (xi_lo + xi_hi));
                    x1 = d_recip(1.0 + ee);
                    x0 = ee * x1;
                }
                narrow = xi_hi - xi_lo < SS_TOL;
            }
            p0 = d_recip(sum);
            ss++;
            const bool settled = (dx < SS_TOL && fabs(res) < SS_RES_TOL) || narrow;
            if (settled || ss >= ss_max) {
                if (!settled && ss_max < SS_MAX_IT) { rc = BD_CAP; stage = S_DONE; return; }
                ri0 = x0 * rl;
                ri1 = x1 * rl;
                rs = p0;
                if (MODE == BD_MODE_INIT) handover(); else enter_newton();
            }
            return;
        }

        if (MODE == BD_MODE_INIT) return;  // (no Newton stages in an INIT lane)

        if (stage == S_NEWTON_S) {
            sv = e;
            stage = S_NEWTON_N;
            return;
        }

        if (stage == S_NEWTON_N) {
            const PhaseEval& s = sv;
            const PhaseEval& n = e;
            double A[3][4];
            A[0][0] = rs * (z0 * (d_recip(s.r0) + s.h00) + z1 * s.h01);
            A[1][0] = rs * (z0 * s.h01 + z1 * (d_recip(s.r1) + s.h11));
            A[2][0] = rs * (z0 * s.dp0() + z1 * s.dp1());
            A[0][1] = -(1.0 + ri0 * n.h00);
            A[1][1] = -ri0 * n.h01;
            A[2][1] = -ri0 * n.dp0();
            A[0][2] = -ri1 * n.h01;
            A[1][2] = -(1.0 + ri1 * n.h11);
            A[2][2] = -ri1 * n.dp1();
            A[0][3] = -(s.mu0() - n.mu0());
            A[1][3] = -(s.mu1() - n.mu1());
            A[2][3] = -(s.p() - n.p());
            double du[3];
            if (!solve3(A, du)) { newton_failed(); return; }
            const double mx = fmax(fabs(du[0]), fmax(fabs(du[1]), fabs(du[2])));
            if (!is_finite_bits(mx)) { newton_failed(); return; }
            if (damped) {
                if (it > 0 && mx > m_prev && n_bt < NEWTON_DAMPED_HALVINGS && m_prev > 1e-3) {
                    st0 *= 0.5; st1 *= 0.5; st2 *= 0.5;
                    rs *= exp(-st0); ri0 *= exp(-st1); ri1 *= exp(-st2);
                    n_bt++;
                    it++;
                    if (it >= NEWTON_DAMPED_MAX_IT) newton_failed(); else stage = S_NEWTON_S;
                    return;
                }
                m_prev = mx;
                n_bt = 0;
            }
            if (mx < NEWTON_PROGRESS * err_best) { err_best = mx; it_best = it; }
            else if (it - it_best >= np_limit) { newton_failed(); return; }
            // at most a factor e per iteration -- except for a trace component of the incipient phase (mole fraction below
            // NEWTON_TRACE): its chemical potential is linear in ln rho_i there (ideal dilution), so the Newton step lands on
            // the solution however long it is and limiting it only makes the iteration march (rows with p ~ 1e-10 Pa and
            // x_i ~ 1e-30 needed 35 ... 90 iterations of unit steps, the longer ones ran into the cap)
            const double rtot = ri0 + ri1;
            const bool tr0 = ri0 < NEWTON_TRACE * rtot, tr1 = ri1 < NEWTON_TRACE * rtot;
            const double mxl = fmax(fabs(du[0]), fmax(tr0 ? 0.0 : fabs(du[1]), tr1 ? 0.0 : fabs(du[2])));
            const double scale = mxl > 1.0 ? d_recip(mxl) : 1.0;
            st0 = scale * du[0];
            st1 = tr0 ? fmin(fmax(du[1], -NEWTON_TRACE_MAX), NEWTON_TRACE_MAX) : scale * du[1];
            st2 = tr1 ? fmin(fmax(du[2], -NEWTON_TRACE_MAX), NEWTON_TRACE_MAX) : scale * du[2];
            rs *= exp(st0);
            ri0 *= exp(st1);
            ri1 *= exp(st2);
            out.iters = it + 1;
            // the iteration has collapsed onto the trivial solution (both phases identical): the Jacobian is singular there
            // and the steps wander along its null direction until a cap stops them -> give the row up now
            if (fabs(ri0 + ri1 - rs) <= 1e-6 * rs && fabs(ri0 - z0 * rs) <= 1e-6 * rs) { newton_failed(); return; }
            const bool stagnated = it >= 3 && mx < NEWTON_FLOOR && mx >= 0.25 * err_prev;
            err_prev = mx;
            it++;
            if (mx <= NEWTON_ACCEPT || stagnated) {
                const double dens_i = ri0 + ri1;
                const double lo = DEW ? rs : dens_i, hi = DEW ? dens_i : rs;
                if (!(lo < hi * (1.0 - 1e-6))) { newton_failed(); return; }  // trivial solution
                stage = S_DONE;
                // converged: the state these two evaluations were taken at is within mx of the solution and the
                // reference's final formula is second order in that error, so it is applied to them directly (no
                // further evaluation); the densities handed out carry the last step
                out.spec0 = z0 * rs; out.spec1 = z1 * rs; out.inc0 = ri0; out.inc1 = ri1;
                out.p = bubble_dew_formula(s, n);
                rc = is_finite_bits(out.p) ? BD_OK : BD_FAILED;
            } else if (it >= (damped ? NEWTON_DAMPED_MAX_IT : newton_max)) {
                if (newton_max < NEWTON_MAX_IT) { rc = BD_CAP; stage = S_DONE; }
                else newton_failed();
            } else {
                stage = S_NEWTON_S;
            }
            return;
        }

#undef PCS_SM_START_ROOT
#undef PCS_SM_START_ROOT_WARM
    }
};

// Zero-pressure liquid fugacities of the two PURE components for Raoult's law (dew points), found on the ONE-variable
// evaluation along each component's axis before the state machine starts -- the same Newton iteration as the plain form's
// S_ROOT stage (same start, scaled function, dense restart, acceptance rule and first-order carry of the chemical potential
// to the root), so the state machine continues exactly where it would have been, to rounding.  All lanes of the wave run
// it together (the roots need 2-4 evaluations whatever the row).  A root the plain form would give up on leaves NaN, and
// the state machine takes its own route (robust second attempt).  fug[i] = rho_L,i exp(mu_res,i), rho[i] = rho_L,i.
template <class Model>
PCS_DEV void pure_fugacities_on_the_line(const Model& m, double fug[2], double rho_root[2]) {
#pragma unroll 1
    for (int comp = 0; comp < 2; comp++) {
        const double x0 = comp == 0 ? 1.0 : 0.0, x1 = 1.0 - x0;
        const double pk = m.packing(x0, x1);
        double rho = 0.5 / pk, err_prev = 1.0, f = __longlong_as_double(0x7ff8000000000000LL), rr = 0.0;
        bool dense = false, active = true;
        int it = 0;
        for (int guard = 0; guard < LIQ_ROOT_MAX_IT + 2; guard++) {
            if (__ballot(active) == 0ull) break;
            if (!active) continue;
            const D2<double> a = line_eval(m, x0, x1, rho);
            const double p = rho - a.v + rho * a.d1, dp = 1.0 + rho * a.d2;
            if (it == 0 && !dense && !(p > 0.0)) {
                rho = 0.62 / pk;
                dense = true;
                continue;
            }
            const double den = dense ? dp : dp - 4.0 * p * pk / (1.0 - rho * pk);
            bool bad = !(dp > 0.0) || !(den > 0.0) || !is_finite_bits(p);
            const double step = p / den, rho_new = rho - step;
            bad = bad || !(rho_new > 0.0) || !is_finite_bits(rho_new);
            bad = bad || (dense && !(rho_new * pk > 0.5 && rho_new * pk < 0.62));
            bool done = false;
            if (!bad) {
                const double err = fabs(step) / rho;
                done = err <= LIQ_ROOT_TOL || (it >= 3 && err < 1e-7 && err >= 0.25 * err_prev);
                err_prev = err;
                it++;
                if (!done && it >= LIQ_ROOT_MAX_IT) bad = true;
            }
            if (bad) {
                active = false;
            } else if (done) {
                f = rho_new * exp(a.d1 - a.d2 * step);
                rr = rho_new;
                active = false;
            } else {
                rho = rho_new;
            }
        }
        fug[comp] = f;
        rho_root[comp] = rr;
    }
}

// Upper bound of the evaluations one row can need: 2 pure roots + (SS_MAX_IT sweeps each with a double root re-solve)
// + 2 evaluations per Newton iteration.  Every driver of the state machine (single pass below, work queue in
// mix_kernels.hip) gives a row up beyond it, so a stage transition that fails to advance a counter fails the row
// instead of hanging the wave.
constexpr int BD_EVAL_GUARD = 4 * LIQ_ROOT_MAX_IT + SS_MAX_IT * (2 * LIQ_ROOT_MAX_IT + 2) + 2 * (NEWTON_MAX_IT + NEWTON_DAMPED_MAX_IT) + 8;

// One row per lane: every pass of the wave-level loop evaluates once for every unfinished lane.
// Two attempts: the plain form, then -- only if one of its liquid roots failed -- the robust form (try_robust).
template <bool DEW, class Model>
PCS_DEV int bubble_dew_solve_sm(const Model& m, double z0, double p_init, MixResult& out, int ss_max = SS_MAX_IT,
                                int newton_max = NEWTON_MAX_IT, bool robust = false, bool* root_failed = nullptr,
                                const double* fug = nullptr, const double* rho_pure = nullptr, bool may_damp = true,
                                int no_progress = 0) {
    BdLane<DEW> L;
    if (fug) L.start(m, z0, p_init, ss_max, newton_max, robust, fug[0], fug[1], rho_pure[0], rho_pure[1]);
    else L.start(m, z0, p_init, ss_max, newton_max, robust);
    L.may_damp = may_damp;
    if (no_progress > 0) L.np_limit = no_progress;
    for (int guard = 0; guard < (robust ? robust_eval_budget<DEW>() : BD_EVAL_GUARD); guard++) {
        if (__ballot(!L.done()) == 0ull) break;
        if (L.done()) continue;
        double e0, e1;
        L.point(e0, e1);
        PhaseEval e = phase_eval(m, e0, e1);  // the only evaluation site
        L.consume(m, e);
    }
    out = L.out;
    if (root_failed) *root_failed = L.root_failed;
    return L.done() ? L.rc : BD_FAILED;
}

// The same with ONE loop and one INLINED evaluation site for both attempts: the plain form and -- `second_attempt`, only if one
// of its liquid roots failed -- the robust form, restarted in place on the lanes that need it while the others idle (as the
// work-queue kernels do).  Used by the gc bubble-point kernel (1.48 -> 1.43 ms per 1e6 rows); the gc dew-point kernel, which also
// carries the one-variable evaluation of the pure-liquid fugacities, is faster with the two calls above (3.63 vs 3.73 ms, and 4.0
// with the evaluation inlined).
template <bool DEW, class Model>
PCS_DEV int bubble_dew_solve_sm_both(const Model& m, double z0, double p_init, MixResult& out, int ss_max = SS_MAX_IT,
                                int newton_max = NEWTON_MAX_IT, bool second_attempt = true, const double* fug = nullptr,
                                const double* rho_pure = nullptr, bool may_damp = true) {
    BdLane<DEW> L;
    if (fug) L.start(m, z0, p_init, ss_max, newton_max, false, fug[0], fug[1], rho_pure[0], rho_pure[1]);
    else L.start(m, z0, p_init, ss_max, newton_max, false);
    L.may_damp = may_damp;
    int evals = 0;
    for (int guard = 0; guard < BD_EVAL_GUARD + robust_eval_budget<DEW>(); guard++) {
        if (__ballot(!L.done()) == 0ull) break;
        if (L.done()) continue;
        double e0, e1;
        L.point(e0, e1);
        PhaseEval e = phase_eval_inline(m, e0, e1);  // the only evaluation site, inlined
        L.consume(m, e);
        if (++evals >= (L.robust ? robust_eval_budget<DEW>() : BD_EVAL_GUARD) && !L.done()) L.idle();  // rc = BD_FAILED
        if (second_attempt && L.done() && L.rc != BD_OK && !L.robust && L.root_failed) {
            L.start(m, z0, p_init, SS_MAX_IT, NEWTON_MAX_IT, true);
            L.may_damp = may_damp;
            evals = 0;
        }
    }
    out = L.out;
    return L.done() ? L.rc : BD_FAILED;
}

}  // namespace pcs
