// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
// Flat C entry points over the CPU restatement so tests / smoke / bench's cpu_baseline leg
// can drive it through ctypes.  OpenMP over rows mirrors the reference's rayon
// par_map_collect (src/pcsaft.rs:86-92).  `prec` selects the solver arithmetic:
// 0 = double, 1 = long double (x87 80-bit, tighter tolerance; results rounded to double).
#include <cstdint>
#include <cstring>
#include <omp.h>
#include "pcsaft_pure.hpp"
#include "pcsaft_mix.hpp"
#include "mix_solver.hpp"
#include "mix_continuation.hpp"
#include "pcsaft_mixn.hpp"
#include "gc_pcsaft.hpp"

using namespace oracle;

namespace {

template <class F>
PureParams<F> load_pure(const double* row) {
    F p[8];
    for (int k = 0; k < 8; k++) p[k] = F(row[k]);
    return make_pure_params<F>(p);
}

template <class F>
void pure_vle_row(const double* row, double T, F tol, double* rho_v, double* rho_l, uint8_t* status,
                  int32_t* iters, int32_t* path) {
    PureParams<F> q = load_pure<F>(row);
    F rv = 0, rl = 0;
    SolveInfo info;
    bool ok = vle_pure<F>(q, F(T), rv, rl, info, tol);
    *status = ok ? 0 : 1;
    *rho_v = ok ? double(rv) : 0.0;
    *rho_l = ok ? double(rl) : 0.0;
    if (iters) *iters = info.iters;
    if (path) *path = info.path;
}

}  // namespace

// pure-component property gradients at fixed densities, F = double (the reference's fp64 autograd) or long double (exact)
template <class F>
static void pure_property_grad_impl(int which, const double* params, const double* T, const double* p_pa,
                                    const double* rho_v, const double* rho_l, int64_t n, double* value, double* grad) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        typedef DualN<F, 10> G;
        G p[8];
        for (int k = 0; k < 8; k++) p[k] = G::var((F)params[8 * i + k], k);
        PureParams<G> q = make_pure_params<G>(p);
        G Tg = G::var((F)T[i], 8);
        G r;
        if (which == 0) {
            r = vapor_pressure_formula<G>(q, Tg, G((F)rho_v[i]), G((F)rho_l[i]));
        } else if (which == 1) {
            // liquid_density_formula needs Dual3<G>; written out here
            typedef Dual3<G> D;
            PureParams<D> qd = {D(q.m, G(0.0), G(0.0)), D(q.sigma, G(0.0), G(0.0)), D(q.epsilon_k, G(0.0), G(0.0)),
                                D(q.mu2, G(0.0), G(0.0)), D(q.kappa_ab, G(0.0), G(0.0)),
                                D(q.epsilon_k_ab, G(0.0), G(0.0)), D(q.na, G(0.0), G(0.0)), D(q.nb, G(0.0), G(0.0))};
            G rho = G((F)rho_l[i]);
            D h = helmholtz_energy(qd, D(Tg, G(0.0), G(0.0)), D::diff(rho));
            G pg = rho - h.re + rho * h.v1;
            G dpg = 1.0 + rho * h.v2;
            G pspec = G::var((F)p_pa[i], 9) / Tg * (1.0 / P_UNIT);
            r = (rho - (pg - pspec) / dpg) / RHO_UNIT;
        } else {
            typedef Dual3<G> D;
            PureParams<D> qd = {D(q.m, G(0.0), G(0.0)), D(q.sigma, G(0.0), G(0.0)), D(q.epsilon_k, G(0.0), G(0.0)),
                                D(q.mu2, G(0.0), G(0.0)), D(q.kappa_ab, G(0.0), G(0.0)),
                                D(q.epsilon_k_ab, G(0.0), G(0.0)), D(q.na, G(0.0), G(0.0)), D(q.nb, G(0.0), G(0.0))};
            G rl = G((F)rho_l[i]), rv = G((F)rho_v[i]);
            D h = helmholtz_energy(qd, D(Tg, G(0.0), G(0.0)), D::diff(rl));
            G p_l = rl - h.re + rl * h.v1;
            G dp_l = 1.0 + rl * h.v2;
            G a_l = h.re / rl;
            G a_v = helmholtz_energy(q, Tg, rv) / rv;
            G pp = -(a_v - a_l + log(rv / rl)) / (1.0 / rv - 1.0 / rl);
            r = (rl - (p_l - pp) / dp_l) / RHO_UNIT;
        }
        value[i] = (double)r.re;
        for (int k = 0; k < 10; k++) grad[10 * i + k] = (double)r.eps[k];
    }
}


extern "C" {

int orc_num_threads() { return omp_get_max_threads(); }

// (a, p, dp) at given (T, rho): feos_torch/pcsaft_pure.py:180-182
void orc_pure_derivatives(const double* params, const double* T, const double* rho, int64_t n, double* a,
                          double* p, double* dp) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        PureParams<double> q = load_pure<double>(params + 8 * i);
        derivatives<double>(q, T[i], rho[i], a[i], p[i], dp[i]);
    }
}

// plain Helmholtz energy density: feos_torch/pcsaft_pure.py:106-178
void orc_pure_helmholtz(const double* params, const double* T, const double* rho, int64_t n, double* a) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        PureParams<double> q = load_pure<double>(params + 8 * i);
        a[i] = helmholtz_energy<double>(q, T[i], rho[i]);
    }
}

// converged phase densities (the role of src/pcsaft.rs:82-103 + feos PhaseEquilibrium::pure),
// dense output with status (1 = failed) instead of dropped rows.
void orc_pure_vle(const double* params, const double* T, int64_t n, int prec, double* rho_v, double* rho_l,
                  uint8_t* status, int32_t* iters, int32_t* path) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        if (prec == 1)
            pure_vle_row<long double>(params + 8 * i, T[i], 1e-17L, rho_v + i, rho_l + i, status + i,
                                      iters ? iters + i : nullptr, path ? path + i : nullptr);
        else
            pure_vle_row<double>(params + 8 * i, T[i], 1e-13, rho_v + i, rho_l + i, status + i,
                                 iters ? iters + i : nullptr, path ? path + i : nullptr);
    }
}

// PcSaftPure.vapor_pressure (feos_torch/pcsaft_pure.py:201-215): solve + final formula, Pa.
void orc_pure_vapor_pressure(const double* params, const double* T, int64_t n, int prec, double* p_out,
                             uint8_t* status) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        if (prec == 1) {
            PureParams<long double> q = load_pure<long double>(params + 8 * i);
            long double rv, rl;
            SolveInfo info;
            bool ok = vle_pure<long double>(q, T[i], rv, rl, info, 1e-17L);
            status[i] = ok ? 0 : 1;
            p_out[i] = ok ? double(vapor_pressure_formula<long double>(q, T[i], rv, rl)) : 0.0;
        } else {
            PureParams<double> q = load_pure<double>(params + 8 * i);
            double rv, rl;
            SolveInfo info;
            bool ok = vle_pure<double>(q, T[i], rv, rl, info, 1e-13);
            status[i] = ok ? 0 : 1;
            p_out[i] = ok ? vapor_pressure_formula<double>(q, T[i], rv, rl) : 0.0;
        }
    }
}

// the final formula alone at caller-supplied densities (feos_torch/pcsaft_pure.py:212-215)
void orc_pure_vapor_pressure_at(const double* params, const double* T, const double* rho_v,
                                const double* rho_l, int64_t n, double* p_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        PureParams<double> q = load_pure<double>(params + 8 * i);
        p_out[i] = vapor_pressure_formula<double>(q, T[i], rho_v[i], rho_l[i]);
    }
}

// PcSaftPure.liquid_density (feos_torch/pcsaft_pure.py:184-199): kmol/m3
void orc_pure_liquid_density(const double* params, const double* T, const double* p_pa, int64_t n, int prec,
                             double* rho_out, uint8_t* status) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        int it;
        if (prec == 1) {
            PureParams<long double> q = load_pure<long double>(params + 8 * i);
            long double rho, Tl = T[i];
            long double pr = (long double)p_pa[i] / Tl * (long double)(1.0 / P_UNIT);
            bool ok = liquid_density_at_p<long double>(q, Tl, pr, rho, it, 1e-17L);
            status[i] = ok ? 0 : 1;
            rho_out[i] = ok ? double(liquid_density_formula<long double>(q, Tl, p_pa[i], rho)) : 0.0;
        } else {
            PureParams<double> q = load_pure<double>(params + 8 * i);
            double rho;
            double pr = p_pa[i] / T[i] * (1.0 / P_UNIT);
            bool ok = liquid_density_at_p<double>(q, T[i], pr, rho, it, 1e-13);
            status[i] = ok ? 0 : 1;
            rho_out[i] = ok ? liquid_density_formula<double>(q, T[i], p_pa[i], rho) : 0.0;
        }
    }
}

// converged liquid density [A^-3] at (T, p) — the role of src/pcsaft.rs:105-129 (dense + status)
void orc_pure_liquid_density_root(const double* params, const double* T, const double* p_pa, int64_t n,
                                  int prec, double* rho_out, uint8_t* status) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        int it;
        if (prec == 1) {
            PureParams<long double> q = load_pure<long double>(params + 8 * i);
            long double rho = 0, Tl = T[i];
            long double pr = (long double)p_pa[i] / Tl * (long double)(1.0 / P_UNIT);
            bool ok = liquid_density_at_p<long double>(q, Tl, pr, rho, it, 1e-17L);
            status[i] = ok ? 0 : 1;
            rho_out[i] = ok ? double(rho) : 0.0;
        } else {
            PureParams<double> q = load_pure<double>(params + 8 * i);
            double rho = 0;
            double pr = p_pa[i] / T[i] * (1.0 / P_UNIT);
            bool ok = liquid_density_at_p<double>(q, T[i], pr, rho, it, 1e-13);
            status[i] = ok ? 0 : 1;
            rho_out[i] = ok ? rho : 0.0;
        }
    }
}

// PcSaftPure.equilibrium_liquid_density (feos_torch/pcsaft_pure.py:217-233): kmol/m3
void orc_pure_equilibrium_liquid_density(const double* params, const double* T, int64_t n, int prec,
                                         double* rho_out, uint8_t* status) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        if (prec == 1) {
            PureParams<long double> q = load_pure<long double>(params + 8 * i);
            long double rv, rl;
            SolveInfo info;
            bool ok = vle_pure<long double>(q, T[i], rv, rl, info, 1e-17L);
            status[i] = ok ? 0 : 1;
            rho_out[i] = ok ? double(equilibrium_liquid_density_formula<long double>(q, T[i], rv, rl)) : 0.0;
        } else {
            PureParams<double> q = load_pure<double>(params + 8 * i);
            double rv, rl;
            SolveInfo info;
            bool ok = vle_pure<double>(q, T[i], rv, rl, info, 1e-13);
            status[i] = ok ? 0 : 1;
            rho_out[i] = ok ? equilibrium_liquid_density_formula<double>(q, T[i], rv, rl) : 0.0;
        }
    }
}

// Gradients of the three properties w.r.t. (8 parameters, T[, p]) with the densities held
// fixed — what torch reverse mode through feos_torch/pcsaft_pure.py:196-199/:212-215/:228-233
// yields.  grad is [n, 10]: d/d(m, sigma, eps_k, mu, kappa_ab, eps_k_ab, na, nb, T, p_spec).
// which: 0 = vapor_pressure, 1 = liquid_density, 2 = equilibrium_liquid_density.
// For which = 1, rho_v is unused and rho_l is the converged density, p_pa the specification.
void orc_pure_property_grad(int which, const double* params, const double* T, const double* p_pa,
                            const double* rho_v, const double* rho_l, int64_t n, double* value, double* grad) {
    pure_property_grad_impl<double>(which, params, T, p_pa, rho_v, rho_l, n, value, grad);
}
// the same in long double (with the cancellation-free site fractions of that instantiation): the exact gradient
void orc_pure_property_grad_ld(int which, const double* params, const double* T, const double* p_pa,
                               const double* rho_v, const double* rho_l, int64_t n, double* value, double* grad) {
    pure_property_grad_impl<long double>(which, params, T, p_pa, rho_v, rho_l, n, value, grad);
}

// The reference's own Dual3 known-answer vectors (tests/test_dual.py:5-24: x = diff(4), y = 5),
// exact comparisons.  Returns 0 when every triple matches, else the 1-based index of the
// first failing check.
int orc_dual3_selftest() {
    typedef Dual3<double> D;
    auto eq = [](const D& a, double r, double v1, double v2) { return a.re == r && a.v1 == v1 && a.v2 == v2; };
    D x = D::diff(4.0);
    double y = 5.0;
    int k = 0;
    k++; if (!eq(x + x, 8, 2, 0)) return k;
    k++; if (!eq(x + y, 9, 1, 0)) return k;
    k++; if (!eq(y + x, 9, 1, 0)) return k;
    k++; if (!eq(-x, -4, -1, 0)) return k;
    k++; if (!eq(x - x, 0, 0, 0)) return k;
    k++; if (!eq(x - y, -1, 1, 0)) return k;
    k++; if (!eq(y - x, 1, -1, 0)) return k;
    k++; if (!eq(x * x, 16, 8, 2)) return k;
    k++; if (!eq(x * y, 20, 5, 0)) return k;
    k++; if (!eq(y * x, 20, 5, 0)) return k;
    k++; if (!eq(x.recip(), 0.25, -1.0 / 16, 1.0 / 32)) return k;
    k++; if (!eq(x / x, 1, 0, 0)) return k;
    k++; if (!eq(x / y, 0.8, 0.2, 0)) return k;
    k++; if (!eq(y / x, 1.25, -5.0 / 16, 5.0 / 32)) return k;
    k++; if (!eq(log(x), std::log(4.0), 0.25, -1.0 / 16)) return k;
    k++; if (!eq(exp(x), std::exp(4.0), std::exp(4.0), std::exp(4.0))) return k;
    k++; if (!eq(sqrt(x), 2, 0.25, -1.0 / 32)) return k;
    return 0;
}

// ======================================================================================
// binary mixtures (feos_torch/pcsaft_mix.py)
// ======================================================================================
}  // extern "C"

namespace {

template <class S, class F>
S lift_to(const F& x) { S s(0.0); s.re = x; return s; }
template <> [[maybe_unused]] double lift_to<double, double>(const double& x) { return x; }
template <> [[maybe_unused]] long double lift_to<long double, long double>(const long double& x) { return x; }

template <class S, class F>
MixParams<S> lift_mix(const MixParams<F>& q) {
    MixParams<S> r;
    for (int i = 0; i < 2; i++) {
        r.m[i] = lift_to<S, F>(q.m[i]); r.sigma[i] = lift_to<S, F>(q.sigma[i]); r.epsilon_k[i] = lift_to<S, F>(q.epsilon_k[i]);
        r.mu2[i] = lift_to<S, F>(q.mu2[i]); r.kappa_ab[i] = lift_to<S, F>(q.kappa_ab[i]);
        r.epsilon_k_ab[i] = lift_to<S, F>(q.epsilon_k_ab[i]); r.na[i] = lift_to<S, F>(q.na[i]); r.nb[i] = lift_to<S, F>(q.nb[i]);
    }
    r.kij = lift_to<S, F>(q.kij);
    r.eps_aibj = lift_to<S, F>(q.eps_aibj);
    r.robust = q.robust;
    return r;
}

// adapter: PcSaftMix as a `Model` for mix_solver.hpp
template <class F>
struct MixModel {
    MixParams<F> q;
    template <class S> S a(const S& T, const S* rho) const {
        MixParams<S> qs = lift_mix<S, F>(q);
        return helmholtz_energy_density_mix<S>(qs, T, rho);
    }
    F packing(F T, const F* x) const {
        F s = 0;
        for (int i = 0; i < 2; i++) {
            F d = q.sigma[i] * (F(1) - F(0.12) * oracle::exp(F(-3) * q.epsilon_k[i] / T));
            s += x[i] * q.m[i] * d * d * d;
        }
        return F(PI) / F(6) * s;
    }
};

template <class F>
MixParams<F> load_mix(const double* par16, const double* kij2) {
    F p[16];
    for (int k = 0; k < 16; k++) p[k] = F(par16[k]);
    return make_mix_params<F>(p, F(kij2[0]), F(kij2[1]));
}

template <class F>
void mix_bd_row(const double* par16, const double* kij2, double T, double z, double p_pa, bool dew, F tol,
                double* rho4, double* p_out, uint8_t* status) {
    MixModel<F> model{load_mix<F>(par16, kij2)};
    model.q.robust = true;
    F rs[2], ri[2];
    MixSolveInfo info;
    F p_red = F(p_pa) / F(T) * F(1.0 / P_UNIT);
    bool ok = bubble_dew<F>(model, F(T), F(z), p_red, dew, rs, ri, info, tol) || (info.root_failed && bubble_dew<F>(model, F(T), F(z), p_red, dew, rs, ri, info, tol, true));
    *status = ok ? 0 : 1;
    // reference layout (src/pcsaft.rs:225-228): [rhoV_1, rhoV_2, rhoL_1, rhoL_2]
    const F* v = dew ? rs : ri;
    const F* l = dew ? ri : rs;
    if (rho4) {
        rho4[0] = ok ? double(v[0]) : 0.0; rho4[1] = ok ? double(v[1]) : 0.0;
        rho4[2] = ok ? double(l[0]) : 0.0; rho4[3] = ok ? double(l[1]) : 0.0;
    }
    if (p_out) *p_out = ok ? double(bubble_dew_formula<F>(model.q, F(T), rs, ri)) : 0.0;
}

}  // namespace

extern "C" {

// PcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420)
void orc_mix_derivatives(const double* params, const double* kij, const double* T, const double* rho, int64_t n,
                         int robust, double* a, double* p, double* mu, double* v) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        MixParams<double> q = load_mix<double>(params + 16 * i, kij + 2 * i);
        q.robust = robust != 0;
        derivatives_mix<double>(q, T[i], rho + 2 * i, a[i], p[i], mu + 2 * i, v + 2 * i);
    }
}

// plain helmholtz_energy_density (:31-154)
void orc_mix_helmholtz(const double* params, const double* kij, const double* T, const double* rho, int64_t n,
                       double* a) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        MixParams<double> q = load_mix<double>(params + 16 * i, kij + 2 * i);
        a[i] = helmholtz_energy_density_mix<double>(q, T[i], rho + 2 * i);
    }
}

// The SECOND solver (mix_continuation.hpp): continuation in composition from the pure-component ends.  code [n]:
// 0 = solution found, 1 = no pure-fluid VLE next to either end, 2 = every route ended in a critical point,
// 3 = stalled (limit of stability / liquid-liquid region).  rho4 / p_out as orc_mix_bubble_dew; info [n,3] = steps, Newton
// iterations, route.
void orc_mix_bubble_dew_continuation(const double* params, const double* kij, const double* T, const double* z, int64_t n,
                                     int dew, int prec, double* rho4, double* p_out, int32_t* code, int32_t* info) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < n; i++) {
        auto run = [&](auto zero) {
            typedef decltype(zero) F;
            MixModel<F> model{load_mix<F>(params + 16 * i, kij + 2 * i)};
            model.q.robust = true;
            F rs[2] = {0, 0}, ri[2] = {0, 0};
            ContInfo ci = bubble_dew_continuation<F>(model, F(T[i]), F(z[i]), dew != 0, rs, ri, prec == 1 ? F(1e-15) : F(1e-12));
            const bool ok = ci.code == CONT_OK;
            code[i] = ci.code;
            if (info) { info[3 * i] = ci.steps; info[3 * i + 1] = ci.newton; info[3 * i + 2] = ci.route; }
            const F* v = dew ? rs : ri;
            const F* l = dew ? ri : rs;
            if (rho4) {
                rho4[4 * i] = ok ? double(v[0]) : 0.0; rho4[4 * i + 1] = ok ? double(v[1]) : 0.0;
                rho4[4 * i + 2] = ok ? double(l[0]) : 0.0; rho4[4 * i + 3] = ok ? double(l[1]) : 0.0;
            }
            if (p_out) p_out[i] = ok ? double(bubble_dew_formula<F>(model.q, F(T[i]), rs, ri)) : 0.0;
        };
        if (prec == 1) run((long double)0);
        else run(double(0));
    }
}

// converged partial densities [n,4] = (rhoV_1, rhoV_2, rhoL_1, rhoL_2) + status, and the property
// (bubble / dew pressure, Pa) from the reference's final formula.  Either output may be NULL.
void orc_mix_bubble_dew(const double* params, const double* kij, const double* T, const double* z,
                        const double* p_init_pa, int64_t n, int dew, int prec, double* rho4, double* p_out,
                        uint8_t* status) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; i++) {
        if (prec == 1)
            mix_bd_row<long double>(params + 16 * i, kij + 2 * i, T[i], z[i], p_init_pa[i], dew != 0, 1e-17L,
                                    rho4 ? rho4 + 4 * i : nullptr, p_out ? p_out + i : nullptr, status + i);
        else
            mix_bd_row<double>(params + 16 * i, kij + 2 * i, T[i], z[i], p_init_pa[i], dew != 0, 1e-7,  // the kernels' Newton tolerance (NEWTON_ACCEPT, csrc/mix_solver.hpp): the double instantiation mirrors their decisions
                               
                               rho4 ? rho4 + 4 * i : nullptr, p_out ? p_out + i : nullptr, status + i);
    }
}

// n-component PcSaftMix.derivatives (feos_torch/pcsaft_mix.py:395-420 with parameters [n, nc, 8], kij = None): a [n], p [n],
// mu [n, nc], v [n, nc].  prec = 1: long double.  Returns 1 if a row asks for two associating components (binary only in the
// reference, :250 / :336) or nc is out of range.
int orc_mixn_derivatives(const double* params, const double* T, const double* rho, int nc, int64_t n, int prec, double* a,
                         double* p, double* mu, double* v) {
    if (nc < 1 || nc > MIXN_MAX) return 1;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < n; i++) {
        auto run = [&](auto zero) {
            typedef decltype(zero) F;
            F r[MIXN_MAX], aa, pp, m2[MIXN_MAX], v2[MIXN_MAX];
            for (int k = 0; k < nc; k++) r[k] = F(rho[nc * i + k]);
            if (!derivatives_mixn<F>(nc, params + 8 * nc * i, F(T[i]), r, aa, pp, m2, v2)) bad |= 1;
            a[i] = double(aa); p[i] = double(pp);
            for (int k = 0; k < nc; k++) { mu[nc * i + k] = double(m2[k]); v[nc * i + k] = double(v2[k]); }
        };
        if (prec == 1) run((long double)0);
        else run(double(0));
    }
    return bad;
}

// PcSaftMix.derivatives in long double with the safeguarded association iterations and the cancellation-free site
// fractions: the exact values of the model, against which both the kernels and the reference's own fp64 evaluation
// (orc_mix_derivatives, robust = 0) are measured row by row.
void orc_mix_derivatives_ld(const double* params, const double* kij, const double* T, const double* rho, int64_t n,
                            double* a, double* p, double* mu, double* v) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        typedef long double F;
        MixParams<F> q = load_mix<F>(params + 16 * i, kij + 2 * i);
        q.robust = true;
        F r[2] = {F(rho[2 * i]), F(rho[2 * i + 1])}, aa, pp, m2[2], v2[2];
        derivatives_mix<F>(q, F(T[i]), r, aa, pp, m2, v2);
        a[i] = double(aa); p[i] = double(pp);
        for (int k = 0; k < 2; k++) { mu[2 * i + k] = double(m2[k]); v[2 * i + k] = double(v2[k]); }
    }
}

// orc_mix_bubble_dew_grad in long double (exact gradient of the reference's formula at the given densities)
void orc_mix_bubble_dew_grad_ld(const double* params, const double* kij, const double* T, const double* rho4,
                                int64_t n, int dew, double* value, double* grad) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        typedef DualN<long double, 19> G;
        G p[16];
        for (int k = 0; k < 16; k++) p[k] = G::var((long double)params[16 * i + k], k);
        MixParams<G> q = make_mix_params<G>(p, G::var((long double)kij[2 * i], 16), G::var((long double)kij[2 * i + 1], 17));
        q.robust = true;
        G Tg = G::var((long double)T[i], 18);
        G rv[2] = {G((long double)rho4[4 * i]), G((long double)rho4[4 * i + 1])}, rl[2] = {G((long double)rho4[4 * i + 2]), G((long double)rho4[4 * i + 3])};
        G r = dew ? bubble_dew_formula<G>(q, Tg, rv, rl) : bubble_dew_formula<G>(q, Tg, rl, rv);
        value[i] = double(r.re);
        for (int k = 0; k < 19; k++) grad[19 * i + k] = double(r.eps[k]);
    }
}

// Value and gradient of the bubble / dew pressure formula (pcsaft_mix.py:435-444 / :459-468) at
// FIXED densities rho4 = (rhoV_1, rhoV_2, rhoL_1, rhoL_2): grad[n,19] = d/d(16 parameters
// [comp0 x 8, comp1 x 8], kij[0], kij[1], T) — what torch reverse mode yields in the reference.
void orc_mix_bubble_dew_grad(const double* params, const double* kij, const double* T, const double* rho4,
                             int64_t n, int dew, double* value, double* grad) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        typedef DualN<double, 19> G;
        G p[16];
        for (int k = 0; k < 16; k++) p[k] = G::var(params[16 * i + k], k);
        MixParams<G> q = make_mix_params<G>(p, G::var(kij[2 * i], 16), G::var(kij[2 * i + 1], 17));
        // `var` on eps_aibj == 0 keeps re == 0, so the mean combining rule is still selected
        q.robust = true;
        G Tg = G::var(T[i], 18);
        G rv[2] = {G(rho4[4 * i]), G(rho4[4 * i + 1])}, rl[2] = {G(rho4[4 * i + 2]), G(rho4[4 * i + 3])};
        G r = dew ? bubble_dew_formula<G>(q, Tg, rv, rl) : bubble_dew_formula<G>(q, Tg, rl, rv);
        value[i] = r.re;
        for (int k = 0; k < 19; k++) grad[19 * i + k] = r.eps[k];
    }
}

// ======================================================================================
// heterosegmented gc-PC-SAFT (feos_torch/gc_pcsaft.py)
// ======================================================================================
}  // extern "C"

namespace {

// adapter: GcPcSaftMix as a `Model` for mix_solver.hpp.  F-valued T and densities; the
// structure and segment table stay plain doubles.
template <class F>
struct GcModel {
    GcRow r;
    template <class S> S a(const S& T, const S* rho) const { return gc_helmholtz_energy_density<S>(r, T, rho); }
    F packing(F T, const F* x) const {
        F s = 0;
        for (int i = 0; i < 2; i++)
            for (int al = 0; al < r.S; al++) {
                double m = r.counts[i * r.S + al] * r.seg[8 * al];
                if (m == 0.0) continue;
                F d = F(r.seg[8 * al + 1]) * (F(1) - F(0.12) * oracle::exp(F(-3) * F(r.seg[8 * al + 2]) / T));
                s += x[i] * F(m) * d * d * d;
            }
        return F(PI) / F(6) * s;
    }
};

bool gc_row(GcRow& r, int S, const double* seg, const double* kab, const double* counts, const double* bonds,
            const double* phi, int64_t i) {
    r.S = S;
    r.seg = seg;
    r.kab = kab;
    r.counts = counts + (size_t)i * 2 * S;
    r.bonds = bonds + (size_t)i * 2 * S * S;
    r.phi[0] = phi[2 * i];
    r.phi[1] = phi[2 * i + 1];
    return gc_prepare(r);
}

template <class F>
void gc_bd_row(GcRow& r, double T, double z, double p_pa, bool dew, F tol, double* rho4, double* p_out,
               uint8_t* status) {
    r.robust = true;
    GcModel<F> model{r};
    F rs[2], ri[2];
    MixSolveInfo info;
    F p_red = F(p_pa) / F(T) * F(1.0 / P_UNIT);
    // (no damped second run of the Newton stage for gc rows: as csrc/gc_kernels.hip)
    const int np = dew ? 10 : 0;  // GC_NO_PROGRESS_DEW of csrc/gc_kernels.hip
    bool ok = bubble_dew<F>(model, F(T), F(z), p_red, dew, rs, ri, info, tol, false, false, np) || (info.root_failed && bubble_dew<F>(model, F(T), F(z), p_red, dew, rs, ri, info, tol, true, false, np));
    *status = ok ? 0 : 1;
    const F* v = dew ? rs : ri;
    const F* l = dew ? ri : rs;
    if (rho4) {
        rho4[0] = ok ? double(v[0]) : 0.0; rho4[1] = ok ? double(v[1]) : 0.0;
        rho4[2] = ok ? double(l[0]) : 0.0; rho4[3] = ok ? double(l[1]) : 0.0;
    }
    if (p_out) *p_out = ok ? double(bubble_dew_formula_generic<F>(model, F(T), rs, ri) * F(T) * F(P_UNIT)) : 0.0;
}

}  // namespace

extern "C" {

}  // extern "C" (reopened below)

namespace {
template <class X, class G> struct LiftG;
template <class G> struct LiftG<G, G> { static G go(const G& g) { return g; } };
template <class G, int N> struct LiftG<HyperDual<G, N>, G> { static HyperDual<G, N> go(const G& g) { HyperDual<G, N> h; h.re = g; return h; } };
// gradient model: kab[ka,kb], phi_0, phi_1 carried as DualN tangents (direction 0, 1, 2; T is 3); GcG = DualN<double, 4> or,
// for the exact gradient, DualN<long double, 4>
template <class GcG>
struct GcGradModel {
    GcRow r; GcG kv; GcG ph[2]; int ka, kb;
    template <class X> X a(const X& Tt, const X* rho) const {
        X ks = LiftG<X, GcG>::go(kv), p2[2] = {LiftG<X, GcG>::go(ph[0]), LiftG<X, GcG>::go(ph[1])};
        return gc_helmholtz_energy_density<X>(r, Tt, rho, p2, ka, kb, &ks);
    }
};
}  // namespace

extern "C" {

// GcPcSaftMix.derivatives (feos_torch/gc_pcsaft.py:443-468).  seg [S,8], kab [S,S],
// counts [n,2,S], bonds [n,2,S,S] (lower triangle), phi [n,2].  Returns 1 if a row violates
// "only up to one associating segment per component" (:77-80).
int orc_gc_derivatives(int S, const double* seg, const double* kab, const double* counts, const double* bonds,
                       const double* phi, const double* T, const double* rho, int64_t n, int robust, double* a,
                       double* p, double* mu, double* v) {
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < n; i++) {
        GcRow r;
        if (!gc_row(r, S, seg, kab, counts, bonds, phi, i)) { bad |= 1; continue; }
        r.robust = robust != 0;
        GcModel<double> model{r};
        derivatives_generic<double>(model, T[i], rho + 2 * i, a[i], p[i], mu + 2 * i, v + 2 * i);
    }
    return bad;
}

// converged partial densities + bubble/dew pressure [Pa]; either output may be NULL
int orc_gc_bubble_dew(int S, const double* seg, const double* kab, const double* counts, const double* bonds,
                      const double* phi, const double* T, const double* z, const double* p_init_pa, int64_t n,
                      int dew, int prec, double* rho4, double* p_out, uint8_t* status) {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(| : bad)
    for (int64_t i = 0; i < n; i++) {
        GcRow r;
        if (!gc_row(r, S, seg, kab, counts, bonds, phi, i)) { bad |= 1; status[i] = 1; continue; }
        if (prec == 1)
            gc_bd_row<long double>(r, T[i], z[i], p_init_pa[i], dew != 0, 1e-17L, rho4 ? rho4 + 4 * i : nullptr,
                                   p_out ? p_out + i : nullptr, status + i);
        else
            gc_bd_row<double>(r, T[i], z[i], p_init_pa[i], dew != 0, 1e-13, rho4 ? rho4 + 4 * i : nullptr,
                              p_out ? p_out + i : nullptr, status + i);
    }
    return bad;
}

// value and gradient of the bubble/dew formula at fixed densities w.r.t.
// (kab[ka,kb] (= kab[kb,ka]), phi_0, phi_1, T): grad [n,4]
// prec = 1: long double (exact gradient of the reference's formula; its fp64 evaluation amplifies rounding on a few
// ill-conditioned rows)
void orc_gc_bubble_dew_grad(int S, const double* seg, const double* kab, const double* counts, const double* bonds,
                            const double* phi, const double* T, const double* rho4, int64_t n, int dew, int ka,
                            int kb, int prec, double* value, double* grad) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        auto run = [&](auto zero) {
            typedef decltype(zero) F;
            typedef DualN<F, 4> G;
            GcRow r;
            gc_row(r, S, seg, kab, counts, bonds, phi, i);
            r.robust = true;
            GcGradModel<G> model{r, G::var(F(kab[ka * S + kb]), 0), {G::var(F(phi[2 * i]), 1), G::var(F(phi[2 * i + 1]), 2)}, ka, kb};
            G Tg = G::var(F(T[i]), 3);
            G rv[2] = {G(F(rho4[4 * i])), G(F(rho4[4 * i + 1]))}, rl[2] = {G(F(rho4[4 * i + 2])), G(F(rho4[4 * i + 3]))};
            G pr = dew ? bubble_dew_formula_generic<G>(model, Tg, rv, rl) : bubble_dew_formula_generic<G>(model, Tg, rl, rv);
            G res = pr * Tg * F(P_UNIT);
            value[i] = double(res.re);
            for (int k = 0; k < 4; k++) grad[4 * i + k] = double(res.eps[k]);
        };
        if (prec == 1) run((long double)0);
        else run(double(0));
    }
}

}  // extern "C"
