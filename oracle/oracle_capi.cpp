// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
// Flat C entry points over the CPU restatement so tests / smoke / bench's cpu_baseline leg
// can drive it through ctypes.  OpenMP over rows mirrors the reference's rayon
// par_map_collect (src/pcsaft.rs:86-92).  `prec` selects the solver arithmetic:
// 0 = double, 1 = long double (x87 80-bit, tighter tolerance; results rounded to double).
#include <cstdint>
#include <cstring>
#include <omp.h>
#include "pcsaft_pure.hpp"
// #include "pcsaft_mix.hpp"  (added with the mixture oracle)

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
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        typedef DualN<double, 10> G;
        G p[8];
        for (int k = 0; k < 8; k++) p[k] = G::var(params[8 * i + k], k);
        PureParams<G> q = make_pure_params<G>(p);
        G Tg = G::var(T[i], 8);
        G r;
        if (which == 0) {
            r = vapor_pressure_formula<G>(q, Tg, G(rho_v[i]), G(rho_l[i]));
        } else if (which == 1) {
            // liquid_density_formula needs Dual3<G>; written out here
            typedef Dual3<G> D;
            PureParams<D> qd = {D(q.m, G(0.0), G(0.0)), D(q.sigma, G(0.0), G(0.0)), D(q.epsilon_k, G(0.0), G(0.0)),
                                D(q.mu2, G(0.0), G(0.0)), D(q.kappa_ab, G(0.0), G(0.0)),
                                D(q.epsilon_k_ab, G(0.0), G(0.0)), D(q.na, G(0.0), G(0.0)), D(q.nb, G(0.0), G(0.0))};
            G rho = G(rho_l[i]);
            D h = helmholtz_energy(qd, D(Tg, G(0.0), G(0.0)), D::diff(rho));
            G pg = rho - h.re + rho * h.v1;
            G dpg = 1.0 + rho * h.v2;
            G pspec = G::var(p_pa[i], 9) / Tg * (1.0 / P_UNIT);
            r = (rho - (pg - pspec) / dpg) / RHO_UNIT;
        } else {
            typedef Dual3<G> D;
            PureParams<D> qd = {D(q.m, G(0.0), G(0.0)), D(q.sigma, G(0.0), G(0.0)), D(q.epsilon_k, G(0.0), G(0.0)),
                                D(q.mu2, G(0.0), G(0.0)), D(q.kappa_ab, G(0.0), G(0.0)),
                                D(q.epsilon_k_ab, G(0.0), G(0.0)), D(q.na, G(0.0), G(0.0)), D(q.nb, G(0.0), G(0.0))};
            G rl = G(rho_l[i]), rv = G(rho_v[i]);
            D h = helmholtz_energy(qd, D(Tg, G(0.0), G(0.0)), D::diff(rl));
            G p_l = rl - h.re + rl * h.v1;
            G dp_l = 1.0 + rl * h.v2;
            G a_l = h.re / rl;
            G a_v = helmholtz_energy(q, Tg, rv) / rv;
            G pp = -(a_v - a_l + log(rv / rl)) / (1.0 / rv - 1.0 / rl);
            r = (rl - (p_l - pp) / dp_l) / RHO_UNIT;
        }
        value[i] = r.re;
        for (int k = 0; k < 10; k++) grad[10 * i + k] = r.eps[k];
    }
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

}  // extern "C"
