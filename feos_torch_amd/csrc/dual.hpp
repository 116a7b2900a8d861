// Forward-mode dual numbers for the gfx950 kernels (device only).
//
// Roles of the reference types (feos_torch/dual.py, feos_torch/dual_torch.py):
//   D2<T>      value + 1st + 2nd derivative along ONE direction     (Dual3, dual.py:5-78)
//   DN<T,N>    value + N first derivatives                          (parameter tangents; what
//              torch reverse mode delivers in the reference is delivered forward here)
//   T2<T>      value, gradient and Hessian in the two partial densities (supersedes DualTensor,
//              dual_torch.py:4-158, which carries only the volume-mixed second derivatives)
//   T1<T>      value and gradient in the two partial densities (with T = DN: parameter tangents)
// All are plain aggregates living in VGPRs; every loop over N is fully unrolled.  The types
// nest (D2<DN<double,3>> = d/drho, d2/drho2 and their parameter tangents) and mix with plain
// double on either side so model code is written once (pure_model.hpp, mix_model.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#define PCS_DEV __device__ __forceinline__

namespace pcs {

// ---- plain-double shims so generic code can call the same names ------------------------
PCS_DEV double re(double x) { return x; }
PCS_DEV double d_exp(double x) { return exp(x); }
// fp64 logarithm.  The library log() is a double-double evaluation (~100 VALU, 60 of them dependent v_add_f64 two-sums)
// for < 1 ulp; where PCS_FAST_LOG is defined (the pure-component unit) the textbook reduction is used instead:
// x = 2^e m, m in [sqrt(1/2), sqrt 2), s = (m-1)/(m+1), ln m = 2 s + s R(s^2) with the 7-term minimax R of the classical
// e_log (error of the polynomial 2^-58), ln x = e ln 2 + ln m.  ~30 VALU, <= 2 ulp (4.1e-16 relative, checked against
// numpy over 1e-13 .. 1e13 and 1 + 1e-12 .. 1 + 1e-3); arguments are finite, positive and normal (g_hs >= 1, site
// fractions in (0, 1]).
#ifdef PCS_FAST_LOG
PCS_DEV double d_recip(double x);
PCS_DEV double d_log(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    const bool low = m < 0.70710678118654752440;
    m = low ? 2.0 * m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * d_recip(m + 1.0);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = __builtin_fma(R, z, 1.531383769920937332e-01);
    R = __builtin_fma(R, z, 1.818357216161805012e-01);
    R = __builtin_fma(R, z, 2.222219843214978396e-01);
    R = __builtin_fma(R, z, 2.857142874366239149e-01);
    R = __builtin_fma(R, z, 3.999999999940941908e-01);
    R = __builtin_fma(R, z, 6.666666666666735130e-01);
    double r = __builtin_fma((double)e, 0.69314718055994530942, __builtin_fma(s, R * z, s + s));
#if PCS_FAST_LOG == 2
    // the mixture / gc solvers detect degenerate states by the IEEE results of the library log: keep them
    // (positive normal and subnormal arguments take the short form; frexp is exact on subnormals)
    if (!__builtin_amdgcn_class(x, 0x180)) r = (x == 0.0) ? -__builtin_inf() : (x > 0.0 ? x : __builtin_nan(""));
#else
    // pure-component unit: an iterate that overshoots the packing-fraction pole makes the argument negative; the solvers
    // recognise such states by a non-finite result (bit test), so the short form must not return a finite number there
    if (!__builtin_amdgcn_class(x, 0x380)) r = __longlong_as_double(0x7ff8000000000000LL);  // not +normal / +subnormal / +inf
#endif
    return r;
}
#else
PCS_DEV double d_log(double x) { return log(x); }
#endif
PCS_DEV double d_sqrt(double x) { return sqrt(x); }
PCS_DEV double d_cbrt(double x) { return cbrt(x); }
// fp64 reciprocal.  IEEE 1.0/x lowers to v_div_scale x2 + v_rcp_f64 + 5 FMA + v_div_fmas +
// v_div_fixup (~11 VALU); the kernels only ever divide well-scaled finite numbers, so the
// hardware estimate refined by two Newton steps (5 VALU, <= 1 ulp, measured 1.1e-16 over 1e-300..1e300)
// is used where PCS_FAST_RCP is defined: the pure-component translation unit (build.py).  It maps
// 0 and denormals to NaN instead of +-inf, which changes the path the mixture / gc solvers take
// through degenerate pure-component limits, so those units keep the IEEE division.
#ifdef PCS_FAST_RCP
PCS_DEV double d_recip(double x) {
    double r = __builtin_amdgcn_rcp(x);
#if PCS_FAST_RCP == 2
    // mixture / gc units: the refinement only where the estimate is an ordinary number, so that 1/0 = inf, 1/inf = 0 and
    // NaN propagate as with the IEEE division (the solvers walk through degenerate pure-component limits on them)
    const double r0 = r;
#endif
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#if PCS_FAST_RCP == 2
    if (!__builtin_amdgcn_class(r0, 0x108)) r = r0;  // not a +-normal number: keep the hardware result
#endif
    return r;
}
#else
PCS_DEV double d_recip(double x) { return 1.0 / x; }
#endif
// NaN / inf test that survives -fno-honor-nans style flags (pure bit test)
PCS_DEV bool is_finite_bits(double x) {
    return ((unsigned long long)__double_as_longlong(x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
}

// =========================================================================================
// D2<T>
// =========================================================================================
template <class T>
struct D2 {
    T v, d1, d2;
    PCS_DEV D2() {}
    PCS_DEV D2(double x) : v(x), d1(0.0), d2(0.0) {}
    PCS_DEV D2(const T& a, const T& b, const T& c) : v(a), d1(b), d2(c) {}
    // f(g): (f0, f1 g', f2 g'^2 + f1 g'')
    PCS_DEV D2 chain(const T& f0, const T& f1, const T& f2) const { return D2(f0, f1 * d1, f2 * (d1 * d1) + f1 * d2); }
};
template <class T> struct is_dual { static constexpr bool value = false; };
template <class T> struct is_dual<D2<T>> { static constexpr bool value = true; };

template <class T> PCS_DEV double re(const D2<T>& a) { return re(a.v); }
template <class T> PCS_DEV D2<T> operator+(const D2<T>& a, const D2<T>& b) { return D2<T>(a.v + b.v, a.d1 + b.d1, a.d2 + b.d2); }
template <class T> PCS_DEV D2<T> operator-(const D2<T>& a, const D2<T>& b) { return D2<T>(a.v - b.v, a.d1 - b.d1, a.d2 - b.d2); }
template <class T> PCS_DEV D2<T> operator-(const D2<T>& a) { return D2<T>(-a.v, -a.d1, -a.d2); }
template <class T> PCS_DEV D2<T> operator*(const D2<T>& a, const D2<T>& b) {
    return D2<T>(a.v * b.v, a.d1 * b.v + a.v * b.d1, a.d2 * b.v + 2.0 * (a.d1 * b.d1) + a.v * b.d2);
}
// scalar (double) on either side
template <class T> PCS_DEV D2<T> operator+(const D2<T>& a, double b) { return D2<T>(a.v + b, a.d1, a.d2); }
template <class T> PCS_DEV D2<T> operator+(double b, const D2<T>& a) { return D2<T>(a.v + b, a.d1, a.d2); }
template <class T> PCS_DEV D2<T> operator-(const D2<T>& a, double b) { return D2<T>(a.v - b, a.d1, a.d2); }
template <class T> PCS_DEV D2<T> operator-(double b, const D2<T>& a) { return D2<T>(b - a.v, -a.d1, -a.d2); }
template <class T> PCS_DEV D2<T> operator*(const D2<T>& a, double b) { return D2<T>(a.v * b, a.d1 * b, a.d2 * b); }
template <class T> PCS_DEV D2<T> operator*(double b, const D2<T>& a) { return D2<T>(a.v * b, a.d1 * b, a.d2 * b); }
template <class T> PCS_DEV D2<T> d_recip(const D2<T>& a) {
    T r = d_recip(a.v);
    T r2 = r * r;
    return a.chain(r, -r2, 2.0 * (r2 * r));
}
template <class T> PCS_DEV D2<T> operator/(const D2<T>& a, const D2<T>& b) { return a * d_recip(b); }
template <class T> PCS_DEV D2<T> operator/(const D2<T>& a, double b) { double r = 1.0 / b; return a * r; }
template <class T> PCS_DEV D2<T> operator/(double b, const D2<T>& a) { return d_recip(a) * b; }
template <class T> PCS_DEV D2<T> d_log(const D2<T>& a) { T r = d_recip(a.v); return a.chain(d_log(a.v), r, -(r * r)); }
template <class T> PCS_DEV D2<T> d_exp(const D2<T>& a) { T e = d_exp(a.v); return a.chain(e, e, e); }
template <class T> PCS_DEV D2<T> d_sqrt(const D2<T>& a) {
    T s = d_sqrt(a.v);
    T h = 0.5 * d_recip(s);              // 1/(2 sqrt x)
    return a.chain(s, h, -(h * d_recip(a.v)) * 0.5);  // f'' = -1/(4 x sqrt x)
}

// D2<T> (x) T for a dual component type T (e.g. D2<DN<double,2>> with DN<double,2> coefficients)
#define PCS_IFDUAL(T) std::enable_if_t<!std::is_same<T, double>::value, int> = 0
template <class T, PCS_IFDUAL(T)> PCS_DEV D2<T> operator*(const D2<T>& a, const T& b) { return D2<T>(a.v * b, a.d1 * b, a.d2 * b); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D2<T> operator*(const T& b, const D2<T>& a) { return D2<T>(a.v * b, a.d1 * b, a.d2 * b); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D2<T> operator+(const D2<T>& a, const T& b) { return D2<T>(a.v + b, a.d1, a.d2); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D2<T> operator+(const T& b, const D2<T>& a) { return D2<T>(a.v + b, a.d1, a.d2); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D2<T> operator-(const D2<T>& a, const T& b) { return D2<T>(a.v - b, a.d1, a.d2); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D2<T> operator-(const T& b, const D2<T>& a) { return D2<T>(b - a.v, -a.d1, -a.d2); }

// =========================================================================================
// D1<T>: value + first derivative along one direction over a component type T (the Jacobian kernels use
// D1<DN<double,C>>: a and da/drho with their parameter tangents)
// =========================================================================================
template <class T>
struct D1 {
    T v, d1;
    PCS_DEV D1() {}
    PCS_DEV D1(double x) : v(x), d1(0.0) {}
    PCS_DEV D1(const T& a, const T& b) : v(a), d1(b) {}
    PCS_DEV D1 chain(const T& f0, const T& f1) const { return D1(f0, f1 * d1); }
};
template <class T> struct is_dual<D1<T>> { static constexpr bool value = true; };
template <class T> PCS_DEV double re(const D1<T>& a) { return re(a.v); }
template <class T> PCS_DEV D1<T> operator+(const D1<T>& a, const D1<T>& b) { return D1<T>(a.v + b.v, a.d1 + b.d1); }
template <class T> PCS_DEV D1<T> operator-(const D1<T>& a, const D1<T>& b) { return D1<T>(a.v - b.v, a.d1 - b.d1); }
template <class T> PCS_DEV D1<T> operator-(const D1<T>& a) { return D1<T>(-a.v, -a.d1); }
template <class T> PCS_DEV D1<T> operator*(const D1<T>& a, const D1<T>& b) { return D1<T>(a.v * b.v, a.d1 * b.v + a.v * b.d1); }
template <class T> PCS_DEV D1<T> operator+(const D1<T>& a, double b) { return D1<T>(a.v + b, a.d1); }
template <class T> PCS_DEV D1<T> operator+(double b, const D1<T>& a) { return D1<T>(a.v + b, a.d1); }
template <class T> PCS_DEV D1<T> operator-(const D1<T>& a, double b) { return D1<T>(a.v - b, a.d1); }
template <class T> PCS_DEV D1<T> operator-(double b, const D1<T>& a) { return D1<T>(b - a.v, -a.d1); }
template <class T> PCS_DEV D1<T> operator*(const D1<T>& a, double b) { return D1<T>(a.v * b, a.d1 * b); }
template <class T> PCS_DEV D1<T> operator*(double b, const D1<T>& a) { return D1<T>(a.v * b, a.d1 * b); }
template <class T> PCS_DEV D1<T> d_recip(const D1<T>& a) { T r = d_recip(a.v); return a.chain(r, -(r * r)); }
template <class T> PCS_DEV D1<T> d_log(const D1<T>& a) { return a.chain(d_log(a.v), d_recip(a.v)); }
template <class T> PCS_DEV D1<T> d_sqrt(const D1<T>& a) { T sq = d_sqrt(a.v); return a.chain(sq, 0.5 * d_recip(sq)); }
template <class T> PCS_DEV D1<T> d_exp(const D1<T>& a) { T e = d_exp(a.v); return a.chain(e, e); }
template <class T> PCS_DEV D1<T> d_cbrt(const D1<T>& a) { T s = d_cbrt(a.v); return a.chain(s, s * d_recip(a.v) * (1.0 / 3.0)); }
template <class T> PCS_DEV D1<T> operator/(const D1<T>& a, const D1<T>& b) { return a * d_recip(b); }
template <class T> PCS_DEV D1<T> operator/(const D1<T>& a, double b) { double r = 1.0 / b; return a * r; }
template <class T> PCS_DEV D1<T> operator/(double b, const D1<T>& a) { return d_recip(a) * b; }
template <class T, PCS_IFDUAL(T)> PCS_DEV D1<T> operator*(const D1<T>& a, const T& b) { return D1<T>(a.v * b, a.d1 * b); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D1<T> operator*(const T& b, const D1<T>& a) { return D1<T>(a.v * b, a.d1 * b); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D1<T> operator+(const D1<T>& a, const T& b) { return D1<T>(a.v + b, a.d1); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D1<T> operator+(const T& b, const D1<T>& a) { return D1<T>(a.v + b, a.d1); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D1<T> operator-(const D1<T>& a, const T& b) { return D1<T>(a.v - b, a.d1); }
template <class T, PCS_IFDUAL(T)> PCS_DEV D1<T> operator-(const T& b, const D1<T>& a) { return D1<T>(b - a.v, -a.d1); }

// =========================================================================================
// D1s: value + first derivative along one direction, plain doubles (the solver's fp64 finish, which takes the
// second derivative from the fp32 pre-solve).  Uses the same d_recip(double) as D2<double>.
// =========================================================================================
struct D1s {
    double v, d1;
    PCS_DEV D1s() {}
    PCS_DEV D1s(double x) : v(x), d1(0.0) {}
    PCS_DEV D1s(double a, double b) : v(a), d1(b) {}
    PCS_DEV D1s chain(double f0, double f1) const { return D1s(f0, f1 * d1); }
};
template <> struct is_dual<D1s> { static constexpr bool value = true; };
PCS_DEV double re(const D1s& a) { return a.v; }
PCS_DEV D1s operator+(const D1s& a, const D1s& b) { return D1s(a.v + b.v, a.d1 + b.d1); }
PCS_DEV D1s operator-(const D1s& a, const D1s& b) { return D1s(a.v - b.v, a.d1 - b.d1); }
PCS_DEV D1s operator-(const D1s& a) { return D1s(-a.v, -a.d1); }
PCS_DEV D1s operator*(const D1s& a, const D1s& b) { return D1s(a.v * b.v, a.d1 * b.v + a.v * b.d1); }
PCS_DEV D1s operator+(const D1s& a, double b) { return D1s(a.v + b, a.d1); }
PCS_DEV D1s operator+(double b, const D1s& a) { return D1s(a.v + b, a.d1); }
PCS_DEV D1s operator-(const D1s& a, double b) { return D1s(a.v - b, a.d1); }
PCS_DEV D1s operator-(double b, const D1s& a) { return D1s(b - a.v, -a.d1); }
PCS_DEV D1s operator*(const D1s& a, double b) { return D1s(a.v * b, a.d1 * b); }
PCS_DEV D1s operator*(double b, const D1s& a) { return D1s(a.v * b, a.d1 * b); }
PCS_DEV D1s d_recip(const D1s& a) { double r = d_recip(a.v); return a.chain(r, -(r * r)); }
PCS_DEV D1s d_log(const D1s& a) { return a.chain(d_log(a.v), d_recip(a.v)); }
PCS_DEV D1s d_sqrt(const D1s& a) { double s = d_sqrt(a.v); return a.chain(s, 0.5 * d_recip(s)); }

// =========================================================================================
// DN<T,N>
// =========================================================================================
template <class T, int N>
struct DN {
    T v;
    T e[N];
    PCS_DEV DN() {}
    PCS_DEV DN(double x) : v(x) {
#pragma unroll
        for (int i = 0; i < N; i++) e[i] = T(0.0);
    }
    PCS_DEV DN chain(const T& f0, const T& f1) const {
        DN r;
        r.v = f0;
#pragma unroll
        for (int i = 0; i < N; i++) r.e[i] = f1 * e[i];
        return r;
    }
};
template <class T, int N> struct is_dual<DN<T, N>> { static constexpr bool value = true; };
template <class T, int N> PCS_DEV double re(const DN<T, N>& a) { return re(a.v); }
#define PCS_DN_LOOP _Pragma("unroll") for (int i = 0; i < N; i++)
template <class T, int N> PCS_DEV DN<T, N> operator+(const DN<T, N>& a, const DN<T, N>& b) { DN<T, N> r; r.v = a.v + b.v; PCS_DN_LOOP r.e[i] = a.e[i] + b.e[i]; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator-(const DN<T, N>& a, const DN<T, N>& b) { DN<T, N> r; r.v = a.v - b.v; PCS_DN_LOOP r.e[i] = a.e[i] - b.e[i]; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator-(const DN<T, N>& a) { DN<T, N> r; r.v = -a.v; PCS_DN_LOOP r.e[i] = -a.e[i]; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator*(const DN<T, N>& a, const DN<T, N>& b) { DN<T, N> r; r.v = a.v * b.v; PCS_DN_LOOP r.e[i] = a.e[i] * b.v + a.v * b.e[i]; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator+(const DN<T, N>& a, double b) { DN<T, N> r = a; r.v = a.v + b; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator+(double b, const DN<T, N>& a) { DN<T, N> r = a; r.v = a.v + b; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator-(const DN<T, N>& a, double b) { DN<T, N> r = a; r.v = a.v - b; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator-(double b, const DN<T, N>& a) { DN<T, N> r; r.v = b - a.v; PCS_DN_LOOP r.e[i] = -a.e[i]; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator*(const DN<T, N>& a, double b) { DN<T, N> r; r.v = a.v * b; PCS_DN_LOOP r.e[i] = a.e[i] * b; return r; }
template <class T, int N> PCS_DEV DN<T, N> operator*(double b, const DN<T, N>& a) { return a * b; }
// Parameter-tangent duals use the IEEE division: the fast reciprocal (<= 1 ulp) perturbs the
// implicit-differentiation Newton updates of the induced-association site fraction enough to
// shift dp/dT by ~1e-5 on those rows (measured, tests/test_gc_gpu.py); the gradient kernels are
// not on the headline path, so exactness wins here.
template <int N> PCS_DEV DN<double, N> d_recip(const DN<double, N>& a) { double r = 1.0 / a.v; return a.chain(r, -(r * r)); }
template <class T, int N, PCS_IFDUAL(T)> PCS_DEV DN<T, N> d_recip(const DN<T, N>& a) { T r = d_recip(a.v); return a.chain(r, -(r * r)); }
template <class T, int N> PCS_DEV DN<T, N> operator/(const DN<T, N>& a, const DN<T, N>& b) { return a * d_recip(b); }
template <class T, int N> PCS_DEV DN<T, N> operator/(const DN<T, N>& a, double b) { double r = 1.0 / b; return a * r; }
template <class T, int N> PCS_DEV DN<T, N> operator/(double b, const DN<T, N>& a) { return d_recip(a) * b; }
template <class T, int N> PCS_DEV DN<T, N> d_log(const DN<T, N>& a) { return a.chain(d_log(a.v), d_recip(a.v)); }
template <class T, int N> PCS_DEV DN<T, N> d_exp(const DN<T, N>& a) { T e = d_exp(a.v); return a.chain(e, e); }
template <class T, int N> PCS_DEV DN<T, N> d_sqrt(const DN<T, N>& a) { T s = d_sqrt(a.v); return a.chain(s, 0.5 * d_recip(s)); }
template <class T, int N> PCS_DEV DN<T, N> d_cbrt(const DN<T, N>& a) { T s = d_cbrt(a.v); return a.chain(s, s * d_recip(a.v) * (1.0 / 3.0)); }

// =========================================================================================
// T2<T>: second-order Taylor coefficients in TWO variables (the partial densities rho_1, rho_2):
//        v, g[2] = d/drho_i, h[3] = d2/drho_1^2, d2/drho_1 drho_2, d2/drho_2^2.
// Carries everything the bubble/dew Newton needs (chemical potentials, pressure and their full
// Jacobian); the reference's DualTensor (dual_torch.py:4-158) carries only the V-mixed part.
// =========================================================================================
template <class T>
struct T2 {
    T v, g0, g1, h00, h01, h11;
    PCS_DEV T2() {}
    PCS_DEV T2(double x) : v(x), g0(0.0), g1(0.0), h00(0.0), h01(0.0), h11(0.0) {}
    PCS_DEV T2(const T& a, const T& b, const T& c, const T& d, const T& e, const T& f) : v(a), g0(b), g1(c), h00(d), h01(e), h11(f) {}
    // f(u): value f0, f' = f1, f'' = f2
    PCS_DEV T2 chain(const T& f0, const T& f1, const T& f2) const {
        return T2(f0, f1 * g0, f1 * g1, f1 * h00 + f2 * (g0 * g0), f1 * h01 + f2 * (g0 * g1), f1 * h11 + f2 * (g1 * g1));
    }
};
template <class T> struct is_dual<T2<T>> { static constexpr bool value = true; };
template <class T> PCS_DEV double re(const T2<T>& a) { return re(a.v); }
template <class T> PCS_DEV T2<T> operator+(const T2<T>& a, const T2<T>& b) { return T2<T>(a.v + b.v, a.g0 + b.g0, a.g1 + b.g1, a.h00 + b.h00, a.h01 + b.h01, a.h11 + b.h11); }
template <class T> PCS_DEV T2<T> operator-(const T2<T>& a, const T2<T>& b) { return T2<T>(a.v - b.v, a.g0 - b.g0, a.g1 - b.g1, a.h00 - b.h00, a.h01 - b.h01, a.h11 - b.h11); }
template <class T> PCS_DEV T2<T> operator-(const T2<T>& a) { return T2<T>(-a.v, -a.g0, -a.g1, -a.h00, -a.h01, -a.h11); }
template <class T> PCS_DEV T2<T> operator*(const T2<T>& a, const T2<T>& b) {
    return T2<T>(a.v * b.v, a.g0 * b.v + a.v * b.g0, a.g1 * b.v + a.v * b.g1,
                 a.h00 * b.v + 2.0 * (a.g0 * b.g0) + a.v * b.h00,
                 a.h01 * b.v + a.g0 * b.g1 + a.g1 * b.g0 + a.v * b.h01,
                 a.h11 * b.v + 2.0 * (a.g1 * b.g1) + a.v * b.h11);
}
template <class T> PCS_DEV T2<T> operator+(const T2<T>& a, double b) { T2<T> r = a; r.v = a.v + b; return r; }
template <class T> PCS_DEV T2<T> operator+(double b, const T2<T>& a) { T2<T> r = a; r.v = a.v + b; return r; }
template <class T> PCS_DEV T2<T> operator-(const T2<T>& a, double b) { T2<T> r = a; r.v = a.v - b; return r; }
template <class T> PCS_DEV T2<T> operator-(double b, const T2<T>& a) { T2<T> r = -a; r.v = b - a.v; return r; }
template <class T> PCS_DEV T2<T> operator*(const T2<T>& a, double b) { return T2<T>(a.v * b, a.g0 * b, a.g1 * b, a.h00 * b, a.h01 * b, a.h11 * b); }
template <class T> PCS_DEV T2<T> operator*(double b, const T2<T>& a) { return a * b; }
template <class T> PCS_DEV T2<T> d_recip(const T2<T>& a) { T r = d_recip(a.v); T r2 = r * r; return a.chain(r, -r2, 2.0 * (r2 * r)); }
template <class T> PCS_DEV T2<T> operator/(const T2<T>& a, const T2<T>& b) { return a * d_recip(b); }
template <class T> PCS_DEV T2<T> operator/(const T2<T>& a, double b) { double r = 1.0 / b; return a * r; }
template <class T> PCS_DEV T2<T> operator/(double b, const T2<T>& a) { return d_recip(a) * b; }
template <class T> PCS_DEV T2<T> d_log(const T2<T>& a) { T r = d_recip(a.v); return a.chain(d_log(a.v), r, -(r * r)); }
template <class T> PCS_DEV T2<T> d_exp(const T2<T>& a) { T e = d_exp(a.v); return a.chain(e, e, e); }
template <class T> PCS_DEV T2<T> d_sqrt(const T2<T>& a) { T s = d_sqrt(a.v); T h = 0.5 * d_recip(s); return a.chain(s, h, -(h * d_recip(a.v)) * 0.5); }
template <class T> PCS_DEV T2<T> d_cbrt(const T2<T>& a) { T s = d_cbrt(a.v); T rx = d_recip(a.v); T f1 = s * rx * (1.0 / 3.0); return a.chain(s, f1, f1 * rx * (-2.0 / 3.0)); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T2<T> operator*(const T2<T>& a, const T& b) { return T2<T>(a.v * b, a.g0 * b, a.g1 * b, a.h00 * b, a.h01 * b, a.h11 * b); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T2<T> operator*(const T& b, const T2<T>& a) { return a * b; }
template <class T, PCS_IFDUAL(T)> PCS_DEV T2<T> operator+(const T2<T>& a, const T& b) { T2<T> r = a; r.v = a.v + b; return r; }
template <class T, PCS_IFDUAL(T)> PCS_DEV T2<T> operator+(const T& b, const T2<T>& a) { T2<T> r = a; r.v = a.v + b; return r; }
template <class T, PCS_IFDUAL(T)> PCS_DEV T2<T> operator-(const T2<T>& a, const T& b) { T2<T> r = a; r.v = a.v - b; return r; }
template <class T, PCS_IFDUAL(T)> PCS_DEV T2<T> operator-(const T& b, const T2<T>& a) { T2<T> r = -a; r.v = b - a.v; return r; }

// =========================================================================================
// T1<T>: first-order Taylor coefficients in the two partial densities (v, g0, g1).  With
// T = DN<double,C> it carries d(a, da/drho_i)/d(parameter) — what the implicit-function
// gradients of the bubble/dew pressure need.
// =========================================================================================
template <class T>
struct T1 {
    T v, g0, g1;
    PCS_DEV T1() {}
    PCS_DEV T1(double x) : v(x), g0(0.0), g1(0.0) {}
    PCS_DEV T1(const T& a, const T& b, const T& c) : v(a), g0(b), g1(c) {}
    PCS_DEV T1 chain(const T& f0, const T& f1) const { return T1(f0, f1 * g0, f1 * g1); }
};
template <class T> struct is_dual<T1<T>> { static constexpr bool value = true; };
template <class T> PCS_DEV double re(const T1<T>& a) { return re(a.v); }
template <class T> PCS_DEV T1<T> operator+(const T1<T>& a, const T1<T>& b) { return T1<T>(a.v + b.v, a.g0 + b.g0, a.g1 + b.g1); }
template <class T> PCS_DEV T1<T> operator-(const T1<T>& a, const T1<T>& b) { return T1<T>(a.v - b.v, a.g0 - b.g0, a.g1 - b.g1); }
template <class T> PCS_DEV T1<T> operator-(const T1<T>& a) { return T1<T>(-a.v, -a.g0, -a.g1); }
template <class T> PCS_DEV T1<T> operator*(const T1<T>& a, const T1<T>& b) { return T1<T>(a.v * b.v, a.g0 * b.v + a.v * b.g0, a.g1 * b.v + a.v * b.g1); }
template <class T> PCS_DEV T1<T> operator+(const T1<T>& a, double b) { return T1<T>(a.v + b, a.g0, a.g1); }
template <class T> PCS_DEV T1<T> operator+(double b, const T1<T>& a) { return T1<T>(a.v + b, a.g0, a.g1); }
template <class T> PCS_DEV T1<T> operator-(const T1<T>& a, double b) { return T1<T>(a.v - b, a.g0, a.g1); }
template <class T> PCS_DEV T1<T> operator-(double b, const T1<T>& a) { return T1<T>(b - a.v, -a.g0, -a.g1); }
template <class T> PCS_DEV T1<T> operator*(const T1<T>& a, double b) { return T1<T>(a.v * b, a.g0 * b, a.g1 * b); }
template <class T> PCS_DEV T1<T> operator*(double b, const T1<T>& a) { return a * b; }
template <class T> PCS_DEV T1<T> d_recip(const T1<T>& a) { T r = d_recip(a.v); return a.chain(r, -(r * r)); }
template <class T> PCS_DEV T1<T> operator/(const T1<T>& a, const T1<T>& b) { return a * d_recip(b); }
template <class T> PCS_DEV T1<T> operator/(const T1<T>& a, double b) { double r = 1.0 / b; return a * r; }
template <class T> PCS_DEV T1<T> operator/(double b, const T1<T>& a) { return d_recip(a) * b; }
template <class T> PCS_DEV T1<T> d_log(const T1<T>& a) { return a.chain(d_log(a.v), d_recip(a.v)); }
template <class T> PCS_DEV T1<T> d_exp(const T1<T>& a) { T e = d_exp(a.v); return a.chain(e, e); }
template <class T> PCS_DEV T1<T> d_sqrt(const T1<T>& a) { T s = d_sqrt(a.v); return a.chain(s, 0.5 * d_recip(s)); }
template <class T> PCS_DEV T1<T> d_cbrt(const T1<T>& a) { T s = d_cbrt(a.v); return a.chain(s, s * d_recip(a.v) * (1.0 / 3.0)); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T1<T> operator*(const T1<T>& a, const T& b) { return T1<T>(a.v * b, a.g0 * b, a.g1 * b); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T1<T> operator*(const T& b, const T1<T>& a) { return a * b; }
template <class T, PCS_IFDUAL(T)> PCS_DEV T1<T> operator+(const T1<T>& a, const T& b) { return T1<T>(a.v + b, a.g0, a.g1); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T1<T> operator+(const T& b, const T1<T>& a) { return T1<T>(a.v + b, a.g0, a.g1); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T1<T> operator-(const T1<T>& a, const T& b) { return T1<T>(a.v - b, a.g0, a.g1); }
template <class T, PCS_IFDUAL(T)> PCS_DEV T1<T> operator-(const T& b, const T1<T>& a) { return T1<T>(b - a.v, -a.g0, -a.g1); }

// ---- D2 (x) T2: functions of the FIRST coordinate only ---------------------------------------------------------------
// The solvers' evaluation takes its two-variable Taylor coefficients in the coordinates (zeta_3, rho_2) instead of
// (rho_1, rho_2) (mix_solver.hpp::phase_eval_inline).  Everything that depends on the packing fraction alone -- 1/(1 - zeta_3)
// and its powers, ln(1 - zeta_3), the dispersion and dipole polynomials, the pieces of C1 -- is then a function of the
// first coordinate only: value, first and second derivative (a D2) instead of six numbers, D2 x D2 products (6 flops) among
// themselves and 11-flop products with general T2 quantities instead of 17.  A D2 z stands for the T2 (z.v, z.d1, 0, z.d2, 0, 0).
template <class T> PCS_DEV T2<T> operator*(const T2<T>& a, const D2<T>& b) {
    return T2<T>(a.v * b.v, a.g0 * b.v + a.v * b.d1, a.g1 * b.v, a.h00 * b.v + 2.0 * (a.g0 * b.d1) + a.v * b.d2,
                 a.h01 * b.v + a.g1 * b.d1, a.h11 * b.v);
}
template <class T> PCS_DEV T2<T> operator*(const D2<T>& b, const T2<T>& a) { return a * b; }
template <class T> PCS_DEV T2<T> operator+(const T2<T>& a, const D2<T>& b) { return T2<T>(a.v + b.v, a.g0 + b.d1, a.g1, a.h00 + b.d2, a.h01, a.h11); }
template <class T> PCS_DEV T2<T> operator+(const D2<T>& b, const T2<T>& a) { return a + b; }
template <class T> PCS_DEV T2<T> operator-(const T2<T>& a, const D2<T>& b) { return T2<T>(a.v - b.v, a.g0 - b.d1, a.g1, a.h00 - b.d2, a.h01, a.h11); }
template <class T> PCS_DEV T2<T> operator-(const D2<T>& b, const T2<T>& a) { return T2<T>(b.v - a.v, b.d1 - a.g0, -a.g1, b.d2 - a.h00, -a.h01, -a.h11); }

// second-order <-> first-order two-variable types (drop / zero the Hessian part)
template <class T> PCS_DEV T1<T> lower(const T2<T>& a) { return T1<T>(a.v, a.g0, a.g1); }
template <class T> PCS_DEV T2<T> raise(const T1<T>& a) { return T2<T>(a.v, a.g0, a.g1, T(0.0), T(0.0), T(0.0)); }

// ---- lifting a parameter-type value P into a result type R -------------------------------
// Model code is templated on <P, R>: P is the type of parameters / temperature-only
// coefficients (double in the solvers, a dual in the gradient kernels), R the type of the
// density (D2<double> in the solvers).  Either P is double, or P == R.
template <class R, class P> struct Lift;
template <class R> struct Lift<R, R> { static PCS_DEV const R& go(const R& p) { return p; } };
template <class T> struct Lift<D2<T>, double> { static PCS_DEV D2<T> go(double p) { return D2<T>(p); } };
template <class T, int N> struct Lift<DN<T, N>, double> { static PCS_DEV DN<T, N> go(double p) { return DN<T, N>(p); } };
template <class T> struct Lift<D1<T>, T> { static PCS_DEV D1<T> go(const T& p) { return D1<T>(p, T(0.0)); } };
template <class T> struct Lift<T1<T>, T> { static PCS_DEV T1<T> go(const T& p) { return T1<T>(p, T(0.0), T(0.0)); } };
template <class T> struct Lift<T2<T>, double> { static PCS_DEV T2<T> go(double p) { return T2<T>(p); } };
template <class T, int N> struct Lift<T2<DN<T, N>>, DN<T, N>> {
    static PCS_DEV T2<DN<T, N>> go(const DN<T, N>& p) { const DN<T, N> z(0.0); return T2<DN<T, N>>(p, z, z, z, z, z); }
};
template <class T, int N> struct Lift<D2<DN<T, N>>, DN<T, N>> {
    static PCS_DEV D2<DN<T, N>> go(const DN<T, N>& p) { const DN<T, N> z(0.0); return D2<DN<T, N>>(p, z, z); }
};

}  // namespace pcs
