// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called from the
// product path (feos_torch_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of the reference's forward-mode dual numbers:
//   Dual3      <- feos_torch/dual.py:5-78        (value, d/dx, d2/dx2 of one scalar)
//   HyperDual  <- feos_torch/dual_torch.py:4-158 (DualTensor: re, eps1[N], eps2, eps1eps2[N])
//   Dual2      <- feos_torch/dual_torch.py:165-208 (first order, two directions)
//   DualN      -- first-order N-direction dual (stands in for torch reverse mode when the
//                 oracle needs d(result)/d(parameters); not a reference type)
// All types are generic over the component type T so they nest (e.g. Dual3<DualN<double,9>>
// gives d2a/drho dtheta) and so the same model code runs in double and long double.
#pragma once
#include <cmath>
#include <type_traits>

namespace oracle {

// ---- scalar helpers so generic code can call re()/log()/exp()/sqrt() on plain floats
inline double re(double x) { return x; }
inline long double re(long double x) { return x; }
inline double log(double x) { return std::log(x); }
inline double exp(double x) { return std::exp(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double cbrt(double x) { return std::cbrt(x); }
inline long double log(long double x) { return std::log(x); }
inline long double exp(long double x) { return std::exp(x); }
inline long double sqrt(long double x) { return std::sqrt(x); }
inline long double cbrt(long double x) { return std::cbrt(x); }

// Site fractions of the two-site-type association model (feos_torch/pcsaft_pure.py:172-175, pcsaft_mix.py:235-238):
//   xa = 2/(sq + 1 + t),  xb = 2/(sq + 1 - t),  t = (rhob - rhoa) delta,  sq = sqrt((1 - t)^2 + 4 rhob delta).
// One of the two denominators cancels when |t| >> 1 (strong association at low T: t ~ 1e10 loses 10 digits, i.e. 1e-7
// relative in double and still 1e-10 in x87 long double -- found in round 2 with a 50-digit mpmath referee,
// tests/tools/mp_pure_check.py).  The plain-double instantiations keep the formula AS WRITTEN (they play the role of the
// reference's Python); the long-double instantiations, which are the parity reference, use the algebraically identical
// conjugate forms ((sq+1+t)(sq-1-t) = 4 rhoa delta, (sq+1-t)(sq-1+t) = 4 rhob delta) so that they are exact to ~1e-18.
template <class S>
void site_fractions_two_types(const S& rhoa, const S& rhob, const S& delta, S& xa, S& xb) {
    S t = (rhob - rhoa) * delta;
    S aux = 1.0 - t;
    S sq = sqrt(aux * aux + 4.0 * rhob * delta);
    if constexpr (std::is_same<decltype(re(rhoa)), long double>::value) {
        const long double tr = re(t);
        if (tr > 0.5L) {
            xa = 2.0 / (sq + 1.0 + t);
            xb = (sq - 1.0 + t) / (2.0 * (rhob * delta));
            return;
        }
        if (tr < -0.5L) {
            xa = (sq - 1.0 - t) / (2.0 * (rhoa * delta));
            xb = 2.0 / (sq + 1.0 - t);
            return;
        }
    }
    xa = 2.0 / (sq + 1.0 + (rhob - rhoa) * delta);
    xb = 2.0 / (sq + 1.0 - (rhob - rhoa) * delta);
}

// ------------------------------------------------------------------------------------
// Dual3: feos_torch/dual.py:5-78
// ------------------------------------------------------------------------------------
template <class T>
struct Dual3 {
    T re, v1, v2;
    Dual3() : re(0), v1(0), v2(0) {}
    Dual3(double x) : re(x), v1(0), v2(0) {}
    Dual3(T r, T a, T b) : re(r), v1(a), v2(b) {}
    static Dual3 diff(T x) { return Dual3(x, T(1.0), T(0.0)); }  // dual.py:12-13
    // dual.py:48-49
    Dual3 chain_rule(T f0, T f1, T f2) const { return Dual3(f0, f1 * v1, f2 * v1 * v1 + f1 * v2); }
    // dual.py:51-53
    Dual3 recip() const {
        T rec = T(1.0) / re;
        return chain_rule(rec, -(rec * rec), T(2.0) * rec * rec * rec);
    }
};
template <class T> auto re(const Dual3<T>& a) { return re(a.re); }
template <class T> Dual3<T> operator+(const Dual3<T>& a, const Dual3<T>& b) { return {a.re + b.re, a.v1 + b.v1, a.v2 + b.v2}; }
template <class T> Dual3<T> operator+(const Dual3<T>& a, double b) { return {a.re + b, a.v1, a.v2}; }
template <class T> Dual3<T> operator+(double b, const Dual3<T>& a) { return a + b; }
template <class T> Dual3<T> operator-(const Dual3<T>& a) { return {-a.re, -a.v1, -a.v2}; }
template <class T> Dual3<T> operator-(const Dual3<T>& a, const Dual3<T>& b) { return {a.re - b.re, a.v1 - b.v1, a.v2 - b.v2}; }
template <class T> Dual3<T> operator-(const Dual3<T>& a, double b) { return {a.re - b, a.v1, a.v2}; }
template <class T> Dual3<T> operator-(double b, const Dual3<T>& a) { return b + (-a); }  // dual.py:36-37
// dual.py:39-46
template <class T> Dual3<T> operator*(const Dual3<T>& a, const Dual3<T>& b) {
    return {a.re * b.re, a.v1 * b.re + a.re * b.v1, a.v2 * b.re + T(2.0) * a.v1 * b.v1 + a.re * b.v2};
}
template <class T> Dual3<T> operator*(const Dual3<T>& a, double b) { return {a.re * b, a.v1 * b, a.v2 * b}; }
template <class T> Dual3<T> operator*(double b, const Dual3<T>& a) { return a * b; }
// dual.py:55-61
template <class T> Dual3<T> operator/(const Dual3<T>& a, const Dual3<T>& b) { return a * b.recip(); }
template <class T> Dual3<T> operator/(const Dual3<T>& a, double b) { return {a.re / b, a.v1 / b, a.v2 / b}; }
template <class T> Dual3<T> operator/(double b, const Dual3<T>& a) { return a.recip() * b; }
// dual.py:63-74
template <class T> Dual3<T> log(const Dual3<T>& a) {
    T rec = T(1.0) / a.re;
    return a.chain_rule(log(a.re), rec, -(rec * rec));
}
template <class T> Dual3<T> exp(const Dual3<T>& a) {
    T e = exp(a.re);
    return a.chain_rule(e, e, e);
}
template <class T> Dual3<T> sqrt(const Dual3<T>& a) {
    T s = sqrt(a.re);
    return a.chain_rule(s, T(0.5) / s, T(-0.25) / (s * s * s));
}

// ------------------------------------------------------------------------------------
// DualN: first-order dual with N directions (oracle-only helper for parameter gradients)
// ------------------------------------------------------------------------------------
template <class T, int N>
struct DualN {
    T re;
    T eps[N];
    DualN() : re(0) { for (int i = 0; i < N; i++) eps[i] = T(0); }
    DualN(double x) : re(x) { for (int i = 0; i < N; i++) eps[i] = T(0); }
    static DualN var(T x, int k) { DualN r; r.re = x; r.eps[k] = T(1); return r; }
    DualN chain(T f0, T f1) const { DualN r; r.re = f0; for (int i = 0; i < N; i++) r.eps[i] = f1 * eps[i]; return r; }
};
template <class T, int N> auto re(const DualN<T, N>& a) { return re(a.re); }
template <class T, int N> DualN<T, N> operator+(const DualN<T, N>& a, const DualN<T, N>& b) { DualN<T, N> r; r.re = a.re + b.re; for (int i = 0; i < N; i++) r.eps[i] = a.eps[i] + b.eps[i]; return r; }
template <class T, int N> DualN<T, N> operator+(const DualN<T, N>& a, double b) { DualN<T, N> r = a; r.re = a.re + b; return r; }
template <class T, int N> DualN<T, N> operator+(double b, const DualN<T, N>& a) { return a + b; }
template <class T, int N> DualN<T, N> operator-(const DualN<T, N>& a) { DualN<T, N> r; r.re = -a.re; for (int i = 0; i < N; i++) r.eps[i] = -a.eps[i]; return r; }
template <class T, int N> DualN<T, N> operator-(const DualN<T, N>& a, const DualN<T, N>& b) { return a + (-b); }
template <class T, int N> DualN<T, N> operator-(const DualN<T, N>& a, double b) { return a + (-b); }
template <class T, int N> DualN<T, N> operator-(double b, const DualN<T, N>& a) { return (-a) + b; }
template <class T, int N> DualN<T, N> operator*(const DualN<T, N>& a, const DualN<T, N>& b) { DualN<T, N> r; r.re = a.re * b.re; for (int i = 0; i < N; i++) r.eps[i] = a.eps[i] * b.re + a.re * b.eps[i]; return r; }
template <class T, int N> DualN<T, N> operator*(const DualN<T, N>& a, double b) { DualN<T, N> r; r.re = a.re * b; for (int i = 0; i < N; i++) r.eps[i] = a.eps[i] * b; return r; }
template <class T, int N> DualN<T, N> operator*(double b, const DualN<T, N>& a) { return a * b; }
template <class T, int N> DualN<T, N> recip(const DualN<T, N>& a) { T rec = T(1.0) / a.re; return a.chain(rec, -(rec * rec)); }
template <class T, int N> DualN<T, N> operator/(const DualN<T, N>& a, const DualN<T, N>& b) { return a * recip(b); }
template <class T, int N> DualN<T, N> operator/(const DualN<T, N>& a, double b) { return a * (1.0 / b); }
template <class T, int N> DualN<T, N> operator/(double b, const DualN<T, N>& a) { return recip(a) * b; }
template <class T, int N> DualN<T, N> log(const DualN<T, N>& a) { return a.chain(log(a.re), T(1.0) / a.re); }
template <class T, int N> DualN<T, N> exp(const DualN<T, N>& a) { T e = exp(a.re); return a.chain(e, e); }
template <class T, int N> DualN<T, N> sqrt(const DualN<T, N>& a) { T s = sqrt(a.re); return a.chain(s, T(0.5) / s); }
template <class T, int N> DualN<T, N> cbrt(const DualN<T, N>& a) { T s = cbrt(a.re); return a.chain(s, s / (T(3.0) * a.re)); }

// ------------------------------------------------------------------------------------
// HyperDual: feos_torch/dual_torch.py:4-158 (DualTensor), one batch element.
//   eps1[N]      : N first-order directions (mole numbers N_1..N_n, then volume)
//   eps2         : one first-order direction (volume)
//   eps1eps2[N]  : mixed second derivatives
// ------------------------------------------------------------------------------------
template <class T, int N>
struct HyperDual {
    T re;
    T eps1[N];
    T eps2;
    T eps1eps2[N];
    HyperDual() : re(0), eps2(0) { for (int i = 0; i < N; i++) { eps1[i] = T(0); eps1eps2[i] = T(0); } }
    HyperDual(double x) : re(x), eps2(0) { for (int i = 0; i < N; i++) { eps1[i] = T(0); eps1eps2[i] = T(0); } }
    // dual_torch.py:109-117
    HyperDual chain_rule(T f0, T f1, T f2) const {
        HyperDual r;
        r.re = f0;
        r.eps2 = f1 * eps2;
        for (int i = 0; i < N; i++) {
            r.eps1[i] = f1 * eps1[i];
            r.eps1eps2[i] = f1 * eps1eps2[i] + f2 * eps1[i] * eps2;
        }
        return r;
    }
    // dual_torch.py:119-122
    HyperDual recip() const {
        T rec = T(1.0) / re;
        T rec2 = rec * rec;
        return chain_rule(rec, -rec2, T(2.0) * rec2 * rec);
    }
};
template <class T, int N> auto re(const HyperDual<T, N>& a) { return re(a.re); }
template <class T, int N> HyperDual<T, N> operator+(const HyperDual<T, N>& a, const HyperDual<T, N>& b) {
    HyperDual<T, N> r; r.re = a.re + b.re; r.eps2 = a.eps2 + b.eps2;
    for (int i = 0; i < N; i++) { r.eps1[i] = a.eps1[i] + b.eps1[i]; r.eps1eps2[i] = a.eps1eps2[i] + b.eps1eps2[i]; }
    return r;
}
template <class T, int N> HyperDual<T, N> operator+(const HyperDual<T, N>& a, double b) { HyperDual<T, N> r = a; r.re = a.re + b; return r; }
template <class T, int N> HyperDual<T, N> operator+(double b, const HyperDual<T, N>& a) { return a + b; }
template <class T, int N> HyperDual<T, N> operator-(const HyperDual<T, N>& a) {
    HyperDual<T, N> r; r.re = -a.re; r.eps2 = -a.eps2;
    for (int i = 0; i < N; i++) { r.eps1[i] = -a.eps1[i]; r.eps1eps2[i] = -a.eps1eps2[i]; }
    return r;
}
template <class T, int N> HyperDual<T, N> operator-(const HyperDual<T, N>& a, const HyperDual<T, N>& b) { return a + (-b); }
template <class T, int N> HyperDual<T, N> operator-(const HyperDual<T, N>& a, double b) { return a + (-b); }
template <class T, int N> HyperDual<T, N> operator-(double b, const HyperDual<T, N>& a) { return b + (-a); }
// dual_torch.py:80-107
template <class T, int N> HyperDual<T, N> operator*(const HyperDual<T, N>& a, const HyperDual<T, N>& b) {
    HyperDual<T, N> r;
    r.re = a.re * b.re;
    r.eps2 = a.re * b.eps2 + b.re * a.eps2;
    for (int i = 0; i < N; i++) {
        r.eps1[i] = a.re * b.eps1[i] + b.re * a.eps1[i];
        r.eps1eps2[i] = a.re * b.eps1eps2[i] + a.eps1[i] * b.eps2 + a.eps2 * b.eps1[i] + a.eps1eps2[i] * b.re;
    }
    return r;
}
template <class T, int N> HyperDual<T, N> operator*(const HyperDual<T, N>& a, double b) {
    HyperDual<T, N> r; r.re = a.re * b; r.eps2 = a.eps2 * b;
    for (int i = 0; i < N; i++) { r.eps1[i] = a.eps1[i] * b; r.eps1eps2[i] = a.eps1eps2[i] * b; }
    return r;
}
template <class T, int N> HyperDual<T, N> operator*(double b, const HyperDual<T, N>& a) { return a * b; }
// dual_torch.py:124-145
template <class T, int N> HyperDual<T, N> operator/(const HyperDual<T, N>& a, const HyperDual<T, N>& b) { return a * b.recip(); }
template <class T, int N> HyperDual<T, N> operator/(const HyperDual<T, N>& a, double b) {
    HyperDual<T, N> r; r.re = a.re / b; r.eps2 = a.eps2 / b;
    for (int i = 0; i < N; i++) { r.eps1[i] = a.eps1[i] / b; r.eps1eps2[i] = a.eps1eps2[i] / b; }
    return r;
}
template <class T, int N> HyperDual<T, N> operator/(double b, const HyperDual<T, N>& a) { return a.recip() * b; }
// dual_torch.py:147-158
template <class T, int N> HyperDual<T, N> log(const HyperDual<T, N>& a) { T rec = T(1.0) / a.re; return a.chain_rule(log(a.re), rec, -rec * rec); }
template <class T, int N> HyperDual<T, N> exp(const HyperDual<T, N>& a) { T e = exp(a.re); return a.chain_rule(e, e, e); }
template <class T, int N> HyperDual<T, N> sqrt(const HyperDual<T, N>& a) { T s = sqrt(a.re); return a.chain_rule(s, T(0.5) / s, T(-0.25) / s / a.re); }
template <class T, int N> HyperDual<T, N> cbrt(const HyperDual<T, N>& a) {
    T s = cbrt(a.re);  // x^(1/3): f1 = s/(3x), f2 = -2 s/(9 x^2)   (torch .pow(1/3) on a plain tensor in the reference)
    return a.chain_rule(s, s / (T(3.0) * a.re), T(-2.0) * s / (T(9.0) * a.re * a.re));
}

// ------------------------------------------------------------------------------------
// Dual2: feos_torch/dual_torch.py:165-208 — Jacobian carrier for the association Newton.
// ------------------------------------------------------------------------------------
template <class T>
struct Dual2 {
    T re, eps1, eps2;
    Dual2(T r, T a, T b) : re(r), eps1(a), eps2(b) {}
};
template <class T> Dual2<T> operator*(const Dual2<T>& a, const Dual2<T>& b) { return {a.re * b.re, a.re * b.eps1 + a.eps1 * b.re, a.re * b.eps2 + a.eps2 * b.re}; }
template <class T> Dual2<T> operator*(const Dual2<T>& a, const T& b) { return {a.re * b, a.eps1 * b, a.eps2 * b}; }
template <class T> Dual2<T> operator*(const T& b, const Dual2<T>& a) { return a * b; }
template <class T> Dual2<T> operator/(const Dual2<T>& a, const Dual2<T>& b) {
    return {a.re / b.re, (a.eps1 * b.re - a.re * b.eps1) / (b.re * b.re), (a.eps2 * b.re - a.re * b.eps2) / (b.re * b.re)};
}
template <class T> Dual2<T> operator+(const Dual2<T>& a, const Dual2<T>& b) { return {a.re + b.re, a.eps1 + b.eps1, a.eps2 + b.eps2}; }
template <class T> Dual2<T> operator+(const Dual2<T>& a, const T& b) { return {a.re + b, a.eps1, a.eps2}; }
template <class T> Dual2<T> operator+(const T& b, const Dual2<T>& a) { return a + b; }
template <class T> Dual2<T> operator-(const Dual2<T>& a, const Dual2<T>& b) { return {a.re - b.re, a.eps1 - b.eps1, a.eps2 - b.eps2}; }
template <class T> Dual2<T> operator-(const Dual2<T>& a, const T& b) { return {a.re - b, a.eps1, a.eps2}; }

}  // namespace oracle
