// fp32 pre-solve of the pure-component VLE (device only).
//
// On gfx950 an fp32 VALU op issues in half the cycles of an fp64 one and 1/x, log, sqrt are single
// (quarter-rate) instructions instead of ~5 / ~45 / ~20-instruction fp64 sequences, so one fp32
// evaluation of a(rho), a', a'' costs ~0.35 of the fp64 one.  Newton's method is self-correcting:
// the zero-pressure liquid root, the ideal-gas vapour estimate and the first coupled iterations
// are therefore run in fp32 down to its noise floor (~1e-6 relative), and the fp64 iteration of
// pure_solver.hpp starts from that point and needs 1-2 iterations instead of 4-5 + initialiser.
// The result is defined by the fp64 iterations alone; a lane whose fp32 pass leaves the
// representable range or misbehaves simply takes the all-fp64 path.
#pragma once
#include "pure_model.hpp"

namespace pcs {

constexpr float PCS_F32_TAYLOR_MAX = 1e-3f;
constexpr float PCS_F32_PREDICT_TOL_L = 5e-6f;
constexpr float PCS_F32_PREDICT_TOL_V = 5e-5f;
constexpr float PCS_F32_PREDICT_CMAX = 1e3f;
constexpr int PCS_F32_DENSE_LEVELS = 3;  // restarts of the fp32 liquid root on the dense side (1 = eta 0.58 only)
constexpr float PCS_F32_LIQ_TOL = 1e-1f;  // relative (scaled-Newton) step at which the fp32 liquid initialiser hands over to the coupled iteration

struct F2 {  // value, d/drho, d2/drho2 in fp32
    float v, d1, d2;
};
PCS_DEV F2 f2(float v, float d1, float d2) { F2 r; r.v = v; r.d1 = d1; r.d2 = d2; return r; }
PCS_DEV F2 operator+(F2 a, F2 b) { return f2(a.v + b.v, a.d1 + b.d1, a.d2 + b.d2); }
PCS_DEV F2 operator-(F2 a, F2 b) { return f2(a.v - b.v, a.d1 - b.d1, a.d2 - b.d2); }
PCS_DEV F2 operator+(F2 a, float b) { return f2(a.v + b, a.d1, a.d2); }
PCS_DEV F2 operator-(float b, F2 a) { return f2(b - a.v, -a.d1, -a.d2); }
PCS_DEV F2 operator*(F2 a, float b) { return f2(a.v * b, a.d1 * b, a.d2 * b); }
PCS_DEV F2 operator*(F2 a, F2 b) {
    return f2(a.v * b.v, fmaf(a.d1, b.v, a.v * b.d1), fmaf(a.d2, b.v, fmaf(2.0f * a.d1, b.d1, a.v * b.d2)));
}
PCS_DEV F2 chainf(F2 a, float f0, float f1, float f2_) { return f2(f0, f1 * a.d1, fmaf(f2_, a.d1 * a.d1, f1 * a.d2)); }
// raw v_log_f32 / v_exp_f32 (the library __logf / __expf wrap them in a denormal rescue: v_cmp + v_cndmask + v_ldexp,
// ~5 VALU each).  Arguments here are normal fp32 numbers or the lane falls back to the fp64 path (non-finite result).
PCS_DEV float f_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }
PCS_DEV float f_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
PCS_DEV F2 recipf(F2 a) {
    float r = __builtin_amdgcn_rcpf(a.v);
    float r2 = r * r;
    return chainf(a, r, -r2, 2.0f * r2 * r);
}
PCS_DEV F2 logf2(F2 a) {
    float r = __builtin_amdgcn_rcpf(a.v);
    return chainf(a, f_log(a.v), r, -r * r);
}
PCS_DEV F2 sqrtf2(F2 a) {
    float s = __builtin_amdgcn_sqrtf(a.v);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    return chainf(a, s, h, -0.5f * h * __builtin_amdgcn_rcpf(a.v));
}
template <int N>
PCS_DEV F2 hornerf(const float* coef, F2 x) {  // x.d2 == 0 (x = eta = ceta * rho)
    float p = coef[N - 1], d1 = 0.0f, d2 = 0.0f;
#pragma unroll
    for (int i = N - 2; i >= 0; i--) {
        d2 = fmaf(d2, x.v, d1);
        d1 = fmaf(d1, x.v, p);
        p = fmaf(p, x.v, coef[i]);
    }
    return f2(p, d1 * x.d1, 2.0f * d2 * (x.d1 * x.d1));
}

struct PureCoefF {
    float m, mm1, ceta, ai[7], bi[7], kd1, kd2, j1[5], j2[4], qm, da, na, nb;
    bool polar, assoc;
};

PCS_DEV void to_f32(const PureCoef<double>& c, PureCoefF& f) {
    f.m = (float)c.m; f.mm1 = (float)c.mm1; f.ceta = (float)c.ceta;
#pragma unroll
    for (int i = 0; i < 7; i++) { f.ai[i] = (float)c.ai[i]; f.bi[i] = (float)c.bi[i]; }
    f.kd1 = (float)c.kd1; f.kd2 = (float)c.kd2;
    f.polar = c.polar; f.assoc = c.assoc;
    if (c.polar) {
#pragma unroll
        for (int i = 0; i < 5; i++) f.j1[i] = (float)c.j1[i];
#pragma unroll
        for (int i = 0; i < 4; i++) f.j2[i] = (float)c.j2[i];
        f.qm = (float)c.qm;
    }
    f.da = (float)c.da; f.na = (float)c.na; f.nb = (float)c.nb;
}

// Native fp32 version of pure_coef() (pure_model.hpp): the pre-solve needs its coefficients to ~1e-6 only, and
// computing them from the fp32 parameters keeps the 33 fp64 coefficients out of the registers until the fp64
// finish needs them.
PCS_DEV void pure_coef_f32(PureCoefF& f, const double* par, double T64) {
    const float m = (float)par[0], sigma = (float)par[1], eps = (float)par[2], mu = (float)par[3];
    const float rT = __builtin_amdgcn_rcpf((float)T64);
    const float s3 = sigma * sigma * sigma;
    const float e = eps * rT;
    const float d = sigma * (1.0f - 0.12f * f_exp(-3.0f * e));
    f.m = m;
    f.mm1 = m - 1.0f;
    f.ceta = (float)FRAC_PI_6 * (m * (d * d * d));
    const float rm = __builtin_amdgcn_rcpf(m);
    const float m1 = f.mm1 * rm;
    const float m2 = (m - 2.0f) * rm;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        f.ai[i] = fmaf(m1, fmaf(m2, (float)A2[i], (float)A1[i]), (float)A0[i]);
        f.bi[i] = fmaf(m1, fmaf(m2, (float)B2[i], (float)B1[i]), (float)B0[i]);
    }
    const float pref = (float)(-PI) * ((m * m) * (e * s3));
    f.kd1 = 2.0f * pref;
    f.kd2 = pref * (m * e);
    f.polar = mu != 0.0f;
    f.qm = 0.0f;
    if (f.polar) {
        const float mu2t = (mu * mu) * (rm * rT) * (float)MU2_UNIT;
        const bool clamp = m > 2.0f;
        const float md1 = clamp ? 0.5f : m1;
        const float md2 = clamp ? 0.0f : md1 * m2;
        const float rs3 = __builtin_amdgcn_rcpf(s3);
        const float f2c = (float)(-PI) * rs3;
        const float f3c = (float)(-PI_SQ_43) * (rs3 * mu2t);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            float a = (float)AD[i][0] + md1 * (float)AD[i][1] + md2 * (float)AD[i][2];
            if (i < 3) a = a + ((float)BD[i][0] + md1 * (float)BD[i][1] + md2 * (float)BD[i][2]) * e;
            f.j1[i] = a * f2c;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) f.j2[i] = ((float)CD[i][0] + md1 * (float)CD[i][1] + md2 * (float)CD[i][2]) * f3c;
        f.qm = mu2t * mu2t;
    }
    f.na = (float)par[6];
    f.nb = (float)par[7];
    const bool sites = (f.na != 0.0f) || (f.nb != 0.0f);
    f.da = (f_exp((float)par[5] * rT) - 1.0f) * s3 * (float)par[4];
    f.assoc = sites && f.da != 0.0f;
}

struct EvalF { float a, p, dp, mu; };

// Hard sphere, hard chain and dispersion of a pure component are a = rho F(eta) + rho^2 G(eta), eta = ceta rho, with
//   F = m HS - (m-1) ln g,   HS = (4 eta - 3 eta^2) u^2,   g = (1 - eta/2) u^3,   u = 1/(1-eta)
//   G = kd1 I1 + kd2 C I2,   C = 1/D,  D = 1 + m A - (m-1) B,  A = (8 eta - 2 eta^2) u^4,  B = poly u^2 w^2,  w = 1/(2-eta)
// Their first and second eta-derivatives in closed form (checked symbolically) cost about a third of the generic
// value/d1/d2 arithmetic: HS' = (4-2eta)u^3, HS'' = (10-4eta)u^4, (ln g)' = 3u - w, (ln g)'' = 3u^2 - w^2,
// A' = (8+20eta-4eta^2)u^5, A'' = (60+72eta-12eta^2)u^6, and with q = u^2 w^2, s = u + w:
// B' = q (poly' + 2 poly s), B'' = 2 q s (poly' + 2 poly s) + q (poly'' + 2 poly' s + 2 poly (u^2 + w^2)).
template <int N>
PCS_DEV void horner3f(const float* coef, float x, float& p, float& d1, float& d2) {  // p, p', p''
    p = coef[N - 1];
    d1 = 0.0f;
    float h = 0.0f;
#pragma unroll
    for (int i = N - 2; i >= 0; i--) {
        h = fmaf(h, x, d1);
        d1 = fmaf(d1, x, p);
        p = fmaf(p, x, coef[i]);
    }
    d2 = 2.0f * h;
}
PCS_DEV F2 core_closed_f32(const PureCoefF& c, float rho) {
    const float eta = rho * c.ceta;
    const float u = __builtin_amdgcn_rcpf(1.0f - eta), w = __builtin_amdgcn_rcpf(2.0f - eta);
    const float u2 = u * u, u3 = u2 * u, u4 = u2 * u2, w2 = w * w;
    const float HS = eta * (4.0f - 3.0f * eta) * u2, HS1 = (4.0f - 2.0f * eta) * u3, HS2 = (10.0f - 4.0f * eta) * u4;
    const float LG = f_log((1.0f - 0.5f * eta) * u3), LG1 = 3.0f * u - w, LG2 = 3.0f * u2 - w2;
    const float F = c.m * HS - c.mm1 * LG, F1 = c.m * HS1 - c.mm1 * LG1, F2_ = c.m * HS2 - c.mm1 * LG2;
    float I1, I1a, I1b, I2, I2a, I2b;
    horner3f<7>(c.ai, eta, I1, I1a, I1b);
    horner3f<7>(c.bi, eta, I2, I2a, I2b);
    const float A = eta * (8.0f - 2.0f * eta) * u4, A1 = (8.0f + eta * (20.0f - 4.0f * eta)) * (u4 * u),
                A2 = (60.0f + eta * (72.0f - 12.0f * eta)) * (u4 * u2);
    const float poly = eta * (20.0f + eta * (-27.0f + eta * (12.0f - 2.0f * eta)));
    const float poly1 = 20.0f + eta * (-54.0f + eta * (36.0f - 8.0f * eta)), poly2 = -54.0f + eta * (72.0f - 24.0f * eta);
    const float q = u2 * w2, s = u + w;
    const float t = poly1 + 2.0f * poly * s;
    const float B = poly * q, B1 = q * t, B2 = q * (2.0f * s * t + poly2 + 2.0f * poly1 * s + 2.0f * poly * (u2 + w2));
    const float D = 1.0f + c.m * A - c.mm1 * B, D1 = c.m * A1 - c.mm1 * B1, D2 = c.m * A2 - c.mm1 * B2;
    const float C = __builtin_amdgcn_rcpf(D), Csq = C * C;
    const float C1 = -D1 * Csq, C2 = (2.0f * D1 * D1 * C - D2) * Csq;
    const float G = c.kd1 * I1 + c.kd2 * (C * I2);
    const float G1 = c.kd1 * I1a + c.kd2 * (C1 * I2 + C * I2a);
    const float G2 = c.kd1 * I1b + c.kd2 * (C2 * I2 + 2.0f * C1 * I2a + C * I2b);
    const float ce = c.ceta, rc = rho * ce;  // eta-derivatives -> rho-derivatives
    F2 a;
    a.v = rho * (F + rho * G);
    a.d1 = F + rc * F1 + rho * (2.0f * G + rc * G1);
    a.d2 = ce * (2.0f * F1 + rc * F2_) + 2.0f * G + rc * (4.0f * G1 + rc * G2);
    return a;
}

// Association term of a pure component in closed form (value, first and second density derivative).
//   a_assoc = rho q(S),  q = na (ln XA - XA/2 + 1/2) + nb (ln XB - XB/2 + 1/2),  S = rho Delta(eta) = rho da h(eta),
//   h = u + 1.5 eta u^2 + 0.5 eta^2 u^3,  u = 1/(1-eta)                                  (pcsaft_pure.py:163-176)
// XA = 1/(1 + nb S XB), XB = 1/(1 + na S XA) depend on rho through S only.  The association energy is stationary in the
// site fractions at the mass-action solution (Michelsen's Q function), so
//   dq/dS = -na nb XA XB,   d2q/dS2 = -na nb (XA' XB + XA XB'),
//   XA' = alpha (beta S XA - XB) / (1 - alpha beta S^2),  XB' = beta (alpha S XB - XA) / (1 - alpha beta S^2),
//   alpha = nb XA^2, beta = na XB^2,
// and a' = q + rho q_S S',  a'' = 2 q_S S' + rho (q_SS S'^2 + q_S S''),  S' = da (h + eta h'),  S'' = da ceta (2 h' + eta h''),
//   h' = 2.5 u^2 + 4 eta u^3 + 1.5 eta^2 u^4,   h'' = 9 u^3 + 15 eta u^4 + 6 eta^2 u^5.
// About a third of the generic value/d1/d2 arithmetic of the term (checked against it: tests/test_pure_gpu.py goldens and
// the 1e6-row parity runs).  XA, XB themselves from the cancellation-free closed forms of pure_model.hpp.
PCS_DEV F2 assoc_closed_f32(const PureCoefF& c, float rho) {
    const float eta = rho * c.ceta;
    const float u = __builtin_amdgcn_rcpf(1.0f - eta);
    const float u2 = u * u, eu = eta * u;
    const float h = u * (1.0f + eu * (1.5f + 0.5f * eu));
    const float h1 = u2 * (2.5f + eu * (4.0f + 1.5f * eu));
    const float h2 = u2 * u * (9.0f + eu * (15.0f + 6.0f * eu));
    const float S = rho * c.da * h;
    const float S1 = c.da * (h + eta * h1);
    const float S2 = c.da * c.ceta * (2.0f * h1 + eta * h2);
    const float sa = c.na * S, sb = c.nb * S;  // rho_a Delta, rho_b Delta
    const float t = sb - sa;
    const float aux = 1.0f - t;
    const float sq = __builtin_amdgcn_sqrtf(fmaf(aux, aux, 4.0f * sb));
    float xa, xb;
    if (t > 0.5f) {
        xa = 2.0f * __builtin_amdgcn_rcpf(sq + 1.0f + t);
        xb = (sq - 1.0f + t) * __builtin_amdgcn_rcpf(2.0f * sb);
    } else if (t < -0.5f) {
        xa = (sq - 1.0f - t) * __builtin_amdgcn_rcpf(2.0f * sa);
        xb = 2.0f * __builtin_amdgcn_rcpf(sq + 1.0f - t);
    } else {
        xa = 2.0f * __builtin_amdgcn_rcpf(sq + 1.0f + t);
        xb = 2.0f * __builtin_amdgcn_rcpf(sq + 1.0f - t);
    }
    const float q = c.na * (f_log(xa) - 0.5f * xa + 0.5f) + c.nb * (f_log(xb) - 0.5f * xb + 0.5f);
    const float nn = c.na * c.nb;
    const float q1 = -nn * xa * xb;
    const float al = c.nb * xa * xa, be = c.na * xb * xb;
    const float rden = __builtin_amdgcn_rcpf(1.0f - al * be * S * S);
    const float xa1 = al * (be * S * xa - xb) * rden, xb1 = be * (al * S * xb - xa) * rden;
    const float q2 = -nn * (xa1 * xb + xa * xb1);
    F2 r;
    r.v = rho * q;
    r.d1 = q + rho * q1 * S1;
    r.d2 = 2.0f * q1 * S1 + rho * (q2 * S1 * S1 + q1 * S2);
    return r;
}

// same model as pure_a() (pure_model.hpp), fp32
PCS_DEV EvalF pure_eval_f32(const PureCoefF& c, float rho) {
    F2 a = core_closed_f32(c, rho);
    if (c.polar || c.assoc) {
        F2 r = f2(rho, 1.0f, 0.0f);
        F2 eta = r * c.ceta;
        if (c.polar) {
            F2 rho2 = r * r;
            F2 J1 = hornerf<5>(c.j1, eta);
            F2 J2 = hornerf<4>(c.j2, eta);
            a = a + (rho2 * c.qm) * ((J1 * J1) * recipf(J1 - r * J2));
        }
        if (c.assoc) {
            a = a + assoc_closed_f32(c, rho);
        }
    }
    EvalF ec;
    ec.a = a.v;
    ec.p = rho - a.v + rho * a.d1;
    ec.dp = 1.0f + rho * a.d2;
    ec.mu = a.d1;
    return ec;
}

PCS_DEV bool finitef(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

// fp32 liquid root of p(rho) = p_spec from eta = 0.5.  Newton on (p - p_spec)(1-eta)^4 = 0 (same root):
// the hard-sphere pole makes p(rho) very steep on the dense side, the scaled function is close to
// linear -> 2-3 evaluations instead of 4-6 to a 10 % step.  Strongly attractive rows (large dipole /
// association at low T) have their liquid above eta = 0.5: they restart at eta = 0.58 on its dense side
// with plain Newton (monotone from there) and the tighter `tol_dense`.
// Wave-uniform loop; returns false when the lane must use the fp64 initialiser.
PCS_DEV bool liquid_root_f32(const PureCoefF& f, float p_spec, float tol, float tol_dense, int cap, float& rl,
                             int& n_eval) {
    bool ok = finitef(f.da) && finitef(f.kd2) && finitef(f.ceta) && f.ceta > 0.0f && finitef(p_spec);
    rl = 0.5f / f.ceta;
    bool done = !ok, dense = false;
    int first = 0;  // iteration at which the current start density is evaluated
    for (int it = 0; it < cap; it++) {
        if (!done) {
            EvalF e = pure_eval_f32(f, rl);
            n_eval++;
            float res = e.p - p_spec;
            if (it == first && it < PCS_F32_DENSE_LEVELS && finitef(e.p) && !(res > 0.0f)) {
                // still on the dilute side of the root: next start 0.08 further up (eta = 0.58, 0.66, 0.74)
                dense = true;
                first = it + 1;
                rl = (0.58f + 0.08f * (float)it) / f.ceta;
            } else if (!finitef(e.p) || !(e.dp > 0.0f) || (it == first && !(res > 0.0f))) {
                ok = false;
                done = true;
            } else {
                float den = dense ? e.dp : e.dp - 4.0f * res * f.ceta * __builtin_amdgcn_rcpf(1.0f - rl * f.ceta);
                float step = res * __builtin_amdgcn_rcpf(den);
                if (!(den > 0.0f)) step = 2.0f * rl;  // -> rn < 0 -> this lane takes the fp64 initialiser
                float rn = rl - step;
                if (!(rn > 0.0f)) { ok = false; done = true; }
                else { done = fabsf(step) <= (dense ? tol_dense : tol) * rl; rl = rn; }
            }
        }
        if (__ballot(!done) == 0ull) break;
    }
    return ok && done;
}

// fp32 pass, in two resumable parts so that a kernel can hand the few lanes that need more coupled iterations than their
// wave-mates to another wave (k_pure_vle<true>: block-level straggler exchange):
//   presolve_begin:   zero-pressure liquid root, liquid state at it, virial-corrected ideal-gas vapour estimate
//   presolve_coupled: coupled Newton towards the equal-area pressure from iteration s.it up to (excluding) it_end, or
//                     until every lane of the wave is done
// vle_presolve_f32 = begin + coupled(8).  s.ok: every step behaved (otherwise the lane uses the fp64 initialiser);
// s.done && s.ok: converged to the fp32 noise floor (typically 1e-6 relative).  Not converged within the cap is fine: the
// fp64 iteration continues from there.  l.dp / dpv: dp/drho of the two phases from the last evaluations (one small step
// before the returned densities): the second derivative the fp64 finish uses for its Newton steps.
struct PreState {
    float rl, rv;
    EvalF l;              // liquid state at rl (re-evaluated or carried by the Taylor expansion)
    float dpv;            // dp/drho of the vapour at its last evaluation
    float sl_prev, sv_prev;
    int it, n_liq, code;  // coupled iterations done; diagnostics
    bool ok, done;
};

PCS_DEV void presolve_begin(const PureCoefF& f, PreState& s) {
    s.n_liq = 0; s.it = 0; s.code = 0;
    s.dpv = 1.0f; s.sl_prev = 1.0f; s.sv_prev = 1.0f;
    // zero-pressure liquid, handed over to the coupled iteration at a loose step
    bool ok = liquid_root_f32(f, 0.0f, PCS_F32_LIQ_TOL, 1e-2f, 12, s.rl, s.n_liq);
    const float rl = s.rl;
    s.l = pure_eval_f32(f, rl);
    float rv = rl * f_exp(s.l.mu);
    {
        // second-virial correction of the ideal-gas estimate: ln rho + 2 B rho = ln rho_L + mu_L^res with
        // B = lim a/rho^2 from the coefficients (no model evaluation); three scalar Newton steps
        float B = (4.0f * f.m - 2.5f * f.mm1) * f.ceta + f.kd1 * f.ai[0] + f.kd2 * f.bi[0];
        if (f.polar) B += f.qm * f.j1[0];
        if (f.assoc) B -= f.na * f.nb * f.da;
        const float Lg = f_log(rv);
        float r = rv;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float den = fmaxf(1.0f + 2.0f * B * r, 0.3f);
            float fr = f_log(r) + 2.0f * B * r - Lg;
            r = r * fmaxf(1.0f - fr * __builtin_amdgcn_rcpf(den), 0.2f);
        }
        if (finitef(r) && r > 0.0f) rv = r;
    }
    s.code = !ok ? 1 : !finitef(rv) ? 6 : !(s.l.dp > 0.0f) ? 7 : !(rv < 0.5f * rl) ? 8 : !(rv > 1e-30f) ? 9 : 0;
    ok = ok && finitef(rv) && (s.l.dp > 0.0f) && (rv < 0.5f * rl) && (rv > 1e-30f);
    s.rv = rv;
    s.ok = ok;
    s.done = !ok;
}

PCS_DEV void presolve_coupled(const PureCoefF& f, PreState& s, int it_end) {
    float rl = s.rl, rv = s.rv;
    EvalF l = s.l;
    bool ok = s.ok, done = s.done;
    float dpv_last = s.dpv, dl_taken = 0.0f;
    float sl_prev = s.sl_prev, sv_prev = s.sv_prev;
    int n_cpl = s.it;
    // lanes of one wave may resume at different iteration numbers (straggler exchange): `it` below only bounds the loop,
    // the lane's own count n_cpl decides what the first-iteration rule of the stop criterion sees
    for (int it = 0; it < it_end; it++) {
        const bool act = !done && n_cpl < it_end;
        if (act) {
            EvalF v = pure_eval_f32(f, rv);
            dpv_last = v.dp;
            float iv = __builtin_amdgcn_rcpf(rv), il = __builtin_amdgcn_rcpf(rl);
            float ps = -(v.a * iv - l.a * il + f_log(rv * il)) * __builtin_amdgcn_rcpf(iv - il);
            float dl = -(l.p - ps) * __builtin_amdgcn_rcpf(l.dp);
            float dv = -(v.p - ps) * __builtin_amdgcn_rcpf(v.dp);
            float rln = rl + dl, rvn = rv + dv;
            // a large downward vapour step (poor first estimate at very low pressures) is taken in ln(rho) instead
            if (rvn < 0.3f * rv) rvn = rv * f_exp(dv * iv);
            if (!finitef(rln) || !finitef(rvn) || !(v.dp > 0.0f) || !(l.dp > 0.0f) || !(rvn > 1e-30f) || !(rvn < 0.6f * rln)) {
                s.code = (!finitef(rln) || !finitef(rvn)) ? 10 : !(v.dp > 0.0f) ? 11 : !(l.dp > 0.0f) ? 12 : !(rvn > 1e-30f) ? 13 : 14;
                ok = false;
                done = true;
            } else {
                // quadratic convergence: |next step| ~ C step^2 with C estimated from the last two steps
                float sl = fabsf(dl) * il, sv = fabsf(dv) * iv;
                float pl = sl * sl * fminf(sl * __builtin_amdgcn_rcpf(sl_prev * sl_prev), PCS_F32_PREDICT_CMAX);
                float pv = sv * sv * fminf(sv * __builtin_amdgcn_rcpf(sv_prev * sv_prev), PCS_F32_PREDICT_CMAX);
                done = ((sl <= 2e-6f) && (sv <= 3e-5f)) || (n_cpl > 0 && sl < 1e-2f && sv < 1e-2f && pl <= PCS_F32_PREDICT_TOL_L && pv <= PCS_F32_PREDICT_TOL_V);
                sl_prev = sl; sv_prev = sv;
                dl_taken = rln - rl;
                rl = rln;
                rv = rvn;
            }
            n_cpl++;
        }
        // the liquid state follows every taken step (also the last one of this call: a resumed lane starts from it)
        // the liquid barely moves after the first iteration: a lane whose liquid step was below 1e-3 carries its
        // liquid state to the new density by the Taylor expansion (a to 2nd, p to 1st order, dp kept) instead of a
        // re-evaluation; the error (~2.5 (dl/rho)^2 in the density) is below the fp32 noise the pass stops at.  The
        // choice is per lane (a row's result does not depend on its wave-mates); the evaluation is skipped when no
        // lane of the wave needs it.
        {
            const bool moved = act && !done;
            const bool reeval = moved && !(fabsf(dl_taken) <= PCS_F32_TAYLOR_MAX * rl);
            if (moved && !reeval) {
                const float a2 = (l.dp - 1.0f) * __builtin_amdgcn_rcpf(rl - dl_taken);  // a'' at the expansion point
                l.a = fmaf(dl_taken, fmaf(0.5f * a2, dl_taken, l.mu), l.a);
                l.mu = fmaf(a2, dl_taken, l.mu);
                l.p = fmaf(l.dp, dl_taken, l.p);
            }
            if (__ballot(reeval) != 0ull) {
                if (reeval) l = pure_eval_f32(f, rl);
            }
        }
        if (__ballot(!done && n_cpl < it_end) == 0ull) break;
    }
    s.rl = rl; s.rv = rv; s.l = l; s.dpv = dpv_last; s.sl_prev = sl_prev; s.sv_prev = sv_prev;
    s.it = n_cpl; s.ok = ok; s.done = done;
}

PCS_DEV bool vle_presolve_f32(const PureCoefF& f, double& rl_out, double& rv_out, float& dpl_out, float& dpv_out,
                              int* diag = nullptr) {
    PreState s;
    presolve_begin(f, s);
    presolve_coupled(f, s, 8);
    rl_out = (double)s.rl;
    rv_out = (double)s.rv;
    dpl_out = s.l.dp;
    dpv_out = s.dpv;
    if (diag) *diag = s.n_liq | (s.it << 8) | (s.code << 16);  // diagnostics builds only
    return s.ok;
}

}  // namespace pcs
