// Would two rows per lane in packed fp32 pay?  The hard-sphere / chain / dispersion part of the fp32 evaluation of the pure
// VLE kernel (core_closed_f32, csrc/pure_f32.hpp: value, first and second density derivative in closed form) written once
// for a scalar float and once for a 2-vector (row A in .x, row B in .y; v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, the
// reciprocals and the logarithm per element), evaluated REPS times per lane by persistent waves.  Prints row-evaluations/s.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o scratch/ab/packed_eval scripts/microbench/packed_eval.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

struct Sc {  // scalar flavour
    typedef float T;
    static __device__ __forceinline__ T rcp(T x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ T lg(T x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }
    static __device__ __forceinline__ T bc(float x) { return x; }
};
struct Pk {  // two rows per lane
    typedef f2 T;
    static __device__ __forceinline__ T rcp(T x) { return T{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
    static __device__ __forceinline__ T lg(T x) { return T{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)} * 0.69314718f; }
    static __device__ __forceinline__ T bc(float x) { return T{x, x}; }
};

template <class V>
struct Coef { typename V::T m, mm1, ceta, ai[7], bi[7], kd1, kd2; };

template <class V>
__device__ __forceinline__ void core(const Coef<V>& c, typename V::T rho, typename V::T& a0, typename V::T& a1, typename V::T& a2) {
    typedef typename V::T T;
    const T one = V::bc(1.0f);
    const T eta = rho * c.ceta;
    const T u = V::rcp(one - eta), w = V::rcp(V::bc(2.0f) - eta);
    const T u2 = u * u, u3 = u2 * u, u4 = u2 * u2, w2 = w * w;
    const T HS = eta * (V::bc(4.0f) - 3.0f * eta) * u2, HS1 = (V::bc(4.0f) - 2.0f * eta) * u3, HS2 = (V::bc(10.0f) - 4.0f * eta) * u4;
    const T LG = V::lg((one - 0.5f * eta) * u3), LG1 = 3.0f * u - w, LG2 = 3.0f * u2 - w2;
    const T F = c.m * HS - c.mm1 * LG, F1 = c.m * HS1 - c.mm1 * LG1, F2 = c.m * HS2 - c.mm1 * LG2;
    T I1 = c.ai[6], I1a = V::bc(0.0f), I1h = V::bc(0.0f), I2 = c.bi[6], I2a = V::bc(0.0f), I2h = V::bc(0.0f);
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        I1h = I1h * eta + I1a; I1a = I1a * eta + I1; I1 = I1 * eta + c.ai[i];
        I2h = I2h * eta + I2a; I2a = I2a * eta + I2; I2 = I2 * eta + c.bi[i];
    }
    const T I1b = 2.0f * I1h, I2b = 2.0f * I2h;
    const T A = eta * (V::bc(8.0f) - 2.0f * eta) * u4, A1 = (V::bc(8.0f) + eta * (V::bc(20.0f) - 4.0f * eta)) * (u4 * u),
            A2 = (V::bc(60.0f) + eta * (V::bc(72.0f) - 12.0f * eta)) * (u4 * u2);
    const T poly = eta * (V::bc(20.0f) + eta * (V::bc(-27.0f) + eta * (V::bc(12.0f) - 2.0f * eta)));
    const T poly1 = V::bc(20.0f) + eta * (V::bc(-54.0f) + eta * (V::bc(36.0f) - 8.0f * eta)), poly2 = V::bc(-54.0f) + eta * (V::bc(72.0f) - 24.0f * eta);
    const T q = u2 * w2, s = u + w;
    const T t = poly1 + 2.0f * poly * s;
    const T B = poly * q, B1 = q * t, B2 = q * (2.0f * s * t + poly2 + 2.0f * poly1 * s + 2.0f * poly * (u2 + w2));
    const T D = one + c.m * A - c.mm1 * B, D1 = c.m * A1 - c.mm1 * B1, D2 = c.m * A2 - c.mm1 * B2;
    const T C = V::rcp(D), Csq = C * C;
    const T C1 = -D1 * Csq, C2 = (2.0f * D1 * D1 * C - D2) * Csq;
    const T G = c.kd1 * I1 + c.kd2 * (C * I2);
    const T G1 = c.kd1 * I1a + c.kd2 * (C1 * I2 + C * I2a);
    const T G2 = c.kd1 * I1b + c.kd2 * (C2 * I2 + 2.0f * C1 * I2a + C * I2b);
    const T ce = c.ceta, rc = rho * ce;
    a0 = rho * (F + rho * G);
    a1 = F + rc * F1 + rho * (2.0f * G + rc * G1);
    a2 = ce * (2.0f * F1 + rc * F2) + 2.0f * G + rc * (4.0f * G1 + rc * G2);
}

template <class V, int ROWS>
__global__ __launch_bounds__(256) void k(float* out, int reps, float seed) {
    typedef typename V::T T;
    Coef<V> c;
    const float t = seed + 1e-3f * threadIdx.x;
    c.m = V::bc(1.5f + t); c.mm1 = c.m - V::bc(1.0f); c.ceta = V::bc(20.0f + t);
    for (int i = 0; i < 7; i++) { c.ai[i] = V::bc(0.9f - 0.1f * i + t); c.bi[i] = V::bc(0.7f - 0.05f * i + t); }
    c.kd1 = V::bc(-300.0f); c.kd2 = V::bc(-500.0f);
    T rho = V::bc(0.02f + 1e-5f * threadIdx.x);
    T acc = V::bc(0.0f);
    for (int r = 0; r < reps; r++) {
        T a0, a1, a2;
        core<V>(c, rho, a0, a1, a2);
        acc = acc + a0 + a1 * 1e-3f + a2 * 1e-6f;
        rho = rho + a1 * 1e-9f;  // dependent chain like a Newton iteration
    }
    float v;
    if constexpr (ROWS == 2) v = acc.x + acc.y; else v = acc;
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int reps = 2000;
    float* out; hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int blocks_per_cu = 2; blocks_per_cu <= 4; blocks_per_cu += 2) {  // 256-thread blocks: 2 -> 2 waves/SIMD, 4 -> 4 waves/SIMD
            const int grid = cus * blocks_per_cu;
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((k<Sc, 1>), dim3(grid), dim3(256), 0, 0, out, reps, 0.01f);
                else hipLaunchKernelGGL((k<Pk, 2>), dim3(grid), dim3(256), 0, 0, out, reps, 0.01f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double rows = (double)grid * 256 * (mode ? 2 : 1) * reps;
            printf("{\"flavour\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"row_evals_per_s\": %.4e}\n", mode ? "packed (2 rows/lane)" : "scalar", blocks_per_cu, best, rows / (best * 1e-3));
        }
    }
    return 0;
}
