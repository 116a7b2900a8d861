// Issue cost of the VALU instruction classes the pure-VLE kernel is made of, in shader cycles per wave-instruction per SIMD,
// measured with s_memtime inside the kernel at WAVES waves per SIMD (the headline kernel runs 3): every wave times a loop
// of independent instructions of one class; cost per SIMD = elapsed cycles / (instructions per wave x waves per SIMD).
// Feeds the VALU roofline of scripts/summarise_profile.py (valu.frac = sum_class N_class x cost_class / SIMD-cycles).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_cost scripts/microbench/issue_cost.hip && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
template <int CLS>
__global__ __launch_bounds__(256) void k(long long* cyc, double* sink, int iters, double seed) {
    double d0 = seed + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    float f0 = (float)d0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double db = 0.9999, dc = 1e-3;
    const float fb = 0.9999f, fc = 1e-3f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define D8(op) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc))
#define D8b(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db))
#define D8u(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define F8(op) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" \
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb), "v"(fc))
#define F8b(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb))
#define F8u(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7))
#define I8b(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
        : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0 | 1))
        if (CLS == 0) { REP8(D8("v_fma_f64");) }
        if (CLS == 1) { REP8(D8b("v_mul_f64");) }
        if (CLS == 2) { REP8(D8b("v_add_f64");) }
        if (CLS == 3) { REP8(F8("v_fma_f32");) }
        if (CLS == 4) { REP8(F8b("v_mul_f32");) }
        if (CLS == 5) { REP8(F8b("v_add_f32");) }
        if (CLS == 6) { REP8(I8b("v_and_b32");) }
        if (CLS == 7) { REP8(F8u("v_mov_b32");) }
        if (CLS == 8) { REP8(F8u("v_rcp_f32");) }
        if (CLS == 9) { REP8(F8u("v_sqrt_f32");) }
        if (CLS == 10) { REP8(F8u("v_exp_f32");) }
        if (CLS == 11) { REP8(F8u("v_log_f32");) }
        if (CLS == 12) { REP8(D8u("v_rcp_f64");) }
        if (CLS == 13) { REP8(D8u("v_sqrt_f64");) }
        if (CLS == 14) { REP8(D8u("v_rsq_f64");) }
        if (CLS == 15) { REP8(F8b("v_cndmask_b32");) }
        if (CLS == 16) { REP8(D8u("v_fract_f64");) }
        if (CLS == 17) { REP8(D8b("v_max_f64");) }
        if (CLS == 21) { REP8(D8("v_pk_fma_f32");) }  // packed fp32: register pairs
        if (CLS == 22) { REP8(D8b("v_pk_mul_f32");) }
        if (CLS == 18) {  // v_cvt_f32_f64: double sources, float destinations
            REP8(asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                              : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));)
        }
        if (CLS == 19) {  // v_cvt_f64_f32
            REP8(asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\nv_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                              : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));)
        }
        if (CLS == 20) {  // v_ldexp_f64: the exponent operand is a 32-bit integer
            REP8(asm volatile("v_ldexp_f64 %0, %0, %8\nv_ldexp_f64 %1, %1, %8\nv_ldexp_f64 %2, %2, %8\nv_ldexp_f64 %3, %3, %8\nv_ldexp_f64 %4, %4, %8\nv_ldexp_f64 %5, %5, %8\nv_ldexp_f64 %6, %6, %8\nv_ldexp_f64 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(i0 & 1));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int CLS>
void run(const char* name, int waves_per_simd) {
    // 256-thread blocks = 4 waves = 1 per SIMD; `waves_per_simd` blocks per CU, 256 CUs
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * waves_per_simd, iters = 4096;
    long long* cyc; double* sink;
    hipMalloc(&cyc, blocks * 4 * sizeof(long long));
    hipMalloc(&sink, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CLS>), dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<CLS>), dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3.f;
    // wall-clock view: wave-instructions of the whole chip per second, and the time one SIMD spends per wave-instruction
    const double wave_instr = (double)blocks * 4.0 * iters * 64.0;
    const double ns_per_simd = ms * 1e6 * (cus * 4.0) / wave_instr;
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double per_wave = med / (iters * 64.0);  // cycles per instruction as seen by one wave
    printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_instr_per_wave\": %.3f, \"ticks_per_wave_instr_per_simd\": %.3f, "
           "\"gwave_instr_per_s\": %.2f, \"ns_per_wave_instr_per_simd\": %.4f}\n", name, waves_per_simd, per_wave, per_wave / waves_per_simd,
           wave_instr / ms / 1e6, ns_per_simd);
    hipFree(cyc); hipFree(sink);
}

int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 3;
    run<0>("v_fma_f64", w);  run<1>("v_mul_f64", w);  run<2>("v_add_f64", w);
    run<3>("v_fma_f32", w);  run<4>("v_mul_f32", w);  run<5>("v_add_f32", w);
    run<6>("v_and_b32", w);  run<7>("v_mov_b32", w);  run<15>("v_cndmask_b32", w);
    run<8>("v_rcp_f32", w);  run<9>("v_sqrt_f32", w); run<10>("v_exp_f32", w); run<11>("v_log_f32", w);
    run<12>("v_rcp_f64", w); run<13>("v_sqrt_f64", w); run<14>("v_rsq_f64", w);
    run<16>("v_fract_f64", w); run<17>("v_max_f64", w);
    run<18>("v_cvt_f32_f64", w); run<19>("v_cvt_f64_f32", w); run<20>("v_ldexp_f64", w);
    run<21>("v_pk_fma_f32", w); run<22>("v_pk_mul_f32", w);
    return 0;
}
