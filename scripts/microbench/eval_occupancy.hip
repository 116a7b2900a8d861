// How much would two waves per SIMD buy the mixture evaluation?  Persistent waves call the solvers' evaluation function
// (phase_eval: T2<double>, 256 VGPRs, not inlined) REPS times per lane on fixed rows, at 1 and at 2 waves per SIMD; prints the
// evaluation rate of the chip.  The kernel keeps almost nothing live across the call, so both occupancies fit.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DPCS_FAST_LOG=2 -DPCS_FAST_RCP=2 -Ifeos_torch_amd/csrc -o scratch/ab/eval_occ scripts/microbench/eval_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "mix_jacobian.hpp"
using namespace pcs;

template <int W>
__global__ __launch_bounds__(64, W) void k_eval(const double* __restrict__ par16, double T, double* __restrict__ out, int reps) {
    double par[16];
    for (int k = 0; k < 16; k++) par[k] = par16[16 * (threadIdx.x & 3) + k];  // four row classes per wave
    MixModelD m;
    mix_coef<double>(m.c, par, 0.01, 0.0, T);
    double r0 = 2.0e-3 + 1e-6 * threadIdx.x, r1 = 3.0e-3;
    double acc = 0.0;
    for (int k = 0; k < reps; k++) {
        PhaseEval e = phase_eval(m, r0, r1);
        acc += e.a + e.h01;
        r0 += 1e-9 * e.g0;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    // non-polar / polar / self-associating / cross-associating pairs
    const double rows[4][16] = {
        {1.5, 3.5, 250, 0, 0, 0, 0, 0, 2.5, 3.8, 280, 0, 0, 0, 0, 0},
        {1.5, 3.5, 250, 1.8, 0, 0, 0, 0, 2.5, 3.8, 280, 1.2, 0, 0, 0, 0},
        {1.5, 3.5, 250, 0, 0.03, 2500, 1, 1, 2.5, 3.8, 280, 0, 0, 0, 0, 0},
        {1.5, 3.5, 250, 0, 0.03, 2500, 1, 1, 2.5, 3.8, 280, 0, 0.02, 2000, 1, 1}};
    double* d_par; double* d_out;
    hipMalloc(&d_par, sizeof(rows)); hipMemcpy(d_par, rows, sizeof(rows), hipMemcpyHostToDevice);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int reps = 300;
    hipMalloc(&d_out, sizeof(double) * cus * 4 * 2 * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 1; w <= 2; w++) {
        const int grid = cus * 4 * w;
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (w == 1) hipLaunchKernelGGL(k_eval<1>, dim3(grid), dim3(64), 0, 0, d_par, 300.0, d_out, reps);
            else hipLaunchKernelGGL(k_eval<2>, dim3(grid), dim3(64), 0, 0, d_par, 300.0, d_out, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("{\"waves_per_simd\": %d, \"ms\": %.3f, \"lane_evals_per_s\": %.4e}\n", w, best, (double)grid * 64 * reps / (best * 1e-3));
    }
    return 0;
}
