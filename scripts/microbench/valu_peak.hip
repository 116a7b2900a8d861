// Measured VALU issue rates on this GPU: independent FMA chains, 256-thread blocks, 8 blocks/CU resident.
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T, int MODE>
__global__ __launch_bounds__(256) void k(T* out, int iters, T seed) {
    T a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = seed + (T)(threadIdx.x + j);
    T b = seed * (T)0.999, c = (T)1e-3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) a[j] = a[j] * b + c;                               // fma
            if (MODE == 1) a[j] = (sizeof(T) == 4) ? (T)__builtin_amdgcn_rcpf((float)a[j]) + c : (T)__builtin_amdgcn_rcp((double)a[j]) + c;  // rcp + add
            if (MODE == 2) a[j] = (T)__builtin_amdgcn_logf((float)a[j]) * b + c;   // f32 log2 + fma
        }
    }
    T s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class T, int MODE>
double run(const char* name, int ops_per_iter_per_lane) {
    const int blocks = 256 * 8, iters = 4096;
    T* out; hipMalloc(&out, blocks * 256 * sizeof(T));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, (T)1.0001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, (T)1.0001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double wave_instr = (double)blocks * 4 * iters * 8 * ops_per_iter_per_lane;  // wave-level instructions of interest
    double per_simd_cyc = ms * 1e-3 * 2.4e9 / (wave_instr / 1024.0);
    printf("%-28s %.3f ms  %.2f Gwave-instr/s  ~%.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, wave_instr / ms / 1e6, per_simd_cyc);
    hipFree(out); return ms;
}
int main() {
    run<double, 0>("v_fma_f64", 1);
    run<float, 0>("v_fma_f32", 1);
    run<double, 1>("v_rcp_f64 + v_add_f64", 2);
    run<float, 1>("v_rcp_f32 + v_add_f32", 2);
    run<float, 2>("v_log_f32 + v_fma_f32", 2);
}
