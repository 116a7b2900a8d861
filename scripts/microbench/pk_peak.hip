// packed fp32 FMA issue rate vs scalar fp32 FMA on this GPU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk(v2f* out, int iters, float seed) {
    v2f a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = (v2f){seed + threadIdx.x + j, seed - j};
    v2f b = {seed * 0.999f, seed * 0.998f}, c = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = __builtin_elementwise_fma(a[j], b, c);
    }
    v2f s = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sc(float* out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = seed + threadIdx.x + j;
    float b = seed * 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = fmaf(a[j], b, c);
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    const int blocks = 256 * 8, iters = 4096;
    void* out; hipMalloc(&out, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        float ms = 0;
        for (int r = 0; r < 6; r++) {
            if (r == 1) hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_sc, dim3(blocks), dim3(256), 0, 0, (float*)out, iters, 1.0001f);
            else hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), 0, 0, (v2f*)out, iters, 1.0001f);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double wi = (double)blocks * 4 * iters * 8;
        printf("%s: %.3f ms  %.1f Gwave-instr/s  (%.1f TFLOP/s)\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", ms, wi / ms / 1e6, wi * 64 * 2 * (which ? 2 : 1) / ms / 1e9);
    }
    return 0;
}
