// Shared helpers of the C-ABI translation units (error string, argument checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace pcs_abi {

extern thread_local char g_err[256];  // defined in pure_kernels.hip; read through pcs_last_error()

inline int fail(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return 1;
}
inline int fail_msg(const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return 2;
}
inline int check_n(int64_t n) {
    if (n < 0) return fail_msg("n must be >= 0");
    if (n >= (int64_t)1 << 31) return fail_msg("n must be < 2^31 rows per call (shard larger batches)");
    return 0;
}
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Device-side counters (retry-list count, work-queue control block) are zeroed by a KERNEL on the call's stream, not by
// hipMemsetAsync: every operation of a call is then a kernel node of a hipGraph capture.  Established in round 2 with one
// A/B run (profiles/r02_capture_ab.log): a capture of pcs_pure_vle replayed behind pending launches is bit-identical to the
// eager call 3/3 times with this reset, while the same library with the hipMemsetAsync reset of round 1 ended its second
// replay in a GPU memory access fault -- although every list consumer bounds count and entries by n, i.e. the fault is in
// the replayed memset node itself, not in a kernel reading a stale list.
template <int DUMMY = 0>
__global__ void k_zero_ints(int32_t* __restrict__ p, int count) {
    for (int k = threadIdx.x; k < count; k += blockDim.x) p[k] = 0;
}
inline int zero_ints(int32_t* p, int count, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_ints<0>, dim3(1), dim3(64), 0, s, p, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_zero_ints launch", e);
    return 0;
}

}  // namespace pcs_abi
