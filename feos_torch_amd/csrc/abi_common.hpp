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
// hipMemsetAsync: every operation of a call is then a kernel node of a hipGraph capture.  What round 2's one A/B run shows
// (profiles/r02_capture_ab.log, scripts/dev/capture_ab.py plants a stale count of 0x7FFFFFF0 before every replay): with this
// reset a capture of pcs_pure_vle replayed behind pending launches is bit-identical to the eager call 3/3 times; with the
// hipMemsetAsync reset of round 1 the second replay ended in a GPU memory access fault.  The log does not say which access
// faulted.  The reading that fits the code: the replayed memset was not reliably ordered before / visible to the first
// kernel, and the list PRODUCERS of that build appended without a bound (retry[1 + atomicAdd(&retry[0], 1)] = i), i.e. ~8 GB
// past the workspace on the planted count.  Since round 3 every producer bounds its slot by n (pure_kernels.hip,
// mix_kernels.hip, gc_kernels.hip) like the consumers always did, so a stale or foreign counter can no longer send a store
// out of the list whatever resets it.
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
