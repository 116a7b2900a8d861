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
// hipMemsetAsync: every operation of a call is then a kernel node, so a hipGraph capture of the call replays them in
// the captured order (a memset node is executed by the runtime's blit path, which round 1 saw reordered against the
// kernels of a replay issued behind pending work).
template <int DUMMY = 0>
__global__ void k_zero_ints(int32_t* __restrict__ p, int count) {
    for (int k = threadIdx.x; k < count; k += blockDim.x) p[k] = 0;
}
inline int zero_ints(int32_t* p, int count, hipStream_t s) {
#ifdef PCS_ZERO_WITH_MEMSET  // A/B builds only (scripts/dev/capture_ab.py): round 1's hipMemsetAsync reset
    hipError_t em = hipMemsetAsync(p, 0, sizeof(int32_t) * count, s);
    return em == hipSuccess ? 0 : fail("hipMemsetAsync", em);
#endif
    hipLaunchKernelGGL(k_zero_ints<0>, dim3(1), dim3(64), 0, s, p, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_zero_ints launch", e);
    return 0;
}

}  // namespace pcs_abi
