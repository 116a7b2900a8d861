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

}  // namespace pcs_abi
