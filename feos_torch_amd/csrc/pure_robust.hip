// K1b of the pure-component VLE: the robust pass over the compacted retry list (rare rows:
// near-critical temperatures, strongly non-ideal vapour).  Own translation unit because it is
// compiled with strict IEEE comparison semantics (its bracketing / bisection loops rely on them;
// measured 2x slower under the relaxed flags of pure_kernels.hip), see feos_torch_amd/build.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abi_common.hpp"
#include "pure_solver.hpp"

using namespace pcs;

namespace {

constexpr int RETRY_GRID = 1024;  // 64-thread workgroups; the retry count is read on the device

// grid-stride over the list, count read on the device: no host synchronisation
__global__ __launch_bounds__(64) void k_pure_vle_robust(const double* __restrict__ params,
                                                        const double* __restrict__ temp,
                                                        double* __restrict__ p_sat, double* __restrict__ rho_eq,
                                                        double* __restrict__ rho_vl, uint8_t* __restrict__ status,
                                                        int32_t* __restrict__ iters,
                                                        const int32_t* __restrict__ retry, int64_t n) {
    const int count = (int)min((int64_t)max(retry[0], 0), n);  // bounded by n, as the entries below (see k_pure_vle_fallback)
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t entry = (uint32_t)retry[1 + k];
        if (entry & 0x80000000u) continue;  // solved by k_pure_vle_fallback (pure_kernels.hip)
        const int64_t i = (int64_t)entry;
        if (i >= n) continue;
        double par[8];
#pragma unroll
        for (int j = 0; j < 8; j++) par[j] = params[8 * i + j];
        const double T = temp[i];
        PureCoef<double> c;
        pure_coef<double>(c, par, T, false);
        VleResult r;
        int st = vle_robust(c, r, rho_eq ? 1e-8 : TOL_STEP);
        if (st == ST_OK) {
            if (p_sat) p_sat[i] = r.p_star * T * P_UNIT;
            if (rho_eq) rho_eq[i] = r.rho_l * (1.0 / RHO_UNIT);
            if (rho_vl) {
                rho_vl[2 * i] = r.rho_v;
                rho_vl[2 * i + 1] = r.rho_l;
            }
            if (iters) iters[i] = 1000 + r.iters;
            status[i] = 0;
        } else {
            if (p_sat) p_sat[i] = 0.0;
            if (rho_eq) rho_eq[i] = 0.0;
            if (rho_vl) {
                rho_vl[2 * i] = 0.0;
                rho_vl[2 * i + 1] = 0.0;
            }
            if (iters) iters[i] = -1;
            status[i] = 1;
        }
    }
}

}  // namespace

namespace pcs_abi {

int launch_pure_vle_retry(const double* params, const double* temp, double* p_sat, double* rho_eq, double* rho_vl,
                          uint8_t* status, int32_t* iters, const int32_t* retry, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_pure_vle_robust, dim3(RETRY_GRID), dim3(64), 0, s, params, temp, p_sat, rho_eq, rho_vl, status,
                       iters, retry, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("k_pure_vle_robust launch", e);
    return 0;
}

}  // namespace pcs_abi
