// Helpers shared by the gc-PC-SAFT translation units (gc_kernels.hip, gc_gradient.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abi_common.hpp"
#include "gc_model.hpp"

namespace {

using namespace pcs;
using namespace pcs_abi;

template <class P>
struct GcModelT {
    GcCoef<P> c;
    template <class R> PCS_DEV R a(const R& r0, const R& r1) const { return gc_a<P, R>(c, r0, r1); }
    template <class R, class Z> PCS_DEV R a_z(const R& r0, const R& r1, const Z& zeta3) const { return gc_a_z<P, R, Z>(c, r0, r1, zeta3); }
    PCS_DEV double packing(double x0, double x1) const { return x0 * re(c.zk[3][0]) + x1 * re(c.zk[3][1]); }
};

// stage the batch table (S*8 + 3*S*S doubles) into LDS
__device__ __forceinline__ GcTable stage_table(const double* __restrict__ table, int S, double* lds) {
    const int nd = gc_table_doubles(S);
    for (int k = threadIdx.x; k < nd; k += blockDim.x) lds[k] = table[k];
    __syncthreads();
    GcTable tb;
    tb.S = S;
    tb.seg = lds;
    tb.E1 = lds + S * 8;
    tb.E2 = tb.E1 + S * S;
    tb.K = tb.E2 + S * S;
    return tb;
}

// The 80 structure bytes of this lane's row, staged in LDS.  The model set-up (gc_mol) reads single bytes of the row inside its
// loops -- 32 for the molecule sums, up to 2 x 192 in the dispersion double sums, 48 for the bonds -- and as global byte loads
// each of them waits for its own round trip to the cache (one wave per SIMD: nothing hides it).  Lane stride 84 B = 21 dwords
// (odd: conflict-free for equal offsets); the area is lane-private, no barrier needed.  `area` = GC_ROW_LDS_DOUBLES doubles per
// thread of the workgroup.
constexpr int GC_ROW_LDS_STRIDE = 84;
constexpr int GC_ROW_LDS_DOUBLES = 11;
__device__ __forceinline__ const unsigned char* stage_row(const unsigned char* __restrict__ row, double* area) {
    unsigned int* dst = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(area) + threadIdx.x * GC_ROW_LDS_STRIDE);
    const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
    for (int k = 0; k < GC_ROW_BYTES / 16; k++) {
        const uint4 v = src[k];
        dst[4 * k] = v.x; dst[4 * k + 1] = v.y; dst[4 * k + 2] = v.z; dst[4 * k + 3] = v.w;
    }
    return reinterpret_cast<const unsigned char*>(dst);
}

// the one dual-number evaluation site of the gradient kernels, not inlined (register pressure, see mix_jacobian.hpp)
template <class G, class R>
__device__ __attribute__((noinline)) R gc_a_tangent(const GcCoef<G>& c, const R& r0, const R& r1) {
    return gc_a<G, R>(c, r0, r1);
}

inline size_t gc_lds_bytes(int S, int block, int per_thread_doubles) {
    return sizeof(double) * ((size_t)(S * 8 + 3 * S * S) + (size_t)per_thread_doubles * block);
}

inline int gc_check(int S, int64_t n) {
    if (int e = check_n(n)) return e;
    if (S < 1 || S > GC_MAXS) return fail_msg("gc: number of segment types must be in [1, 32]");
    return 0;
}

}  // namespace
