// Shared device helpers for libpn2hip (gfx950 only).
//
// fp32 operation order is part of the contract (oracle/pn2_oracle.c): the translation unit is
// compiled with -ffp-contract=off and every rounding step below is spelled out, the only fused multiply-adds
// being the two explicit fmaf() of the K=3 dot product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pn2_hip.h"

#define PN2_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

#define PN2_HIP_CHECK(expr)                     \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

namespace pn2 {

constexpr int kWave = 64;

__device__ __forceinline__ float norm2(float x, float y, float z) {
    float a = __fmul_rn(x, x);
    float b = __fmul_rn(y, y);
    float c = __fmul_rn(z, z);
    return __fadd_rn(__fadd_rn(a, b), c);
}

// dot product in the order MKL's sgemm uses for K = 3: fma(z,z', fma(y,y', x*x'))
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, __fmul_rn(ax, bx)));
}

// square_distance(src, dst)[n][m] = ((-2*dot) + |src_n|^2) + |dst_m|^2   (pointnet2_utils.py:39-41)
__device__ __forceinline__ float sqdist(float sx, float sy, float sz, float sn2, float dx, float dy, float dz,
                                        float dn2) {
    float d = __fmul_rn(-2.0f, dot3(sx, sy, sz, dx, dy, dz));
    d = __fadd_rn(d, sn2);
    return __fadd_rn(d, dn2);
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace pn2
