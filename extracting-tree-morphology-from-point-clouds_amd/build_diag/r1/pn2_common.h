// Shared device helpers for libpn2hip (gfx950 only).
//
// fp32 operation order is part of the contract (oracle/pn2_oracle.c): the translation unit is
// compiled with -ffp-contract=off and every rounding step below is spelled out, the only fused multiply-adds
// being the two explicit fmaf() of the K=3 dot product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pn2_hip_r1.h"

#define PN2_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

// Launch KERNEL (parenthesise template-ids) under a profiling scope carrying the launch's algorithmic bytes/flops.
#define PN2_LAUNCH(NAME, BYTES, FLOPS, KERNEL, GRID, BLOCK, STREAM, ...)                 \
    do {                                                                                 \
        pn2::prof::Scope sc_(NAME, STREAM, (double)(BYTES), (double)(FLOPS));             \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, __VA_ARGS__);                  \
    } while (0)

#define PN2_HIP_CHECK(expr)                     \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

namespace pn2 {

namespace prof {
extern bool g_enabled;
// RAII bracket around one kernel launch (see prof.hip); a no-op unless profiling is enabled.
class Scope {
   public:
    Scope(const char* name, hipStream_t s, double bytes, double flops);
    ~Scope();

   private:
    bool active_;
    hipStream_t s_;
    size_t idx_ = 0;
};
}  // namespace prof

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

// Two square distances per instruction (v_pk_mul/fma/add_f32).  The dst side is passed pre-scaled by -2: scaling by
// a power of two commutes with every rounding of the fma chain, so fma(-2z,z', fma(-2y,y', (-2x)*x')) is bit for
// bit -2*dot3(...) (coordinates far from the subnormal range).
using f2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 sqdist2(f2 sx, f2 sy, f2 sz, f2 sn2, f2 dx_m2, f2 dy_m2, f2 dz_m2, f2 dn2) {
    const f2 m = __builtin_elementwise_fma(sz, dz_m2, __builtin_elementwise_fma(sy, dy_m2, sx * dx_m2));
    return (m + sn2) + dn2;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// Wavefront max of an unsigned 64-bit key, result in every lane.  DPP row shifts / row broadcasts (full-rate
// VALU, ~10 cycles a step) instead of __shfl_xor, which lowers to ds_bpermute through the LDS crossbar
// (~100+ cycles a step in a dependent chain -- it was 2/3 of an FPS step).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;   // lanes without a source read 0
    return o > v ? o : v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned identity, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_u32<0x111, 0xF>(0u, v));
    v = max(v, dpp_u32<0x112, 0xF>(0u, v));
    v = max(v, dpp_u32<0x114, 0xF>(0u, v));
    v = max(v, dpp_u32<0x118, 0xF>(0u, v));
    v = max(v, dpp_u32<0x142, 0xA>(0u, v));
    v = max(v, dpp_u32<0x143, 0xC>(0u, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, dpp_u32<0x111, 0xF>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<0x112, 0xF>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<0x114, 0xF>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<0x118, 0xF>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<0x142, 0xA>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<0x143, 0xC>(0xFFFFFFFFu, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Wavefront max of the 64-bit key (hi << 32 | lo) as two 32-bit DPP reductions (each step is one v_max/v_min with a
// DPP source): max of the high words, then max of the low words among the lanes that hold it.
__device__ __forceinline__ unsigned long long wave_max_key(unsigned hi, unsigned lo) {
    const unsigned mh = wave_max_u32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = dpp_max_step<0x111, 0xF>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xF>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xF>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of each row holds the row max
    v = dpp_max_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave max
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace pn2
