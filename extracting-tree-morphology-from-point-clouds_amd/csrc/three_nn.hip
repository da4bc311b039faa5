// three_nn + three_interpolate for gfx950 -- replaces Modules/PointNet2/blocks.py:194-204.
//
// three_nn: the reference builds the [B,N,S] distance matrix and fully sorts every row to keep 3 entries.
// Here one thread owns one dense point, the S sampled points (x, y, z, |p|^2) are staged once per workgroup in
// LDS and read as wave-uniform 16-byte broadcasts (conflict free), and a 3-entry insertion with strict '<'
// keeps the lower index ahead on equal distances (the reference's order among exact ties is whatever
// torch.sort(stable=False) does and is not specified; see tests/test_oracle_golden.py).  The inverse-distance
// weights are produced in the same kernel, in the reference's operation order.
//
// three_interpolate: out = (p[i0]*w0 + p[i1]*w1) + p[i2]*w2 with separate multiplies and adds, 16-byte
// vectorised over channels, written straight into the (optional) skip-connection concat buffer.
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 2048;  // sampled points staged per pass: 32 KiB of LDS

__global__ __launch_bounds__(kBlock) void three_nn_kernel(const float* __restrict__ xyz1, int64_t ab, int64_t an, int64_t ac,
                                                          const float* __restrict__ xyz2, int64_t bb, int64_t bn, int64_t bc,
                                                          int N, int S, int32_t* __restrict__ out_idx,
                                                          float* __restrict__ out_w, float* __restrict__ out_dist) {
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    const int n = blockIdx.x * kBlock + threadIdx.x;
    const bool ok = n < N;
    const float* p = xyz1 + (int64_t)b * ab + (int64_t)(ok ? n : 0) * an;
    const float px = p[0], py = p[ac], pz = p[2 * ac];
    const float pn = pn2::norm2(px, py, pz);
    float d0 = __builtin_inff(), d1 = d0, d2 = d0;
    int i0 = 0, i1 = 0, i2 = 0;

    for (int s0 = 0; s0 < S; s0 += kTile) {
        const int cnt = (S - s0) < kTile ? (S - s0) : kTile;
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += kBlock) {
            const float* q = xyz2 + (int64_t)b * bb + (int64_t)(s0 + t) * bn;
            const float x = q[0], y = q[bc], z = q[2 * bc];
            tile[t] = make_float4(x, y, z, pn2::norm2(x, y, z));
        }
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const float4 q = tile[t];
            const float d = pn2::sqdist(px, py, pz, pn, q.x, q.y, q.z, q.w);
            const bool c2 = d < d2;
            if (__ballot(c2)) {  // wave-uniform skip of the insertion once the top-3 has settled
                const bool c1 = d < d1, c0 = d < d0;
                const int s = s0 + t;
                d2 = c1 ? d1 : (c2 ? d : d2);
                i2 = c1 ? i1 : (c2 ? s : i2);
                d1 = c0 ? d0 : (c1 ? d : d1);
                i1 = c0 ? i0 : (c1 ? s : i1);
                d0 = c0 ? d : d0;
                i0 = c0 ? s : i0;
            }
        }
    }
    if (!ok) return;
    const size_t o = ((size_t)b * N + n) * 3;
    out_idx[o] = i0;
    out_idx[o + 1] = i1;
    out_idx[o + 2] = i2;
    if (out_dist) {
        out_dist[o] = d0;
        out_dist[o + 1] = d1;
        out_dist[o + 2] = d2;
    }
    // blocks.py:200-203: clamp(min=1e-6), reciprocal, (r0 + r1) + r2, divide
    const float r0 = __fdiv_rn(1.0f, d0 < 1e-6f ? 1e-6f : d0);
    const float r1 = __fdiv_rn(1.0f, d1 < 1e-6f ? 1e-6f : d1);
    const float r2 = __fdiv_rn(1.0f, d2 < 1e-6f ? 1e-6f : d2);
    const float sum = __fadd_rn(__fadd_rn(r0, r1), r2);
    out_w[o] = __fdiv_rn(r0, sum);
    out_w[o + 1] = __fdiv_rn(r1, sum);
    out_w[o + 2] = __fdiv_rn(r2, sum);
}

template <int V>
struct Vec;
template <>
struct Vec<1> {
    using T = float;
};
template <>
struct Vec<4> {
    using T = float4;
};

__device__ __forceinline__ float interp1(float a, float b, float c, float w0, float w1, float w2) {
    return __fadd_rn(__fadd_rn(__fmul_rn(a, w0), __fmul_rn(b, w1)), __fmul_rn(c, w2));
}

// V = 4 requires unit channel stride and 16-byte aligned rows on both sides
template <int V>
__global__ __launch_bounds__(kBlock) void three_interpolate_kernel(const float* __restrict__ points2, int64_t pb, int64_t pn,
                                                                   int64_t pc, const int32_t* __restrict__ idx,
                                                                   const float* __restrict__ w, int N, int D,
                                                                   float* __restrict__ out, int64_t out_stride,
                                                                   int64_t out_offset, long long total) {
    const int DV = D / V;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / DV;  // (b, n)
        const int c = (int)(e - r * DV) * V;
        const int b = (int)(r / N);
        const int j0 = idx[r * 3], j1 = idx[r * 3 + 1], j2 = idx[r * 3 + 2];
        const float w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
        const float* base = points2 + (int64_t)b * pb;
        float* o = out + r * out_stride + out_offset + c;
        if (V == 4) {
            const float4 a = *(const float4*)(base + (int64_t)j0 * pn + c);
            const float4 bq = *(const float4*)(base + (int64_t)j1 * pn + c);
            const float4 cq = *(const float4*)(base + (int64_t)j2 * pn + c);
            float4 v;
            v.x = interp1(a.x, bq.x, cq.x, w0, w1, w2);
            v.y = interp1(a.y, bq.y, cq.y, w0, w1, w2);
            v.z = interp1(a.z, bq.z, cq.z, w0, w1, w2);
            v.w = interp1(a.w, bq.w, cq.w, w0, w1, w2);
            *(float4*)o = v;
        } else {
            *o = interp1(base[(int64_t)j0 * pn + c * pc], base[(int64_t)j1 * pn + c * pc], base[(int64_t)j2 * pn + c * pc],
                         w0, w1, w2);
        }
    }
}

__global__ __launch_bounds__(kBlock) void three_interpolate_grad_kernel(const float* __restrict__ dout, int64_t out_stride,
                                                                        int64_t out_offset, const int32_t* __restrict__ idx,
                                                                        const float* __restrict__ w, int N, int S, int D,
                                                                        float* __restrict__ dpoints2, long long total) {
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / D;
        const int c = (int)(e - r * D);
        const int b = (int)(r / N);
        const float g = dout[r * out_stride + out_offset + c];
        float* base = dpoints2 + (int64_t)b * S * D + c;
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(base + (int64_t)idx[r * 3 + k] * D, __fmul_rn(g, w[r * 3 + k]));
    }
}

// square_distance as a materialised [B,N,M] matrix: API completeness only (pointnet2_utils.py:21-42); the fused
// kernels above never build it.
__global__ __launch_bounds__(kBlock) void square_distance_kernel(const float* __restrict__ src, int64_t ab, int64_t an,
                                                                 int64_t ac, const float* __restrict__ dst, int64_t bb,
                                                                 int64_t bn, int64_t bc, int N, int M,
                                                                 float* __restrict__ out, long long total) {
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / M;  // (b, n)
        const int m = (int)(e - r * M);
        const int b = (int)(r / N);
        const int n = (int)(r - (long long)b * N);
        const float* s = src + (int64_t)b * ab + (int64_t)n * an;
        const float* d = dst + (int64_t)b * bb + (int64_t)m * bn;
        const float sx = s[0], sy = s[ac], sz = s[2 * ac];
        const float dx = d[0], dy = d[bc], dz = d[2 * bc];
        out[e] = pn2::sqdist(sx, sy, sz, pn2::norm2(sx, sy, sz), dx, dy, dz, pn2::norm2(dx, dy, dz));
    }
}

inline unsigned grid_for(long long total) {
    long long g = (total + kBlock - 1) / kBlock;
    return (unsigned)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" int pn2_three_nn_f32(const float* xyz1, int64_t ab, int64_t an, int64_t ac, const float* xyz2, int64_t bb,
                                int64_t bn, int64_t bc, int B, int N, int S, int32_t* out_idx, float* out_w,
                                float* out_dist, void* stream) {
    if (!xyz1 || !xyz2 || !out_idx || !out_w || B <= 0 || N <= 0 || S < 3 || B > 65535) return PN2_E_BADARG;
    PN2_LAUNCH("three_nn", (double)B * (12.0 * N + 12.0 * S + 36.0 * N), 0, three_nn_kernel, dim3(pn2::ceil_div(N, kBlock), B),
               dim3(kBlock), (hipStream_t)stream, xyz1, ab, an, ac, xyz2, bb, bn, bc, N, S, out_idx, out_w, out_dist);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_square_distance_f32(const float* src, int64_t ab, int64_t an, int64_t ac, const float* dst, int64_t bb,
                                       int64_t bn, int64_t bc, int B, int N, int M, float* out, void* stream) {
    if (!src || !dst || !out || B <= 0 || N <= 0 || M <= 0) return PN2_E_BADARG;
    const long long total = (long long)B * N * M;
    PN2_LAUNCH("square_distance", 4.0 * total + 12.0 * B * (N + M), 0, square_distance_kernel, dim3(grid_for(total)),
               dim3(kBlock), (hipStream_t)stream, src, ab, an, ac, dst, bb, bn, bc, N, M, out, total);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_three_interpolate_f32(const float* points2, int64_t pb, int64_t pn, int64_t pc, const int32_t* idx,
                                         const float* w, int B, int N, int S, int D, float* out, int64_t out_stride,
                                         int64_t out_offset, void* stream) {
    if (!points2 || !idx || !w || !out || B <= 0 || N <= 0 || S <= 0 || D <= 0 || out_stride < out_offset + D)
        return PN2_E_BADARG;
    const bool vec = pc == 1 && D % 4 == 0 && pn % 4 == 0 && pb % 4 == 0 && out_stride % 4 == 0 && out_offset % 4 == 0 &&
                     ((uintptr_t)points2 % 16 == 0) && ((uintptr_t)out % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    const double ti_bytes = (double)B * N * (36.0 + 4.0 * D) + 4.0 * B * S * D;
    if (vec) {
        const long long total = (long long)B * N * (D / 4);
        PN2_LAUNCH("three_interpolate", ti_bytes, 0, (three_interpolate_kernel<4>), dim3(grid_for(total)), dim3(kBlock), s,
                   points2, pb, pn, pc, idx, w, N, D, out, out_stride, out_offset, total);
    } else {
        const long long total = (long long)B * N * D;
        PN2_LAUNCH("three_interpolate", ti_bytes, 0, (three_interpolate_kernel<1>), dim3(grid_for(total)), dim3(kBlock), s,
                   points2, pb, pn, pc, idx, w, N, D, out, out_stride, out_offset, total);
    }
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_three_interpolate_grad_f32(const float* dout, int64_t out_stride, int64_t out_offset,
                                              const int32_t* idx, const float* w, int B, int N, int S, int D,
                                              float* dpoints2, void* stream) {
    if (!dout || !idx || !w || !dpoints2 || B <= 0 || N <= 0 || S <= 0 || D <= 0 || out_stride < out_offset + D)
        return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    PN2_HIP_CHECK(hipMemsetAsync(dpoints2, 0, (size_t)B * S * D * sizeof(float), s));
    const long long total = (long long)B * N * D;
    PN2_LAUNCH("three_interpolate_grad", (double)B * N * (36.0 + 4.0 * D) + 4.0 * B * S * D, 0, three_interpolate_grad_kernel,
               dim3(grid_for(total)), dim3(kBlock), s, dout, out_stride, out_offset, idx, w, N, S, D, dpoints2, total);
    PN2_LAUNCH_CHECK();
    return 0;
}
