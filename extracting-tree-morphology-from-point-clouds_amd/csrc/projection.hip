// Closest-cylinder projection for gfx950 -- replaces Modules/Projection.py:19-114 (closest_cylinder_cuda_batch; the
// same code is duplicated in PreProcessing/LabelGenerationCuda.py:20-110 and called from the QSM fitting).
//
// The reference broadcasts a batch of 1024 points against all M cylinders of a QSM: ~25 temporaries of shape
// [1024, M, 3] per batch, an argmin over M, fancy indexing of the winner, and a host round trip per batch.  Here one
// thread owns one point; the cylinders stream through LDS (8 floats each: start, unit axis, length, radius) as
// wave-uniform broadcasts; the thread evaluates the reference's distance expression for every cylinder with a running
// arg-min (strict '<': the first minimum, like torch.argmin) and then re-evaluates the winner once more to produce its
// mantle projection.  Nothing of size N x M is ever written.  VALU-bound: ~90 flop per (point, cylinder) pair.
//
// fp32 operation order is spelled out (the TU is built with -ffp-contract=off): 3-term sums are (x + y) + z, norms are
// sqrt of that sum, divisions and square roots are correctly rounded (sqrtf, not __fsqrt_rn: HIP maps that one to the
// approximate native square root) -- the same sequence as oracle/pn2_oracle.c
// (pn2o_cylinder_project), which restates the reference's torch expressions line by line.
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;   // cylinders per LDS pass (2 x 16 KiB)

struct Cyl {
    float sx, sy, sz, ux, uy, uz, len, rad;
};

__device__ __forceinline__ float dot3p(float ax, float ay, float az, float bx, float by, float bz) {
    return __fadd_rn(__fadd_rn(__fmul_rn(ax, bx), __fmul_rn(ay, by)), __fmul_rn(az, bz));
}
__device__ __forceinline__ float norm3(float x, float y, float z) { return __builtin_sqrtf(dot3p(x, y, z, x, y, z)); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// The reference's per-pair geometry (Projection.py:33-84).  Returns the distance; when WANT_POINT also the final
// projection point (mantle variant :90-105 if `mantle`, else the distance's own projection point).
template <bool WANT_POINT>
__device__ __forceinline__ float pair_distance(float px, float py, float pz, const Cyl& c, bool mantle, float& fx, float& fy,
                                               float& fz) {
    // :33-41  projection of (p - start) on the axis, clamped to the segment
    const float vx = __fsub_rn(px, c.sx), vy = __fsub_rn(py, c.sy), vz = __fsub_rn(pz, c.sz);
    const float pl = clampf(dot3p(vx, vy, vz, c.ux, c.uy, c.uz), 0.0f, c.len);
    const float qx = __fadd_rn(c.sx, __fmul_rn(pl, c.ux)), qy = __fadd_rn(c.sy, __fmul_rn(pl, c.uy)),
                qz = __fadd_rn(c.sz, __fmul_rn(pl, c.uz));
    // :44-48  projection vector, its axial component; "perpendicular" = |dot| <= 1e-3 (torch.isclose with atol)
    const float wx = __fsub_rn(px, qx), wy = __fsub_rn(py, qy), wz = __fsub_rn(pz, qz);
    const float dp = dot3p(wx, wy, wz, c.ux, c.uy, c.uz);
    const bool perp = fabsf(dp) <= 1e-3f;
    // :51-61  rejection (radial) direction
    const float rx = __fsub_rn(wx, __fmul_rn(dp, c.ux)), ry = __fsub_rn(wy, __fmul_rn(dp, c.uy)),
                rz = __fsub_rn(wz, __fmul_rn(dp, c.uz));
    float nr = norm3(rx, ry, rz);
    nr = nr < 1e-8f ? 1e-8f : nr;
    const float ax = __fdiv_rn(rx, nr), ay = __fdiv_rn(ry, nr), az = __fdiv_rn(rz, nr);
    // :64-68  the diameter segment through the clamped axis point
    const float two_r = __fmul_rn(2.0f, c.rad);
    const float hx = __fmul_rn(0.5f, __fmul_rn(ax, two_r)), hy = __fmul_rn(0.5f, __fmul_rn(ay, two_r)),
                hz = __fmul_rn(0.5f, __fmul_rn(az, two_r));
    const float s0x = __fsub_rn(qx, hx), s0y = __fsub_rn(qy, hy), s0z = __fsub_rn(qz, hz);
    // :71-76  projection of p on that segment
    const float t = clampf(dot3p(__fsub_rn(px, s0x), __fsub_rn(py, s0y), __fsub_rn(pz, s0z), ax, ay, az), 0.0f, two_r);
    const float ox = __fadd_rn(s0x, __fmul_rn(t, ax)), oy = __fadd_rn(s0y, __fmul_rn(t, ay)), oz = __fadd_rn(s0z, __fmul_rn(t, az));
    // :79  surface point for the perpendicular case
    const float ux = __fadd_rn(qx, __fmul_rn(ax, c.rad)), uy = __fadd_rn(qy, __fmul_rn(ay, c.rad)),
                uz = __fadd_rn(qz, __fmul_rn(az, c.rad));
    // :82-85
    const float gx = perp ? ux : ox, gy = perp ? uy : oy, gz = perp ? uz : oz;
    const float dist = norm3(__fsub_rn(px, gx), __fsub_rn(py, gy), __fsub_rn(pz, gz));
    if (WANT_POINT) {
        if (mantle) {   // :90-105  non-perpendicular points go to the nearer end of the diameter segment
            const float s1x = __fadd_rn(qx, hx), s1y = __fadd_rn(qy, hy), s1z = __fadd_rn(qz, hz);
            const float d0 = norm3(__fsub_rn(ox, s0x), __fsub_rn(oy, s0y), __fsub_rn(oz, s0z));
            const float d1 = norm3(__fsub_rn(ox, s1x), __fsub_rn(oy, s1y), __fsub_rn(oz, s1z));
            const bool to_start = d0 < d1;
            fx = perp ? ux : (to_start ? s0x : s1x);
            fy = perp ? uy : (to_start ? s0y : s1y);
            fz = perp ? uz : (to_start ? s0z : s1z);
        } else {
            fx = gx, fy = gy, fz = gz;
        }
    }
    return dist;
}

__global__ __launch_bounds__(kBlock) void cylinder_project_kernel(const float* __restrict__ points, int64_t ps, int N,
                                                                  const float* __restrict__ start, const float* __restrict__ unit,
                                                                  const float* __restrict__ length, const float* __restrict__ radius,
                                                                  const int32_t* __restrict__ ids, int M, int mantle,
                                                                  int32_t* __restrict__ out_id, float* __restrict__ out_dist,
                                                                  float* __restrict__ out_off) {
    __shared__ float4 ta[kTile], tb[kTile];   // {sx, sy, sz, len}, {ux, uy, uz, rad}
    const int n = blockIdx.x * kBlock + threadIdx.x;
    const bool ok = n < N;
    const float* p = points + (int64_t)(ok ? n : 0) * ps;
    const float px = p[0], py = p[1], pz = p[2];
    float best = __builtin_inff();
    int bi = 0;
    for (int m0 = 0; m0 < M; m0 += kTile) {
        const int cnt = M - m0 < kTile ? M - m0 : kTile;
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += kBlock) {
            const int m = m0 + t;
            ta[t] = make_float4(start[3 * m], start[3 * m + 1], start[3 * m + 2], length[m]);
            tb[t] = make_float4(unit[3 * m], unit[3 * m + 1], unit[3 * m + 2], radius[m]);
        }
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const float4 a = ta[t], b = tb[t];
            const Cyl c{a.x, a.y, a.z, b.x, b.y, b.z, a.w, b.w};
            float fx, fy, fz;
            const float d = pair_distance<false>(px, py, pz, c, false, fx, fy, fz);
            if (d < best) {     // strict: the first minimum wins, like torch.argmin; NaN never wins
                best = d;
                bi = m0 + t;
            }
        }
    }
    if (!ok) return;
    const Cyl c{start[3 * bi], start[3 * bi + 1], start[3 * bi + 2], unit[3 * bi], unit[3 * bi + 1], unit[3 * bi + 2], length[bi],
                radius[bi]};
    float fx, fy, fz;
    const float d = pair_distance<true>(px, py, pz, c, mantle != 0, fx, fy, fz);
    out_id[n] = ids ? ids[bi] : bi;
    if (out_dist) out_dist[n] = d;
    out_off[3 * n] = __fsub_rn(fx, px);
    out_off[3 * n + 1] = __fsub_rn(fy, py);
    out_off[3 * n + 2] = __fsub_rn(fz, pz);
}

}  // namespace

extern "C" int pn2_cylinder_project_f32(const float* points, int64_t point_stride, int N, const float* start, const float* axis_unit,
                                        const float* axis_length, const float* radius, const int32_t* ids, int M,
                                        int move_points_to_mantle, int32_t* out_id, float* out_dist, float* out_offset,
                                        void* stream) {
    if (!points || !start || !axis_unit || !axis_length || !radius || !out_id || !out_offset || N <= 0 || M <= 0 || point_stride < 3)
        return PN2_E_BADARG;
    PN2_LAUNCH("cylinder_project", 12.0 * N + 32.0 * M + 20.0 * N, 90.0 * (double)N * M, cylinder_project_kernel,
               dim3(pn2::ceil_div(N, kBlock)), dim3(kBlock), (hipStream_t)stream, points, point_stride, N, start, axis_unit,
               axis_length, radius, ids, M, move_points_to_mantle, out_id, out_dist, out_offset);
    PN2_LAUNCH_CHECK();
    return 0;
}
