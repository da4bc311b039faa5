// group_points / index_points for gfx950 -- replaces Modules/PointNet2/pointnet2_utils.py:45-63 and the
// gather + centring + concat of sample_and_group (lines 156-161; MSG variant blocks.py:141-146).
//
// Pure data movement: one thread per output element, consecutive threads write consecutive channels of one
// grouped row, so stores are fully coalesced and the gathered source row (channels-last) is read in
// contiguous runs.  The backward is a float atomic scatter-add (order differs from the reference's sequential
// index_put_, so gradients agree to rounding, not bitwise).
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;

// Indices handed to a gather are data, not trusted: one outside [0, N) is not followed (the element is taken from row 0 and the
// caller's status word gets PN2_STATUS_BAD_INDEX) instead of faulting the GPU -- e.g. the -1 rows a dead FPS launch
// leaves behind.  The reference asserts the range on the host (pointnet2_utils.py:54), which costs a sync per call.
__device__ __forceinline__ int checked(int j, int N, int32_t* status) {
    if ((unsigned)j < (unsigned)N) return j;
    if (status) atomicOr(status, PN2_STATUS_BAD_INDEX);
    return 0;
}

__global__ __launch_bounds__(kBlock) void group_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                       const float* __restrict__ new_xyz, const float* __restrict__ feats,
                                                       int64_t fb, int64_t fn, int64_t fc, const int32_t* __restrict__ idx,
                                                       int B, int N, int S, int K, int D, int xyz_last,
                                                       float* __restrict__ out, long long total, int32_t* status,
                                                       const int* __restrict__ coff) {
    const int C = 3 + D;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / C;  // (b, s, k)
        const int c = (int)(e - r * C);
        const long long bs = r / K;
        const int b = (int)(bs / S);
        const int cx = xyz_last ? c - D : c;  // coordinate channel if in [0,3)
        float v;
        if (coff) {  // ragged batch: both tensors channel-first per cloud (pn2_common.h: CloudView)
            const int o = coff[b], n = coff[b + 1] - o;
            const int j = checked(idx[r], n, status);
            if (cx >= 0 && cx < 3)
                v = __fsub_rn(xyz[3LL * o + (int64_t)cx * n + j], new_xyz[bs * 3 + cx]);
            else
                v = feats[(int64_t)D * o + (int64_t)(xyz_last ? c : c - 3) * n + j];
        } else {
            const int j = checked(idx[r], N, status);
            if (cx >= 0 && cx < 3) {
                v = __fsub_rn(xyz[(int64_t)b * sb + (int64_t)j * sn + cx * sc], new_xyz[bs * 3 + cx]);
            } else {
                const int cf = xyz_last ? c : c - 3;
                v = feats[(int64_t)b * fb + (int64_t)j * fn + cf * fc];
            }
        }
        out[e] = v;
    }
}

__global__ __launch_bounds__(kBlock) void group_grad_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                            int B, int N, int S, int K, int D, int xyz_last,
                                                            float* __restrict__ dfeats, long long total) {
    const int C = 3 + D;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / D;
        const int cf = (int)(e - r * D);
        const int b = (int)(r / ((long long)S * K));
        const int j = idx[r];
        if ((unsigned)j >= (unsigned)N) continue;   // flagged by the forward pass
        const float g = dout[r * C + (xyz_last ? cf : cf + 3)];
        atomicAdd(dfeats + ((int64_t)b * N + j) * D + cf, g);
    }
}

// One wavefront per GROUP (K <= 64 neighbours of one sampled point), lanes across the channels: a ball with fewer than
// nsample points is padded with its first hit (pointnet2_utils.py:126-130), so most of a sparse ball's K rows go to the same
// destination -- those are summed in registers and leave as ONE atomic per channel instead of serialising in the L2.
__global__ __launch_bounds__(kBlock) void group_grad_groups_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                                   long long groups, int SK_per_cloud, int N, int K, int D,
                                                                   int xyz_last, float* __restrict__ dfeats) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int C = 3 + D;
    const long long r0 = g * K;                              // first (b, s, k) row of the group
    const int b = (int)(r0 / SK_per_cloud);
    const int mine = lane < K ? idx[r0 + lane] : -1;
    const int j0 = __builtin_amdgcn_readfirstlane(mine);
    const unsigned long long same = __ballot(lane < K && mine == j0);          // rows that go where row 0 goes
    const unsigned long long other = __ballot(lane < K && mine != j0 && (unsigned)mine < (unsigned)N);
    const float* base = dout + r0 * C + (xyz_last ? 0 : 3);
    float* dst = dfeats + (long long)b * N * D;
    for (int c0 = 0; c0 < D; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < D;
        float acc = 0.f;
        unsigned long long m = same;
        while (m) {   // wave-uniform; four rows in flight
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = m ? (int)__builtin_ctzll(m) : -1;
                if (m) m &= m - 1ull;
                v[u] = (k >= 0 && live) ? base[(long long)k * C + c] : 0.f;
            }
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        if (live && (unsigned)j0 < (unsigned)N) atomicAdd(dst + (long long)j0 * D + c, acc);
        m = other;
        while (m) {   // four rows in flight here too
            float v[4];
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = m ? (int)__builtin_ctzll(m) : -1;
                if (m) m &= m - 1ull;
                j[u] = k >= 0 ? __builtin_amdgcn_readlane(mine, k) : -1;
                v[u] = (k >= 0 && live) ? base[(long long)k * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j[u] >= 0 && live) atomicAdd(dst + (long long)j[u] * D + c, v[u]);
        }
    }
}

__global__ __launch_bounds__(kBlock) void gather_kernel(const float* __restrict__ points, int64_t pb, int64_t pn, int64_t pc,
                                                        const int32_t* __restrict__ idx, int B, int N, int S, int C,
                                                        float* __restrict__ out, long long total, int32_t* status) {
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / C;
        const int c = (int)(e - r * C);
        const int b = (int)(r / S);
        out[e] = points[(int64_t)b * pb + (int64_t)checked(idx[r], N, status) * pn + c * pc];
    }
}

__global__ __launch_bounds__(kBlock) void gather_grad_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                             int B, int N, int S, int C, float* __restrict__ dpoints,
                                                             long long total) {
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
        const long long r = e / C;
        const int c = (int)(e - r * C);
        const int b = (int)(r / S);
        const int j = idx[r];
        if ((unsigned)j >= (unsigned)N) continue;   // flagged by the forward pass
        atomicAdd(dpoints + ((int64_t)b * N + j) * C + c, dout[e]);
    }
}

inline unsigned grid_for(long long total) {
    long long g = (total + kBlock - 1) / kBlock;
    return (unsigned)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" int pn2_group_f32(const float* xyz, int64_t sb, int64_t sn, int64_t sc, const float* new_xyz,
                             const float* feats, int64_t fb, int64_t fn, int64_t fc, const int32_t* idx, int B, int N,
                             int S, int K, int D, int xyz_last, float* out, int32_t* status, void* stream) {
    if (!xyz || !new_xyz || !idx || !out || B <= 0 || N <= 0 || S <= 0 || K <= 0 || D < 0 || (D > 0 && !feats))
        return PN2_E_BADARG;
    const long long total = (long long)B * S * K * (3 + D);
    PN2_LAUNCH("group", 8.0 * total + 4.0 * B * S * K, 0, group_kernel, dim3(grid_for(total)), dim3(kBlock), (hipStream_t)stream,
               xyz, sb, sn, sc, new_xyz, feats, fb, fn, fc, idx, B, N, S, K, D, xyz_last, out, total, status,
               (const int*)nullptr);
    PN2_LAUNCH_CHECK();
    return 0;
}

// Ragged batch: xyz_cf / feats_cf are the flat channel-first buffers (3 resp. D planes per cloud), idx is cloud-local.
extern "C" int pn2_group_ragged_f32(const float* xyz_cf, const float* feats_cf, int D, const int32_t* coff,
                                    const float* new_xyz, const int32_t* idx, int C, int S, int K, int xyz_last, float* out,
                                    int32_t* status, void* stream) {
    if (!xyz_cf || !coff || !new_xyz || !idx || !out || C <= 0 || S <= 0 || K <= 0 || D < 0 || (D > 0 && !feats_cf))
        return PN2_E_BADARG;
    const long long total = (long long)C * S * K * (3 + D);
    PN2_LAUNCH("group", 8.0 * total + 4.0 * C * S * K, 0, group_kernel, dim3(grid_for(total)), dim3(kBlock), (hipStream_t)stream,
               xyz_cf, 0, 1, 0, new_xyz, feats_cf, 0, 1, 0, idx, C, 0, S, K, D, xyz_last, out, total, status, (const int*)coff);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_group_grad_f32(const float* dout, const int32_t* idx, int B, int N, int S, int K, int D,
                                  int xyz_last, float* dfeats, void* stream) {
    if (!dout || !idx || !dfeats || B <= 0 || N <= 0 || S <= 0 || K <= 0 || D <= 0) return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    PN2_HIP_CHECK(hipMemsetAsync(dfeats, 0, (size_t)B * N * D * sizeof(float), s));
    const long long total = (long long)B * S * K * D;
    // padded balls pre-summed per group -- when there are enough groups to hide a wavefront's walk over its K rows (a
    // single 1024-point cloud with 256 groups: 13 us against 8 us row-wise; 398 raster clouds: 72 against 217 us)
    const bool by_group = getenv("PN2_GROUP_GRAD_GROUPS") != nullptr || ((long long)B * S >= 1024 && !getenv("PN2_GROUP_GRAD_ROWS"));
    if (K <= 64 && (long long)S * K <= 0x7FFFFFFF && by_group) {
        const long long groups = (long long)B * S;
        PN2_LAUNCH("group_grad", 8.0 * total + 4.0 * B * S * K + 4.0 * B * N * D, 0, group_grad_groups_kernel,
                   dim3((unsigned)pn2::ceil_div(groups, kBlock / 64)), dim3(kBlock), s, dout, idx, groups, S * K, N, K, D, xyz_last,
                   dfeats);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    PN2_LAUNCH("group_grad", 8.0 * total + 4.0 * B * S * K + 4.0 * B * N * D, 0, group_grad_kernel, dim3(grid_for(total)),
               dim3(kBlock), s, dout, idx, B, N, S, K, D, xyz_last, dfeats, total);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_gather_f32(const float* points, int64_t pb, int64_t pn, int64_t pc, const int32_t* idx, int B,
                              int N, int S, int C, float* out, int32_t* status, void* stream) {
    if (!points || !idx || !out || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PN2_E_BADARG;
    const long long total = (long long)B * S * C;
    PN2_LAUNCH("gather", 8.0 * total + 4.0 * B * S, 0, gather_kernel, dim3(grid_for(total)), dim3(kBlock), (hipStream_t)stream,
               points, pb, pn, pc, idx, B, N, S, C, out, total, status);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_gather_grad_f32(const float* dout, const int32_t* idx, int B, int N, int S, int C, float* dpoints,
                                   void* stream) {
    if (!dout || !idx || !dpoints || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    PN2_HIP_CHECK(hipMemsetAsync(dpoints, 0, (size_t)B * N * C * sizeof(float), s));
    const long long total = (long long)B * S * C;
    PN2_LAUNCH("gather_grad", 8.0 * total + 4.0 * B * S + 4.0 * B * N * C, 0, gather_grad_kernel, dim3(grid_for(total)),
               dim3(kBlock), s, dout, idx, B, N, S, C, dpoints, total);
    PN2_LAUNCH_CHECK();
    return 0;
}
