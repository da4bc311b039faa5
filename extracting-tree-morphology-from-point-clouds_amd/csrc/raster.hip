// Raster data path for gfx950 -- replaces the host loops of Modules/Pipeline/ModelPredicting.py:98-163 (rasterize_clouds),
// Modules/DataLoading/RasterizedTreeSet.py:201-268 (__getitem__: one boolean box mask over the whole cloud per raster)
// and :390-459 (collate_fn_streaming: per-mini-batch zero padding on the host).
//
// A raster is the set of points inside a half-open axis-aligned box [x_k, x_k + size) x [y_l, ...) x [z_m, ...) of a grid
// with origin at the cloud's minimum corner and step `stride` (np.arange(min, max, stride)); the dataset tests membership
// on the float32 coordinates against the float32-rounded bounds.  Per axis the boxes' lower and upper bounds are
// ascending, so the boxes that contain a coordinate are an index RANGE found by two binary searches with exactly that
// predicate -- lo[k] <= p < hi[k] -- whatever the rounding of the bounds did (a point on a seam may belong to two boxes
// or to none, like in the reference).  The reference spends O(#rasters x N) on masks per tree; here:
//   raster_ranges_kernel   per point: the 3 index ranges and the number of boxes it falls into (1 when size == stride);
//   (exclusive scan of the counts: torch.cumsum)
//   raster_keys_kernel     per point: one 64-bit key (box id * N + point id) per membership;
//   (one radix sort of the keys: torch.sort -> rasters in the reference's x-major order, ascending point ids inside)
//   raster_pack_kernel     straight from the sorted ids to the network's level-0 input: the zero-padded channel-first
//                          buffers of ALL mini-batches of the tree, laid end to end (= ops.RaggedClouds; each mini-batch's
//                          [B_j, CH, N_j] tensor is a view of it), plus the padding mask.
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;

// number of entries <= v in the ascending array a[0..n)
__device__ __forceinline__ int upper_bound_f32(const float* __restrict__ a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// bounds: [lo_x (nx) | hi_x (nx) | lo_y (ny) | hi_y (ny) | lo_z (nz) | hi_z (nz)]
__global__ __launch_bounds__(kBlock) void raster_ranges_kernel(const float* __restrict__ points, int64_t ps, int N,
                                                               const float* __restrict__ bounds, int nx, int ny, int nz,
                                                               int32_t* __restrict__ ranges, int32_t* __restrict__ count) {
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float* p = points + (int64_t)n * ps;
    const int dims[3] = {nx, ny, nz};
    const float* b = bounds;
    int total = 1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = p[a];
        const int first = upper_bound_f32(b + dims[a], dims[a], v);      // first box whose upper bound is above v
        const int last = upper_bound_f32(b, dims[a], v) - 1;             // last box whose lower bound is <= v
        ranges[6 * (size_t)n + 2 * a] = first;
        ranges[6 * (size_t)n + 2 * a + 1] = last;
        total *= last >= first ? last - first + 1 : 0;
        b += 2 * dims[a];
    }
    count[n] = total;    // NaN coordinates fall into no box
}

__global__ __launch_bounds__(kBlock) void raster_keys_kernel(const int32_t* __restrict__ ranges, const int64_t* __restrict__ offset,
                                                             int N, int ny, int nz, int64_t* __restrict__ keys) {
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int32_t* r = ranges + 6 * (size_t)n;
    int64_t o = offset[n];
    for (int kx = r[0]; kx <= r[1]; ++kx)
        for (int ky = r[2]; ky <= r[3]; ++ky)
            for (int kz = r[4]; kz <= r[5]; ++kz) keys[o++] = (((int64_t)kx * ny + ky) * nz + kz) * (int64_t)N + n;
}

// cloud table entry: {flat row offset (padded rows before this raster), padded length, first entry in the sorted id list,
// real length}
__global__ __launch_bounds__(kBlock) void raster_pack_kernel(const float* __restrict__ points, int64_t ps, const float* __restrict__ feats,
                                                             int64_t fs, int F, const int64_t* __restrict__ sorted_ids,
                                                             const int4* __restrict__ table, float* __restrict__ xyz_cf,
                                                             float* __restrict__ feats_cf, uint8_t* __restrict__ mask) {
    const int4 t = table[blockIdx.y];
    const int off = t.x, npad = t.y, first = t.z, nreal = t.w;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < npad; i += gridDim.x * kBlock) {
        const bool real = i < nreal;
        const int64_t id = real ? sorted_ids[first + i] : 0;
        const float* p = points + id * ps;
        float* x = xyz_cf + 3LL * off;
        x[i] = real ? p[0] : 0.0f;
        x[npad + i] = real ? p[1] : 0.0f;
        x[2LL * npad + i] = real ? p[2] : 0.0f;
        if (F > 0) {
            const float* f = feats + id * fs;
            float* y = feats_cf + (int64_t)F * off;
            for (int c = 0; c < F; ++c) y[(int64_t)c * npad + i] = real ? f[c] : 0.0f;
        }
        mask[off + i] = real ? 1 : 0;
    }
}

}  // namespace

extern "C" int pn2_raster_ranges_f32(const float* points, int64_t point_stride, int N, const float* bounds, int nx, int ny, int nz,
                                     int32_t* ranges, int32_t* count, void* stream) {
    if (!points || !bounds || !ranges || !count || N <= 0 || nx <= 0 || ny <= 0 || nz <= 0 || point_stride < 3) return PN2_E_BADARG;
    PN2_LAUNCH("raster_ranges", 40.0 * N, 0, raster_ranges_kernel, dim3(pn2::ceil_div(N, kBlock)), dim3(kBlock), (hipStream_t)stream,
               points, point_stride, N, bounds, nx, ny, nz, ranges, count);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_raster_keys(const int32_t* ranges, const int64_t* offset, int N, int ny, int nz, int64_t* keys, void* stream) {
    if (!ranges || !offset || !keys || N <= 0 || ny <= 0 || nz <= 0) return PN2_E_BADARG;
    PN2_LAUNCH("raster_keys", 40.0 * N, 0, raster_keys_kernel, dim3(pn2::ceil_div(N, kBlock)), dim3(kBlock), (hipStream_t)stream,
               ranges, offset, N, ny, nz, keys);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_raster_pack_f32(const float* points, int64_t point_stride, const float* feats, int64_t feat_stride, int F,
                                   const int64_t* sorted_ids, const int32_t* cloud_table, int C, int n_max, float* xyz_cf,
                                   float* feats_cf, uint8_t* masks_pad, void* stream) {
    if (!points || !sorted_ids || !cloud_table || !xyz_cf || !masks_pad || C <= 0 || C > 65535 || n_max <= 0 || F < 0 ||
        (F > 0 && (!feats || !feats_cf)) || point_stride < 3)
        return PN2_E_BADARG;
    int gx = pn2::ceil_div(n_max, kBlock);
    gx = gx > 64 ? 64 : gx;
    PN2_LAUNCH("raster_pack", 0, 0, raster_pack_kernel, dim3(gx, C), dim3(kBlock), (hipStream_t)stream, points, point_stride, feats,
               feat_stride, F, sorted_ids, (const int4*)cloud_table, xyz_cf, feats_cf, masks_pad);
    PN2_LAUNCH_CHECK();
    return 0;
}
