// PointTransformerV3 conditional positional encoding for gfx950 (third stage): the submanifold 3 x 3 x 3 sparse
// convolution in front of every Block, Modules/PointTransformerV3/blocks.py:561-568 (spconv.SubMConv3d(channels, channels,
// kernel_size=3, bias=True) on the voxels of Point.sparsify, :153-190).  spconv is a CUDA-only dependency that is absent here;
// what is built is the operator's definition: the dense cross-correlation conv3d(padding = 1) of the voxel grid, evaluated at
// the ACTIVE voxels only, inactive voxels contributing zeros (tests/test_ptv3_cpe.py pins the restatement to
// torch.nn.functional.conv3d).
//
//   out[i] = bias + sum over the 27 offsets d = (dx, dy, dz) in {-1, 0, 1}^3 of  W[d] feat[j(i, d)],
//   j(i, d) = the active voxel of i's cloud at grid_coord[i] + d, if there is one.
//
// Two steps, two entry points:
//   pn2_ptv3_subm_neighbors_i32   j(i, d) for every voxel and offset: the voxels' (cloud, x, y, z) keys go into an open-addressing
//                                 hash table (one 64-bit compare-and-swap per voxel, linear probing at load <= 1/2), then every
//                                 (voxel, offset) pair is one lookup.  The table is per call; the neighbour table [N][27] is what
//                                 every Block of a stage shares (spconv's indice_key).
//   pn2_ptv3_subm_conv_f32        a GEMM whose A operand is GATHERED: rows = voxels (tiles of 128), contraction over (offset,
//                                 input channel) = 27 * C_in, columns = output channels.  Four wavefronts own 32 rows each and all
//                                 the tile's columns; A rows are staged K-major in LDS through the neighbour table (a missing
//                                 neighbour is a row of zeros), the offset's weight slab [C_in][C_out] beside it.  An offset none of
//                                 the tile's 128 voxels has a neighbour at is skipped as a whole (voxels of a surface have ~9 of
//                                 their 26 neighbours: most of the 27 slabs are skipped for most tiles when the rows are in
//                                 serialized order).  fp32: v_mfma_f32_32x32x2_f32.
#include <stdlib.h>

#include "pn2_common.h"

namespace {

using u64 = unsigned long long;
constexpr u64 kEmpty = ~0ull;
constexpr int kCoordBits = 16, kCoordMax = (1 << kCoordBits) - 2;   // grid coordinates 0 .. 65533 (one cell of margin each side)

__device__ __forceinline__ u64 voxel_key(long long b, int x, int y, int z) {
    return ((u64)b << 48) | ((u64)(unsigned)(x + 1) << 32) | ((u64)(unsigned)(y + 1) << 16) | (u64)(unsigned)(z + 1);
}
__device__ __forceinline__ unsigned hash64(u64 k) {   // murmur3 finalizer
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k;
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const long long* __restrict__ batch, const int* __restrict__ grid, int N,
                                                           u64* __restrict__ keys, int* __restrict__ vals, unsigned mask,
                                                           int32_t* status) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int x = grid[3 * i], y = grid[3 * i + 1], z = grid[3 * i + 2];
    const long long b = batch ? batch[i] : 0;
    if ((unsigned)x > (unsigned)kCoordMax || (unsigned)y > (unsigned)kCoordMax || (unsigned)z > (unsigned)kCoordMax || b < 0 || b > 65535) {
        if (status) atomicOr(status, PN2_STATUS_BAD_INDEX);   // outside the key's range: the voxel has no neighbours and is nobody's
        return;
    }
    const u64 key = voxel_key(b, x, y, z);
    unsigned h = hash64(key) & mask;
    for (;;) {   // terminates: the table has at least 2 N slots
        const u64 old = atomicCAS(keys + h, kEmpty, key);
        if (old == kEmpty) {
            vals[h] = i;
            return;
        }
        if (old == key) {   // a duplicate voxel: the lowest index represents the cell (the reference's voxels are unique)
            atomicMin(vals + h, i);
            return;
        }
        h = (h + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void voxel_neighbors_kernel(const long long* __restrict__ batch, const int* __restrict__ grid, int N,
                                                              const u64* __restrict__ keys, const int* __restrict__ vals,
                                                              unsigned mask, int32_t* __restrict__ nbr) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= 27ll * N) return;
    const int i = (int)(e / 27), d = (int)(e - 27ll * i);
    const int dx = d / 9 - 1, dy = (d / 3) % 3 - 1, dz = d % 3 - 1;
    const int x = grid[3 * i], y = grid[3 * i + 1], z = grid[3 * i + 2];
    const long long b = batch ? batch[i] : 0;
    int found = -1;
    if ((unsigned)x <= (unsigned)kCoordMax && (unsigned)y <= (unsigned)kCoordMax && (unsigned)z <= (unsigned)kCoordMax && b >= 0 &&
        b <= 65535) {
        const u64 key = voxel_key(b, x + dx, y + dy, z + dz);
        unsigned h = hash64(key) & mask;
        for (;;) {
            const u64 k = keys[h];
            if (k == key) {
                found = vals[h];
                break;
            }
            if (k == kEmpty) break;
            h = (h + 1) & mask;
        }
    }
    nbr[e] = found;
}

// ------------------------------------------------------------------------------------------------ the gathered GEMM
constexpr int CT_ROWS = 128, CBK = 16, CLD = CT_ROWS + 4;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// CT: columns per workgroup (32, 64 or 128).  grid: (row tiles, column tiles).
template <int CT>
__global__ __launch_bounds__(256) void subm_conv_kernel(const float* __restrict__ feat, long long ldf, const int32_t* __restrict__ nbr,
                                                        const float* __restrict__ weight, const float* __restrict__ bias, int N,
                                                        int Cin, int Cout, float* __restrict__ out, long long ldo) {
    constexpr int NJ = CT / 32, BLD = CT + 4;
    __shared__ __attribute__((aligned(16))) float As[2][CBK * CLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][CBK * BLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * CT_ROWS, n0 = blockIdx.y * CT;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    // staging roles: A -- two rows per thread (r_a + 64 p), four consecutive k; B -- CT * 16 / 256 floats per thread
    const int ra = tid >> 2, ka = 4 * (tid & 3);
    constexpr int BPT = CT * CBK / 256;              // floats per thread of the weight slab tile: 2, 4 or 8
    constexpr int BV = BPT >= 4 ? 4 : BPT;           // vector width of its loads
    constexpr int BN = BPT / BV;                     // loads per thread
    const int kb = (tid * BPT) / CT, nb = (tid * BPT) % CT;
    const int ktiles = Cin / CBK;
    float4 va[2];
    float vb[BPT];
    int buf = 0;
    bool first = true;
    for (int d = 0; d < 27; ++d) {
        // the two neighbours this thread gathers for offset d; does anybody in the tile have one?
        int j[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = m0 + ra + 64 * p;
            j[p] = row < N ? nbr[27ll * row + d] : -1;
        }
        if (!__syncthreads_or(j[0] >= 0 || j[1] >= 0)) continue;   // uniform: nobody in the tile needs this offset's slab
        const float* wd = weight + (long long)d * Cin * Cout;
        for (int kt = 0; kt < ktiles; ++kt) {
            const int k0 = kt * CBK;
            // fetch
#pragma unroll
            for (int p = 0; p < 2; ++p)
                va[p] = j[p] >= 0 ? *(const float4*)(feat + (long long)j[p] * ldf + k0 + ka) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < BN; ++q) {
                const float* src = wd + (long long)(k0 + kb) * Cout + n0 + nb + q * BV;
                if (BV == 4) {
                    const float4 t = *(const float4*)src;
                    vb[4 * q] = t.x, vb[4 * q + 1] = t.y, vb[4 * q + 2] = t.z, vb[4 * q + 3] = t.w;
                } else {
                    const float2 t = *(const float2*)src;
                    vb[2 * q] = t.x, vb[2 * q + 1] = t.y;
                }
            }
            // multiply the previous tile while these loads fly
            if (!first) {
                const float* a = As[buf ^ 1] + wave * 32 + l31;
                const float* b = Bs[buf ^ 1] + l31;
#pragma unroll
                for (int kk = 0; kk < CBK; kk += 2) {
                    const float av = a[(kk + half) * CLD];
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn)
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[(kk + half) * BLD + 32 * jn], acc[jn], 0, 0, 0);
                }
            }
            // commit into the other buffer
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float* dst = As[buf] + ka * CLD + ra + 64 * p;
                dst[0] = va[p].x, dst[CLD] = va[p].y, dst[2 * CLD] = va[p].z, dst[3 * CLD] = va[p].w;
            }
#pragma unroll
            for (int q = 0; q < BPT; ++q) Bs[buf][kb * BLD + nb + q] = vb[q];
            __syncthreads();
            buf ^= 1;
            first = false;
        }
    }
    if (!first) {   // the last staged tile
        const float* a = As[buf ^ 1] + wave * 32 + l31;
        const float* b = Bs[buf ^ 1] + l31;
#pragma unroll
        for (int kk = 0; kk < CBK; kk += 2) {
            const float av = a[(kk + half) * CLD];
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn)
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[(kk + half) * BLD + 32 * jn], acc[jn], 0, 0, 0);
        }
    }
    // epilogue.  C/D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
        const int col = n0 + 32 * jn + l31;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < N) out[(long long)row * ldo + col] = acc[jn][r] + bv;
        }
    }
}

inline size_t table_slots(int N) {
    size_t t = 1024;
    while (t < 2 * (size_t)N) t <<= 1;
    return t;
}

}  // namespace

extern "C" size_t pn2_ptv3_subm_workspace_bytes(int N) {
    if (N <= 0) return 0;
    return table_slots(N) * (sizeof(u64) + sizeof(int));
}

extern "C" int pn2_ptv3_subm_neighbors_i32(const int64_t* batch, const int32_t* grid_coord, int N, int32_t* nbr, void* workspace,
                                           size_t workspace_bytes, int32_t* status, void* stream) {
    if (!grid_coord || !nbr || N <= 0 || N > (1 << 30) / 27) return PN2_E_BADARG;
    if (!workspace || workspace_bytes < pn2_ptv3_subm_workspace_bytes(N)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t T = table_slots(N);
    u64* keys = (u64*)workspace;
    int* vals = (int*)(keys + T);
    PN2_HIP_CHECK(hipMemsetAsync(keys, 0xFF, T * sizeof(u64), s));
    PN2_HIP_CHECK(hipMemsetAsync(vals, 0x7F, T * sizeof(int), s));   // (atomicMin target of duplicate voxels)
    PN2_LAUNCH("ptv3_voxel_insert", 28.0 * N, 0, voxel_insert_kernel, dim3(pn2::ceil_div(N, 256)), dim3(256), s,
               (const long long*)batch, grid_coord, N, keys, vals, (unsigned)(T - 1), status);
    PN2_LAUNCH("ptv3_voxel_neighbors", 27.0 * 16.0 * N, 0, voxel_neighbors_kernel, dim3(pn2::ceil_div(27ll * N, 256)), dim3(256), s,
               (const long long*)batch, grid_coord, N, (const u64*)keys, (const int*)vals, (unsigned)(T - 1), nbr);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_ptv3_subm_conv_f32(const float* feat, int64_t ldf, const int32_t* nbr, const float* weight, const float* bias, int N,
                                      int Cin, int Cout, float* out, int64_t ldo, void* stream) {
    if (!feat || !nbr || !weight || !out || N <= 0 || Cin <= 0 || Cout <= 0 || Cin % CBK || Cout % 32 || ldf % 4 || ldf < Cin ||
        ldo < Cout || ((uintptr_t)feat & 15) || ((uintptr_t)weight & 15))
        return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int ct = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 32);
    const dim3 grid(pn2::ceil_div(N, CT_ROWS), Cout / ct);
    const double flops = 2.0 * 27.0 * N * (double)Cin * Cout;   // upper bound: the dense stencil (skipped slabs do no work)
    const double bytes = 4.0 * N * (27.0 + Cin + Cout) + 4.0 * 27.0 * Cin * Cout;
    if (ct == 128)
        PN2_LAUNCH("ptv3_subm_conv", bytes, flops, (subm_conv_kernel<128>), grid, dim3(256), s, feat, (long long)ldf, nbr, weight, bias, N,
                   Cin, Cout, out, (long long)ldo);
    else if (ct == 64)
        PN2_LAUNCH("ptv3_subm_conv", bytes, flops, (subm_conv_kernel<64>), grid, dim3(256), s, feat, (long long)ldf, nbr, weight, bias, N,
                   Cin, Cout, out, (long long)ldo);
    else
        PN2_LAUNCH("ptv3_subm_conv", bytes, flops, (subm_conv_kernel<32>), grid, dim3(256), s, feat, (long long)ldf, nbr, weight, bias, N,
                   Cin, Cout, out, (long long)ldo);
    PN2_LAUNCH_CHECK();
    return 0;
}
