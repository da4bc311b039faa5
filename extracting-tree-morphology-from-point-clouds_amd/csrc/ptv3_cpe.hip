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
        if (old == kEmpty || old == key) {   // the slot is this voxel's cell; duplicate voxels: the lowest index represents the cell
            atomicMin(vals + h, i);          // (the reference's voxels are unique; vals starts at 0x7F7F7F7F)
            return;
        }
        h = (h + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void voxel_neighbors_kernel(const long long* __restrict__ batch, const int* __restrict__ grid, int N,
                                                              const u64* __restrict__ keys, const int* __restrict__ vals,
                                                              unsigned mask, int32_t* __restrict__ nbr, int ks) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int noff = ks * ks * ks, rad = ks / 2;
    if (e >= (long long)noff * N) return;
    const int i = (int)(e / noff), d = (int)(e - (long long)noff * i);
    const int dx = d / (ks * ks) - rad, dy = (d / ks) % ks - rad, dz = d % ks - rad;
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
                                                        int Cin, int Cout, float* __restrict__ out, long long ldo, int noff) {
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
    for (int d = 0; d < noff; ++d) {
        // the two neighbours this thread gathers for offset d; does anybody in the tile have one?
        int j[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = m0 + ra + 64 * p;
            j[p] = row < N ? nbr[(long long)noff * row + d] : -1;
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

// ---- the same contraction with the rows COMPACTED per offset ------------------------------------------------------------------
// At the occupancy of real plots (2-13 of the 27 neighbours present) a 128-voxel tile keeps nearly every offset slab although only
// 7-47 % of its rows have that neighbour, and the kernel above multiplies the zero rows too.  Here a tile's voxels that HAVE a
// neighbour at offset d are compacted into blocks of 32 (ballots over the tile's neighbour table in LDS), and a block is one
// wavefront's item: it gathers the block's 32 neighbour rows through a private LDS buffer (K-major, double buffered, no
// workgroup barrier), takes the offset's weight fragments straight from the L2-resident slab, and adds its 32 x CT products to an
// LDS copy of the output tile with ds_add_f32 (an output row receives at most one product row per offset; items of different
// offsets may meet in a row, hence the atomic -- the order of those <= 27 additions is not fixed, so the result may differ in the
// last bits from run to run, as every float-atomic reduction in this library).  The four wavefronts take items round-robin.
// Executed matrix work: ~27-33 blocks per tile instead of 95-108 at 2-5 present neighbours (tools/bench_ptv3_cpe.py).
__device__ __forceinline__ int nth_set_bit(u64 m, int n) {   // position of the n-th (0-based) set bit; n < popcount(m)
    int pos = 0;
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        const int c = __popcll((m >> pos) & ((1ull << w) - 1ull));
        if (n >= c) {
            n -= c;
            pos += w;
        }
    }
    return pos;
}
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

constexpr int ALD = 36;   // row pitch (floats) of a wavefront's [k][32 rows] staging buffer

// BF16 (throughput mode, like the attention's): the gathered rows are rounded to bfloat16 on their way into the wavefront's LDS
// image ([32 rows][16 k], 48-byte pitch: the matrix-core operand is one 16-byte read), the weights come as a bfloat16 copy
// [27][C_out][C_in] (eight consecutive k of a column = one 16-byte load), v_mfma_f32_32x32x16_bf16, fp32 accumulation and output.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
template <int CT, bool BF16>   // columns per workgroup: 32 or 64
__global__ __launch_bounds__(256) void subm_conv_compact_kernel(const float* __restrict__ feat, long long ldf,
                                                                const int32_t* __restrict__ nbr, const float* __restrict__ weight,
                                                                const float* __restrict__ bias, int N, int Cin, int Cout,
                                                                float* __restrict__ out, long long ldo) {
    constexpr int NJ = CT / 32, OLD = CT + 4;
    __shared__ __attribute__((aligned(16))) float O[CT_ROWS * OLD];
    __shared__ int s_nbr[CT_ROWS * 27];
    __shared__ __attribute__((aligned(16))) float Aw[4][1][2 * CBK * ALD];   // per wavefront: fp32 [32 k][36] or bfloat16 [32 rows][144 B]
    __shared__ int s_rows[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * CT_ROWS, n0 = blockIdx.y * CT;
    for (int e = tid; e < CT_ROWS * 27; e += 256) {
        const int r = e / 27;
        s_nbr[e] = m0 + r < N ? nbr[27ll * m0 + e] : -1;
    }
    for (int e = tid; e < CT_ROWS * CT; e += 256) {
        const int r = e / CT, c = e - r * CT;
        O[r * OLD + c] = bias ? bias[n0 + c] : 0.0f;
    }
    __syncthreads();
    const int ktiles = Cin / CBK;
    int item = 0;   // running (offset, block) item number of the tile: wavefront w takes items w, w + 4, ...
    for (int d = 0; d < 27; ++d) {
        const u64 mlo = __ballot(s_nbr[lane * 27 + d] >= 0), mhi = __ballot(s_nbr[(lane + 64) * 27 + d] >= 0);
        const int clo = __popcll(mlo), count = clo + __popcll(mhi);
        const int nblk = (count + 31) >> 5;
        const float* wd = weight + (long long)d * Cin * Cout + n0;
        const __bf16* wd16 = (const __bf16*)weight + ((long long)d * Cout + n0) * Cin;   // BF16: [27][C_out][C_in]
        for (int blk = 0; blk < nblk; ++blk, ++item) {
            if ((item & 3) != wave) continue;   // wave-uniform
            // this lane's compacted row (both half-wavefronts hold the same 32 rows)
            const int q = 32 * blk + l31;
            const bool valid = q < count;
            int row = 0;
            if (valid) row = q < clo ? nth_set_bit(mlo, q) : 64 + nth_set_bit(mhi, q - clo);
            const int j = valid ? s_nbr[row * 27 + d] : -1;
            if (lane < 32) s_rows[wave][lane] = valid ? row : -1;
            // staging roles: chunk c = lane (rows 0..15) and lane + 64 (rows 16..31), four consecutive k each
            const int jA = __shfl(j, lane >> 2, 64), jB = __shfl(j, 16 + (lane >> 2), 64);
            const int ka = 4 * (lane & 3);
            f32x16 acc[NJ];
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jn][r] = 0.0f;
            // The block's rows come KC K-tiles at a time: all of a chunk's gathers are in flight together and a chunk ahead of their
            // use (one K-tile per round trip made the kernel latency-bound: bfloat16 operands bought only 1.37 x).  One LDS image
            // per wavefront: its reads are issued, in order, before the next chunk's writes.
            constexpr int KC = BF16 ? 4 : 2;                  // K-tiles per chunk (fp32: the image must stay 4.6 KB per wavefront)
            constexpr int P16 = 2 * CBK * KC + 16;            // bfloat16 image: row pitch in bytes (16 bytes of padding)
            const int nchunks = ktiles / KC;                  // (the launcher guarantees Cin % (16 KC) == 0)
            float4 va[KC], vb[KC];
            auto gather = [&](int c) {
#pragma unroll
                for (int t = 0; t < KC; ++t) {
                    const int k = (c * KC + t) * CBK + ka;
                    va[t] = jA >= 0 ? *(const float4*)(feat + (long long)jA * ldf + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                    vb[t] = jB >= 0 ? *(const float4*)(feat + (long long)jB * ldf + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            // ... and so do the chunk's weight fragments (L2-resident slab): bw / bw16 hold the CURRENT chunk's, loaded a chunk ago
            bf16x8 bw16[KC][NJ], bn16[KC][NJ];
            float bw[BF16 ? 1 : KC * CBK / 2][NJ], bn[BF16 ? 1 : KC * CBK / 2][NJ];
            auto weights = [&](int c, bf16x8 (&w16)[KC][NJ], float (&w32)[BF16 ? 1 : KC * CBK / 2][NJ]) {
                const int k0 = c * KC * CBK;
                if constexpr (BF16) {
#pragma unroll
                    for (int t = 0; t < KC; ++t)
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn)
                            w16[t][jn] = *(const bf16x8*)(wd16 + (long long)(32 * jn + l31) * Cin + k0 + t * CBK + 8 * half);
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < KC * CBK / 2; ++s2)
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn) w32[s2][jn] = wd[(long long)(k0 + 2 * s2 + half) * Cout + 32 * jn + l31];
                }
            };
            gather(0);
            weights(0, bw16, bw);
            float* A = Aw[wave][0];
            char* A16 = (char*)A;
            for (int c = 0; c < nchunks; ++c) {
                if constexpr (BF16) {
#pragma unroll
                    for (int t = 0; t < KC; ++t) {
                        *(bf16x4*)(A16 + (lane >> 2) * P16 + 2 * (t * CBK + ka)) =
                            bf16x4{(__bf16)va[t].x, (__bf16)va[t].y, (__bf16)va[t].z, (__bf16)va[t].w};
                        *(bf16x4*)(A16 + (16 + (lane >> 2)) * P16 + 2 * (t * CBK + ka)) =
                            bf16x4{(__bf16)vb[t].x, (__bf16)vb[t].y, (__bf16)vb[t].z, (__bf16)vb[t].w};
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < KC; ++t) {
                        float* dst = A + (t * CBK + ka) * ALD + (lane >> 2);
                        dst[0] = va[t].x, dst[ALD] = va[t].y, dst[2 * ALD] = va[t].z, dst[3 * ALD] = va[t].w;
                        dst[16] = vb[t].x, dst[ALD + 16] = vb[t].y, dst[2 * ALD + 16] = vb[t].z, dst[3 * ALD + 16] = vb[t].w;
                    }
                }
                if (c + 1 < nchunks) {
                    gather(c + 1);
                    weights(c + 1, bn16, bn);
                }
                wave_lds_sync();
                if constexpr (BF16) {
#pragma unroll
                    for (int t = 0; t < KC; ++t) {
                        const bf16x8 af = *(const bf16x8*)(A16 + l31 * P16 + 2 * t * CBK + 16 * half);
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw16[t][jn], acc[jn], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < KC * CBK / 2; ++s2) {
                        const float av = A[(2 * s2 + half) * ALD + l31];
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[s2][jn], acc[jn], 0, 0, 0);
                    }
                }
                if (c + 1 < nchunks) {
                    if constexpr (BF16) {
#pragma unroll
                        for (int t = 0; t < KC; ++t)
#pragma unroll
                            for (int jn = 0; jn < NJ; ++jn) bw16[t][jn] = bn16[t][jn];
                    } else {
#pragma unroll
                        for (int s2 = 0; s2 < KC * CBK / 2; ++s2)
#pragma unroll
                            for (int jn = 0; jn < NJ; ++jn) bw[s2][jn] = bn[s2][jn];
                    }
                }
                wave_lds_sync();   // the image is rewritten by the next chunk
            }
            wave_lds_sync();
            // add the block's products to the output tile: accumulator row p = (r & 3) + 8 (r >> 2) + 4 half -> voxel s_rows[p]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = s_rows[wave][(r & 3) + 8 * (r >> 2) + 4 * half];
                if (orow >= 0) {
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn) atomicAdd(&O[orow * OLD + 32 * jn + l31], acc[jn][r]);
                }
            }
            wave_lds_sync();   // s_rows is rewritten by this wavefront's next item
        }
    }
    __syncthreads();
    for (int e = tid; e < CT_ROWS * CT; e += 256) {
        const int r = e / CT, c = e - r * CT;
        if (m0 + r < N) out[(long long)(m0 + r) * ldo + n0 + c] = O[r * OLD + c];
    }
}

inline size_t table_slots(int N) {
    size_t t = 1024;
    while (t < 2 * (size_t)N) t <<= 1;
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// Training: the weight gradient  dW[o][ci][co] = sum_i X[nbr[i][o]][ci] * dY[i][co]  (offset-major like the forward's packed
// weights).  One workgroup per (offset, 64 x 64 or 128 x 128 tile of (ci, co), row range): 32 rows per step -- the gathered X rows (absent
// neighbour: zeros) and the dY rows go to LDS as [row][channel], four wavefronts own 32 x 32 each (v_mfma_f32_32x32x2_f32, the
// contraction runs over the rows); the rows of a range that have the neighbour are first compacted in row order (2048 rows at a
// time), so the steps multiply present pairs only.  Row ranges leave slabs that a
// second launch sums in fixed order (deterministic, like the chains' split-K weight gradients).
// The input gradient needs no kernel of its own: nbr[i][o] = j <=> nbr[j][noff - 1 - o] = i for distinct voxels, so
// dX = subm_conv(dY, nbr, W') with W'[o][co][ci] = W[noff - 1 - o][ci][co] (PointTransformerV3/cpe.py).
constexpr int WGK = 32;
constexpr int WG_SPLIT_ROWS = 4096, WG_MAX_SPLITS = 64;
inline int wgrad_splits(int N) {
    const int n = pn2::ceil_div(N, WG_SPLIT_ROWS);
    return n < 1 ? 1 : (n > WG_MAX_SPLITS ? WG_MAX_SPLITS : n);
}
// T = tile edge: 64 (four wavefronts own 32 x 32 each) for layers up to 64 wide, 128 (64 x 64 each: four accumulators) for the
// wide ones -- a (ci, co) tile re-reads the gathered rows once per tile of the OTHER dimension, so doubling the edge halves the
// bytes of a 256-wide layer (2 x 2 x 27 instead of 4 x 4 x 27 passes over the valid rows)
template <int T>
__global__ __launch_bounds__(256) void subm_wgrad_kernel(const float* __restrict__ feat, long long ldf, const int* __restrict__ nbr,
                                                         int noff, const float* __restrict__ dout, long long ldo, int N, int Cin,
                                                         int Cout, int rows_per_split, float* __restrict__ slab) {
    constexpr int LD = T + 4, NI = T / 64, TPR = T / 4, RPP = 256 / TPR, PASSES = WGK / RPP;
    __shared__ float sX[WGK * LD], sY[WGK * LD];
    // the rows of a super-step that HAVE the neighbour, compacted in row order: (row, neighbour) pairs
    constexpr int SUP = 2048, RPT = SUP / 256;
    __shared__ int sI[SUP], sJ[SUP];
    __shared__ int sWave[4];
    const int tco_n = (Cout + T - 1) / T;
    const int o = blockIdx.x % noff, tile = blockIdx.x / noff;
    const int ci0 = (tile / tco_n) * T, co0 = (tile % tco_n) * T;
    const int split = blockIdx.y;
    const int r_begin = split * rows_per_split, r_end = min(N, r_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    f32x16 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int s0 = r_begin; s0 < r_end; s0 += SUP) {
        // ---- ordered compaction: thread t looks at rows s0 + RPT t .. + RPT - 1 (the multiplication then runs over present pairs
        //      only: at 2-6 of 27 neighbours present the dense steps were 80-93 % zeros)
        int jj[RPT], cnt = 0;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int i = s0 + RPT * tid + u;
            jj[u] = i < r_end ? nbr[(long long)i * noff + o] : -1;
            cnt += jj[u] >= 0 ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();   // (also: the previous super-step's staging reads of sI / sJ are over)
        int pos = incl - cnt, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            if (w2 < wave) pos += sWave[w2];
            total += sWave[w2];
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u)
            if (jj[u] >= 0) {
                sI[pos] = s0 + RPT * tid + u;
                sJ[pos] = jj[u];
                ++pos;
            }
        __syncthreads();
        for (int p0 = 0; p0 < total; p0 += WGK) {
#pragma unroll
            for (int pass = 0; pass < PASSES; ++pass) {
                const int rr = tid / TPR + RPP * pass, c4 = (tid % TPR) * 4;
                float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f), y4 = x4;
                if (p0 + rr < total) {
                    const int i = sI[p0 + rr], j = sJ[p0 + rr];
                    if (ci0 + c4 < Cin) x4 = *(const float4*)(feat + (long long)j * ldf + ci0 + c4);
                    if (co0 + c4 < Cout) y4 = *(const float4*)(dout + (long long)i * ldo + co0 + c4);
                }
                *(float4*)(sX + rr * LD + c4) = x4;
                *(float4*)(sY + rr * LD + c4) = y4;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < WGK; k += 2) {
                float a[NI], b[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    a[i] = sX[(k + (lane >> 5)) * LD + (T / 2) * wm + 32 * i + (lane & 31)];
                    b[i] = sY[(k + (lane >> 5)) * LD + (T / 2) * wn + 32 * i + (lane & 31)];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();   // the tiles are rewritten by the next step
        }
    }
    float* dst = slab + ((long long)split * noff + o) * Cin * Cout;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (T / 2) * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int co = co0 + (T / 2) * wn + 32 * j + (lane & 31);
                if (ci < Cin && co < Cout) dst[(long long)ci * Cout + co] = acc[i][j][r];
            }
}
__global__ __launch_bounds__(256) void subm_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, long long mn,
                                                                float* __restrict__ dw) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < mn; e += (long long)gridDim.x * 256) {
        float sum = 0.0f;
        for (int sidx = 0; sidx < nsplit; ++sidx) sum += slab[(long long)sidx * mn + e];
        dw[e] = sum;
    }
}

}  // namespace

extern "C" size_t pn2_ptv3_subm_workspace_bytes(int N) {
    if (N <= 0) return 0;
    return table_slots(N) * (sizeof(u64) + sizeof(int));
}

extern "C" int pn2_ptv3_subm_neighbors_i32(const int64_t* batch, const int32_t* grid_coord, int N, int kernel_size, int32_t* nbr,
                                           void* workspace, size_t workspace_bytes, int32_t* status, void* stream) {
    if (!grid_coord || !nbr || N <= 0 || (kernel_size != 3 && kernel_size != 5)) return PN2_E_BADARG;
    const int noff = kernel_size * kernel_size * kernel_size;
    if (N > (1 << 30) / noff) return PN2_E_BADARG;
    if (!workspace || workspace_bytes < pn2_ptv3_subm_workspace_bytes(N)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t T = table_slots(N);
    u64* keys = (u64*)workspace;
    int* vals = (int*)(keys + T);
    PN2_HIP_CHECK(hipMemsetAsync(keys, 0xFF, T * sizeof(u64), s));
    PN2_HIP_CHECK(hipMemsetAsync(vals, 0x7F, T * sizeof(int), s));   // (atomicMin target of duplicate voxels)
    PN2_LAUNCH("ptv3_voxel_insert", 28.0 * N, 0, voxel_insert_kernel, dim3(pn2::ceil_div(N, 256)), dim3(256), s,
               (const long long*)batch, grid_coord, N, keys, vals, (unsigned)(T - 1), status);
    PN2_LAUNCH("ptv3_voxel_neighbors", 16.0 * noff * N, 0, voxel_neighbors_kernel, dim3(pn2::ceil_div((long long)noff * N, 256)),
               dim3(256), s, (const long long*)batch, grid_coord, N, (const u64*)keys, (const int*)vals, (unsigned)(T - 1), nbr,
               kernel_size);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_ptv3_subm_conv_f32(const float* feat, int64_t ldf, const int32_t* nbr, int kernel_size, const float* weight,
                                      const void* weight_bf16, const float* bias, int N, int Cin, int Cout, float* out,
                                      int64_t ldo, void* stream) {
    if (!feat || !nbr || !weight || !out || N <= 0 || Cin <= 0 || Cout <= 0 || Cin % CBK || Cout % 32 || ldf % 4 || ldf < Cin ||
        ldo < Cout || ((uintptr_t)feat & 15) || ((uintptr_t)weight & 15) || (kernel_size != 3 && kernel_size != 5))
        return PN2_E_BADARG;
    const int noff = kernel_size * kernel_size * kernel_size;
    hipStream_t s = (hipStream_t)stream;
    // rows compacted per offset from 128 input channels on (measured at 3.7 present neighbours, 1 M voxels: C = 128 7.0 -> 5.0 ms;
    // C = 64 2.06 -> 2.00; C = 32 0.68 -> 0.92: an item's per-K-tile round trips are not amortised by narrow rows);
    // PN2_CPE_DENSE_TILES=1 / PN2_CPE_COMPACT=1 force either kernel (A/B aid)
    if (weight_bf16 && noff == 27 && Cin >= 64 && Cin % 64 == 0) {   // bf16 mode: the compacted kernel on bfloat16 operands (narrower layers: fp32)
        if ((uintptr_t)weight_bf16 & 15) return PN2_E_BADARG;
        const int ctc = Cout % 64 == 0 ? 64 : 32;
        const dim3 gridc(pn2::ceil_div(N, CT_ROWS), Cout / ctc);
        const double fl = 2.0 * 27.0 * N * (double)Cin * Cout, by = 4.0 * N * (27.0 + Cin + Cout) + 2.0 * 27.0 * Cin * Cout;
        if (ctc == 64)
            PN2_LAUNCH("ptv3_subm_conv_bf16", by, fl, (subm_conv_compact_kernel<64, true>), gridc, dim3(256), s, feat, (long long)ldf, nbr,
                       (const float*)weight_bf16, bias, N, Cin, Cout, out, (long long)ldo);
        else
            PN2_LAUNCH("ptv3_subm_conv_bf16", by, fl, (subm_conv_compact_kernel<32, true>), gridc, dim3(256), s, feat, (long long)ldf, nbr,
                       (const float*)weight_bf16, bias, N, Cin, Cout, out, (long long)ldo);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    if (noff == 27 && Cin % 32 == 0 && (Cin >= 128 || getenv("PN2_CPE_COMPACT")) && !getenv("PN2_CPE_DENSE_TILES")) {
        const int ctc = Cout % 64 == 0 ? 64 : 32;
        const dim3 gridc(pn2::ceil_div(N, CT_ROWS), Cout / ctc);
        const double fl = 2.0 * 27.0 * N * (double)Cin * Cout, by = 4.0 * N * (27.0 + Cin + Cout) + 4.0 * 27.0 * Cin * Cout;
        if (ctc == 64)
            PN2_LAUNCH("ptv3_subm_conv", by, fl, (subm_conv_compact_kernel<64, false>), gridc, dim3(256), s, feat, (long long)ldf, nbr, weight, bias,
                       N, Cin, Cout, out, (long long)ldo);
        else
            PN2_LAUNCH("ptv3_subm_conv", by, fl, (subm_conv_compact_kernel<32, false>), gridc, dim3(256), s, feat, (long long)ldf, nbr, weight, bias,
                       N, Cin, Cout, out, (long long)ldo);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    const int ct = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 32);
    const dim3 grid(pn2::ceil_div(N, CT_ROWS), Cout / ct);
    const double flops = 2.0 * noff * N * (double)Cin * Cout;   // upper bound: the dense stencil (skipped slabs do no work)
    const double bytes = 4.0 * N * ((double)noff + Cin + Cout) + 4.0 * noff * Cin * Cout;
    if (ct == 128)
        PN2_LAUNCH("ptv3_subm_conv", bytes, flops, (subm_conv_kernel<128>), grid, dim3(256), s, feat, (long long)ldf, nbr, weight, bias, N,
                   Cin, Cout, out, (long long)ldo, noff);
    else if (ct == 64)
        PN2_LAUNCH("ptv3_subm_conv", bytes, flops, (subm_conv_kernel<64>), grid, dim3(256), s, feat, (long long)ldf, nbr, weight, bias, N,
                   Cin, Cout, out, (long long)ldo, noff);
    else
        PN2_LAUNCH("ptv3_subm_conv", bytes, flops, (subm_conv_kernel<32>), grid, dim3(256), s, feat, (long long)ldf, nbr, weight, bias, N,
                   Cin, Cout, out, (long long)ldo, noff);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pn2_ptv3_subm_wgrad_workspace_bytes(int N, int kernel_size, int Cin, int Cout) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || (kernel_size != 3 && kernel_size != 5)) return 0;
    return (size_t)wgrad_splits(N) * kernel_size * kernel_size * kernel_size * Cin * Cout * sizeof(float);
}

extern "C" int pn2_ptv3_subm_wgrad_f32(const float* feat, int64_t ldf, const int32_t* nbr, int kernel_size, const float* dout,
                                       int64_t ldo, int N, int Cin, int Cout, float* dweight, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!feat || !nbr || !dout || !dweight || N <= 0 || Cin <= 0 || Cout <= 0 || Cin % 4 || Cout % 4 || ldf % 4 || ldo % 4 ||
        ldf < Cin || ldo < Cout || ((uintptr_t)feat & 15) || ((uintptr_t)dout & 15) || (kernel_size != 3 && kernel_size != 5))
        return PN2_E_BADARG;
    if (!workspace || workspace_bytes < pn2_ptv3_subm_wgrad_workspace_bytes(N, kernel_size, Cin, Cout)) return PN2_E_WORKSPACE;
    const int noff = kernel_size * kernel_size * kernel_size;
    const int nsplit = wgrad_splits(N);
    const int rps = pn2::ceil_div(pn2::ceil_div(N, nsplit), WGK) * WGK;
    hipStream_t s = (hipStream_t)stream;
    const long long mn = (long long)noff * Cin * Cout;
    const int T = (Cin > 64 && Cout > 64) ? 128 : 64;
    const dim3 grid((unsigned)(noff * pn2::ceil_div(Cin, T) * pn2::ceil_div(Cout, T)), (unsigned)nsplit);
    const double by = 4.0 * N * noff * (1.0 + Cout) + 4.0 * nsplit * mn, fl = 2.0 * noff * N * (double)Cin * Cout;
    if (T == 128)
        PN2_LAUNCH("ptv3_subm_wgrad", by, fl, (subm_wgrad_kernel<128>), grid, dim3(256), s, feat, (long long)ldf, nbr, noff, dout,
                   (long long)ldo, N, Cin, Cout, rps, (float*)workspace);
    else
        PN2_LAUNCH("ptv3_subm_wgrad", by, fl, (subm_wgrad_kernel<64>), grid, dim3(256), s, feat, (long long)ldf, nbr, noff, dout,
                   (long long)ldo, N, Cin, Cout, rps, (float*)workspace);
    long long blocks = (mn + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    PN2_LAUNCH("ptv3_subm_wgrad_reduce", 4.0 * (nsplit + 1) * mn, 0, subm_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), s,
               (const float*)workspace, nsplit, mn, dweight);
    PN2_LAUNCH_CHECK();
    return 0;
}
