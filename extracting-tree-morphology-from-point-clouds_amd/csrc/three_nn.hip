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
#include "interp.h"
#include "pn2_common.h"
#include "segments.h"
#include <cstdlib>

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 2048;  // sampled points staged per pass: 32 KiB of LDS

// P (1, 2 or 4) dense points per thread: one broadcast LDS read of a sampled point serves P distance tests, evaluated two
// at a time with packed fp32 instructions.
template <int P>
__global__ __launch_bounds__(kBlock) void three_nn_kernel(const float* __restrict__ xyz1, int64_t ab, int64_t an, int64_t ac,
                                                          const float* __restrict__ xyz2, int64_t bb, int64_t bn, int64_t bc,
                                                          int N, int S, int32_t* __restrict__ out_idx,
                                                          float* __restrict__ out_w, float* __restrict__ out_dist,
                                                          const int* __restrict__ coff, const int* __restrict__ order) {
    __shared__ float4 tile[kTile + 1];  // +1: the loop below reads one entry ahead
    const int b = blockIdx.y;
    // ragged batch: the dense side is a flat channel-first buffer, outputs are packed rows (coff[b] + n)
    const pn2::CloudView cv = pn2::cloud_view(xyz1, ab, an, ac, N, coff, b, 3);
    N = cv.n;
    if ((int)(blockIdx.x * P * kBlock) >= N) return;   // uniform: the grid is sized for the longest cloud
    const size_t row0 = coff ? (size_t)coff[b] : (size_t)b * N;
    // `order` (a permutation of the cloud's points, regular batches only): thread t of the grid takes point order[t].  With
    // a SPATIAL order the 64 dense points of a wavefront are neighbours, their top-3 lists improve on the same few sampled
    // points and the wave-uniform skip of the insertion below holds for most of the scan (in index order nearly every
    // sampled point improves SOME lane's list: 19 instead of 7 instructions per pair).
    float px[P], py[P], pz[P], pn[P], d0[P], d1[P], d2[P];
    int i0[P], i1[P], i2[P], nn[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int pos = (blockIdx.x * P + j) * kBlock + threadIdx.x;
        const int n = pos < N ? (order ? order[(size_t)b * N + pos] : pos) : -1;
        nn[j] = n;
        const float* p = cv.p + (int64_t)(n >= 0 ? n : 0) * cv.sn;
        px[j] = p[0], py[j] = p[cv.sc], pz[j] = p[2 * cv.sc];
        pn[j] = pn2::norm2(px[j], py[j], pz[j]);
        d0[j] = d1[j] = d2[j] = __builtin_inff();
        i0[j] = i1[j] = i2[j] = 0;
    }

    for (int s0 = 0; s0 < S; s0 += kTile) {
        const int cnt = (S - s0) < kTile ? (S - s0) : kTile;
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += kBlock) {
            const float* q = xyz2 + (int64_t)b * bb + (int64_t)(s0 + t) * bn;
            const float x = q[0], y = q[bc], z = q[2 * bc];
            tile[t] = make_float4(-2.0f * x, -2.0f * y, -2.0f * z, pn2::norm2(x, y, z));  // see pn2::sqdist2
        }
        __syncthreads();
        float4 q_next = tile[0];
        for (int t = 0; t < cnt; ++t) {
            const float4 q = q_next;
            q_next = tile[t + 1];  // software pipelining: the LDS latency overlaps the P tests below
            float d[P];
            bool any = false;
            if constexpr (P == 1) {
                const float m = __builtin_fmaf(pz[0], q.z, __builtin_fmaf(py[0], q.y, __fmul_rn(px[0], q.x)));
                d[0] = __fadd_rn(__fadd_rn(m, pn[0]), q.w);
                any = d[0] < d2[0];
            } else {
                const pn2::f2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z}, qw = {q.w, q.w};
#pragma unroll
                for (int j = 0; j + 1 < P; j += 2) {
                    const pn2::f2 dd = pn2::sqdist2(pn2::f2{px[j], px[j + 1]}, pn2::f2{py[j], py[j + 1]},
                                                    pn2::f2{pz[j], pz[j + 1]}, pn2::f2{pn[j], pn[j + 1]}, qx, qy, qz, qw);
                    d[j] = dd.x;
                    d[j + 1] = dd.y;
                    any |= (d[j] < d2[j]) | (d[j + 1] < d2[j + 1]);
                }
            }
            if (__ballot(any)) {  // wave-uniform skip of the insertion once the top-3 lists have settled
                const int s = s0 + t;
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    // scalar copies first: selects between array elements would be lowered to selects of addresses
                    const float D = d[j], D0 = d0[j], D1 = d1[j], D2 = d2[j];
                    const int I0 = i0[j], I1 = i1[j], I2 = i2[j];
                    const bool c2 = D < D2, c1 = D < D1, c0 = D < D0;
                    const float n2 = c2 ? D : D2;
                    const int m2 = c2 ? s : I2;
                    const float n1 = c1 ? D : D1;
                    const int m1 = c1 ? s : I1;
                    d2[j] = c1 ? D1 : n2;
                    i2[j] = c1 ? I1 : m2;
                    d1[j] = c0 ? D0 : n1;
                    i1[j] = c0 ? I0 : m1;
                    d0[j] = c0 ? D : D0;
                    i0[j] = c0 ? s : I0;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int n = nn[j];
        const size_t o = (row0 + (n >= 0 ? n : 0)) * 3;
        if (n >= 0) {
        out_idx[o] = i0[j];
        out_idx[o + 1] = i1[j];
        out_idx[o + 2] = i2[j];
        if (out_dist) {
            out_dist[o] = d0[j];
            out_dist[o + 1] = d1[j];
            out_dist[o + 2] = d2[j];
        }
        // blocks.py:200-203: clamp(min=1e-6), reciprocal, (r0 + r1) + r2, divide
        const float r0 = __fdiv_rn(1.0f, d0[j] < 1e-6f ? 1e-6f : d0[j]);
        const float r1 = __fdiv_rn(1.0f, d1[j] < 1e-6f ? 1e-6f : d1[j]);
        const float r2 = __fdiv_rn(1.0f, d2[j] < 1e-6f ? 1e-6f : d2[j]);
        const float sum = __fadd_rn(__fadd_rn(r0, r1), r2);
        out_w[o] = __fdiv_rn(r0, sum);
        out_w[o + 1] = __fdiv_rn(r1, sum);
        out_w[o + 2] = __fdiv_rn(r2, sum);
        }
    }
}

// (distance, index) order: what an ascending scan with strict comparisons keeps
__device__ __forceinline__ bool lex_lt(float d, int i, float e, int j) { return d < e || (d == e && i < j); }

// Small clouds (the deep levels: a few hundred dense points): TPP threads share one dense point, thread j of the point scans
// the samples j, j + TPP, ... and the TPP top-3 lists are merged in (distance, index) order through lane exchanges -- the
// same three neighbours as one ascending scan, TPP times the wavefronts.  S <= kTile, regular batches.
template <int TPP>
__global__ __launch_bounds__(kBlock) void three_nn_small_kernel(const float* __restrict__ xyz1, int64_t ab, int64_t an, int64_t ac,
                                                                const float* __restrict__ xyz2, int64_t bb, int64_t bn, int64_t bc,
                                                                int N, int S, int32_t* __restrict__ out_idx,
                                                                float* __restrict__ out_w, float* __restrict__ out_dist) {
    __shared__ float4 tile[kTile + TPP];
    const int b = blockIdx.y;
    for (int t = threadIdx.x; t < S + TPP; t += kBlock) {
        const float* q = xyz2 + (int64_t)b * bb + (int64_t)(t < S ? t : S - 1) * bn;
        const float x = q[0], y = q[bc], z = q[2 * bc];
        tile[t] = make_float4(-2.0f * x, -2.0f * y, -2.0f * z, pn2::norm2(x, y, z));
    }
    __syncthreads();
    const int gt = blockIdx.x * kBlock + threadIdx.x, n = gt / TPP, sub = gt % TPP;
    const bool live = n < N;
    const float* p = xyz1 + (int64_t)b * ab + (int64_t)(live ? n : 0) * an;
    const float px = p[0], py = p[ac], pz = p[2 * ac];
    const float pn = pn2::norm2(px, py, pz);
    float d0 = __builtin_inff(), d1 = d0, d2 = d0;
    int i0 = 0, i1 = 0, i2 = 0;   // like three_nn_kernel: an infinite distance never enters a list
    auto insert = [&](float D, int s) {
        const float D0 = d0, D1 = d1, D2 = d2;
        const int I0 = i0, I1 = i1, I2 = i2;
        const bool c2 = lex_lt(D, s, D2, I2), c1 = lex_lt(D, s, D1, I1), c0 = lex_lt(D, s, D0, I0);
        const float n2 = c2 ? D : D2;
        const int m2 = c2 ? s : I2;
        const float n1 = c1 ? D : D1;
        const int m1 = c1 ? s : I1;
        d2 = c1 ? D1 : n2;
        i2 = c1 ? I1 : m2;
        d1 = c0 ? D0 : n1;
        i1 = c0 ? I0 : m1;
        d0 = c0 ? D : D0;
        i0 = c0 ? s : I0;
    };
    float4 q_next = tile[sub];
    for (int t = sub; t < S; t += TPP) {
        const float4 q = q_next;
        q_next = tile[t + TPP];
        const float m = __builtin_fmaf(pz, q.z, __builtin_fmaf(py, q.y, __fmul_rn(px, q.x)));
        const float D = __fadd_rn(__fadd_rn(m, pn), q.w);
        if (__ballot(D < d2)) insert(D, t);   // ascending t per thread: a tie never displaces an earlier sample
    }
#pragma unroll
    for (int o = 1; o < TPP; o <<= 1) {   // the TPP lanes of a point are adjacent: butterfly merge, all end with the full list
        const float e0 = __shfl_xor(d0, o), e1 = __shfl_xor(d1, o), e2 = __shfl_xor(d2, o);
        const int j0 = __shfl_xor(i0, o), j1 = __shfl_xor(i1, o), j2 = __shfl_xor(i2, o);
        insert(e0, j0);
        insert(e1, j1);
        insert(e2, j2);
    }
    if (!live || sub != 0) return;
    const size_t o = ((size_t)b * N + n) * 3;
    out_idx[o] = i0, out_idx[o + 1] = i1, out_idx[o + 2] = i2;
    if (out_dist) out_dist[o] = d0, out_dist[o + 1] = d1, out_dist[o + 2] = d2;
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
                                                                   const float* __restrict__ w, int N, int S, int D,
                                                                   float* __restrict__ out, int64_t out_stride,
                                                                   int64_t out_offset, long long total, int32_t* status,
                                                                   const int* __restrict__ coff, const float* __restrict__ skip,
                                                                   int64_t kb, int64_t kn, int64_t kc, int DK) {
    // skip (regular batches only): [B,N,DK] rows copied into columns [0, DK) of out -- the concat of blocks.py:208 in the
    // same launch; the thread that interpolates channel group c of a row also copies the skip groups c, c + DV, ...
    const int DV = D / V;
    // ragged batch: blockIdx.y = cloud, the x-range covers the longest cloud; idx / w / out are packed rows
    long long e_begin = (long long)blockIdx.x * kBlock + threadIdx.x, e_step = (long long)gridDim.x * kBlock, row0 = 0;
    if (coff) {
        row0 = coff[blockIdx.y];
        total = (long long)(coff[blockIdx.y + 1] - coff[blockIdx.y]) * DV;
    }
    for (long long e = e_begin; e < total; e += e_step) {
        const long long r = row0 + e / DV;  // (b, n)
        const int c = (int)(e % DV) * V;
        const int b = coff ? (int)blockIdx.y : (int)(r / N);
        int j0 = idx[r * 3], j1 = idx[r * 3 + 1], j2 = idx[r * 3 + 2];
        if (((unsigned)j0 >= (unsigned)S) | ((unsigned)j1 >= (unsigned)S) | ((unsigned)j2 >= (unsigned)S)) {
            // untrusted index: do not follow it (row 0 instead) and tell the caller (PN2_STATUS_BAD_INDEX)
            if (status) atomicOr(status, PN2_STATUS_BAD_INDEX);
            j0 = (unsigned)j0 < (unsigned)S ? j0 : 0;
            j1 = (unsigned)j1 < (unsigned)S ? j1 : 0;
            j2 = (unsigned)j2 < (unsigned)S ? j2 : 0;
        }
        const float w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
        const float* base = points2 + (int64_t)b * pb;
        float* o = out + r * out_stride + out_offset + c;
        if (skip) {
            const float* sk = skip + (int64_t)b * kb + (r - (long long)b * N) * kn;
            for (int q = c; q < DK; q += D) {
                if (V == 4)
                    *(float4*)(out + r * out_stride + q) = *(const float4*)(sk + q);
                else
                    out[r * out_stride + q] = sk[(int64_t)q * kc];
            }
        }
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

// Backward of the interpolation: dpoints2[b][idx[n][k]][c] += w[n][k] * dout[n][c].  ~768 dense points feed every
// sampled point at FP1; issued as 3*N*D float atomics (100 M of them onto 512 KB) the op runs at the memory-side
// atomic rate (0.5 ms), and LDS float atomics are no better (~2.6 LDS cycles per LANE, measured).  A full bucketing of
// the 3*N (row, weight) pairs by destination (round 1) removes the atomics but reads every `dout` row three times
// (PMC: 410 MB against 144 MB algorithmic).  Now the rows are bucketed by their NEAREST sampled point only (counting
// sort of N row ids: integer atomics only), one wavefront owns a 64-row chunk of a bucket and reads each of its rows
// ONCE (coalesced 512-byte rows, lane = channel):
//   * the nearest-neighbour term accumulates in registers;
//   * the second and third neighbours of the rows of one bucket are a handful of adjacent sampled points: they
//     accumulate in a per-wavefront LDS table of TIG_SLOTS rows, slot = position of the destination in a lane-distributed
//     tag list (one compare + ballot), lane-private read-modify-write, no atomics;
//   * registers and slots are flushed with one float atomic row each per chunk (a destination beyond the table's
//     capacity falls back to direct atomics).
// Histogram of destinations.  Integer atomics on ~1000 hot global addresses serialise at ~235 ns each, so every
// workgroup first counts its contiguous share of one cloud's rows in LDS and then adds S totals to global memory.
// blockIdx.y = cloud, blockIdx.x = share.
constexpr int TIG_T = 1024;
// the backward kernels see the indices the forward pass already validated (and flagged); clamping keeps a bad one from
// becoming an out-of-bounds LDS / global access
__device__ __forceinline__ int clamp_idx(int j, int S) { return (unsigned)j < (unsigned)S ? j : 0; }
__global__ __launch_bounds__(TIG_T) void tig_count_kernel(const int32_t* __restrict__ idx, int N, int S, int per_block,
                                                          int* __restrict__ hist, const int* __restrict__ coff) {
    extern __shared__ int lh[];  // [S]
    const int b = blockIdx.y;
    const long long row0 = coff ? coff[b] : (long long)b * N;   // ragged batch: packed rows, n_b = coff[b+1] - coff[b]
    if (coff) N = coff[b + 1] - coff[b];
    if ((long long)blockIdx.x * per_block >= N) return;         // uniform
    for (int e = threadIdx.x; e < S; e += TIG_T) lh[e] = 0;
    __syncthreads();
    const long long p0 = (long long)blockIdx.x * per_block, p1 = p0 + per_block < N ? p0 + per_block : N;
    const int32_t* ib = idx + row0 * 3;
    for (long long e = p0 + threadIdx.x; e < p1; e += TIG_T) atomicAdd(&lh[clamp_idx(ib[3 * e], S)], 1);
    __syncthreads();
    for (int e = threadIdx.x; e < S; e += TIG_T)
        if (lh[e]) atomicAdd(hist + (long long)b * S + e, lh[e]);
}

// exclusive scans over hist[0..total) (one block; total = B*S is a few thousand): offs = list offsets (cursor starts
// as a copy), coffs = offsets in units of 64-entry chunks (what the reduce kernel's wavefronts index)
constexpr int TIG_CHUNK = 64;
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
// One block, 16 wavefronts: wavefront w owns a contiguous range of the histogram and walks it in 64-wide (coalesced)
// tiles with a shuffle scan and a register carry -- no block barrier per tile; the 16 range totals are scanned once.
__global__ __launch_bounds__(1024) void tig_scan_kernel(const int* __restrict__ hist, int total, int* __restrict__ offs,
                                                        int* __restrict__ cursor, int* __restrict__ coffs) {
    __shared__ int wsum[16], wcsum[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = ((total + 15) / 16 + 63) / 64 * 64;           // range length, a multiple of the tile
    const int r0 = wv * per, r1 = r0 + per < total ? r0 + per : total;
    int sum = 0, csum = 0;
    for (int base = r0; base < r1; base += 64) {
        const int e = base + lane;
        const int h = e < r1 ? hist[e] : 0;
        sum += h;
        csum += (h + TIG_CHUNK - 1) / TIG_CHUNK;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        csum += __shfl_xor(csum, off, 64);
    }
    if (lane == 0) {
        wsum[wv] = sum;
        wcsum[wv] = csum;
    }
    __syncthreads();
    int carry = 0, ccarry = 0;
    for (int k = 0; k < wv; ++k) {
        carry += wsum[k];
        ccarry += wcsum[k];
    }
    for (int base = r0; base < r1; base += 64) {
        const int e = base + lane;
        const int h = e < r1 ? hist[e] : 0;
        const int c = (h + TIG_CHUNK - 1) / TIG_CHUNK;
        const int incl = wave_incl_scan(h, lane), cincl = wave_incl_scan(c, lane);
        if (e < r1) {
            offs[e] = carry + incl - h;
            cursor[e] = carry + incl - h;
            coffs[e] = ccarry + cincl - c;
        }
        carry += __shfl(incl, 63, 64);
        ccarry += __shfl(cincl, 63, 64);
    }
    if (threadIdx.x == 0) {
        int t = 0, ct = 0;
        for (int k = 0; k < 16; ++k) {
            t += wsum[k];
            ct += wcsum[k];
        }
        offs[total] = t;
        coffs[total] = ct;
    }
}

// Scatter the row ids into their nearest neighbour's list: the workgroup counts its share in LDS again, reserves one
// contiguous range per destination with a single returning global atomic, then ranks its rows inside that range with
// LDS atomics.
__global__ __launch_bounds__(TIG_T) void tig_fill_kernel(const int32_t* __restrict__ idx, int N, int S, int per_block,
                                                         int* __restrict__ cursor, int* __restrict__ list,
                                                         const int* __restrict__ coff) {
    extern __shared__ int lh[];  // [S] counts, then [S] running positions
    const int b = blockIdx.y;
    const long long row0 = coff ? coff[b] : (long long)b * N;
    if (coff) N = coff[b + 1] - coff[b];
    if ((long long)blockIdx.x * per_block >= N) return;         // uniform
    for (int e = threadIdx.x; e < S; e += TIG_T) lh[e] = 0;
    __syncthreads();
    const long long p0 = (long long)blockIdx.x * per_block, p1 = p0 + per_block < N ? p0 + per_block : N;
    const int32_t* ib = idx + row0 * 3;
    for (long long e = p0 + threadIdx.x; e < p1; e += TIG_T) atomicAdd(&lh[clamp_idx(ib[3 * e], S)], 1);
    __syncthreads();
    for (int e = threadIdx.x; e < S; e += TIG_T) {
        const int c = lh[e];
        lh[e] = c ? atomicAdd(cursor + (long long)b * S + e, c) : 0;   // base of this workgroup's range
    }
    __syncthreads();
    for (long long e = p0 + threadIdx.x; e < p1; e += TIG_T) {
        const int pos = atomicAdd(&lh[clamp_idx(ib[3 * e], S)], 1);
        list[pos] = (int)(row0 + e);
    }
}

// one wavefront per 64-row chunk of a destination's list; lane l owns channels l and l + 64 of a 128-channel block
// 16 table slots: the table is what limits the wavefronts -- and with them the bytes in flight -- per compute unit (32 slots:
// 120 us at 262 144 x 1024 x 128, 16: 100 us; scattered neighbours beyond the table go to memory as atomics)
constexpr int TIG_SLOTS = 16;
// DY: `dout` is the gradient with respect to relu(bn(y)); the row that is scattered is the BatchNorm(+ReLU) backward of it,
// dz = scale * (mask(dout) - a - (y - mean) * b) -- the TR_DY operand transform of the chain GEMMs (mlp_tile.h) applied to
// the two channels a lane owns.  The rows of one destination belong to one cloud, hence to one segment: its coefficient
// block is picked once per wavefront.
struct TigDy {
    const float* y;
    const float* coef;
    int relu;
};
template <bool DY, bool H16 = false>   // H16 (with DY): dout and y are __bf16 rows
__global__ __launch_bounds__(kBlock) void tig_reduce_kernel(const float* __restrict__ dout, int64_t out_stride, int64_t out_offset,
                                                            const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                            const int* __restrict__ offs, const int* __restrict__ coffs,
                                                            const int* __restrict__ list, int BS, int S, int D,
                                                            float* __restrict__ dpoints2, const TigDy dy, const SegTable st) {
    __shared__ float tab[kBlock / 64][TIG_SLOTS][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + wv));
    if (wave >= coffs[BS]) return;
    int lo_d = 0, hi_d = BS;  // destination with coffs[dest] <= wave < coffs[dest + 1]
    while (hi_d - lo_d > 1) {
        const int mid = (lo_d + hi_d) >> 1;
        if (coffs[mid] <= wave) lo_d = mid; else hi_d = mid;
    }
    const int dest = lo_d, cloud_base = (dest / S) * S;
    const int lo = offs[dest] + (wave - coffs[dest]) * TIG_CHUNK, end = offs[dest + 1];
    const int hi = lo + TIG_CHUNK < end ? lo + TIG_CHUNK : end;
    const float* coef = dy.coef;
    if (DY && st.nseg > 1 && lo < hi) coef += (long long)seg_of_row(st, __builtin_amdgcn_readfirstlane(list[lo])) * ST_ROWS * D;
    for (int c0 = 0; c0 < D; c0 += 128) {
        float a0 = 0.0f, a1 = 0.0f;
        // the lane's two channels of the 128-channel block: l and l + 64, or -- bfloat16 rows -- the adjacent pair 2l, 2l + 1
        // (one 4-byte load per row instead of two 2-byte ones)
        const int ca = H16 ? c0 + 2 * lane : c0 + lane, cb = H16 ? c0 + 2 * lane + 1 : c0 + 64 + lane;
        float km[2], ks[2], kb[2], ka[2], kq[2];    // DY: the coefficients of channels ca, cb
        if (DY) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = (h ? cb : ca) < D ? (h ? cb : ca) : D - 1;
                km[h] = coef[ST_MEAN * D + cc], ks[h] = coef[ST_SCALE * D + cc], kb[h] = coef[ST_BETA * D + cc];
                ka[h] = coef[ST_A * D + cc], kq[h] = coef[ST_B * D + cc];
            }
        }
        int tag = -1, nslots = 0;          // lane s < nslots: destination held by table slot s
        constexpr int U = 4;               // rows in flight: their ids, neighbour lists and data are fetched together
        for (int e0 = lo; e0 < hi; e0 += U) {
            int row[U], i1[U], i2[U];
            float w0[U], w1[U], w2[U], v0[U], v1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) row[u] = __builtin_amdgcn_readfirstlane(list[e0 + u < hi ? e0 + u : hi - 1]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* src = dout + (long long)row[u] * out_stride + out_offset;
                if (H16) {   // (D and the row strides are even, ca is even: 4-byte aligned pairs)
                    const unsigned short* s16 = (const unsigned short*)dout + (long long)row[u] * out_stride + out_offset;
                    const unsigned pr = cb < D ? *(const unsigned*)(s16 + ca) : 0u;
                    v0[u] = __uint_as_float(pr << 16);
                    v1[u] = __uint_as_float(pr & 0xFFFF0000u);
                } else {
                    v0[u] = ca < D ? src[ca] : 0.0f;
                    v1[u] = cb < D ? src[cb] : 0.0f;
                }
                if (DY) {
                    const float* yr = dy.y + (long long)row[u] * D;
                    const unsigned short* yh = (const unsigned short*)dy.y + (long long)row[u] * D;
                    const unsigned ypr = (H16 && cb < D) ? *(const unsigned*)(yh + ca) : 0u;
                    const float y0 = H16 ? __uint_as_float(ypr << 16) : (ca < D ? yr[ca] : 0.0f);
                    const float y1 = H16 ? __uint_as_float(ypr & 0xFFFF0000u) : (cb < D ? yr[cb] : 0.0f);
                    const float t0 = __builtin_fmaf(y0 - km[0], ks[0], kb[0]), t1 = __builtin_fmaf(y1 - km[1], ks[1], kb[1]);
                    const float g0 = (!dy.relu || t0 > 0.0f) ? v0[u] : 0.0f, g1 = (!dy.relu || t1 > 0.0f) ? v1[u] : 0.0f;
                    v0[u] = ca < D ? ks[0] * (g0 - ka[0] - (y0 - km[0]) * kq[0]) : 0.0f;
                    v1[u] = cb < D ? ks[1] * (g1 - ka[1] - (y1 - km[1]) * kq[1]) : 0.0f;
                }
                const int32_t* ir = idx + (long long)row[u] * 3;
                const float* wr = w + (long long)row[u] * 3;
                i1[u] = ir[1], i2[u] = ir[2];
                w0[u] = wr[0], w1[u] = wr[1], w2[u] = wr[2];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (e0 + u >= hi) break;   // uniform
                a0 += __fmul_rn(v0[u], w0[u]);
                a1 += __fmul_rn(v1[u], w0[u]);
#pragma unroll
                for (int k = 1; k < 3; ++k) {
                    const int d = cloud_base + clamp_idx(k == 1 ? i1[u] : i2[u], S);
                    const float wk = k == 1 ? w1[u] : w2[u];
                    const unsigned long long hit = __ballot(tag == d);
                    int slot;
                    if (hit) {
                        slot = (int)__builtin_ctzll(hit);
                    } else if (nslots < TIG_SLOTS) {
                        slot = nslots++;
                        if (lane == slot) tag = d;
                        tab[wv][slot][lane] = 0.0f;
                        tab[wv][slot][lane + 64] = 0.0f;
                    } else {   // table full (scattered neighbours): this contribution goes straight to memory
                        float* q = dpoints2 + (long long)d * D;
                        if (ca < D) atomicAdd(q + ca, __fmul_rn(v0[u], wk));
                        if (cb < D) atomicAdd(q + cb, __fmul_rn(v1[u], wk));
                        continue;
                    }
                    tab[wv][slot][lane] += __fmul_rn(v0[u], wk);       // lane-private cells: plain read-modify-write
                    tab[wv][slot][lane + 64] += __fmul_rn(v1[u], wk);
                }
            }
        }
        float* dst = dpoints2 + (long long)dest * D;
        if (ca < D) atomicAdd(dst + ca, a0);
        if (cb < D) atomicAdd(dst + cb, a1);
        for (int sl = 0; sl < nslots; ++sl) {
            const int d = __builtin_amdgcn_readlane(tag, sl);
            float* q = dpoints2 + (long long)d * D;
            if (ca < D) atomicAdd(q + ca, tab[wv][sl][lane]);
            if (cb < D) atomicAdd(q + cb, tab[wv][sl][lane + 64]);
        }
    }
}

// fallback for sampled sets too large for an LDS slice: plain global atomics
__global__ __launch_bounds__(kBlock) void three_interpolate_grad_global_kernel(const float* __restrict__ dout, int64_t out_stride,
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
        for (int k = 0; k < 3; ++k) atomicAdd(base + (int64_t)clamp_idx(idx[r * 3 + k], S) * D, __fmul_rn(g, w[r * 3 + k]));
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
                                float* out_dist, const int32_t* order, void* stream) {
    if (!xyz1 || !xyz2 || !out_idx || !out_w || B <= 0 || N <= 0 || S < 3 || B > 65535) return PN2_E_BADARG;
    // points per thread: as many as still leave every SIMD at least one wavefront
    // measured on MI355X at 262144 x 1024: P = 1 128 us, P = 2 152 us, P = 4 190 us -- the top-3 insertion, not the
    // LDS read or the distance, is what the loop spends its issue slots on, and fewer, fatter wavefronts hide less
    int P = 1;
    if (const char* e = getenv("PN2_TNN_P")) P = atoi(e);
    const double tnn_bytes = (double)B * (12.0 * N + 12.0 * S + 36.0 * N);
    // small clouds: several threads per dense point (a launch of a handful of wavefronts is all scan latency otherwise)
    if (!order && S <= kTile && (long long)B * N <= 16384 && S >= 32 && !getenv("PN2_TNN_NO_SMALL")) {
        constexpr int TPP = 8;
        PN2_LAUNCH("three_nn", tnn_bytes, 8.0 * B * (double)N * S, (three_nn_small_kernel<TPP>),
                   dim3(pn2::ceil_div((long long)N * TPP, kBlock), B), dim3(kBlock), (hipStream_t)stream, xyz1, ab, an, ac, xyz2, bb,
                   bn, bc, N, S, out_idx, out_w, out_dist);
        PN2_LAUNCH_CHECK();
        return 0;
    }
#define PN2_TNN_CASE(P_)                                                                                              \
    if (P == P_)                                                                                                      \
        PN2_LAUNCH("three_nn", tnn_bytes, 8.0 * B * (double)N * S, (three_nn_kernel<P_>), dim3(pn2::ceil_div(N, kBlock * P_), B), dim3(kBlock), \
                   (hipStream_t)stream, xyz1, ab, an, ac, xyz2, bb, bn, bc, N, S, out_idx, out_w, out_dist, (const int*)nullptr, (const int*)order);
    PN2_TNN_CASE(1)
    PN2_TNN_CASE(2)
    PN2_TNN_CASE(4)
#undef PN2_TNN_CASE
    if (P != 1 && P != 2 && P != 4) return PN2_E_BADARG;
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

// skip == nullptr: plain interpolation into columns [out_offset, out_offset + D)
static int interpolate_run(const float* points2, int64_t pb, int64_t pn, int64_t pc, const int32_t* idx, const float* w, int B,
                           int N, int S, int D, float* out, int64_t out_stride, int64_t out_offset, int32_t* status, void* stream,
                           const float* skip, int64_t kb, int64_t kn, int64_t kc, int DK) {
    if (!points2 || !idx || !w || !out || B <= 0 || N <= 0 || S <= 0 || D <= 0 || out_stride < out_offset + D)
        return PN2_E_BADARG;
    if (skip && (DK <= 0 || DK > out_offset)) return PN2_E_BADARG;
    bool vec = pc == 1 && D % 4 == 0 && pn % 4 == 0 && pb % 4 == 0 && out_stride % 4 == 0 && out_offset % 4 == 0 &&
               ((uintptr_t)points2 % 16 == 0) && ((uintptr_t)out % 16 == 0);
    if (skip) vec = vec && kc == 1 && DK % 4 == 0 && kn % 4 == 0 && kb % 4 == 0 && ((uintptr_t)skip % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    const double ti_bytes = (double)B * N * (36.0 + 4.0 * D + (skip ? 8.0 * DK : 0.0)) + 4.0 * B * S * D;
    if (vec) {
        const long long total = (long long)B * N * (D / 4);
        PN2_LAUNCH("three_interpolate", ti_bytes, 0, (three_interpolate_kernel<4>), dim3(grid_for(total)), dim3(kBlock), s,
                   points2, pb, pn, pc, idx, w, N, S, D, out, out_stride, out_offset, total, status, (const int*)nullptr, skip, kb,
                   kn, kc, DK);
    } else {
        const long long total = (long long)B * N * D;
        PN2_LAUNCH("three_interpolate", ti_bytes, 0, (three_interpolate_kernel<1>), dim3(grid_for(total)), dim3(kBlock), s,
                   points2, pb, pn, pc, idx, w, N, S, D, out, out_stride, out_offset, total, status, (const int*)nullptr, skip, kb,
                   kn, kc, DK);
    }
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_three_interpolate_f32(const float* points2, int64_t pb, int64_t pn, int64_t pc, const int32_t* idx,
                                         const float* w, int B, int N, int S, int D, float* out, int64_t out_stride,
                                         int64_t out_offset, int32_t* status, void* stream) {
    return interpolate_run(points2, pb, pn, pc, idx, w, B, N, S, D, out, out_stride, out_offset, status, stream, nullptr, 0, 0, 0,
                           0);
}

// ... and the skip connection's rows points1 [B,N,D1] (strided) into columns [0, D1) of out in the same launch: the whole
// cat([points1, interpolated], -1) of blocks.py:208.  out_offset >= D1.
extern "C" int pn2_three_interpolate_concat_f32(const float* points1, int64_t kb, int64_t kn, int64_t kc, int D1,
                                                const float* points2, int64_t pb, int64_t pn, int64_t pc, const int32_t* idx,
                                                const float* w, int B, int N, int S, int D, float* out, int64_t out_stride,
                                                int64_t out_offset, int32_t* status, void* stream) {
    if (!points1) return PN2_E_BADARG;
    return interpolate_run(points2, pb, pn, pc, idx, w, B, N, S, D, out, out_stride, out_offset, status, stream, points1, kb, kn,
                           kc, D1);
}

// bucketing pays off once the op is large; below this many atomics the direct kernel is faster
static bool tig_sorted(int B, int N, int S, int D) { return (long long)B * N * D >= (1LL << 22) && S <= 8192 && B <= 65535; }

// hist, offs (+1), cursor, coffs: ints; list: row ids
static size_t tig_workspace(int B, long long rows, int S) {
    return (size_t)(4 * ((size_t)B * S + 4)) * sizeof(int) + 16 + (size_t)rows * sizeof(int);
}

extern "C" size_t pn2_three_interpolate_grad_workspace_bytes(int B, int N, int S, int D) {
    if (B <= 0 || N <= 0 || S <= 0 || D <= 0 || !tig_sorted(B, N, S, D)) return 16;
    return tig_workspace(B, (long long)B * N, S);
}

// Shared body: regular batches (coff == nullptr, B clouds of N rows) and ragged ones (coff [B+1], N = longest cloud, `rows`
// packed rows in total).  Ragged batches always take the bucketed path.
static int tig_run(const float* dout, int64_t out_stride, int64_t out_offset, const int32_t* idx, const float* w, int B, int N,
                   int S, int D, float* dpoints2, void* workspace, size_t workspace_bytes, void* stream, const int* coff,
                   long long rows, const pn2::interp::DySource* dy = nullptr) {
    hipStream_t s = (hipStream_t)stream;
    if (dy && (!dy->y || !dy->coef || out_stride != D || out_offset != 0 || S > 8192 || B > 65535 || dy->nseg > kMaxSegs ||
               (dy->nseg > 1 && !dy->row_off) || (dy->rows_bf16 && D % 2)))
        return PN2_E_BADARG;
    PN2_HIP_CHECK(hipMemsetAsync(dpoints2, 0, (size_t)B * S * D * sizeof(float), s));
    const double bytes = (double)rows * (36.0 + (dy ? (dy->rows_bf16 ? 4.0 : 8.0) : 4.0) * D) + 4.0 * B * S * D;
    if (!dy && !coff && !tig_sorted(B, N, S, D)) {
        const long long total = (long long)B * N * D;
        PN2_LAUNCH("three_interpolate_grad", bytes, 0, three_interpolate_grad_global_kernel, dim3(grid_for(total)), dim3(kBlock), s,
                   dout, out_stride, out_offset, idx, w, N, S, D, dpoints2, total);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    if (!workspace || workspace_bytes < tig_workspace(B, rows, S)) return PN2_E_WORKSPACE;
    const int BS = B * S;
    int* hist = (int*)workspace;
    int* offs = hist + BS + 4;
    int* cursor = offs + BS + 4;
    int* coffs = cursor + BS + 4;
    int* list = (int*)(((uintptr_t)(coffs + BS + 4) + 15) & ~(uintptr_t)15);
    PN2_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)(BS + 4) * sizeof(int), s));
    // ~64 workgroups per launch keep the per-destination global atomics few; S ints of LDS each
    int shares = 64 / B;
    if (shares < 1) shares = 1;
    int per_block = pn2::ceil_div(pn2::ceil_div(N, shares), TIG_T) * TIG_T;
    shares = pn2::ceil_div(N, per_block);
    const size_t lds = (size_t)S * sizeof(int);
    {
        pn2::prof::Scope sc_("tig_count", s, 8.0 * rows, 0);
        hipLaunchKernelGGL(tig_count_kernel, dim3(shares, B), dim3(TIG_T), lds, s, idx, N, S, per_block, hist, coff);
    }
    PN2_LAUNCH("tig_scan", 12.0 * BS, 0, tig_scan_kernel, dim3(1), dim3(1024), s, (const int*)hist, BS, offs, cursor, coffs);
    {
        pn2::prof::Scope sc_("tig_fill", s, 8.0 * rows, 0);
        hipLaunchKernelGGL(tig_fill_kernel, dim3(shares, B), dim3(TIG_T), lds, s, idx, N, S, per_block, cursor, list, coff);
    }
    // sum over destinations of ceil(len / 64) <= rows / 64 + B*S: wavefronts beyond the real chunk count exit at once
    const long long waves = rows / TIG_CHUNK + BS + 1;
    if (waves > 0x7FFFFFFFLL / 64) return PN2_E_BADARG;
    SegTable st{};
    st.nseg = 1;
    if (dy) {
        st.nseg = dy->nseg > 1 ? dy->nseg : 1;
        for (int i = 0; i <= st.nseg && dy->nseg > 1; ++i) st.row_off[i] = dy->row_off[i];
        const TigDy td{dy->y, dy->coef, dy->relu};
        if (dy->rows_bf16)
            PN2_LAUNCH("three_interpolate_grad", bytes, 0, (tig_reduce_kernel<true, true>), dim3((unsigned)pn2::ceil_div(waves, kBlock / 64)),
                       dim3(kBlock), s, dout, out_stride, out_offset, idx, w, (const int*)offs, (const int*)coffs, (const int*)list,
                       BS, S, D, dpoints2, td, st);
        else
        PN2_LAUNCH("three_interpolate_grad", bytes, 0, (tig_reduce_kernel<true>), dim3((unsigned)pn2::ceil_div(waves, kBlock / 64)),
                   dim3(kBlock), s, dout, out_stride, out_offset, idx, w, (const int*)offs, (const int*)coffs, (const int*)list, BS,
                   S, D, dpoints2, td, st);
    } else {
        PN2_LAUNCH("three_interpolate_grad", bytes, 0, (tig_reduce_kernel<false>), dim3((unsigned)pn2::ceil_div(waves, kBlock / 64)),
                   dim3(kBlock), s, dout, out_stride, out_offset, idx, w, (const int*)offs, (const int*)coffs, (const int*)list, BS,
                   S, D, dpoints2, TigDy{nullptr, nullptr, 0}, st);
    }
    PN2_LAUNCH_CHECK();
    return 0;
}

size_t pn2::interp::grad_workspace_bytes(int B, long long rows, int S) { return tig_workspace(B, rows, S); }
int pn2::interp::grad(const float* dout, int64_t out_stride, int64_t out_offset, const int32_t* idx, const float* w, int B, int N,
                      int S, int D, float* dpoints2, void* workspace, size_t workspace_bytes, hipStream_t stream, const int* coff,
                      long long rows, const DySource* dy) {
    return tig_run(dout, out_stride, out_offset, idx, w, B, N, S, D, dpoints2, workspace, workspace_bytes, (void*)stream, coff,
                   rows, dy);
}

extern "C" int pn2_three_interpolate_grad_f32(const float* dout, int64_t out_stride, int64_t out_offset,
                                              const int32_t* idx, const float* w, int B, int N, int S, int D,
                                              float* dpoints2, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !idx || !w || !dpoints2 || B <= 0 || N <= 0 || S <= 0 || D <= 0 || out_stride < out_offset + D)
        return PN2_E_BADARG;
    return tig_run(dout, out_stride, out_offset, idx, w, B, N, S, D, dpoints2, workspace, workspace_bytes, stream, nullptr,
                   (long long)B * N);
}

// ------------------------------------------------------------------------------------------------ ragged batches
// Whole-tree execution (pn2_hip.h "Ragged clouds"): the dense side is C clouds of n_b = coff[b+1] - coff[b] points in one
// flat channel-first buffer, the sampled side is regular [C,S,*]; idx / w / interpolated rows are PACKED rows
// (row = coff[b] + n), `rows` = coff[C].
extern "C" int pn2_three_nn_ragged_f32(const float* xyz1_cf, const int32_t* coff, const float* xyz2, int C, int n_max, int S,
                                       int32_t* out_idx, float* out_w, void* stream) {
    if (!xyz1_cf || !coff || !xyz2 || !out_idx || !out_w || C <= 0 || n_max <= 0 || S < 3 || C > 65535) return PN2_E_BADARG;
    PN2_LAUNCH("three_nn", (double)C * (48.0 * n_max + 12.0 * S), 8.0 * C * (double)n_max * S, (three_nn_kernel<1>),
               dim3(pn2::ceil_div(n_max, kBlock), C), dim3(kBlock), (hipStream_t)stream, xyz1_cf, 0, 1, 0, xyz2, (int64_t)S * 3, 3,
               1, n_max, S, out_idx, out_w, (float*)nullptr, (const int*)coff, (const int*)nullptr);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_three_interpolate_ragged_f32(const float* points2, const int32_t* idx, const float* w, const int32_t* coff,
                                                int C, int n_max, long long rows, int S, int D, float* out, int64_t out_stride,
                                                int64_t out_offset, int32_t* status, void* stream) {
    if (!points2 || !idx || !w || !coff || !out || C <= 0 || n_max <= 0 || rows <= 0 || S <= 0 || D <= 0 ||
        out_stride < out_offset + D || C > 65535)
        return PN2_E_BADARG;
    const bool vec = D % 4 == 0 && out_stride % 4 == 0 && out_offset % 4 == 0 && ((uintptr_t)points2 % 16 == 0) &&
                     ((uintptr_t)out % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    const double ti_bytes = (double)rows * (36.0 + 4.0 * D) + 4.0 * C * S * D;
    const int64_t pb = (int64_t)S * D, pn = D;
    if (vec) {
        const long long per = (long long)n_max * (D / 4);
        PN2_LAUNCH("three_interpolate", ti_bytes, 0, (three_interpolate_kernel<4>), dim3(grid_for(per), C), dim3(kBlock), s, points2,
                   pb, pn, (int64_t)1, idx, w, n_max, S, D, out, out_stride, out_offset, per, status, (const int*)coff, (const float*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, 0);
    } else {
        const long long per = (long long)n_max * D;
        PN2_LAUNCH("three_interpolate", ti_bytes, 0, (three_interpolate_kernel<1>), dim3(grid_for(per), C), dim3(kBlock), s, points2,
                   pb, pn, (int64_t)1, idx, w, n_max, S, D, out, out_stride, out_offset, per, status, (const int*)coff, (const float*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, 0);
    }
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pn2_three_interpolate_grad_ragged_workspace_bytes(int C, long long rows, int S) {
    if (C <= 0 || rows <= 0 || S <= 0) return 16;
    return tig_workspace(C, rows, S);
}

extern "C" int pn2_three_interpolate_grad_ragged_f32(const float* dout, int64_t out_stride, int64_t out_offset, const int32_t* idx,
                                                     const float* w, const int32_t* coff, int C, int n_max, long long rows, int S,
                                                     int D, float* dpoints2, void* workspace, size_t workspace_bytes,
                                                     void* stream) {
    if (!dout || !idx || !w || !coff || !dpoints2 || C <= 0 || n_max <= 0 || rows <= 0 || S <= 0 || S > 8192 || D <= 0 ||
        out_stride < out_offset + D || C > 65535)
        return PN2_E_BADARG;
    return tig_run(dout, out_stride, out_offset, idx, w, C, n_max, S, D, dpoints2, workspace, workspace_bytes, stream,
                   (const int*)coff, rows);
}
