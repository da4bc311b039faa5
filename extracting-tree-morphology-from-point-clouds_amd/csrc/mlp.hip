// Pointwise MLP chains (1x1 conv -> train/eval BatchNorm -> ReLU [-> max over K]) for gfx950.
// Replaces the Conv2d/BatchNorm2d/ReLU/max stack of set abstraction (Modules/PointNet2/blocks.py:93-98), the
// Conv1d/BatchNorm1d/ReLU stack of feature propagation (:213-215) and ConvHead (:7-35), forward and backward.
//
// Every layer is a rows x C_in -> rows x C_out contraction over channels-last rows, so all of them run on one
// LDS-tiled fp32 MFMA kernel (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, the precision the reference
// computes in).  Block tile 128 x 128 x 32, four wavefronts, each owning a 64 x 64 sub-tile = 2 x 2 MFMA
// accumulators; operands are staged K-major in LDS ([k][m], row pad 4 floats) so that the MFMA operand read
// (lane -> m = lane & 31, k = lane >> 5) is one conflict-free ds_read_b32; the next K-tile is fetched into
// registers while the current one is multiplied (one barrier per K-tile, two LDS buffers).
//
// What makes it a *chain* kernel is the operand transform applied while a tile is staged:
//   TR_BNRELU   x = relu((y - mean) * scale + beta)      the previous layer's BatchNorm+ReLU, never materialised
//   TR_DY       dy = scale * (mask(dz) - a - (y - mean) * b)   BatchNorm+ReLU backward, mask = [(y-mean)*scale+beta > 0]
// so a layer's pre-BN output Y is written once and read by whoever needs it, and train-mode BatchNorm costs no
// extra pass over the activations: the forward epilogue emits per-64-row (mean, M2) partials that a small
// kernel merges in float64 (Chan's parallel variance: as accurate as two-pass).
//
// Three GEMM roles share the kernel:  forward  Y = X W^T + b        (A rows x Cin, B = W [Cout][Cin])
//                                     dgrad    dX = dY W            (A rows x Cout, B = W as [K][N])
//                                     wgrad    dW = dY^T X          (reduction over rows, split over blocks,
//                                                                    fixed-order slab sum -> deterministic)
#include "chain_coop.h"
#include "mlp_tile.h"

namespace {

// (chunk, which) element of one channel's statistics partials in either layout (GemmArgs::pstride)
struct PartialView {
    const float* p;
    int C, c;
    long long cm;
    __device__ __forceinline__ float operator()(int k, int which) const {
        return cm ? p[(long long)c * cm + 2 * (long long)k + which] : p[((long long)k * 2 + which) * C + c];
    }
};

// ------------------------------------------------------------------------------------------- BatchNorm: forward
// One block per channel: merge the CH-row (mean, M2) partials in float64, write the coefficient block and
// update the running statistics exactly like nn.BatchNorm (biased variance for normalisation, unbiased for
// running_var).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial_, int rows, int CH, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float eps, float momentum, float* __restrict__ coef, long long cm) {
    __shared__ double red[256];
    __shared__ double s_mean;
    const int c = blockIdx.x, t = threadIdx.x;
    // cm != 0: channel-major partials (the GEMM epilogues of single-segment chains): this channel's (mean, M2) pairs are
    // contiguous -- coalesced reads instead of one useful float per cache line.  Same values, same summation order.
    const PartialView partial{partial_, C, c, cm};
    const int nchunk = (rows + CH - 1) / CH;
    // Up to KEEP chunks per thread stay in registers between the two reductions (the strided partials are read once:
    // one useful float per cache line, the read IS the kernel's time); longer columns re-read.
    constexpr int KEEP = 16;
    const bool keep = nchunk <= KEEP * 256;   // uniform
    float pm[KEEP], pv[KEEP];
    double acc = 0.0;
    if (keep) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int k = t + 256 * i;
            const bool on = k < nchunk;
            pm[i] = on ? partial(k, 0) : 0.0f;
            pv[i] = on ? partial(k, 1) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int k = t + 256 * i;
            const int n = k < nchunk ? (rows - k * CH < CH ? rows - k * CH : CH) : 0;
            acc += (double)n * (double)pm[i];
        }
    } else {
#pragma unroll 8
        for (int k = t; k < nchunk; k += 256) {  // independent strided loads: keep several in flight
            const int n = rows - k * CH < CH ? rows - k * CH : CH;
            acc += (double)n * (double)partial(k, 0);
        }
    }
    red[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) s_mean = red[0] / (double)rows;
    __syncthreads();
    const double mean = s_mean;
    acc = 0.0;
    if (keep) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int k = t + 256 * i;
            if (k < nchunk) {
                const int n = rows - k * CH < CH ? rows - k * CH : CH;
                const double d = (double)pm[i] - mean;
                acc += (double)pv[i] + (double)n * d * d;
            }
        }
    } else {
#pragma unroll 8
        for (int k = t; k < nchunk; k += 256) {
            const int n = rows - k * CH < CH ? rows - k * CH : CH;
            const double d = (double)partial(k, 0) - mean;
            acc += (double)partial(k, 1) + (double)n * d * d;
        }
    }
    __syncthreads();
    red[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) {
        const double var = red[0] / (double)rows;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
        coef[ST_MEAN * C + c] = (float)mean;
        coef[ST_VAR * C + c] = (float)var;
        coef[ST_INVSTD * C + c] = invstd;
        coef[ST_SCALE * C + c] = gm * invstd;
        coef[ST_BETA * C + c] = bt;
        if (running_mean) {
            const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    }
}

// one block per channel: s1, s2 in float64 -> dgamma, dbeta and the dY coefficients (a = s1/R, b = invstd*s2/R)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial_, int nblk, int rows, int C,
                                                              float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, long long cm) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const PartialView partial{partial_, C, c, cm};
    double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
    for (int k = t; k < nblk; k += 256) {
        a1 += (double)partial(k, 0);
        a2 += (double)partial(k, 1);
    }
    r1[t] = a1;
    r2[t] = a2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            r1[t] += r1[t + s];
            r2[t] += r2[t + s];
        }
        __syncthreads();
    }
    if (t == 0) {
        const double s1 = r1[0], s2 = r2[0];
        coef[ST_A * C + c] = (float)(s1 / (double)rows);
        coef[ST_B * C + c] = (float)((double)coef[ST_INVSTD * C + c] * s2 / (double)rows);
        if (dbeta) dbeta[c] += (float)s1;
        if (dgamma) dgamma[c] += (float)s2;
    }
}

// Small layers (at most 64 partial chunks): one WAVEFRONT per channel, lane = chunk, float64 sums by shuffles -- no LDS, no
// workgroup barriers (the block-per-channel kernels spend their 5 us on two 8-step barrier reductions over mostly empty
// threads).  Same formulas as bn_finalize_kernel / bn_bwd_finalize_kernel.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__global__ __launch_bounds__(256) void bn_finalize_wave_kernel(const float* __restrict__ partial_, int rows, int CH, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float eps, float momentum, float* __restrict__ coef, long long cm) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), k = threadIdx.x & 63;
    if (c >= C) return;
    const int nchunk = (rows + CH - 1) / CH;
    const PartialView partial{partial_, C, c, cm};
    const bool on = k < nchunk;
    const int n = on ? (rows - k * CH < CH ? rows - k * CH : CH) : 0;
    const double pm = on ? (double)partial(k, 0) : 0.0, pv = on ? (double)partial(k, 1) : 0.0;
    const double mean = wave_sum_f64((double)n * pm) / (double)rows;
    const double d = pm - mean;
    const double var = wave_sum_f64(pv + (double)n * d * d) / (double)rows;
    if (k == 0) {
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
        coef[ST_MEAN * C + c] = (float)mean;
        coef[ST_VAR * C + c] = (float)var;
        coef[ST_INVSTD * C + c] = invstd;
        coef[ST_SCALE * C + c] = gm * invstd;
        coef[ST_BETA * C + c] = bt;
        if (running_mean) {
            const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    }
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_wave_kernel(const float* __restrict__ partial_, int nblk, int rows, int C,
                                                                   float* __restrict__ coef, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, long long cm) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), k = threadIdx.x & 63;
    if (c >= C) return;
    const PartialView partial{partial_, C, c, cm};
    const bool on = k < nblk;
    const double s1 = wave_sum_f64(on ? (double)partial(k, 0) : 0.0), s2 = wave_sum_f64(on ? (double)partial(k, 1) : 0.0);
    if (k == 0) {
        coef[ST_A * C + c] = (float)(s1 / (double)rows);
        coef[ST_B * C + c] = (float)((double)coef[ST_INVSTD * C + c] * s2 / (double)rows);
        if (dbeta) dbeta[c] += (float)s1;
        if (dgamma) dgamma[c] += (float)s2;
    }
}

// rows of statistics chunk k of segment s: blocks of `cpb` chunks of `crows` rows tile the segment from its first row
__device__ __forceinline__ int chunk_rows_in_seg(const SegTable& st, int s, int k, int cpb, int crows) {
    const int row0 = st.row_off[s] + (k - st.blk_off[s] * cpb) * crows;
    const int left = st.row_off[s + 1] - row0;
    return left < crows ? left : crows;
}

// ---- BatchNorm finalize, forward and backward, in two coalesced stages -----------------------------------------------
// The GEMM epilogues (and the reduction kernels) leave per-chunk partials [chunk][2][C].  A block per CHANNEL reading
// them walks memory with a stride of 2*C floats -- one useful float per cache line (20 us at 4096 chunks, 100+ us with
// 13 000 chunks of a whole tree).  Instead:
//   stage 1  grid (C/64, slices, segments): a block sums a SLICE of 64 consecutive chunks of one segment for 64 adjacent
//            channels -- lane = channel, so every load is a full 256-byte row of the partial array -- and writes three
//            (two) float64 sums per channel;
//   stage 2  grid C/64: one wavefront per segment (lane = channel) adds the segment's slices in order, writes the
//            coefficient block, then one thread per channel walks the segments in order for what is sequential by
//            nature (running statistics: one momentum update per mini-batch; dgamma / dbeta: a fixed-order sum).
// Forward statistics use the segment's first chunk mean as a pivot p: with d_k = m_k - p,
//   mean = p + sum(n_k d_k) / N,   var = (sum M2_k + sum n_k d_k^2 - N (mean - p)^2) / N      (float64)
// -- the single-pass form of Chan's merge; the pivot removes the cancellation a raw sum of squares would have.
constexpr int SLICE = 64;   // chunks per slice
__device__ __forceinline__ int seg_chunks(const SegTable& st, int s, int cpb) { return (st.blk_off[s + 1] - st.blk_off[s]) * cpb; }
__device__ __forceinline__ int seg_slices(const SegTable& st, int s, int cpb) { return (seg_chunks(st, s, cpb) + SLICE - 1) / SLICE; }

constexpr int FIN_W = 16;   // wavefronts per block of stage 2

template <bool FWD>
__global__ __launch_bounds__(256) void bn_slice_reduce_kernel(const float* __restrict__ partial, const SegTable st, int cpb,
                                                              int crows, int C, double* __restrict__ sp) {
    constexpr int NV = FWD ? 3 : 2;
    __shared__ double red[4][NV][64];
    const int s = blockIdx.z, j = blockIdx.y;
    const int nchunks = seg_chunks(st, s, cpb);
    if (j * SLICE >= nchunks) return;                        // uniform: the grid is sized for the longest segment
    int slot = j;
    for (int t = 0; t < s; ++t) slot += seg_slices(st, t, cpb);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < C;
    const int k0 = st.blk_off[s] * cpb;
    const int kb = k0 + j * SLICE, ke = (j + 1) * SLICE < nchunks ? kb + SLICE : k0 + nchunks;
    const float pivot = (FWD && ok) ? partial[((long long)k0 * 2) * C + c] : 0.0f;   // chunk k0 always holds rows
    double a = 0.0, b = 0.0, q = 0.0;
    for (int k = kb + w; k < ke; k += 4) {
        const int n = chunk_rows_in_seg(st, s, k, cpb, crows);
        if (n <= 0 || !ok) continue;
        const float v0 = partial[((long long)k * 2) * C + c], v1 = partial[((long long)k * 2 + 1) * C + c];
        if (FWD) {
            const double d = (double)v0 - (double)pivot;
            a += (double)n * d;
            b += (double)n * d * d;
            q += (double)v1;
        } else {
            a += (double)v0;
            b += (double)v1;
        }
    }
    red[w][0][lane] = a;
    red[w][1][lane] = b;
    if (FWD) red[w][NV - 1][lane] = q;
    __syncthreads();
    if (w == 0 && ok) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            sp[((long long)slot * NV + v) * C + c] = (red[0][v][lane] + red[1][v][lane]) + (red[2][v][lane] + red[3][v][lane]);
    }
}

__device__ __forceinline__ float ld_coherent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Stage 2: grid (C/64, segments), one wavefront per slice of the segment (lane = channel), sums combined through LDS in
// slice order -> the segment's coefficient block (and its (s1, s2) sums for stage 3).

template <bool FWD>
__global__ __launch_bounds__(64 * FIN_W) void bn_seg_finalize_kernel(const float* __restrict__ partial, const double* __restrict__ sp,
                                                                     const SegTable st, int cpb, int C,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float eps, float* __restrict__ coef,
                                                                     double* __restrict__ segsum) {
    constexpr int NV = FWD ? 3 : 2;
    __shared__ double red[FIN_W][NV][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, s = blockIdx.y;
    const bool ok = c < C;
    int slot = 0;
    for (int t = 0; t < s; ++t) slot += seg_slices(st, t, cpb);
    const int ns = seg_slices(st, s, cpb);
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    for (int j = w; j < ns; j += FIN_W)
        if (ok) {
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += sp[((long long)(slot + j) * NV + v) * C + c];
        }
#pragma unroll
    for (int v = 0; v < NV; ++v) red[w][v][lane] = acc[v];
    __syncthreads();
    if (w == 0 && ok) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double t = 0.0;
            for (int k = 0; k < FIN_W; ++k) t += red[k][v][lane];
            acc[v] = t;
        }
        const double rows = (double)(st.row_off[s + 1] - st.row_off[s]);
        float* cf = coef + (long long)s * ST_ROWS * C;
        if (FWD) {
            const double dm = acc[0] / rows;
            const double mean = (double)partial[((long long)st.blk_off[s] * cpb * 2) * C + c] + dm;
            double var = (acc[2] + acc[1] - rows * dm * dm) / rows;
            var = var > 0.0 ? var : 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            cf[ST_MEAN * C + c] = (float)mean;
            cf[ST_VAR * C + c] = (float)var;
            cf[ST_INVSTD * C + c] = invstd;
            cf[ST_SCALE * C + c] = (gamma ? gamma[c] : 1.0f) * invstd;
            cf[ST_BETA * C + c] = beta ? beta[c] : 0.0f;
        } else {
            cf[ST_A * C + c] = (float)(acc[0] / rows);
            cf[ST_B * C + c] = (float)((double)cf[ST_INVSTD * C + c] * acc[1] / rows);
            segsum[((long long)s * 2 + 0) * C + c] = acc[0];
            segsum[((long long)s * 2 + 1) * C + c] = acc[1];
        }
    }
}

// Stage 3 (tiny): what is sequential over the segments.  One thread per channel walks the segments in order: one momentum
// update of the running statistics per segment (rounded to float each time, like nn.BatchNorm's buffer updates), or the
// fixed-order sums dgamma / dbeta.  A separate launch rather than a "last block" inside stage 2: the device-scope fences
// that pattern needs cost ~40 us each behind a GEMM (they write back / invalidate the L2s).
template <bool FWD>
__global__ __launch_bounds__(64) void bn_seg_ordered_kernel(const SegTable st, int C, const float* __restrict__ coef,
                                                            const double* __restrict__ segsum, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float momentum,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    constexpr int U = 8;   // segments fetched per round: the loads of a round are in flight together
    if (FWD) {
        float rm = running_mean[c], rv = running_var[c];
        for (int t0 = 0; t0 < st.nseg; t0 += U) {
            float m[U], v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u < st.nseg ? t0 + u : st.nseg - 1;
                const float* cf = coef + (long long)t * ST_ROWS * C;
                m[u] = cf[ST_MEAN * C + c];
                v[u] = cf[ST_VAR * C + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u >= st.nseg) break;
                const int rows = st.row_off[t0 + u + 1] - st.row_off[t0 + u];
                const double var = (double)v[u];
                const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
                rm = (float)((1.0 - momentum) * (double)rm + (double)momentum * (double)m[u]);
                rv = (float)((1.0 - momentum) * (double)rv + (double)momentum * unbiased);
            }
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    } else {
        double t1 = 0.0, t2 = 0.0;
        for (int t0 = 0; t0 < st.nseg; t0 += U) {
            double a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u < st.nseg ? t0 + u : st.nseg - 1;
                a[u] = segsum[((long long)t * 2 + 0) * C + c];
                b[u] = segsum[((long long)t * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (t0 + u < st.nseg) {
                    t1 += a[u];
                    t2 += b[u];
                }
        }
        if (dbeta) dbeta[c] += (float)t1;
        if (dgamma) dgamma[c] += (float)t2;
    }
}

// eval mode: coefficients from the running statistics
__global__ void bn_eval_coef_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                    float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
    coef[ST_MEAN * C + c] = running_mean[c];
    coef[ST_VAR * C + c] = running_var[c];
    coef[ST_INVSTD * C + c] = invstd;
    coef[ST_SCALE * C + c] = (gamma ? gamma[c] : 1.0f) * invstd;
    coef[ST_BETA * C + c] = beta ? beta[c] : 0.0f;
}

// z = relu((y - mean) * scale + beta), elementwise, 16 B per lane
// segmented variant: one block per APPLY_R-row block of one segment (its coefficient block)
constexpr int APPLY_R = 64;
__global__ __launch_bounds__(256) void bn_relu_apply_seg_kernel(const float* __restrict__ y, const SegTable st, int C,
                                                                const float* __restrict__ coef, int relu,
                                                                float* __restrict__ z, int y16) {
    const RowBlock rb = row_block(st, (int)blockIdx.x, APPLY_R);
    const int r1 = rb.row0 + APPLY_R < rb.row_end ? rb.row0 + APPLY_R : rb.row_end;
    const float* cf = coef + (long long)rb.seg * ST_ROWS * C;
    const long long e0 = (long long)rb.row0 * C / 4, e1 = (long long)r1 * C / 4;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
        const int c = (int)((e * 4) % C);
        const float4 v = ld_row4(y, e * 4, y16 != 0);
        const float4 mean = *(const float4*)(cf + ST_MEAN * C + c);
        const float4 sc = *(const float4*)(cf + ST_SCALE * C + c);
        const float4 bt = *(const float4*)(cf + ST_BETA * C + c);
        float4 o;
        o.x = __builtin_fmaf(v.x - mean.x, sc.x, bt.x);
        o.y = __builtin_fmaf(v.y - mean.y, sc.y, bt.y);
        o.z = __builtin_fmaf(v.z - mean.z, sc.z, bt.z);
        o.w = __builtin_fmaf(v.w - mean.w, sc.w, bt.w);
        if (relu) {
            o.x = fmaxf(o.x, 0.f);
            o.y = fmaxf(o.y, 0.f);
            o.z = fmaxf(o.z, 0.f);
            o.w = fmaxf(o.w, 0.f);
        }
        ((float4*)z)[e] = o;
    }
}

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ y, long long total4, int C,
                                                            const float* __restrict__ coef, int relu, float* __restrict__ z, int y16) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int c = (int)((e * 4) % C);
        const float4 v = ld_row4(y, e * 4, y16 != 0);
        const float4 mean = *(const float4*)(coef + ST_MEAN * C + c);
        const float4 sc = *(const float4*)(coef + ST_SCALE * C + c);
        const float4 bt = *(const float4*)(coef + ST_BETA * C + c);
        float4 o;
        o.x = __builtin_fmaf(v.x - mean.x, sc.x, bt.x);
        o.y = __builtin_fmaf(v.y - mean.y, sc.y, bt.y);
        o.z = __builtin_fmaf(v.z - mean.z, sc.z, bt.z);
        o.w = __builtin_fmaf(v.w - mean.w, sc.w, bt.w);
        if (relu) {
            o.x = fmaxf(o.x, 0.f);
            o.y = fmaxf(o.y, 0.f);
            o.z = fmaxf(o.z, 0.f);
            o.w = fmaxf(o.w, 0.f);
        }
        ((float4*)z)[e] = o;
    }
}

// out[g][c] = max_k relu(bn(y[g*K + k][c])), arg = first k attaining it (torch.max(dim) keeps the first maximum)
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ y, long long groups, int K, int C,
                                                              const float* __restrict__ coef, int relu,
                                                              float* __restrict__ out, int32_t* __restrict__ arg,
                                                              const SegTable st) {
    const long long total = groups * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long gi = e / C;
        const int c = (int)(e - gi * C);
        // a group never straddles a segment (segments are whole clouds)
        const float* cf = st.nseg > 1 ? coef + (long long)seg_of_row(st, (int)(gi * K)) * ST_ROWS * C : coef;
        const float mean = cf[ST_MEAN * C + c], sc = cf[ST_SCALE * C + c], bt = cf[ST_BETA * C + c];
        const float* p = y + gi * K * C + c;
        float best = -__builtin_inff();
        int bk = 0;
        for (int k0 = 0; k0 < K; k0 += 8) {   // eight independent loads in flight, then the (ordered) comparisons
            float raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = p[(long long)(k0 + u < K ? k0 + u : K - 1) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float v = __builtin_fmaf(raw[u] - mean, sc, bt);
                if (relu) v = fmaxf(v, 0.f);
                if (k0 + u < K && v > best) {
                    best = v;
                    bk = k0 + u;
                }
            }
        }
        out[e] = best;
        arg[e] = bk;
    }
}

// dz[g*K + k][c] = (k == arg[g][c]) ? dout[g][c] : 0
__global__ __launch_bounds__(256) void maxpool_scatter_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                              long long groups, int K, int C, float* __restrict__ dz) {
    const long long total = groups * K * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / C;
        const int c = (int)(e - r * C);
        const long long gi = r / K;
        const int k = (int)(r - gi * K);
        dz[e] = arg[gi * C + c] == k ? dout[gi * C + c] : 0.0f;
    }
}

// The same scatter for a pooled layer that ends in a BatchNorm, with that BatchNorm's backward column sums taken on the
// way: only the arg-max row of a group carries a gradient, so the sums need (dout, arg, Y at the arg-max rows) -- S x C
// elements instead of a second pass over the K times larger (dz, Y).  One workgroup per R rows (whole groups: K divides
// R <= PS_ROWS; the host picks R so that even the deep levels give every compute unit a block),
// partial[blk][0 / 1][c] like bn_bwd_reduce_kernel.
constexpr int PS_ROWS = 256;
__global__ __launch_bounds__(256) void maxpool_scatter_sums_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                                   int K, int C, int R, const float* __restrict__ y,
                                                                   const float* __restrict__ coef, int relu,
                                                                   float* __restrict__ dz, float* __restrict__ partial,
                                                                   const SegTable st, float* __restrict__ lead, long long ldlead,
                                                                   int nlead, unsigned char* __restrict__ arg8) {
    // arg8: the consumers rebuild dz themselves (TR_DYP operands, mlp_tile.h) -- no dense dz; the arg-max rows go out as bytes
    __shared__ float red[2][256];
    const RowBlock rb = row_block(st, (int)blockIdx.x, R);
    const int r0 = rb.row0, r1 = r0 + R < rb.row_end ? r0 + R : rb.row_end;
    // PN2_CHAIN_ZERO_LEAD: the block's rows of the chain's input gradient, columns [0, nlead)
    for (int e = threadIdx.x; e < (r1 - r0) * nlead; e += 256) lead[(long long)(r0 + e / nlead) * ldlead + e % nlead] = 0.0f;
    coef += (long long)rb.seg * ST_ROWS * C;
    const int g0 = r0 / K, g1 = r1 / K;   // segments and row blocks hold whole groups
    const int cw = C > 128 ? 256 : C > 64 ? 128 : 64;   // channels worked on in parallel
    const int ng = 256 / cw;              // groups worked on in parallel
    const int tc = threadIdx.x % cw, tg = threadIdx.x / cw;
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + tc;
        float s1 = 0.f, s2 = 0.f;
        if (c < C) {
            const float mean = coef[ST_MEAN * C + c], sc = coef[ST_SCALE * C + c], bt = coef[ST_BETA * C + c],
                        is = coef[ST_INVSTD * C + c];
            for (int gi = g0 + tg; gi < g1; gi += ng) {
                const int ka = arg[(long long)gi * C + c];
                const float g = dout[(long long)gi * C + c];
                const long long base = (long long)gi * K;
                if ((unsigned)ka < (unsigned)K) {
                    const float yy = y[(base + ka) * C + c];
                    const float t = __builtin_fmaf(yy - mean, sc, bt);
                    const float gz = (!relu || t > 0.f) ? g : 0.f;
                    s1 += gz;
                    s2 += gz * ((yy - mean) * is);
                }
                if (arg8) arg8[(long long)gi * C + c] = (unsigned char)ka;
                else
                    for (int k = 0; k < K; ++k) dz[(base + k) * C + c] = k == ka ? g : 0.0f;
            }
        }
        red[0][threadIdx.x] = s1;
        red[1][threadIdx.x] = s2;
        __syncthreads();
        if (tg == 0 && c < C) {
            for (int q = 1; q < ng; ++q) s1 += red[0][q * cw + tc], s2 += red[1][q * cw + tc];
            partial[((long long)blockIdx.x * 2 + 0) * C + c] = s1;
            partial[((long long)blockIdx.x * 2 + 1) * C + c] = s2;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void zero_lead_kernel(float* __restrict__ lead, long long ldlead, int nlead, long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
        lead[(e / nlead) * ldlead + e % nlead] = 0.0f;
}

// ------------------------------------------------------------------------------------------ BatchNorm: backward
// partial[blk][0][c] = sum_r dzhat, partial[blk][1][c] = sum_r dzhat * xhat over the block's RB rows.
// Each thread owns 4 consecutive channels (16-byte loads of dz and y), row groups are combined through LDS.
constexpr int RB = 64;
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, long long lddz,
                                                            const float* __restrict__ y, long long ldy, int rows, int C,
                                                            const float* __restrict__ coef, int relu,
                                                            float* __restrict__ partial, const SegTable st, int dz16, int y16) {
    __shared__ float red[2][256][4];
    const RowBlock rb = row_block(st, (int)blockIdx.x, RB);
    const int r0 = rb.row0;
    const int r1 = r0 + RB < rb.row_end ? r0 + RB : rb.row_end;
    coef += (long long)rb.seg * ST_ROWS * C;
    const int C4 = (C + 3) / 4;
    const int cw = C4 < 256 ? (C4 <= 8 ? 8 : C4 <= 16 ? 16 : C4 <= 32 ? 32 : C4 <= 64 ? 64 : C4 <= 128 ? 128 : 256) : 256;
    const int rg = 256 / cw;                       // row groups working in parallel
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    for (int q0 = 0; q0 < C4; q0 += cw) {
        const int c = 4 * (q0 + tc);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            float mean[4], sc[4], bt[4], is[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = c + j < C ? c + j : C - 1;
                mean[j] = coef[ST_MEAN * C + cc];
                sc[j] = coef[ST_SCALE * C + cc];
                bt[j] = coef[ST_BETA * C + cc];
                is[j] = coef[ST_INVSTD * C + cc];
            }
            for (int rr = r0 + tr; rr < r1; rr += 4 * rg) {   // four rows' loads in flight, summed in row order
                float4 yv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = rr + u * rg < r1 ? rr + u * rg : r1 - 1;
                    yv[u] = y16 ? ld4h<VEC>(y, ldy, r, c, rows, C) : ld4<VEC>(y, ldy, r, c, rows, C);      // (bfloat16 storage: uniform)
                    dv[u] = dz16 ? ld4h<VEC>(dz, lddz, r, c, rows, C) : ld4<VEC>(dz, lddz, r, c, rows, C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rr + u * rg >= r1) break;
                    const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float t = __builtin_fmaf(yy[j] - mean[j], sc[j], bt[j]);
                        const float g = (!relu || t > 0.f) ? dd[j] : 0.f;
                        s1[j] += g;
                        s2[j] += g * ((yy[j] - mean[j]) * is[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][threadIdx.x][j] = s1[j];
            red[1][threadIdx.x][j] = s2[j];
        }
        __syncthreads();
        if (tr == 0 && c < C) {
            for (int k = 1; k < rg; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1[j] += red[0][k * cw + tc][j];
                    s2[j] += red[1][k * cw + tc][j];
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < C) {
                    partial[((long long)blockIdx.x * 2 + 0) * C + c + j] = s1[j];
                    partial[((long long)blockIdx.x * 2 + 1) * C + c + j] = s2[j];
                }
        }
        __syncthreads();
    }
}

// dW[m][n] (+)= sum over splits of slab[s][m][n], fixed order; optional column sums for a bias without BatchNorm
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long long mn,
                                                          float* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < mn; e += (long long)gridDim.x * 256) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= nsplit; k += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += slab[(long long)(k + u) * mn + e];
        for (; k < nsplit; ++k) a[0] += slab[(long long)k * mn + e];
        out[e] += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}

// Many splits: the four wavefronts of a block each sum a quarter of the splits for the same 64 elements (256-byte
// coalesced reads per slab), then add up in wavefront order -> mn/64 blocks instead of mn/256, 4x shorter chains.
__global__ __launch_bounds__(256) void slab_reduce_sliced_kernel(const float* __restrict__ slab, int nsplit, long long mn,
                                                                 float* __restrict__ out) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    const int per = (nsplit + 3) / 4;
    const int k0 = w * per, k1 = (k0 + per) < nsplit ? (k0 + per) : nsplit;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < mn) {
        int k = k0;
        for (; k + 8 <= k1; k += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += slab[(long long)(k + u) * mn + e];
        for (; k < k1; ++k) a[0] += slab[(long long)k * mn + e];
    }
    part[w][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (w == 0 && e < mn) out[e] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// All weight-gradient slab reductions of one chain call in ONE launch (a launch per layer is 5-8 us of mostly latency, 23
// of them per backward pass): block b serves 64 elements of the task whose block range holds b, exactly like
// slab_reduce_sliced_kernel (same summation order, so the result does not depend on how the work is batched).
constexpr int SLAB_TASKS = 32;
struct SlabTask {
    const float* slab;
    float* out;
    long long mn;
    int nsplit, first_block;
};
struct SlabTasks {
    int n, blocks;
    SlabTask t[SLAB_TASKS];
};
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const SlabTasks T) {
    __shared__ float part[4][64];
    int k = 0;
#pragma unroll
    for (int i = 1; i < SLAB_TASKS; ++i)
        if (i < T.n && (int)blockIdx.x >= T.t[i].first_block) k = i;
    const float* __restrict__ slab = T.t[k].slab;
    const long long mn = T.t[k].mn;
    const int nsplit = T.t[k].nsplit;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)((int)blockIdx.x - T.t[k].first_block) * 64 + lane;
    const int per = (nsplit + 3) / 4;
    const int k0 = w * per, k1 = (k0 + per) < nsplit ? (k0 + per) : nsplit;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < mn) {
        int q = k0;
        for (; q + 8 <= k1; q += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += slab[(long long)(q + u) * mn + e];
        for (; q < k1; ++q) a[0] += slab[(long long)q * mn + e];
    }
    part[w][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (w == 0 && e < mn) T.t[k].out[e] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// partial[blk][c] = sum over the block's rows of dz[r][c]  (bias of a layer WITHOUT BatchNorm, e.g. the last
// conv of a head); summed in fixed order by colsum_finalize_kernel -> deterministic
constexpr int CS_ROWS = 256;   // (1024: 179 workgroups for a 183 000-row layer -- fewer than compute units; 0.43 -> 0.2 ms at 768 channels)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dz, long long ld, int rows, int C,
                                                     float* __restrict__ partial) {
    // cw consecutive lanes walk cw consecutive channels of a row (coalesced), 256 / cw rows in parallel; the row groups are
    // joined through LDS in fixed order.  (The first version walked one channel at a time with a row per thread: one pass over
    // the tensor's cache lines PER CHANNEL -- fine for a 3-wide head, 1.6 ms per call on a 1 M x 96 PointTransformerV3 layer.)
    __shared__ float red[256];
    const int r0 = blockIdx.x * CS_ROWS;
    const int r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    int cw = 1;
    while (cw < C && cw < 64) cw <<= 1;
    const int nr = 256 / cw, cx = (int)threadIdx.x % cw, ry = (int)threadIdx.x / cw;
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + cx;
        float a = 0.f;
        if (c < C) {
            float a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four independent loads in flight
            int r = r0 + ry;
            for (; r + 3 * nr < r1; r += 4 * nr) {
                a += dz[(long long)r * ld + c];
                a1 += dz[(long long)(r + nr) * ld + c];
                a2 += dz[(long long)(r + 2 * nr) * ld + c];
                a3 += dz[(long long)(r + 3 * nr) * ld + c];
            }
            for (; r < r1; r += nr) a += dz[(long long)r * ld + c];
            a = (a + a1) + (a2 + a3);
        }
        red[threadIdx.x] = a;
        __syncthreads();
        for (int sft = nr >> 1; sft > 0; sft >>= 1) {
            if (ry < sft) red[threadIdx.x] += red[threadIdx.x + sft * cw];
            __syncthreads();
        }
        if (ry == 0 && c < C) partial[(long long)blockIdx.x * C + c] = red[cx];
        __syncthreads();
    }
}

// one 64-thread block per channel: lanes split the partial blocks, DPP-free shuffle tree, float64 accumulation
__global__ __launch_bounds__(64) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                             float* __restrict__ out) {
    const int c = blockIdx.x;
    double a = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) a += (double)partial[(long long)k * C + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (threadIdx.x == 0) out[c] += (float)a;
}

// ------------------------------------------------------------------------------------- narrow last layer (heads)
// The 128 -> 2 / 128 -> 3 convs that end the two heads are matrix-vector shaped: on the 128-wide MFMA tile they cost
// as much matrix-pipe time as a full layer for 2 % of its useful work.  They run on the vector ALU instead, one
// pass over the rows each way: a row's K channels are spread over KL = K/4 lanes (16-byte loads), the N <= 4
// outputs are reduced across those lanes with DPP row operations.
//   forward : out[r][n] = b[n] + sum_k act(Y[r][k]) W[n][k]
//   backward: dX[r][k] = sum_n dO[r][n] W[n][k]  (written);  per-block partials of dW[n][k] = sum_r dO[r][n] act(Y[r][k]),
//             db[n] = sum_r dO[r][n] and, when the input comes through a BatchNorm, that layer's backward sums
//             s1[k] = sum_r mask*dX, s2[k] = sum_r mask*dX*xhat -- everything the generic path needs three GEMM
//             launches and a reduction pass for.
constexpr int NR_ROWS = 1024;  // rows per block
constexpr int NR_T = 1024;     // threads per block: 16 waves per CU hide the load latency, 256 blocks keep the partials few
constexpr int NMAX = 4;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// sum over each aligned group of KL lanes (KL = 16, 32 or 64); valid in the LAST lane of the group
template <int KL>
__device__ __forceinline__ float group_sum(float v) {
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);                      // lane 15 of every 16-lane row
    if (KL >= 32) v = dpp_add<0x142, 0xA>(v);        // row_bcast:15 -> lanes 31 and 63
    if (KL >= 64) v = dpp_add<0x143, 0xC>(v);        // row_bcast:31 -> lane 63
    return v;
}

template <int KL, bool HAS_BN, bool Y16 = false>
__global__ __launch_bounds__(NR_T) void narrow_fwd_kernel(const float* __restrict__ y, long long ldy, int rows, int K,
                                                         const float* __restrict__ coef, int relu,
                                                         const float* __restrict__ W, const float* __restrict__ bias, int N,
                                                         float* __restrict__ out, const SegTable st) {
    constexpr int RPW = 64 / KL;                      // rows per wave per step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane % KL, rsub = lane / KL, k = 4 * kq;
    const RowBlock rb = row_block(st, (int)blockIdx.x, NR_ROWS);
    if (HAS_BN) coef += (long long)rb.seg * ST_ROWS * K;
    float w[NMAX][4], cm[4], cs[4], cb[4];
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[n][j] = n < N ? W[(long long)n * K + k + j] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cm[j] = HAS_BN ? coef[ST_MEAN * K + k + j] : 0.0f;
        cs[j] = HAS_BN ? coef[ST_SCALE * K + k + j] : 1.0f;
        cb[j] = HAS_BN ? coef[ST_BETA * K + k + j] : 0.0f;
    }
    const int r0 = rb.row0;
    const int r1 = r0 + NR_ROWS < rb.row_end ? r0 + NR_ROWS : rb.row_end;
    constexpr int RGF = (NR_T / 64) * RPW;
    const int iters = (r1 - r0 + RGF - 1) / RGF;   // uniform trip count: the DPP sums need every lane
    constexpr int UF = 4;   // rows in flight per lane: the loop is one 16-byte load and a DPP sum per row otherwise
    for (int it0 = 0; it0 < iters; it0 += UF) {
        float4 v[UF];
        int rbu[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            rbu[u] = r0 + (it0 + u) * RGF + wave * RPW + rsub;
            const int r = rbu[u] < r1 ? rbu[u] : r1 - 1;
            v[u] = ld_row4(y, (long long)r * ldy + k, Y16);
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {   // steps beyond iters compute on the last row and store nothing
            const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            float p[NMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = e[j];
                if (HAS_BN) {
                    x = __builtin_fmaf(x - cm[j], cs[j], cb[j]);
                    if (relu) x = fmaxf(x, 0.0f);
                }
#pragma unroll
                for (int n = 0; n < NMAX; ++n) p[n] = __builtin_fmaf(x, w[n][j], p[n]);
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) p[n] = group_sum<KL>(p[n]);
            if (kq == KL - 1 && rbu[u] < r1) {
                for (int n = 0; n < N; ++n) out[(long long)rbu[u] * N + n] = p[n] + (bias ? bias[n] : 0.0f);
            }
        }
    }
}

template <int KL, bool HAS_BN, bool Y16 = false, bool DX16 = false>
__global__ __launch_bounds__(NR_T) void narrow_bwd_kernel(const float* __restrict__ dout, int rows, int K, int N,
                                                         const float* __restrict__ y, long long ldy,
                                                         const float* __restrict__ coef, int relu,
                                                         const float* __restrict__ W, float* __restrict__ dx,
                                                         float* __restrict__ part_s, float* __restrict__ part_w,
                                                         float* __restrict__ part_b, const SegTable st) {
    constexpr int RPW = 64 / KL, RG = (NR_T / 64) * RPW;       // row groups per block
    __shared__ float red[RG][KL][4 * NMAX + 8];
    __shared__ float redb[4][NMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane % KL, rsub = lane / KL, k = 4 * kq, rgp = wave * RPW + rsub;
    const RowBlock rb = row_block(st, (int)blockIdx.x, NR_ROWS);
    if (HAS_BN) coef += (long long)rb.seg * ST_ROWS * K;
    float w[NMAX][4], cm[4], cs[4], cb[4], ci[4];
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[n][j] = n < N ? W[(long long)n * K + k + j] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cm[j] = HAS_BN ? coef[ST_MEAN * K + k + j] : 0.0f;
        cs[j] = HAS_BN ? coef[ST_SCALE * K + k + j] : 1.0f;
        cb[j] = HAS_BN ? coef[ST_BETA * K + k + j] : 0.0f;
        ci[j] = HAS_BN ? coef[ST_INVSTD * K + k + j] : 0.0f;
    }
    float dw[NMAX][4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, db[NMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) dw[n][j] = 0.0f;
    const int r0 = rb.row0;
    const int r1 = r0 + NR_ROWS < rb.row_end ? r0 + NR_ROWS : rb.row_end;
    for (int r = r0 + rgp; r < r1; r += RG) {
        const float4 v = ld_row4(y, (long long)r * ldy + k, Y16);
        const float e[4] = {v.x, v.y, v.z, v.w};
        float g[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) g[n] = n < N ? dout[(long long)r * N + n] : 0.0f;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = e[j], t = 1.0f;
            if (HAS_BN) {
                t = __builtin_fmaf(x - cm[j], cs[j], cb[j]);
                x = relu ? fmaxf(t, 0.0f) : t;
            }
            float d = 0.0f;
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                d = __builtin_fmaf(g[n], w[n][j], d);
                dw[n][j] = __builtin_fmaf(g[n], x, dw[n][j]);
            }
            o[j] = d;
            if (HAS_BN) {
                const float dzh = (!relu || t > 0.0f) ? d : 0.0f;
                s1[j] += dzh;
                s2[j] += dzh * ((e[j] - cm[j]) * ci[j]);
            }
        }
        st_row4(dx, (long long)r * K + k, make_float4(o[0], o[1], o[2], o[3]), DX16);
        if (kq == 0)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) db[n] += g[n];
    }
    // block reduction over the RG row groups, fixed order
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rgp][kq][4 * n + j] = dw[n][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[rgp][kq][4 * NMAX + j] = s1[j];
        red[rgp][kq][4 * NMAX + 4 + j] = s2[j];
    }
    __syncthreads();
    if (rgp == 0) {
        float acc[4 * NMAX + 8];
#pragma unroll
        for (int q = 0; q < 4 * NMAX + 8; ++q) acc[q] = red[0][kq][q];
        for (int gI = 1; gI < RG; ++gI)
#pragma unroll
            for (int q = 0; q < 4 * NMAX + 8; ++q) acc[q] += red[gI][kq][q];
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) part_w[((long long)blockIdx.x * N + n) * K + k + j] = acc[4 * n + j];
        if (HAS_BN)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part_s[((long long)blockIdx.x * 2 + 0) * K + k + j] = acc[4 * NMAX + j];
                part_s[((long long)blockIdx.x * 2 + 1) * K + k + j] = acc[4 * NMAX + 4 + j];
            }
    }
    // bias column sums: lanes with kq == 0 hold them, one per row group
    __syncthreads();
    float* rbias = &red[0][0][0];
    if (kq == 0)
#pragma unroll
        for (int n = 0; n < NMAX; ++n) rbias[rgp * NMAX + n] = db[n];
    __syncthreads();
    if (threadIdx.x < (unsigned)N) {
        float a = 0.0f;
        for (int gI = 0; gI < RG; ++gI) a += rbias[gI * NMAX + threadIdx.x];
        part_b[(long long)blockIdx.x * N + threadIdx.x] = a;
    }
    (void)redb;
}

inline bool narrow_ok(const pn2_mlp_layer& L, bool last, int pool_k) {
    return last && !L.has_bn && pool_k <= 1 && L.cout <= NMAX && (L.cin == 64 || L.cin == 128 || L.cin == 256);
}

inline unsigned grid1d(long long total, int per_block = 256) {
    long long g = (total + per_block - 1) / per_block;
    return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

// Host view of the row segments of a chain (pn2_segments of the C ABI; one segment = the ordinary batch).
struct Segs {
    int nseg;
    const int32_t* row_off;   // host, nseg + 1 entries, row_off[0] == 0, row_off[nseg] == rows
    int one[2];               // storage for the single-segment case
};
Segs make_segs(int rows, const pn2_segments* sg) {
    Segs S{};
    if (sg && sg->nseg > 1) {
        S.nseg = sg->nseg;
        S.row_off = sg->row_off;
    } else {
        S.nseg = 1;
        S.one[0] = 0;
        S.one[1] = rows;
        S.row_off = nullptr;   // patched by the accessor below (S is returned by value)
    }
    return S;
}
inline int seg_off(const Segs& S, int i) { return S.row_off ? S.row_off[i] : S.one[i]; }
bool segs_valid(int rows, const pn2_segments* sg, int pool_k) {
    if (!sg || sg->nseg <= 1) return true;
    if (sg->nseg > kMaxSegs || !sg->row_off || sg->row_off[0] != 0 || sg->row_off[sg->nseg] != rows) return false;
    for (int i = 0; i < sg->nseg; ++i) {
        const int n = sg->row_off[i + 1] - sg->row_off[i];
        if (n <= 0 || (pool_k > 1 && n % pool_k)) return false;
    }
    return true;
}
// blocks of R rows that never straddle a segment; returns the table, *nblk = number of blocks
SegTable make_table(const Segs& S, int R, int* nblk) {
    SegTable t;
    t.nseg = S.nseg;
    int b = 0;
    for (int i = 0; i < S.nseg; ++i) {
        t.row_off[i] = seg_off(S, i);
        t.blk_off[i] = b;
        b += pn2::ceil_div(seg_off(S, i + 1) - seg_off(S, i), R);
    }
    t.row_off[S.nseg] = seg_off(S, S.nseg);
    t.blk_off[S.nseg] = b;
    *nblk = b;
    return t;
}

inline void launch_slab_reduce(const float* slab, int nsplit, long long mn, float* out, hipStream_t s) {
    if (nsplit >= 32)
        PN2_LAUNCH("slab_reduce", 4.0 * (nsplit + 2) * mn, 0, slab_reduce_sliced_kernel, dim3((unsigned)((mn + 63) / 64)),
                   dim3(256), s, slab, nsplit, mn, out);
    else
        PN2_LAUNCH("slab_reduce", 4.0 * (nsplit + 2) * mn, 0, slab_reduce_kernel, dim3(grid1d(mn)), dim3(256), s, slab, nsplit,
                   mn, out);
}

inline void add_slab_task(SlabTasks& T, const float* slab, int nsplit, long long mn, float* out) {
    SlabTask& t = T.t[T.n++];
    t.slab = slab, t.out = out, t.mn = mn, t.nsplit = nsplit, t.first_block = T.blocks;
    T.blocks += (int)((mn + 63) / 64);
}
inline void flush_slab_tasks(SlabTasks& T, hipStream_t s) {
    if (!T.n) return;
    double bytes = 0.0;
    for (int i = 0; i < T.n; ++i) bytes += 4.0 * (T.t[i].nsplit + 2) * T.t[i].mn;
    PN2_LAUNCH("slab_reduce", bytes, 0, slab_reduce_multi_kernel, dim3(T.blocks), dim3(256), s, T);
    T.n = T.blocks = 0;
}

// Tile choice: 128-tiles unless they would leave most of the chip idle (deep levels have a few hundred rows).
inline int pick_tile(int M, int N, int nsplit) {
    // (128-row tiles for narrow outputs of long matrices were measured: 20.8 -> 21.3 .. 22.5 ms per raster-mode tree)
    if (M <= 64 || N <= 64) return 64;   // a narrow output (the 128 -> 2/3 head convs) wastes less of a 64-tile
    const long long big = (long long)pn2::ceil_div(M, 128) * pn2::ceil_div(N, 128) * nsplit;
    return big >= 96 ? 128 : 64;
}

// Row-tiled roles (forward, dgrad): grid.x = row tiles of the segments; wgrad: grid.z = reduction ranges of the segments
// (`g.k_per_split` rows each), grid.x tiles the output rows (= cout).
template <bool A_T, int A_KIND, bool B_T, int B_KIND, int EPI, int TILE, bool VEC, int TEAMS = 1, bool BF16 = false, int NTT = NT, int ST = 0,
          bool DUAL = false>
int launch_gemm_tv(GemmArgs& g, const Segs& S, hipStream_t s, int* nblk_out) {
    int nblk = 0;
    const SegTable st = make_table(S, EPI == EPI_SLAB ? g.k_per_split : TILE, &nblk);
    const dim3 grid = EPI == EPI_SLAB ? dim3(pn2::ceil_div(g.M, TILE), pn2::ceil_div(g.N, TILE), nblk)
                                      : dim3(nblk, pn2::ceil_div(g.N, TILE), 1);
    if (nblk_out) *nblk_out = nblk;
    // (the dgrad that closes a pair of linked chains also reads C -- its accumulators start from the sibling's gradient --
    // and the producing layer's rows for its BatchNorm-backward sums: its own group in the profile, with those bytes)
    const bool closing = EPI == EPI_STORE && g.accumulate;
    const char* name = DUAL ? (BF16 ? "gemm_dgrad_pair_bf16" : "gemm_dgrad_pair")
                       : BF16 ? (EPI == EPI_FWD ? "gemm_fwd_bf16" : EPI == EPI_STORE ? (closing ? "gemm_dgrad_acc_bf16" : "gemm_dgrad_bf16")
                                                                                     : "gemm_wgrad_bf16")
                            : (EPI == EPI_FWD ? "gemm_fwd" : EPI == EPI_STORE ? (closing ? "gemm_dgrad_acc" : "gemm_dgrad") : "gemm_wgrad");
    const double mk = (double)g.M * g.K * (A_KIND == TR_DY ? 2 : 1), kn = (double)g.K * g.N * (B_KIND == TR_DY ? 2 : 1);
    // algorithmic bytes with the rows' storage type (ST: bfloat16 A / B / C / ey rows are 2 bytes per element; slabs stay fp32)
    const double ea = (ST & 1) ? 2.0 : 4.0, eb = (ST & 2) ? 2.0 : 4.0, ec = ((ST & 4) && EPI != EPI_SLAB) ? 2.0 : 4.0, ee = (ST & 8) ? 2.0 : 4.0;
    const double mn = (double)g.M * g.N;
    const double bytes = ea * mk + eb * kn + ec * mn * ((EPI == EPI_SLAB ? nblk : 1) + (closing ? 1.0 : 0.0)) +
                         ((EPI == EPI_STORE && g.partial) ? ee * mn : 0.0);   // (closing: C is read too; partial: the ey rows)
    if (closing && !DUAL)
        PN2_LAUNCH(name, bytes, 2.0 * g.M * g.N * g.K,
                   (gemm_kernel<A_T, A_KIND, B_T, B_KIND, EPI, TILE, VEC, TEAMS, BF16, EPI == EPI_STORE && !DUAL, NTT, ST>), grid,
                   dim3(NTT * TEAMS), s, g, st);
    else
        PN2_LAUNCH(name, bytes, 2.0 * g.M * g.N * g.K, (gemm_kernel<A_T, A_KIND, B_T, B_KIND, EPI, TILE, VEC, TEAMS, BF16, false, NTT, ST, DUAL>),
                   grid, dim3(NTT * TEAMS), s, g, st);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

inline long long grid_blocks(int M, int N, int tile) { return (long long)pn2::ceil_div(M, tile) * pn2::ceil_div(N, tile); }

template <bool A_T, int A_KIND, bool B_T, int B_KIND, int EPI>
int launch_gemm(GemmArgs& g, const Segs& S, int tile, hipStream_t s, int* nblk_out = nullptr) {
    const bool vec = vec_ok(g.A) && vec_ok(g.B);
    const int st16 = (g.A.p16 ? 1 : 0) | (g.B.p16 ? 2 : 0) | (g.c16 ? 4 : 0) | (g.ey16 ? 8 : 0);
    if (st16 && !(tile == 128 && g.precision == PN2_PRECISION_BF16 && vec)) return PN2_E_BADARG;   // bfloat16 rows: these kernels only
    if (st16) {
        // the storage layouts the chains produce (pn2_hip.h PN2_CHAIN_STORE_BF16), each a kernel of its own: the row type is a
        // compile-time property (a run-time branch between a K-tile's loads serialises their round trips)
        if (g.A.q && g.A.p16 != g.A.q16) return PN2_E_BADARG;
#define PN2_ST_CASE(V) case V: return launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 128, true, 1, true, NT, V>(g, S, s, nblk_out)
        if constexpr (EPI == EPI_FWD) {
            switch (st16) { PN2_ST_CASE(4); PN2_ST_CASE(5); default: return PN2_E_BADARG; }
        } else if constexpr (EPI == EPI_STORE) {
            if (g.partial && !g.ey16) return PN2_E_BADARG;   // (a bfloat16 chain is linked to bfloat16 rows only)
            switch (st16 | 8) { PN2_ST_CASE(9); PN2_ST_CASE(13); default: return PN2_E_BADARG; }
        } else {
            switch (st16) { PN2_ST_CASE(1); PN2_ST_CASE(3); default: return PN2_E_BADARG; }
        }
#undef PN2_ST_CASE
    }
    if (tile == 128 && g.precision == PN2_PRECISION_BF16 && vec)   // the small-problem tiles always run fp32
        return launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 128, true, 1, true>(g, S, s, nblk_out);
    if (tile == 128)
        return vec ? launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 128, true>(g, S, s, nblk_out)
                   : launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 128, false>(g, S, s, nblk_out);
    // The smallest problems (a deep level: <= 128 tiles of 64 x 64): 32 x 32 tiles whose four K-teams are ONE wavefront each -- four
    // times the workgroups, i.e. four times the compute units whose matrix cores take part (chunks of statistics stay 32 rows, so
    // the callers' 64-tile bookkeeping -- two chunks per 64 rows -- still describes the partials).
    if (EPI != EPI_SLAB && S.nseg == 1 && !g.accumulate && g.K >= 8 * BK && (long long)grid_blocks(g.M, g.N, 64) <= 128 &&
        !getenv("PN2_NO_TILE32"))
        return vec ? launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 32, true, 4, false, 64>(g, S, s, nblk_out)
                   : launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 32, false, 4, false, 64>(g, S, s, nblk_out);
    // small problem with a long contraction: split K over four teams inside the workgroup
    if (EPI != EPI_SLAB && g.K >= 8 * BK && (long long)grid_blocks(g.M, g.N, 64) <= 512)
        return vec ? launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 64, true, 4>(g, S, s, nblk_out)
                   : launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 64, false, 4>(g, S, s, nblk_out);
    return vec ? launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 64, true>(g, S, s, nblk_out)
               : launch_gemm_tv<A_T, A_KIND, B_T, B_KIND, EPI, 64, false>(g, S, s, nblk_out);
}

// The pooled layer's weight / input gradient with the max-pool's gradient rebuilt while staging (TR_DYP): fp32 rows, 16-byte
// staging, the 128- and the plain 64-tile kernels only -- pooled_dyp_ok() below is the same test, made before the scatter
template <bool A_T, bool B_T, int B_KIND, int EPI>
int launch_gemm_dyp(GemmArgs& g, const Segs& S, int tile, hipStream_t s, int* nblk_out = nullptr) {
    if (!(vec_ok(g.A) && vec_ok(g.B)) || g.A.p16 || g.B.p16 || g.c16 || g.ey16) return PN2_E_BADARG;
    if (tile == 128 && g.precision == PN2_PRECISION_BF16)
        return launch_gemm_tv<A_T, TR_DYP, B_T, B_KIND, EPI, 128, true, 1, true>(g, S, s, nblk_out);
    if (tile == 128) return launch_gemm_tv<A_T, TR_DYP, B_T, B_KIND, EPI, 128, true>(g, S, s, nblk_out);
    return launch_gemm_tv<A_T, TR_DYP, B_T, B_KIND, EPI, 64, true>(g, S, s, nblk_out);
}
// would launch_gemm pick one of those kernels for this problem?  (the small-problem variants -- 32-tiles, K teams -- keep dz)
inline bool dyp_tile_ok(int tile, int M, int N, int K, int epi, int nseg) {
    if (tile == 128) return true;
    if (epi != EPI_SLAB && K >= 8 * BK && (long long)grid_blocks(M, N, 64) <= 512) return false;
    (void)nseg;
    return true;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <int KL>
int launch_narrow_fwd_kl(const Act& in, int rows, const pn2_mlp_layer& L, float* out, const Segs& S, hipStream_t s) {
    int nblk = 0;
    const SegTable st = make_table(S, NR_ROWS, &nblk);
    const dim3 grid(nblk), block(NR_T);
    const double bytes = 4.0 * rows * (L.cin + L.cout), flops = 2.0 * rows * L.cin * L.cout;
    if (in.coef && in.h16)
        PN2_LAUNCH("narrow_fwd", bytes * 0.5, flops, (narrow_fwd_kernel<KL, true, true>), grid, block, s, in.p, in.ld, rows, L.cin, in.coef,
                   in.relu, L.weight, L.bias, L.cout, out, st);
    else if (in.h16)
        return PN2_E_BADARG;
    else if (in.coef)
        PN2_LAUNCH("narrow_fwd", bytes, flops, (narrow_fwd_kernel<KL, true>), grid, block, s, in.p, in.ld, rows, L.cin, in.coef,
                   in.relu, L.weight, L.bias, L.cout, out, st);
    else
        PN2_LAUNCH("narrow_fwd", bytes, flops, (narrow_fwd_kernel<KL, false>), grid, block, s, in.p, in.ld, rows, L.cin, in.coef,
                   in.relu, L.weight, L.bias, L.cout, out, st);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int launch_narrow_fwd(const Act& in, int rows, const pn2_mlp_layer& L, float* out, const Segs& S, hipStream_t s) {
    return L.cin == 64 ? launch_narrow_fwd_kl<16>(in, rows, L, out, S, s)
                       : L.cin == 128 ? launch_narrow_fwd_kl<32>(in, rows, L, out, S, s) : launch_narrow_fwd_kl<64>(in, rows, L, out, S, s);
}

template <int KL>
int launch_narrow_bwd_kl(const Act& in, int rows, const pn2_mlp_layer& L, const float* dout, float* dx, float* part_s,
                         float* part_w, float* part_b, const SegTable& st, int nblk, hipStream_t s, int dx16) {
    const dim3 grid(nblk), block(NR_T);
    const double bytes = 4.0 * rows * (2.0 * L.cin + L.cout), flops = 4.0 * rows * L.cin * L.cout;
    if (in.h16 != dx16 || (in.h16 && !in.coef)) return PN2_E_BADARG;   // bfloat16 storage: rows in and gradient rows out together
    if (in.h16)
        PN2_LAUNCH("narrow_bwd", bytes * 0.5, flops, (narrow_bwd_kernel<KL, true, true, true>), grid, block, s, dout, rows, L.cin, L.cout,
                   in.p, in.ld, in.coef, in.relu, L.weight, dx, part_s, part_w, part_b, st);
    else if (in.coef)
        PN2_LAUNCH("narrow_bwd", bytes, flops, (narrow_bwd_kernel<KL, true>), grid, block, s, dout, rows, L.cin, L.cout, in.p, in.ld,
                   in.coef, in.relu, L.weight, dx, part_s, part_w, part_b, st);
    else
        PN2_LAUNCH("narrow_bwd", bytes, flops, (narrow_bwd_kernel<KL, false>), grid, block, s, dout, rows, L.cin, L.cout, in.p, in.ld,
                   in.coef, in.relu, L.weight, dx, part_s, part_w, part_b, st);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int launch_narrow_bwd(const Act& in, int rows, const pn2_mlp_layer& L, const float* dout, float* dx, float* part_s,
                      float* part_w, float* part_b, const SegTable& st, int nblk, hipStream_t s, int dx16) {
    return L.cin == 64 ? launch_narrow_bwd_kl<16>(in, rows, L, dout, dx, part_s, part_w, part_b, st, nblk, s, dx16)
           : L.cin == 128 ? launch_narrow_bwd_kl<32>(in, rows, L, dout, dx, part_s, part_w, part_b, st, nblk, s, dx16)
                          : launch_narrow_bwd_kl<64>(in, rows, L, dout, dx, part_s, part_w, part_b, st, nblk, s, dx16);
}

struct WgradPlan {
    int tile, nsplit, kps;
};

// dW = dY^T X reduces over `rows`; the reduction is cut into row ranges of kps rows that stay inside one segment (one
// slab each, summed in fixed order); nsplit = upper bound of the number of ranges (exact for one segment)
WgradPlan plan_wgrad(int rows, int cout, int cin, int nseg) {
    const int tiles = pn2::ceil_div(cout, 128) * pn2::ceil_div(cin, 128);
    // one slab per workgroup; long reductions get two workgroups per CU (measured at 262144 rows, 128x128: 136 us with
    // 256 workgroups = one wavefront per SIMD, 111 us with 512, +3.6 us of slab reduction)
    int target = rows >= 65536 ? 512 : 256;
    if (const char* e = getenv("PN2_WGRAD_BLOCKS")) target = atoi(e) > 0 ? atoi(e) : target;  // tuning aid
    int nsplit = target / tiles;
    if (nsplit < 1) nsplit = 1;
    const int kps = pn2::ceil_div(pn2::ceil_div(rows, nsplit), BK) * BK;
    nsplit = pn2::ceil_div(rows, kps) + (nseg > 1 ? nseg : 0);
    return WgradPlan{pick_tile(cout, cin, nsplit), nsplit, kps};
}

// Segmented finalizes keep their float64 scratch at the start of the workspace, ahead of everything else:
// [256 B spare][per-segment (s1, s2) sums: nseg x 2 x cmax][slice sums: slices x 3 x cmax]
struct FinScratch {
    double* segsum;
    double* slices;
};
size_t slice_region_bytes(int rows, int nseg, size_t cmax) {
    if (nseg <= 1) return 0;                                  // one segment: the single-kernel finalizes need none
    const size_t chunks = (size_t)pn2::ceil_div(rows, 32) + 2 * (size_t)nseg;
    const size_t slices = (chunks + SLICE - 1) / SLICE + (size_t)nseg;
    return align256(256 + ((size_t)nseg * 2 + slices * 3) * cmax * sizeof(double));
}
size_t chain_cmax(const pn2_mlp_layer* layers, int nlayers) {
    size_t cmax = 0;
    for (int i = 0; i < nlayers; ++i) {
        const size_t m = layers[i].cin > layers[i].cout ? layers[i].cin : layers[i].cout;
        cmax = cmax > m ? cmax : m;
    }
    return cmax;
}
size_t slice_region_of(int rows, const pn2_mlp_layer* layers, int nlayers, int nseg) {
    return slice_region_bytes(rows, nseg, chain_cmax(layers, nlayers));
}
FinScratch fin_scratch(void* base, int nseg, size_t cmax) {
    FinScratch f;
    f.segsum = (double*)((char*)base + 256);
    f.slices = f.segsum + (size_t)nseg * 2 * cmax;
    return f;
}

// grid of stage 1: (channel groups of 64, slices of the longest segment, segments)
dim3 slice_grid(const SegTable& st, int cpb, int C) {
    int longest = 0;
    for (int s = 0; s < st.nseg; ++s) {
        const int n = (st.blk_off[s + 1] - st.blk_off[s]) * cpb;
        longest = longest > n ? longest : n;
    }
    return dim3(pn2::ceil_div(C, 64), pn2::ceil_div(longest, SLICE), st.nseg);
}

// forward finalize of one BatchNorm layer: (mean, M2) partial chunks of tile/2 rows, two per row tile of the table
int launch_bn_finalize(const float* partial, const FinScratch& fs, const Segs& S, int tile, int rows, const pn2_mlp_layer& L,
                       hipStream_t s, long long cm) {
    const int ch = tile / 2;
    if (S.nseg == 1) {
        if (pn2::ceil_div(rows, ch) <= 64)
            PN2_LAUNCH("bn_finalize", 8.0 * pn2::ceil_div(rows, ch) * L.cout, 0, bn_finalize_wave_kernel, dim3(pn2::ceil_div(L.cout, 4)),
                       dim3(256), s, partial, rows, ch, L.cout, L.gamma, L.beta, L.running_mean, L.running_var, L.eps, L.momentum,
                       L.stats, cm);
        else
        PN2_LAUNCH("bn_finalize", 8.0 * pn2::ceil_div(rows, ch) * L.cout, 0, bn_finalize_kernel, dim3(L.cout), dim3(256), s,
                   partial, rows, ch, L.cout, L.gamma, L.beta, L.running_mean, L.running_var, L.eps, L.momentum, L.stats, cm);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    if (L.cout > 512) return PN2_E_BADARG;
    int nblk = 0;
    const SegTable st = make_table(S, tile, &nblk);
    PN2_LAUNCH("bn_stats_reduce", 16.0 * nblk * L.cout, 0, (bn_slice_reduce_kernel<true>), slice_grid(st, 2, L.cout), dim3(256), s,
               partial, st, 2, ch, L.cout, fs.slices);
    PN2_LAUNCH("bn_finalize", 24.0 * (nblk * 2 / SLICE + S.nseg) * L.cout, 0, (bn_seg_finalize_kernel<true>),
               dim3(pn2::ceil_div(L.cout, 64), S.nseg), dim3(64 * FIN_W), s, partial, (const double*)fs.slices, st, 2, L.cout,
               L.gamma, L.beta, L.eps, L.stats, fs.segsum);
    if (L.running_mean)
        PN2_LAUNCH("bn_running", 8.0 * S.nseg * L.cout, 0, (bn_seg_ordered_kernel<true>), dim3(pn2::ceil_div(L.cout, 64)), dim3(64),
                   s, st, L.cout, (const float*)L.stats, (const double*)fs.segsum, L.running_mean, L.running_var, L.momentum,
                   (float*)nullptr, (float*)nullptr);
    PN2_LAUNCH_CHECK();
    return 0;
}

// backward finalize: row blocks of `R` rows hold `cpb` partial chunks each
int launch_bn_bwd_finalize(const float* partial, const FinScratch& fs, const Segs& S, int R, int cpb, int rows,
                           const pn2_mlp_layer& L, hipStream_t s, long long cm) {
    int nblk = 0;
    const SegTable st = make_table(S, R, &nblk);
    if (S.nseg == 1) {
        if (pn2::ceil_div(rows, R / cpb) <= 64)
            PN2_LAUNCH("bn_bwd_finalize", 8.0 * nblk * cpb * L.cout, 0, bn_bwd_finalize_wave_kernel, dim3(pn2::ceil_div(L.cout, 4)),
                       dim3(256), s, partial, pn2::ceil_div(rows, R / cpb), rows, L.cout, L.stats, L.dgamma, L.dbeta, cm);
        else
        PN2_LAUNCH("bn_bwd_finalize", 8.0 * nblk * cpb * L.cout, 0, bn_bwd_finalize_kernel, dim3(L.cout), dim3(256), s, partial,
                   pn2::ceil_div(rows, R / cpb), rows, L.cout, L.stats, L.dgamma, L.dbeta, cm);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    if (L.cout > 512) return PN2_E_BADARG;
    PN2_LAUNCH("bn_stats_reduce", 8.0 * nblk * cpb * L.cout, 0, (bn_slice_reduce_kernel<false>), slice_grid(st, cpb, L.cout),
               dim3(256), s, partial, st, cpb, R / cpb, L.cout, fs.slices);
    PN2_LAUNCH("bn_bwd_finalize", 16.0 * (nblk * cpb / SLICE + S.nseg) * L.cout, 0, (bn_seg_finalize_kernel<false>),
               dim3(pn2::ceil_div(L.cout, 64), S.nseg), dim3(64 * FIN_W), s, partial, (const double*)fs.slices, st, cpb, L.cout,
               (const float*)nullptr, (const float*)nullptr, 0.0f, L.stats, fs.segsum);
    if (L.dgamma || L.dbeta)
        PN2_LAUNCH("bn_param_grads", 16.0 * S.nseg * L.cout, 0, (bn_seg_ordered_kernel<false>), dim3(pn2::ceil_div(L.cout, 64)),
                   dim3(64), s, st, L.cout, (const float*)L.stats, (const double*)fs.segsum, (float*)nullptr, (float*)nullptr, 0.0f,
                   L.dgamma, L.dbeta);
    PN2_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// channel-major partials of the GEMM epilogues (two chunks per row tile), single-segment chains only; 0 = row-major
inline long long cm_stride(int rows, int tile, int nseg) {
    static const bool off = getenv("PN2_BN_ROW_MAJOR") != nullptr;   // A/B aid
    return (nseg == 1 && !off) ? 2ll * 2 * pn2::ceil_div(rows, tile) : 0;
}

// row-block size of the dgrad that produces a linked chain's input gradient (two partial chunks per block)
inline int link_tile(int rows, int cin) { return pick_tile(rows, cin, 1); }

// bf16 mode with bfloat16 STORAGE of the chain's pre-BatchNorm rows and gradient rows: every contraction of the chain must be one of
// the 128-tile, 16-byte-staged kernels (the only ones with the bfloat16 load / store paths), no pooling.
bool chain_bf16_storage_ok(int rows, const pn2_mlp_layer* layers, int nlayers, int pool_k) {
    if (pool_k > 1 || rows <= 0 || nlayers <= 0) return false;
    for (int i = 0; i < nlayers; ++i) {
        const pn2_mlp_layer& L = layers[i];
        const bool last = i == nlayers - 1;
        if (narrow_ok(L, last, pool_k)) continue;
        if (!L.has_bn || L.cin % 4 || L.cout % 4) return false;
        if (pick_tile(rows, L.cout, 1) != 128 || pick_tile(rows, L.cin, 1) != 128) return false;
    }
    return true;
}

// cooperative launches: asked for by the caller (a sync buffer) and not switched off (PN2_NO_COOP: A/B and test aid)
inline bool coop_usable(const pn2_coop* c) {
    return c && c->sync && !getenv("PN2_NO_COOP");
}

// bytes of ONE of the two partial regions at the start of a chain workspace (behind the segmented finalizes' scratch)
namespace {
size_t partial_region_bytes(int rows, const pn2_mlp_layer* layers, int nlayers, int nseg);
}

// ===================================================================================================== C ABI
namespace {
// (shared scratch need of the layers, slab arena) of a chain workspace
void workspace_parts(int rows, const pn2_mlp_layer* layers, int nlayers, int nseg, size_t* need_out, size_t* arena_out) {
    size_t need = 0, arena = 0;
    for (int i = 0; i < nlayers; ++i) {
        const size_t cin = layers[i].cin, cout = layers[i].cout;
        // per-row-chunk partials (forward stats, backward sums): chunks are >= 32 rows (+ 2 per segment boundary);
        // channels up to max(cin, cout)
        const size_t cmax = cin > cout ? cin : cout;
        const size_t part = ((size_t)pn2::ceil_div(rows, 32) + 2 * (size_t)nseg) * 2 * cmax * sizeof(float);
        const size_t cs = (size_t)pn2::ceil_div(rows, CS_ROWS) * cout * sizeof(float);         // bias column sums
        const WgradPlan wp = plan_wgrad(rows, (int)cout, (int)cin, nseg);
        const size_t slab = (size_t)wp.nsplit * cout * cin * sizeof(float);
        const size_t nblk = (size_t)pn2::ceil_div(rows, NR_ROWS) + nseg;
        const size_t narrow = nblk * (2 * cin + cout * cin + cout) * sizeof(float);           // narrow_bwd partials
        size_t m = part > cs ? part : cs;
        m = m > narrow ? m : narrow;
        need = need > m ? need : m;
        arena += align256(slab);   // the weight-gradient slabs of ALL layers stay until the chain's one reduction launch
    }
    *need_out = align256(need);
    *arena_out = arena;
}
size_t partial_region_bytes(int rows, const pn2_mlp_layer* layers, int nlayers, int nseg) {
    size_t need = 0, arena = 0;
    workspace_parts(rows, layers, nlayers, nseg, &need, &arena);
    return need;
}
}  // namespace

extern "C" int pn2_mlp_chain_bf16_storage(int rows, const pn2_mlp_layer* layers, int nlayers, int pool_k) {
    return layers && chain_bf16_storage_ok(rows, layers, nlayers, pool_k) ? 1 : 0;
}

extern "C" size_t pn2_mlp_workspace_bytes(int rows, const pn2_mlp_layer* layers, int nlayers, int nseg) {
    if (rows <= 0 || !layers || nlayers <= 0) return 0;
    if (nseg < 1) nseg = 1;
    size_t need = 0, arena = 0;
    workspace_parts(rows, layers, nlayers, nseg, &need, &arena);
    // (two partial regions: the cooperative chain kernels alternate between them from layer to layer)
    return slice_region_of(rows, layers, nlayers, nseg) + 2 * need + 256 + arena;
}

namespace {
// start of the slab arena inside a chain workspace (behind the shared scratch region)
size_t slab_arena_offset(int rows, const pn2_mlp_layer* layers, int nlayers, int nseg) {
    size_t arena = 0;
    for (int i = 0; i < nlayers; ++i) {
        const WgradPlan wp = plan_wgrad(rows, layers[i].cout, layers[i].cin, nseg);
        arena += align256((size_t)wp.nsplit * layers[i].cout * layers[i].cin * sizeof(float));
    }
    return pn2_mlp_workspace_bytes(rows, layers, nlayers, nseg) - arena;
}
}  // namespace

extern "C" int pn2_mlp_chain_fwd_f32(const float* x, int64_t ldx, int rows, const pn2_mlp_layer* layers, int nlayers,
                                     int training, int pool_k, float* out, int32_t* pool_arg, const pn2_segments* segments,
                                     int precision, const pn2_coop* coop, void* workspace, size_t workspace_bytes, void* stream) {
    const bool lazy_out = (precision & PN2_CHAIN_LAZY_OUT) != 0;
    const int x16 = (precision & PN2_CHAIN_X_BF16) ? 1 : 0, s16 = (precision & PN2_CHAIN_STORE_BF16) ? 1 : 0;
    precision &= ~(PN2_CHAIN_LAZY_OUT | PN2_CHAIN_X_BF16 | PN2_CHAIN_STORE_BF16);
    if ((x16 || s16) && (precision != PN2_PRECISION_BF16 || !s16 || !chain_bf16_storage_ok(rows, layers, nlayers, pool_k)))
        return PN2_E_BADARG;
    if (!x || !layers || nlayers <= 0 || rows <= 0 || (!out && !lazy_out) || (pool_k > 1 && (!pool_arg || rows % pool_k)))
        return PN2_E_BADARG;
    if (lazy_out && (pool_k > 1 || !layers[nlayers - 1].has_bn)) return PN2_E_BADARG;
    if (precision != PN2_PRECISION_F32 && precision != PN2_PRECISION_BF16) return PN2_E_BADARG;
    if (!segs_valid(rows, segments, pool_k)) return PN2_E_BADARG;
    const Segs S = make_segs(rows, training ? segments : nullptr);   // eval mode: one coefficient block serves every row
    if (workspace_bytes < pn2_mlp_workspace_bytes(rows, layers, nlayers, S.nseg) || !workspace) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const FinScratch sp = fin_scratch(workspace, S.nseg, chain_cmax(layers, nlayers));   // segmented finalizes' scratch
    workspace = (char*)workspace + slice_region_of(rows, layers, nlayers, S.nseg);
    if (coop_usable(coop) && training && S.nseg == 1 && !lazy_out && !layers[0].in_stats && out &&
        pn2::coop::shapes_ok(rows, layers, nlayers, pool_k)) {
        // a deep level: the whole chain as one persistent launch (chain_coop.hip)
        pn2::coop::FwdCall c{};
        c.x = x, c.ldx = ldx, c.rows = rows, c.layers = layers, c.nlayers = nlayers, c.pool_k = pool_k, c.out = out, c.arg = pool_arg;
        c.partial[0] = (float*)workspace;
        c.partial[1] = (float*)((char*)workspace + partial_region_bytes(rows, layers, nlayers, S.nseg));
        c.ctl = coop, c.stream = s;
        return pn2::coop::forward(c);
    }
    if (x16 && !layers[0].in_stats) return PN2_E_BADARG;   // bfloat16 input rows only exist as a linked producer's rows
    Act in{x, ldx, layers[0].in_stats, layers[0].in_stats ? layers[0].in_relu : 0, x16};   // linked chain: BN(+ReLU) while staging
    for (int i = 0; i < nlayers; ++i) {
        const pn2_mlp_layer& L = layers[i];
        if (!L.weight || L.cin <= 0 || L.cout <= 0) return PN2_E_BADARG;
        const bool last = i == nlayers - 1;
        const bool direct_out = last && !L.has_bn && pool_k <= 1;   // a bare conv at the end writes `out` itself
        float* y = direct_out ? out : L.y;
        if (!y || (L.has_bn && !L.stats)) return PN2_E_BADARG;
        GemmArgs g{};
        g.precision = precision;
        g.A = act_operand(in, rows, L.cin);
        g.B = plain(L.weight, L.cin, L.cout, L.cin);
        g.M = rows;
        g.N = L.cout;
        g.K = L.cin;
        g.C = y;
        g.ldc = L.cout;
        g.c16 = (s16 && !direct_out) ? 1 : 0;
        g.bias = L.bias;
        g.partial = (L.has_bn && training) ? (float*)workspace : nullptr;
        const int tile = pick_tile(rows, L.cout, 1);
        g.pstride = cm_stride(rows, tile, S.nseg);
        int st = 0;
        if (narrow_ok(L, last, pool_k) && in.ld % 4 == 0 && aligned16(in.p)) {
            st = launch_narrow_fwd(in, rows, L, y, S, s);
        } else {
            st = in.coef ? launch_gemm<true, TR_BNRELU, true, TR_PLAIN, EPI_FWD>(g, S, tile, s)
                         : launch_gemm<true, TR_PLAIN, true, TR_PLAIN, EPI_FWD>(g, S, tile, s);
        }
        if (st) return st;
        if (L.has_bn) {
            if (training) {
                if ((st = launch_bn_finalize((const float*)workspace, sp, S, tile, rows, L, s, cm_stride(rows, tile, S.nseg)))) return st;
            } else {
                if (!L.running_mean || !L.running_var) return PN2_E_BADARG;
                PN2_LAUNCH("bn_eval_coef", 36.0 * L.cout, 0, bn_eval_coef_kernel, dim3(pn2::ceil_div(L.cout, 256)), dim3(256), s,
                           L.cout, L.gamma, L.beta, L.running_mean, L.running_var, L.eps, L.stats);
                PN2_LAUNCH_CHECK();
            }
            in = Act{y, L.cout, L.stats, L.relu, s16};
        } else {
            if (L.relu && !last) return PN2_E_BADARG;  // ReLU without BatchNorm only exists fused into a BN layer here
            in = Act{y, L.cout, nullptr, 0, 0};
        }
        if (last && !direct_out && !lazy_out) {
            const int C = L.cout;
            if (!L.has_bn || C % 4) return PN2_E_BADARG;
            int nblk = 0;
            if (pool_k > 1) {
                const long long groups = rows / pool_k;
                const SegTable tb = make_table(S, pool_k, &nblk);
                PN2_LAUNCH("bn_relu_maxpool", 4.0 * rows * C + 8.0 * groups * C, 0, bn_relu_maxpool_kernel,
                           dim3(grid1d(groups * C)), dim3(256), s, (const float*)y, groups, pool_k, C, (const float*)L.stats,
                           L.relu, out, pool_arg, tb);
            } else if (S.nseg > 1) {
                const SegTable tb = make_table(S, APPLY_R, &nblk);
                PN2_LAUNCH("bn_relu_apply", 8.0 * rows * C, 0, bn_relu_apply_seg_kernel, dim3(nblk), dim3(256), s, (const float*)y, tb,
                           C, (const float*)L.stats, L.relu, out, s16);
            } else {
                PN2_LAUNCH("bn_relu_apply", 8.0 * rows * C, 0, bn_relu_apply_kernel, dim3(grid1d((long long)rows * C / 4)),
                           dim3(256), s, (const float*)y, (long long)rows * C / 4, C, (const float*)L.stats, L.relu, out, s16);
            }
            PN2_LAUNCH_CHECK();
        }
    }
    return 0;
}

extern "C" int pn2_mlp_chain_bwd_f32(const float* x, int64_t ldx, int rows, const pn2_mlp_layer* layers, int nlayers,
                                     int pool_k, const float* dout, const int32_t* pool_arg, float* dx, int64_t lddx,
                                     int dx_first_col, float* scratch_a, float* scratch_b, const pn2_segments* segments,
                                     int precision, pn2_wgrad_tasks* deferred, const pn2_coop* coop, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!x || !layers || nlayers <= 0 || rows <= 0 || !dout || !scratch_a || !scratch_b) return PN2_E_BADARG;
    if (dx_first_col < 0 || dx_first_col >= layers[0].cin) return PN2_E_BADARG;
    const int accumulate_dx = (precision & PN2_CHAIN_ACCUMULATE_DX) ? 1 : 0;
    // deferred reductions go to the caller's list (pn2_hip.h); a list without room makes this call reduce in place
    const bool defer_wgrad = deferred && deferred->n >= 0 && deferred->n + nlayers <= PN2_WGRAD_TASKS_MAX;
    bool zero_lead = (precision & PN2_CHAIN_ZERO_LEAD) != 0 && dx && dx_first_col > 0;
    if (zero_lead && accumulate_dx) return PN2_E_BADARG;
    // bfloat16 storage (pn2_hip.h): which of the row tensors of this call are __bf16
    const int x16 = (precision & PN2_CHAIN_X_BF16) ? 1 : 0, s16 = (precision & PN2_CHAIN_STORE_BF16) ? 1 : 0;
    const int dout16 = (precision & PN2_CHAIN_DOUT_BF16) ? 1 : 0, dx16 = (precision & PN2_CHAIN_DX_BF16) ? 1 : 0;
    precision &= ~(PN2_CHAIN_ACCUMULATE_DX | PN2_CHAIN_ZERO_LEAD | PN2_CHAIN_X_BF16 | PN2_CHAIN_STORE_BF16 | PN2_CHAIN_DOUT_BF16 |
                   PN2_CHAIN_DX_BF16);
    if ((x16 || s16 || dout16 || dx16) &&
        (precision != PN2_PRECISION_BF16 || !s16 || dx_first_col || !chain_bf16_storage_ok(rows, layers, nlayers, pool_k)))
        return PN2_E_BADARG;
    if (x16 && !layers[0].in_stats) return PN2_E_BADARG;
    if (precision != PN2_PRECISION_F32 && precision != PN2_PRECISION_BF16) return PN2_E_BADARG;
    if (accumulate_dx && !dx) return PN2_E_BADARG;
    if (!segs_valid(rows, segments, pool_k)) return PN2_E_BADARG;
    const Segs S = make_segs(rows, segments);
    if (workspace_bytes < pn2_mlp_workspace_bytes(rows, layers, nlayers, S.nseg) || !workspace) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const FinScratch sp = fin_scratch(workspace, S.nseg, chain_cmax(layers, nlayers));
    float* ws = (float*)((char*)workspace + slice_region_of(rows, layers, nlayers, S.nseg));
    char* arena = (char*)workspace + slab_arena_offset(rows, layers, nlayers, S.nseg);
    SlabTasks tasks{};
    if (coop_usable(coop) && S.nseg == 1 && !accumulate_dx && !layers[0].in_stats && !layers[0].in_partial &&
        !layers[nlayers - 1].out_partial && (pool_k <= 1 || pool_arg) && (dx_first_col == 0 || zero_lead || !dx) &&
        pn2::coop::shapes_ok(rows, layers, nlayers, pool_k)) {
        // a deep level: the whole backward chain as one persistent launch (chain_coop.hip); the slab reductions stay with
        // the caller exactly as on the launch-per-layer path
        pn2::coop::BwdCall c{};
        c.x = x, c.ldx = ldx, c.rows = rows, c.layers = layers, c.nlayers = nlayers, c.pool_k = pool_k, c.dout = dout, c.arg = pool_arg;
        c.dx = dx, c.lddx = lddx, c.dx_first_col = dx_first_col, c.zero_lead = zero_lead ? 1 : 0;
        c.scratch[0] = scratch_a, c.scratch[1] = scratch_b;
        c.partial[0] = ws;
        c.partial[1] = (float*)((char*)ws + partial_region_bytes(rows, layers, nlayers, S.nseg));
        c.ctl = coop, c.stream = s;
        for (int i = nlayers - 1; i >= 0; --i) {
            const pn2_mlp_layer& L = layers[i];
            if (!L.dweight) continue;
            const WgradPlan wp = plan_wgrad(rows, L.cout, L.cin, S.nseg);      // sizes the arena (pn2_mlp_workspace_bytes)
            int kps = 0, nsplit = 0;
            pn2::coop::plan_slabs(rows, L.cout, L.cin, &kps, &nsplit);
            if (nsplit > wp.nsplit) {
                kps = pn2::ceil_div(pn2::ceil_div(rows, wp.nsplit), 4 * BK) * 4 * BK;
                nsplit = pn2::ceil_div(rows, kps);
            }
            c.slab[i] = (float*)arena, c.kps[i] = kps, c.nsplit[i] = nsplit;
            if (!defer_wgrad) {
                bool clash = tasks.n == SLAB_TASKS;
                for (int t = 0; t < tasks.n && !clash; ++t) clash = tasks.t[t].out == L.dweight;
                if (clash) return PN2_E_BADARG;   // (a weight shared by two layers of one chain: not cooperative)
            }
            add_slab_task(tasks, (const float*)arena, nsplit, (long long)L.cout * L.cin, L.dweight);
            arena += align256((size_t)wp.nsplit * L.cout * L.cin * sizeof(float));
        }
        const int st = pn2::coop::backward(c);
        if (st) return st;
        if (defer_wgrad) {
            for (int i = 0; i < tasks.n; ++i) {
                pn2_wgrad_task& t = deferred->t[deferred->n++];
                t.slab = tasks.t[i].slab, t.out = tasks.t[i].out, t.mn = tasks.t[i].mn, t.nsplit = tasks.t[i].nsplit, t.reserved = 0;
            }
        } else {
            flush_slab_tasks(tasks, s);
        }
        PN2_LAUNCH_CHECK();
        return 0;
    }
    // dz of the last layer: upstream gradient, or the max-pool scatter of it
    const float* dz = dout;
    int dz16 = dout16;
    long long lddz = layers[nlayers - 1].cout;
    float* bufs[2] = {scratch_a, scratch_b};
    int which = 0;
    int pooled_R = 0;   // > 0: the max-pool scatter left the pooled layer's BatchNorm-backward sums, row blocks of this size
    bool pooled_dyp = false;   // ... and no dense dz: bufs[0] holds the arg-max rows as bytes, the pooled layer's GEMMs rebuild dz
    if (pool_k > 1) {
        if (!pool_arg) return PN2_E_BADARG;
        const pn2_mlp_layer& PL = layers[nlayers - 1];
        const int C = PL.cout;
        const bool no_sums = getenv("PN2_POOL_NO_SUMS") != nullptr;   // tests: the two-pass path
        if (PL.has_bn && PL.y && PL.stats && pool_k <= PS_ROWS && !no_sums) {
            // the BatchNorm-backward sums of the pooled layer come out of the scatter (no bn_bwd_reduce pass); row blocks
            // of whole groups, as many as keep ~256 workgroups busy
            int gpb = (rows / pool_k) / 256;
            if (gpb > PS_ROWS / pool_k) gpb = PS_ROWS / pool_k;
            if (gpb * pool_k < 32) gpb = pn2::ceil_div(32, pool_k);   // the partial buffer is sized for chunks of >= 32 rows
            pooled_R = gpb * pool_k;
            int nblk = 0;
            const SegTable tb = make_table(S, pooled_R, &nblk);
            // no dense dz at all when both GEMMs of the pooled layer can rebuild it while staging (TR_DYP, mlp_tile.h)
            {
                const bool need_dx = nlayers > 1 || dx != nullptr;
                const int skip = nlayers == 1 ? dx_first_col : 0;
                const long long in_ld = nlayers == 1 ? ldx : layers[nlayers - 2].cout;
                const float* in_p = nlayers == 1 ? x : layers[nlayers - 2].y;
                const bool pow2 = (pool_k & (pool_k - 1)) == 0;
                bool ok = pow2 && pool_k <= 256 && C % 4 == 0 && !dout16 && !s16 && !(nlayers == 1 && x16) && aligned16(dout) &&
                          aligned16(PL.y) && in_ld % 4 == 0 && PL.cin % 4 == 0 && aligned16(in_p) && aligned16(PL.weight) &&
                          skip % 4 == 0 && !getenv("PN2_NO_POOL_DYP");
                // (a deep level's GEMMs are latency-bound: the extra address arithmetic costs them more than the dense tensor's
                // bytes -- measured on the headline step, whose pooled chains have <= 32 768 rows: 3.94 -> 4.09 ms with it)
                long long min_rows = 65536;
                if (const char* e = getenv("PN2_POOL_DYP_MIN_ROWS")) min_rows = atoll(e);
                if (rows < min_rows) ok = false;
                if (ok && PL.dweight) ok = dyp_tile_ok(plan_wgrad(rows, PL.cout, PL.cin, S.nseg).tile, PL.cout, PL.cin, rows, EPI_SLAB, S.nseg);
                if (ok && need_dx) ok = dyp_tile_ok(pick_tile(rows, PL.cin - skip, 1), rows, PL.cin - skip, PL.cout, EPI_STORE, S.nseg);
                if (ok && need_dx && nlayers == 1 && (lddx % 4 != 0 || !aligned16(dx))) ok = false;
                pooled_dyp = ok;
            }
            PN2_LAUNCH("maxpool_scatter", (pooled_dyp ? 0.0 : 4.0 * rows * C) + 17.0 * (rows / pool_k) * C, 0, maxpool_scatter_sums_kernel,
                       dim3(nblk), dim3(256), s, dout, pool_arg, pool_k, C, pooled_R, (const float*)PL.y, (const float*)PL.stats,
                       PL.relu, bufs[which], ws, tb, zero_lead ? dx : nullptr, (long long)lddx, zero_lead ? dx_first_col : 0,
                       pooled_dyp ? (unsigned char*)bufs[which] : (unsigned char*)nullptr);
            zero_lead = false;
        } else {
            PN2_LAUNCH("maxpool_scatter", 4.0 * rows * C + 8.0 * (rows / pool_k) * C, 0, maxpool_scatter_kernel,
                       dim3(grid1d((long long)rows * C)), dim3(256), s, dout, pool_arg, (long long)(rows / pool_k), pool_k, C,
                       bufs[which]);
        }
        PN2_LAUNCH_CHECK();
        dz = bufs[which];
        which ^= 1;
    }
    if (zero_lead) {
        const long long total = (long long)rows * dx_first_col;
        PN2_LAUNCH("zero_lead", 4.0 * total, 0, zero_lead_kernel, dim3(grid1d(total)), dim3(256), s, dx, (long long)lddx,
                   dx_first_col, total);
        PN2_LAUNCH_CHECK();
    }
    // BatchNorm-backward partials of the current layer already in ws?  (R, cpb): row-block size and chunks per block
    int fused_R = pooled_R, fused_cpb = pooled_R ? 1 : 0;
    long long fused_cm = 0;   // layout of those partials (cm_stride)
    for (int i = nlayers - 1; i >= 0; --i) {
        const pn2_mlp_layer& L = layers[i];
        const bool last = i == nlayers - 1;
        const float* y = (last && !L.has_bn && pool_k <= 1) ? nullptr : L.y;
        // ---- BatchNorm backward reductions -> coefficients a, b and dgamma, dbeta
        if (L.has_bn) {
            int R = fused_R, cpb = fused_cpb, st;
            long long cm = fused_cm;
            const float* partial = (const float*)ws;
            if (last && L.out_partial && pool_k <= 1) {   // written by the linked consumer chain's dgrad epilogue
                partial = L.out_partial;
                R = L.out_partial_rows;
                cpb = L.out_partial_cpb;
                if (R <= 0 || cpb != 2) return PN2_E_BADARG;
                cm = cm_stride(rows, R, S.nseg);
            } else if (!fused_R) {   // otherwise written by the dgrad epilogue (or the narrow backward kernel) of layer i + 1
                R = RB;
                cpb = 1;
                cm = 0;
                int nblk = 0;
                const SegTable tb = make_table(S, RB, &nblk);
                const bool vec = (L.cout % 4 == 0) && (lddz % 4 == 0) && aligned16(dz) && aligned16(y);
                if (vec)
                    PN2_LAUNCH("bn_bwd_reduce", 8.0 * rows * L.cout, 0, (bn_bwd_reduce_kernel<true>), dim3(nblk), dim3(256), s, dz,
                               lddz, y, (long long)L.cout, rows, L.cout, (const float*)L.stats, L.relu, ws, tb, dz16, s16);
                else
                    PN2_LAUNCH("bn_bwd_reduce", 8.0 * rows * L.cout, 0, (bn_bwd_reduce_kernel<false>), dim3(nblk), dim3(256), s, dz,
                               lddz, y, (long long)L.cout, rows, L.cout, (const float*)L.stats, L.relu, ws, tb, dz16, s16);
                PN2_LAUNCH_CHECK();
            }
            if ((st = launch_bn_bwd_finalize(partial, sp, S, R, cpb, rows, L, s, cm))) return st;
        } else if (L.dbias && !(narrow_ok(L, last, pool_k) && lddz == L.cout)) {
            const int nblk = pn2::ceil_div(rows, CS_ROWS);
            PN2_LAUNCH("colsum", 4.0 * rows * L.cout, 0, colsum_kernel, dim3(nblk), dim3(256), s, dz, lddz, rows, L.cout, ws);
            PN2_LAUNCH("colsum_finalize", 4.0 * nblk * L.cout, 0, colsum_finalize_kernel, dim3(L.cout), dim3(64), s,
                       (const float*)ws, nblk, L.cout, L.dbias);
            PN2_LAUNCH_CHECK();
        }
        fused_R = fused_cpb = 0;
        fused_cm = 0;
        // layer input as an activation source
        Act in = i == 0 ? Act{x, ldx, layers[0].in_stats, layers[0].in_stats ? layers[0].in_relu : 0, x16}
                        : Act{layers[i - 1].y, layers[i - 1].cout, layers[i - 1].has_bn ? layers[i - 1].stats : nullptr,
                              layers[i - 1].relu, s16};
        // ---- narrow last layer: one vector-ALU pass does dgrad, wgrad, bias sums and the previous layer's BN sums
        if (narrow_ok(L, last, pool_k) && lddz == L.cout && in.ld % 4 == 0 && aligned16(in.p)) {
            int nblk = 0;
            const SegTable tb = make_table(S, NR_ROWS, &nblk);
            float* part_s = ws;
            float* part_w = part_s + (size_t)nblk * 2 * L.cin;
            float* part_b = part_w + (size_t)nblk * L.cout * L.cin;
            float* target = i > 0 ? bufs[which] : (dx ? dx : bufs[which]);
            if (i == 0 && dx && (lddx != L.cin || accumulate_dx)) return PN2_E_BADARG;   // the narrow kernel stores
            if (dz16) return PN2_E_BADARG;                                               // (its upstream gradient is fp32)
            const int t16 = (i > 0 || !dx) ? s16 : dx16;
            int st = launch_narrow_bwd(in, rows, L, dz, target, part_s, part_w, part_b, tb, nblk, s, t16);
            if (st) return st;
            if (L.dweight)
                launch_slab_reduce((const float*)part_w, nblk, (long long)L.cout * L.cin, L.dweight, s);
            if (L.dbias)
                PN2_LAUNCH("colsum_finalize", 4.0 * nblk * L.cout, 0, colsum_finalize_kernel, dim3(L.cout), dim3(64), s,
                           (const float*)part_b, nblk, L.cout, L.dbias);
            PN2_LAUNCH_CHECK();
            if (i > 0 && layers[i - 1].has_bn) {
                fused_R = NR_ROWS;
                fused_cpb = 1;
            }
            dz = target;
            dz16 = t16;
            lddz = L.cin;
            which ^= 1;
            continue;
        }
        // dY operand (through BatchNorm+ReLU backward when the layer has one)
        Operand dy = plain(dz, lddz, rows, L.cout);
        dy.p16 = dz16;
        if (L.has_bn) {
            dy.q = y;
            dy.q16 = s16;
            dy.ldq = L.cout;
            dy.coef = L.stats;
            dy.cstride = L.cout;
            dy.relu = L.relu;
        }
        const bool dyp = last && pooled_dyp;
        if (dyp) {
            dy.p = dout;
            dy.parg8 = (const unsigned char*)dz;
            dy.pool_shift = __builtin_ctz((unsigned)pool_k);
        }
        // ---- wgrad: dW[cout][cin] += dY^T X, reduction over rows split across blocks
        if (L.dweight) {
            const WgradPlan wp = plan_wgrad(rows, L.cout, L.cin, S.nseg);
            GemmArgs g{};
            g.precision = precision;
            g.A = dy;                       // [rows = K][cout = M], direct layout
            g.B = act_operand(in, rows, L.cin);
            g.M = L.cout;
            g.N = L.cin;
            g.K = rows;
            g.C = (float*)arena;
            g.ldc = L.cin;
            g.k_per_split = wp.kps;
            int st, nsplit = 0;
            if (dyp)
                st = in.coef ? launch_gemm_dyp<false, false, TR_BNRELU, EPI_SLAB>(g, S, wp.tile, s, &nsplit)
                             : launch_gemm_dyp<false, false, TR_PLAIN, EPI_SLAB>(g, S, wp.tile, s, &nsplit);
            else if (L.has_bn)
                st = in.coef ? launch_gemm<false, TR_DY, false, TR_BNRELU, EPI_SLAB>(g, S, wp.tile, s, &nsplit)
                             : launch_gemm<false, TR_DY, false, TR_PLAIN, EPI_SLAB>(g, S, wp.tile, s, &nsplit);
            else
                st = in.coef ? launch_gemm<false, TR_PLAIN, false, TR_BNRELU, EPI_SLAB>(g, S, wp.tile, s, &nsplit)
                             : launch_gemm<false, TR_PLAIN, false, TR_PLAIN, EPI_SLAB>(g, S, wp.tile, s, &nsplit);
            if (st) return st;
            if (!defer_wgrad) {   // never two reductions into the same gradient in one launch (shared weights)
                bool clash = tasks.n == SLAB_TASKS;
                for (int t = 0; t < tasks.n && !clash; ++t) clash = tasks.t[t].out == L.dweight;
                if (clash) flush_slab_tasks(tasks, s);
            }
            if (tasks.n == SLAB_TASKS) return PN2_E_BADARG;   // more layers than a task table holds
            add_slab_task(tasks, (const float*)arena, nsplit, (long long)L.cout * L.cin, L.dweight);
            arena += align256((size_t)wp.nsplit * L.cout * L.cin * sizeof(float));
            PN2_LAUNCH_CHECK();
        }
        // ---- dgrad: dX[rows][cin] = dY W; when the previous layer has a BatchNorm its backward column sums are
        //      taken from the accumulators in the epilogue (no separate pass over dX and Y)
        const bool need_dx = i > 0 || dx != nullptr;
        if (need_dx) {
            // the chain's input gradient may be wanted from column dx_first_col on only (the centred coordinates in
            // front of a grouped set-abstraction input carry no gradient): fewer -- and better filled -- column tiles
            const int skip = i == 0 ? dx_first_col : 0;
            float* target = i > 0 ? bufs[which] : dx + skip;
            const long long ldt = i > 0 ? L.cin : lddx;
            GemmArgs g{};
            g.precision = precision;
            g.A = dy;                       // [rows = M][cout = K]
            g.B = plain(L.weight + skip, L.cin, L.cout, L.cin - skip);   // [cout = K][cin = N], direct layout
            g.M = rows;
            g.N = L.cin - skip;
            g.K = L.cout;
            g.C = target;
            g.ldc = ldt;
            g.c16 = i > 0 ? s16 : dx16;
            g.ey16 = i > 0 ? s16 : x16;
            g.accumulate = i == 0 ? accumulate_dx : 0;
            const int tile = pick_tile(rows, L.cin - skip, 1);
            if (i > 0 && layers[i - 1].has_bn) {
                g.partial = ws;
                g.ey = layers[i - 1].y;
                g.ldey = layers[i - 1].cout;
                g.ecoef = layers[i - 1].stats;
                g.erelu = layers[i - 1].relu;
                fused_R = tile;
                fused_cpb = 2;
                g.pstride = fused_cm = cm_stride(rows, tile, S.nseg);
            } else if (i == 0 && in.coef && L.in_partial) {   // linked chain: the producing layer's BatchNorm-backward sums
                if (skip || tile != link_tile(rows, L.cin)) return PN2_E_BADARG;
                g.partial = L.in_partial;
                g.ey = x;
                g.ldey = ldx;
                g.ecoef = in.coef;
                g.erelu = in.relu;
                g.pstride = cm_stride(rows, tile, S.nseg);
            }
            int st = dyp      ? launch_gemm_dyp<true, false, TR_PLAIN, EPI_STORE>(g, S, tile, s)
                     : L.has_bn ? launch_gemm<true, TR_DY, false, TR_PLAIN, EPI_STORE>(g, S, tile, s)
                              : launch_gemm<true, TR_PLAIN, false, TR_PLAIN, EPI_STORE>(g, S, tile, s);
            if (st) return st;
            dz = target;
            dz16 = i > 0 ? s16 : dx16;
            lddz = ldt;
            which ^= 1;
        }
    }
    if (defer_wgrad) {
        for (int i = 0; i < tasks.n; ++i) {
            pn2_wgrad_task& t = deferred->t[deferred->n++];
            t.slab = tasks.t[i].slab, t.out = tasks.t[i].out, t.mn = tasks.t[i].mn, t.nsplit = tasks.t[i].nsplit, t.reserved = 0;
        }
    } else {
        flush_slab_tasks(tasks, s);
    }
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pn2_mlp_link_partial_bytes(int rows, int cin, int nseg, int32_t* block_rows, int32_t* chunks_per_block) {
    if (rows <= 0 || cin <= 0) return 0;
    if (nseg < 1) nseg = 1;
    if (block_rows) *block_rows = link_tile(rows, cin);
    if (chunks_per_block) *chunks_per_block = 2;
    return ((size_t)pn2::ceil_div(rows, 32) + 2 * (size_t)nseg) * 2 * (size_t)cin * sizeof(float);
}

// The reductions pn2_mlp_chain_bwd_f32 calls left in their callers' lists (`deferred`), in as few launches as the task table
// allows (SLAB_TASKS each).  The workspaces of those calls must still be alive.  Stateless: everything arrives by argument.
extern "C" int pn2_mlp_reduce_wgrad(const pn2_wgrad_task* tasks, int n, void* stream) {
    if (n < 0 || (n > 0 && !tasks)) return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    SlabTasks T{};
    for (int k = 0; k < n; ++k) {
        const pn2_wgrad_task& t = tasks[k];
        if (!t.slab || !t.out || t.mn <= 0 || t.nsplit <= 0) return PN2_E_BADARG;
        // two reductions into the SAME gradient (a layer that ran more than once in this backward pass: the mini-batches of
        // forward_hierarchical) must not share a launch -- their blocks would read-modify-write the same elements
        bool clash = T.n == SLAB_TASKS;
        for (int i = 0; i < T.n && !clash; ++i) clash = T.t[i].out == t.out;
        if (clash) flush_slab_tasks(T, s);
        add_slab_task(T, t.slab, t.nsplit, t.mn, t.out);
    }
    flush_slab_tasks(T, s);
    PN2_LAUNCH_CHECK();
    return 0;
}

// ================================================================== feature propagation, first convolution hoisted
// (pn2_hip.h "Feature propagation with the first convolution HOISTED").  Forward: one workgroup interpolates a block of R rows
// of the C-wide layer -- a thread owns four channels of IB_PASSES rows per statistics chunk -- writes them and reduces the two
// chunks' (mean, M2) from the registers (two-pass inside the chunk, as the GEMM epilogue does).  Backward: three_nn.hip's
// bucketed scatter with the TR_DY transform on its loads (interp.h).
#include "interp.h"

namespace {

constexpr int IB_PASSES = 8;
inline int interp_block_rows(int C) { return 2 * IB_PASSES * (256 / (C / 4)); }   // 512 / 256 / 128 / 64 rows at C = 32 / 64 / 128 / 256

__device__ __forceinline__ int checked_index(int j, int n, int32_t* status) {
    if ((unsigned)j < (unsigned)n) return j;
    if (status) atomicOr(status, PN2_STATUS_BAD_INDEX);   // untrusted index: row 0 instead, and say so
    return 0;
}

// Where a row of the layer comes from.  InterpSrc: the three-neighbour interpolation of the sampled rows q (feature
// propagation).  GroupSrc: a grouped row (b, s, k) of a set-abstraction level, gf[b][idx] + Wx (xyz[b][idx] - new_xyz[b][s]) --
// the features' share of the first conv was applied to the SOURCE points (gf, bias included), the centred coordinates'
// share (three multiply-adds per channel, the same centred differences the reference forms) is added here.
struct InterpSrc {
    const float* q;
    const int32_t* idx;
    const float* w;
    const int32_t* row_cloud;
    int N, S;
    __device__ __forceinline__ void init(int, int) {}
    __device__ __forceinline__ float4 row(int row, int c, int C, int32_t* status) const {
        const long long r3 = 3ll * row;
        const int j0 = checked_index(idx[r3], S, status), j1 = checked_index(idx[r3 + 1], S, status),
                  j2 = checked_index(idx[r3 + 2], S, status);
        const float w0 = w[r3], w1 = w[r3 + 1], w2 = w[r3 + 2];
        const float* qb = q + (long long)(row_cloud ? row_cloud[row] : row / N) * S * C + c;
        const float4 a = *(const float4*)(qb + (long long)j0 * C);
        const float4 b = *(const float4*)(qb + (long long)j1 * C);
        const float4 d = *(const float4*)(qb + (long long)j2 * C);
        // the reference's order (blocks.py:204): (p0*w0 + p1*w1) + p2*w2, separate multiplies and adds
        float4 v;
        v.x = __fadd_rn(__fadd_rn(__fmul_rn(a.x, w0), __fmul_rn(b.x, w1)), __fmul_rn(d.x, w2));
        v.y = __fadd_rn(__fadd_rn(__fmul_rn(a.y, w0), __fmul_rn(b.y, w1)), __fmul_rn(d.y, w2));
        v.z = __fadd_rn(__fadd_rn(__fmul_rn(a.z, w0), __fmul_rn(b.z, w1)), __fmul_rn(d.z, w2));
        v.w = __fadd_rn(__fadd_rn(__fmul_rn(a.w, w0), __fmul_rn(b.w, w1)), __fmul_rn(d.w, w2));
        return v;
    }
};
struct GroupSrc {
    const float* gf;       // [B][N][C]
    const float* xyz;      // source coordinates, element (b, n, d) at b*sb + n*sn + d*sc
    long long sb, sn, sc;
    const float* new_xyz;  // [B*S][3]
    const float* wx;       // coordinate columns of the conv weight: wx[c * ldw + d]
    long long ldw;
    const int32_t* idx;    // [B*S*K] cloud-local
    int N, S, K;
    float k[4][3];         // the thread's four channels
    __device__ __forceinline__ void init(int c, int) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int d = 0; d < 3; ++d) k[j][d] = wx[(long long)(c + j) * ldw + d];
    }
    __device__ __forceinline__ float4 row(int row, int c, int C, int32_t* status) const {
        const int bs = row / K, b = bs / S;
        const int j = checked_index(idx[row], N, status);
        const float* px = xyz + (long long)b * sb + (long long)j * sn;
        const float* pc = new_xyz + 3ll * bs;
        const float dx = __fsub_rn(px[0], pc[0]), dy = __fsub_rn(px[sc], pc[1]), dz = __fsub_rn(px[2 * sc], pc[2]);
        float4 v = *(const float4*)(gf + ((long long)b * N + j) * C + c);
        v.x += __builtin_fmaf(k[0][2], dz, __builtin_fmaf(k[0][1], dy, k[0][0] * dx));
        v.y += __builtin_fmaf(k[1][2], dz, __builtin_fmaf(k[1][1], dy, k[1][0] * dx));
        v.z += __builtin_fmaf(k[2][2], dz, __builtin_fmaf(k[2][1], dy, k[2][0] * dx));
        v.w += __builtin_fmaf(k[3][2], dz, __builtin_fmaf(k[3][1], dy, k[3][0] * dx));
        return v;
    }
};

// Y16: the rows are stored as bfloat16 (bf16 mode with bfloat16 storage; the statistics come from the fp32 registers, as in
// the GEMM epilogue)
template <class SRC, bool Y16 = false, bool STATS = true>   // STATS = false (eval mode): the rows only
__global__ __launch_bounds__(256) void rows_stats_kernel(SRC src, int C, float* __restrict__ y, float* __restrict__ partial,
                                                         long long pchunk, long long pcol, long long pwhich, const SegTable st,
                                                         int32_t* status) {
    __shared__ float red[1024];    // [rows per pass][C]
    __shared__ float smean[256];
    const int tpr = C / 4, rpp = 256 / tpr, ch = IB_PASSES * rpp;
    const int cg = threadIdx.x % tpr, rs = threadIdx.x / tpr, c = 4 * cg;
    src.init(c, C);
    const RowBlock rb = row_block(st, (int)blockIdx.x, 2 * ch);
    for (int h = 0; h < 2; ++h) {
        const int base = rb.row0 + h * ch;
        const int left = rb.row_end - base;
        const int cnt = left < 0 ? 0 : (left < ch ? left : ch);
        float4 v[IB_PASSES];
#pragma unroll
        for (int p = 0; p < IB_PASSES; ++p) {
            const int row = base + p * rpp + rs;
            v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rb.row_end) {
                v[p] = src.row(row, c, C, status);
                if (Y16) {
                    using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
                    *(bf16x4*)((__bf16*)y + (long long)row * C + c) = bf16x4{(__bf16)v[p].x, (__bf16)v[p].y, (__bf16)v[p].z, (__bf16)v[p].w};
                } else {
                    *(float4*)(y + (long long)row * C + c) = v[p];
                }
            }
        }
        if (!STATS) continue;
        float4 s = v[0];
#pragma unroll
        for (int p = 1; p < IB_PASSES; ++p) s.x += v[p].x, s.y += v[p].y, s.z += v[p].z, s.w += v[p].w;
        *(float4*)(red + rs * C + c) = s;
        __syncthreads();
        if (rs == 0) {
            for (int k = 1; k < rpp; ++k) {
                const float4 o = *(const float4*)(red + k * C + c);
                s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
            }
            const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.0f;
            *(float4*)(smean + c) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        }
        __syncthreads();
        const float4 m = *(const float4*)(smean + c);
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < IB_PASSES; ++p) {
            if (base + p * rpp + rs < rb.row_end) {
                const float dx = v[p].x - m.x, dy = v[p].y - m.y, dz = v[p].z - m.z, dw = v[p].w - m.w;
                m2.x += dx * dx, m2.y += dy * dy, m2.z += dz * dz, m2.w += dw * dw;
            }
        }
        *(float4*)(red + rs * C + c) = m2;
        __syncthreads();
        if (rs == 0) {
            for (int k = 1; k < rpp; ++k) {
                const float4 o = *(const float4*)(red + k * C + c);
                m2.x += o.x, m2.y += o.y, m2.z += o.z, m2.w += o.w;
            }
            const long long chunk = 2ll * blockIdx.x + h;
            const float mm[4] = {m.x, m.y, m.z, m.w}, qq[4] = {m2.x, m2.y, m2.z, m2.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* pp = partial + chunk * pchunk + (long long)(c + j) * pcol;
                pp[0] = cnt > 0 ? mm[j] : 0.0f;
                pp[pwhich] = cnt > 0 ? qq[j] : 0.0f;
            }
        }
        __syncthreads();
    }
}

// eval mode of the hoisted layers: the coefficient block from the running statistics (one block serves every row)
int hoisted_eval_coef(const pn2_mlp_layer& L, hipStream_t s) {
    if (!L.running_mean || !L.running_var) return PN2_E_BADARG;
    PN2_LAUNCH("bn_eval_coef", 36.0 * L.cout, 0, bn_eval_coef_kernel, dim3(pn2::ceil_div(L.cout, 256)), dim3(256), s, L.cout, L.gamma,
               L.beta, L.running_mean, L.running_var, L.eps, L.stats);
    PN2_LAUNCH_CHECK();
    return 0;
}

bool interp_bn_args_ok(const void* idx, const void* w, const int32_t* coff, int B, int N, int S, long long rows,
                       const pn2_mlp_layer* L, const pn2_segments* sg) {
    if (!idx || !w || !L || B <= 0 || N <= 0 || S <= 0 || rows <= 0 || rows >= (1ll << 31) / 4 || B > 65535 || S > 8192) return false;
    if (!coff && rows != (long long)B * N) return false;
    if (!(L->cout == 64 || L->cout == 128 || L->cout == 256) || !L->has_bn || !L->y || !L->stats || !aligned16(L->y)) return false;
    if (!segs_valid((int)rows, sg, 1)) return false;
    for (int i = 0; !coff && sg && sg->nseg > 1 && i <= sg->nseg; ++i)
        if (sg->row_off[i] % N) return false;   // whole clouds per segment (ragged clouds: the caller's promise)
    return true;
}

}  // namespace

extern "C" size_t pn2_interp_bn_workspace_bytes(int B, long long rows, int S, int C, int nseg) {
    if (B <= 0 || rows <= 0 || S <= 0 || C <= 0) return 0;
    if (nseg < 1) nseg = 1;
    const size_t part = ((size_t)pn2::ceil_div(rows, 32) + 2 * (size_t)nseg) * 2 * (size_t)C * sizeof(float);
    return slice_region_bytes((int)rows, nseg, (size_t)C) + align256(part) + align256(pn2::interp::grad_workspace_bytes(B, rows, S));
}

extern "C" int pn2_interp_bn_fwd_f32(const float* q, const int32_t* idx, const float* w, const int32_t* coff,
                                     const int32_t* row_cloud, int B, int N, int S, long long nrows, const pn2_mlp_layer* layer,
                                     const pn2_segments* segments, int training, int rows_bf16, int32_t* status, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!q || !aligned16(q) || (coff == nullptr) != (row_cloud == nullptr) ||
        !interp_bn_args_ok(idx, w, coff, B, N, S, nrows, layer, training ? segments : nullptr))
        return PN2_E_BADARG;
    const pn2_mlp_layer& L = *layer;
    const int rows = (int)nrows, C = L.cout;
    const Segs Sg = make_segs(rows, training ? segments : nullptr);
    if (!workspace || workspace_bytes < pn2_interp_bn_workspace_bytes(B, rows, S, C, Sg.nseg)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const FinScratch fs = fin_scratch(workspace, Sg.nseg, (size_t)C);
    float* part = (float*)((char*)workspace + slice_region_bytes(rows, Sg.nseg, (size_t)C));
    const int R = interp_block_rows(C);
    int nblk = 0;
    const SegTable st = make_table(Sg, R, &nblk);
    const long long cm = cm_stride(rows, R, Sg.nseg);
    const InterpSrc src{q, idx, w, row_cloud, N, S};
    if (!training) {   // eval: rows only, coefficients from the running statistics
        if (rows_bf16) return PN2_E_BADARG;
        PN2_LAUNCH("interp_bn_fwd", (double)rows * (36.0 + 4.0 * C) + 4.0 * B * S * C, 0, (rows_stats_kernel<InterpSrc, false, false>),
                   dim3(nblk), dim3(256), s, src, C, L.y, part, 0ll, 0ll, 0ll, st, status);
        PN2_LAUNCH_CHECK();
        return hoisted_eval_coef(L, s);
    }
    if (rows_bf16)
        PN2_LAUNCH("interp_bn_fwd", (double)rows * (36.0 + 2.0 * C) + 4.0 * B * S * C, 0, (rows_stats_kernel<InterpSrc, true>), dim3(nblk),
                   dim3(256), s, src, C, L.y, part, cm ? 2ll : 2ll * C, cm ? cm : 1ll, cm ? 1ll : (long long)C, st, status);
    else
        PN2_LAUNCH("interp_bn_fwd", (double)rows * (36.0 + 4.0 * C) + 4.0 * B * S * C, 0, (rows_stats_kernel<InterpSrc>), dim3(nblk),
                   dim3(256), s, src, C, L.y, part, cm ? 2ll : 2ll * C, cm ? cm : 1ll, cm ? 1ll : (long long)C, st, status);
    PN2_LAUNCH_CHECK();
    return launch_bn_finalize(part, fs, Sg, R, rows, L, s, cm);
}

extern "C" int pn2_interp_bn_bwd_f32(const float* dout, const int32_t* idx, const float* w, const int32_t* coff, int B, int N,
                                     int S, long long nrows, const pn2_mlp_layer* layer, float* dq, const pn2_segments* segments,
                                     int rows_bf16, void* workspace, size_t workspace_bytes, void* stream) {
    // (the scatter finds a row's cloud through its destination: no row -> cloud table on this side)
    if (!dout || !dq || !aligned16(dout) || !interp_bn_args_ok(idx, w, coff, B, N, S, nrows, layer, segments)) return PN2_E_BADARG;
    const pn2_mlp_layer& L = *layer;
    const int rows = (int)nrows, C = L.cout;
    const Segs Sg = make_segs(rows, segments);
    if (!workspace || workspace_bytes < pn2_interp_bn_workspace_bytes(B, rows, S, C, Sg.nseg)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const FinScratch fs = fin_scratch(workspace, Sg.nseg, (size_t)C);
    char* base = (char*)workspace + slice_region_bytes(rows, Sg.nseg, (size_t)C);
    float* part = (float*)base;
    const size_t part_bytes = align256(((size_t)pn2::ceil_div(rows, 32) + 2 * (size_t)Sg.nseg) * 2 * (size_t)C * sizeof(float));
    // BatchNorm-backward sums: left behind by the linked consumer's dgrad epilogue, or reduced here
    const float* partial = part;
    int R = RB, cpb = 1, st_;
    long long cm = 0;
    if (L.out_partial) {
        partial = L.out_partial;
        R = L.out_partial_rows;
        cpb = L.out_partial_cpb;
        if (R <= 0 || cpb != 2) return PN2_E_BADARG;
        cm = cm_stride(rows, R, Sg.nseg);
    } else {
        int nblk = 0;
        const SegTable tb = make_table(Sg, RB, &nblk);
        PN2_LAUNCH("bn_bwd_reduce", 8.0 * rows * C, 0, (bn_bwd_reduce_kernel<true>), dim3(nblk), dim3(256), s, dout, (long long)C,
                   (const float*)L.y, (long long)C, rows, C, (const float*)L.stats, L.relu, part, tb, rows_bf16 ? 1 : 0,
                   rows_bf16 ? 1 : 0);
        PN2_LAUNCH_CHECK();
    }
    if ((st_ = launch_bn_bwd_finalize(partial, fs, Sg, R, cpb, rows, L, s, cm))) return st_;
    int32_t one[2] = {0, rows};
    const pn2::interp::DySource dy{L.y, L.stats, L.relu, Sg.nseg, Sg.nseg > 1 ? Sg.row_off : one, rows_bf16 ? 1 : 0};
    char* tws = base + part_bytes;
    return pn2::interp::grad(dout, C, 0, idx, w, B, N, S, C, dq, tws, workspace_bytes - (size_t)(tws - (char*)workspace), s,
                             (const int*)coff, rows, &dy);
}

// ================================================================== set abstraction, first convolution hoisted
// (pn2_hip.h "Set abstraction with the first convolution HOISTED").  Forward: rows_stats_kernel<GroupSrc>.  Backward: one
// wavefront per group, lanes across the channels (the layout of group.hip's group_grad_groups_kernel): dZ is rebuilt from
// (dA, Z) row by row; the rows that go where row 0 goes -- the padding of a sparse ball -- are summed in registers and leave as
// one atomic per channel, the others as one atomic each; the coordinate weights' gradient sum_k dZ[k][c] * (xyz_k - centre)
// accumulates in registers over all the groups a wavefront walks and leaves as one partial per workgroup.
namespace {

constexpr int GB_BLOCKS = 1024;   // workgroups of the backward scatter (= partials of the coordinate-weight gradient)

__global__ __launch_bounds__(256) void group_dy_scatter_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                               const float* __restrict__ coef_all, int relu,
                                                               const float* __restrict__ xyz, long long sb, long long sn,
                                                               long long sc, const float* __restrict__ new_xyz,
                                                               const int32_t* __restrict__ idx, long long groups, int N, int S,
                                                               int K, int C, float* __restrict__ dgf,
                                                               float* __restrict__ wpart, const SegTable st) {
    __shared__ float red[4][3][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float aw[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) aw[q][0] = aw[q][1] = aw[q][2] = 0.0f;
    for (long long g = (long long)blockIdx.x * 4 + wv; g < groups; g += (long long)gridDim.x * 4) {
        const long long r0 = g * K;
        const int b = (int)(g / S);
        const float* coef = coef_all + (st.nseg > 1 ? (long long)seg_of_row(st, (int)r0) * ST_ROWS * C : 0);
        const int mine = lane < K ? idx[r0 + lane] : -1;
        const bool ok = (unsigned)mine < (unsigned)N;           // (a bad index was flagged by the forward pass)
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;                  // lane k: centred coordinates of row k
        if (ok) {
            const float* px = xyz + (long long)b * sb + (long long)mine * sn;
            const float* pc = new_xyz + 3ll * g;
            dx = __fsub_rn(px[0], pc[0]), dy = __fsub_rn(px[sc], pc[1]), dz = __fsub_rn(px[2 * sc], pc[2]);
        }
        const int j0 = __builtin_amdgcn_readfirstlane(mine);
        const unsigned long long same = __ballot(lane < K && ok && mine == j0);
        const unsigned long long other = __ballot(lane < K && ok && mine != j0);
        float* dst = dgf + (long long)b * N * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (64 * q >= C) break;                              // uniform
            const bool live = 64 * q + lane < C;                 // (C = 32: half a wavefront)
            const int c = live ? 64 * q + lane : C - 1;
            const float km = coef[ST_MEAN * C + c], ks = coef[ST_SCALE * C + c], kb = coef[ST_BETA * C + c];
            const float ka = coef[ST_A * C + c], kq = coef[ST_B * C + c];
            float acc = 0.0f;
            for (int pass = 0; pass < 2; ++pass) {
                unsigned long long m = pass ? other : same;
                while (m) {   // wave-uniform; four rows in flight
                    int k[4];
                    float dv[4], zv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        k[u] = m ? (int)__builtin_ctzll(m) : -1;
                        if (m) m &= m - 1ull;
                        const long long r = r0 + (k[u] >= 0 ? k[u] : 0);
                        dv[u] = dout[r * C + c];
                        zv[u] = z[r * C + c];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (k[u] < 0) break;                     // uniform
                        const float t = __builtin_fmaf(zv[u] - km, ks, kb);
                        const float gm = (!relu || t > 0.0f) ? dv[u] : 0.0f;
                        const float d = live ? ks * (gm - ka - (zv[u] - km) * kq) : 0.0f;
                        aw[q][0] += d * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dx), k[u]));
                        aw[q][1] += d * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dy), k[u]));
                        aw[q][2] += d * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dz), k[u]));
                        if (pass == 0)
                            acc += d;
                        else if (live)
                            atomicAdd(dst + (long long)__builtin_amdgcn_readlane(mine, k[u]) * C + c, d);
                    }
                }
                if (pass == 0 && same && live) atomicAdd(dst + (long long)j0 * C + c, acc);
            }
        }
    }
    // one partial of the coordinate-weight gradient per workgroup: [3][C]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) red[wv][d][64 * q + lane] = aw[q][d];
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * C; e += 256) {
        const int d = e / C, c = e - d * C;
        wpart[(long long)blockIdx.x * 3 * C + e] = (red[0][d][c] + red[1][d][c]) + (red[2][d][c] + red[3][d][c]);
    }
}

// dwx[c * ld + d] = sum over the scatter's workgroups, fixed order: 16 wavefronts (lane = element) take every 16th partial each,
// four loads in flight, and are joined in order (one thread per element walking all 1024 partials was 256 dependent rounds: 77 us)
constexpr int GWX_W = 16;
__global__ __launch_bounds__(64 * GWX_W) void group_wx_reduce_kernel(const float* __restrict__ wpart, int nblk, int C,
                                                                     float* __restrict__ dwx, long long ld) {
    __shared__ float red[GWX_W][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool ok = e < 3 * C;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (ok) {
        int k = w;
        for (; k + 3 * GWX_W < nblk; k += 4 * GWX_W) {
            a0 += wpart[(long long)k * 3 * C + e];
            a1 += wpart[(long long)(k + GWX_W) * 3 * C + e];
            a2 += wpart[(long long)(k + 2 * GWX_W) * 3 * C + e];
            a3 += wpart[(long long)(k + 3 * GWX_W) * 3 * C + e];
        }
        for (; k < nblk; k += GWX_W) a0 += wpart[(long long)k * 3 * C + e];
    }
    red[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && ok) {
        float t = 0.0f;
        for (int i = 0; i < GWX_W; ++i) t += red[i][lane];
        const int d = e / C, c = e - d * C;
        dwx[(long long)c * ld + d] = t;
    }
}

bool group_bn_args_ok(const void* xyz, const void* new_xyz, const void* idx, int B, int N, int S, int K, const pn2_mlp_layer* L,
                      const pn2_segments* sg) {
    if (!xyz || !new_xyz || !idx || !L || B <= 0 || N <= 0 || S <= 0 || K <= 0 || K > 64) return false;
    const long long rows = (long long)B * S * K;
    if (rows >= (1ll << 31) / 4) return false;
    if (!(L->cout == 32 || L->cout == 64 || L->cout == 128 || L->cout == 256) || !L->has_bn || !L->y || !L->stats || !aligned16(L->y))
        return false;
    if (!segs_valid((int)rows, sg, K)) return false;
    for (int i = 0; sg && sg->nseg > 1 && i <= sg->nseg; ++i)
        if (sg->row_off[i] % (S * K)) return false;   // whole clouds per segment
    return true;
}

size_t group_part_bytes(long long rows, int nseg, int C) {
    return align256(((size_t)pn2::ceil_div(rows, 32) + 2 * (size_t)nseg) * 2 * (size_t)C * sizeof(float));
}

}  // namespace

extern "C" size_t pn2_group_bn_workspace_bytes(int B, int S, int K, int C, int nseg) {
    if (B <= 0 || S <= 0 || K <= 0 || C <= 0) return 0;
    if (nseg < 1) nseg = 1;
    const long long rows = (long long)B * S * K;
    return slice_region_bytes((int)rows, nseg, (size_t)C) + group_part_bytes(rows, nseg, C) +
           align256((size_t)GB_BLOCKS * 3 * C * sizeof(float));
}

extern "C" int pn2_group_bn_fwd_f32(const float* gf, const float* xyz, int64_t sb, int64_t sn, int64_t sc, const float* new_xyz,
                                    const int32_t* idx, const float* wx, int64_t ldw, int B, int N, int S, int K,
                                    const pn2_mlp_layer* layer, const pn2_segments* segments, int training, int32_t* status,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!gf || !aligned16(gf) || !wx || !group_bn_args_ok(xyz, new_xyz, idx, B, N, S, K, layer, training ? segments : nullptr))
        return PN2_E_BADARG;
    const pn2_mlp_layer& L = *layer;
    const int rows = B * S * K, C = L.cout;
    const Segs Sg = make_segs(rows, training ? segments : nullptr);
    if (!workspace || workspace_bytes < pn2_group_bn_workspace_bytes(B, S, K, C, Sg.nseg)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const FinScratch fs = fin_scratch(workspace, Sg.nseg, (size_t)C);
    float* part = (float*)((char*)workspace + slice_region_bytes(rows, Sg.nseg, (size_t)C));
    const int R = interp_block_rows(C);
    int nblk = 0;
    const SegTable st = make_table(Sg, R, &nblk);
    const long long cm = cm_stride(rows, R, Sg.nseg);
    GroupSrc src{};
    src.gf = gf, src.xyz = xyz, src.sb = sb, src.sn = sn, src.sc = sc, src.new_xyz = new_xyz, src.wx = wx, src.ldw = ldw;
    src.idx = idx, src.N = N, src.S = S, src.K = K;
    if (!training) {
        PN2_LAUNCH("group_bn_fwd", (double)rows * (8.0 * C + 20.0), 6.0 * rows * C, (rows_stats_kernel<GroupSrc, false, false>), dim3(nblk),
                   dim3(256), s, src, C, L.y, part, 0ll, 0ll, 0ll, st, status);
        PN2_LAUNCH_CHECK();
        return hoisted_eval_coef(L, s);
    }
    PN2_LAUNCH("group_bn_fwd", (double)rows * (8.0 * C + 20.0), 6.0 * rows * C, (rows_stats_kernel<GroupSrc>), dim3(nblk), dim3(256), s,
               src, C, L.y, part, cm ? 2ll : 2ll * C, cm ? cm : 1ll, cm ? 1ll : (long long)C, st, status);
    PN2_LAUNCH_CHECK();
    return launch_bn_finalize(part, fs, Sg, R, rows, L, s, cm);
}

extern "C" int pn2_group_bn_bwd_f32(const float* dout, const float* xyz, int64_t sb, int64_t sn, int64_t sc, const float* new_xyz,
                                    const int32_t* idx, int B, int N, int S, int K, const pn2_mlp_layer* layer, float* dgf,
                                    float* dwx, int64_t lddw, const pn2_segments* segments, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!dout || !dgf || !dwx || !aligned16(dout) || !group_bn_args_ok(xyz, new_xyz, idx, B, N, S, K, layer, segments))
        return PN2_E_BADARG;
    const pn2_mlp_layer& L = *layer;
    const int rows = B * S * K, C = L.cout;
    const Segs Sg = make_segs(rows, segments);
    if (!workspace || workspace_bytes < pn2_group_bn_workspace_bytes(B, S, K, C, Sg.nseg)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const FinScratch fs = fin_scratch(workspace, Sg.nseg, (size_t)C);
    char* base = (char*)workspace + slice_region_bytes(rows, Sg.nseg, (size_t)C);
    float* part = (float*)base;
    float* wpart = (float*)(base + group_part_bytes(rows, Sg.nseg, C));
    const float* partial = part;
    int R = RB, cpb = 1, st_;
    long long cm = 0;
    if (L.out_partial) {   // BatchNorm-backward sums left behind by the linked consumer's dgrad epilogue
        partial = L.out_partial;
        R = L.out_partial_rows;
        cpb = L.out_partial_cpb;
        if (R <= 0 || cpb != 2) return PN2_E_BADARG;
        cm = cm_stride(rows, R, Sg.nseg);
    } else {
        int nblk = 0;
        const SegTable tb = make_table(Sg, RB, &nblk);
        PN2_LAUNCH("bn_bwd_reduce", 8.0 * rows * C, 0, (bn_bwd_reduce_kernel<true>), dim3(nblk), dim3(256), s, dout, (long long)C,
                   (const float*)L.y, (long long)C, rows, C, (const float*)L.stats, L.relu, part, tb, 0, 0);
        PN2_LAUNCH_CHECK();
    }
    if ((st_ = launch_bn_bwd_finalize(partial, fs, Sg, R, cpb, rows, L, s, cm))) return st_;
    PN2_HIP_CHECK(hipMemsetAsync(dgf, 0, (size_t)B * N * C * sizeof(float), s));
    const long long groups = (long long)B * S;
    int nb = (int)((groups + 3) / 4);
    if (nb > GB_BLOCKS) nb = GB_BLOCKS;
    int dummy = 0;
    const SegTable st = make_table(Sg, RB, &dummy);   // (row offsets only: seg_of_row)
    PN2_LAUNCH("group_bn_bwd", (double)rows * (8.0 * C + 20.0) + 4.0 * B * N * C, 8.0 * rows * C, group_dy_scatter_kernel, dim3(nb),
               dim3(256), s, dout, (const float*)L.y, (const float*)L.stats, L.relu, xyz, (long long)sb, (long long)sn, (long long)sc,
               new_xyz, idx, groups, N, S, K, C, dgf, wpart, st);
    PN2_LAUNCH("group_bn_wx", 12.0 * nb * C, 0, group_wx_reduce_kernel, dim3(pn2::ceil_div(3 * C, 64)), dim3(64 * GWX_W), s,
               (const float*)wpart, nb, C, dwx, (long long)lddw);
    PN2_LAUNCH_CHECK();
    return 0;
}

#ifdef PN2_GEMM_DIAG
// diagnostic build only: copy the stamp table to the host (synchronises the device)
extern "C" int pn2_gemm_diag_read(unsigned long long* host, int clear) {
    PN2_HIP_CHECK(hipDeviceSynchronize());
    PN2_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_diag), sizeof(unsigned long long) * 16384 * 8));
    if (clear) {
        static unsigned long long zeros[16384 * 8];
        PN2_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_diag), zeros, sizeof(zeros)));
    }
    return 0;
}
#endif

// ================================================================== the two heads' first-layer dgrad as one contraction
// (pn2_hip.h pn2_mlp_pair_dgrad_f32)
extern "C" int pn2_mlp_pair_dgrad_f32(int rows, const pn2_mlp_layer* la, const float* dza, const pn2_mlp_layer* lb, const float* dzb,
                                      const float* x, int64_t ldx, float* dx, int64_t lddx, const pn2_segments* segments,
                                      int precision, void* stream) {
    const int x16 = (precision & PN2_CHAIN_X_BF16) ? 1 : 0, s16 = (precision & PN2_CHAIN_STORE_BF16) ? 1 : 0;
    const int dx16 = (precision & PN2_CHAIN_DX_BF16) ? 1 : 0;
    precision &= ~(PN2_CHAIN_X_BF16 | PN2_CHAIN_STORE_BF16 | PN2_CHAIN_DX_BF16);
    if (!la || !lb || !dza || !dzb || !dx || rows <= 0) return PN2_E_BADARG;
    if (la->cin != lb->cin || la->cout != lb->cout || !la->has_bn || !lb->has_bn || la->relu != lb->relu || !la->y || !lb->y ||
        !la->stats || !lb->stats || !la->weight || !lb->weight || la->cin % 4 || la->cout % BK)
        return PN2_E_BADARG;
    if (precision != PN2_PRECISION_F32 && precision != PN2_PRECISION_BF16) return PN2_E_BADARG;
    if ((x16 || s16 || dx16) && (precision != PN2_PRECISION_BF16 || !s16)) return PN2_E_BADARG;
    if (pick_tile(rows, la->cin, 1) != 128) return PN2_E_BADARG;      // the big-tile kernels only
    if (!segs_valid(rows, segments, 1)) return PN2_E_BADARG;
    const Segs S = make_segs(rows, segments);
    const int cin = la->cin, cout = la->cout;
    GemmArgs g{};
    g.precision = precision;
    g.A = plain(dza, cout, rows, cout);
    g.A.p16 = s16, g.A.q = la->y, g.A.q16 = s16, g.A.ldq = cout, g.A.coef = la->stats, g.A.cstride = cout, g.A.relu = la->relu;
    g.A2p = dzb, g.A2q = lb->y, g.A2coef = lb->stats;
    g.B = plain(la->weight, cin, cout, cin);
    g.B2p = lb->weight;
    g.ksplit = cout;
    g.M = rows, g.N = cin, g.K = 2 * cout;
    g.C = dx, g.ldc = lddx, g.c16 = dx16, g.ey16 = x16;
    if (la->in_stats && la->in_partial) {   // linked heads: the producing layer's BatchNorm-backward sums from the epilogue
        if (!x) return PN2_E_BADARG;
        g.partial = la->in_partial;
        g.ey = x, g.ldey = ldx, g.ecoef = la->in_stats, g.erelu = la->in_relu;
        g.pstride = cm_stride(rows, 128, S.nseg);
    }
    if (!(vec_ok(g.A) && vec_ok(g.B) && aligned16(dzb) && aligned16(lb->y) && aligned16(lb->weight))) return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int st16 = (g.A.p16 ? 1 : 0) | (g.c16 ? 4 : 0) | (g.ey16 ? 8 : 0);
    if (st16) {
        if (g.partial && !g.ey16) return PN2_E_BADARG;
        switch (st16 | 8) {
            case 9: return launch_gemm_tv<true, TR_DY, false, TR_PLAIN, EPI_STORE, 128, true, 1, true, NT, 9, true>(g, S, s, nullptr);
            case 13: return launch_gemm_tv<true, TR_DY, false, TR_PLAIN, EPI_STORE, 128, true, 1, true, NT, 13, true>(g, S, s, nullptr);
            default: return PN2_E_BADARG;
        }
    }
    if (precision == PN2_PRECISION_BF16)
        return launch_gemm_tv<true, TR_DY, false, TR_PLAIN, EPI_STORE, 128, true, 1, true, NT, 0, true>(g, S, s, nullptr);
    return launch_gemm_tv<true, TR_DY, false, TR_PLAIN, EPI_STORE, 128, true, 1, false, NT, 0, true>(g, S, s, nullptr);
}
