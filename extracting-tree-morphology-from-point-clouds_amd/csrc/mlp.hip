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
#include "pn2_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LD = 132, NT = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { TR_PLAIN = 0, TR_BNRELU = 1, TR_DY = 2 };
enum { EPI_FWD = 0, EPI_STORE = 1, EPI_SLAB = 2 };
// rows of a layer's coefficient block (each `C` floats): see pn2_mlp_layer.stats in pn2_hip.h
enum { ST_MEAN = 0, ST_VAR = 1, ST_INVSTD = 2, ST_SCALE = 3, ST_BETA = 4, ST_A = 5, ST_B = 6, ST_ROWS = 8 };

struct Operand {
    const float* p;   // X, Y or dZ: [rows][cols], cols (= channels) contiguous
    const float* q;   // TR_DY: the layer's pre-BN output Y
    long long ld, ldq;
    int rows, cols;
    const float* coef;  // coefficient block [ST_ROWS][cstride] of the BatchNorm involved, or null
    int cstride;
    int relu;
};

template <int KIND>
__device__ __forceinline__ float xform(const Operand& o, float v, float y, int ch) {
    if (KIND == TR_PLAIN) return v;
    const float mean = o.coef[ST_MEAN * o.cstride + ch];
    const float scale = o.coef[ST_SCALE * o.cstride + ch];
    const float beta = o.coef[ST_BETA * o.cstride + ch];
    if (KIND == TR_BNRELU) {
        const float t = __builtin_fmaf(v - mean, scale, beta);
        return o.relu ? fmaxf(t, 0.0f) : t;
    }
    // TR_DY
    const float t = __builtin_fmaf(y - mean, scale, beta);
    const float dz = (!o.relu || t > 0.0f) ? v : 0.0f;
    const float a = o.coef[ST_A * o.cstride + ch], b = o.coef[ST_B * o.cstride + ch];
    return scale * (dz - a - (y - mean) * b);
}

__device__ __forceinline__ float4 load4(const float* base, long long ld, int r, int c, int nrows, int ncols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= nrows || c >= ncols) return v;
    const float* p = base + (long long)r * ld + c;
    if (vec && c + 3 < ncols) return *(const float4*)p;
    v.x = p[0];
    if (c + 1 < ncols) v.y = p[1];
    if (c + 2 < ncols) v.z = p[2];
    if (c + 3 < ncols) v.w = p[3];
    return v;
}

// One operand's staging state: 4 float4 (+4 for the second source) per thread per K-tile.
template <bool T_LAYOUT, int KIND>
struct Stager {
    float4 v[4], y[4];
    // T layout: global [outer][k]; thread -> (kq = t & 7, o_sub = t >> 3), pass p: outer = o0 + 32 p + o_sub
    // D layout: global [k][outer]; thread -> (oq = t & 31, k_sub = t >> 5), pass p: k = k0 + 8 p + k_sub
    __device__ __forceinline__ void fetch(const Operand& o, int o0, int k0, bool vec) {
        const int t = threadIdx.x;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int r, c;
            if (T_LAYOUT) {
                r = o0 + 32 * p + (t >> 3);
                c = k0 + 4 * (t & 7);
            } else {
                r = k0 + 8 * p + (t >> 5);
                c = o0 + 4 * (t & 31);
            }
            v[p] = load4(o.p, o.ld, r, c, o.rows, o.cols, vec);
            if (KIND == TR_DY) y[p] = load4(o.q, o.ldq, r, c, o.rows, o.cols, vec);
        }
    }
    __device__ __forceinline__ void commit(const Operand& o, float* S, int o0, int k0) {
        const int t = threadIdx.x;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int r, c;
            if (T_LAYOUT) {
                r = o0 + 32 * p + (t >> 3);
                c = k0 + 4 * (t & 7);
            } else {
                r = k0 + 8 * p + (t >> 5);
                c = o0 + 4 * (t & 31);
            }
            float e[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
            if (KIND != TR_PLAIN) {
                const float yy[4] = {y[p].x, y[p].y, y[p].z, y[p].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // elements outside the matrix must stay exactly zero (they pad the contraction)
                    const bool in = r < o.rows && c + j < o.cols;
                    e[j] = in ? xform<KIND>(o, e[j], KIND == TR_DY ? yy[j] : 0.0f, c + j) : 0.0f;
                }
            }
            if (T_LAYOUT) {
                const int m = 32 * p + (t >> 3), k = 4 * (t & 7);
#pragma unroll
                for (int j = 0; j < 4; ++j) S[(k + j) * LD + m] = e[j];
            } else {
                const int k = 8 * p + (t >> 5), m = 4 * (t & 31);
                *(float4*)(S + k * LD + m) = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
};

struct GemmArgs {
    Operand A, B;
    int M, N, K;          // C[M][N] = sum_k A[m][k] B[k][n]
    float* C;             // EPI_FWD: Y [M][ldc]; EPI_STORE: [M][ldc]; EPI_SLAB: [split][M][ldc]
    long long ldc;
    const float* bias;    // EPI_FWD, may be null
    float* partial;       // EPI_FWD + stats: [ceil(M/64)][2][N] (mean, M2) per 64-row chunk, or null
    int k_per_split;      // EPI_SLAB: K range per blockIdx.z
    int vecA, vecB;
};

template <bool A_T, int A_KIND, bool B_T, int B_KIND, int EPI>
__global__ __launch_bounds__(NT) void gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LD];

    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    int k_begin = 0, k_end = g.K;
    if (EPI == EPI_SLAB) {
        k_begin = blockIdx.z * g.k_per_split;
        k_end = k_begin + g.k_per_split < g.K ? k_begin + g.k_per_split : g.K;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    Stager<A_T, A_KIND> sa;
    Stager<B_T, B_KIND> sb;
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        sa.fetch(g.A, m0, k_begin, g.vecA);
        sb.fetch(g.B, n0, k_begin, g.vecB);
        sa.commit(g.A, As[0], m0, k_begin);
        sb.commit(g.B, Bs[0], n0, k_begin);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int knext = k_begin + (kt + 1) * BK;
        if (kt + 1 < nk) {
            sa.fetch(g.A, m0, knext, g.vecA);
            sb.fetch(g.B, n0, knext, g.vecB);
        }
        const float* a = As[cur] + wm * 64 + l31;
        const float* b = Bs[cur] + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int ko = (kk + half) * LD;
            const float a0 = a[ko], a1 = a[ko + 32], b0 = b[ko], b1 = b[ko + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.commit(g.A, As[cur ^ 1], m0, knext);
            sb.commit(g.B, Bs[cur ^ 1], n0, knext);
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* C = g.C;
    if (EPI == EPI_SLAB) C += (long long)blockIdx.z * g.M * g.ldc;
    const int rbase = m0 + wm * 64 + 4 * half;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + 32 * j + l31;
        const bool cok = col < g.N;
        const float bias = (EPI == EPI_FWD && g.bias && cok) ? g.bias[col] : 0.0f;
        float sum = 0.0f;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2);
                const float val = acc[i][j][r] + bias;
                acc[i][j][r] = val;
                if (row < g.M) {
                    if (cok) C[(long long)row * g.ldc + col] = val;
                    sum += val;
                    ++cnt;
                }
            }
        if (EPI == EPI_FWD && g.partial) {
            // per-(64-row chunk, column) mean and M2; the other half-wave holds the other 32 rows of the column
            sum += __shfl_xor(sum, 32, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mean = cnt > 0 ? sum / (float)cnt : 0.0f;
            float m2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2);
                    const float d = acc[i][j][r] - mean;
                    if (row < g.M) m2 += d * d;
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (half == 0 && cok && m0 + wm * 64 < g.M) {
                const long long chunk = (m0 + wm * 64) / 64;
                g.partial[(chunk * 2 + 0) * g.N + col] = mean;
                g.partial[(chunk * 2 + 1) * g.N + col] = m2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- BatchNorm: forward
// One block per channel: merge the 64-row (mean, M2) partials in float64, write the coefficient block and
// update the running statistics exactly like nn.BatchNorm (biased variance for normalisation, unbiased for
// running_var).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float eps, float momentum, float* __restrict__ coef) {
    __shared__ double red[256];
    __shared__ double s_mean;
    const int c = blockIdx.x, t = threadIdx.x;
    const int nchunk = (rows + 63) / 64;
    double acc = 0.0;
    for (int k = t; k < nchunk; k += 256) {
        const int n = rows - k * 64 < 64 ? rows - k * 64 : 64;
        acc += (double)n * (double)partial[((long long)k * 2) * C + c];
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
    for (int k = t; k < nchunk; k += 256) {
        const int n = rows - k * 64 < 64 ? rows - k * 64 : 64;
        const double d = (double)partial[((long long)k * 2) * C + c] - mean;
        acc += (double)partial[((long long)k * 2 + 1) * C + c] + (double)n * d * d;
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
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ y, long long total4, int C,
                                                            const float* __restrict__ coef, int relu, float* __restrict__ z) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int c = (int)((e * 4) % C);
        const float4 v = ((const float4*)y)[e];
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
                                                              float* __restrict__ out, int32_t* __restrict__ arg) {
    const long long total = groups * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long gi = e / C;
        const int c = (int)(e - gi * C);
        const float mean = coef[ST_MEAN * C + c], sc = coef[ST_SCALE * C + c], bt = coef[ST_BETA * C + c];
        const float* p = y + gi * K * C + c;
        float best = -__builtin_inff();
        int bk = 0;
        for (int k = 0; k < K; ++k) {
            float v = __builtin_fmaf(p[(long long)k * C] - mean, sc, bt);
            if (relu) v = fmaxf(v, 0.f);
            if (v > best) {
                best = v;
                bk = k;
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

// ------------------------------------------------------------------------------------------ BatchNorm: backward
// partial[blk][0][c] = sum_r dzhat, partial[blk][1][c] = sum_r dzhat * xhat over the block's 256 rows
constexpr int RB = 256;
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, long long lddz,
                                                            const float* __restrict__ y, long long ldy, int rows, int C,
                                                            const float* __restrict__ coef, int relu,
                                                            float* __restrict__ partial) {
    __shared__ float red[2][256];
    const int r0 = blockIdx.x * RB;
    const int r1 = r0 + RB < rows ? r0 + RB : rows;
    // threads along channels: groups of `cw` columns, `rg` row groups
    const int cw = C < 256 ? (C <= 32 ? 32 : C <= 64 ? 64 : C <= 128 ? 128 : 256) : 256;
    const int rg = 256 / cw;
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + tc;
        float s1 = 0.f, s2 = 0.f;
        if (c < C) {
            const float mean = coef[ST_MEAN * C + c], sc = coef[ST_SCALE * C + c], bt = coef[ST_BETA * C + c];
            const float invstd = coef[ST_INVSTD * C + c];
            for (int r = r0 + tr; r < r1; r += rg) {
                const float yy = y[(long long)r * ldy + c];
                const float t = __builtin_fmaf(yy - mean, sc, bt);
                const float g = (!relu || t > 0.f) ? dz[(long long)r * lddz + c] : 0.f;
                s1 += g;
                s2 += g * ((yy - mean) * invstd);
            }
        }
        red[0][threadIdx.x] = s1;
        red[1][threadIdx.x] = s2;
        __syncthreads();
        if (tr == 0 && c < C) {
            for (int k = 1; k < rg; ++k) {
                s1 += red[0][k * cw + tc];
                s2 += red[1][k * cw + tc];
            }
            partial[((long long)blockIdx.x * 2 + 0) * C + c] = s1;
            partial[((long long)blockIdx.x * 2 + 1) * C + c] = s2;
        }
        __syncthreads();
    }
}

// one block per channel: s1, s2 in float64 -> dgamma, dbeta and the dY coefficients (a = s1/R, b = invstd*s2/R)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int rows, int C,
                                                              float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int k = t; k < nblk; k += 256) {
        a1 += (double)partial[((long long)k * 2) * C + c];
        a2 += (double)partial[((long long)k * 2 + 1) * C + c];
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

// dW[m][n] (+)= sum over splits of slab[s][m][n], fixed order; optional column sums for a bias without BatchNorm
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long long mn,
                                                          float* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < mn; e += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slab[(long long)k * mn + e];
        out[e] += s;
    }
}

// partial[blk][c] = sum over the block's rows of dz[r][c]  (bias of a layer WITHOUT BatchNorm, e.g. the last
// conv of a head); summed in fixed order by colsum_finalize_kernel -> deterministic
constexpr int CS_ROWS = 1024;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dz, long long ld, int rows, int C,
                                                     float* __restrict__ partial) {
    __shared__ float red[256];
    const int r0 = blockIdx.x * CS_ROWS;
    const int r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    for (int c = 0; c < C; ++c) {
        float a = 0.f;
        for (int r = r0 + threadIdx.x; r < r1; r += 256) a += dz[(long long)r * ld + c];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(long long)blockIdx.x * C + c] = red[0];
        __syncthreads();
    }
}

__global__ void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0;
    for (int k = 0; k < nblk; ++k) a += (double)partial[(long long)k * C + c];
    out[c] += (float)a;
}

inline unsigned grid1d(long long total, int per_block = 256) {
    long long g = (total + per_block - 1) / per_block;
    return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// vector (16 B) staging is legal when rows start 16-byte aligned and the contiguous extent is a multiple of 4
inline int vec_ok(const Operand& o) {
    return (o.ld % 4 == 0) && (o.cols % 4 == 0) && aligned16(o.p) && (!o.q || (o.ldq % 4 == 0 && aligned16(o.q)));
}

Operand plain(const float* p, long long ld, int rows, int cols) {
    Operand o{};
    o.p = p;
    o.ld = ld;
    o.rows = rows;
    o.cols = cols;
    return o;
}

// activation source of a layer: raw rows (first layer) or the previous layer's Y seen through its BN+ReLU
struct Act {
    const float* p;
    long long ld;
    const float* coef;  // null -> plain
    int relu;
};

Operand act_operand(const Act& a, int rows, int cols) {
    Operand o = plain(a.p, a.ld, rows, cols);
    o.coef = a.coef;
    o.cstride = cols;
    o.relu = a.relu;
    return o;
}

template <bool A_T, int A_KIND, bool B_T, int B_KIND, int EPI>
int launch_gemm(GemmArgs& g, int nsplit, hipStream_t s) {
    g.vecA = vec_ok(g.A);
    g.vecB = vec_ok(g.B);
    dim3 grid(pn2::ceil_div(g.M, BM), pn2::ceil_div(g.N, BN), nsplit);
    const char* name = EPI == EPI_FWD ? "gemm_fwd" : EPI == EPI_STORE ? "gemm_dgrad" : "gemm_wgrad";
    const double mk = (double)g.M * g.K * (A_KIND == TR_DY ? 2 : 1), kn = (double)g.K * g.N * (B_KIND == TR_DY ? 2 : 1);
    const double bytes = 4.0 * (mk + kn + (double)g.M * g.N * (EPI == EPI_SLAB ? nsplit : 1));
    PN2_LAUNCH(name, bytes, 2.0 * g.M * g.N * g.K, (gemm_kernel<A_T, A_KIND, B_T, B_KIND, EPI>), grid, dim3(NT), s, g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// ===================================================================================================== C ABI
extern "C" size_t pn2_mlp_workspace_bytes(int rows, const pn2_mlp_layer* layers, int nlayers) {
    if (rows <= 0 || !layers || nlayers <= 0) return 0;
    size_t need = 0;
    for (int i = 0; i < nlayers; ++i) {
        const size_t cin = layers[i].cin, cout = layers[i].cout;
        const size_t fwd = (size_t)pn2::ceil_div(rows, 64) * 2 * cout * sizeof(float);      // stats partials
        size_t bwd_red = (size_t)pn2::ceil_div(rows, RB) * 2 * cout * sizeof(float);         // s1/s2 partials
        const size_t cs = (size_t)pn2::ceil_div(rows, CS_ROWS) * cout * sizeof(float);         // bias column sums
        bwd_red = bwd_red > cs ? bwd_red : cs;
        // wgrad slabs: at most 256 row splits of one 128x128 tile grid
        const size_t tiles = (size_t)pn2::ceil_div((int)cout, BM) * pn2::ceil_div((int)cin, BN);
        size_t splits = 256 / tiles;
        if (splits < 1) splits = 1;
        const size_t maxs = (size_t)pn2::ceil_div(rows, BK);
        if (splits > maxs) splits = maxs;
        const size_t slab = splits * cout * cin * sizeof(float);
        size_t m = fwd > bwd_red ? fwd : bwd_red;
        m = m > slab ? m : slab;
        need = need > m ? need : m;
    }
    return align256(need) + 256;
}

extern "C" int pn2_mlp_chain_fwd_f32(const float* x, int64_t ldx, int rows, const pn2_mlp_layer* layers, int nlayers,
                                     int training, int pool_k, float* out, int32_t* pool_arg, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!x || !layers || nlayers <= 0 || rows <= 0 || !out || (pool_k > 1 && (!pool_arg || rows % pool_k))) return PN2_E_BADARG;
    if (workspace_bytes < pn2_mlp_workspace_bytes(rows, layers, nlayers) || !workspace) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Act in{x, ldx, nullptr, 0};
    for (int i = 0; i < nlayers; ++i) {
        const pn2_mlp_layer& L = layers[i];
        if (!L.weight || L.cin <= 0 || L.cout <= 0) return PN2_E_BADARG;
        const bool last = i == nlayers - 1;
        const bool direct_out = last && !L.has_bn && pool_k <= 1;   // a bare conv at the end writes `out` itself
        float* y = direct_out ? out : L.y;
        if (!y || (L.has_bn && !L.stats)) return PN2_E_BADARG;
        GemmArgs g{};
        g.A = act_operand(in, rows, L.cin);
        g.B = plain(L.weight, L.cin, L.cout, L.cin);
        g.M = rows;
        g.N = L.cout;
        g.K = L.cin;
        g.C = y;
        g.ldc = L.cout;
        g.bias = L.bias;
        g.partial = (L.has_bn && training) ? (float*)workspace : nullptr;
        int st = in.coef ? launch_gemm<true, TR_BNRELU, true, TR_PLAIN, EPI_FWD>(g, 1, s)
                         : launch_gemm<true, TR_PLAIN, true, TR_PLAIN, EPI_FWD>(g, 1, s);
        if (st) return st;
        if (L.has_bn) {
            if (training) {
                PN2_LAUNCH("bn_finalize", 8.0 * pn2::ceil_div(rows, 64) * L.cout, 0, bn_finalize_kernel, dim3(L.cout), dim3(256), s,
                           (const float*)workspace, rows, L.cout, L.gamma, L.beta, L.running_mean, L.running_var, L.eps,
                           L.momentum, L.stats);
            } else {
                if (!L.running_mean || !L.running_var) return PN2_E_BADARG;
                PN2_LAUNCH("bn_eval_coef", 36.0 * L.cout, 0, bn_eval_coef_kernel, dim3(pn2::ceil_div(L.cout, 256)), dim3(256), s,
                           L.cout, L.gamma, L.beta, L.running_mean, L.running_var, L.eps, L.stats);
            }
            PN2_LAUNCH_CHECK();
            in = Act{y, L.cout, L.stats, L.relu};
        } else {
            if (L.relu && !last) return PN2_E_BADARG;  // ReLU without BatchNorm only exists fused into a BN layer here
            in = Act{y, L.cout, nullptr, 0};
        }
        if (last && !direct_out) {
            const int C = L.cout;
            if (!L.has_bn || C % 4) return PN2_E_BADARG;
            if (pool_k > 1) {
                const long long groups = rows / pool_k;
                PN2_LAUNCH("bn_relu_maxpool", 4.0 * rows * C + 8.0 * groups * C, 0, bn_relu_maxpool_kernel,
                           dim3(grid1d(groups * C)), dim3(256), s, (const float*)y, groups, pool_k, C, (const float*)L.stats,
                           L.relu, out, pool_arg);
            } else {
                PN2_LAUNCH("bn_relu_apply", 8.0 * rows * C, 0, bn_relu_apply_kernel, dim3(grid1d((long long)rows * C / 4)),
                           dim3(256), s, (const float*)y, (long long)rows * C / 4, C, (const float*)L.stats, L.relu, out);
            }
            PN2_LAUNCH_CHECK();
        }
    }
    return 0;
}

extern "C" int pn2_mlp_chain_bwd_f32(const float* x, int64_t ldx, int rows, const pn2_mlp_layer* layers, int nlayers,
                                     int pool_k, const float* dout, const int32_t* pool_arg, float* dx, int64_t lddx,
                                     float* scratch_a, float* scratch_b, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!x || !layers || nlayers <= 0 || rows <= 0 || !dout || !scratch_a || !scratch_b) return PN2_E_BADARG;
    if (workspace_bytes < pn2_mlp_workspace_bytes(rows, layers, nlayers) || !workspace) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    // dz of the last layer: upstream gradient, or the max-pool scatter of it
    const float* dz = dout;
    long long lddz = layers[nlayers - 1].cout;
    float* bufs[2] = {scratch_a, scratch_b};
    int which = 0;
    if (pool_k > 1) {
        if (!pool_arg) return PN2_E_BADARG;
        const int C = layers[nlayers - 1].cout;
        PN2_LAUNCH("maxpool_scatter", 4.0 * rows * C + 8.0 * (rows / pool_k) * C, 0, maxpool_scatter_kernel,
                   dim3(grid1d((long long)rows * C)), dim3(256), s, dout, pool_arg, (long long)(rows / pool_k), pool_k, C,
                   bufs[which]);
        PN2_LAUNCH_CHECK();
        dz = bufs[which];
        which ^= 1;
    }
    for (int i = nlayers - 1; i >= 0; --i) {
        const pn2_mlp_layer& L = layers[i];
        const bool last = i == nlayers - 1;
        const float* y = (last && !L.has_bn && pool_k <= 1) ? nullptr : L.y;
        // ---- BatchNorm backward reductions -> coefficients a, b and dgamma, dbeta
        if (L.has_bn) {
            const int nblk = pn2::ceil_div(rows, RB);
            PN2_LAUNCH("bn_bwd_reduce", 8.0 * rows * L.cout, 0, bn_bwd_reduce_kernel, dim3(nblk), dim3(256), s, dz, lddz, y,
                       (long long)L.cout, rows, L.cout, (const float*)L.stats, L.relu, ws);
            PN2_LAUNCH("bn_bwd_finalize", 8.0 * nblk * L.cout, 0, bn_bwd_finalize_kernel, dim3(L.cout), dim3(256), s,
                       (const float*)ws, nblk, rows, L.cout, L.stats, L.dgamma, L.dbeta);
            PN2_LAUNCH_CHECK();
        } else if (L.dbias) {
            const int nblk = pn2::ceil_div(rows, CS_ROWS);
            PN2_LAUNCH("colsum", 4.0 * rows * L.cout, 0, colsum_kernel, dim3(nblk), dim3(256), s, dz, lddz, rows, L.cout, ws);
            PN2_LAUNCH("colsum_finalize", 4.0 * nblk * L.cout, 0, colsum_finalize_kernel, dim3(pn2::ceil_div(L.cout, 64)),
                       dim3(64), s, (const float*)ws, nblk, L.cout, L.dbias);
            PN2_LAUNCH_CHECK();
        }
        // dY operand (through BatchNorm+ReLU backward when the layer has one)
        Operand dy = plain(dz, lddz, rows, L.cout);
        if (L.has_bn) {
            dy.q = y;
            dy.ldq = L.cout;
            dy.coef = L.stats;
            dy.cstride = L.cout;
            dy.relu = L.relu;
        }
        // layer input as an activation source
        Act in = i == 0 ? Act{x, ldx, nullptr, 0}
                        : Act{layers[i - 1].y, layers[i - 1].cout, layers[i - 1].has_bn ? layers[i - 1].stats : nullptr,
                              layers[i - 1].relu};
        // ---- wgrad: dW[cout][cin] += dY^T X, reduction over rows split across blocks
        if (L.dweight) {
            const int tiles = pn2::ceil_div(L.cout, BM) * pn2::ceil_div(L.cin, BN);
            int nsplit = 256 / tiles;
            if (nsplit < 1) nsplit = 1;
            int kps = pn2::ceil_div(pn2::ceil_div(rows, nsplit), BK) * BK;
            nsplit = pn2::ceil_div(rows, kps);
            GemmArgs g{};
            g.A = dy;                       // [rows = K][cout = M], direct layout
            g.B = act_operand(in, rows, L.cin);
            g.M = L.cout;
            g.N = L.cin;
            g.K = rows;
            g.C = ws;
            g.ldc = L.cin;
            g.k_per_split = kps;
            int st;
            if (L.has_bn)
                st = in.coef ? launch_gemm<false, TR_DY, false, TR_BNRELU, EPI_SLAB>(g, nsplit, s)
                             : launch_gemm<false, TR_DY, false, TR_PLAIN, EPI_SLAB>(g, nsplit, s);
            else
                st = in.coef ? launch_gemm<false, TR_PLAIN, false, TR_BNRELU, EPI_SLAB>(g, nsplit, s)
                             : launch_gemm<false, TR_PLAIN, false, TR_PLAIN, EPI_SLAB>(g, nsplit, s);
            if (st) return st;
            PN2_LAUNCH("slab_reduce", 4.0 * (nsplit + 2) * L.cout * L.cin, 0, slab_reduce_kernel,
                       dim3(grid1d((long long)L.cout * L.cin)), dim3(256), s, (const float*)ws, nsplit,
                       (long long)L.cout * L.cin, L.dweight);
            PN2_LAUNCH_CHECK();
        }
        // ---- dgrad: dX[rows][cin] = dY W
        const bool need_dx = i > 0 || dx != nullptr;
        if (need_dx) {
            float* target = i > 0 ? bufs[which] : dx;
            const long long ldt = i > 0 ? L.cin : lddx;
            GemmArgs g{};
            g.A = dy;                       // [rows = M][cout = K]
            g.B = plain(L.weight, L.cin, L.cout, L.cin);   // [cout = K][cin = N], direct layout
            g.M = rows;
            g.N = L.cin;
            g.K = L.cout;
            g.C = target;
            g.ldc = ldt;
            int st = L.has_bn ? launch_gemm<true, TR_DY, false, TR_PLAIN, EPI_STORE>(g, 1, s)
                              : launch_gemm<true, TR_PLAIN, false, TR_PLAIN, EPI_STORE>(g, 1, s);
            if (st) return st;
            dz = target;
            lddz = ldt;
            which ^= 1;
        }
    }
    return 0;
}
