// LayerNorm over the channels of point rows, forward and backward, for gfx950 (the `norm_layer` of every
// PointTransformerV3 Block, Modules/PointTransformerV3/blocks.py:551-597 -- nn.LayerNorm on [N, C] rows, C = 32 ... 512).
// torch's kernel gives a row to a workgroup whatever its width: at C = 32 and a million rows that is 0.5 ms per call where the
// rows are 0.26 GB of traffic (66 calls per forward, 35 ms; 91 ms per training step with the backward).  Here L = C / 4 (<= 64)
// lanes own a row -- a lane keeps its one or two float4 of it in registers -- so a wavefront handles 64 / L rows per step with
// 16-byte accesses, the row sums are xor-shuffles inside the L lanes, and the statistics are the two-pass ones (mean first, then
// the centred squares: what torch computes, not E[x^2] - mean^2).
//   forward   y = (x - mean) * rstd * gamma + beta;  mean, rstd [rows] kept for the backward when asked for
//   backward  g = dy * gamma;  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat));  dgamma = sum_rows dy * xhat, dbeta = sum_rows dy:
//             every lane accumulates its channels over the rows it sees, the row groups of a wavefront and the four wavefronts
//             of a workgroup are joined in fixed order, a second launch adds the workgroups' partials in order (deterministic).
#include "pn2_common.h"

namespace {

constexpr int LN_T = 256;
constexpr int LN_MAX_BLOCKS = 512;

template <int L>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int L, int NCH>
__global__ __launch_bounds__(LN_T) void layer_norm_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, long long rows,
                                                              float* __restrict__ y, long long ldy, float* __restrict__ mean,
                                                              float* __restrict__ rstd) {
    constexpr int C = 4 * L * NCH, RPW = 64 / L;
    const int lane = threadIdx.x & 63, sub = lane % L, rg = lane / L;
    const long long wave = (long long)blockIdx.x * (LN_T / 64) + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * (LN_T / 64);
    float4 gm[NCH], bt[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        gm[h] = gamma ? *(const float4*)(gamma + 4 * (sub + L * h)) : make_float4(1.f, 1.f, 1.f, 1.f);
        bt[h] = beta ? *(const float4*)(beta + 4 * (sub + L * h)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
        const long long r = r0 + rg;
        const bool in = r < rows;
        const long long rr = in ? r : rows - 1;
        float4 v[NCH];
        float s = 0.0f;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            v[h] = *(const float4*)(x + rr * ldx + 4 * (sub + L * h));
            s += (v[h].x + v[h].y) + (v[h].z + v[h].w);
        }
        const float m = row_sum<L>(s) * (1.0f / C);
        float q = 0.0f;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            v[h].x -= m, v[h].y -= m, v[h].z -= m, v[h].w -= m;
            q += (v[h].x * v[h].x + v[h].y * v[h].y) + (v[h].z * v[h].z + v[h].w * v[h].w);
        }
        const float rs = 1.0f / sqrtf(row_sum<L>(q) * (1.0f / C) + eps);
        if (in) {
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                float4 o;
                o.x = v[h].x * rs * gm[h].x + bt[h].x;
                o.y = v[h].y * rs * gm[h].y + bt[h].y;
                o.z = v[h].z * rs * gm[h].z + bt[h].z;
                o.w = v[h].w * rs * gm[h].w + bt[h].w;
                *(float4*)(y + r * ldy + 4 * (sub + L * h)) = o;
            }
            if (mean && sub == 0) mean[r] = m, rstd[r] = rs;
        }
    }
}

template <int L, int NCH>
__global__ __launch_bounds__(LN_T) void layer_norm_bwd_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                              long long ldx, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              long long rows, float* __restrict__ dx, long long lddx,
                                                              float* __restrict__ partial) {
    constexpr int C = 4 * L * NCH, RPW = 64 / L;
    __shared__ __attribute__((aligned(16))) float red[LN_T / 64][2][C];
    const int lane = threadIdx.x & 63, sub = lane % L, rg = lane / L, w = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * (LN_T / 64) + w, nwaves = (long long)gridDim.x * (LN_T / 64);
    float4 gm[NCH], ag[NCH], ab[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        gm[h] = gamma ? *(const float4*)(gamma + 4 * (sub + L * h)) : make_float4(1.f, 1.f, 1.f, 1.f);
        ag[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[h] = ag[h];
    }
    for (long long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
        const long long r = r0 + rg;
        const bool in = r < rows;
        const long long rr = in ? r : rows - 1;
        const float m = mean[rr], rs = rstd[rr];
        float4 g[NCH], xh[NCH];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            float4 d = *(const float4*)(dy + rr * lddy + 4 * (sub + L * h));
            const float4 xv = *(const float4*)(x + rr * ldx + 4 * (sub + L * h));
            if (!in) d = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[h] = make_float4((xv.x - m) * rs, (xv.y - m) * rs, (xv.z - m) * rs, (xv.w - m) * rs);
            ab[h].x += d.x, ab[h].y += d.y, ab[h].z += d.z, ab[h].w += d.w;
            ag[h].x += d.x * xh[h].x, ag[h].y += d.y * xh[h].y, ag[h].z += d.z * xh[h].z, ag[h].w += d.w * xh[h].w;
            g[h] = make_float4(d.x * gm[h].x, d.y * gm[h].y, d.z * gm[h].z, d.w * gm[h].w);
            s1 += (g[h].x + g[h].y) + (g[h].z + g[h].w);
            s2 += (g[h].x * xh[h].x + g[h].y * xh[h].y) + (g[h].z * xh[h].z + g[h].w * xh[h].w);
        }
        s1 = row_sum<L>(s1) * (1.0f / C);
        s2 = row_sum<L>(s2) * (1.0f / C);
        if (in) {
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                float4 o;
                o.x = rs * (g[h].x - s1 - xh[h].x * s2);
                o.y = rs * (g[h].y - s1 - xh[h].y * s2);
                o.z = rs * (g[h].z - s1 - xh[h].z * s2);
                o.w = rs * (g[h].w - s1 - xh[h].w * s2);
                *(float4*)(dx + r * lddx + 4 * (sub + L * h)) = o;
            }
        }
    }
    // join the wavefront's row groups (lanes with the same lane % L), then the workgroup's wavefronts, in fixed order
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
#pragma unroll
        for (int o = L; o < 64; o <<= 1) {
            ag[h].x += __shfl_xor(ag[h].x, o, 64), ag[h].y += __shfl_xor(ag[h].y, o, 64);
            ag[h].z += __shfl_xor(ag[h].z, o, 64), ag[h].w += __shfl_xor(ag[h].w, o, 64);
            ab[h].x += __shfl_xor(ab[h].x, o, 64), ab[h].y += __shfl_xor(ab[h].y, o, 64);
            ab[h].z += __shfl_xor(ab[h].z, o, 64), ab[h].w += __shfl_xor(ab[h].w, o, 64);
        }
        if (rg == 0) {
            *(float4*)(&red[w][0][4 * (sub + L * h)]) = ag[h];
            *(float4*)(&red[w][1][4 * (sub + L * h)]) = ab[h];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += LN_T) {
        const int which = e / C, c = e - which * C;
        partial[((long long)blockIdx.x * 2 + which) * C + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// dgamma[c] = sum over the workgroups' partials, likewise dbeta: 16 wavefronts (lane = channel) take every 16th partial each, four
// loads in flight, and are joined in order -- a fixed order of additions whatever the launch (the first version walked the
// partials with ONE thread per channel: 2048 dependent loads, 0.5 ms per call, more than forward and backward together)
constexpr int LNP_W = 16;
__global__ __launch_bounds__(64 * LNP_W) void layer_norm_param_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[LNP_W][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double g[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int k = w;
        for (; k + 3 * LNP_W < nblk; k += 4 * LNP_W) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] += (double)partial[((long long)(k + u * LNP_W) * 2 + 0) * C + c];
                b[u] += (double)partial[((long long)(k + u * LNP_W) * 2 + 1) * C + c];
            }
        }
        for (; k < nblk; k += LNP_W) {
            g[0] += (double)partial[((long long)k * 2 + 0) * C + c];
            b[0] += (double)partial[((long long)k * 2 + 1) * C + c];
        }
    }
    red[w][0][lane] = (g[0] + g[1]) + (g[2] + g[3]);
    red[w][1][lane] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    if (w == 0 && c < C) {
        double sg = 0.0, sb = 0.0;
        for (int t = 0; t < LNP_W; ++t) sg += red[t][0][lane], sb += red[t][1][lane];
        if (dgamma) dgamma[c] = (float)sg;
        if (dbeta) dbeta[c] = (float)sb;
    }
}

inline int ln_blocks(long long rows, int L, int cap = LN_MAX_BLOCKS) {
    const long long per_block = (long long)(LN_T / 64) * (64 / L);   // rows per workgroup and step
    long long b = (rows + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

inline bool ln_width_ok(int C) { return C == 32 || C == 64 || C == 128 || C == 256 || C == 512; }

}  // namespace

extern "C" int pn2_layer_norm_supported(int C) { return ln_width_ok(C) ? 1 : 0; }

extern "C" int pn2_layer_norm_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                                      float* y, int64_t ldy, float* mean, float* rstd, void* stream) {
    if (!x || !y || rows <= 0 || !ln_width_ok(C) || ldx % 4 || ldy % 4 || ldx < C || ldy < C || ((uintptr_t)x & 15) ||
        ((uintptr_t)y & 15) || (gamma && ((uintptr_t)gamma & 15)) || (beta && ((uintptr_t)beta & 15)) || (!mean != !rstd))
        return PN2_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const double by = 8.0 * rows * C;
#define PN2_LN_FWD(LL, NN)                                                                                                         \
    PN2_LAUNCH("layer_norm_fwd", by, 0, (layer_norm_fwd_kernel<LL, NN>), dim3(ln_blocks(rows, LL, 4096)), dim3(LN_T), s, x, (long long)ldx, gamma, \
               beta, eps, (long long)rows, y, (long long)ldy, mean, rstd)
    switch (C) {
        case 32: PN2_LN_FWD(8, 1); break;
        case 64: PN2_LN_FWD(16, 1); break;
        case 128: PN2_LN_FWD(32, 1); break;
        case 256: PN2_LN_FWD(64, 1); break;
        default: PN2_LN_FWD(64, 2); break;
    }
#undef PN2_LN_FWD
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pn2_layer_norm_bwd_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || !ln_width_ok(C)) return 0;
    return (size_t)LN_MAX_BLOCKS * 2 * C * sizeof(float);
}

extern "C" int pn2_layer_norm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean, const float* rstd,
                                      const float* gamma, int64_t rows, int C, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !dx || rows <= 0 || !ln_width_ok(C) || lddy % 4 || ldx % 4 || lddx % 4 || lddy < C || ldx < C ||
        lddx < C || ((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dx & 15) || (gamma && ((uintptr_t)gamma & 15)))
        return PN2_E_BADARG;
    if (!workspace || workspace_bytes < pn2_layer_norm_bwd_workspace_bytes(rows, C)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const double by = 12.0 * rows * C;
    int nblk = 0;
#define PN2_LN_BWD(LL, NN)                                                                                                          \
    nblk = ln_blocks(rows, LL);                                                                                                     \
    PN2_LAUNCH("layer_norm_bwd", by, 0, (layer_norm_bwd_kernel<LL, NN>), dim3(nblk), dim3(LN_T), s, dy, (long long)lddy, x, (long long)ldx, \
               mean, rstd, gamma, (long long)rows, dx, (long long)lddx, (float*)workspace)
    switch (C) {
        case 32: PN2_LN_BWD(8, 1); break;
        case 64: PN2_LN_BWD(16, 1); break;
        case 128: PN2_LN_BWD(32, 1); break;
        case 256: PN2_LN_BWD(64, 1); break;
        default: PN2_LN_BWD(64, 2); break;
    }
#undef PN2_LN_BWD
    if (dgamma || dbeta)
        PN2_LAUNCH("layer_norm_params", 8.0 * nblk * C, 0, layer_norm_param_kernel, dim3((C + 63) / 64), dim3(64 * LNP_W), s,
                   (const float*)workspace, nblk, C, dgamma, dbeta);
    PN2_LAUNCH_CHECK();
    return 0;
}
