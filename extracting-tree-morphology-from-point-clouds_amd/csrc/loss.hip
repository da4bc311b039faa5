// Masked point-wise loss of the flattened mode in two launches (+ a tiny finalize) -- the arithmetic of
// Modules/Loss.py:6-36 (point_wise_loss) under the masks of Modules/PointNet2/PointNet2.py:180-207 (get_loss).
//
// Rows r of the padded batch: pad[r] says "real point", off_mask[r] "its offset counts"; cum_pad / cum_off are the
// inclusive prefix sums of those masks, so cum - 1 is the row's position in the COMPACTED label arrays the reference
// indexes after boolean masking.
//   semantic = sum_pad( logsumexp(sem_r) - sem_r[label_r] ) / n_valid              (F.cross_entropy(sum) / len)
//   offset   = sum_offmask( sqrt(max(|off_r - lab_r|^2, 1e-8)) ) / n_off            (sqrt(clamp(.)).mean())
// Backward recomputes the per-row terms: d sem = g0 / n_valid (softmax - onehot), d off = g1 / n_off (off - lab) / dist
// where the clamp is inactive, 0 elsewhere.
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;

struct Row {
    float ce, dist;
};

__device__ __forceinline__ void row_terms(const float* __restrict__ sem, const float* __restrict__ off, int r,
                                          const int64_t* __restrict__ sem_lab, long long n_sem, const float* __restrict__ off_lab,
                                          long long n_off_lab, bool is_pad, bool is_off, long long rank, long long rank_off,
                                          float& ce, float& dist, float& p0, float& p1, int& label, float& dx, float& dy,
                                          float& dz, float& sq) {
    ce = dist = 0.f;
    p0 = p1 = dx = dy = dz = sq = 0.f;
    label = 0;
    if (is_pad) {
        const long long k = rank < 0 ? 0 : (rank >= n_sem ? n_sem - 1 : rank);
        label = (int)sem_lab[k];
        const float a = sem[(size_t)r * 2], b = sem[(size_t)r * 2 + 1];
        const float m = fmaxf(a, b);
        const float ea = expf(a - m), eb = expf(b - m);
        const float lse = m + logf(ea + eb);
        ce = lse - (label ? b : a);
        const float inv = 1.0f / (ea + eb);
        p0 = ea * inv, p1 = eb * inv;
    }
    if (is_off) {
        const long long k = rank_off < 0 ? 0 : (rank_off >= n_off_lab ? n_off_lab - 1 : rank_off);
        dx = off[(size_t)r * 3] - off_lab[(size_t)k * 3];
        dy = off[(size_t)r * 3 + 1] - off_lab[(size_t)k * 3 + 1];
        dz = off[(size_t)r * 3 + 2] - off_lab[(size_t)k * 3 + 2];
        sq = (dx * dx + dy * dy) + dz * dz;
        dist = sqrtf(fmaxf(sq, 1e-8f));
    }
}

__global__ __launch_bounds__(kBlock) void point_loss_fwd_kernel(const float* __restrict__ sem, const float* __restrict__ off,
                                                                const bool* __restrict__ pad, const bool* __restrict__ off_mask,
                                                                const int64_t* __restrict__ cum_pad,
                                                                const int64_t* __restrict__ cum_off,
                                                                const int64_t* __restrict__ sem_lab, long long n_sem,
                                                                const float* __restrict__ off_lab, long long n_off_lab, int R,
                                                                double* __restrict__ partial) {
    __shared__ double red[2][kBlock / 64];
    double s1 = 0.0, s2 = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < R; r += gridDim.x * kBlock) {
        float ce, dist, p0, p1, dx, dy, dz, sq;
        int label;
        row_terms(sem, off, r, sem_lab, n_sem, off_lab, n_off_lab, pad[r], off_mask[r], cum_pad[r] - 1, cum_off[r] - 1, ce, dist,
                  p0, p1, label, dx, dy, dz, sq);
        s1 += (double)ce;
        s2 += (double)dist;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1, red[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// out[0] = semantic, out[1] = offset; counts clamped to >= 1 like the host expression
__global__ void point_loss_finalize_kernel(const double* __restrict__ partial, int nblk, const int64_t* __restrict__ cum_pad,
                                           const int64_t* __restrict__ cum_off, int R, float* __restrict__ out) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s1 += partial[i * 2], s2 += partial[i * 2 + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (threadIdx.x == 0) {
        const long long nv = cum_pad[R - 1] > 1 ? cum_pad[R - 1] : 1, no = cum_off[R - 1] > 1 ? cum_off[R - 1] : 1;
        out[0] = (float)(s1 / (double)nv);
        out[1] = (float)(s2 / (double)no);
    }
}

__global__ __launch_bounds__(kBlock) void point_loss_bwd_kernel(const float* __restrict__ sem, const float* __restrict__ off,
                                                                const bool* __restrict__ pad, const bool* __restrict__ off_mask,
                                                                const int64_t* __restrict__ cum_pad,
                                                                const int64_t* __restrict__ cum_off,
                                                                const int64_t* __restrict__ sem_lab, long long n_sem,
                                                                const float* __restrict__ off_lab, long long n_off_lab, int R,
                                                                const float* __restrict__ g, float* __restrict__ dsem,
                                                                float* __restrict__ doff) {
    const long long nv = cum_pad[R - 1] > 1 ? cum_pad[R - 1] : 1, no = cum_off[R - 1] > 1 ? cum_off[R - 1] : 1;
    const float gs = g[0] / (float)nv, go = g[1] / (float)no;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < R; r += gridDim.x * kBlock) {
        float ce, dist, p0, p1, dx, dy, dz, sq;
        int label;
        const bool is_pad = pad[r], is_off = off_mask[r];
        row_terms(sem, off, r, sem_lab, n_sem, off_lab, n_off_lab, is_pad, is_off, cum_pad[r] - 1, cum_off[r] - 1, ce, dist, p0,
                  p1, label, dx, dy, dz, sq);
        dsem[(size_t)r * 2] = is_pad ? gs * (p0 - (label ? 0.f : 1.f)) : 0.f;
        dsem[(size_t)r * 2 + 1] = is_pad ? gs * (p1 - (label ? 1.f : 0.f)) : 0.f;
        const float w = (is_off && sq >= 1e-8f) ? go / dist : 0.f;
        doff[(size_t)r * 3] = w * dx;
        doff[(size_t)r * 3 + 1] = w * dy;
        doff[(size_t)r * 3 + 2] = w * dz;
    }
}

inline int loss_blocks(int R) {
    const int b = pn2::ceil_div(R, kBlock * 4);
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

}  // namespace

extern "C" size_t pn2_point_loss_workspace_bytes(int R) { return R > 0 ? (size_t)loss_blocks(R) * 2 * sizeof(double) : 0; }

extern "C" int pn2_point_loss_fwd_f32(const float* sem, const float* off, const unsigned char* pad, const unsigned char* off_mask,
                                      const int64_t* cum_pad, const int64_t* cum_off, const int64_t* sem_labels, int64_t n_sem,
                                      const float* off_labels, int64_t n_off, int R, float* out2, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (!sem || !off || !pad || !off_mask || !cum_pad || !cum_off || !sem_labels || !off_labels || !out2 || !workspace || R <= 0 ||
        n_sem <= 0 || n_off <= 0)
        return PN2_E_BADARG;
    if (workspace_bytes < pn2_point_loss_workspace_bytes(R)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = loss_blocks(R);
    PN2_LAUNCH("point_loss_fwd", 38.0 * R, 0, point_loss_fwd_kernel, dim3(nblk), dim3(kBlock), s, sem, off, (const bool*)pad,
               (const bool*)off_mask, cum_pad, cum_off, sem_labels, (long long)n_sem, off_labels, (long long)n_off, R,
               (double*)workspace);
    PN2_LAUNCH("point_loss_fwd", 16.0 * nblk, 0, point_loss_finalize_kernel, dim3(1), dim3(64), s, (const double*)workspace, nblk,
               cum_pad, cum_off, R, out2);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_point_loss_bwd_f32(const float* sem, const float* off, const unsigned char* pad, const unsigned char* off_mask,
                                      const int64_t* cum_pad, const int64_t* cum_off, const int64_t* sem_labels, int64_t n_sem,
                                      const float* off_labels, int64_t n_off, int R, const float* grad2, float* dsem, float* doff,
                                      void* stream) {
    if (!sem || !off || !pad || !off_mask || !cum_pad || !cum_off || !sem_labels || !off_labels || !grad2 || !dsem || !doff ||
        R <= 0 || n_sem <= 0 || n_off <= 0)
        return PN2_E_BADARG;
    PN2_LAUNCH("point_loss_bwd", 58.0 * R, 0, point_loss_bwd_kernel, dim3(loss_blocks(R)), dim3(kBlock), (hipStream_t)stream, sem,
               off, (const bool*)pad, (const bool*)off_mask, cum_pad, cum_off, sem_labels, (long long)n_sem, off_labels,
               (long long)n_off, R, grad2, dsem, doff);
    PN2_LAUNCH_CHECK();
    return 0;
}
