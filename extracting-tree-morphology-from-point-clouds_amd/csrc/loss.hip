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

// out[0] = semantic, out[1] = offset; counts clamped to >= 1 like the host expression.  With weights: out[k] *= weights[k]
// and *total = out[0] + out[1] (the loss multipliers and the sum of get_loss, PointNet2.py:198-207).
__global__ void point_loss_finalize_kernel(const double* __restrict__ partial, int nblk, const int64_t* __restrict__ cum_pad,
                                           const int64_t* __restrict__ cum_off, int R, float* __restrict__ out,
                                           const float* __restrict__ weights, float* __restrict__ total) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s1 += partial[i * 2], s2 += partial[i * 2 + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (threadIdx.x == 0) {
        const long long nv = cum_pad[R - 1] > 1 ? cum_pad[R - 1] : 1, no = cum_off[R - 1] > 1 ? cum_off[R - 1] : 1;
        float a = (float)(s1 / (double)nv), b = (float)(s2 / (double)no);
        if (weights) {
            a = __fmul_rn(a, weights[0]);
            b = __fmul_rn(b, weights[1]);
            *total = __fadd_rn(a, b);
        }
        out[0] = a;
        out[1] = b;
    }
}

__global__ __launch_bounds__(kBlock) void point_loss_bwd_kernel(const float* __restrict__ sem, const float* __restrict__ off,
                                                                const bool* __restrict__ pad, const bool* __restrict__ off_mask,
                                                                const int64_t* __restrict__ cum_pad,
                                                                const int64_t* __restrict__ cum_off,
                                                                const int64_t* __restrict__ sem_lab, long long n_sem,
                                                                const float* __restrict__ off_lab, long long n_off_lab, int R,
                                                                const float* __restrict__ g, float* __restrict__ dsem,
                                                                float* __restrict__ doff, const float* __restrict__ weights) {
    const long long nv = cum_pad[R - 1] > 1 ? cum_pad[R - 1] : 1, no = cum_off[R - 1] > 1 ? cum_off[R - 1] : 1;
    // weights: g is the gradient of the weighted TOTAL (one float)
    const float g0 = weights ? __fmul_rn(g[0], weights[0]) : g[0], g1 = weights ? __fmul_rn(g[0], weights[1]) : g[1];
    const float gs = g0 / (float)nv, go = g1 / (float)no;
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

// ---------------------------------------------------------------------------------------------- mask ranks
// cum_pad, off_mask, cum_off of get_loss (PointNet2.py:188-196: boolean masking of the padded rows, then of the rows whose
// offsets count) in ONE launch instead of two prefix sums, a gather and the element-wise glue between them.  Workgroup i
// owns rows [1024 i, 1024 i + 1024); what precedes it is two plain counts -- R_i = real rows before the chunk and, because
// the real rows consume masks_off in order, O_i = set entries of masks_off[0 : R_i) -- which every workgroup takes for
// itself from memory the caches hold anyway (<= 2 x R bytes), so there is no carry between workgroups to wait for.
constexpr int kRankBlock = 1024;

__device__ __forceinline__ int nonzero_bytes(unsigned x) {
    return __popc((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u);
}

// number of non-zero bytes of m[0 : n), counted by the whole workgroup; `red` = 16 ints of LDS
__device__ __forceinline__ long long count_prefix(const unsigned char* __restrict__ m, long long n, int* red) {
    const int t = threadIdx.x;
    long long c = 0;
    long long head = (16 - ((uintptr_t)m & 15)) & 15;
    if (head > n) head = n;
    if (t < head) c += m[t] != 0;
    const uint4* v = (const uint4*)(m + head);
    const long long nv = (n - head) / 16;
    for (long long e0 = t; e0 < nv; e0 += 8 * kRankBlock) {   // eight loads in flight per thread
        uint4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long e = e0 + (long long)u * kRankBlock;
            q[u] = v[e < nv ? e : nv - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + (long long)u * kRankBlock < nv)
                c += nonzero_bytes(q[u].x) + nonzero_bytes(q[u].y) + nonzero_bytes(q[u].z) + nonzero_bytes(q[u].w);
    }
    const long long tail = head + nv * 16;
    if (tail + t < n) c += m[tail + t] != 0;   // < 16 bytes
    int w = (int)c;   // <= 16 * ceil(n / 16384) + 2 per thread (n <= 2^30)
    for (int o = 32; o; o >>= 1) w += __shfl_xor(w, o);
    __syncthreads();   // red may still be read from the previous use
    if ((t & 63) == 0) red[t >> 6] = w;
    __syncthreads();
    long long total = 0;
    for (int k = 0; k < kRankBlock / 64; ++k) total += red[k];
    return total;
}

// inclusive scan of one flag per thread over the workgroup
__device__ __forceinline__ int block_scan_flag(bool f, int* red) {
    const int t = threadIdx.x, lane = t & 63;
    const unsigned long long b = __ballot(f);
    const int incl = __popcll(b & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
    __syncthreads();
    if (lane == 0) red[t >> 6] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int k = 0; k < (t >> 6); ++k) before += red[k];
    return before + incl;
}

__global__ __launch_bounds__(kRankBlock) void mask_ranks_kernel(const unsigned char* __restrict__ pad,
                                                               const unsigned char* __restrict__ masks_off, long long R,
                                                               long long n_mask, int64_t* __restrict__ cum_pad,
                                                               unsigned char* __restrict__ off_mask, int64_t* __restrict__ cum_off) {
    __shared__ int red[kRankBlock / 64];
    const long long r0 = (long long)blockIdx.x * kRankBlock, r = r0 + threadIdx.x;
    const long long real_before = count_prefix(pad, r0, red);
    // rows beyond n_mask (a masks_off shorter than the number of real rows) all read its last entry, like the clamp of the
    // torch expression this replaces
    const long long in_range = real_before < n_mask ? real_before : n_mask;
    long long off_before = count_prefix(masks_off, in_range, red);
    if (real_before > n_mask && masks_off[n_mask - 1] != 0) off_before += real_before - n_mask;
    const bool p = r < R && pad[r] != 0;
    const long long cp = real_before + block_scan_flag(p, red);
    long long rank = cp - 1;
    rank = rank < 0 ? 0 : (rank >= n_mask ? n_mask - 1 : rank);
    const bool o = p && masks_off[rank] != 0;
    const long long co = off_before + block_scan_flag(o, red);
    if (r < R) {
        cum_pad[r] = cp;
        off_mask[r] = o ? 1 : 0;
        cum_off[r] = co;
    }
}

}  // namespace

extern "C" size_t pn2_point_loss_workspace_bytes(int R) { return R > 0 ? (size_t)loss_blocks(R) * 2 * sizeof(double) : 0; }

static int point_loss_fwd(const float* sem, const float* off, const unsigned char* pad, const unsigned char* off_mask,
                          const int64_t* cum_pad, const int64_t* cum_off, const int64_t* sem_labels, int64_t n_sem,
                          const float* off_labels, int64_t n_off, int R, float* out2, void* workspace, size_t workspace_bytes,
                          void* stream, const float* weights, float* total) {
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
               cum_pad, cum_off, R, out2, weights, total);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_point_loss_fwd_f32(const float* sem, const float* off, const unsigned char* pad, const unsigned char* off_mask,
                                      const int64_t* cum_pad, const int64_t* cum_off, const int64_t* sem_labels, int64_t n_sem,
                                      const float* off_labels, int64_t n_off, int R, float* out2, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return point_loss_fwd(sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, n_sem, off_labels, n_off, R, out2, workspace,
                          workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int pn2_point_loss_weighted_fwd_f32(const float* sem, const float* off, const unsigned char* pad,
                                               const unsigned char* off_mask, const int64_t* cum_pad, const int64_t* cum_off,
                                               const int64_t* sem_labels, int64_t n_sem, const float* off_labels, int64_t n_off,
                                               int R, const float* weights2, float* out2, float* total, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    if (!weights2 || !total) return PN2_E_BADARG;
    return point_loss_fwd(sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, n_sem, off_labels, n_off, R, out2, workspace,
                          workspace_bytes, stream, weights2, total);
}

static int point_loss_bwd(const float* sem, const float* off, const unsigned char* pad, const unsigned char* off_mask,
                          const int64_t* cum_pad, const int64_t* cum_off, const int64_t* sem_labels, int64_t n_sem,
                          const float* off_labels, int64_t n_off, int R, const float* grad2, float* dsem, float* doff, void* stream,
                          const float* weights) {
    if (!sem || !off || !pad || !off_mask || !cum_pad || !cum_off || !sem_labels || !off_labels || !grad2 || !dsem || !doff ||
        R <= 0 || n_sem <= 0 || n_off <= 0)
        return PN2_E_BADARG;
    PN2_LAUNCH("point_loss_bwd", 58.0 * R, 0, point_loss_bwd_kernel, dim3(loss_blocks(R)), dim3(kBlock), (hipStream_t)stream, sem,
               off, (const bool*)pad, (const bool*)off_mask, cum_pad, cum_off, sem_labels, (long long)n_sem, off_labels,
               (long long)n_off, R, grad2, dsem, doff, weights);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_point_loss_bwd_f32(const float* sem, const float* off, const unsigned char* pad, const unsigned char* off_mask,
                                      const int64_t* cum_pad, const int64_t* cum_off, const int64_t* sem_labels, int64_t n_sem,
                                      const float* off_labels, int64_t n_off, int R, const float* grad2, float* dsem, float* doff,
                                      void* stream) {
    return point_loss_bwd(sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, n_sem, off_labels, n_off, R, grad2, dsem, doff,
                          stream, nullptr);
}

extern "C" int pn2_point_loss_weighted_bwd_f32(const float* sem, const float* off, const unsigned char* pad,
                                               const unsigned char* off_mask, const int64_t* cum_pad, const int64_t* cum_off,
                                               const int64_t* sem_labels, int64_t n_sem, const float* off_labels, int64_t n_off,
                                               int R, const float* grad_total, const float* weights2, float* dsem, float* doff,
                                               void* stream) {
    if (!weights2) return PN2_E_BADARG;
    return point_loss_bwd(sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, n_sem, off_labels, n_off, R, grad_total, dsem,
                          doff, stream, weights2);
}

extern "C" int pn2_mask_ranks(const unsigned char* pad, const unsigned char* masks_off, long long R, long long n_mask,
                              int64_t* cum_pad, unsigned char* off_mask, int64_t* cum_off, void* stream) {
    if (!pad || !masks_off || !cum_pad || !off_mask || !cum_off || R <= 0 || n_mask <= 0 || R > (1LL << 30)) return PN2_E_BADARG;
    const long long blocks = (R + kRankBlock - 1) / kRankBlock;
    if (blocks > 0x7FFFFFFFLL) return PN2_E_BADARG;
    PN2_LAUNCH("mask_ranks", 18.0 * R, 0, mask_ranks_kernel, dim3((unsigned)blocks), dim3(kRankBlock), (hipStream_t)stream, pad,
               masks_off, R, n_mask, cum_pad, off_mask, cum_off);
    PN2_LAUNCH_CHECK();
    return 0;
}
