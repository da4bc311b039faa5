// query_ball_point for gfx950 -- replaces Modules/PointNet2/pointnet2_utils.py:92-136.
//
// The reference materialises a [B,S,N] distance matrix, masks it and sorts S*N int64 keys to obtain "the first
// nsample indices, ascending, with d <= r^2".  Here one wavefront owns Q queries and walks the cloud in index
// order, 64 points per step: every lane holds one point (coalesced 4-byte loads per coordinate plane), tests
// it against the Q queries with the reference's exact fp32 expression, and the hits are appended in index
// order with a wavefront ballot + prefix popcount -- no sort, no distance matrix, early exit once all Q
// queries hold nsample hits.  When there are few queries the index range is cut into segments scanned by
// different wavefronts and a small second kernel stitches the per-segment lists back together in order.
//
// Empty balls (only possible when the query is not a member of the cloud) take the row argmin, first minimum,
// like torch.argmin (lines 113-122); short rows are padded with their first hit (lines 126-130).
#include "pn2_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

using u64 = unsigned long long;
constexpr int kBlock = 256;  // 4 wavefronts
constexpr int kMaxSeg = 64;

struct Cloud {
    const float* p;
    int64_t sn, sc;
    __device__ __forceinline__ void load(int n, float& x, float& y, float& z) const {
        const float* q = p + (int64_t)n * sn;
        x = q[0];
        y = q[sc];
        z = q[2 * sc];
    }
};

// argmin over the whole cloud for one query, first minimum wins; executed by a full wavefront.
__device__ int wave_argmin(const Cloud& c, int N, float qx, float qy, float qz, float qn) {
    const int lane = threadIdx.x & 63;
    float bd = __builtin_inff();
    int bi = 0x7FFFFFFF;
    for (int n = lane; n < N; n += 64) {
        float x, y, z;
        c.load(n, x, y, z);
        const float d = pn2::sqdist(qx, qy, qz, qn, x, y, z, pn2::norm2(x, y, z));
        if (d < bd) {
            bd = d;
            bi = n;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    return bi == 0x7FFFFFFF ? 0 : bi;
}

// pad a finished row: lanes k in [cnt, Keff) copy the first hit; cnt == 0 -> argmin fallback
__device__ void finish_row(const Cloud& c, int N, int32_t* row, int cnt, int first, int Keff, float qx, float qy,
                           float qz, float qn) {
    const int lane = threadIdx.x & 63;
    if (cnt == 0) first = wave_argmin(c, N, qx, qy, qz, qn);
    for (int k = cnt + lane; k < Keff; k += 64) row[k] = first;
}

template <int Q>
__global__ __launch_bounds__(kBlock) void ball_query_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn,
                                                            int64_t sc, const float* __restrict__ new_xyz, int64_t qb,
                                                            int64_t qs, int64_t qc, int B, int N, int S, float r2,
                                                            int Keff, int32_t* __restrict__ out_idx, int seg_len,
                                                            int nseg, int32_t* __restrict__ part_idx,
                                                            int32_t* __restrict__ part_cnt, const int* __restrict__ coff) {
    const int lane = threadIdx.x & 63;
    const int nqg = (S + Q - 1) / Q;
    // wave order [b][seg][query group]: the 4 waves of a workgroup scan the same segment
    long long w = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    w = __builtin_amdgcn_readfirstlane((int)w);
    const long long total = (long long)B * nseg * nqg;
    if (w >= total) return;
    const int qg = (int)(w % nqg);
    const int seg = (int)((w / nqg) % nseg);
    const int b = (int)(w / ((long long)nqg * nseg));

    const pn2::CloudView cv = pn2::cloud_view(xyz, sb, sn, sc, N, coff, b, 3);   // ragged batches: nseg == 1
    const Cloud c{cv.p, cv.sn, cv.sc};
    N = cv.n;
    static_assert(Q % 2 == 0, "queries are tested in pairs");
    float qx[Q], qy[Q], qz[Q], qn[Q];
    int cnt[Q], first[Q];
    int32_t* row[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const int s = qg * Q + i;
        const bool ok = s < S;
        const float* q = new_xyz + (int64_t)b * qb + (int64_t)(ok ? s : 0) * qs;
        qx[i] = q[0];
        qy[i] = q[qc];
        qz[i] = q[2 * qc];
        qn[i] = pn2::norm2(qx[i], qy[i], qz[i]);
        cnt[i] = ok ? 0 : Keff;  // a slot beyond S is born full
        first[i] = 0;
        const size_t qid = (size_t)b * S + (ok ? s : 0);
        row[i] = nseg == 1 ? out_idx + qid * Keff : part_idx + (qid * nseg + seg) * Keff;
    }

    const int n_begin = seg * seg_len;
    const int n_end = (n_begin + seg_len) < N ? (n_begin + seg_len) : N;
    const u64 lt = pn2::lanemask_lt();
    pn2::f2 qx2[Q / 2], qy2[Q / 2], qz2[Q / 2], qn2[Q / 2];
#pragma unroll
    for (int i = 0; i < Q; i += 2) {
        qx2[i / 2] = pn2::f2{qx[i], qx[i + 1]};
        qy2[i / 2] = pn2::f2{qy[i], qy[i + 1]};
        qz2[i / 2] = pn2::f2{qz[i], qz[i + 1]};
        qn2[i / 2] = pn2::f2{qn[i], qn[i + 1]};
    }
    // the loads of block n0 + 64 are in flight while block n0 is tested
    float nx, ny, nz;
    c.load(n_begin + lane < n_end ? n_begin + lane : n_begin, nx, ny, nz);
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
        const int n = n0 + lane;
        const bool ok = n < n_end;
        const float x = nx, y = ny, z = nz;
        c.load(n + 64 < n_end ? n + 64 : n_begin, nx, ny, nz);
        const float pn = pn2::norm2(x, y, z);
        const pn2::f2 x2 = {-2.0f * x, -2.0f * x}, y2 = {-2.0f * y, -2.0f * y}, z2 = {-2.0f * z, -2.0f * z},
                      pn2v = {pn, pn};
        bool in[Q];
        bool any = false;
#pragma unroll
        for (int i = 0; i < Q; i += 2) {
            const pn2::f2 d = pn2::sqdist2(qx2[i / 2], qy2[i / 2], qz2[i / 2], qn2[i / 2], x2, y2, z2, pn2v);
            in[i] = ok && !(d.x > r2);
            in[i + 1] = ok && !(d.y > r2);
            any |= in[i] | in[i + 1];
        }
        if (__ballot(any) == 0) continue;  // common case: no lane holds a hit for any of the Q queries
        bool all_full = true;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            if (cnt[i] < Keff) {
                const bool hit = in[i];
                const u64 m = __ballot(hit);
                if (m) {
                    if (cnt[i] == 0) first[i] = n0 + __builtin_ctzll(m);
                    const int pos = cnt[i] + __popcll(m & lt);
                    if (hit && pos < Keff) row[i][pos] = n;
                    cnt[i] += __popcll(m);
                }
                all_full = all_full && (cnt[i] >= Keff);
            }
        }
        if (all_full) break;
    }

#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const int s = qg * Q + i;
        if (s >= S) continue;
        const int have = cnt[i] < Keff ? cnt[i] : Keff;
        if (nseg == 1) {
            finish_row(c, N, row[i], have, first[i], Keff, qx[i], qy[i], qz[i], qn[i]);
        } else if (lane == 0) {
            part_cnt[((size_t)b * S + s) * nseg + seg] = have;
        }
    }
}

// One wavefront per query: concatenate the per-segment hit lists in segment (= index) order.  Lane g reads the count
// of segment g (nseg <= 64), a wavefront scan turns the counts into offsets, and every output slot finds its segment
// by bisection of the offsets in LDS -- two dependent global loads per query instead of one per segment.
__global__ __launch_bounds__(kBlock) void ball_query_merge_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn,
                                                                  int64_t sc, const float* __restrict__ new_xyz,
                                                                  int64_t qb, int64_t qs, int64_t qc, int B, int N,
                                                                  int S, int Keff, int32_t* __restrict__ out_idx,
                                                                  int nseg, const int32_t* __restrict__ part_idx,
                                                                  const int32_t* __restrict__ part_cnt) {
    __shared__ int s_off[kBlock / 64][kMaxSeg + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long w = (long long)blockIdx.x * (kBlock / 64) + wv;
    w = __builtin_amdgcn_readfirstlane((int)w);
    if (w >= (long long)B * S) return;
    const int b = (int)(w / S), s = (int)(w % S);
    const size_t qid = (size_t)w;
    int32_t* row = out_idx + qid * Keff;
    const int c = lane < nseg ? part_cnt[qid * nseg + lane] : 0;
    int incl = c;  // inclusive scan over the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    int* offs = s_off[wv];
    offs[lane + 1] = incl;
    if (lane == 0) offs[0] = 0;
    __builtin_amdgcn_wave_barrier();
    const int all = offs[nseg];
    const int total = all < Keff ? all : Keff;
    int first = 0;
    for (int j = lane; j < total; j += 64) {
        int lo = 0, hi = nseg;  // largest seg with offs[seg] <= j
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= j) lo = mid; else hi = mid;
        }
        row[j] = part_idx[(qid * nseg + lo) * Keff + (j - offs[lo])];
    }
    if (total > 0 && total < Keff) {
        // first hit = first entry of the first non-empty segment
        const u64 nonempty = __ballot(c > 0);
        const int seg0 = __builtin_ctzll(nonempty);
        first = part_idx[(qid * nseg + seg0) * Keff];
    }
    if (total == Keff) return;
    const float* q = new_xyz + (int64_t)b * qb + (int64_t)s * qs;
    const float qx = q[0], qy = q[qc], qz = q[2 * qc];
    const Cloud cl{xyz + (int64_t)b * sb, sn, sc};
    finish_row(cl, N, row, total, first, Keff, qx, qy, qz, pn2::norm2(qx, qy, qz));
}

struct Plan {
    int Q, nseg, seg_len;
};

Plan plan(int B, int N, int S) {
    // 8 queries per wavefront amortise each point load over 8 tests; the index range is cut into segments until
    // ~8192 wavefronts exist (measured on MI355X: B=1, N=262144, S=1024 -> (8,64); B=8, N=8192 -> (8,8))
    const long long queries = (long long)B * S;
    int Q = 8;
    while (Q > 2 && queries < 8 * Q) Q >>= 1;
    const long long waves = (queries + Q - 1) / Q;
    long long nseg = 8192 / (waves > 0 ? waves : 1);
    const int nblk = pn2::ceil_div(N, 64);
    if (nseg > kMaxSeg) nseg = kMaxSeg;
    if (nseg > nblk / 8) nseg = nblk / 8;  // at least 512 points per segment
    if (nseg < 1) nseg = 1;
    if (const char* e = getenv("PN2_BQ_PLAN")) {  // "Q,nseg" -- tuning aid
        int q = 0, g = 0;
        if (sscanf(e, "%d,%d", &q, &g) == 2 && (q == 2 || q == 4 || q == 8) && g >= 1 && g <= kMaxSeg) {
            Q = q;
            nseg = g;
        }
    }
    int seg_len = pn2::ceil_div(pn2::ceil_div(N, nseg), 64) * 64;
    nseg = pn2::ceil_div(N, seg_len);
    return Plan{Q, (int)nseg, seg_len};
}

}  // namespace

extern "C" size_t pn2_ball_query_workspace_bytes(int B, int N, int S, int nsample) {
    if (B <= 0 || N <= 0 || S <= 0 || nsample <= 0) return 0;
    const Plan p = plan(B, N, S);
    const int Keff = nsample < N ? nsample : N;
    if (p.nseg == 1) return 16;
    return (size_t)B * S * p.nseg * (Keff + 1) * sizeof(int32_t) + 16;
}

extern "C" int pn2_ball_query_f32(const float* xyz, int64_t sb, int64_t sn, int64_t sc, const float* new_xyz,
                                  int64_t qb, int64_t qn, int64_t qc, int B, int N, int S, float r2, int nsample,
                                  int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    if (!xyz || !new_xyz || !out_idx || B <= 0 || N <= 0 || S <= 0 || nsample <= 0) return PN2_E_BADARG;
    const Plan p = plan(B, N, S);
    const int Keff = nsample < N ? nsample : N;
    if (workspace_bytes < pn2_ball_query_workspace_bytes(B, N, S, nsample) || (p.nseg > 1 && !workspace))
        return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int32_t* part_idx = (int32_t*)workspace;
    int32_t* part_cnt = part_idx ? part_idx + (size_t)B * S * p.nseg * Keff : nullptr;
    const int nqg = pn2::ceil_div(S, p.Q);
    const long long waves = (long long)B * p.nseg * nqg;
    if (waves > 0x7FFFFFFFll) return PN2_E_BADARG;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(kBlock);
    // algorithmic bytes per SURVEY 8d; "flops" = 8 per (query, point) distance test of the brute-force scan
    const double bq_bytes = (double)B * (12.0 * N + 12.0 * S + 8.0 * S * Keff);
#define PN2_BQ_CASE(Q_)                                                                                               \
    if (p.Q == Q_)                                                                                                    \
        PN2_LAUNCH("ball_query", bq_bytes, 8.0 * B * (double)S * N, (ball_query_kernel<Q_>), grid, block, s, xyz, sb, sn, sc, new_xyz, qb, qn, \
                   qc, B, N, S, r2, Keff, out_idx, p.seg_len, p.nseg, part_idx, part_cnt, (const int*)nullptr);
    PN2_BQ_CASE(2)
    PN2_BQ_CASE(4)
    PN2_BQ_CASE(8)
#undef PN2_BQ_CASE
    PN2_LAUNCH_CHECK();
    if (p.nseg > 1) {
        const long long q = (long long)B * S;
        PN2_LAUNCH("ball_query_merge", (double)q * Keff * 8.0, 0, ball_query_merge_kernel, dim3((unsigned)((q + 3) / 4)), block,
                   s, xyz, sb, sn, sc, new_xyz, qb, qn, qc, B, N, S, Keff, out_idx, p.nseg, part_idx, part_cnt);
        PN2_LAUNCH_CHECK();
    }
    return 0;
}

// Ragged batch (whole-tree execution): C small clouds of n_b >= nsample points each, S queries per cloud.  Thousands of
// (cloud, query group) wavefronts exist already, so every wavefront scans its whole cloud (one segment, no merge pass).
extern "C" int pn2_ball_query_ragged_f32(const float* xyz_cf, const int32_t* coff, const float* new_xyz, int C, int n_max, int S,
                                         float r2, int nsample, int32_t* out_idx, void* stream) {
    if (!xyz_cf || !coff || !new_xyz || !out_idx || C <= 0 || n_max <= 0 || S <= 0 || nsample <= 0) return PN2_E_BADARG;
    int Q = 8;
    while (Q > 2 && (long long)C * S < 8 * Q) Q >>= 1;
    const int nqg = pn2::ceil_div(S, Q);
    const long long waves = (long long)C * nqg;
    if (waves > 0x7FFFFFFFll) return PN2_E_BADARG;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const double bq_bytes = (double)C * (12.0 * n_max + 12.0 * S + 8.0 * S * nsample);
    const int seg_len = pn2::ceil_div(n_max, 64) * 64;
#define PN2_BQ_CASE(Q_)                                                                                                  \
    if (Q == Q_)                                                                                                         \
        PN2_LAUNCH("ball_query", bq_bytes, 8.0 * C * (double)S * n_max, (ball_query_kernel<Q_>), grid, block, s, xyz_cf, 0, 1, 0, \
                   new_xyz, (int64_t)S * 3, 3, 1, C, n_max, S, r2, nsample, out_idx, seg_len, 1, (int32_t*)nullptr,        \
                   (int32_t*)nullptr, (const int*)coff);
    PN2_BQ_CASE(2)
    PN2_BQ_CASE(4)
    PN2_BQ_CASE(8)
#undef PN2_BQ_CASE
    PN2_LAUNCH_CHECK();
    return 0;
}
