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
#include "pn2_cells.h"
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
    // the loads of the next PF blocks of 64 points are in flight while PF blocks are tested (deep levels: a few wavefronts
    // walk a small cloud end to end, one exposed load latency per step would be all of their time)
    constexpr int PF = 4;
    float fx[PF], fy[PF], fz[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int n = n_begin + 64 * u + lane;
        c.load(n < n_end ? n : n_begin, fx[u], fy[u], fz[u]);
    }
    bool done = false;
    for (int s0 = n_begin; s0 < n_end && !done; s0 += 64 * PF) {
        float cx[PF], cy[PF], cz[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) cx[u] = fx[u], cy[u] = fy[u], cz[u] = fz[u];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int n = s0 + 64 * (PF + u) + lane;
            c.load(n < n_end ? n : n_begin, fx[u], fy[u], fz[u]);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int n0 = s0 + 64 * u;
            if (n0 >= n_end || done) break;   // uniform
            const int n = n0 + lane;
            const bool ok = n < n_end;
            const float x = cx[u], y = cy[u], z = cz[u];
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
            done = all_full;
        }
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
    if (N <= 2048) {   // the deep levels: a scan is 32 steps at most -- more wavefronts (2 queries each), no merge launch
        Q = 2;         // (1 x 1024 points, 256 queries: (8,2) + merge 19 us, (2,1) 11 us)
        nseg = 1;
    }
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

// ------------------------------------------------------------------------------------------------------------
// Ball query over the cell structure an ordered FPS call left behind (fps.hip: the cloud's points counting-sorted into the
// 16 x 16 x 16 cells of its bounding box).  One wavefront per query.
//
// The index-order walk stops as soon as nsample hits are found -- after N nsample / hits points, a microsecond for the
// dense balls of these clouds -- but a ball with FEWER than nsample hits makes it scan the whole cloud (68 us at 262144
// points), and a handful of such queries set the kernel's time.  Here a query first adds up the points of the cells its
// ball can reach; if they are few (kCandMax) it tests exactly those, in cell order, and keeps the nsample SMALLEST hit
// indices (what the index-order walk would have produced): hits go to an LDS list, at 4 nsample entries the list is cut
// back to its nsample smallest (ranks by counting) and from then on only indices below the largest kept one are admitted.
// Otherwise (a ball in a very dense region, or one whose rounding reach covers many cells) it walks in index order.
//   reach: a point counts as inside when fl(-2 q.p + |q|^2 + |p|^2) <= r^2; that value differs from |q - p|^2 by at most
//   err = 2^-20 (|q| + |p|)^2 (a generous bound on its seven roundings), so every hit has |q - p|_inf <= sqrt(r^2 + err)
//   and lies in the cells of [q - R, q + R] per axis (cell numbers are monotone in the coordinate).
constexpr int kList = 512, kMaxCells = 64, kCandMax = 8192;

// leaves the min(keep, m) smallest entries of list[0..m) in ascending order in list[0..); returns how many.  m <= kList.
__device__ int rank_select(int* list, int m, int keep, int* tmp, int lane) {
    for (int e = lane; e < m; e += 64) {
        const int v = list[e];
        int r = 0;
        for (int j = 0; j < m; ++j) r += list[j] < v ? 1 : 0;   // broadcast reads; entries are distinct point indices
        tmp[e] = r;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int vals[kList / 64], ranks[kList / 64];
#pragma unroll
    for (int k = 0; k < kList / 64; ++k) {
        const int e = lane + 64 * k;
        vals[k] = e < m ? list[e] : 0;
        ranks[k] = e < m ? tmp[e] : keep;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < kList / 64; ++k)
        if (ranks[k] < keep) list[ranks[k]] = vals[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return m < keep ? m : keep;
}

__global__ __launch_bounds__(kBlock) void ball_query_cells_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                                  const float* __restrict__ new_xyz, int64_t qb, int64_t qs,
                                                                  int64_t qc, int B, int N, int S, float r2, int Keff,
                                                                  const unsigned* __restrict__ box,
                                                                  const int* __restrict__ cellstart,
                                                                  const int* __restrict__ order,
                                                                  const float* __restrict__ sorted_xyz,
                                                                  int32_t* __restrict__ out_idx) {
    __shared__ int s_list[kBlock / 64][kList];
    __shared__ int s_tmp[kBlock / 64][kList];
    __shared__ int s_cnt[kBlock / 64];
    // one WORKGROUP per query: its four wavefronts share the cells' runs (the heaviest query sets the kernel's time, and a
    // query is a chain of L2 round trips), keep a list each and merge them at the end
    constexpr int NWV = kBlock / 64;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long w = (long long)blockIdx.x;
    const int b = (int)(w / S), sq = (int)(w % S);
    const Cloud c{xyz + (int64_t)b * sb, sn, sc};
    const float* q = new_xyz + (int64_t)b * qb + (int64_t)sq * qs;
    const float qx = q[0], qy = q[qc], qz = q[2 * qc];
    const float qn = pn2::norm2(qx, qy, qz);
    int32_t* row = out_idx + ((size_t)b * S + sq) * Keff;
    int* list = s_list[wv];
    int* tmp = s_tmp[wv];
    const u64 lt = pn2::lanemask_lt();

    const pn2::CellGrid cg = pn2::cell_grid(box + b * 8);
    // largest |coordinate| of the cloud and of the query -> error bound of the expanded distance -> reach R
    float mp = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) mp = fmaxf(mp, fmaxf(fabsf(cg.lo[a]), fabsf(pn2::ord_dec(box[b * 8 + a]))));
    const float mq = fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
    const float norms = 1.7320509f * (mp + mq) * 1.0001f;                    // >= |q| + |p|
    const float R = sqrtf(r2 + 9.5367432e-7f * norms * norms) * 1.0001f + 1e-30f;
    int c0[3], c1[3];
    const float qv[3] = {qx, qy, qz};
    bool usable = R == R && R < 3.0e38f && Keff * NWV <= kList;
    const int cut_at = 4 * Keff < kList - 64 ? 4 * Keff : kList - 64;   // the list never outgrows kList (64 appended at most)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c0[a] = pn2::cell_axis(cg, a, qv[a] - R);
        c1[a] = pn2::cell_axis(cg, a, qv[a] + R);
        usable = usable && qv[a] == qv[a];
    }
    const int nx = c1[0] - c0[0] + 1, ny = c1[1] - c0[1] + 1, nz = c1[2] - c0[2] + 1;
    const int ncell = nx * ny * nz;
    usable = usable && ncell <= kMaxCells;
    const int* cs = cellstart + (size_t)b * (pn2::kCells + 1);
    // lane l < ncell: cell number l of the box -> its run of sorted positions
    int p0 = 0, p1 = 0;
    if (usable && lane < ncell) {
        const int cx = c0[0] + lane % nx, cy = c0[1] + (lane / nx) % ny, cz = c0[2] + lane / (nx * ny);
        const unsigned cell = pn2::spread3((unsigned)cx) | (pn2::spread3((unsigned)cy) << 1) | (pn2::spread3((unsigned)cz) << 2);
        p0 = cs[cell];
        p1 = cs[cell + 1];
    }
    int cand = p1 - p0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cand += __shfl_xor(cand, off, 64);
    // Which way?  The cell search costs cand / 64 steps whatever the ball holds; the index-order walk costs
    // N nsample / hits / 64 steps -- less only when the ball is really full.  hits ~ cand x (ball volume / volume of the
    // reachable cells), a crude estimate (points sit on surfaces): walk only when it promises 8 nsample hits or more.
    float cellvol = (float)ncell;
#pragma unroll
    for (int a = 0; a < 3; ++a) cellvol = cg.scale[a] > 0.0f ? cellvol / cg.scale[a] : 0.0f;
    const float rr = sqrtf(r2);
    const float est = cellvol > 0.0f ? (float)cand * (4.18879f * rr * rr * rr) / cellvol : 0.0f;
    int m = 0;   // entries in the list / hits so far (wave-uniform)
    if (usable && (cand <= kCandMax || est < 8.0f * (float)Keff)) {
        const int* ord = order + (size_t)b * N;
        const float* sx = sorted_xyz + (size_t)b * 3 * N;
        int limit = 0x7FFFFFFF;   // after the first cut: only indices below the largest kept one can still matter
        // wavefront v takes blocks v, v + 4, ... of every cell's run, the next block's loads in flight while this one is tested
        for (int ci = 0; ci < ncell; ++ci) {
            const int a0 = __builtin_amdgcn_readlane(p0, ci) + 64 * wv, a1 = __builtin_amdgcn_readlane(p1, ci);
            int nn = 0;
            float xx = 0.f, yy = 0.f, zz = 0.f;
            auto fetch = [&](int pos) {
                const int at = pos + lane;
                const size_t sa = at < a1 ? (size_t)at : (size_t)(a1 - 1);
                nn = ord[sa];
                xx = sx[sa], yy = sx[(size_t)N + sa], zz = sx[2 * (size_t)N + sa];
            };
            if (a0 < a1) fetch(a0);
            for (int pos = a0; pos < a1; pos += 64 * NWV) {
                const int cn = nn;
                const float cx = xx, cy = yy, cz = zz;
                if (pos + 64 * NWV < a1) fetch(pos + 64 * NWV);
                const bool ok = pos + lane < a1;
                const float d = pn2::sqdist(qx, qy, qz, qn, cx, cy, cz, pn2::norm2(cx, cy, cz));
                const bool hit = ok && !(d > r2) && cn < limit;
                const u64 hits = __ballot(hit);
                if (hits) {   // wave-uniform
                    if (hit) list[m + __popcll(hits & lt)] = cn;
                    m += __popcll(hits);
                    if (m >= cut_at) {   // keep the nsample smallest so far
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        m = rank_select(list, m, Keff, tmp, lane);
                        limit = list[Keff - 1];
                    }
                }
            }
        }
        // every wavefront's nsample smallest, then the nsample smallest of those
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        m = rank_select(list, m, Keff, tmp, lane);
        if (lane == 0) s_cnt[wv] = m;
        __syncthreads();
        if (wv != 0) return;
        int total = s_cnt[0];
        for (int v = 1; v < NWV; ++v) {
            const int mv = s_cnt[v];
            for (int k = lane; k < mv; k += 64) list[total + k] = s_list[v][k];
            total += mv;
        }
        m = total;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        m = rank_select(list, m, Keff, tmp, lane);
        for (int k = lane; k < m; k += 64) row[k] = list[k];
        const int first = m > 0 ? list[0] : 0;
        finish_row(c, N, row, m, first, Keff, qx, qy, qz, qn);
        return;
    }
    // ---- index order, stop at nsample hits (the loads of the next block are in flight while this one is tested)
    if (wv != 0) return;
    int first = 0;
    float nxv, nyv, nzv;
    c.load(lane < N ? lane : 0, nxv, nyv, nzv);
    for (int n0 = 0; n0 < N && m < Keff; n0 += 64) {
        const int n = n0 + lane;
        const bool ok = n < N;
        const float x = nxv, y = nyv, z = nzv;
        c.load(n + 64 < N ? n + 64 : 0, nxv, nyv, nzv);
        const float d = pn2::sqdist(qx, qy, qz, qn, x, y, z, pn2::norm2(x, y, z));
        const bool hit = ok && !(d > r2);
        const u64 hits = __ballot(hit);
        const int slot = m + __popcll(hits & lt);
        if (hit && slot < Keff) row[slot] = n;
        if (m == 0 && hits) first = n0 + (int)__builtin_ctzll(hits);
        m += __popcll(hits);
    }
    if (m > Keff) m = Keff;
    finish_row(c, N, row, m, first, Keff, qx, qy, qz, qn);
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

extern "C" int pn2_ball_query_cells_f32(const float* xyz, int64_t sb, int64_t sn, int64_t sc, const float* new_xyz, int64_t qb,
                                        int64_t qn, int64_t qc, int B, int N, int S, float r2, int nsample, const uint32_t* box,
                                        const int32_t* cellstart, const int32_t* order, const float* sorted_xyz,
                                        int32_t* out_idx, void* stream) {
    if (!xyz || !new_xyz || !out_idx || !box || !cellstart || !order || !sorted_xyz || B <= 0 || N <= 0 || S <= 0 || nsample <= 0)
        return PN2_E_BADARG;
    const int Keff = nsample < N ? nsample : N;
    const long long waves = (long long)B * S;
    if (waves > 0x7FFFFFFFll) return PN2_E_BADARG;
    // flops = 0: the cell search does not run the S x N distance tests of the index walk (its candidate count is data
    // dependent), so a "brute-force equivalent" rate would be a meaningless number in the profile; the HBM figure stays
    PN2_LAUNCH("ball_query", (double)B * (12.0 * N + 12.0 * S + 8.0 * S * Keff), 0, ball_query_cells_kernel,
               dim3((unsigned)waves), dim3(kBlock), (hipStream_t)stream, xyz, sb, sn, sc, new_xyz, qb, qn, qc, B, N, S, r2,
               Keff, box, (const int*)cellstart, (const int*)order, sorted_xyz, out_idx);
    PN2_LAUNCH_CHECK();
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
