// Neighbour search through a hashed cell grid -- the fast path behind pn2_knn_radius_f64 (Modules/Features.py:111-175).
//
// The brute-force scan in features.hip tests every pair (46 ms for 262144 points).  Here the cloud is binned once into
// hashed cubic cells (float64 like everything on this path):
//   kg_bbox / kg_init            bounding box, first cell edge from the box volume
//   kg_count / kg_decide (x3)    histogram of the cells + refinement of the edge until a cell holds ~6 points on
//                                average (a tree fills a few percent of its box, the box-derived edge is far too large)
//   kg_scan / kg_scatter         counting sort: points (coordinates, index, cell key) grouped by hash bucket
//   kg_knn_kernel                one WAVEFRONT per query: the 27 cells around the query are flattened into one
//                                candidate list (no lane divergence), every lane keeps a sorted 4-entry list of its
//                                candidates and the k nearest are popped with k wavefront arg-min reductions
//   kg_count_kernel              radius count over the 27 cells of a second grid whose edge is the radius
// Exactness: a point outside the 27 cells differs from the query by more than one cell edge h along some axis, so a
// result whose k-th distance is below h^2 (1 - 1e-9) cannot be improved from outside; distances are the brute-force
// kernel's expression bit for bit and ties are ordered by (distance, index) exactly like the scan.  A query that fails the
// test (sparse surroundings), sees more than 8192 candidate entries, or whose popping exhausts a lane that had to drop
// candidates is redone by kg_slow_kernel: its wavefront scans the whole cloud with full k-entry lists per lane.
#include "pn2_common.h"

namespace {

using u64 = unsigned long long;
constexpr int kBlock = 256;
constexpr int kMaxK = 16;
constexpr int kLane = 4;  // candidates a lane keeps on the fast path
constexpr int kMaxCand = 8192;  // candidate entries of the 27 buckets one wavefront is willing to scan

struct GridHdr {
    u64 bmin[3], bmax[3];   // ordered-integer images of the bounding box (atomicMin / atomicMax)
    double ox, oy, oz, h, inv_h, extent;
    unsigned occupied, unresolved;
    int rounds_left, pad;
};

// survives the grid rebuilds between passes: the unsettled queries of every pass and the previous pass's cell edge
struct Ctl {
    unsigned n_todo[4];
    double prev_h;
    double pad;
};

__device__ __forceinline__ u64 ord(double v) {  // monotone double -> u64
    const u64 b = (u64)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | (1ull << 63));
}
__device__ __forceinline__ double unord(u64 k) {
    const u64 b = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ void kg_bbox(const double* __restrict__ pts, int N, GridHdr* hdr) {
    __shared__ double smin[3][kBlock / 64], smax[3][kBlock / 64];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = pts[(size_t)i * 3 + a];
            mn[a] = fmin(mn[a], v);
            mx[a] = fmax(mx[a], v);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fmin(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmax(mx[a], __shfl_xor(mx[a], off, 64));
        }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) smin[a][w] = mn[a], smax[a][w] = mx[a];
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        double lo = smin[a][0], hi = smax[a][0];
        for (int j = 1; j < kBlock / 64; ++j) lo = fmin(lo, smin[a][j]), hi = fmax(hi, smax[a][j]);
        atomicMin(&hdr->bmin[a], ord(lo));
        atomicMax(&hdr->bmax[a], ord(hi));
    }
}

// first edge: ~8 points per cell if the cloud filled its box; or the caller's fixed edge (radius grid)
__global__ void kg_init(GridHdr* hdr, int N, double fixed_h, int rounds, Ctl* ctl, double scale_prev) {
    if (scale_prev > 0.0) fixed_h = ctl->prev_h * scale_prev;  // coarser grid for the queries the last one left open
    const double lo[3] = {unord(hdr->bmin[0]), unord(hdr->bmin[1]), unord(hdr->bmin[2])};
    const double hi[3] = {unord(hdr->bmax[0]), unord(hdr->bmax[1]), unord(hdr->bmax[2])};
    const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const double emax = fmax(fmax(ex, ey), fmax(ez, 1e-300));
    const double fl = emax / 1024.0;
    double h = fixed_h > 0.0 ? fixed_h : cbrt(fmax(ex, fl) * fmax(ey, fl) * fmax(ez, fl) * 8.0 / (double)N);
    h = fmax(h, emax / 1048575.0);  // 20-bit cell coordinates
    hdr->ox = lo[0], hdr->oy = lo[1], hdr->oz = lo[2];
    hdr->h = h, hdr->inv_h = 1.0 / h, hdr->extent = emax;
    hdr->occupied = 0, hdr->unresolved = 0;
    hdr->rounds_left = fixed_h > 0.0 ? 0 : rounds;
}

__global__ void kg_save_edge(const GridHdr* hdr, Ctl* ctl) { ctl->prev_h = hdr->h; }

__device__ __forceinline__ void cell_of(const GridHdr& g, double x, double y, double z, int& cx, int& cy, int& cz) {
    cx = (int)fmin(fmax((x - g.ox) * g.inv_h, 0.0), 1048575.0);
    cy = (int)fmin(fmax((y - g.oy) * g.inv_h, 0.0), 1048575.0);
    cz = (int)fmin(fmax((z - g.oz) * g.inv_h, 0.0), 1048575.0);
}
__device__ __forceinline__ u64 cell_key(int x, int y, int z) { return ((u64)z << 42) | ((u64)y << 21) | (u64)x; }
__device__ __forceinline__ unsigned bucket_of(int x, int y, int z, unsigned mask) {
    return (((unsigned)x * 73856093u) ^ ((unsigned)y * 19349663u) ^ ((unsigned)z * 83492791u)) & mask;
}

__global__ void kg_count(const double* __restrict__ pts, int N, GridHdr* hdr, int* __restrict__ counts, unsigned mask) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const GridHdr g = *hdr;
    int cx, cy, cz;
    cell_of(g, pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2], cx, cy, cz);
    if (atomicAdd(&counts[bucket_of(cx, cy, cz, mask)], 1) == 0) atomicAdd(&hdr->occupied, 1u);
}

// shrink the edge while a bucket holds more than ~10 points on average; `final` = no further round follows
__global__ void kg_decide(GridHdr* hdr, int N) {
    if (hdr->rounds_left > 0) {
        const double m = (double)N / (double)(hdr->occupied > 0 ? hdr->occupied : 1);
        if (m > 10.0) {
            double h = hdr->h * fmax(0.35, pow(6.0 / m, 0.45));
            h = fmax(h, hdr->extent / 1048575.0);
            hdr->h = h, hdr->inv_h = 1.0 / h;
        } else {
            hdr->rounds_left = 1;  // settled: the remaining rounds recount the same grid
        }
        hdr->rounds_left -= 1;
    }
    hdr->occupied = 0;
}

// exclusive scan of counts[T] -> start[T + 1]; one workgroup
__global__ __launch_bounds__(1024) void kg_scan(const int* __restrict__ counts, int T, int* __restrict__ start) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (T + 1023) / 1024;
    const int b0 = t * per, b1 = (b0 + per) < T ? (b0 + per) : T;
    int sum = 0;
    for (int i = b0; i < b1; ++i) sum += counts[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = b0; i < b1; ++i) {
        start[i] = run;
        run += counts[i];
    }
    if (t == 1023) start[T] = part[1023];
}

__global__ void kg_scatter(const double* __restrict__ pts, int N, const GridHdr* hdr, const int* __restrict__ start,
                           int* __restrict__ cursor, unsigned mask, double* __restrict__ sx, double* __restrict__ sy,
                           double* __restrict__ sz, int* __restrict__ sidx, u64* __restrict__ skey) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const GridHdr g = *hdr;
    const double x = pts[(size_t)i * 3], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
    int cx, cy, cz;
    cell_of(g, x, y, z, cx, cy, cz);
    const unsigned b = bucket_of(cx, cy, cz, mask);
    const int pos = start[b] + atomicAdd(&cursor[b], 1);
    sx[pos] = x, sy[pos] = y, sz[pos] = z;
    sidx[pos] = i;
    skey[pos] = cell_key(cx, cy, cz);
}

// wavefront arg-min of (d, idx), d >= 0: the owner lane and whether any finite entry exists
__device__ __forceinline__ int wave_argmin(double d, int idx, bool& any) {
    const u64 bits = (u64)__double_as_longlong(d);
    const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
    const unsigned mh = pn2::wave_min_u32(hi);
    any = mh < 0x7FF00000u;  // below +inf
    u64 who = __ballot(hi == mh);
    if (__popcll(who) > 1) {
        const unsigned ml = pn2::wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
        who = __ballot(hi == mh && lo == ml);
        if (__popcll(who) > 1) {
            const unsigned mi = pn2::wave_min_u32((hi == mh && lo == ml) ? (unsigned)idx : 0xFFFFFFFFu);
            who = __ballot(hi == mh && lo == ml && (unsigned)idx == mi);
        }
    }
    return (int)__builtin_ctzll(who);
}

__device__ __forceinline__ bool lt_key(double d, int s, double dk, int ik) { return d < dk || (d == dk && s < ik); }

template <int K>
__global__ __launch_bounds__(kBlock) void kg_knn_kernel(const double* __restrict__ pts, int N, int k, const GridHdr* hdr,
                                                        const int* __restrict__ start, unsigned mask,
                                                        const double* __restrict__ sx, const double* __restrict__ sy,
                                                        const double* __restrict__ sz, const int* __restrict__ sidx,
                                                        const u64* __restrict__ skey, int32_t* __restrict__ nn_idx,
                                                        double* __restrict__ nn_d2, const int* __restrict__ todo_in,
                                                        const unsigned* __restrict__ n_in, int* __restrict__ todo_out,
                                                        unsigned* __restrict__ n_out) {
    __shared__ int s_excl[kBlock / 64][28], s_beg[kBlock / 64][27];
    __shared__ u64 s_ckey[kBlock / 64][27];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * (kBlock / 64) + wv;
    if (w >= (todo_in ? (int)*n_in : N)) return;
    const int q = todo_in ? todo_in[w] : w;
    const GridHdr g = *hdr;
    const double qx = pts[(size_t)q * 3], qy = pts[(size_t)q * 3 + 1], qz = pts[(size_t)q * 3 + 2];
    int cx, cy, cz;
    cell_of(g, qx, qy, qz, cx, cy, cz);
    // ---- the 27 neighbour cells: lane c < 27 looks up its bucket
    int beg = 0, cnt = 0;
    u64 ckey = 0;
    if (lane < 27) {
        const int x = cx + lane % 3 - 1, y = cy + (lane / 3) % 3 - 1, z = cz + lane / 9 - 1;
        if (x >= 0 && y >= 0 && z >= 0 && x <= 1048575 && y <= 1048575 && z <= 1048575) {
            const unsigned b = bucket_of(x, y, z, mask);
            beg = start[b];
            cnt = start[b + 1] - beg;
            ckey = cell_key(x, y, z);
        }
    }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane < 27) {
        s_excl[wv][lane] = incl - cnt;
        s_beg[wv][lane] = beg;
        s_ckey[wv][lane] = ckey;
    }
    if (lane == 26) s_excl[wv][27] = incl;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int M = s_excl[wv][27];
    bool fail = M > kMaxCand || M < k;
    // ---- candidates, flattened: lane handles j = it * 64 + lane
    double ld[kLane];
    int li[kLane];
#pragma unroll
    for (int s = 0; s < kLane; ++s) ld[s] = __builtin_inf(), li[s] = 0x7FFFFFFF;
    bool dropped = false;  // this lane had to discard a candidate (it saw more than kLane)
    if (!fail) {
        for (int j = lane; j < M; j += 64) {
            int c = 0;  // largest cell with s_excl[c] <= j
#pragma unroll
            for (int step = 16; step > 0; step >>= 1) {
                const int t = c + step;
                if (t < 27 && s_excl[wv][t] <= j) c = t;
            }
            const int pos = s_beg[wv][c] + (j - s_excl[wv][c]);
            if (skey[pos] != s_ckey[wv][c]) continue;  // another cell hashed into this bucket
            const double dx = sx[pos] - qx, dy = sy[pos] - qy, dz = sz[pos] - qz;
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
            const int s = sidx[pos];
            dropped = dropped || li[kLane - 1] != 0x7FFFFFFF;  // list already full: this one or the last one goes
            // sorted insertion into the lane's kLane slots by (d, index)
#pragma unroll
            for (int u = kLane - 1; u >= 1; --u) {
                const bool below = lt_key(d, s, ld[u - 1], li[u - 1]);
                const bool here = lt_key(d, s, ld[u], li[u]);
                const double nd = below ? ld[u - 1] : (here ? d : ld[u]);
                const int ni = below ? li[u - 1] : (here ? s : li[u]);
                ld[u] = nd, li[u] = ni;
            }
            if (lt_key(d, s, ld[0], li[0])) ld[0] = d, li[0] = s;
        }
    }
    // ---- pop the k nearest.  Candidates are dealt to lanes round-robin, so a lane rarely owns more than one or two
    // of them; should a lane run empty after discarding candidates, the result is not trusted (slow path).
    double last = 0.0;
    int popped = 0;
    if (!fail) {
        for (int t = 0; t < k; ++t) {
            bool any;
            const int owner = wave_argmin(ld[0], li[0], any);
            if (!any) {  // fewer real candidates than k (hash collisions filtered out)
                fail = true;
                break;
            }
            if (lane == owner) {
                nn_idx[(size_t)q * k + t] = li[0];
                if (nn_d2) nn_d2[(size_t)q * k + t] = ld[0];
                last = ld[0];
                ++popped;
#pragma unroll
                for (int u = 0; u < kLane - 1; ++u) ld[u] = ld[u + 1], li[u] = li[u + 1];
                ld[kLane - 1] = __builtin_inf(), li[kLane - 1] = 0x7FFFFFFF;
            }
            last = __shfl(last, owner, 64);
        }
        if (__ballot(dropped && popped == kLane)) fail = true;
    }
    const double reach = g.h * (1.0 - 1e-9);
    if (!fail && !(last < reach * reach)) fail = true;
    if (fail && lane == 0) todo_out[atomicAdd(n_out, 1u)] = q;
}

// ---- queries the grid could not settle: the wavefront scans the whole cloud, every lane with a full k-entry list
template <int K>
__global__ __launch_bounds__(kBlock) void kg_slow_kernel(const double* __restrict__ pts, int N, int k,
                                                         const unsigned* __restrict__ n_todo, const int* __restrict__ todo,
                                                         int32_t* __restrict__ nn_idx, double* __restrict__ nn_d2) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned total = *n_todo;
    for (unsigned w = blockIdx.x * (kBlock / 64) + wv; w < total; w += gridDim.x * (kBlock / 64)) {
        const int q = todo[w];
        const double qx = pts[(size_t)q * 3], qy = pts[(size_t)q * 3 + 1], qz = pts[(size_t)q * 3 + 2];
        double ld[K];
        int li[K];
#pragma unroll
        for (int s = 0; s < K; ++s) ld[s] = __builtin_inf(), li[s] = 0x7FFFFFFF;
        for (int n = lane; n < N; n += 64) {
            const double dx = pts[(size_t)n * 3] - qx, dy = pts[(size_t)n * 3 + 1] - qy, dz = pts[(size_t)n * 3 + 2] - qz;
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
            if (d < ld[K - 1]) {  // ascending n within a lane: strict '<' keeps the lower index on ties
#pragma unroll
                for (int u = K - 1; u >= 1; --u) {
                    const bool below = d < ld[u - 1], here = d < ld[u];
                    const double nd = below ? ld[u - 1] : (here ? d : ld[u]);
                    const int ni = below ? li[u - 1] : (here ? n : li[u]);
                    ld[u] = nd, li[u] = ni;
                }
                if (d < ld[0]) ld[0] = d, li[0] = n;
            }
        }
        for (int t = 0; t < k; ++t) {
            bool any;
            const int owner = wave_argmin(ld[0], li[0], any);
            if (lane == owner) {
                nn_idx[(size_t)q * k + t] = li[0];
                if (nn_d2) nn_d2[(size_t)q * k + t] = ld[0];
#pragma unroll
                for (int u = 0; u < K - 1; ++u) ld[u] = ld[u + 1], li[u] = li[u + 1];
                ld[K - 1] = __builtin_inf(), li[K - 1] = 0x7FFFFFFF;
            }
        }
    }
}

// radius count on a grid whose edge is >= the radius: thread per query, 27 cells
__global__ __launch_bounds__(kBlock) void kg_count_kernel(const double* __restrict__ pts, int N, double r2, const GridHdr* hdr,
                                                          const int* __restrict__ start, unsigned mask,
                                                          const double* __restrict__ sx, const double* __restrict__ sy,
                                                          const double* __restrict__ sz, const u64* __restrict__ skey,
                                                          int32_t* __restrict__ count) {
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= N) return;
    const GridHdr g = *hdr;
    const double qx = pts[(size_t)q * 3], qy = pts[(size_t)q * 3 + 1], qz = pts[(size_t)q * 3 + 2];
    int cx, cy, cz;
    cell_of(g, qx, qy, qz, cx, cy, cz);
    int c = 0;
    for (int n = 0; n < 27; ++n) {
        const int x = cx + n % 3 - 1, y = cy + (n / 3) % 3 - 1, z = cz + n / 9 - 1;
        if (x < 0 || y < 0 || z < 0 || x > 1048575 || y > 1048575 || z > 1048575) continue;
        const unsigned b = bucket_of(x, y, z, mask);
        const u64 key = cell_key(x, y, z);
        const int e1 = start[b + 1];
        for (int e = start[b]; e < e1; ++e) {
            if (skey[e] != key) continue;
            const double dx = sx[e] - qx, dy = sy[e] - qy, dz = sz[e] - qz;
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
            c += d <= r2 ? 1 : 0;
        }
    }
    count[q] = c;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int table_size(int N) {
    int t = 1024;
    while (t < N / 2 && t < (1 << 22)) t <<= 1;
    return t;
}

struct Layout {
    size_t hdr, ctl, counts, start, cursor, sx, sy, sz, sidx, skey, todo, todo2, total;
};
inline Layout layout(int N) {
    const int T = table_size(N);
    Layout L{};
    size_t o = 0;
    L.hdr = o, o += align256(sizeof(GridHdr));
    L.ctl = o, o += align256(sizeof(Ctl));
    L.counts = o, o += align256((size_t)T * 4);
    L.start = o, o += align256((size_t)(T + 1) * 4);
    L.cursor = o, o += align256((size_t)T * 4);
    L.sx = o, o += align256((size_t)N * 8);
    L.sy = o, o += align256((size_t)N * 8);
    L.sz = o, o += align256((size_t)N * 8);
    L.sidx = o, o += align256((size_t)N * 4);
    L.skey = o, o += align256((size_t)N * 8);
    L.todo = o, o += align256((size_t)N * 4);
    L.todo2 = o, o += align256((size_t)N * 4);
    L.total = o;
    return L;
}

// bin the cloud; fixed_h > 0 = that edge, else adaptive
int build_grid(const double* pts, int N, double fixed_h, double scale_prev, char* ws, hipStream_t s) {
    const Layout L = layout(N);
    const int T = table_size(N);
    const unsigned mask = (unsigned)T - 1u;
    GridHdr* hdr = (GridHdr*)(ws + L.hdr);
    int* counts = (int*)(ws + L.counts);
    int* start = (int*)(ws + L.start);
    int* cursor = (int*)(ws + L.cursor);
    const dim3 grid(pn2::ceil_div(N, kBlock)), block(kBlock);
    PN2_HIP_CHECK(hipMemsetAsync(hdr, 0xFF, 3 * sizeof(u64), s));                       // bmin = +max
    PN2_HIP_CHECK(hipMemsetAsync((char*)hdr + 3 * sizeof(u64), 0, sizeof(GridHdr) - 3 * sizeof(u64), s));
    PN2_LAUNCH("knn_grid_build", 24.0 * N, 0, kg_bbox, dim3(256), block, s, pts, N, hdr);
    const int rounds = (fixed_h > 0.0 || scale_prev > 0.0) ? 0 : 3;
    PN2_LAUNCH("knn_grid_build", 0, 0, kg_init, dim3(1), dim3(1), s, hdr, N, fixed_h, rounds, (Ctl*)(ws + L.ctl), scale_prev);
    for (int r = 0; r <= rounds; ++r) {
        PN2_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)T * 4, s));
        PN2_LAUNCH("knn_grid_build", 24.0 * N, 0, kg_count, grid, block, s, pts, N, hdr, counts, mask);
        if (r < rounds) PN2_LAUNCH("knn_grid_build", 0, 0, kg_decide, dim3(1), dim3(1), s, hdr, N);
    }
    PN2_LAUNCH("knn_grid_build", 8.0 * T, 0, kg_scan, dim3(1), dim3(1024), s, (const int*)counts, T, start);
    PN2_HIP_CHECK(hipMemsetAsync(cursor, 0, (size_t)T * 4, s));
    PN2_LAUNCH("knn_grid_build", 64.0 * N, 0, kg_scatter, grid, block, s, pts, N, (const GridHdr*)hdr, (const int*)start, cursor,
               mask, (double*)(ws + L.sx), (double*)(ws + L.sy), (double*)(ws + L.sz), (int*)(ws + L.sidx), (u64*)(ws + L.skey));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" size_t pn2_knn_grid_workspace_bytes(int N) { return N > 0 ? layout(N).total : 0; }

// same contract as pn2_knn_radius_f64 (include/pn2_hip.h); r2 < 0 = no radius count
extern "C" int pn2_knn_radius_grid_f64(const double* points, int N, int k, double r2, int32_t* nn_idx, double* nn_d2,
                                       int32_t* radius_count, void* workspace, size_t workspace_bytes, void* stream) {
    if (!points || !nn_idx || !workspace || N <= 0 || k <= 0 || k > kMaxK || k > N) return PN2_E_BADARG;
    if (workspace_bytes < pn2_knn_grid_workspace_bytes(N)) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const Layout L = layout(N);
    const unsigned mask = (unsigned)table_size(N) - 1u;
    GridHdr* hdr = (GridHdr*)(ws + L.hdr);
    Ctl* ctl = (Ctl*)(ws + L.ctl);
    PN2_HIP_CHECK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
    const dim3 qgrid(pn2::ceil_div(N, kBlock / 64)), block(kBlock);
    const double qbytes = (double)N * (24.0 + 12.0 * k + 27.0 * 8.0);
    int* todo[2] = {(int*)(ws + L.todo), (int*)(ws + L.todo2)};
    // pass 0: every query on the adaptive grid; passes 1, 2: what is left on grids 3x and 9x coarser (sparse
    // surroundings: the 27 cells did not reach the k-th neighbour); the rest by full scan
    int st = 0;
    for (int pass = 0; pass < 3; ++pass) {
        st = build_grid(points, N, -1.0, pass == 0 ? -1.0 : 3.0, ws, s);
        if (st) return st;
        PN2_LAUNCH("knn_grid_build", 0, 0, kg_save_edge, dim3(1), dim3(1), s, (const GridHdr*)hdr, ctl);
        const int* tin = pass == 0 ? nullptr : todo[(pass - 1) & 1];
        const unsigned* nin = pass == 0 ? nullptr : &ctl->n_todo[pass - 1];
#define PN2_KG_ARGS points, N, k, (const GridHdr*)hdr, (const int*)(ws + L.start), mask, (const double*)(ws + L.sx),            \
                    (const double*)(ws + L.sy), (const double*)(ws + L.sz), (const int*)(ws + L.sidx), (const u64*)(ws + L.skey), \
                    nn_idx, nn_d2, tin, nin, todo[pass & 1], &ctl->n_todo[pass]
        if (k <= 8)
            PN2_LAUNCH("knn_grid", qbytes, 0, (kg_knn_kernel<8>), qgrid, block, s, PN2_KG_ARGS);
        else
            PN2_LAUNCH("knn_grid", qbytes, 0, (kg_knn_kernel<16>), qgrid, block, s, PN2_KG_ARGS);
#undef PN2_KG_ARGS
    }
    if (k <= 8)
        PN2_LAUNCH("knn_grid_slow", 0, 0, (kg_slow_kernel<8>), dim3(1024), block, s, points, N, k, (const unsigned*)&ctl->n_todo[2],
                   (const int*)todo[0], nn_idx, nn_d2);
    else
        PN2_LAUNCH("knn_grid_slow", 0, 0, (kg_slow_kernel<16>), dim3(1024), block, s, points, N, k, (const unsigned*)&ctl->n_todo[2],
                   (const int*)todo[0], nn_idx, nn_d2);
    PN2_LAUNCH_CHECK();
    if (radius_count && r2 >= 0.0) {
        // second grid, edge = radius (a hair more): the 27 cells then hold every point within the radius
        const double h = sqrt(r2) * (1.0 + 1e-9) + 1e-300;
        st = build_grid(points, N, h, -1.0, ws, s);
        if (st) return st;
        PN2_LAUNCH("knn_grid_count", (double)N * 28.0, 0, kg_count_kernel, dim3(pn2::ceil_div(N, kBlock)), block, s, points, N, r2,
                   (const GridHdr*)hdr, (const int*)(ws + L.start), mask, (const double*)(ws + L.sx), (const double*)(ws + L.sy),
                   (const double*)(ws + L.sz), (const u64*)(ws + L.skey), radius_count);
        PN2_LAUNCH_CHECK();
    }
    return 0;
}
