// Cooperative MLP chains: a whole (conv -> BatchNorm -> ReLU) x n [-> max over K] chain of a deep level as ONE persistent
// launch per direction (include/pn2_hip.h "Cooperative chain launches"; replaces, for the small levels, the launch-per-layer
// path of mlp.hip -- Modules/PointNet2/blocks.py:93-98, 213-215 forward and backward).
//
// Why.  At 512 ... 8192 rows a layer is a few hundred MFLOP: 1-2 us of matrix-core time.  Run as GEMM, finalize, GEMM, ...
// every launch still pays its ramp (kernel arguments, first tile's HBM round trip), its drain and a kernel boundary: 7-27 us
// per GEMM, 4-6 us per finalize, ~46 + 92 such launches per step of the depth-4 model.  Here G workgroups (one per tile of
// the widest layer, at most one per compute unit: all co-resident) stay resident and walk the layers:
//
//   forward   for each layer: 64 x 64 tiles (four K-teams of 256 threads, the tile body of mlp_tile.h) -> Y tile + per-chunk
//             (mean, M2) partials, stored WRITE-THROUGH; grid barrier; every workgroup merges the partials of all channels it
//             is about to normalise into an LDS coefficient block (float64, one pass with a pivot) -- workgroup 0 also writes
//             the block and the running statistics to memory for the backward pass; ... ; max over K / activation pass.
//   backward  max-pool scatter (or a column-sum pass) -> BatchNorm-backward partials; then for each layer, last to first:
//             coefficients (a, b) into LDS, then ONE tile list holding the layer's weight-gradient tiles (split-K slabs) and
//             input-gradient tiles (whose epilogue leaves the next layer's BatchNorm-backward partials); grid barrier.
//
// The barrier is an arrival counter in caller-owned memory (one atomic add + polling by one lane per workgroup: 1.7 / 2.7 /
// 5.4 us at 16 / 64 / 256 workgroups, tools/ubench/gridbar.hip) -- no device-scope release fence (an L2 write-back: 40 us
// behind dirty lines): what crosses workgroups is written with sc1 stores and read with sc1 loads instead, everything else
// (weights, inputs, tensors of earlier launches) takes the ordinary cached path.  Every wait is bounded: a launch whose
// workgroups are not co-resident raises PN2_STATUS_COOP_BARRIER and drains.
#include "chain_coop.h"

#include <cstdlib>

#include "mlp_tile.h"

namespace {

constexpr int NTT = 64;                             // threads per K-team: one wavefront (the 32 x 32 tile of mlp_tile.h)
constexpr int CT = NTT * 4;                         // threads per workgroup: four K-teams
constexpr int TL = 32;                              // tile edge
constexpr int kLdsFloats = 4 * 4 * BK * (TL + 4);   // staging buffers of the four teams (re-used as their reduction buffer)
constexpr int kMaxL = pn2::coop::kMaxLayers;
constexpr int kMaxC = pn2::coop::kMaxC;
constexpr int kChunk = TL;                          // rows per statistics chunk (one wavefront's rows of a tile)

struct Sync {
    unsigned* w;        // [0] departures, [1] dead flag, phase p: [4 + 2 p] tickets handed out, [5 + 2 p] items done
    int* status;
    unsigned spin_limit;
};

// Work queues instead of a static tile-to-workgroup map.  Every phase (a layer's tile list, the pooling pass, ...) has two
// counters in the caller's sync words: tickets handed out and items DONE.  A workgroup takes items by ticket until the tickets
// run out, then waits until the phase's done-count is complete -- that wait is the layer barrier.  Nothing requires the
// launch's workgroups to be co-resident: a workgroup that is scheduled late (a GPU shared with another process, another
// stream's kernel on some compute units) finds the tickets of the early phases gone, sees them done and catches up; the
// resident ones never wait for it.  The next ticket is always requested one item ahead (and the first ticket of the next
// phase before the wait), so a claim's round trip (~1 us, a memory-side atomic) hides behind the item it follows.
struct Walker {
    Sync sy;
    int* s_slot;        // LDS: [0] the ticket being handed to the workgroup, [1] dead
    unsigned pre;       // thread 0: the ticket requested ahead
    int phase;

    __device__ __forceinline__ unsigned* ticket_ctr() const { return sy.w + 4 + 2 * phase; }
    __device__ __forceinline__ unsigned* done_ctr() const { return sy.w + 5 + 2 * phase; }
    __device__ __forceinline__ void init(const Sync& s, int* slot) {
        sy = s;
        s_slot = slot;
        phase = 0;
        pre = 0;
        if (threadIdx.x == 0) {
            slot[1] = 0;
            pre = __hip_atomic_fetch_add(ticket_ctr(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // first ticket of the current phase (requested when the previous phase ran out)
    __device__ __forceinline__ int take() {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_slot[0] = (int)pre;
            pre = __hip_atomic_fetch_add(ticket_ctr(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        return s_slot[0];
    }
    // the item just worked on is complete (its write-through stores acknowledged): count it, hand out the next ticket
    __device__ __forceinline__ int done_and_take() {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(done_ctr(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_slot[0] = (int)pre;
            pre = __hip_atomic_fetch_add(ticket_ctr(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        return s_slot[0];
    }
    // Tickets of this phase ran out: request the first ticket of the next phase, then wait until all `nitems` items are done.
    // Returns false when the launch is dead (a bounded wait ran out: an item's owner vanished -- should not happen).
    __device__ __forceinline__ bool finish(int nitems) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* d = done_ctr();
            ++phase;
            pre = __hip_atomic_fetch_add(ticket_ctr(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nitems) {
                if (++spins > sy.spin_limit) {
                    __hip_atomic_store(sy.w + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sy.status) atomicOr(sy.status, PN2_STATUS_COOP_BARRIER);
                    s_slot[1] = 1;
                    break;
                }
                if ((spins & 63u) == 0 && __hip_atomic_load(sy.w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    s_slot[1] = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        } else {
            ++phase;
        }
        __syncthreads();
        return s_slot[1] == 0;
    }
    // Last act of a launch: the workgroup that departs last leaves every counter zeroed for the next launch on this stream.
    __device__ __forceinline__ void depart(int nphases) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned d = __hip_atomic_fetch_add(sy.w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1) {
                for (int k = 0; k < 2 * nphases + 2; ++k) __hip_atomic_store(sy.w + 4 + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sy.w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};

// a (first, second) pair of one channel's partials, read past the L2 in one 8-byte load
__device__ __forceinline__ float2 ld_pair_coh(const float* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

constexpr int kFly = 32;   // partial pairs a lane has in flight in the merges
// lanes per channel for the partial merges: a power of two <= 64 so that a channel's lanes share a wavefront
__device__ __forceinline__ int lanes_per_channel(int C, int nchunk) {
    int t = 1;
    while (t < 64 && 2 * t * C <= CT && 2 * t <= nchunk) t *= 2;
    return t;
}
__device__ __forceinline__ double group_sum_f64(double v, int T) {
    for (int off = T >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Forward BatchNorm coefficients of one layer from the tile epilogues' channel-major (mean, M2) pairs (channel c, chunk k at
// partial[c * pstride + 2 k]; chunk k = rows [32 k, 32 k + 32)) -> LDS block rows ST_MEAN / ST_SCALE / ST_BETA of `coef`
// ([ST_ROWS][C]).  Every workgroup does this for itself; `writer` also stores the block and updates the running statistics
// like nn.BatchNorm.  float64, one pass: with the first chunk's mean as pivot p and d_k = m_k - p,
//   mean = p + sum(n_k d_k) / N,   var = (sum M2_k + sum n_k d_k^2 - N (mean - p)^2) / N        (Chan's merge, single pass)
__device__ __forceinline__ void finalize_fwd(const float* __restrict__ partial, long long pstride, int rows, int C, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float eps,
                             float momentum, float* __restrict__ gstats, bool writer, float* __restrict__ coef) {
    const int nchunk = (rows + kChunk - 1) / kChunk;
    const int T = lanes_per_channel(C, nchunk), per = CT / T, tid = (int)threadIdx.x;
    for (int c0 = 0; c0 < C; c0 += per) {
        const int c = c0 + tid / T, sub = tid % T;
        const bool ok = c < C;
        const float* pc = partial + (long long)(ok ? c : 0) * pstride;
        // (no dummy loads for idle lanes or short columns: these loads go past the L2, and a thousand of them on ONE address
        // serialise at the memory side -- 7-14 us per merge measured, tools/diag_coop.py)
        const float pivot = ok ? ld_pair_coh(pc).x : 0.0f;
        const int mine = ok ? (nchunk - sub + T - 1) / T : 0;   // this lane's chunks: sub, sub + T, ...
        double a = 0.0, b = 0.0, q = 0.0;
        for (int j0 = 0; j0 < mine; j0 += kFly) {   // up to kFly independent loads in flight
            float2 v[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u)
                if (j0 + u < mine) v[u] = ld_pair_coh(pc + 2 * (sub + (j0 + u) * T));
#pragma unroll
            for (int u = 0; u < kFly; ++u)
                if (j0 + u < mine) {
                    const int k = sub + (j0 + u) * T;
                    const int left = rows - k * kChunk;
                    const double n = (double)(left < kChunk ? left : kChunk);
                    const double d = (double)v[u].x - (double)pivot;
                    a += n * d;
                    b += n * d * d;
                    q += (double)v[u].y;
                }
        }
        a = group_sum_f64(a, T);
        b = group_sum_f64(b, T);
        q = group_sum_f64(q, T);
        if (ok && sub == 0) {
            const double N = (double)rows, dm = a / N, mean = (double)pivot + dm;
            double var = (q + b - N * dm * dm) / N;
            var = var > 0.0 ? var : 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = (gamma ? gamma[c] : 1.0f) * invstd, bt = beta ? beta[c] : 0.0f;
            coef[ST_MEAN * C + c] = (float)mean;
            coef[ST_SCALE * C + c] = sc;
            coef[ST_BETA * C + c] = bt;
            if (writer) {
                gstats[ST_MEAN * C + c] = (float)mean;
                gstats[ST_VAR * C + c] = (float)var;
                gstats[ST_INVSTD * C + c] = invstd;
                gstats[ST_SCALE * C + c] = sc;
                gstats[ST_BETA * C + c] = bt;
                if (rmean) {
                    const double unbiased = rows > 1 ? var * N / (N - 1.0) : var;
                    rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + (double)momentum * mean);
                    rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + (double)momentum * unbiased);
                }
            }
        }
    }
}

// Backward coefficients of one layer from channel-major (s1, s2) pairs over `nblk` row blocks: a = s1 / N, b = invstd s2 / N
// (what TR_DY needs next to the forward's mean / scale / beta, copied here from the layer's stored block) -> LDS block;
// `writer` also accumulates dbeta += s1, dgamma += s2 and stores (a, b).
__device__ __forceinline__ void finalize_bwd(const float* __restrict__ partial, long long pstride, int nblk, int rows, int C, float* __restrict__ gstats,
                             float* __restrict__ dgamma, float* __restrict__ dbeta, bool writer, float* __restrict__ coef) {
    const int T = lanes_per_channel(C, nblk), per = CT / T, tid = (int)threadIdx.x;
    for (int c0 = 0; c0 < C; c0 += per) {
        const int c = c0 + tid / T, sub = tid % T;
        const bool ok = c < C;
        const float* pc = partial + (long long)(ok ? c : 0) * pstride;
        const int mine = ok ? (nblk - sub + T - 1) / T : 0;
        double s1 = 0.0, s2 = 0.0;
        for (int j0 = 0; j0 < mine; j0 += kFly) {
            float2 v[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u)
                if (j0 + u < mine) v[u] = ld_pair_coh(pc + 2 * (sub + (j0 + u) * T));
#pragma unroll
            for (int u = 0; u < kFly; ++u)
                if (j0 + u < mine) {
                    s1 += (double)v[u].x;
                    s2 += (double)v[u].y;
                }
        }
        s1 = group_sum_f64(s1, T);
        s2 = group_sum_f64(s2, T);
        if (ok && sub == 0) {
            const float invstd = gstats[ST_INVSTD * C + c];
            const float av = (float)(s1 / (double)rows), bv = (float)((double)invstd * s2 / (double)rows);
            coef[ST_MEAN * C + c] = gstats[ST_MEAN * C + c];
            coef[ST_SCALE * C + c] = gstats[ST_SCALE * C + c];
            coef[ST_BETA * C + c] = gstats[ST_BETA * C + c];
            coef[ST_A * C + c] = av;
            coef[ST_B * C + c] = bv;
            if (writer) {
                gstats[ST_A * C + c] = av;
                gstats[ST_B * C + c] = bv;
                if (dbeta) dbeta[c] += (float)s1;
                if (dgamma) dgamma[c] += (float)s2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------ forward
struct FwdLayer {
    int cin, cout, relu, vec;
    const float *W, *bias, *gamma, *beta;
    float *rmean, *rvar;
    float eps, momentum;
    float *y, *stats;
};
struct FwdArgs {
    const float* x;
    long long ldx;
    int rows, nlayers, pool_k;
    FwdLayer L[kMaxL];
    float* out;
    int32_t* arg;
    float* partial[2];
    Sync sy;
    unsigned long long* diag;   // tuning aid (PN2_COOP_DIAG): [workgroup][phase][8] 100 MHz time stamps, or null
};

#define COOP_STAMP(slot)                                                                                      \
    do {                                                                                                      \
        if (a.diag && tid == 0) a.diag[((long long)blockIdx.x * 8 + dphase) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

template <int A_KIND, bool VEC>
__device__ __forceinline__ void fwd_tile(const GemmArgs& g, const OneSeg& st, int bx, int by, float* lds) {
    gemm_body<true, A_KIND, true, TR_PLAIN, EPI_FWD, TL, VEC, 4, false, false, true, NTT, 0>(g, st, bx, by, 0, lds);
}

// (The argument block is read through the kernarg segment pointer: a by-value struct indexed with a run-time layer number
// would be copied to private memory first -- 430 bytes of scratch per lane and every pointer in it a vector register.)
__global__ __launch_bounds__(CT, 2) void chain_coop_fwd_kernel(const FwdArgs) {
    const FwdArgs& a = *(const FwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    __shared__ __attribute__((aligned(16))) float s_coef[ST_ROWS * kMaxC];
    __shared__ int s_slot[2];
    const int tid = (int)threadIdx.x;
    Walker wk;
    wk.init(a.sy, s_slot);
    const OneSeg st{a.rows};
    const int ntx = (a.rows + TL - 1) / TL;
    const long long pstride = 2ll * ntx;             // one chunk per row tile, two floats per chunk
    int dphase = 0;
    for (int i = 0; i < a.nlayers; ++i) {
        const FwdLayer& L = a.L[i];
        const int nty = (L.cout + TL - 1) / TL, nt = ntx * nty;
        dphase = i;
        COOP_STAMP(0);
        int t = wk.take();
        COOP_STAMP(1);
        if (a.diag && tid == 0) a.diag[((long long)blockIdx.x * 8 + dphase) * 8 + 7] = (unsigned long long)t;
        if (t < nt) {
            if (i > 0) {   // the BatchNorm of the layer whose rows this one reads; ticket 0's owner keeps the books
                const FwdLayer& P = a.L[i - 1];
                finalize_fwd(a.partial[(i - 1) & 1], pstride, a.rows, P.cout, P.gamma, P.beta, P.rmean, P.rvar, P.eps, P.momentum,
                             P.stats, t == 0, s_coef);
                __syncthreads();
            }
            GemmArgs g{};
            g.A.p = i == 0 ? a.x : a.L[i - 1].y;
            g.A.ld = i == 0 ? a.ldx : (long long)a.L[i - 1].cout;
            g.A.rows = a.rows;
            g.A.cols = L.cin;
            g.A.coef = i == 0 ? nullptr : s_coef;
            g.A.cstride = L.cin;
            g.A.relu = i == 0 ? 0 : a.L[i - 1].relu;
            g.B.p = L.W;
            g.B.ld = L.cin;
            g.B.rows = L.cout;
            g.B.cols = L.cin;
            g.M = a.rows;
            g.N = L.cout;
            g.K = L.cin;
            g.C = L.y;
            g.ldc = L.cout;
            g.bias = L.bias;
            g.partial = a.partial[i & 1];
            g.pstride = pstride;
            COOP_STAMP(2);
            while (t < nt) {
                const int bx = t / nty, by = t - bx * nty;
                if (i == 0) {
                    if (L.vec) fwd_tile<TR_PLAIN, true>(g, st, bx, by, lds); else fwd_tile<TR_PLAIN, false>(g, st, bx, by, lds);
                } else {
                    if (L.vec) fwd_tile<TR_BNRELU, true>(g, st, bx, by, lds); else fwd_tile<TR_BNRELU, false>(g, st, bx, by, lds);
                }
                COOP_STAMP(3);
                t = wk.done_and_take();   // (its barriers also separate the tile's reduction buffer from the next tile's staging)
                COOP_STAMP(4);
            }
        }
        COOP_STAMP(5);
        if (!wk.finish(nt)) return;
        COOP_STAMP(6);
    }
    dphase = a.nlayers;
    COOP_STAMP(0);
    // ---- the chain's output: max over each group of pool_k rows, or the plain activation, in items of CT elements
    const FwdLayer& E = a.L[a.nlayers - 1];
    const int C = E.cout, K = a.pool_k;
    const long long total = K > 1 ? (long long)(a.rows / K) * C : (long long)a.rows * C / 4;
    const int nitems = (int)((total + CT - 1) / CT);
    int t = wk.take();
    COOP_STAMP(1);
    if (t < nitems) {
        finalize_fwd(a.partial[(a.nlayers - 1) & 1], pstride, a.rows, E.cout, E.gamma, E.beta, E.rmean, E.rvar, E.eps, E.momentum,
                     E.stats, t == 0, s_coef);
        __syncthreads();
        COOP_STAMP(2);
        while (t < nitems) {
            const long long e = (long long)t * CT + tid;
            if (e < total) {
                if (K > 1) {
                    const long long gi = e / C;
                    const int c = (int)(e - gi * C);
                    const float mean = s_coef[ST_MEAN * C + c], sc = s_coef[ST_SCALE * C + c], bt = s_coef[ST_BETA * C + c];
                    const float* p = E.y + gi * K * C + c;
                    float best = -__builtin_inff();
                    int bk = 0;
                    for (int k0 = 0; k0 < K; k0 += 32) {   // a whole group's rows in flight (each load is a trip past the L2)
                        float raw[32];
#pragma unroll
                        for (int u = 0; u < 32; ++u) raw[u] = ld_coh(p + (long long)(k0 + u < K ? k0 + u : K - 1) * C);
#pragma unroll
                        for (int u = 0; u < 32; ++u) {
                            float v = __builtin_fmaf(raw[u] - mean, sc, bt);
                            if (E.relu) v = fmaxf(v, 0.f);
                            if (k0 + u < K && v > best) {
                                best = v;
                                bk = k0 + u;
                            }
                        }
                    }
                    a.out[e] = best;
                    a.arg[e] = bk;
                } else {
                    const int c = (int)((e * 4) % C);
                    const int r = (int)((e * 4) / C);
                    const float4 v = ld4_coh<true>(E.y, C, r, c, a.rows, C);
                    float4 o;
                    o.x = __builtin_fmaf(v.x - s_coef[ST_MEAN * C + c], s_coef[ST_SCALE * C + c], s_coef[ST_BETA * C + c]);
                    o.y = __builtin_fmaf(v.y - s_coef[ST_MEAN * C + c + 1], s_coef[ST_SCALE * C + c + 1], s_coef[ST_BETA * C + c + 1]);
                    o.z = __builtin_fmaf(v.z - s_coef[ST_MEAN * C + c + 2], s_coef[ST_SCALE * C + c + 2], s_coef[ST_BETA * C + c + 2]);
                    o.w = __builtin_fmaf(v.w - s_coef[ST_MEAN * C + c + 3], s_coef[ST_SCALE * C + c + 3], s_coef[ST_BETA * C + c + 3]);
                    if (E.relu) {
                        o.x = fmaxf(o.x, 0.f);
                        o.y = fmaxf(o.y, 0.f);
                        o.z = fmaxf(o.z, 0.f);
                        o.w = fmaxf(o.w, 0.f);
                    }
                    ((float4*)a.out)[e] = o;
                }
            }
            COOP_STAMP(3);
            t = wk.done_and_take();
            COOP_STAMP(4);
        }
    }
    COOP_STAMP(5);
    wk.depart(a.nlayers + 1);
}

// ----------------------------------------------------------------------------------------------------- backward
struct BwdLayer {
    int cin, cout, relu;
    int vec_w, vec_d;              // 16-byte staging legal for the weight-gradient / input-gradient contraction
    const float* W;
    const float* y;                // [rows][cout] pre-BatchNorm rows of the forward pass
    float* stats;                  // [8][cout]
    float *dgamma, *dbeta;
    float* slab;                   // [nsplit][cout][cin] or null (no weight gradient wanted)
    int kps, nsplit;
    const float* dz;               // gradient w.r.t. this layer's ACTIVATED output rows ([rows][cout], row stride cout)
    float* dx;                     // where the input gradient goes (null: not wanted), row stride lddx, first column `skip`
    long long lddx;
    int skip;
};
struct BwdArgs {
    const float* x;
    long long ldx;
    int rows, nlayers, pool_k;
    BwdLayer L[kMaxL];
    const float* dout;
    const int32_t* arg;
    float* dz_last;                // pooled chains: the scatter's output (= L[n-1].dz)
    float* lead;                   // columns [0, nlead) of the chain's input gradient are zeroed, or null
    long long ldlead;
    int nlead;
    int gpb;                       // pooled: groups per block of the scatter
    float* partial[2];
    Sync sy;
};

template <bool VEC>
__device__ __forceinline__ void dgrad_tile(const GemmArgs& g, const OneSeg& st, int bx, int by, float* lds) {
    gemm_body<true, TR_DY, false, TR_PLAIN, EPI_STORE, TL, VEC, 4, false, false, true, NTT, 0>(g, st, bx, by, 0, lds);
}
template <int B_KIND, bool VEC>
__device__ __forceinline__ void wgrad_tile(const GemmArgs& g, const OneSeg& st, int bx, int by, int bz, float* lds) {
    gemm_body<false, TR_DY, false, B_KIND, EPI_SLAB, TL, VEC, 4, false, false, true, NTT, 0>(g, st, bx, by, bz, lds);
}

__global__ __launch_bounds__(CT, 2) void chain_coop_bwd_kernel(const BwdArgs) {
    const BwdArgs& a = *(const BwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    __shared__ __attribute__((aligned(16))) float s_coef[ST_ROWS * kMaxC];
    __shared__ int s_slot[2];
    const int tid = (int)threadIdx.x;
    Walker wk;
    wk.init(a.sy, s_slot);
    const OneSeg st{a.rows};
    const int n = a.nlayers;
    int nblk0;                     // row blocks behind the last layer's BatchNorm-backward partials
    long long pstride0;
    {
        // ---- phase 0: BatchNorm-backward sums of the LAST layer, s1 = sum mask dz, s2 = sum mask dz xhat per row block
        const BwdLayer& E = a.L[n - 1];
        const int C = E.cout, K = a.pool_k;
        const float* cf = E.stats;
        float* red = lds;
        if (K > 1) {
            // max-pool scatter dz[g K + k][c] = (k == arg[g][c]) ? dout[g][c] : 0 with the sums taken on the way: only the arg-max
            // row of a group carries a gradient
            const int groups = a.rows / K;
            nblk0 = (groups + a.gpb - 1) / a.gpb;
            pstride0 = 2ll * nblk0;
            const int cw = C > 128 ? 256 : C > 64 ? 128 : 64, ng = CT / cw;
            const int tc = tid % cw, tg = tid / cw;
            for (int blk = wk.take(); blk < nblk0; blk = wk.done_and_take()) {
                const int g0 = blk * a.gpb, g1 = g0 + a.gpb < groups ? g0 + a.gpb : groups;
                if (a.lead)   // the unused leading columns of the chain's input gradient, this block's rows
                    for (int e = tid; e < (g1 - g0) * K * a.nlead; e += CT)
                        a.lead[((long long)g0 * K + e / a.nlead) * a.ldlead + e % a.nlead] = 0.0f;
                for (int c0 = 0; c0 < C; c0 += cw) {
                    const int c = c0 + tc;
                    float s1 = 0.f, s2 = 0.f;
                    if (c < C) {
                        const float mean = cf[ST_MEAN * C + c], sc = cf[ST_SCALE * C + c], bt = cf[ST_BETA * C + c],
                                    is = cf[ST_INVSTD * C + c];
                        for (int gi = g0 + tg; gi < g1; gi += ng) {
                            const int ka = a.arg[(long long)gi * C + c];
                            const float gv = a.dout[(long long)gi * C + c];
                            const long long base = (long long)gi * K;
                            if ((unsigned)ka < (unsigned)K) {
                                const float yy = E.y[(base + ka) * C + c];
                                const float t = __builtin_fmaf(yy - mean, sc, bt);
                                const float gz = (!E.relu || t > 0.f) ? gv : 0.f;
                                s1 += gz;
                                s2 += gz * ((yy - mean) * is);
                            }
                            for (int k = 0; k < K; ++k) st_coh(&a.dz_last[(base + k) * C + c], k == ka ? gv : 0.0f);
                        }
                    }
                    __syncthreads();
                    red[tid] = s1;
                    red[CT + tid] = s2;
                    __syncthreads();
                    if (tg == 0 && c < C) {
                        for (int q = 1; q < ng; ++q) s1 += red[q * cw + tc], s2 += red[CT + q * cw + tc];
                        float* pp = a.partial[0] + (long long)c * pstride0 + 2 * blk;
                        st_coh(pp, s1);
                        st_coh(pp + 1, s2);
                    }
                }
            }
        } else {
            // no pooling: dz of the last layer IS dout; one pass over (dout, y) for the sums, blocks of 64 rows
            constexpr int RBk = 64;
            nblk0 = (a.rows + RBk - 1) / RBk;
            pstride0 = 2ll * nblk0;
            const int C4 = C / 4;
            int cw = 8;
            while (cw < C4 && cw < 256) cw *= 2;
            const int rg = CT / cw, tc = tid % cw, tr = tid / cw;
            for (int blk = wk.take(); blk < nblk0; blk = wk.done_and_take()) {
                const int r0 = blk * RBk, r1 = r0 + RBk < a.rows ? r0 + RBk : a.rows;
                if (a.lead)
                    for (int e = tid; e < (r1 - r0) * a.nlead; e += CT) a.lead[((long long)r0 + e / a.nlead) * a.ldlead + e % a.nlead] = 0.0f;
                for (int q0 = 0; q0 < C4; q0 += cw) {
                    const int c = 4 * (q0 + tc);
                    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                    if (c < C) {
                        const float4 mean = *(const float4*)(cf + ST_MEAN * C + c), sc = *(const float4*)(cf + ST_SCALE * C + c),
                                     bt = *(const float4*)(cf + ST_BETA * C + c), is = *(const float4*)(cf + ST_INVSTD * C + c);
                        const float m[4] = {mean.x, mean.y, mean.z, mean.w}, sv[4] = {sc.x, sc.y, sc.z, sc.w},
                                    b[4] = {bt.x, bt.y, bt.z, bt.w}, iv[4] = {is.x, is.y, is.z, is.w};
                        for (int r = r0 + tr; r < r1; r += rg) {
                            const float4 yv = *(const float4*)(E.y + (long long)r * C + c);
                            const float4 dv = *(const float4*)(a.dout + (long long)r * C + c);
                            const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float t = __builtin_fmaf(yy[j] - m[j], sv[j], b[j]);
                                const float gz = (!E.relu || t > 0.f) ? dd[j] : 0.f;
                                s1[j] += gz;
                                s2[j] += gz * ((yy[j] - m[j]) * iv[j]);
                            }
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        red[(j * 2 + 0) * CT + tid] = s1[j];
                        red[(j * 2 + 1) * CT + tid] = s2[j];
                    }
                    __syncthreads();
                    if (tr == 0 && c < C) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float t1 = s1[j], t2 = s2[j];
                            for (int q = 1; q < rg; ++q) t1 += red[(j * 2 + 0) * CT + q * cw + tc], t2 += red[(j * 2 + 1) * CT + q * cw + tc];
                            float* pp = a.partial[0] + (long long)(c + j) * pstride0 + 2 * blk;
                            st_coh(pp, t1);
                            st_coh(pp + 1, t2);
                        }
                    }
                }
            }
        }
        if (!wk.finish(nblk0)) return;
    }
    const int ntx = (a.rows + TL - 1) / TL;
    const long long pstride = 2ll * ntx;   // partials of the input-gradient epilogues: one chunk per row tile
    for (int i = n - 1; i >= 0; --i) {
        const BwdLayer& L = a.L[i];
        const int pi = (n - 1 - i) & 1;    // which partial buffer holds THIS layer's sums
        const int wx = (L.cout + TL - 1) / TL, wy = (L.cin + TL - 1) / TL;
        const int nw = L.slab ? wx * wy * L.nsplit : 0;
        const int dyt = (L.cin - L.skip + TL - 1) / TL;
        const int nd = L.dx ? ntx * dyt : 0;
        int t = wk.take();
        if (t < nw + nd || t == 0) {       // (ticket 0's owner keeps the books -- dgamma, dbeta -- even of a layer without tiles)
            if (i == n - 1)
                finalize_bwd(a.partial[0], pstride0, nblk0, a.rows, L.cout, L.stats, L.dgamma, L.dbeta, t == 0, s_coef);
            else
                finalize_bwd(a.partial[pi], pstride, (a.rows + kChunk - 1) / kChunk, a.rows, L.cout, L.stats, L.dgamma, L.dbeta, t == 0,
                             s_coef);
            __syncthreads();
        }
        if (t < nw + nd) {
            // dY operand: dz through this layer's BatchNorm + ReLU backward
            Operand dy{};
            dy.p = L.dz;
            dy.ld = L.cout;
            dy.q = L.y;
            dy.ldq = L.cout;
            dy.rows = a.rows;
            dy.cols = L.cout;
            dy.coef = s_coef;
            dy.cstride = L.cout;
            dy.relu = L.relu;
            // the layer's input as an activation source
            Operand in{};
            in.p = i == 0 ? a.x : a.L[i - 1].y;
            in.ld = i == 0 ? a.ldx : (long long)a.L[i - 1].cout;
            in.rows = a.rows;
            in.cols = L.cin;
            in.coef = i == 0 ? nullptr : a.L[i - 1].stats;
            in.cstride = L.cin;
            in.relu = i == 0 ? 0 : a.L[i - 1].relu;
            GemmArgs gw{};   // dW[cout][cin] = dY^T X over row ranges -> slabs
            gw.A = dy;
            gw.B = in;
            gw.M = L.cout;
            gw.N = L.cin;
            gw.K = a.rows;
            gw.C = L.slab;
            gw.ldc = L.cin;
            gw.k_per_split = L.kps;
            GemmArgs gd{};   // dX[rows][cin - skip] = dY W
            gd.A = dy;
            gd.B.p = L.W + L.skip;
            gd.B.ld = L.cin;
            gd.B.rows = L.cout;
            gd.B.cols = L.cin - L.skip;
            gd.M = a.rows;
            gd.N = L.cin - L.skip;
            gd.K = L.cout;
            gd.C = L.dx;
            gd.ldc = L.lddx;
            if (i > 0) {   // the epilogue leaves the BatchNorm-backward sums of layer i - 1
                gd.partial = a.partial[pi ^ 1];
                gd.pstride = pstride;
                gd.ey = a.L[i - 1].y;
                gd.ldey = a.L[i - 1].cout;
                gd.ecoef = a.L[i - 1].stats;
                gd.erelu = a.L[i - 1].relu;
            }
            while (t < nw + nd) {
                if (t < nd) {          // input-gradient tiles first: the next layer waits for them, the slabs for nobody
                    const int bx = t / dyt, by = t - bx * dyt;
                    if (L.vec_d) dgrad_tile<true>(gd, st, bx, by, lds); else dgrad_tile<false>(gd, st, bx, by, lds);
                } else {
                    const int u = t - nd, bz = u / (wx * wy), r = u - bz * (wx * wy), bx = r / wy, by = r - bx * wy;
                    if (i == 0) {
                        if (L.vec_w) wgrad_tile<TR_PLAIN, true>(gw, st, bx, by, bz, lds); else wgrad_tile<TR_PLAIN, false>(gw, st, bx, by, bz, lds);
                    } else {
                        if (L.vec_w) wgrad_tile<TR_BNRELU, true>(gw, st, bx, by, bz, lds); else wgrad_tile<TR_BNRELU, false>(gw, st, bx, by, bz, lds);
                    }
                }
                t = wk.done_and_take();
            }
        }
        // (every tile, not just the input gradient's: a straggling weight-gradient tile still reads the dz buffer that the next
        // layer's input-gradient tiles overwrite)
        if (i > 0 && !wk.finish(nw + nd)) return;
    }
    wk.depart(n + 1);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

Sync make_sync(const pn2_coop* ctl) {
    Sync s;
    s.w = ctl->sync;
    s.status = ctl->status;
    s.spin_limit = ctl->spin_limit ? ctl->spin_limit : (1u << 22);
    return s;
}
int max_wg(const pn2_coop* ctl) {
    int m = ctl->max_workgroups > 0 ? ctl->max_workgroups : 128;
    if (const char* e = getenv("PN2_COOP_MAX_WG")) m = atoi(e) > 0 ? atoi(e) : m;   // tuning aid
    return m < 1 ? 1 : (m > 1024 ? 1024 : m);
}

}  // namespace

namespace pn2 {
namespace coop {

bool shapes_ok(int rows, const pn2_mlp_layer* layers, int nlayers, int pool_k) {
    if (nlayers < 1 || nlayers > kMaxLayers || rows < 1) return false;
    int max_rows = PN2_COOP_MAX_ROWS;
    if (const char* e = getenv("PN2_COOP_MAX_ROWS")) max_rows = atoi(e);   // tuning aid (0: never cooperative)
    if (rows > max_rows) return false;
    for (int i = 0; i < nlayers; ++i) {
        const pn2_mlp_layer& L = layers[i];
        if (!L.has_bn || L.cout > kMaxC || L.cout % 4 || L.cin < 1 || !L.weight || !L.y || !L.stats) return false;
        if (!al16(L.y) || !al16(L.stats)) return false;
    }
    if (pool_k > 1 && (rows % pool_k || pool_k > 1024)) return false;
    return true;
}

void plan_slabs(int rows, int cout, int cin, int* kps, int* nsplit) {
    const int tiles = ceil_div(cout, TL) * ceil_div(cin, TL);
    int want = 256 / tiles;                       // ~256 weight-gradient tiles per layer next to the input-gradient tiles
    if (want < 1) want = 1;
    int k = ceil_div(ceil_div(rows, want), 4 * BK) * 4 * BK;   // whole K-tiles for each of the four teams
    if (k < 4 * BK) k = 4 * BK;
    *kps = k;
    *nsplit = ceil_div(rows, k);
}

int forward(const FwdCall& c) {
    FwdArgs a{};
    a.x = c.x;
    a.ldx = c.ldx;
    a.rows = c.rows;
    a.nlayers = c.nlayers;
    a.pool_k = c.pool_k;
    a.out = c.out;
    a.arg = c.arg;
    a.partial[0] = c.partial[0];
    a.partial[1] = c.partial[1];
    a.sy = make_sync(c.ctl);
    const int ntx = ceil_div(c.rows, TL);
    int tiles = 1;
    double flops = 0.0, bytes = 4.0 * c.rows * c.layers[0].cin;
    for (int i = 0; i < c.nlayers; ++i) {
        const pn2_mlp_layer& S = c.layers[i];
        FwdLayer& L = a.L[i];
        L.cin = S.cin, L.cout = S.cout, L.relu = S.relu;
        L.W = S.weight, L.bias = S.bias, L.gamma = S.gamma, L.beta = S.beta;
        L.rmean = S.running_mean, L.rvar = S.running_var, L.eps = S.eps, L.momentum = S.momentum;
        L.y = S.y, L.stats = S.stats;
        const bool a_ok = i == 0 ? (c.ldx % 4 == 0 && al16(c.x)) : true;
        L.vec = (S.cin % 4 == 0) && a_ok && al16(S.weight);
        const int t = ntx * ceil_div(S.cout, TL);
        tiles = t > tiles ? t : tiles;
        flops += 2.0 * c.rows * S.cin * S.cout;
        bytes += 8.0 * c.rows * S.cout;
    }
    const int cap = max_wg(c.ctl);
    const int G = tiles < cap ? tiles : cap;
    if (const char* e = getenv("PN2_COOP_DIAG")) a.diag = (unsigned long long*)strtoull(e, nullptr, 0);   // device buffer, 256 x 8 x 8 u64
    PN2_LAUNCH("chain_coop_fwd", bytes, flops, chain_coop_fwd_kernel, dim3(G), dim3(CT), c.stream, a);
    PN2_LAUNCH_CHECK();
    return 0;
}

int backward(const BwdCall& c) {
    BwdArgs a{};
    a.x = c.x;
    a.ldx = c.ldx;
    a.rows = c.rows;
    a.nlayers = c.nlayers;
    a.pool_k = c.pool_k;
    a.dout = c.dout;
    a.arg = c.arg;
    a.partial[0] = c.partial[0];
    a.partial[1] = c.partial[1];
    a.sy = make_sync(c.ctl);
    const int n = c.nlayers, ntx = ceil_div(c.rows, TL);
    int which = 0, tiles = 1;
    const float* dz = c.dout;
    if (c.pool_k > 1) {
        a.dz_last = c.scratch[which];
        dz = c.scratch[which];
        which ^= 1;
    }
    if (c.zero_lead && c.dx && c.dx_first_col > 0) {
        a.lead = c.dx;
        a.ldlead = c.lddx;
        a.nlead = c.dx_first_col;
    }
    double flops = 0.0, bytes = 0.0;
    for (int i = n - 1; i >= 0; --i) {
        const pn2_mlp_layer& S = c.layers[i];
        BwdLayer& L = a.L[i];
        L.cin = S.cin, L.cout = S.cout, L.relu = S.relu;
        L.W = S.weight, L.y = S.y, L.stats = S.stats, L.dgamma = S.dgamma, L.dbeta = S.dbeta;
        L.slab = S.dweight ? c.slab[i] : nullptr;
        L.kps = c.kps[i], L.nsplit = c.nsplit[i];
        L.dz = dz;
        L.skip = i == 0 ? c.dx_first_col : 0;
        if (i > 0) {
            L.dx = c.scratch[which];
            L.lddx = S.cin;
        } else {
            L.dx = c.dx ? c.dx + L.skip : nullptr;
            L.lddx = c.lddx;
        }
        const bool in_ok = i == 0 ? (c.ldx % 4 == 0 && al16(c.x)) : true;
        L.vec_w = (S.cout % 4 == 0) && (S.cin % 4 == 0) && in_ok && al16(dz);
        L.vec_d = (S.cout % 4 == 0) && (S.cin % 4 == 0) && ((S.cin - L.skip) % 4 == 0) && al16(S.weight + L.skip) && al16(dz);
        int t = L.dx ? ntx * ceil_div(S.cin - L.skip, TL) : 0;
        if (L.slab) t += ceil_div(S.cout, TL) * ceil_div(S.cin, TL) * L.nsplit;
        tiles = t > tiles ? t : tiles;
        flops += 4.0 * c.rows * S.cin * S.cout;
        bytes += 4.0 * c.rows * (3.0 * S.cout + 2.0 * S.cin);
        if (i > 0) {
            dz = L.dx;
            which ^= 1;
        }
    }
    const int cap = max_wg(c.ctl);
    const int G = tiles < cap ? tiles : cap;
    if (c.pool_k > 1) {   // groups per scatter block: every workgroup one block, and no more blocks than the partial buffer has chunks
        a.gpb = ceil_div(c.rows / c.pool_k, G);
        const int floor_gpb = ceil_div(kChunk, c.pool_k);
        if (a.gpb < floor_gpb) a.gpb = floor_gpb;
    }
    PN2_LAUNCH("chain_coop_bwd", bytes, flops, chain_coop_bwd_kernel, dim3(G), dim3(CT), c.stream, a);
    PN2_LAUNCH_CHECK();
    return 0;
}

}  // namespace coop
}  // namespace pn2
