// farthest_point_sample for gfx950 -- replaces Modules/PointNet2/pointnet2_utils.py:66-89.
//
// FPS is npoint strictly sequential argmax steps, so the bound is the latency of one step, not bandwidth.
// Layout: every cloud is owned by a *group* of G workgroups (G = 1 for small clouds).  Each thread keeps
// PPT points and their running minimum distance in registers for the whole kernel (xyz is read from HBM
// exactly once: 12 B/point); a step is
//     register update -> wave max (shuffles) -> LDS max over the workgroup's waves
//     -> [G > 1] four 8-byte granules per workgroup {key | x | y | z}, stored write-through and polled by one
//        wave of every member workgroup (data-is-the-flag hand-off, placement independent, no grid barrier).
// The winner's coordinates travel WITH the key (registers -> LDS -> granules), so the next step never waits
// on a dependent global load.
// The argmax key is (dist_bits << 32) | ~index: distances are non-negative so their bit patterns order like
// the values, and the complemented index makes the LOWEST index win ties exactly like torch.max does
// (zero-padded clouds have many identical points).
//
// Granules are never reused inside a launch (one slot per (cloud, step, member)), the slot array is zeroed
// by a memset node ahead of the kernel, every granule carries its own validity (bit 63 of the key granule,
// the step number in the coordinate granules) and is written by ONE 8-byte store, and every spin is bounded.
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include "pn2_cells.h"
#include "pn2_common.h"

namespace {

using pn2::cell_grid;
using pn2::cell_of;
using pn2::CellGrid;
using pn2::kCells;
using pn2::ord_dec;
using pn2::ord_enc;

using u64 = unsigned long long;
using f2 = __attribute__((ext_vector_type(2))) float;
using lds_f32 = __attribute__((address_space(3))) float;
constexpr u64 kValid = 1ull << 63;
constexpr int kMaxG = 64;
constexpr unsigned kSpinLimitDefault = 1u << 22;   // polls before a hand-off is declared dead (PN2_FPS_SPIN_LIMIT overrides)
// status bits OR-ed into the caller's status word when a launch dies (include/pn2_hip.h: PN2_STATUS_*)
constexpr int kStatusHandoff = PN2_STATUS_FPS_HANDOFF, kStatusArrival = PN2_STATUS_FPS_ARRIVAL;

// Launch-wide knobs passed by value to every multi-workgroup kernel.
struct Knobs {
    unsigned spin_limit;   // bounded spins: a member that never shows up kills the launch instead of hanging it
    int force_fallback;    // PN2_FPS_FORCE_FALLBACK: take the placement-independent grouping even when XCD-local groups exist
    int* status;           // caller's sticky status word (device), may be null
};

// Rows a dead launch never produces must not look like samples.  The XCD kernels fill ALL output rows with -1 (indices) /
// NaN (centroids) before their grid-wide arrival count -- workgroup i takes clouds i, i + grid, ... --, so the fill is
// complete and released before any workgroup learns its role and writes a sample; a launch that dies in the arrival phase
// leaves the fill behind.  (In the kernel rather than as two memset nodes ahead of it: those cost ~30 us per call.)
__device__ __forceinline__ void poison_rows(int32_t* out_idx, float* out_xyz, size_t first, int count, int tid, int nthreads) {
    for (int e = tid; e < count; e += nthreads) out_idx[first + e] = -1;
    if (out_xyz)
        for (int e = tid; e < 3 * count; e += nthreads) out_xyz[3 * first + e] = __uint_as_float(0xFFFFFFFFu);
}

// One poll of a bounded spin.  Returns false when the launch is dead: this waiter ran out of polls (it raises the
// launch's error word, which every other waiter of the launch polls) or somebody else already did.  A dead launch is
// never waited on again: every workgroup leaves at its next barrier, the rows it did not produce keep the -1 / NaN
// fill written ahead of the kernel, and the caller's status word says why (ops.check_status raises on it).
__device__ __forceinline__ bool spin_alive(unsigned& spins, const Knobs& kn, unsigned* err, int why) {
    if (++spins > kn.spin_limit) {
        atomicOr(err, 1u);
        if (kn.status) atomicOr(kn.status, why);
        return false;
    }
    if ((spins & 127u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    return true;
}
#ifdef PN2_FPS_DIAG
constexpr size_t kHdr = 256;
#else
constexpr size_t kHdr = 16;
#endif

__device__ __forceinline__ u64 ld_granule(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for the
// round trip of the (fire-and-forget) result stores of the step, which costs more than the step itself.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ u64 fps_key(float d, int n) {
    return d < 0.0f ? 0ull : (((u64)__float_as_uint(d)) << 32) | (u64)(0xFFFFFFFFu - (unsigned)n);
}
// wavefront max of a 64-bit key and the lane that holds it; the low word is only reduced when the high words tie
__device__ __forceinline__ u64 wave_max_key_owner(u64 k, int& owner) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned mh = pn2::wave_max_u32(hi);
    u64 who = __ballot(hi == mh);
    unsigned ml;
    if (__popcll(who) == 1) {
        owner = (int)__builtin_ctzll(who);
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, owner);
    } else {
        ml = pn2::wave_max_u32(hi == mh ? lo : 0u);
        who = __ballot(hi == mh && lo == ml);
        owner = (int)__builtin_ctzll(who);
    }
    return ((u64)mh << 32) | ml;
}
__device__ __forceinline__ u64 wave_max_key_only(u64 k) {
    int owner;
    return wave_max_key_owner(k, owner);
}

// GROUPED = false: one workgroup per cloud (G == 1) -- the exchange code is not even compiled in, the loop stays tight.
// RAGGED: per-cloud offsets (whole-tree batches, G == 1 only); compiled separately so that the regular kernels keep
// wave-uniform (scalar) cloud sizes and strides.
template <int PPT, int T, bool GROUPED, bool RAGGED = false>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                int B, int N, int npoint, const int64_t* __restrict__ start,
                                                int32_t* __restrict__ out_idx, float* __restrict__ out_xyz,
                                                u64* gran, unsigned* err, int G, int groups, Knobs kn, const int* coff) {
    constexpr int NW = T / 64;
    __shared__ u64 s_key[2][NW];
    __shared__ float s_xyz[2][NW][3];
    __shared__ u64 s_win[2];
    __shared__ float s_wxyz[2][3];
    __shared__ int s_dead;

    const int tid = threadIdx.x;
    if (tid == 0) s_dead = 0;
    __syncthreads();
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = blockIdx.x % G;
    const int grp = blockIdx.x / G;
    const int base = g * (T * PPT) + tid;

    for (int b = grp; b < B; b += groups) {
        const pn2::CloudView cv = RAGGED ? pn2::cloud_view(xyz, sb, sn, sc, N, coff, b, 3)
                                         : pn2::CloudView{xyz + (int64_t)b * sb, sn, sc, N};
        const float* p = cv.p;
        float x[PPT], y[PPT], z[PPT], d[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int n = base + j * T;
            const bool ok = n < cv.n;
            const float* q = p + (int64_t)(ok ? n : 0) * cv.sn;
            x[j] = q[0];
            y[j] = q[cv.sc];
            z[j] = q[2 * cv.sc];
            d[j] = ok ? 1e10f : -1.0f;  // -1 marks a slot beyond N: never a maximum, never updated
        }
        int far = (int)start[b];
        float cx, cy, cz;
        {
            const float* c = p + (int64_t)far * cv.sn;
            cx = c[0];
            cy = c[cv.sc];
            cz = c[2 * cv.sc];
        }
        u64* gb = gran + (size_t)b * npoint * 4 * G;

#ifdef PN2_FPS_DIAG
        unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#define STAMP(k)                                                  \
    do {                                                          \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        st[k] += t_ - tprev;                                      \
        tprev = t_;                                               \
    } while (0)
#else
#define STAMP(k)
#endif
        for (int i = 0; i < npoint; ++i) {
#ifdef PN2_FPS_DIAG
            tprev = __builtin_amdgcn_s_memtime();
#endif
            if (g == 0 && tid == 0) {
                out_idx[(size_t)b * npoint + i] = far;
                if (out_xyz) {
                    float* o = out_xyz + ((size_t)b * npoint + i) * 3;
                    o[0] = cx;
                    o[1] = cy;
                    o[2] = cz;
                }
            }
            if (i == npoint - 1) break;

            float bestd = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
            int bestj = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const float dx = __fsub_rn(x[j], cx), dy = __fsub_rn(y[j], cy), dz = __fsub_rn(z[j], cz);
                const float dist = pn2::norm2(dx, dy, dz);
                d[j] = dist < d[j] ? dist : d[j];
                if (d[j] > bestd) {  // strict: the lowest index of this thread wins
                    bestd = d[j];
                    bestj = j;
                    bx = x[j];
                    by = y[j];
                    bz = z[j];
                }
            }
            const int bestn = base + bestj * T;
            const u64 mykey =
                bestd < 0.0f ? 0ull : (((u64)__float_as_uint(bestd)) << 32) | (u64)(0xFFFFFFFFu - (unsigned)bestn);
            // one reduction of the distance words; the index words only when two lanes share the largest distance
            int owner;
            const u64 wkey = wave_max_key_owner(mykey, owner);
            const int buf = i & 1;
            // the lane that owns the wave maximum (keys are unique unless all are 0) publishes key + coordinates
            if (lane == owner) {
                s_key[buf][wave] = wkey;
                s_xyz[buf][wave][0] = bx;
                s_xyz[buf][wave][1] = by;
                s_xyz[buf][wave][2] = bz;
            }
            STAMP(0);  // compute + wave reduce + LDS write
            lds_barrier();
            STAMP(1);  // barrier 1
            // workgroup winner: lane w reads wave w's candidate, one DPP reduction (instead of a 16-deep compare
            // chain in every thread).  With G > 1 only the publishing wave needs it.
            u64 k = 0;
            float nx = 0.f, ny = 0.f, nz = 0.f;
            if (!GROUPED || wave == 0) {
                int kw = 0;
                if (NW <= 4) {
                    k = s_key[buf][0];
#pragma unroll
                    for (int w = 1; w < NW; ++w) {
                        const u64 o = s_key[buf][w];
                        if (o > k) {
                            k = o;
                            kw = w;
                        }
                    }
                } else {
                    const u64 cand = lane < NW ? s_key[buf][lane] : 0ull;
                    k = pn2::wave_max_key((unsigned)(cand >> 32), (unsigned)cand);
                    kw = (int)__builtin_ctzll(__ballot(lane < NW && cand == k));
                }
                // explicit LDS address space: with a run-time wave index the compiler otherwise falls back to FLAT loads here
                // (src_shared_base + vmcnt wait: +130 ns per sample, measured against the round-1 build)
                const lds_f32* cand_xyz = (const lds_f32*)&s_xyz[buf][0][0];
                nx = cand_xyz[kw * 3];
                ny = cand_xyz[kw * 3 + 1];
                nz = cand_xyz[kw * 3 + 2];
            }
            if (GROUPED) {
                if (wave == 0) {
                    u64* slot = gb + (size_t)i * 4 * G;
                    const unsigned tag = (unsigned)(i + 1);
                    if (lane < 4) {
                        const float cxyz = lane == 1 ? nx : lane == 2 ? ny : nz;
                        const u64 v = lane == 0 ? (k | kValid) : (((u64)tag) << 32) | (u64)__float_as_uint(cxyz);
                        st_granule(slot + (size_t)lane * G + g, v);
                    }
                    STAMP(2);  // LDS scan + publish
                    u64 v0 = kValid, v1 = 0, v2 = 0, v3 = 0;
                    if (lane < G) {
                        unsigned spins = 0;
                        for (;;) {
                            // four independent loads in flight per poll (a branch per load serialises them)
                            v0 = ld_granule(slot + lane);
                            v1 = ld_granule(slot + G + lane);
                            v2 = ld_granule(slot + 2 * G + lane);
                            v3 = ld_granule(slot + 3 * G + lane);
                            const bool ok = ((v0 & kValid) != 0) & ((unsigned)(v1 >> 32) == tag) &
                                            ((unsigned)(v2 >> 32) == tag) & ((unsigned)(v3 >> 32) == tag);
                            if (ok) break;
                            if (!spin_alive(spins, kn, err, kStatusHandoff)) {  // a member never arrived: the launch is dead
                                s_dead = 1;
                                v0 = kValid;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    STAMP(3);  // poll
                    const u64 mine = v0 & ~kValid;
                    const u64 best = pn2::wave_max_key((unsigned)(mine >> 32), (unsigned)mine);
                    const u64 own = __ballot(lane < G && mine == best);
                    if (lane == (int)__builtin_ctzll(own)) {
                        s_win[buf] = best;
                        s_wxyz[buf][0] = __uint_as_float((unsigned)v1);
                        s_wxyz[buf][1] = __uint_as_float((unsigned)v2);
                        s_wxyz[buf][2] = __uint_as_float((unsigned)v3);
                    }
                }
                STAMP(4);  // group reduce + LDS write
                lds_barrier();
                STAMP(5);  // barrier 2
                if (s_dead) return;   // uniform: every thread reads the flag after the same barrier
                k = s_win[buf];
                nx = s_wxyz[buf][0];
                ny = s_wxyz[buf][1];
                nz = s_wxyz[buf][2];
            }
            far = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            cx = nx;
            cy = ny;
            cz = nz;
        }
#ifdef PN2_FPS_DIAG
        if (tid == 0 && g == 0 && b == 0) {
            unsigned long long* dbg = (unsigned long long*)err + 2;   // workspace bytes 16.. (diag builds reserve 256)
            for (int k = 0; k < 6; ++k) dbg[k] = st[k];
            dbg[6] = __builtin_amdgcn_s_memtime() - ct0;
            dbg[7] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
#endif
        __syncthreads();  // LDS slots are reused by the next cloud of this group
    }
}

// ------------------------------------------------------------------------------------------------------------
// XCD-local variant for medium clouds (8192 < N <= 524288).
//
// A granule hand-off between workgroups costs ~1400 cycles per hop when it has to cross XCDs (write-through store,
// memory-side load) but only ~500 cycles when both workgroups share an XCD's L2: a plain store lands in that L2
// and an L1-bypassing (sc1) load of a same-XCD reader hits it (tools/ubench/handoff.hip).  Correctness of that
// cheaper form requires every member of a group to run on the SAME XCD, which HIP does not let a kernel choose --
// so membership is decided at run time from where the workgroups actually landed: each workgroup reads its
// HW_REG_XCC_ID, takes a rank on that XCD with a device-scope atomic, and after one grid-wide arrival count the
// launch knows how many workgroups every XCD holds.  Groups of G consecutive ranks of one XCD own one cloud each
// (plain-store granules); if no XCD holds G workgroups (busy GPU, odd placement) the launch falls back, as a
// whole, to groups of consecutive block ids with the placement-independent write-through granules.  Either way
// results are identical; only the speed depends on placement.
constexpr int kXT = 512, kXPPT = 16, kXGrid = 256;
struct XcdHeader {
    unsigned err, arrived, cnt[8], pad[6];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// Phase 0 of the XCD-local kernels, run by one lane: which XCD did this workgroup land on, which group of G same-XCD
// workgroups (or, failing that, of consecutive block ids) does it belong to.  s_role = {group or -1, rank, #groups, local}.
__device__ void xcd_roles(XcdHeader* hdr, int G, int* s_role, const Knobs& kn) {
        const unsigned x = xcc_id();
        const unsigned rank = atomicAdd(&hdr->cnt[x], 1u);
        __threadfence();
        atomicAdd(&hdr->arrived, 1u);
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(&hdr->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            if (!spin_alive(spins, kn, &hdr->err, kStatusArrival)) {  // the grid is not co-resident (busy GPU)
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        int total = 0, before = 0, mine = 0;
        for (unsigned k = 0; k < 8; ++k) {
            const int c = (int)__hip_atomic_load(&hdr->cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / G;
            if (k < x) before += c;
            if (k == x) mine = c;
            total += c;
        }
        int group = -1, grank = 0, ngroups = 0, local = 0;
        if (ok && total > 0 && !kn.force_fallback) {  // XCD-local groups
            local = 1;
            ngroups = total;
            if ((int)rank < mine * G) {
                group = before + (int)rank / G;
                grank = (int)rank % G;
            }
        } else if (ok) {        // placement-independent fallback: consecutive block ids
            ngroups = (int)gridDim.x / G;
            if ((int)blockIdx.x < ngroups * G) {
                group = (int)blockIdx.x / G;
                grank = (int)blockIdx.x % G;
            }
        }
        s_role[0] = group;
        s_role[1] = grank;
        s_role[2] = ngroups;
        s_role[3] = local;
    }

template <bool PERWAVE>
__global__ __launch_bounds__(kXT) void fps_xcd_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                      int B, int N, int npoint, const int64_t* __restrict__ start,
                                                      int32_t* __restrict__ out_idx, float* __restrict__ out_xyz,
                                                      u64* gran, XcdHeader* hdr, int G, Knobs kn) {
    constexpr int T = kXT, NW = T / 64;
    __shared__ int s_dead;
    __shared__ u64 s_key[2][NW];
    __shared__ float s_xyz[2][NW][3];
    __shared__ u64 s_win[2];
    __shared__ float s_wxyz[2][3];
    __shared__ int s_role[4];  // {group id or -1, rank in group, number of groups, local?}
    // the workgroup's points once more in LDS: the lane that wins a step fetches its coordinates from here instead
    // of every thread dragging (x, y, z) of its running best through the update loop
    __shared__ float s_px[kXPPT * kXT], s_py[kXPPT * kXT], s_pz[kXPPT * kXT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- phase 0: fill the outputs, then: where did this workgroup land?  (one lane; everything below is wave-uniform)
    for (int b = blockIdx.x; b < B; b += gridDim.x) poison_rows(out_idx, out_xyz, (size_t)b * npoint, npoint, tid, T);
    __syncthreads();   // every store of the fill has left before thread 0's release + arrival in xcd_roles
    if (tid == 0) {
        s_dead = 0;
        xcd_roles(hdr, G, s_role, kn);
    }
    __syncthreads();
    const int group = s_role[0], g = s_role[1], ngroups = s_role[2];
    const bool local = s_role[3] != 0;
    if (group < 0 || ngroups <= 0) return;

    const int ppt = (N + G * T - 1) / (G * T);  // <= kXPPT by the host's choice of G
    const int base = g * (T * ppt) + tid;

    for (int b = group; b < B; b += ngroups) {
        const float* p = xyz + (int64_t)b * sb;
        // two points per register pair: the subtract / multiply / add of the update run as packed fp32 ops
        f2 x[kXPPT / 2], y[kXPPT / 2], z[kXPPT / 2], d[kXPPT / 2];
#pragma unroll
        for (int j = 0; j < kXPPT; ++j) {
            const int n = base + j * T;
            const bool ok = j < ppt && n < N;
            const float* q = p + (int64_t)(ok ? n : 0) * sn;
            const float qx = q[0], qy = q[sc], qz = q[2 * sc];
            x[j >> 1][j & 1] = qx;
            y[j >> 1][j & 1] = qy;
            z[j >> 1][j & 1] = qz;
            d[j >> 1][j & 1] = ok ? 1e10f : -1.0f;
            s_px[j * T + tid] = qx;
            s_py[j * T + tid] = qy;
            s_pz[j * T + tid] = qz;
        }
        int far = (int)start[b];
        float cx, cy, cz;
        {
            const float* c = p + (int64_t)far * sn;
            cx = c[0];
            cy = c[sc];
            cz = c[2 * sc];
        }
        u64* gb = gran + (size_t)b * npoint * 4 * G * (PERWAVE ? NW : 1);

#ifdef PN2_FPS_DIAG
        unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#endif
        for (int i = 0; i < npoint; ++i) {
#ifdef PN2_FPS_DIAG
            tprev = __builtin_amdgcn_s_memtime();
#endif
            if (g == 0 && tid == 0) {
                out_idx[(size_t)b * npoint + i] = far;
                if (out_xyz) {
                    float* o = out_xyz + ((size_t)b * npoint + i) * 3;
                    o[0] = cx;
                    o[1] = cy;
                    o[2] = cz;
                }
            }
            if (i == npoint - 1) break;

            float bestd = -1.0f;
            int bestj = 0;
            const f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
            for (int q = 0; q < kXPPT / 2; ++q) {
                // ((dx*dx + dy*dy) + dz*dz), unfused (the TU is built with -ffp-contract=off), two points at a time
                const f2 dx = x[q] - c2x, dy = y[q] - c2y, dz = z[q] - c2z;
                const f2 dist = (dx * dx + dy * dy) + dz * dz;
                d[q][0] = fminf(d[q][0], dist[0]);   // slots beyond N hold -1 and stay -1
                d[q][1] = fminf(d[q][1], dist[1]);
                if (d[q][0] > bestd) {  // strict: the lowest index of this thread wins
                    bestd = d[q][0];
                    bestj = 2 * q;
                }
                if (d[q][1] > bestd) {
                    bestd = d[q][1];
                    bestj = 2 * q + 1;
                }
            }
            const int bestn = base + bestj * T;
            const u64 mykey =
                bestd < 0.0f ? 0ull : (((u64)__float_as_uint(bestd)) << 32) | (u64)(0xFFFFFFFFu - (unsigned)bestn);
            const u64 wkey = pn2::wave_max_key((unsigned)(mykey >> 32), (unsigned)mykey);
            u64 k = 0;
            float nx = 0.f, ny = 0.f, nz = 0.f;
            if (PERWAVE) {
            const int buf = i & 1;
            // Every WAVE publishes its own candidate (no workgroup-level reduction, no barrier before the exchange):
            // entry e = g * NW + wave of the step's [4][E] granule block, E = G * NW.  One wave per workgroup polls
            // all E entries (E / 64 per lane, every load of a poll in flight together), reduces them and hands the
            // winner to the other waves through LDS.
            const int E = G * NW;
            u64* slot = gb + (size_t)i * 4 * E;
            const unsigned tag = (unsigned)(i + 1);
            const u64 owners = __ballot(mykey == wkey);
            if (lane == (int)__builtin_ctzll(owners)) {
                const int e = g * NW + wave;
                const u64 gx = (((u64)tag) << 32) | (u64)__float_as_uint(s_px[bestj * T + tid]);
                const u64 gy = (((u64)tag) << 32) | (u64)__float_as_uint(s_py[bestj * T + tid]);
                const u64 gz = (((u64)tag) << 32) | (u64)__float_as_uint(s_pz[bestj * T + tid]);
                if (local) {  // plain stores: they stay in this XCD's L2, where the group's sc1 loads find them
                    __hip_atomic_store(slot + e, wkey | kValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(slot + E + e, gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(slot + 2 * E + e, gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(slot + 3 * E + e, gz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    st_granule(slot + e, wkey | kValid);
                    st_granule(slot + E + e, gx);
                    st_granule(slot + 2 * E + e, gy);
                    st_granule(slot + 3 * E + e, gz);
                }
            }
            STAMP(0);
            if (wave == 0) {
                constexpr int kMaxT = kMaxG * NW / 64;   // entries per lane, at most
                const int nt = (E + 63) / 64;
                u64 bk = 0;
                unsigned bx = 0, by = 0, bz = 0, spins = 0;
                for (;;) {
                    bool ok = true;
                    bk = 0;
#pragma unroll
                    for (int t = 0; t < kMaxT; ++t) {
                        if (t < nt) {  // wave-uniform
                            const int e = lane + 64 * t;
                            const int ec = e < E ? e : E - 1;
                            const u64 v0 = ld_granule(slot + ec), v1 = ld_granule(slot + E + ec);
                            const u64 v2 = ld_granule(slot + 2 * E + ec), v3 = ld_granule(slot + 3 * E + ec);
                            ok = ok & ((v0 & kValid) != 0) & ((unsigned)(v1 >> 32) == tag) & ((unsigned)(v2 >> 32) == tag) &
                                 ((unsigned)(v3 >> 32) == tag);
                            const u64 key = e < E ? (v0 & ~kValid) : 0ull;
                            if (key > bk) {
                                bk = key;
                                bx = (unsigned)v1;
                                by = (unsigned)v2;
                                bz = (unsigned)v3;
                            }
                        }
                    }
                    if (__all(ok)) break;
                    if (!spin_alive(spins, kn, &hdr->err, kStatusHandoff)) {  // wave-uniform
                        s_dead = 1;
                        break;
                    }
                }
                STAMP(3);
                const u64 best = pn2::wave_max_key((unsigned)(bk >> 32), (unsigned)bk);
                const u64 own = __ballot(bk == best);
                if (lane == (int)__builtin_ctzll(own)) {
                    s_win[buf] = best;
                    s_wxyz[buf][0] = __uint_as_float(bx);
                    s_wxyz[buf][1] = __uint_as_float(by);
                    s_wxyz[buf][2] = __uint_as_float(bz);
                }
            }
            STAMP(4);
            lds_barrier();
            STAMP(5);
            if (s_dead) return;
            k = s_win[buf];
            nx = s_wxyz[buf][0];
            ny = s_wxyz[buf][1];
            nz = s_wxyz[buf][2];
            } else {
            const int buf = i & 1;
            const u64 owners = __ballot(mykey == wkey);
            if (lane == (int)__builtin_ctzll(owners)) {
                s_key[buf][wave] = wkey;
                s_xyz[buf][wave][0] = s_px[bestj * T + tid];
                s_xyz[buf][wave][1] = s_py[bestj * T + tid];
                s_xyz[buf][wave][2] = s_pz[bestj * T + tid];
            }
            STAMP(0);
            lds_barrier();
            STAMP(1);
            // workgroup winner: lane w reads wave w's candidate, one DPP reduction (instead of a 16-deep compare
            // chain in every thread).  With G > 1 only the publishing wave needs it.
            k = 0;
            nx = ny = nz = 0.f;
            if (G == 1 || wave == 0) {
                int kw = 0;
                if (NW <= 4) {
                    k = s_key[buf][0];
#pragma unroll
                    for (int w = 1; w < NW; ++w) {
                        const u64 o = s_key[buf][w];
                        if (o > k) {
                            k = o;
                            kw = w;
                        }
                    }
                } else {
                    const u64 cand = lane < NW ? s_key[buf][lane] : 0ull;
                    k = pn2::wave_max_key((unsigned)(cand >> 32), (unsigned)cand);
                    kw = (int)__builtin_ctzll(__ballot(lane < NW && cand == k));
                }
                nx = s_xyz[buf][kw][0];
                ny = s_xyz[buf][kw][1];
                nz = s_xyz[buf][kw][2];
            }
            if (G > 1) {
                if (wave == 0) {
                    u64* slot = gb + (size_t)i * 4 * G;
                    const unsigned tag = (unsigned)(i + 1);
                    if (lane < 4) {
                        const float cxyz = lane == 1 ? nx : lane == 2 ? ny : nz;
                        const u64 v = lane == 0 ? (k | kValid) : (((u64)tag) << 32) | (u64)__float_as_uint(cxyz);
                        u64* dst = slot + (size_t)lane * G + g;
                        if (local)  // plain store: stays in this XCD's L2, where the group's sc1 loads find it
                            __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else
                            st_granule(dst, v);
                    }
                    STAMP(2);
                    u64 v0 = kValid, v1 = 0, v2 = 0, v3 = 0;
                    if (lane < G) {
                        unsigned spins = 0;
                        for (;;) {
                            // four independent loads in flight per poll (a branch per load serialises them)
                            v0 = ld_granule(slot + lane);
                            v1 = ld_granule(slot + G + lane);
                            v2 = ld_granule(slot + 2 * G + lane);
                            v3 = ld_granule(slot + 3 * G + lane);
                            const bool ok = ((v0 & kValid) != 0) & ((unsigned)(v1 >> 32) == tag) &
                                            ((unsigned)(v2 >> 32) == tag) & ((unsigned)(v3 >> 32) == tag);
                            if (ok) break;
                            if (!spin_alive(spins, kn, &hdr->err, kStatusHandoff)) {
                                s_dead = 1;
                                v0 = kValid;
                                break;
                            }
                        }
                    }
                    STAMP(3);
                    const u64 mine = v0 & ~kValid;
                    const u64 best = pn2::wave_max_key((unsigned)(mine >> 32), (unsigned)mine);
                    const u64 own = __ballot(lane < G && mine == best);
                    if (lane == (int)__builtin_ctzll(own)) {
                        s_win[buf] = best;
                        s_wxyz[buf][0] = __uint_as_float((unsigned)v1);
                        s_wxyz[buf][1] = __uint_as_float((unsigned)v2);
                        s_wxyz[buf][2] = __uint_as_float((unsigned)v3);
                    }
                }
                STAMP(4);
                lds_barrier();
                STAMP(5);
                if (s_dead) return;
                k = s_win[buf];
                nx = s_wxyz[buf][0];
                ny = s_wxyz[buf][1];
                nz = s_wxyz[buf][2];
            }
            }
            far = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            cx = nx;
            cy = ny;
            cz = nz;
        }
#ifdef PN2_FPS_DIAG
        if (tid == 0 && g == 0 && b == 0) {
            unsigned long long* dbg = (unsigned long long*)((char*)hdr + 64);   // first granule bytes (diag only)
            for (int k = 0; k < 6; ++k) dbg[k] = st[k];
            dbg[6] = __builtin_amdgcn_s_memtime() - ct0;
            dbg[7] = __builtin_amdgcn_s_memrealtime() - rt0;
            dbg[8] = (unsigned long long)local;
        }
#endif
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// Multi-pick rounds (same placement scheme as fps_xcd_kernel, one exchange per ROUND instead of per sample).
//
// After the update for centroid c the next sample is the point with the largest key.  Suppose the exchange delivers
// the largest keys t1 > t2 > ... > tK (with coordinates).  t1 is the next sample.  If t1's update leaves t2's
// distance unchanged (|t2 - t1|^2 >= d[t2], the very comparison the update makes), then t2 still holds the largest key
// after that update -- every other key was below it and keys only shrink -- so t2 is the sample after t1, without
// another exchange; likewise t3 if neither t1 nor t2 touches it, and so on until the first candidate that is touched.
// The accepted prefix is exactly the sequential algorithm's next samples, in order; the following round applies
// all of them in one update.  Far-apart maxima are the rule (they sit in different unsampled regions): on the
// 262144-point tree 1024 samples take ~350 rounds with K = 4 instead of 1023 steps.
//
// Extracting a true top-K at every level (lane, wavefront, workgroup, group) costs K dependent reductions each, so the
// lists are kept cheap and CERTIFIED instead: a wavefront contributes only its best point plus the key of its
// runner-up as a bound on everything it did not list; a workgroup publishes the best of its wavefront winners plus one
// bound (the largest key it knows of that it did not publish); the group takes the best kMK of the members'
// candidates and H = the largest other candidate or bound.  A candidate is accepted only while its key is above H:
// then nothing unlisted can lie between it and the candidates before it, i.e. it really is the next largest key.
// (The largest keys sit in different unsampled regions and points are dealt to workgroups by index, so they rarely
// share a workgroup: 368 rounds instead of the ideal 352 on the 262144-point tree.)
constexpr int kMK = 4;  // samples accepted per round, at most

template <int PPT>
__global__ __launch_bounds__(kXT) void fps_multi_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                        int B, int N, int npoint, const int64_t* __restrict__ start,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_xyz,
                                                        u64* gran, XcdHeader* hdr, int G, Knobs kn) {
    constexpr int T = kXT, NW = T / 64;
    constexpr int kGran = 5;  // granules per member and round: {key, x, y, z} of its best point + the bound
    __shared__ u64 s_wkey[2][NW], s_wsec[2][NW];
    __shared__ float s_wxyz[2][NW][3];
    __shared__ float s_cent[2][kMK][3];
    __shared__ u64 s_ckey[2][kMK + 1];
    __shared__ int s_m[2];
    __shared__ int s_role[4];
    __shared__ float s_px[PPT * kXT], s_py[PPT * kXT], s_pz[PPT * kXT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = blockIdx.x; b < B; b += gridDim.x) poison_rows(out_idx, out_xyz, (size_t)b * npoint, npoint, tid, T);
    __syncthreads();   // every store of the fill has left before thread 0's release + arrival in xcd_roles
    if (tid == 0) xcd_roles(hdr, G, s_role, kn);
    __syncthreads();
    const int group = s_role[0], g = s_role[1], ngroups = s_role[2];
    const bool local = s_role[3] != 0;
    if (group < 0 || ngroups <= 0) return;

    const int ppt = (N + G * T - 1) / (G * T);
    const int base = g * (T * ppt) + tid;
    for (int b = group; b < B; b += ngroups) {
        const float* p = xyz + (int64_t)b * sb;
        f2 x[PPT / 2], y[PPT / 2], z[PPT / 2], d[PPT / 2];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int n = base + j * T;
            const bool ok = j < ppt && n < N;
            const float* q = p + (int64_t)(ok ? n : 0) * sn;
            const float qx = q[0], qy = q[sc], qz = q[2 * sc];
            x[j >> 1][j & 1] = qx;
            y[j >> 1][j & 1] = qy;
            z[j >> 1][j & 1] = qz;
            d[j >> 1][j & 1] = ok ? 1e10f : -1.0f;
            s_px[j * T + tid] = qx;
            s_py[j * T + tid] = qy;
            s_pz[j * T + tid] = qz;
        }
        int m = 1;  // centroids to apply this round
        float ccx[kMK], ccy[kMK], ccz[kMK];
        {
            const int far = (int)start[b];
            const float* c = p + (int64_t)far * sn;
            ccx[0] = c[0], ccy[0] = c[sc], ccz[0] = c[2 * sc];
#pragma unroll
            for (int t = 1; t < kMK; ++t) ccx[t] = ccx[0], ccy[t] = ccy[0], ccz[t] = ccz[0];
            if (g == 0 && tid == 0) {
                out_idx[(size_t)b * npoint] = far;
                if (out_xyz) {
                    float* o = out_xyz + (size_t)b * npoint * 3;
                    o[0] = ccx[0], o[1] = ccy[0], o[2] = ccz[0];
                }
            }
        }
        int count = 1;
        u64* gb = gran + (size_t)b * npoint * kGran * G;
#ifdef PN2_FPS_DIAG
        unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, nrounds = 0;
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#endif

        for (int round = 0; count < npoint; ++round) {
            const int buf = round & 1;
#ifdef PN2_FPS_DIAG
            tprev = __builtin_amdgcn_s_memtime();
            ++nrounds;
#endif
            // ---- update with the m accepted centroids (m is uniform over the whole group)
#pragma unroll
            for (int t = 0; t < kMK; ++t) {
                if (t < m) {
                    const f2 c2x = {ccx[t], ccx[t]}, c2y = {ccy[t], ccy[t]}, c2z = {ccz[t], ccz[t]};
#pragma unroll
                    for (int q = 0; q < PPT / 2; ++q) {
                        const f2 dx = x[q] - c2x, dy = y[q] - c2y, dz = z[q] - c2z;
                        const f2 dist = (dx * dx + dy * dy) + dz * dz;
                        d[q][0] = fminf(d[q][0], dist[0]);   // slots beyond N hold -1 and stay -1
                        d[q][1] = fminf(d[q][1], dist[1]);
                    }
                }
            }
            STAMP(0);  // update
            // ---- this lane's best two (strict '>': the lower slot = lower index wins ties)
            float b1d = -1.0f, b2d = -1.0f;
            int b1j = 0, b2j = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const float v = d[j >> 1][j & 1];
                const bool g1 = v > b1d, g2 = v > b2d;
                b2d = g1 ? b1d : (g2 ? v : b2d);
                b2j = g1 ? b1j : (g2 ? j : b2j);
                b1d = g1 ? v : b1d;
                b1j = g1 ? j : b1j;
            }
            const u64 k1 = fps_key(b1d, base + b1j * T), k2 = fps_key(b2d, base + b2j * T);
            // ---- wavefront: best point (key + coordinates) and the runner-up's key as the bound on the rest
            int owner;
            const u64 w1 = wave_max_key_owner(k1, owner);
            const u64 w2 = wave_max_key_only(lane == owner ? k2 : k1);
            if (lane == owner) {
                s_wkey[buf][wave] = w1;
                s_wsec[buf][wave] = w2;
                s_wxyz[buf][wave][0] = s_px[b1j * T + tid];
                s_wxyz[buf][wave][1] = s_py[b1j * T + tid];
                s_wxyz[buf][wave][2] = s_pz[b1j * T + tid];
            }
            STAMP(1);  // lane best two + wavefront best/bound
            lds_barrier();
            STAMP(2);  // barrier 1
            if (wave == 0) {
                // ---- workgroup: its best point + the bound on everything else it holds (the other winners and every
                // wavefront's runner-up); the lane that holds the best winner publishes all five granules
                u64* slot = gb + (size_t)round * kGran * G;
                const unsigned tag = (unsigned)(round + 1);
                const u64 mine = lane < NW ? s_wkey[buf][lane] : 0ull;
                const u64 msec = lane < NW ? s_wsec[buf][lane] : 0ull;
                int o1;
                const u64 best = wave_max_key_owner(mine, o1);
                const u64 others = lane == o1 ? msec : (mine > msec ? mine : msec);
                const u64 bound = wave_max_key_only(others);
                if (lane == o1) {
                    u64* dst = slot + g;
                    const u64 v0 = best | kValid;
                    const u64 v1 = (((u64)tag) << 32) | (u64)__float_as_uint(s_wxyz[buf][lane < NW ? lane : 0][0]);
                    const u64 v2 = (((u64)tag) << 32) | (u64)__float_as_uint(s_wxyz[buf][lane < NW ? lane : 0][1]);
                    const u64 v3 = (((u64)tag) << 32) | (u64)__float_as_uint(s_wxyz[buf][lane < NW ? lane : 0][2]);
                    const u64 v4 = bound | kValid;
                    if (local) {
                        __hip_atomic_store(dst, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + G, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + 2 * G, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + 3 * G, v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + 4 * G, v4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        st_granule(dst, v0);
                        st_granule(dst + G, v1);
                        st_granule(dst + 2 * G, v2);
                        st_granule(dst + 3 * G, v3);
                        st_granule(dst + 4 * G, v4);
                    }
                }
                STAMP(3);  // workgroup candidate + publish
                // ---- poll: lane l reads the five granules of members l and l + 64 (G <= 128)
                u64 ek[2] = {0, 0}, hb = 0;
                unsigned ex[2] = {0, 0}, ey[2] = {0, 0}, ez[2] = {0, 0};
                bool dead = false;
                {
                    const bool two = G > 64;  // wave-uniform
                    const u64* src0 = slot + (lane < G ? lane : 0);
                    const u64* src1 = slot + (lane + 64 < G ? lane + 64 : 0);
                    unsigned spins = 0;
                    for (;;) {
                        const u64 v0 = ld_granule(src0), v1 = ld_granule(src0 + G), v2 = ld_granule(src0 + 2 * G),
                                  v3 = ld_granule(src0 + 3 * G), v4 = ld_granule(src0 + 4 * G);
                        bool ok = ((v0 & kValid) != 0) & ((unsigned)(v1 >> 32) == tag) & ((unsigned)(v2 >> 32) == tag) &
                                  ((unsigned)(v3 >> 32) == tag) & ((v4 & kValid) != 0);
                        ek[0] = lane < G ? (v0 & ~kValid) : 0ull;
                        hb = lane < G ? (v4 & ~kValid) : 0ull;
                        ex[0] = (unsigned)v1, ey[0] = (unsigned)v2, ez[0] = (unsigned)v3;
                        if (two) {
                            const u64 w0 = ld_granule(src1), w1 = ld_granule(src1 + G), w2 = ld_granule(src1 + 2 * G),
                                      w3 = ld_granule(src1 + 3 * G), w4 = ld_granule(src1 + 4 * G);
                            ok = ok & ((w0 & kValid) != 0) & ((unsigned)(w1 >> 32) == tag) & ((unsigned)(w2 >> 32) == tag) &
                                 ((unsigned)(w3 >> 32) == tag) & ((w4 & kValid) != 0);
                            const bool on = lane + 64 < G;
                            ek[1] = on ? (w0 & ~kValid) : 0ull;
                            const u64 hb1 = on ? (w4 & ~kValid) : 0ull;
                            hb = hb1 > hb ? hb1 : hb;
                            ex[1] = (unsigned)w1, ey[1] = (unsigned)w2, ez[1] = (unsigned)w3;
                        }
                        if (__all(ok)) break;
                        if (!spin_alive(spins, kn, &hdr->err, kStatusHandoff)) {  // wave-uniform
                            dead = true;   // reported to the whole workgroup through s_m below
                            break;
                        }
                    }
                }
                STAMP(4);  // poll
                // ---- group: the best kMK member candidates by rank (keys through LDS), H = the largest other key
#pragma unroll
                for (int t = 0; t < kMK; ++t) {
                    const bool second = ek[1] > ek[0];
                    const u64 mine = second ? ek[1] : ek[0];
                    int own;
                    const u64 k = wave_max_key_owner(mine, own);
                    if (lane == own) {
                        s_cent[buf][t][0] = __uint_as_float(second ? ex[1] : ex[0]);
                        s_cent[buf][t][1] = __uint_as_float(second ? ey[1] : ey[0]);
                        s_cent[buf][t][2] = __uint_as_float(second ? ez[1] : ez[0]);
                        s_ckey[buf][t] = k;
                        if (second) ek[1] = 0; else ek[0] = 0;
                    }
                }
                // everything that is not in the list: the remaining candidates and all the members' bounds
                const u64 left = ek[1] > ek[0] ? ek[1] : ek[0];
                const u64 hmax = wave_max_key_only(left > hb ? left : hb);
                if (lane == 0) s_ckey[buf][kMK] = 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // ---- accepted prefix: above H (certified next-largest) and untouched by the ones accepted before it
                u64 tk[kMK];
                float tx[kMK], ty[kMK], tz[kMK];
#pragma unroll
                for (int t = 0; t < kMK; ++t)
                    tk[t] = s_ckey[buf][t], tx[t] = s_cent[buf][t][0], ty[t] = s_cent[buf][t][1], tz[t] = s_cent[buf][t][2];
                const u64 fifth = s_ckey[buf][kMK];
                const u64 H = fifth > hmax ? fifth : hmax;
                int acc = 1;
#pragma unroll
                for (int t = 1; t < kMK; ++t) {
                    bool keep = acc == t && tk[t] != 0 && tk[t] > H;
                    const float dt = __uint_as_float((unsigned)(tk[t] >> 32));
#pragma unroll
                    for (int a = 0; a < t; ++a) {
                        const float dx = __fsub_rn(tx[t], tx[a]), dy = __fsub_rn(ty[t], ty[a]), dz = __fsub_rn(tz[t], tz[a]);
                        keep = keep && !(pn2::norm2(dx, dy, dz) < dt);
                    }
                    if (keep) acc = t + 1;
                }
                if (acc > npoint - count) acc = npoint - count;
                if (g == 0 && lane < acc && !dead) {
                    const size_t o = (size_t)b * npoint + count + lane;
                    out_idx[o] = (int)(0xFFFFFFFFu - (unsigned)(s_ckey[buf][lane] & 0xFFFFFFFFull));
                    if (out_xyz)
                        out_xyz[o * 3] = s_cent[buf][lane][0], out_xyz[o * 3 + 1] = s_cent[buf][lane][1],
                                    out_xyz[o * 3 + 2] = s_cent[buf][lane][2];
                }
                if (lane == 0) s_m[buf] = dead ? -1 : acc;   // -1: the launch is dead, everybody leaves after the barrier
            }
            STAMP(5);  // group list + chain (wave 0) / wait (others)
            lds_barrier();
            m = s_m[buf];
            if (m < 0) return;    // dead launch (uniform: every thread reads the same word after the same barrier)
#pragma unroll
            for (int t = 0; t < kMK; ++t) ccx[t] = s_cent[buf][t][0], ccy[t] = s_cent[buf][t][1], ccz[t] = s_cent[buf][t][2];
            count += m;
        }
#ifdef PN2_FPS_DIAG
        if (tid == 0 && g == 0 && b == 0) {
            unsigned long long* dbg = (unsigned long long*)((char*)hdr + 64);   // first granule bytes (diag only)
            for (int k = 0; k < 6; ++k) dbg[k] = st[k];
            dbg[6] = __builtin_amdgcn_s_memtime() - ct0;
            dbg[7] = __builtin_amdgcn_s_memrealtime() - rt0;
            dbg[8] = (unsigned long long)local;
            dbg[9] = nrounds;
        }
#endif
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// Spatially ordered multi-pick rounds: the granule exchange of fps_multi_kernel, but (1) the update is skipped where it
// provably changes nothing, (2) a member lists two candidates and (3) the round's samples come from running the
// sequential algorithm on the listed candidates (see fps_sorted_kernel).
//
// A new centroid c lowers d[p] only where |p - c|^2 < d[p].  If every point a wavefront holds lies in a box whose
// distance to c is at least the wavefront's largest d, nothing it holds changes: its lanes' best-two, its published
// candidates and its bound all stay what they were, and the wavefront goes straight to the barrier.  For that to be the
// common case a wavefront has to hold NEIGHBOURS, so the cloud is put into cell order first -- every axis of the cloud's
// bounding box cut into 16, cells numbered along the Morton curve, one counting sort (fps_box_kernel, fps_hist_kernel,
// fps_scatter_kernel: 22 us for 262144 points; a 38-bit radix sort of (cell, index) keys took 14 launches and 130 us) --
// and wavefront w takes the w-th run of 64 * ppt sorted positions.  On the 262144-point tree a centroid reaches ~12 of
// the 256 wavefronts after the first few dozen samples (tools/sim/fps_rounds.py), so a round is the serial exchange
// plus the update and selection of the few busy wavefronts instead of two wavefronts per SIMD updating 16 points per
// lane each.
// The order only decides who holds which point: keys carry the ORIGINAL index, lanes compare full (d, ~index) keys
// (slots are no longer in index order), so samples and tie-breaks are bit-identical to the unsorted kernels.
// The box test is made safe against the rounding of both sides: the update computes fl((dx*dx + dy*dy) + dz*dz) with
// relative error < 2^-21.4 of the exact value and so does the box distance, hence the skip requires
// box_distance * (1 - 2^-19) >= max d; non-finite values fail the comparison and are treated as "touched".
// The ordering kernels: workgroups of kOT threads, kOP points per thread (loads of a thread are independent: all in flight).
constexpr int kOT = 1024, kOP = 4;   // 262 144 points: 8 per thread 38 us for the three kernels, 4: 31 us, 2: 32 us

// box[b] = {max enc(x), max enc(y), max enc(z), max ~enc(x), max ~enc(y), max ~enc(z)} (zeroed by the workspace memset);
// one atomic per workgroup and value (same-address atomics retire one at a time)
__global__ __launch_bounds__(kOT) void fps_box_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc, int N,
                                                      unsigned* __restrict__ box) {
    __shared__ unsigned red[kOT / 64][6];
    const int b = blockIdx.y, t = threadIdx.x;
    const float* p = xyz + (int64_t)b * sb;
    unsigned v[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kOP; ++k) {
        const int n = (blockIdx.x * kOP + k) * kOT + t;
        if (n < N) {
            const float* q = p + (int64_t)n * sn;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const unsigned e = ord_enc(q[a * sc]);
                v[a] = max(v[a], e);
                v[3 + a] = max(v[3 + a], ~e);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const unsigned w = pn2::wave_max_u32(v[a]);
        if ((t & 63) == 0) red[t >> 6][a] = w;
    }
    __syncthreads();
    if (t < 6) {
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < kOT / 64; ++k) w = max(w, red[k][t]);
        atomicMax(box + b * 8 + t, w);
    }
}

// total[b][cell] = points of cloud b in the cell (LDS histogram per workgroup, one global add per non-empty cell)
__global__ __launch_bounds__(kOT) void fps_hist_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc, int N,
                                                       const unsigned* __restrict__ box, unsigned* __restrict__ total) {
    __shared__ unsigned h[kCells];
    const int b = blockIdx.y, t = threadIdx.x;
    for (int c = t; c < kCells; c += kOT) h[c] = 0;
    __syncthreads();
    const CellGrid cg = cell_grid(box + b * 8);
    const float* p = xyz + (int64_t)b * sb;
#pragma unroll
    for (int k = 0; k < kOP; ++k) {
        const int n = (blockIdx.x * kOP + k) * kOT + t;
        if (n < N) {
            const float* q = p + (int64_t)n * sn;
            atomicAdd(&h[cell_of(cg, q[0], q[sc], q[2 * sc])], 1u);
        }
    }
    __syncthreads();
    for (int c = t; c < kCells; c += kOT)
        if (h[c]) atomicAdd(total + (size_t)b * kCells + c, h[c]);
}

// order[b][first position of the cell + arrival number] = n.  A workgroup counts its points per cell in LDS (every point
// keeps its arrival number), reserves a run per non-empty cell with ONE global atomic, then places its points.  The order
// INSIDE a cell is whatever order the atomics retire in -- it only decides which lane of which wavefront keeps the point,
// never a result.
__global__ __launch_bounds__(kOT) void fps_scatter_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc, int N,
                                                          const unsigned* __restrict__ box, const unsigned* __restrict__ total,
                                                          unsigned* __restrict__ cursor, unsigned* __restrict__ order,
                                                          unsigned* __restrict__ cellstart, float* __restrict__ sorted_xyz) {
    __shared__ unsigned base[kCells];   // first position of the cell; after the reservation: first position of this workgroup's run
    __shared__ unsigned h[kCells];
    __shared__ unsigned part[kOT];
    const int b = blockIdx.y, t = threadIdx.x;
    constexpr int PER = kCells / kOT;
    unsigned v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v[k] = total[(size_t)b * kCells + t * PER + k];
        sum += v[k];
        h[t * PER + k] = 0;
    }
    part[t] = sum;
    __syncthreads();
    for (int s = 1; s < kOT; s <<= 1) {   // inclusive scan of the partial sums
        const unsigned add = t >= s ? part[t - s] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    unsigned run = part[t] - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        base[t * PER + k] = run;
        if (blockIdx.x == 0) cellstart[(size_t)b * (kCells + 1) + t * PER + k] = run;   // kept for other kernels (ball query)
        run += v[k];
    }
    if (blockIdx.x == 0 && t == kOT - 1) cellstart[(size_t)b * (kCells + 1) + kCells] = run;
    const CellGrid cg = cell_grid(box + b * 8);
    const float* p = xyz + (int64_t)b * sb;
    unsigned cell[kOP], arrival[kOP];
    float px[kOP], py[kOP], pz[kOP];
#pragma unroll
    for (int k = 0; k < kOP; ++k) {
        const int n = (blockIdx.x * kOP + k) * kOT + t;
        cell[k] = 0, arrival[k] = 0;
        px[k] = py[k] = pz[k] = 0.0f;
        if (n < N) {
            const float* q = p + (int64_t)n * sn;
            px[k] = q[0], py[k] = q[sc], pz[k] = q[2 * sc];
            cell[k] = cell_of(cg, px[k], py[k], pz[k]);
            arrival[k] = atomicAdd(&h[cell[k]], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = t * PER + k;
        if (h[c]) base[c] += atomicAdd(cursor + (size_t)b * kCells + c, h[c]);
    }
    __syncthreads();
    // the permutation and the coordinates in cell order (three planes per cloud: coalesced reads for whoever walks cells)
    float* sx = sorted_xyz + (size_t)b * 3 * N;
#pragma unroll
    for (int k = 0; k < kOP; ++k) {
        const int n = (blockIdx.x * kOP + k) * kOT + t;
        if (n < N) {
            const size_t pos = base[cell[k]] + arrival[k];
            order[(size_t)b * N + pos] = (unsigned)n;
            sx[pos] = px[k];
            sx[(size_t)N + pos] = py[k];
            sx[2 * (size_t)N + pos] = pz[k];
        }
    }
}

// Round of the ordered kernel (NW wavefronts per member; PER listed candidates per member, G * PER <= 64):
//   every wavefront   box tests (lane t tests centroid t); if touched: update, lane best + runner-up (max / min trees over
//                     the slots), the wavefront's PER best lanes' points + a bound on everything else into LDS -> barrier 1
//   wavefront 0       the member's PER best points + bound, 4 PER + 1 granules published; polls all members' granules
//                     (lane l = candidate l >> 5 of member l & 31 for PER = 2); then SIMULATES the sequential algorithm on
//                     the listed candidates: the largest listed key is the next sample as long as it is above H = the
//                     largest bound (nothing unlisted can beat it -- keys only shrink), it lowers the other listed
//                     candidates' distances exactly as the update will, and so on until the best listed key drops to H
//                     or below.  The accepted samples go to LDS                                            -> barrier 2
// Bounds travel as float bits only (low word taken as all ones): a bound may be loose, never low -- a loose one can
// only shorten the accepted run, which the next round makes up for.  Two listed candidates per member matter: with one,
// H is the largest runner-up of any member and stops the run after 3.4 samples on average; with two it is the largest
// THIRD-best (5.4 samples per round, 191 rounds instead of 302 for 1024 samples of the 262144-point tree,
// tools/sim/fps_listsim.py).
struct alignas(16) ListEntry {
    u64 key;
    float x, y, z, pad;
};
constexpr int kCap = 16;   // samples accepted per round, at most

// ------------------------------------------------------------------------------------------------------------
// Small clouds (N <= 2048: every sampled level of the model): ONE wavefront per cloud.  The whole cloud and its running
// distances sit in one wavefront's registers (PPT points per lane), so a sample is: update + lane best (PPT steps), one DPP
// arg-max over the 64 lanes, three v_readlane for the winner's coordinates -- no LDS, no workgroup barrier, nothing shared.
// The four-wavefront kernel above spends its 0.42-0.49 us per sample on exactly those (barrier + LDS hand-over of the
// wavefront candidates); here a sample of a cloud of <= 256 points is ~0.4 us (256 -> 64 samples: 27 -> 25 us, 64 -> 16: 10.5 ->
// 7.8 us -- small change, kept because it is also the simplest statement of the algorithm).  Same keys, same tie-breaks
// (largest distance, then lowest index), same rounding of the distance: bit-identical samples.
template <int PPT, bool RAGGED>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc, int B,
                                                      int N, int npoint, const int64_t* __restrict__ start,
                                                      int32_t* __restrict__ out_idx, float* __restrict__ out_xyz,
                                                      const int* __restrict__ coff) {
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const pn2::CloudView cv = RAGGED ? pn2::cloud_view(xyz, sb, sn, sc, N, coff, b, 3)
                                         : pn2::CloudView{xyz + (int64_t)b * sb, sn, sc, N};
        const float* p = cv.p;
        float x[PPT], y[PPT], z[PPT], d[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int n = lane + 64 * j;
            const bool ok = n < cv.n;
            const float* q = p + (int64_t)(ok ? n : 0) * cv.sn;
            x[j] = q[0];
            y[j] = q[cv.sc];
            z[j] = q[2 * cv.sc];
            d[j] = ok ? 1e10f : -1.0f;   // -1 marks a slot beyond N: never a maximum, never updated
        }
        int far = (int)start[b];
        float cx, cy, cz;
        {
            const float* c = p + (int64_t)far * cv.sn;
            cx = c[0];
            cy = c[cv.sc];
            cz = c[2 * cv.sc];
        }
        for (int i = 0; i < npoint; ++i) {
            if (lane == 0) {
                out_idx[(size_t)b * npoint + i] = far;
                if (out_xyz) {
                    float* o = out_xyz + ((size_t)b * npoint + i) * 3;
                    o[0] = cx;
                    o[1] = cy;
                    o[2] = cz;
                }
            }
            if (i == npoint - 1) break;
            float bestd = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
            int bestj = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const float dx = __fsub_rn(x[j], cx), dy = __fsub_rn(y[j], cy), dz = __fsub_rn(z[j], cz);
                const float dist = pn2::norm2(dx, dy, dz);
                d[j] = dist < d[j] ? dist : d[j];
                if (d[j] > bestd) {   // strict: the lowest index of this lane wins
                    bestd = d[j];
                    bestj = j;
                    bx = x[j];
                    by = y[j];
                    bz = z[j];
                }
            }
            const int bestn = lane + 64 * bestj;
            const u64 mykey = fps_key(bestd, bestn);
            int owner;
            const u64 wkey = wave_max_key_owner(mykey, owner);
            // all keys zero (npoint > N after every point was taken cannot happen: d >= 0 keeps real keys above the -1 slots)
            far = (int)(0xFFFFFFFFu - (unsigned)(wkey & 0xFFFFFFFFull));
            cx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(bx), owner));
            cy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(by), owner));
            cz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(bz), owner));
        }
    }
}

constexpr int kWaveMaxN = 256;    // four points per lane: beyond that one wavefront's VALU issue rate (4 cycles per instruction) loses to four
                                  // wavefronts on four SIMDs despite their barrier (1024 points, 256 samples: 203 us here, 125 us there)
inline bool use_wave_kernel(int N) { return N <= kWaveMaxN && getenv("PN2_FPS_NO_WAVE") == nullptr; }

template <bool RAGGED>
int launch_wave(const float* xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint, const int64_t* start,
                int32_t* out_idx, float* out_xyz, const int* coff, hipStream_t s) {
    const int ppt = pn2::ceil_div(N, 64);
    const dim3 grid(B < 4096 ? B : 4096), block(64);
    const double fb = (double)B * (12.0 * N + 8.0 * npoint);
#define PN2_WAVE_CASE(P)                                                                                                        \
    if (ppt <= P) {                                                                                                             \
        PN2_LAUNCH("fps", fb, 0, (fps_wave_kernel<P, RAGGED>), grid, block, s, xyz, sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, \
                   coff);                                                                                                       \
    } else
    PN2_WAVE_CASE(1)
    PN2_WAVE_CASE(2)
    PN2_WAVE_CASE(4)
    PN2_WAVE_CASE(8)
    PN2_WAVE_CASE(16)
    PN2_WAVE_CASE(32) { return PN2_E_BADARG; }
#undef PN2_WAVE_CASE
    PN2_LAUNCH_CHECK();
    return 0;
}


// Max over lanes 0..7 / 0..15 (row shifts only), result in every lane.
__device__ __forceinline__ unsigned max8_u32(unsigned v) {
    v = max(v, pn2::dpp_u32<0x111, 0xF>(0u, v));
    v = max(v, pn2::dpp_u32<0x112, 0xF>(0u, v));
    v = max(v, pn2::dpp_u32<0x114, 0xF>(0u, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 7);
}
__device__ __forceinline__ unsigned max16_u32(unsigned v) {
    v = max(v, pn2::dpp_u32<0x111, 0xF>(0u, v));
    v = max(v, pn2::dpp_u32<0x112, 0xF>(0u, v));
    v = max(v, pn2::dpp_u32<0x114, 0xF>(0u, v));
    v = max(v, pn2::dpp_u32<0x118, 0xF>(0u, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
// largest key among lanes 0..W-1 (W = 8 or 16) and the lane that holds it
template <int W>
__device__ __forceinline__ u64 row_max_key_owner(u64 k, int& owner) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const u64 in = (1ull << W) - 1ull;
    const unsigned mh = W == 8 ? max8_u32(hi) : max16_u32(hi);
    u64 who = __ballot(hi == mh) & in;
    unsigned ml;
    if (__popcll(who) == 1) {
        owner = (int)__builtin_ctzll(who);
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, owner);
    } else {
        const unsigned v = hi == mh ? lo : 0u;
        ml = W == 8 ? max8_u32(v) : max16_u32(v);
        who = __ballot(hi == mh && lo == ml) & in;
        owner = (int)__builtin_ctzll(who);
    }
    return ((u64)mh << 32) | ml;
}

template <int PPT, int PER>
__global__ __launch_bounds__(kXT) void fps_sorted_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                         int B, int N, int npoint, const int64_t* __restrict__ start,
                                                         const unsigned* __restrict__ order, int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_xyz, u64* gran, XcdHeader* hdr, int G,
                                                         Knobs kn) {
    constexpr int T = kXT, NW = T / 64;
    constexpr int kGran = 4 * PER + 1;  // granules per member and round: {key, x, y, z} of its PER best points + the bound
    static_assert(PPT <= 16, "slot numbers travel in four bits");
    static_assert(PER == 1 || PER == 2, "one or two listed candidates per member");
    static_assert(NW * PER <= 16, "the member's candidates are reduced inside one row of 16 lanes");
    __shared__ u64 s_wkey[NW][PER];        // a wavefront's candidates and bound: rewritten only in rounds that touch it
    __shared__ float s_wxyz[NW][PER][3];
    __shared__ unsigned s_wbound[NW];
    __shared__ ListEntry s_list[2][kCap];   // the round's accepted samples
    __shared__ int s_m[2];
    __shared__ int s_role[4];
    __shared__ float s_px[PPT * kXT], s_py[PPT * kXT], s_pz[PPT * kXT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int b = blockIdx.x; b < B; b += gridDim.x) poison_rows(out_idx, out_xyz, (size_t)b * npoint, npoint, tid, T);
    __syncthreads();   // every store of the fill has left before thread 0's release + arrival in xcd_roles
    if (tid == 0) xcd_roles(hdr, G, s_role, kn);
    __syncthreads();
    const int group = s_role[0], g = s_role[1], ngroups = s_role[2];
    const bool local = s_role[3] != 0;
    if (group < 0 || ngroups <= 0) return;

    const int ppt = (N + G * T - 1) / (G * T);
    const int first = (g * NW + wave) * (ppt * 64) + lane;   // sorted position of slot 0; slot j is 64 * j further on
    // the candidate this lane polls: candidate pc of member pm
    const int pc = PER == 2 ? (lane >> 5) : 0, pm = PER == 2 ? (lane & 31) : lane;
    const bool polls = pm < G;
    for (int b = group; b < B; b += ngroups) {
        const float* p = xyz + (int64_t)b * sb;
        const unsigned* ord = order + (size_t)b * N;
        f2 x[PPT / 2], y[PPT / 2], z[PPT / 2], d[PPT / 2];
        unsigned pk[PPT];   // original index << 4 | slot
        unsigned ehi[3] = {0, 0, 0}, elo[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int pos = first + j * 64;
            const bool ok = j < ppt && pos < N;
            const int n = ok ? (int)ord[pos] : 0;
            const float* q = p + (int64_t)n * sn;
            const float qx = q[0], qy = q[sc], qz = q[2 * sc];
            pk[j] = ((unsigned)n << 4) | (unsigned)j;
            x[j >> 1][j & 1] = qx;
            y[j >> 1][j & 1] = qy;
            z[j >> 1][j & 1] = qz;
            d[j >> 1][j & 1] = ok ? 1e10f : -1.0f;
            s_px[j * T + tid] = qx;
            s_py[j * T + tid] = qy;
            s_pz[j * T + tid] = qz;
            if (ok) {
                const unsigned e0 = ord_enc(qx), e1 = ord_enc(qy), e2 = ord_enc(qz);
                ehi[0] = max(ehi[0], e0), ehi[1] = max(ehi[1], e1), ehi[2] = max(ehi[2], e2);
                elo[0] = max(elo[0], ~e0), elo[1] = max(elo[1], ~e1), elo[2] = max(elo[2], ~e2);
            }
        }
        // the wavefront's box (wave-uniform)
        float blo[3], bhi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bhi[a] = ord_dec(pn2::wave_max_u32(ehi[a]));
            blo[a] = ord_dec(~pn2::wave_max_u32(elo[a]));
        }
        float wmax = 1e10f;   // the wavefront's largest d (high word of its candidate key)
        int m = 1;            // centroids to apply this round
        float mcx, mcy, mcz;  // lane t: centroid t of the round
        {
            const int far = (int)start[b];
            const float* c = p + (int64_t)far * sn;
            mcx = c[0], mcy = c[sc], mcz = c[2 * sc];
            if (g == 0 && tid == 0) {
                out_idx[(size_t)b * npoint] = far;
                if (out_xyz) {
                    float* o = out_xyz + (size_t)b * npoint * 3;
                    o[0] = mcx, o[1] = mcy, o[2] = mcz;
                }
            }
        }
        int count = 1;
        u64* gb = gran + (size_t)b * npoint * kGran * G;
#ifdef PN2_FPS_DIAG
        unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0, nrounds = 0, ntouch = 0, tupd = 0, tsel = 0;
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#endif

        int rounds_done = 0;
        for (int round = 0; count < npoint; ++round, ++rounds_done) {
            const int buf = round & 1;
#ifdef PN2_FPS_DIAG
            tprev = __builtin_amdgcn_s_memtime();
            ++nrounds;
#endif
            // ---- update with those of the m accepted centroids that can reach this wavefront's box: lane t tests centroid t,
            // the touching ones are applied one by one from scalar registers
            bool touched = false;
            {
                const float bx = fmaxf(fmaxf(blo[0] - mcx, mcx - bhi[0]), 0.0f);
                const float by = fmaxf(fmaxf(blo[1] - mcy, mcy - bhi[1]), 0.0f);
                const float bz = fmaxf(fmaxf(blo[2] - mcz, mcz - bhi[2]), 0.0f);
                const float bd = (bx * bx + by * by) + bz * bz;
                const bool far_away = bd * (1.0f - 1.0f / 524288.0f) >= wmax;   // false for NaN: treated as touched
                u64 reach = __ballot(lane < m && !far_away);
                touched = reach != 0ull;
                while (reach) {   // scalar
                    const int t = (int)__builtin_ctzll(reach);
                    reach &= reach - 1ull;
                    const float cx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(mcx), t));
                    const float cy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(mcy), t));
                    const float cz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(mcz), t));
                    const f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
                    for (int q = 0; q < PPT / 2; ++q) {
                        const f2 dx = x[q] - c2x, dy = y[q] - c2y, dz = z[q] - c2z;
                        const f2 dist = (dx * dx + dy * dy) + dz * dz;
                        d[q][0] = fminf(d[q][0], dist[0]);   // slots beyond N hold -1 and stay -1
                        d[q][1] = fminf(d[q][1], dist[1]);
                    }
                }
            }
#ifdef PN2_FPS_DIAG
            const unsigned long long tu0 = __builtin_amdgcn_s_memtime();
            if (touched) tupd += tu0 - tprev;
#endif
            STAMP(0);  // box tests + update
            if (touched) {
#ifdef PN2_FPS_DIAG
                ++ntouch;
#endif
                // ---- this lane's best (exact key: slots are in cell order, ties go to the lowest ORIGINAL index) and the
                // largest d among its other slots, as max / min trees
                float mx[PPT];
#pragma unroll
                for (int j = 0; j < PPT; ++j) mx[j] = d[j >> 1][j & 1];
#pragma unroll
                for (int w = 1; w < PPT; w <<= 1)
#pragma unroll
                    for (int j = 0; j + w < PPT; j += 2 * w) mx[j] = fmaxf(mx[j], mx[j + w]);
                const float m1 = mx[0];
                unsigned mn[PPT];
#pragma unroll
                for (int j = 0; j < PPT; ++j) mn[j] = d[j >> 1][j & 1] == m1 ? pk[j] : 0xFFFFFFFFu;
#pragma unroll
                for (int w = 1; w < PPT; w <<= 1)
#pragma unroll
                    for (int j = 0; j + w < PPT; j += 2 * w) mn[j] = min(mn[j], mn[j + w]);
                const unsigned c1 = mn[0];
#pragma unroll
                for (int j = 0; j < PPT; ++j) mx[j] = pk[j] == c1 ? -1.0f : d[j >> 1][j & 1];
#pragma unroll
                for (int w = 1; w < PPT; w <<= 1)
#pragma unroll
                    for (int j = 0; j + w < PPT; j += 2 * w) mx[j] = fmaxf(mx[j], mx[j + w]);
                const float m2 = mx[0];
                const int j1 = (int)(c1 & 15u);
                const u64 k1 = m1 < 0.0f ? 0ull : (((u64)__float_as_uint(m1)) << 32) | (u64)(0xFFFFFFFFu - (c1 >> 4));
                const unsigned b2 = m2 < 0.0f ? 0u : __float_as_uint(m2);
                const float px = s_px[j1 * T + tid], py = s_py[j1 * T + tid], pz = s_pz[j1 * T + tid];
                // ---- wavefront: the best points of PER different lanes (key + coordinates) and the float bits of the
                // largest d among everything else (the other lanes' best points, every lane's runner-up)
                int o1, o2 = -1;
                const u64 w1 = wave_max_key_owner(k1, o1);
                u64 w2 = 0;
                if (PER == 2) w2 = wave_max_key_owner(lane == o1 ? 0ull : k1, o2);
                const unsigned wb = pn2::wave_max_u32((lane == o1 || lane == o2) ? b2 : (unsigned)(k1 >> 32));
                wmax = __uint_as_float((unsigned)(w1 >> 32));
                if (lane == o1) {
                    s_wkey[wave][0] = w1;
                    s_wbound[wave] = wb;
                    s_wxyz[wave][0][0] = px, s_wxyz[wave][0][1] = py, s_wxyz[wave][0][2] = pz;
                }
                if (PER == 2 && lane == o2) {
                    s_wkey[wave][PER - 1] = w2;
                    s_wxyz[wave][PER - 1][0] = px, s_wxyz[wave][PER - 1][1] = py, s_wxyz[wave][PER - 1][2] = pz;
                }
            }
#ifdef PN2_FPS_DIAG
            if (touched) tsel += __builtin_amdgcn_s_memtime() - tu0;
#endif
            STAMP(1);  // lane best + runner-up, wavefront candidates
            lds_barrier();
            STAMP(2);  // barrier 1
            if (wave == 0) {
                u64* slot = gb + (size_t)round * kGran * G;
                const unsigned tag = (unsigned)(round + 1);
                // ---- member: its PER best points among the wavefronts' candidates + the bound on everything else it holds
                // (the other candidates and every wavefront's bound); the lanes that hold the winners publish
                constexpr int W = NW * PER;   // candidates in lanes 0 .. W-1 (wavefront l / PER, candidate l % PER)
                const int cw = lane < W ? lane / PER : 0, cc = lane < W ? lane % PER : 0;
                const u64 mine = lane < W ? s_wkey[cw][cc] : 0ull;
                const unsigned mb = lane < NW ? s_wbound[lane] : 0u;
                const float qx = s_wxyz[cw][cc][0], qy = s_wxyz[cw][cc][1], qz = s_wxyz[cw][cc][2];
                int o1, o2 = -1;
                const u64 best = row_max_key_owner<(W <= 8 ? 8 : 16)>(mine, o1);
                u64 best2 = 0;
                if (PER == 2) best2 = row_max_key_owner<(W <= 8 ? 8 : 16)>(lane == o1 ? 0ull : mine, o2);
                const unsigned rest = max((lane == o1 || lane == o2) ? 0u : (unsigned)(mine >> 32), mb);
                const unsigned bound = W <= 8 ? max8_u32(rest) : max16_u32(rest);
                // every candidate slot is published every round (a member with a single point left lists key 0 second)
                const int pub2 = PER == 2 ? ((o2 < 0 || o2 == o1) ? ((o1 + 1) & 15) : o2) : -1;
                if (lane == o1 || lane == pub2) {
                    const int c = lane == o1 ? 0 : 1;
                    u64* dst = slot + (size_t)(4 * c) * G + g;
                    const u64 v0 = (c == 0 ? best : best2) | kValid;
                    const u64 v1 = (((u64)tag) << 32) | (u64)__float_as_uint(qx);
                    const u64 v2 = (((u64)tag) << 32) | (u64)__float_as_uint(qy);
                    const u64 v3 = (((u64)tag) << 32) | (u64)__float_as_uint(qz);
                    const u64 v4 = (((u64)bound) << 32) | 0xFFFFFFFFull | kValid;
                    u64* bdst = slot + (size_t)(4 * PER) * G + g;
                    if (local) {
                        __hip_atomic_store(dst, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + G, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + 2 * G, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(dst + 3 * G, v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (c == 0) __hip_atomic_store(bdst, v4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        st_granule(dst, v0);
                        st_granule(dst + G, v1);
                        st_granule(dst + 2 * G, v2);
                        st_granule(dst + 3 * G, v3);
                        if (c == 0) st_granule(bdst, v4);
                    }
                }
                STAMP(3);  // member candidates + publish
                // ---- poll: lane l reads the four granules of candidate pc of member pm and the member's bound
                u64 ek = 0;
                unsigned hb = 0, ex = 0, ey = 0, ez = 0;
                bool dead = false;
                {
                    const u64* src0 = slot + (size_t)(4 * pc) * G + (polls ? pm : 0);
                    const u64* srcb = slot + (size_t)(4 * PER) * G + (polls ? pm : 0);
                    unsigned spins = 0;
                    for (;;) {
                        const u64 v0 = ld_granule(src0), v1 = ld_granule(src0 + G), v2 = ld_granule(src0 + 2 * G),
                                  v3 = ld_granule(src0 + 3 * G), v4 = ld_granule(srcb);
                        const bool ok = ((v0 & kValid) != 0) & ((unsigned)(v1 >> 32) == tag) & ((unsigned)(v2 >> 32) == tag) &
                                        ((unsigned)(v3 >> 32) == tag) & ((v4 & kValid) != 0);
                        ek = polls ? (v0 & ~kValid) : 0ull;
                        hb = polls ? (unsigned)((v4 & ~kValid) >> 32) : 0u;
                        ex = (unsigned)v1, ey = (unsigned)v2, ez = (unsigned)v3;
                        if (__all(ok)) break;
                        if (!spin_alive(spins, kn, &hdr->err, kStatusHandoff)) {  // wave-uniform
                            dead = true;   // reported to the whole workgroup through s_m below
                            break;
                        }
                    }
                }
                STAMP(4);  // poll
                // ---- the sequential algorithm on the listed candidates
                const unsigned hbmax = pn2::wave_max_u32(hb);
                // (a key is above H = (hbmax, all ones) exactly when its distance bits are above hbmax; the distance bits
                // and the index word are kept apart so that a step is one 32-bit reduction, one ballot, four readlanes)
                const float fx = __uint_as_float(ex), fy = __uint_as_float(ey), fz = __uint_as_float(ez);
                const unsigned klo = (unsigned)ek;
                unsigned chi = (unsigned)(ek >> 32);
                bool alive = ek != 0ull;
                int mypos = -1;         // the position this lane's candidate was accepted at
                unsigned mypos_hi = 0;  // ... and its distance bits at that moment
                int acc = 0;
                const int room = npoint - count < kCap ? npoint - count : kCap;
                for (;;) {
                    const unsigned mh = pn2::wave_max_u32(alive ? chi : 0u);
                    u64 who = __ballot(alive && chi == mh);
                    if (who == 0ull || acc >= room || (acc > 0 && mh <= hbmax)) break;   // scalar
                    if (who & (who - 1ull)) {   // equal distances: the lowest original index (largest low word) wins
                        const unsigned ml = pn2::wave_max_u32((alive && chi == mh) ? klo : 0u);
                        who = __ballot(alive && chi == mh && klo == ml);
                    }
                    const int o = (int)__builtin_ctzll(who);
                    const float cx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ex, o));
                    const float cy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ey, o));
                    const float cz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ez, o));
                    const bool me = lane == o;
                    mypos = me ? acc : mypos;
                    mypos_hi = me ? mh : mypos_hi;
                    alive = alive && !me;
                    // what the update with this sample will do to the other listed points (same expression, same bits)
                    const f2 px2 = {fx, fx}, py2 = {fy, fy}, pz2 = {fz, fz};
                    const f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
                    const f2 dx = px2 - c2x, dy = py2 - c2y, dz = pz2 - c2z;
                    const f2 dist = (dx * dx + dy * dy) + dz * dz;
                    chi = __float_as_uint(fminf(__uint_as_float(chi), dist[0]));
                    ++acc;
                }
                if (mypos >= 0) {
                    ListEntry e;
                    e.key = (((u64)mypos_hi) << 32) | (u64)klo, e.x = fx, e.y = fy, e.z = fz, e.pad = 0.0f;
                    s_list[buf][mypos] = e;
                }
                if (lane == 0) s_m[buf] = dead ? -1 : acc;   // -1: the launch is dead, everybody leaves after the barrier
                STAMP(6);  // simulation
                lds_barrier();
                STAMP(8);  // barrier 2
                if (g == 0 && lane < acc && !dead) {   // off the critical path: the others are released
                    const ListEntry e = s_list[buf][lane];
                    const size_t o = (size_t)b * npoint + count + lane;
                    out_idx[o] = (int)(0xFFFFFFFFu - (unsigned)(e.key & 0xFFFFFFFFull));
                    if (out_xyz) out_xyz[o * 3] = e.x, out_xyz[o * 3 + 1] = e.y, out_xyz[o * 3 + 2] = e.z;
                }
                STAMP(5);  // result stores
            } else {
                lds_barrier();
            }
            m = s_m[buf];
            if (m <= 0) return;    // dead launch (uniform: every thread reads the same word after the same barrier)
            {
                const ListEntry e = s_list[buf][lane < kCap ? lane : 0];
                mcx = e.x, mcy = e.y, mcz = e.z;
            }
            count += m;
        }
        if (tid == 0 && g == 0 && b == 0) hdr->pad[0] = (unsigned)rounds_done;   // exchanges cloud 0 took (pn2_fps_rounds_offset)
#ifdef PN2_FPS_DIAG
        if (tid == 0 && g == 0 && b == 0) {
            unsigned long long* dbg = (unsigned long long*)((char*)hdr + 64);   // first granule bytes (diag only)
            for (int k = 0; k < 6; ++k) dbg[k] = st[k];
            dbg[6] = __builtin_amdgcn_s_memtime() - ct0;
            dbg[7] = __builtin_amdgcn_s_memrealtime() - rt0;
            dbg[8] = (unsigned long long)local;
            dbg[9] = nrounds;
            dbg[10] = ntouch;
            dbg[11] = tupd;
            dbg[12] = tsel;
            dbg[13] = st[6], dbg[14] = st[7], dbg[15] = st[8], dbg[16] = st[9], dbg[17] = st[10];
        }
#endif
        __syncthreads();
    }
}

inline bool use_xcd_kernel(int N) {
    if (getenv("PN2_FPS_NO_XCD")) return false;
    return N > kXT * kXPPT && N <= 64 * kXT * kXPPT;
}
inline int xcd_group_size(int N) { return pn2::ceil_div(N, kXT * kXPPT); }
// Points per lane of the multi-pick kernel: the fewest (4, 8 or 16) whose group still fits the 32 CUs of one XCD.
// More members per cloud = less update / selection work per workgroup AND more far-apart candidates per round, as long
// as the hand-off stays inside an XCD's L2 (measured, 1024 samples: 8 x 65536 points 1.54 ms at 16 points per lane /
// 8 members, 0.96 ms at 4 / 32; one 262144-point cloud 1.37 ms at 16 / 32 but 1.55 / 1.64 ms when 64 / 128 members
// have to cross XCDs).
inline int multi_ppt(int N) {
    if (const char* e = getenv("PN2_FPS_PPT")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16) return v;
    }
    const int cands[3] = {4, 8, 16};
    for (int ppt : cands)
        if (pn2::ceil_div(N, kXT * ppt) <= 32) return ppt;
    return 16;
}
inline int multi_group_size(int N) { return pn2::ceil_div(N, kXT * multi_ppt(N)); }
// Multi-pick rounds pay off once a group has enough members to offer several far-apart candidates per round
// (measured at 16 points per lane: G = 32 1.73 -> 1.35 ms, G = 8 1.64 -> 1.50 ms, G = 3 0.82 -> 0.99 ms).
// ... and once there are enough samples: the first rounds (huge radii) accept one sample each, so at npoint = 100 the
// heavier round does not pay yet (0.20 vs 0.19 ms), at 512 it does (0.57 vs 0.82 ms).
inline bool use_multi_pick(int N, int npoint) {
    return getenv("PN2_FPS_NO_MULTI") == nullptr && npoint >= 200 && multi_group_size(N) >= 8 && multi_group_size(N) <= 128;
}
// every wave publishes its own candidate when one poll still covers all entries with one load set per lane
inline bool xcd_perwave(int N) { return xcd_group_size(N) * (kXT / 64) <= 64; }

struct Config {
    int ppt, t, G, groups;
    double cost;
};

// Pick (PPT, T, G): G members share one cloud; at most 256 workgroups so that every member is resident.
// PN2_FPS_CFG="ppt,t" (tuning aid) pins the candidate.
Config pick(int B, int N) {
    static const int cand[][2] = {{1, 256}, {2, 256}, {4, 256}, {8, 256}, {16, 256}, {4, 1024}, {8, 1024}, {16, 512}, {32, 512}};
    int force_p = 0, force_t = 0;
    if (const char* e = getenv("PN2_FPS_CFG")) sscanf(e, "%d,%d", &force_p, &force_t);
    Config best{0, 0, 0, 0, 1e300};
    for (auto& c : cand) {
        const int ppt = c[0], t = c[1];
        if (force_p && (ppt != force_p || t != force_t)) continue;
        const int G = pn2::ceil_div(N, (long long)ppt * t);
        if (G > kMaxG) continue;
        const int groups = B < (256 / G) ? B : (256 / G);
        const int rounds = pn2::ceil_div(B, groups);
        const double step = 0.35 + (G > 1 ? 1.6 : 0.0) + 0.02 * ppt + 0.0004 * t;  // microseconds, rough
        const double cost = rounds * step;
        if (cost < best.cost) best = Config{ppt, t, G, groups, cost};
    }
    return best;
}

template <int PPT, int T>
void launch(const Config& c, const float* xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
            const int64_t* start, int32_t* out_idx, float* out_xyz, u64* gran, unsigned* err, const Knobs& kn, hipStream_t s,
            const int* coff = nullptr) {
    if (c.G > 1)
        PN2_LAUNCH("fps", (double)B * (12.0 * N + 8.0 * npoint), 0, (fps_kernel<PPT, T, true>), dim3(c.groups * c.G), dim3(T), s, xyz,
                   sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, err, c.G, c.groups, kn, coff);
    else if (coff)
        PN2_LAUNCH("fps", (double)B * (12.0 * N + 8.0 * npoint), 0, (fps_kernel<PPT, T, false, true>), dim3(c.groups * c.G), dim3(T), s,
                   xyz, sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, err, c.G, c.groups, kn, coff);
    else
        PN2_LAUNCH("fps", (double)B * (12.0 * N + 8.0 * npoint), 0, (fps_kernel<PPT, T, false>), dim3(c.groups * c.G), dim3(T), s, xyz,
                   sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, err, c.G, c.groups, kn, coff);
}

Knobs knobs(int32_t* status) {
    Knobs kn{kSpinLimitDefault, 0, status};
    if (const char* e = getenv("PN2_FPS_SPIN_LIMIT")) kn.spin_limit = (unsigned)strtoul(e, nullptr, 10);   // test aid
    kn.force_fallback = getenv("PN2_FPS_FORCE_FALLBACK") != nullptr;
    return kn;
}


}  // namespace

// Workspace of the spatially ordered path behind the header and the granules: [boxes][cell totals][cell cursors][order].
struct OrderLayout {
    size_t box, total, cursor, cellstart, order, sorted_xyz, end;
};
inline OrderLayout order_layout(size_t head, int B, int N) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    OrderLayout L;
    L.box = up(head);
    L.total = up(L.box + (size_t)B * 8 * sizeof(unsigned));
    L.cursor = L.total + (size_t)B * kCells * sizeof(unsigned);
    L.cellstart = up(L.cursor + (size_t)B * kCells * sizeof(unsigned));
    L.order = up(L.cellstart + (size_t)B * (kCells + 1) * sizeof(unsigned));
    L.sorted_xyz = up(L.order + (size_t)B * N * sizeof(unsigned));
    L.end = L.sorted_xyz + (size_t)B * 3 * N * sizeof(float);
    return L;
}
inline bool use_sorted(int B, int N) {
    return getenv("PN2_FPS_NO_SORT") == nullptr && N < (1 << 27) && B <= 4096 && multi_group_size(N) <= 64;
}
// listed candidates per member: two while 2 G candidates fit one wavefront (PN2_FPS_PER=1: A/B aid)
inline int sorted_per(int N) {
    if (const char* e = getenv("PN2_FPS_PER"))
        if (atoi(e) == 1) return 1;
    return multi_group_size(N) <= 32 ? 2 : 1;
}

// header + granules of the XCD kernels
inline size_t xcd_plain_bytes(int B, int N, int npoint) {
    if (use_multi_pick(N, npoint))   // 4 PER + 1 granules per (cloud, round, member): 9 for the ordered kernel with two candidates
        return sizeof(XcdHeader) + (size_t)B * npoint * (multi_group_size(N) <= 32 ? 9 : 5) * multi_group_size(N) * sizeof(u64);
    return sizeof(XcdHeader) + (size_t)B * npoint * 4 * xcd_group_size(N) * (xcd_perwave(N) ? kXT / 64 : 1) * sizeof(u64);
}

extern "C" size_t pn2_fps_workspace_bytes(int B, int N, int npoint) {
    if (B <= 0 || N <= 0 || npoint <= 0) return 0;
    if (use_xcd_kernel(N)) {
        if (use_multi_pick(N, npoint)) return order_layout(xcd_plain_bytes(B, N, npoint), B, N).end;
        return xcd_plain_bytes(B, N, npoint);
    }
    const Config c = pick(B, N);
    if (c.G == 0) return 0;
    // [err word padded to kHdr bytes][granules: 4 per (cloud, step, member)]
    return kHdr + (c.G > 1 ? (size_t)B * npoint * 4 * c.G * sizeof(u64) : 0);
}

// Where pn2_fps_f32 leaves the cell order of the clouds it sampled (int32 [B][N], a permutation of every cloud's point
// indices: neighbours in space are neighbours in the array) inside its workspace, or (size_t)-1 when this problem does not
// take the ordered kernel.  Valid until the workspace is written again; other kernels may use it to schedule by locality
// (pn2_three_nn_f32's `order`) -- only ever as a permutation, never for a result.
extern "C" size_t pn2_fps_order_offset(int B, int N, int npoint) {
    if (B <= 0 || N <= 0 || npoint <= 0 || !use_xcd_kernel(N) || !use_multi_pick(N, npoint) || !use_sorted(B, N)) return (size_t)-1;
    return order_layout(xcd_plain_bytes(B, N, npoint), B, N).order;
}
// ... the number of exchanges the ordered kernel needed for cloud 0 (uint32, written when the launch ends; profiling aid) ...
extern "C" size_t pn2_fps_rounds_offset(int B, int N, int npoint) {
    if (pn2_fps_order_offset(B, N, npoint) == (size_t)-1) return (size_t)-1;
    return offsetof(XcdHeader, pad);
}
// ... and the cell structure behind it: the cloud's box (uint32 [B][8], csrc/pn2_cells.h) and the first sorted position of
// every cell (int32 [B][4097], Morton-numbered 16 x 16 x 16 cells, last entry N).
extern "C" size_t pn2_fps_box_offset(int B, int N, int npoint) {
    if (pn2_fps_order_offset(B, N, npoint) == (size_t)-1) return (size_t)-1;
    return order_layout(xcd_plain_bytes(B, N, npoint), B, N).box;
}
extern "C" size_t pn2_fps_cellstart_offset(int B, int N, int npoint) {
    if (pn2_fps_order_offset(B, N, npoint) == (size_t)-1) return (size_t)-1;
    return order_layout(xcd_plain_bytes(B, N, npoint), B, N).cellstart;
}
// ... and the coordinates in cell order: float [B][3][N] (x, y, z planes; entry p of a plane belongs to point order[p])
extern "C" size_t pn2_fps_sorted_xyz_offset(int B, int N, int npoint) {
    if (pn2_fps_order_offset(B, N, npoint) == (size_t)-1) return (size_t)-1;
    return order_layout(xcd_plain_bytes(B, N, npoint), B, N).sorted_xyz;
}

extern "C" int pn2_fps_f32(const float* xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
                           const int64_t* start, int32_t* out_idx, float* out_xyz, void* workspace,
                           size_t workspace_bytes, int32_t* status, void* stream) {
    if (!xyz || !start || !out_idx || !workspace || B <= 0 || N <= 0 || npoint <= 0) return PN2_E_BADARG;
    const Knobs kn = knobs(status);
    if (use_xcd_kernel(N)) {
        const size_t need = pn2_fps_workspace_bytes(B, N, npoint);
        if (workspace_bytes < need) return PN2_E_WORKSPACE;
        hipStream_t s = (hipStream_t)stream;
        XcdHeader* hdr = (XcdHeader*)workspace;
        u64* gran = (u64*)((char*)workspace + sizeof(XcdHeader));
        if (use_multi_pick(N, npoint) && use_sorted(B, N)) {
            const int ppt = multi_ppt(N), G = multi_group_size(N);
            const OrderLayout L = order_layout(xcd_plain_bytes(B, N, npoint), B, N);
            char* w = (char*)workspace;
            PN2_HIP_CHECK(hipMemsetAsync(workspace, 0, L.cellstart, s));   // header, granules, boxes, cell counters
            unsigned* box = (unsigned*)(w + L.box);
            unsigned* total = (unsigned*)(w + L.total);
            unsigned* cursor = (unsigned*)(w + L.cursor);
            unsigned* order = (unsigned*)(w + L.order);
            {
                pn2::prof::Scope sc_("fps_order", s, (double)B * N * (3 * 12.0 + 4.0), 0.0);
                const dim3 grid(pn2::ceil_div(N, kOT * kOP), B);
                hipLaunchKernelGGL(fps_box_kernel, grid, dim3(kOT), 0, s, xyz, sb, sn, sc, N, box);
                hipLaunchKernelGGL(fps_hist_kernel, grid, dim3(kOT), 0, s, xyz, sb, sn, sc, N, box, total);
                hipLaunchKernelGGL(fps_scatter_kernel, grid, dim3(kOT), 0, s, xyz, sb, sn, sc, N, box, total, cursor, order,
                                   (unsigned*)(w + L.cellstart), (float*)(w + L.sorted_xyz));
            }
            const double fb = (double)B * (12.0 * N + 8.0 * npoint);
            const int per = sorted_per(N);
#define PN2_FPS_SORTED(P, K)                                                                                                     \
    if (ppt == P && per == K)                                                                                                     \
        PN2_LAUNCH("fps", fb, 0, (fps_sorted_kernel<P, K>), dim3(kXGrid), dim3(kXT), s, xyz, sb, sn, sc, B, N, npoint, start,      \
                   (const unsigned*)order, out_idx, out_xyz, gran, hdr, G, kn);
            PN2_FPS_SORTED(4, 1)
            PN2_FPS_SORTED(8, 1)
            PN2_FPS_SORTED(16, 1)
            PN2_FPS_SORTED(4, 2)
            PN2_FPS_SORTED(8, 2)
            PN2_FPS_SORTED(16, 2)
#undef PN2_FPS_SORTED
            PN2_LAUNCH_CHECK();
            return 0;
        }
        PN2_HIP_CHECK(hipMemsetAsync(workspace, 0, xcd_plain_bytes(B, N, npoint), s));   // header + granules
        if (use_multi_pick(N, npoint)) {
            const int ppt = multi_ppt(N), G = multi_group_size(N);
            const double fb = (double)B * (12.0 * N + 8.0 * npoint);
            if (ppt == 4)
                PN2_LAUNCH("fps", fb, 0, (fps_multi_kernel<4>), dim3(kXGrid), dim3(kXT), s, xyz, sb, sn, sc, B, N, npoint, start,
                           out_idx, out_xyz, gran, hdr, G, kn);
            else if (ppt == 8)
                PN2_LAUNCH("fps", fb, 0, (fps_multi_kernel<8>), dim3(kXGrid), dim3(kXT), s, xyz, sb, sn, sc, B, N, npoint, start,
                           out_idx, out_xyz, gran, hdr, G, kn);
            else
                PN2_LAUNCH("fps", fb, 0, (fps_multi_kernel<16>), dim3(kXGrid), dim3(kXT), s, xyz, sb, sn, sc, B, N, npoint, start,
                           out_idx, out_xyz, gran, hdr, G, kn);
        }
        else if (xcd_perwave(N))
            PN2_LAUNCH("fps", (double)B * (12.0 * N + 8.0 * npoint), 0, (fps_xcd_kernel<true>), dim3(kXGrid), dim3(kXT), s, xyz, sb,
                       sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, hdr, xcd_group_size(N), kn);
        else
            PN2_LAUNCH("fps", (double)B * (12.0 * N + 8.0 * npoint), 0, (fps_xcd_kernel<false>), dim3(kXGrid), dim3(kXT), s, xyz, sb,
                       sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, hdr, xcd_group_size(N), kn);
        PN2_LAUNCH_CHECK();
        return 0;
    }
    if (use_wave_kernel(N)) return launch_wave<false>(xyz, sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, nullptr, (hipStream_t)stream);
    const Config c = pick(B, N);
    if (c.G == 0) return PN2_E_BADARG;  // N beyond 64 members x 16384 points (1,048,576)
    const size_t need = pn2_fps_workspace_bytes(B, N, npoint);
    if (workspace_bytes < need) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (c.G > 1) PN2_HIP_CHECK(hipMemsetAsync(workspace, 0, need, s));   // one workgroup per cloud: no hand-off, no error word
    unsigned* err = (unsigned*)workspace;
    u64* gran = (u64*)((char*)workspace + kHdr);
    if (c.G > 1) {   // consecutive-block groups have no arrival phase to order an in-kernel fill: two memset nodes
        PN2_HIP_CHECK(hipMemsetAsync(out_idx, 0xFF, (size_t)B * npoint * sizeof(int32_t), s));
        if (out_xyz) PN2_HIP_CHECK(hipMemsetAsync(out_xyz, 0xFF, (size_t)B * npoint * 3 * sizeof(float), s));
    }
#define PN2_FPS_CASE(P, T_)                                                                                  \
    if (c.ppt == P && c.t == T_) {                                                                           \
        launch<P, T_>(c, xyz, sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, err, kn, s);          \
    } else
    PN2_FPS_CASE(1, 256)
    PN2_FPS_CASE(2, 256)
    PN2_FPS_CASE(4, 256)
    PN2_FPS_CASE(8, 256)
    PN2_FPS_CASE(16, 256)
    PN2_FPS_CASE(4, 1024)
    PN2_FPS_CASE(8, 1024)
    PN2_FPS_CASE(16, 512)
    PN2_FPS_CASE(32, 512) { return PN2_E_BADARG; }
#undef PN2_FPS_CASE
    PN2_LAUNCH_CHECK();
    return 0;
}

// Ragged batch of small clouds (every cloud <= 16384 points: one workgroup each, no hand-off, nothing to time out).
// N_max picks the points-per-lane configuration; clouds shorter than that leave lanes idle.
extern "C" size_t pn2_fps_ragged_workspace_bytes(int C, int n_max, int npoint) {
    if (C <= 0 || n_max <= 0 || n_max > 32 * 512 || npoint <= 0) return 0;
    return kHdr;
}

extern "C" int pn2_fps_ragged_f32(const float* xyz_cf, const int32_t* coff, int C, int n_max, int npoint, const int64_t* start,
                                  int32_t* out_idx, float* out_xyz, void* workspace, size_t workspace_bytes, void* stream) {
    if (!xyz_cf || !coff || !start || !out_idx || !workspace || C <= 0 || n_max <= 0 || n_max > 32 * 512 || npoint <= 0)
        return PN2_E_BADARG;
    if (workspace_bytes < kHdr) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (use_wave_kernel(n_max)) return launch_wave<true>(xyz_cf, 0, 1, 0, C, n_max, npoint, start, out_idx, out_xyz, (const int*)coff, s);
    // one member per cloud: the smallest (points per lane x threads) that covers the longest cloud
    static const int cand[][2] = {{1, 256}, {2, 256}, {4, 256}, {8, 256}, {16, 256}, {4, 1024}, {8, 1024}, {16, 512}, {32, 512}};
    Config c{0, 0, 1, C < 256 ? C : 256, 0.0};
    for (auto& k : cand)
        if (k[0] * k[1] >= n_max) {
            if (!c.ppt || 0.02 * k[0] + 0.0004 * k[1] < 0.02 * c.ppt + 0.0004 * c.t) c.ppt = k[0], c.t = k[1];
        }
    if (!c.ppt) return PN2_E_BADARG;
    const Knobs kn = knobs(nullptr);
    unsigned* err = (unsigned*)workspace;
#define PN2_FPS_CASE(P, T_)                                                                                              \
    if (c.ppt == P && c.t == T_) {                                                                                       \
        launch<P, T_>(c, xyz_cf, 0, 1, 0, C, n_max, npoint, start, out_idx, out_xyz, nullptr, err, kn, s, (const int*)coff); \
    } else
    PN2_FPS_CASE(1, 256)
    PN2_FPS_CASE(2, 256)
    PN2_FPS_CASE(4, 256)
    PN2_FPS_CASE(8, 256)
    PN2_FPS_CASE(16, 256)
    PN2_FPS_CASE(4, 1024)
    PN2_FPS_CASE(8, 1024)
    PN2_FPS_CASE(16, 512)
    PN2_FPS_CASE(32, 512) { return PN2_E_BADARG; }
#undef PN2_FPS_CASE
    PN2_LAUNCH_CHECK();
    return 0;
}
