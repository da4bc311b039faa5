// farthest_point_sample for gfx950 -- replaces Modules/PointNet2/pointnet2_utils.py:66-89.
//
// FPS is npoint strictly sequential argmax steps, so the bound is the latency of one step, not bandwidth.
// Layout: every cloud is owned by a *group* of G workgroups (G = 1 for small clouds).  Each thread keeps
// PPT points and their running minimum distance in registers for the whole kernel (xyz is read from HBM
// exactly once: 12 B/point), a step is
//     register update -> wave max (DPP shuffles) -> LDS max over the workgroup's waves
//     -> [G > 1] one 8-byte granule per workgroup, stored write-through and polled by one wave of every
//        member workgroup (data-is-the-flag hand-off, placement independent, no grid barrier)
//     -> scalar load of the winner's coordinates.
// The argmax key is (dist_bits << 32) | ~index: distances are non-negative so their bit patterns order like
// the values, and the complemented index makes the LOWEST index win ties exactly like torch.max does
// (zero-padded clouds have many identical points).
//
// Granules are never reused inside a launch (one slot per (cloud, step, member)), the slot array is zeroed
// by a memset node ahead of the kernel, bit 63 marks a written slot, and every spin is bounded.
#include "pn2_common.h"

namespace {

using u64 = unsigned long long;
constexpr u64 kValid = 1ull << 63;
constexpr int kMaxG = 64;
constexpr unsigned kSpinLimit = 1u << 24;

template <int PPT, int T>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                int B, int N, int npoint, const int64_t* __restrict__ start,
                                                int32_t* __restrict__ out_idx, float* __restrict__ out_xyz,
                                                u64* gran, unsigned* err, int G, int groups) {
    constexpr int NW = T / 64;
    __shared__ u64 s_key[2][NW];
    __shared__ u64 s_win[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = blockIdx.x % G;
    const int grp = blockIdx.x / G;
    const int base = g * (T * PPT) + tid;

    for (int b = grp; b < B; b += groups) {
        const float* p = xyz + (int64_t)b * sb;
        float x[PPT], y[PPT], z[PPT], d[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int n = base + j * T;
            const bool ok = n < N;
            const float* q = p + (int64_t)(ok ? n : 0) * sn;
            x[j] = q[0];
            y[j] = q[sc];
            z[j] = q[2 * sc];
            d[j] = ok ? 1e10f : -1.0f;  // -1 marks a slot beyond N: never a maximum, never updated
        }
        int far = (int)start[b];
        u64* gb = gran + (size_t)b * npoint * G;

        for (int i = 0; i < npoint; ++i) {
            far = __builtin_amdgcn_readfirstlane(far);
            const float* c = p + (int64_t)far * sn;
            const float cx = c[0], cy = c[sc], cz = c[2 * sc];
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
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const float dx = __fsub_rn(x[j], cx), dy = __fsub_rn(y[j], cy), dz = __fsub_rn(z[j], cz);
                const float dist = pn2::norm2(dx, dy, dz);
                d[j] = dist < d[j] ? dist : d[j];
                if (d[j] > bestd) {  // strict: the lowest index of this thread wins
                    bestd = d[j];
                    bestj = j;
                }
            }
            const int bestn = base + bestj * T;
            u64 key = bestd < 0.0f ? 0ull : (((u64)__float_as_uint(bestd)) << 32) | (u64)(0xFFFFFFFFu - (unsigned)bestn);
            key = pn2::wave_max_u64(key);
            const int buf = i & 1;
            if (lane == 0) s_key[buf][wave] = key;
            __syncthreads();
            u64 k = s_key[buf][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const u64 o = s_key[buf][w];
                k = o > k ? o : k;
            }
            if (G > 1) {
                if (wave == 0) {
                    u64* slot = gb + (size_t)i * G;
                    if (lane == 0) __hip_atomic_store(slot + g, k | kValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 v = kValid;
                    if (lane < G) {
                        unsigned spins = 0;
                        for (;;) {
                            v = __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (v & kValid) break;
                            if (++spins > kSpinLimit) {  // a member never arrived: flag it and let the grid drain
                                atomicOr(err, 1u);
                                v = kValid;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    v = pn2::wave_max_u64(v & ~kValid);
                    if (lane == 0) s_win[buf] = v;
                }
                __syncthreads();
                k = s_win[buf];
            }
            far = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        }
        __syncthreads();  // LDS slots are reused by the next cloud of this group
    }
}

struct Config {
    int ppt, t, G, groups;
    double cost;
};

// Pick (PPT, T, G): G members share one cloud; at most 256 workgroups so that every member is resident.
Config pick(int B, int N) {
    static const int cand[][2] = {{1, 256}, {2, 256}, {4, 256}, {8, 256}, {4, 1024}, {8, 1024}, {16, 512}, {32, 512}};
    Config best{0, 0, 0, 0, 1e300};
    for (auto& c : cand) {
        const int ppt = c[0], t = c[1];
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
            const int64_t* start, int32_t* out_idx, float* out_xyz, u64* gran, unsigned* err, hipStream_t s) {
    hipLaunchKernelGGL((fps_kernel<PPT, T>), dim3(c.groups * c.G), dim3(T), 0, s, xyz, sb, sn, sc, B, N, npoint, start,
                       out_idx, out_xyz, gran, err, c.G, c.groups);
}

}  // namespace

extern "C" size_t pn2_fps_workspace_bytes(int B, int N, int npoint) {
    if (B <= 0 || N <= 0 || npoint <= 0) return 0;
    const Config c = pick(B, N);
    if (c.G == 0) return 0;
    // [err word padded to 16 B][granules]
    return 16 + (c.G > 1 ? (size_t)B * npoint * c.G * sizeof(u64) : 0);
}

extern "C" int pn2_fps_f32(const float* xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
                           const int64_t* start, int32_t* out_idx, float* out_xyz, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!xyz || !start || !out_idx || !workspace || B <= 0 || N <= 0 || npoint <= 0) return PN2_E_BADARG;
    const Config c = pick(B, N);
    if (c.G == 0) return PN2_E_BADARG;  // N beyond 64 members x 16384 points (1,048,576)
    const size_t need = pn2_fps_workspace_bytes(B, N, npoint);
    if (workspace_bytes < need) return PN2_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    PN2_HIP_CHECK(hipMemsetAsync(workspace, 0, need, s));
    unsigned* err = (unsigned*)workspace;
    u64* gran = (u64*)((char*)workspace + 16);
#define PN2_FPS_CASE(P, T_)                                                                                  \
    if (c.ppt == P && c.t == T_) {                                                                           \
        launch<P, T_>(c, xyz, sb, sn, sc, B, N, npoint, start, out_idx, out_xyz, gran, err, s);              \
    } else
    PN2_FPS_CASE(1, 256)
    PN2_FPS_CASE(2, 256)
    PN2_FPS_CASE(4, 256)
    PN2_FPS_CASE(8, 256)
    PN2_FPS_CASE(4, 1024)
    PN2_FPS_CASE(8, 1024)
    PN2_FPS_CASE(16, 512)
    PN2_FPS_CASE(32, 512) { return PN2_E_BADARG; }
#undef PN2_FPS_CASE
    PN2_LAUNCH_CHECK();
    return 0;
}
