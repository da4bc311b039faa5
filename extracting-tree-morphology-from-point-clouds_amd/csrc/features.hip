// kNN feature helpers for gfx950 -- replaces the neighbourhood work of Modules/Features.py:111-175
// (compute_normals_ckdtree, compute_curvature_ckdtree, compute_density_ckdtree; driven by add_features :178-229).
//
// The reference builds a scipy cKDTree over float64 points, queries the k nearest neighbours of every point (the
// point itself comes first), and then runs a Python loop of np.cov + np.linalg.svd / eigvalsh per point; density is a
// radius query per point.  Everything there is float64, so it is float64 here (MI355X runs fp64 vector math at full
// rate):
//   * knn_radius_kernel: one thread per query, the cloud streamed through LDS in tiles, a sorted top-k list in
//     registers (insertion behind a wave-uniform "any lane improves?" ballot -- after the first few hundred points
//     almost no tile position triggers it), the radius count in the same pass.  Neighbours come out ascending by
//     (squared distance, index).
//   * cov_eig_kernel: one thread per point: offsets to its first k neighbours, np.cov's unbiased covariance, cyclic
//     Jacobi on the 3x3 symmetric matrix (relative accuracy down to the smallest eigenvalue), eigenvalues ascending
//     with their eigenvectors; every eigenvector is normalised to a positive largest component (the reference's signs
//     are whatever LAPACK returns and are not reproducible).
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;   // points per LDS tile: 3 x 8 KiB
constexpr int kMaxK = 16;

template <int K>
__global__ __launch_bounds__(kBlock) void knn_radius_kernel(const double* __restrict__ pts, int N, int k, double r2,
                                                            int32_t* __restrict__ nn_idx, double* __restrict__ nn_d2,
                                                            int32_t* __restrict__ count) {
    __shared__ double sx[kTile], sy[kTile], sz[kTile];
    const int q = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = q < N;
    const double qx = pts[(size_t)(valid ? q : 0) * 3], qy = pts[(size_t)(valid ? q : 0) * 3 + 1],
                 qz = pts[(size_t)(valid ? q : 0) * 3 + 2];
    double bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = __builtin_inf();
        bi[j] = 0;
    }
    int cnt = 0;
    for (int t0 = 0; t0 < N; t0 += kTile) {
        const int n = (N - t0) < kTile ? (N - t0) : kTile;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock) {
            sx[i] = pts[(size_t)(t0 + i) * 3];
            sy[i] = pts[(size_t)(t0 + i) * 3 + 1];
            sz[i] = pts[(size_t)(t0 + i) * 3 + 2];
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            const double dx = sx[i] - qx, dy = sy[i] - qy, dz = sz[i] - qz;
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
            cnt += d <= r2 ? 1 : 0;
            if (__ballot(d < bd[K - 1])) {  // wave-uniform skip; strict '<' keeps the lower index on ties
                const int s = t0 + i;
                // sorted insertion, highest slot first so every slot sees its old neighbour
#pragma unroll
                for (int j = K - 1; j >= 1; --j) {
                    const bool below = d < bd[j - 1];   // belongs before slot j-1 -> slot j takes its old left neighbour
                    const bool here = d < bd[j];
                    const double nd = below ? bd[j - 1] : (here ? d : bd[j]);
                    const int ni = below ? bi[j - 1] : (here ? s : bi[j]);
                    bd[j] = nd;
                    bi[j] = ni;
                }
                const bool first = d < bd[0];
                bi[0] = first ? s : bi[0];
                bd[0] = first ? d : bd[0];
            }
        }
    }
    if (!valid) return;
    for (int j = 0; j < k; ++j) {
        nn_idx[(size_t)q * k + j] = bi[j];
        if (nn_d2) nn_d2[(size_t)q * k + j] = bd[j];
    }
    if (count) count[q] = cnt;
}

// cyclic Jacobi for a symmetric 3x3 (a = [a00 a01 a02; . a11 a12; . . a22]); eigenvectors in the columns of v
__device__ void jacobi3(double a[3][3], double v[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = a[p][q];
            if (apq == 0.0) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a[p][p] -= t * apq;
            a[q][q] += t * apq;
            a[p][q] = a[q][p] = 0.0;
            const int r = 3 - p - q;
            const double arp = a[r][p], arq = a[r][q];
            a[r][p] = a[p][r] = c * arp - s * arq;
            a[r][q] = a[q][r] = s * arp + c * arq;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double vip = v[i][p], viq = v[i][q];
                v[i][p] = c * vip - s * viq;
                v[i][q] = s * vip + c * viq;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void cov_eig_kernel(const double* __restrict__ pts, int N,
                                                         const int32_t* __restrict__ nn_idx, int k_stride, int k,
                                                         double* __restrict__ evals, double* __restrict__ evecs) {
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= N) return;
    const double qx = pts[(size_t)q * 3], qy = pts[(size_t)q * 3 + 1], qz = pts[(size_t)q * 3 + 2];
    // np.cov(neighbours.T): rows are variables, mean over the k observations, divide by k - 1
    double ox[kMaxK], oy[kMaxK], oz[kMaxK];
    double mx = 0.0, my = 0.0, mz = 0.0;
    for (int j = 0; j < k; ++j) {
        const int n = nn_idx[(size_t)q * k_stride + j];
        ox[j] = pts[(size_t)n * 3] - qx;
        oy[j] = pts[(size_t)n * 3 + 1] - qy;
        oz[j] = pts[(size_t)n * 3 + 2] - qz;
        mx += ox[j], my += oy[j], mz += oz[j];
    }
    mx /= (double)k, my /= (double)k, mz /= (double)k;
    double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < k; ++j) {
        const double x = ox[j] - mx, y = oy[j] - my, z = oz[j] - mz;
        a[0][0] += x * x, a[0][1] += x * y, a[0][2] += x * z;
        a[1][1] += y * y, a[1][2] += y * z, a[2][2] += z * z;
    }
    const double inv = 1.0 / (double)(k - 1);
    a[0][0] *= inv, a[0][1] *= inv, a[0][2] *= inv, a[1][1] *= inv, a[1][2] *= inv, a[2][2] *= inv;
    a[1][0] = a[0][1], a[2][0] = a[0][2], a[2][1] = a[1][2];
    double v[3][3];
    jacobi3(a, v);
    // ascending eigenvalues
    int o0 = 0, o1 = 1, o2 = 2;
    double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
    if (e0 > e1) { const double t = e0; e0 = e1; e1 = t; const int ti = o0; o0 = o1; o1 = ti; }
    if (e1 > e2) { const double t = e1; e1 = e2; e2 = t; const int ti = o1; o1 = o2; o2 = ti; }
    if (e0 > e1) { const double t = e0; e0 = e1; e1 = t; const int ti = o0; o0 = o1; o1 = ti; }
    evals[(size_t)q * 3] = e0, evals[(size_t)q * 3 + 1] = e1, evals[(size_t)q * 3 + 2] = e2;
    const int ord[3] = {o0, o1, o2};
    for (int r = 0; r < 3; ++r) {
        double x = v[0][ord[r]], y = v[1][ord[r]], z = v[2][ord[r]];
        const double ax = fabs(x), ay = fabs(y), az = fabs(z);
        const double lead = (ax >= ay && ax >= az) ? x : (ay >= az ? y : z);
        if (lead < 0.0) x = -x, y = -y, z = -z;
        evecs[(size_t)q * 9 + r * 3] = x, evecs[(size_t)q * 9 + r * 3 + 1] = y, evecs[(size_t)q * 9 + r * 3 + 2] = z;
    }
}

}  // namespace

extern "C" int pn2_knn_radius_f64(const double* points, int N, int k, double r2, int32_t* nn_idx, double* nn_d2,
                                  int32_t* radius_count, void* stream) {
    if (!points || !nn_idx || N <= 0 || k <= 0 || k > kMaxK || k > N) return PN2_E_BADARG;
    const dim3 grid(pn2::ceil_div(N, kBlock)), block(kBlock);
    const double bytes = 24.0 * N + (double)N * k * (nn_d2 ? 12.0 : 4.0) + (radius_count ? 4.0 * N : 0.0);
    const double flops = 8.0 * (double)N * N;
    hipStream_t s = (hipStream_t)stream;
    if (k <= 8)
        PN2_LAUNCH("knn_radius", bytes, flops, (knn_radius_kernel<8>), grid, block, s, points, N, k, r2, nn_idx, nn_d2, radius_count);
    else
        PN2_LAUNCH("knn_radius", bytes, flops, (knn_radius_kernel<16>), grid, block, s, points, N, k, r2, nn_idx, nn_d2,
                   radius_count);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_cov_eig_f64(const double* points, int N, const int32_t* nn_idx, int k_stride, int k, double* evals,
                               double* evecs, void* stream) {
    if (!points || !nn_idx || !evals || !evecs || N <= 0 || k < 2 || k > kMaxK || k > k_stride) return PN2_E_BADARG;
    PN2_LAUNCH("cov_eig", (double)N * (24.0 * (k + 1) + 4.0 * k + 96.0), 0, cov_eig_kernel, dim3(pn2::ceil_div(N, kBlock)),
               dim3(kBlock), (hipStream_t)stream, points, N, nn_idx, k_stride, k, evals, evecs);
    PN2_LAUNCH_CHECK();
    return 0;
}
