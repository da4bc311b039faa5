/*
 * pn2_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * Scalar C restatement of the index/geometry ops of the reference's PointNet++ hot path,
 * written so that every fp32 rounding step is explicit and machine independent.
 * Compile with  gcc -O2 -ffp-contract=off -mfma  (see oracle/Makefile): the only fused
 * multiply-adds are the explicit fmaf() calls below.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (indices, distances,
 * weights) against the .npz files under tests/golden, which were produced by importing the reference's own
 * Modules/PointNet2 in the build container (tests/golden/make_golden.py).
 *
 * Reference lines restated (paths relative to the reference repo):
 *   pn2o_square_distance   Modules/PointNet2/pointnet2_utils.py:21-42
 *   pn2o_fps               Modules/PointNet2/pointnet2_utils.py:66-89
 *   pn2o_ball_query        Modules/PointNet2/pointnet2_utils.py:92-136
 *   pn2o_gather            Modules/PointNet2/pointnet2_utils.py:45-63   (index_points)
 *   pn2o_group             Modules/PointNet2/pointnet2_utils.py:139-167 (sample_and_group body)
 *   pn2o_three_nn          Modules/PointNet2/blocks.py:194-203
 *   pn2o_three_interpolate Modules/PointNet2/blocks.py:204
 *
 * Numerics contract (measured against the imported reference):
 *   dot(q,p)        = fma(qz,pz, fma(qy,py, qx*px))        (what MKL sgemm does for K=3)
 *   |v|^2           = (vx*vx + vy*vy) + vz*vz              (separate mul/add, no fma)
 *   sqdist(src,dst) = ((-2*dot) + |src|^2) + |dst|^2
 *   FPS distance    = ((dx*dx + dy*dy) + dz*dz)            (no fma)
 *   ties: max/argmin take the lowest index, sort is stable.
 *
 * All arrays are dense row-major; xyz arrays are [B][N][3] float, indices are int64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float dot3_mkl(const float *q, const float *p) {
    return fmaf(q[2], p[2], fmaf(q[1], p[1], q[0] * p[0]));
}

static inline float norm2(const float *v) {
    float a = v[0] * v[0];
    float b = v[1] * v[1];
    float c = v[2] * v[2];
    float ab = a + b;
    return ab + c;
}

static inline float sqdist_ref(const float *src, float src_n2, const float *dst, float dst_n2) {
    float d = -2.0f * dot3_mkl(src, dst);
    d = d + src_n2;
    d = d + dst_n2;
    return d;
}

/* out[b][n][m] = sqdist(src[b][n], dst[b][m]) */
void pn2o_square_distance(const float *src, const float *dst, int B, int N, int M, float *out) {
    for (int b = 0; b < B; ++b) {
        const float *s = src + (size_t)b * N * 3;
        const float *d = dst + (size_t)b * M * 3;
        float *o = out + (size_t)b * N * M;
        for (int n = 0; n < N; ++n) {
            float sn = norm2(s + 3 * n);
            for (int m = 0; m < M; ++m)
                o[(size_t)n * M + m] = sqdist_ref(s + 3 * n, sn, d + 3 * m, norm2(d + 3 * m));
        }
    }
}

/* Iterative farthest point sampling; start[b] is the reference's torch.randint draw. */
void pn2o_fps(const float *xyz, int B, int N, int npoint, const int64_t *start, int64_t *out) {
    float *mind = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int n = 0; n < N; ++n) mind[n] = 1e10f;
        int64_t far = start[b];
        for (int i = 0; i < npoint; ++i) {
            out[(size_t)b * npoint + i] = far;
            const float cx = p[3 * far], cy = p[3 * far + 1], cz = p[3 * far + 2];
            float best = -1.0f;
            int64_t besti = 0;
            for (int n = 0; n < N; ++n) {
                float dx = p[3 * n] - cx, dy = p[3 * n + 1] - cy, dz = p[3 * n + 2] - cz;
                float a = dx * dx, bq = dy * dy, c = dz * dz;
                float ab = a + bq;
                float d = ab + c;
                if (d < mind[n]) mind[n] = d;
                if (mind[n] > best) { best = mind[n]; besti = n; } /* first maximum wins */
            }
            far = besti;
        }
    }
    free(mind);
}

/*
 * Ball query.  out is [B][S][Keff], Keff = min(K, N).  r2 is float32(double(radius)**2).
 * A point is inside when NOT (d > r2).
 */
void pn2o_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int K,
                     int64_t *out) {
    int Keff = K < N ? K : N;
    float *pn = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        const float *q = new_xyz + (size_t)b * S * 3;
        for (int n = 0; n < N; ++n) pn[n] = norm2(p + 3 * n);
        for (int s = 0; s < S; ++s) {
            int64_t *o = out + ((size_t)b * S + s) * Keff;
            float qn = norm2(q + 3 * s);
            int cnt = 0;
            float dmin = INFINITY;
            int64_t imin = 0;
            int have_min = 0;
            for (int n = 0; n < N; ++n) {
                float d = sqdist_ref(q + 3 * s, qn, p + 3 * n, pn[n]);
                if (!(d > r2)) {
                    if (cnt < Keff) o[cnt] = n;
                    ++cnt;
                    if (cnt >= Keff) break;
                }
                if (!have_min || d < dmin) { dmin = d; imin = n; have_min = 1; }
            }
            if (cnt == 0) { /* empty ball: nearest point (argmin, first minimum) fills the row */
                /* the scan above never broke early, so dmin/imin cover all N */
                for (int k = 0; k < Keff; ++k) o[k] = imin;
            } else {
                for (int k = cnt; k < Keff; ++k) o[k] = o[0];
            }
        }
    }
    free(pn);
}

/* out[b][s][c] = points[b][idx[b][s]][c]   (index_points with a flattened index shape) */
void pn2o_gather(const float *points, const int64_t *idx, int B, int N, int C, int S, float *out) {
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            int64_t j = idx[(size_t)b * S + s];
            memcpy(out + ((size_t)b * S + s) * C, points + ((size_t)b * N + j) * C, sizeof(float) * (size_t)C);
        }
}

/* gradient of pn2o_gather: dpoints[b][idx[b][s]][c] += dout[b][s][c], sequential in s */
void pn2o_gather_grad(const float *dout, const int64_t *idx, int B, int N, int C, int S, float *dpoints) {
    memset(dpoints, 0, sizeof(float) * (size_t)B * N * C);
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            int64_t j = idx[(size_t)b * S + s];
            for (int c = 0; c < C; ++c)
                dpoints[((size_t)b * N + j) * C + c] += dout[((size_t)b * S + s) * C + c];
        }
}

/*
 * sample_and_group body after FPS/ball query:
 *   out[b][s][k][0:3]   = xyz[b][idx[b][s][k]] - new_xyz[b][s]
 *   out[b][s][k][3:3+D] = feats[b][idx[b][s][k]]          (D may be 0)
 * xyz_last != 0 gives the MSG channel order [feats, xyz_norm] (blocks.py:143-146).
 */
void pn2o_group(const float *xyz, const float *new_xyz, const float *feats, const int64_t *idx, int B,
                int N, int S, int K, int D, int xyz_last, float *out) {
    int C = 3 + D;
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < K; ++k) {
                int64_t j = idx[((size_t)b * S + s) * K + k];
                float *o = out + (((size_t)b * S + s) * K + k) * C;
                float *ox = xyz_last ? o + D : o;
                float *of = xyz_last ? o : o + 3;
                for (int c = 0; c < 3; ++c)
                    ox[c] = xyz[((size_t)b * N + j) * 3 + c] - new_xyz[((size_t)b * S + s) * 3 + c];
                for (int c = 0; c < D; ++c) of[c] = feats[((size_t)b * N + j) * D + c];
            }
}

/*
 * three_nn: for every xyz1 point the 3 (or kk = min(3,S)) smallest sqdist(xyz1, xyz2) entries in
 * stable-sort order (ties -> lower index first).  dist/idx are [B][N][kk].
 */
void pn2o_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx, float *dist) {
    int kk = S < 3 ? S : 3;
    float *n2 = (float *)malloc(sizeof(float) * (size_t)S);
    for (int b = 0; b < B; ++b) {
        const float *p1 = xyz1 + (size_t)b * N * 3;
        const float *p2 = xyz2 + (size_t)b * S * 3;
        for (int s = 0; s < S; ++s) n2[s] = norm2(p2 + 3 * s);
        for (int n = 0; n < N; ++n) {
            float bd[3] = {INFINITY, INFINITY, INFINITY};
            int64_t bi[3] = {-1, -1, -1};
            float sn = norm2(p1 + 3 * n);
            for (int s = 0; s < S; ++s) {
                float d = sqdist_ref(p1 + 3 * n, sn, p2 + 3 * s, n2[s]);
                /* strict '<' keeps the earlier index ahead on ties (stable sort order) */
                if (bi[0] < 0 || d < bd[0]) {
                    bd[2] = bd[1]; bi[2] = bi[1]; bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = s;
                } else if (bi[1] < 0 || d < bd[1]) {
                    bd[2] = bd[1]; bi[2] = bi[1]; bd[1] = d; bi[1] = s;
                } else if (bi[2] < 0 || d < bd[2]) {
                    bd[2] = d; bi[2] = s;
                }
            }
            for (int k = 0; k < kk; ++k) {
                idx[((size_t)b * N + n) * kk + k] = bi[k];
                dist[((size_t)b * N + n) * kk + k] = bd[k];
            }
        }
    }
    free(n2);
}

/* w = (1/max(d,1e-6)) / sum_k(1/max(d_k,1e-6)); the 3-term sum is (r0 + r1) + r2 */
void pn2o_three_weights(const float *dist, size_t rows, float *w) {
    for (size_t i = 0; i < rows; ++i) {
        float r[3];
        for (int k = 0; k < 3; ++k) {
            float d = dist[3 * i + k];
            if (d < 1e-6f) d = 1e-6f;
            r[k] = 1.0f / d;
        }
        float s = r[0] + r[1];
        s = s + r[2];
        for (int k = 0; k < 3; ++k) w[3 * i + k] = r[k] / s;
    }
}

/* out[b][n][c] = (p[i0][c]*w0 + p[i1][c]*w1) + p[i2][c]*w2, separate mul/add */
void pn2o_three_interpolate(const float *points2, const int64_t *idx, const float *w, int B, int N, int S,
                            int D, float *out) {
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const int64_t *ii = idx + ((size_t)b * N + n) * 3;
            const float *ww = w + ((size_t)b * N + n) * 3;
            for (int c = 0; c < D; ++c) {
                float t0 = points2[((size_t)b * S + ii[0]) * D + c] * ww[0];
                float t1 = points2[((size_t)b * S + ii[1]) * D + c] * ww[1];
                float t2 = points2[((size_t)b * S + ii[2]) * D + c] * ww[2];
                float t01 = t0 + t1;
                out[((size_t)b * N + n) * D + c] = t01 + t2;
            }
        }
}

/* dpoints2[b][idx[b][n][k]][c] += dout[b][n][c] * w[b][n][k]; sequential (n, then k) order */
void pn2o_three_interpolate_grad(const float *dout, const int64_t *idx, const float *w, int B, int N, int S,
                                 int D, float *dpoints2) {
    memset(dpoints2, 0, sizeof(float) * (size_t)B * S * D);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < 3; ++k) {
                int64_t j = idx[((size_t)b * N + n) * 3 + k];
                float wk = w[((size_t)b * N + n) * 3 + k];
                for (int c = 0; c < D; ++c)
                    dpoints2[((size_t)b * S + j) * D + c] += dout[((size_t)b * N + n) * D + c] * wk;
            }
}


/* ------------------------------------------------------------------------------------------------------------
 * kNN feature helpers, float64 -- Modules/Features.py:111-175 (compute_normals_ckdtree, compute_curvature_ckdtree,
 * compute_density_ckdtree).  The neighbour search is scipy's cKDTree in the reference (third party, not under
 * /root/reference; any scipy): its published contract is "the k nearest by Euclidean distance, nearest first", restated
 * here as a brute-force scan ordered by (squared distance, index).  The per-point statistics follow the reference's
 * lines: np.cov of the offsets (mean removed, divided by k-1), eigen-decomposition of the symmetric 3x3. */
void pn2o_knn_radius_f64(const double *pts, int N, int k, double r2, int64_t *nn_idx, double *nn_d2, int64_t *count) {
    for (int q = 0; q < N; ++q) {
        double bd[64];
        int64_t bi[64];
        for (int j = 0; j < k; ++j) { bd[j] = INFINITY; bi[j] = 0; }
        int64_t c = 0;
        for (int n = 0; n < N; ++n) {
            const double dx = pts[3 * n] - pts[3 * q], dy = pts[3 * n + 1] - pts[3 * q + 1], dz = pts[3 * n + 2] - pts[3 * q + 2];
            const double d = (dx * dx + dy * dy) + dz * dz;
            if (d <= r2) ++c;
            if (d < bd[k - 1]) {
                int j = k - 1;
                while (j > 0 && d < bd[j - 1]) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
                bd[j] = d; bi[j] = n;
            }
        }
        for (int j = 0; j < k; ++j) { nn_idx[(size_t)q * k + j] = bi[j]; if (nn_d2) nn_d2[(size_t)q * k + j] = bd[j]; }
        if (count) count[q] = c;
    }
}

static void jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = i == j;
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (a[p][q] == 0.0) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double r[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            r[p][p] = c; r[q][q] = c; r[p][q] = s; r[q][p] = -s;
            double ar[3][3], rtar[3][3], vr[3][3];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                ar[i][j] = 0; vr[i][j] = 0;
                for (int l = 0; l < 3; ++l) { ar[i][j] += a[i][l] * r[l][j]; vr[i][j] += v[i][l] * r[l][j]; }
            }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                rtar[i][j] = 0;
                for (int l = 0; l < 3; ++l) rtar[i][j] += r[l][i] * ar[l][j];
            }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = rtar[i][j]; v[i][j] = vr[i][j]; }
            a[p][q] = a[q][p] = 0.0;
        }
    }
}

/* evals [N][3] ascending; evecs [N][3][3]: row r = unit eigenvector of evals[r], largest component positive */
void pn2o_cov_eig_f64(const double *pts, int N, const int64_t *nn_idx, int k_stride, int k, double *evals, double *evecs) {
    for (int q = 0; q < N; ++q) {
        double o[64][3], m[3] = {0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const int64_t n = nn_idx[(size_t)q * k_stride + j];
            for (int c = 0; c < 3; ++c) { o[j][c] = pts[3 * n + c] - pts[3 * q + c]; m[c] += o[j][c]; }
        }
        for (int c = 0; c < 3; ++c) m[c] /= (double)k;
        double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, v[3][3];
        for (int j = 0; j < k; ++j)
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a[r][c] += (o[j][r] - m[r]) * (o[j][c] - m[c]);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a[r][c] /= (double)(k - 1);
        jacobi3(a, v);
        int ord[3] = {0, 1, 2};
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j)
            if (a[ord[j]][ord[j]] > a[ord[j + 1]][ord[j + 1]]) { const int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
        for (int r = 0; r < 3; ++r) {
            evals[(size_t)q * 3 + r] = a[ord[r]][ord[r]];
            double e[3] = {v[0][ord[r]], v[1][ord[r]], v[2][ord[r]]};
            int lead = 0;
            for (int c = 1; c < 3; ++c) if (fabs(e[c]) > fabs(e[lead])) lead = c;
            const double sgn = e[lead] < 0 ? -1.0 : 1.0;
            for (int c = 0; c < 3; ++c) evecs[(size_t)q * 9 + r * 3 + c] = sgn * e[c];
        }
    }
}

/* =====================================================================================================================
 * Round-2 additions.  PARITY UNPINNED: Modules/Projection.py and Modules/DataLoading/RasterizedTreeSet.py import
 * `fastprogress`, which is not installed here, so the reference cannot be imported to generate golden vectors for them;
 * the reference holds no fixtures or tests of its own for these functions.  What follows restates the reference's source
 * text line by line and is checked by closed-form geometric properties in tests/test_projection.py.
 * ===================================================================================================================== */

/* torch.sum(a * b, dim=2) over 3 components and torch.norm(v, dim=2): (x + y) + z, sqrt of it */
static float dot3p(float ax, float ay, float az, float bx, float by, float bz) { return (ax * bx + ay * by) + az * bz; }
static float norm3(float x, float y, float z) { return sqrtf(dot3p(x, y, z, x, y, z)); }
static float clampf_(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* One (point, cylinder) pair of closest_cylinder_cuda_batch, Modules/Projection.py:33-105.
 * want_point: 0 distance only; 1 also the final projection point f (mantle variant when `mantle`). */
static float cyl_pair(const float *p, const float *s, const float *u, float len, float rad, int want_point, int mantle, float *f) {
    const float vx = p[0] - s[0], vy = p[1] - s[1], vz = p[2] - s[2];                              /* :33 */
    const float pl = clampf_(dot3p(vx, vy, vz, u[0], u[1], u[2]), 0.0f, len);                      /* :36-40 */
    const float qx = s[0] + pl * u[0], qy = s[1] + pl * u[1], qz = s[2] + pl * u[2];               /* :41 */
    const float wx = p[0] - qx, wy = p[1] - qy, wz = p[2] - qz;                                    /* :44 */
    const float dp = dot3p(wx, wy, wz, u[0], u[1], u[2]);                                          /* :47 */
    const int perp = fabsf(dp) <= 1e-3f;                                                           /* :48 isclose(atol=1e-3) */
    const float rx = wx - dp * u[0], ry = wy - dp * u[1], rz = wz - dp * u[2];                     /* :51-52 */
    float nr = norm3(rx, ry, rz);                                                                  /* :55 */
    if (nr < 1e-8f) nr = 1e-8f;                                                                    /* :58-60 */
    const float ax = rx / nr, ay = ry / nr, az = rz / nr;                                          /* :61 */
    const float two_r = 2.0f * rad;
    const float hx = 0.5f * (ax * two_r), hy = 0.5f * (ay * two_r), hz = 0.5f * (az * two_r);      /* :64-68 */
    const float s0x = qx - hx, s0y = qy - hy, s0z = qz - hz;
    const float t = clampf_(dot3p(p[0] - s0x, p[1] - s0y, p[2] - s0z, ax, ay, az), 0.0f, two_r);   /* :71-75 */
    const float ox = s0x + t * ax, oy = s0y + t * ay, oz = s0z + t * az;                           /* :76 */
    const float ux = qx + ax * rad, uy = qy + ay * rad, uz = qz + az * rad;                        /* :79 */
    const float gx = perp ? ux : ox, gy = perp ? uy : oy, gz = perp ? uz : oz;                     /* :82 */
    const float dist = norm3(p[0] - gx, p[1] - gy, p[2] - gz);                                     /* :85 */
    if (want_point) {
        if (mantle) {                                                                              /* :90-105 */
            const float s1x = qx + hx, s1y = qy + hy, s1z = qz + hz;
            const float d0 = norm3(ox - s0x, oy - s0y, oz - s0z), d1 = norm3(ox - s1x, oy - s1y, oz - s1z);
            const int to_start = d0 < d1;
            f[0] = perp ? ux : (to_start ? s0x : s1x);
            f[1] = perp ? uy : (to_start ? s0y : s1y);
            f[2] = perp ? uz : (to_start ? s0z : s1z);
        } else {
            f[0] = gx, f[1] = gy, f[2] = gz;
        }
    }
    return dist;
}

/* closest_cylinder_cuda_batch (Modules/Projection.py:19-114) for all N points: ids (or cylinder index when ids == NULL),
 * distances, offsets = projection point - point.  argmin keeps the first minimum (:88). */
void pn2o_cylinder_project(const float *points, int N, const float *start, const float *unit, const float *length,
                           const float *radius, const int32_t *ids, int M, int mantle, int32_t *out_id, float *out_dist,
                           float *out_off) {
    for (int n = 0; n < N; ++n) {
        const float *p = points + 3 * (size_t)n;
        float best = INFINITY, f[3];
        int bi = 0;
        for (int m = 0; m < M; ++m) {
            const float d = cyl_pair(p, start + 3 * (size_t)m, unit + 3 * (size_t)m, length[m], radius[m], 0, 0, f);
            if (d < best) { best = d; bi = m; }
        }
        const float d = cyl_pair(p, start + 3 * (size_t)bi, unit + 3 * (size_t)bi, length[bi], radius[bi], 1, mantle, f);
        out_id[n] = ids ? ids[bi] : bi;
        out_dist[n] = d;
        for (int c = 0; c < 3; ++c) out_off[3 * (size_t)n + c] = f[c] - p[c];
    }
}

int pn2o_version(void) { return 2; }
