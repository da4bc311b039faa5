/*
 * pn2_hip.h -- C ABI of libpn2hip.so, the MI355X (gfx950) implementation of the PointNet++ hot path.
 *
 * The reference has no FFI boundary of its own for this path: the boundary is the Python API of
 * Modules/PointNet2/pointnet2_utils.py and blocks.py.  Each entry point below replaces the
 * device work of one reference expression group; the Python mirror that keeps the reference's signatures
 * (extracting-tree-morphology-from-point-clouds_amd/PointNet2/) binds these symbols with ctypes, and
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory; the library never allocates, frees or
 *     synchronises.  Work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - return value: 0 ok; negative = rejected arguments (PN2_E_*); positive = hipError_t of a failed launch.
 *   - coordinates/features are addressed as base + b*sb + n*sn + c*sc (strides in ELEMENTS), so both the
 *     reference's [B,N,3] tensors and permuted views of its channel-first [B,3,N] tensors are accepted
 *     without a copy.  Outputs are dense row-major.
 *   - indices are int32 on the device side (the Python mirror widens to torch.long where the reference
 *     API returns it).
 *   - all arithmetic is fp32 with the operation order of the reference's CPU path (see oracle/pn2_oracle.c).
 */
#ifndef PN2_HIP_H
#define PN2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN2_E_BADARG (-1)   /* null pointer / non-positive size / unsupported size */
#define PN2_E_WORKSPACE (-2) /* workspace too small */

#define PN2_ABI_VERSION 6

/* Bits of the caller-owned sticky STATUS word (a device int32 the caller zeroes once and reads at a synchronisation
 * point it has anyway, e.g. the loss read-back; the Python mirror: ops.check_status()).  A kernel ORs a bit in when it
 * gave up instead of producing results -- the rows it did not produce hold -1 indices / NaN coordinates. */
#define PN2_STATUS_FPS_HANDOFF 1 /* farthest_point_sample: a workgroup of a cloud's group never delivered its candidate */
#define PN2_STATUS_FPS_ARRIVAL 2 /* farthest_point_sample: the launch's workgroups were not co-resident (busy GPU) */
#define PN2_STATUS_BAD_INDEX 4   /* a gather / scatter kernel was handed an index outside [0, N) and skipped it */
#define PN2_STATUS_COOP_BARRIER 8 /* a cooperative MLP chain launch: a workgroup never reached a layer barrier (busy GPU) */

/* ABI version of the loaded library (compare with PN2_ABI_VERSION). */
int pn2_version(void);

/* Static description of the code object, e.g. "gfx950". */
const char *pn2_arch(void);

/* ---------------------------------------------------------------------------------------------------
 * square_distance                  replaces Modules/PointNet2/pointnet2_utils.py:21-42
 *   out [B,N,M] = ((-2 * src.dst) + |src|^2) + |dst|^2, the reference's expansion (not the direct distance).
 *   API completeness only: ball query and three_nn evaluate the same expression without materialising it.
 */
int pn2_square_distance_f32(const float *src, int64_t ab, int64_t an, int64_t ac, const float *dst, int64_t bb,
                            int64_t bn, int64_t bc, int B, int N, int M, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * farthest_point_sample            replaces Modules/PointNet2/pointnet2_utils.py:66-89
 *   start    [B] int64   first centroid per cloud (the reference's torch.randint draw, line 79)
 *   out_idx  [B,npoint] int32
 *   out_xyz  [B,npoint,3] f32 or NULL -- the gathered centroids (index_points(xyz, idx), line 154)
 *   workspace: pn2_fps_workspace_bytes(B,N,npoint) bytes, contents irrelevant on entry.
 *   status   device int32 or NULL (PN2_STATUS_*).  Clouds of more than 8192 points are sampled by several
 *            workgroups that hand candidates to each other through memory; every wait is bounded (PN2_FPS_SPIN_LIMIT
 *            polls, default 2^22).  When a wait expires the whole launch stops within microseconds, the rows not yet
 *            produced keep the -1 (out_idx) / NaN (out_xyz) fill written ahead of the kernel, and the status word
 *            gets PN2_STATUS_FPS_HANDOFF / _ARRIVAL.  Environment switches (read per call; all paths produce
 *            identical indices): PN2_FPS_NO_XCD (consecutive-block groups, write-through hand-off), PN2_FPS_NO_MULTI
 *            (one sample per exchange), PN2_FPS_FORCE_FALLBACK (the XCD-local kernels with their placement-independent
 *            grouping).
 *   [r2] Clouds of 8 k - 262 k points with >= 200 samples are first put into CELL ORDER (a counting sort of the points by
 *            the 16^3 cells of the cloud's bounding box along the Morton curve, kept in the workspace) so that a wavefront
 *            holds neighbours; pn2_fps_order_offset() tells where that permutation (int32 [B][N]) lies -- (size_t)-1 when
 *            the problem does not take this path -- for callers that want to schedule other kernels by locality
 *            (pn2_three_nn_f32's `order`).  PN2_FPS_NO_SORT keeps the index-order kernels.
 */
size_t pn2_fps_workspace_bytes(int B, int N, int npoint);
size_t pn2_fps_order_offset(int B, int N, int npoint);
size_t pn2_fps_rounds_offset(int B, int N, int npoint);    /* uint32: exchanges the ordered kernel took for cloud 0 (profiling) */
size_t pn2_fps_box_offset(int B, int N, int npoint);       /* uint32 [B][8]: the clouds' boxes (order-preserving encodings) */
size_t pn2_fps_cellstart_offset(int B, int N, int npoint); /* int32 [B][4097]: first sorted position of every cell, then N */
size_t pn2_fps_sorted_xyz_offset(int B, int N, int npoint); /* float [B][3][N]: x, y, z planes in cell order */
int pn2_fps_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
                const int64_t *start, int32_t *out_idx, float *out_xyz, void *workspace,
                size_t workspace_bytes, int32_t *status, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * query_ball_point                 replaces Modules/PointNet2/pointnet2_utils.py:92-136
 *   r2       float32(double(radius)**2); a point is inside unless d > r2
 *   out_idx  [B,S,Keff] int32, Keff = min(nsample, N): first Keff inside indices ascending, short rows
 *            padded with their first entry, empty balls filled with argmin of the row.
 *   workspace: pn2_ball_query_workspace_bytes(B,N,S,nsample) bytes.
 */
size_t pn2_ball_query_workspace_bytes(int B, int N, int S, int nsample);
int pn2_ball_query_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz,
                       int64_t qb, int64_t qn, int64_t qc, int B, int N, int S, float r2, int nsample,
                       int32_t *out_idx, void *workspace, size_t workspace_bytes, void *stream);
/* The same rows from the cell structure an ordered pn2_fps_f32 call on the SAME cloud left in its workspace (box,
 * cellstart, order, sorted_xyz: see pn2_fps_*_offset).  One wavefront per query.  Unless the ball promises to be full
 * (estimated hits >= 8 nsample among more than 8192 candidates) its reachable cells (rounding of the reference's expanded
 * distance included) are searched: every candidate is tested with the exact expression, the nsample smallest hit indices
 * are kept (ascending).  Full balls are walked in index order until nsample hits.  What this buys: the index-order walk of a ball with FEWER than nsample hits is a scan of the whole
 * cloud (68 us at 262144 points), and those queries set the kernel's time. */
int pn2_ball_query_cells_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz, int64_t qb,
                             int64_t qn, int64_t qc, int B, int N, int S, float r2, int nsample, const uint32_t *box,
                             const int32_t *cellstart, const int32_t *order, const float *sorted_xyz, int32_t *out_idx,
                             void *stream);

/* ---------------------------------------------------------------------------------------------------
 * group_points (+ centring + concat)   replaces index_points x2 and lines 156-161 of sample_and_group
 *   out [B,S,K,3+D]: channels [xyz - new_xyz, feats] or, with xyz_last != 0, [feats, xyz - new_xyz]
 *   (the MSG order, blocks.py:143-146).  feats may be NULL with D = 0.
 *   status (device int32 or NULL): an index outside [0,N) is not followed (row 0 is read instead) and sets
 *   PN2_STATUS_BAD_INDEX -- the sync-free form of the reference's host-side range assert (pointnet2_utils.py:54).
 *   The same holds for pn2_gather_f32 and pn2_three_interpolate_f32; the *_grad entry points skip such entries.
 */
int pn2_group_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz,
                  const float *feats, int64_t fb, int64_t fn, int64_t fc, const int32_t *idx, int B, int N,
                  int S, int K, int D, int xyz_last, float *out, int32_t *status, void *stream);

/* gradient w.r.t. feats: dfeats [B,N,D] (dense, zeroed by the call) += dout[..., feature channels] */
int pn2_group_grad_f32(const float *dout, const int32_t *idx, int B, int N, int S, int K, int D,
                       int xyz_last, float *dfeats, void *stream);

/* plain index_points: out[b][s][:] = points[b][idx[b][s]][:]   (pointnet2_utils.py:45-63) */
int pn2_gather_f32(const float *points, int64_t pb, int64_t pn, int64_t pc, const int32_t *idx, int B, int N,
                   int S, int C, float *out, int32_t *status, void *stream);
int pn2_gather_grad_f32(const float *dout, const int32_t *idx, int B, int N, int S, int C, float *dpoints,
                        void *stream);

/* ---------------------------------------------------------------------------------------------------
 * three_nn                          replaces Modules/PointNet2/blocks.py:194-203
 *   for every xyz1 point the 3 smallest square_distance(xyz1, xyz2) entries, ties -> lower index first;
 *   out_idx [B,N,3] int32, out_w [B,N,3] normalised inverse-distance weights, out_dist [B,N,3] or NULL.
 *   Requires S >= 3.
 *   order   NULL, or int32 [B][N]: a permutation of every cloud's point indices (thread t of cloud b works on point
 *           order[b][t]).  Results do not depend on it; a spatial order (pn2_fps_order_offset) lets a wavefront's 64 points
 *           share their nearest samples, which is what the scan's early-out needs.
 */
int pn2_three_nn_f32(const float *xyz1, int64_t ab, int64_t an, int64_t ac, const float *xyz2, int64_t bb,
                     int64_t bn, int64_t bc, int B, int N, int S, int32_t *out_idx, float *out_w,
                     float *out_dist, const int32_t *order, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * masked point-wise loss            replaces the arithmetic of Modules/Loss.py:6-36 (point_wise_loss) under the masks
 *                                   of Modules/PointNet2/PointNet2.py:180-207 (get_loss)
 *   sem [R,2], off [R,3] predictions of the padded rows; pad / off_mask [R] (1 byte each); cum_pad / cum_off [R]
 *   inclusive int64 prefix sums of the masks (cum - 1 = the row's position in the compacted label arrays sem_labels
 *   [n_sem] int64, off_labels [n_off,3]).  out2 = {sum_pad CE / max(n_valid,1), sum_offmask sqrt(max(|d|^2,1e-8)) /
 *   max(n_off_rows,1)}.  Backward: grad2 = d/d out2 -> dsem [R,2], doff [R,3] (zero on masked-out rows).
 */
/* The masks' prefix sums get_loss needs (PointNet2.py:188-196), one launch: cum_pad[r] = inclusive count of pad[0..r],
 * off_mask[r] = pad[r] && masks_off[clamp(cum_pad[r] - 1, 0, n_mask - 1)], cum_off[r] = inclusive count of off_mask.
 * pad [R], masks_off [n_mask] are torch.bool storage (one byte per element, non-zero = true). */
int pn2_mask_ranks(const unsigned char *pad, const unsigned char *masks_off, long long R, long long n_mask,
                   int64_t *cum_pad, unsigned char *off_mask, int64_t *cum_off, void *stream);
size_t pn2_point_loss_workspace_bytes(int R);
int pn2_point_loss_fwd_f32(const float *sem, const float *off, const unsigned char *pad, const unsigned char *off_mask,
                           const int64_t *cum_pad, const int64_t *cum_off, const int64_t *sem_labels, int64_t n_sem,
                           const float *off_labels, int64_t n_off, int R, float *out2, void *workspace,
                           size_t workspace_bytes, void *stream);
int pn2_point_loss_bwd_f32(const float *sem, const float *off, const unsigned char *pad, const unsigned char *off_mask,
                           const int64_t *cum_pad, const int64_t *cum_off, const int64_t *sem_labels, int64_t n_sem,
                           const float *off_labels, int64_t n_off, int R, const float *grad2, float *dsem, float *doff,
                           void *stream);

/* The same with get_loss's two multipliers and its sum folded in (PointNet2.py:198-207): out2 = (w0 * semantic, w1 * offset),
 * *total = their sum; the backward takes the gradient of the TOTAL (one float on the device). */
int pn2_point_loss_weighted_fwd_f32(const float *sem, const float *off, const unsigned char *pad, const unsigned char *off_mask,
                                    const int64_t *cum_pad, const int64_t *cum_off, const int64_t *sem_labels, int64_t n_sem,
                                    const float *off_labels, int64_t n_off, int R, const float *weights2, float *out2,
                                    float *total, void *workspace, size_t workspace_bytes, void *stream);
int pn2_point_loss_weighted_bwd_f32(const float *sem, const float *off, const unsigned char *pad, const unsigned char *off_mask,
                                    const int64_t *cum_pad, const int64_t *cum_off, const int64_t *sem_labels, int64_t n_sem,
                                    const float *off_labels, int64_t n_off, int R, const float *grad_total,
                                    const float *weights2, float *dsem, float *doff, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * kNN feature helpers               replaces the neighbourhood work of Modules/Features.py:111-175
 *   (compute_normals_ckdtree :111-133, compute_curvature_ckdtree :136-158, compute_density_ckdtree :161-173,
 *    as driven by add_features :178-229).  float64 like the reference.
 *   pn2_knn_radius_f64: points [N][3] contiguous; nn_idx [N][k] int32 = the k nearest points of every point,
 *     ascending by (squared distance, index) -- the point itself first, like cKDTree.query(points, k);
 *     nn_d2 [N][k] squared distances or NULL; radius_count [N] = number of points with d^2 <= r2 (the point
 *     itself included, like len(tree.query_ball_point(p, r))) or NULL.  k <= 16.
 *   pn2_cov_eig_f64: for every point the np.cov (unbiased) covariance of the offsets to its first k neighbours
 *     (rows of nn_idx, k_stride entries apart), evals [N][3] ascending, evecs [N][3][3]: row r = unit eigenvector of
 *     evals[r], sign fixed so that its largest component is positive.
 */
int pn2_knn_radius_f64(const double *points, int N, int k, double r2, int32_t *nn_idx, double *nn_d2,
                       int32_t *radius_count, void *stream);
int pn2_cov_eig_f64(const double *points, int N, const int32_t *nn_idx, int k_stride, int k, double *evals,
                    double *evecs, void *stream);
/* Same contract and bit-identical results through a hashed cell grid (csrc/knn_grid.hip): the cloud is binned into
 * cells (adaptive edge), every query's 27 surrounding cells are searched by one wavefront and certified against the
 * cell edge; queries the grid cannot settle are redone by a full scan.  radius_count uses a second grid whose edge is
 * the radius (r2 < 0 or radius_count == NULL: no count).  workspace: pn2_knn_grid_workspace_bytes(N) bytes. */
size_t pn2_knn_grid_workspace_bytes(int N);
int pn2_knn_radius_grid_f64(const double *points, int N, int k, double r2, int32_t *nn_idx, double *nn_d2,
                            int32_t *radius_count, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * three_interpolate                 replaces Modules/PointNet2/blocks.py:204
 *   points2 [B,S,D] (strided), out rows of out_stride floats; the D interpolated channels are written at
 *   column out_offset (so the skip-connection concat of blocks.py:208 needs no extra copy).
 */
int pn2_three_interpolate_f32(const float *points2, int64_t pb, int64_t pn, int64_t pc, const int32_t *idx,
                              const float *w, int B, int N, int S, int D, float *out, int64_t out_stride,
                              int64_t out_offset, int32_t *status, void *stream);
/* The same with the skip connection's rows points1 [B,N,D1] (strided) copied into columns [0, D1) of out by the same launch:
 * cat([points1, interpolated], -1) of blocks.py:208 in one pass.  out_offset >= D1. */
int pn2_three_interpolate_concat_f32(const float *points1, int64_t kb, int64_t kn, int64_t kc, int D1,
                                     const float *points2, int64_t pb, int64_t pn, int64_t pc, const int32_t *idx,
                                     const float *w, int B, int N, int S, int D, float *out, int64_t out_stride,
                                     int64_t out_offset, int32_t *status, void *stream);
/* dpoints2 [B,S,D] (dense, zeroed by the call) += w * dout[:, out_offset : out_offset+D]
 * workspace: pn2_three_interpolate_grad_workspace_bytes(B,N,S,D) bytes (large calls bucket the contributions by
 * destination before summing them). */
size_t pn2_three_interpolate_grad_workspace_bytes(int B, int N, int S, int D);
int pn2_three_interpolate_grad_f32(const float *dout, int64_t out_stride, int64_t out_offset,
                                   const int32_t *idx, const float *w, int B, int N, int S, int D,
                                   float *dpoints2, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Ragged clouds (whole-tree execution of the reference's streaming mode, Modules/PointNet2/PointNet2.py:210-327).
 *   All mini-batches of a tree are processed by ONE set of launches.  Mini-batch j holds B_j rasters zero-padded to N_j
 *   points (RasterizedTreeSet.py:407-429), so the level-0 clouds have different lengths: cloud b has
 *   n_b = coff[b+1] - coff[b] points (coff: device int32 [C+1], coff[0] = 0), and the level-0 tensors are the reference's
 *   per-mini-batch channel-first tensors [B_j, CH, N_j] laid end to end in one flat buffer: element (b, i, ch) lives at
 *   CH * coff[b] + ch * n_b + i.  Sampled levels are regular [C, S, *].  Per-point results (three_nn indices / weights,
 *   interpolated rows) are PACKED rows, row = coff[b] + i, rows = coff[C].  n_max = the longest cloud.  Results are
 *   bit-identical to calling the regular entry points mini-batch by mini-batch.
 *   pn2_fps_ragged_f32: every cloud <= 16384 points (one workgroup per cloud); start [C]; out_idx [C,npoint] cloud-local.
 *   pn2_ball_query_ragged_f32: every cloud must hold at least nsample points (Keff = nsample); out_idx [C,S,nsample].
 *   pn2_group_ragged_f32: xyz_cf (3 planes) and feats_cf (D planes) flat; out [C,S,K,3+D] as pn2_group_f32.
 *   pn2_three_nn_ragged_f32 / pn2_three_interpolate_ragged_f32 / _grad: FP level 0 (dense side ragged, S >= 3).
 */
size_t pn2_fps_ragged_workspace_bytes(int C, int n_max, int npoint);
int pn2_fps_ragged_f32(const float *xyz_cf, const int32_t *coff, int C, int n_max, int npoint, const int64_t *start,
                       int32_t *out_idx, float *out_xyz, void *workspace, size_t workspace_bytes, void *stream);
int pn2_ball_query_ragged_f32(const float *xyz_cf, const int32_t *coff, const float *new_xyz, int C, int n_max, int S,
                              float r2, int nsample, int32_t *out_idx, void *stream);
int pn2_group_ragged_f32(const float *xyz_cf, const float *feats_cf, int D, const int32_t *coff, const float *new_xyz,
                         const int32_t *idx, int C, int S, int K, int xyz_last, float *out, int32_t *status,
                         void *stream);
int pn2_three_nn_ragged_f32(const float *xyz1_cf, const int32_t *coff, const float *xyz2, int C, int n_max, int S,
                            int32_t *out_idx, float *out_w, void *stream);
int pn2_three_interpolate_ragged_f32(const float *points2, const int32_t *idx, const float *w, const int32_t *coff, int C,
                                     int n_max, long long rows, int S, int D, float *out, int64_t out_stride,
                                     int64_t out_offset, int32_t *status, void *stream);
size_t pn2_three_interpolate_grad_ragged_workspace_bytes(int C, long long rows, int S);
int pn2_three_interpolate_grad_ragged_f32(const float *dout, int64_t out_stride, int64_t out_offset, const int32_t *idx,
                                          const float *w, const int32_t *coff, int C, int n_max, long long rows, int S,
                                          int D, float *dpoints2, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Pointwise MLP chains: (1x1 conv -> BatchNorm -> ReLU) x n [-> max over groups of pool_k rows]
 *   replaces the Conv2d/BatchNorm2d/ReLU/max stack of Modules/PointNet2/blocks.py:93-98 (set abstraction),
 *   the Conv1d/BatchNorm1d/ReLU stack of :213-215 (feature propagation) and ConvHead :7-35, fwd and bwd.
 *
 * Activations are channels-last rows: x [rows][cin_0] with row stride ldx.  Layer i computes
 *   y_i = act_{i-1} W_i^T + b_i,   act_i = relu?(BatchNorm(y_i))  (train: batch statistics over all rows,
 *   running stats updated like nn.BatchNorm; eval: running stats),
 * y_i (pre-BatchNorm) is kept in layers[i].y for the backward pass and the normalised activation is never
 * written except for the chain's output:
 *   pool_k <= 1: out [rows][cout_last] = act_last        (a last layer without BatchNorm writes y itself)
 *   pool_k  > 1: out [rows/pool_k][cout_last] = max over each group of pool_k consecutive rows, pool_arg
 *                [rows/pool_k][cout_last] int32 = first row offset attaining it (torch.max's choice).
 * stats: [8][cout] floats per BatchNorm layer ([nseg][8][cout] with row segments, see below) (mean, biased var, invstd, gamma*invstd, beta, and two rows
 * written by the backward pass); running_mean/var may be NULL (no tracking).
 *
 * Backward: dout has the shape of out; gradients are ACCUMULATED (+=) into dweight/dbias/dgamma/dbeta where
 * non-NULL (dbias of a conv feeding a BatchNorm is analytically zero and is left untouched); dx [rows][cin_0]
 * (row stride lddx) is written when non-NULL, from column dx_first_col on (columns in front of it are left untouched:
 * the 3 centred coordinates that lead a grouped set-abstraction input carry no gradient).  scratch_a/b: two buffers of rows * max(cin_i, cout_last) floats.
 * workspace: pn2_mlp_workspace_bytes(rows, layers, nlayers, nseg) bytes for either direction.
 */
typedef struct pn2_mlp_layer {
    int32_t cin, cout;
    const float *weight;       /* [cout][cin] */
    const float *bias;         /* [cout] or NULL */
    int32_t has_bn, relu;
    const float *gamma, *beta; /* [cout] or NULL (1, 0) */
    float *running_mean, *running_var;
    float eps, momentum;
    float *y;                  /* [rows][cout] pre-BatchNorm output (written by fwd, read by bwd) */
    float *stats;              /* [8][cout] */
    float *dweight, *dbias, *dgamma, *dbeta;
    /* layers[0] only -- a chain LINKED to the one that produced its input: `x` then holds the PRE-activation rows of that
     * chain's last BatchNorm layer (its `y`) and in_stats that layer's coefficient block(s) ([nseg][8][cin], same segments);
     * BatchNorm (+ ReLU when in_relu) is applied while the rows are staged, as between the layers of one chain, and the
     * producing chain need not materialise its output (PN2_CHAIN_LAZY_OUT).  The input gradient `dx` is the gradient with
     * respect to the ACTIVATED rows, i.e. exactly what the producing chain's backward expects as `dout`.  NULL: plain rows. */
    const float *in_stats;
    int32_t in_relu;
    /* Linked chains, backward.  The chain that computes the COMPLETE gradient of the linked rows (the only consumer, or the
     * accumulating second one of a pair) can leave the BatchNorm-backward column sums of the producing layer behind -- its
     * dgrad epilogue has every dX element in registers -- so that the producing chain's backward does not read dout and y
     * once more just to form them.  Consumer: layers[0].in_partial = a buffer of pn2_mlp_link_partial_bytes() (NULL: off;
     * needs dx_first_col == 0).  Producer: layers[nlayers-1].out_partial = that buffer, out_partial_rows / _cpb = the row
     * block size and chunks per block the helper reported. */
    float *in_partial;
    const float *out_partial;
    int32_t out_partial_rows, out_partial_cpb;
} pn2_mlp_layer;

/* Row segments (whole-tree execution).  The reference's streaming mode runs the mini-batches of a tree one after the
 * other (Modules/PointNet2/PointNet2.py:238-306): every mini-batch is a forward pass of its own, with its OWN train-mode
 * BatchNorm statistics and one momentum update of the running statistics.  With `segments` the rows of a chain are the
 * concatenation of nseg such mini-batches: statistics, normalisation and the BatchNorm backward are per segment, the
 * running statistics are updated segment after segment, weight gradients are summed over all rows -- the same numbers as
 * nseg separate calls, in one set of launches.  Per-layer `stats` blocks then hold [nseg][8][cout] floats.
 * row_off is a HOST array of nseg + 1 ascending row offsets (row_off[0] = 0, row_off[nseg] = rows; every segment a
 * multiple of pool_k rows); it is read during the call only.  NULL or nseg <= 1: one segment.  Eval mode ignores it. */
#define PN2_MAX_SEGMENTS 128
typedef struct pn2_segments {
    int32_t nseg;
    const int32_t *row_off;
} pn2_segments;

/* precision of the large (128-row-tile) contractions of a chain call:
 *   PN2_PRECISION_F32   v_mfma_f32_32x32x2_f32: exact fp32 products and sums -- the parity mode, the reference's width
 *                       (PointNet2.py:146 disables autocast around the backbone);
 *   PN2_PRECISION_BF16  operands rounded to bfloat16 (nearest even) in front of v_mfma_f32_32x32x16_bf16, fp32
 *                       accumulation; activations, statistics and gradients stay fp32 in memory.  Throughput mode with
 *                       its own, looser tolerance (tests/test_bf16_mode.py); small layers still run fp32. */
#define PN2_PRECISION_F32 0
#define PN2_PRECISION_BF16 1
/* pn2_mlp_chain_bwd_f32 only, OR-ed into `precision`: ADD the input gradient to what `dx` holds instead of storing it
 * (columns dx_first_col .. cin-1; the first layer must not be a narrow one).  Lets two chains that read the same input --
 * the semantic and the offset head on the backbone features, PointNet2.py:86-87 -- leave one gradient tensor behind
 * without a separate 3 x rows x cin x 4 byte add. */
#define PN2_CHAIN_ACCUMULATE_DX 0x100
/* Deferred weight-gradient reductions.  The wgrad GEMMs of a chain backward leave split-K slabs in the call's workspace; one
 * launch per chain to sum them is 5-8 us of mostly latency, 12 of them per backward pass of the depth-4 model.  With a
 * non-NULL `deferred` (a HOST struct the caller owns) pn2_mlp_chain_bwd_f32 does not reduce: it APPENDS the reductions it
 * owes (slab, nsplit, element count, target) to the list and returns; the caller keeps the call's workspace alive, and
 * pn2_mlp_reduce_wgrad(tasks, n, stream) later performs any number of them in as few launches as possible (32 per launch;
 * two reductions into the same target never share one) -- before anything reads the weight gradients (pn2_amd/mlp.py: an
 * autograd-engine callback at the end of the backward pass, one list per backward pass).  The library keeps NO record of
 * pending reductions: a pass that dies simply drops its list.  A list with fewer than nlayers free entries makes the call
 * reduce in place instead. */
#define PN2_WGRAD_TASKS_MAX 64
typedef struct pn2_wgrad_task {
    const float *slab; /* [nsplit][mn] */
    float *out;        /* [mn], accumulated into */
    int64_t mn;
    int32_t nsplit, reserved;
} pn2_wgrad_task;
typedef struct pn2_wgrad_tasks {
    int32_t n, reserved;
    pn2_wgrad_task t[PN2_WGRAD_TASKS_MAX];
} pn2_wgrad_tasks;
/* pn2_mlp_chain_fwd_f32 only, OR-ed into `precision`: the last layer ends in a BatchNorm and its activation is NOT written
 * (`out` may be NULL): the consumer is a linked chain (pn2_mlp_layer.in_stats) that reads `y` and `stats` of that layer.
 * Saves one read and one write of rows x cout floats.  Not with pool_k > 1. */
#define PN2_CHAIN_LAZY_OUT 0x400
/* pn2_mlp_chain_bwd_f32 only, OR-ed into `precision`: columns [0, dx_first_col) of dx are ZEROED by the call instead of
 * left untouched (inside the max-pool scatter launch when there is one -- no launch of its own).  Not with
 * PN2_CHAIN_ACCUMULATE_DX. */
#define PN2_CHAIN_ZERO_LEAD 0x800

/* Cooperative chain launches (the deep levels).  A chain of a few hundred to a few thousand rows is ~25 launches of 5-25 us
 * each way -- GEMM, BatchNorm finalize, GEMM, ... -- whose time is launch ramp and drain, not work.  With a non-NULL `coop`
 * such a chain (train mode, one segment, every layer a BatchNorm layer, rows <= PN2_COOP_MAX_ROWS -- the default is where it was
 * measured to win; the environment variable PN2_COOP_MAX_ROWS overrides it) runs as ONE persistent
 * launch per direction: workgroups take the layers' 32 x 32 tiles from work queues, layers are separated by the queues' done
 * counts instead of a kernel boundary, tiles that cross workgroups are stored write-through and read past the (per-XCD,
 * mutually incoherent) L2, and every workgroup derives the BatchNorm coefficients it needs from the tile epilogues'
 * partials itself (no finalize launches).  Same arithmetic per element as the launch-per-layer path (same tile code).
 *   sync    >= 64 device uint32 words, ZEROED ONCE by the caller and then left alone; one buffer per stream on which chains
 *           run (launches on one stream are ordered; the kernels leave the words zeroed again when they finish)
 *   status  the sticky PN2_STATUS_* word (PN2_STATUS_COOP_BARRIER: a barrier timed out after spin_limit polls -- the launch
 *           drains, its results are garbage; the caller re-zeroes `sync` before the next launch)
 * Nothing requires the launch's workgroups to be co-resident (work queues; a GPU shared with another process is fine).
 * max_workgroups: cap of the launch's grid (0 = default 128: more pollers make every barrier slower).
 * NULL: always the launch-per-layer path. */
#define PN2_COOP_MAX_ROWS 256
typedef struct pn2_coop {
    uint32_t *sync;
    int32_t *status;
    uint32_t spin_limit;     /* 0 = default (1 << 22 polls, seconds) */
    int32_t max_workgroups;  /* 0 = default (128) */
} pn2_coop;

/* bf16 mode with bfloat16 STORAGE (OR-ed into `precision` next to PN2_PRECISION_BF16).  The bf16 mode of the large chains is bound
 * by HBM, not by the matrix cores, as long as every row tensor is fp32 in memory; with PN2_CHAIN_STORE_BF16 the chain's
 * pre-BatchNorm rows (every layers[i].y) and, in the backward call, the two scratch buffers hold __bf16 elements (same shapes,
 * half the bytes); statistics, coefficient blocks, accumulators, weights and weight gradients stay fp32 (the statistics are
 * taken from the fp32 accumulators before the rows are rounded).  Only chains for which pn2_mlp_chain_bf16_storage() says 1
 * (every contraction a 128-row-tile kernel, no pooling).  The neighbours of a LINKED chain follow:
 *   PN2_CHAIN_X_BF16     x (a producing chain's rows, layers[0].in_stats) is __bf16
 *   PN2_CHAIN_DOUT_BF16  backward: dout is __bf16 (the consumer chain wrote it with PN2_CHAIN_DX_BF16)
 *   PN2_CHAIN_DX_BF16    backward: dx is written (or, with PN2_CHAIN_ACCUMULATE_DX, read and written) as __bf16 */
#define PN2_CHAIN_X_BF16 0x1000
#define PN2_CHAIN_STORE_BF16 0x2000
#define PN2_CHAIN_DOUT_BF16 0x4000
#define PN2_CHAIN_DX_BF16 0x8000
int pn2_mlp_chain_bf16_storage(int rows, const pn2_mlp_layer *layers, int nlayers, int pool_k);

size_t pn2_mlp_workspace_bytes(int rows, const pn2_mlp_layer *layers, int nlayers, int nseg);
int pn2_mlp_chain_fwd_f32(const float *x, int64_t ldx, int rows, const pn2_mlp_layer *layers, int nlayers,
                          int training, int pool_k, float *out, int32_t *pool_arg, const pn2_segments *segments,
                          int precision, const pn2_coop *coop, void *workspace, size_t workspace_bytes, void *stream);
int pn2_mlp_chain_bwd_f32(const float *x, int64_t ldx, int rows, const pn2_mlp_layer *layers, int nlayers,
                          int pool_k, const float *dout, const int32_t *pool_arg, float *dx, int64_t lddx,
                          int dx_first_col, float *scratch_a, float *scratch_b, const pn2_segments *segments,
                          int precision, pn2_wgrad_tasks *deferred, const pn2_coop *coop, void *workspace,
                          size_t workspace_bytes, void *stream);
size_t pn2_mlp_link_partial_bytes(int rows, int cin, int nseg, int32_t *block_rows, int32_t *chunks_per_block);
int pn2_mlp_reduce_wgrad(const pn2_wgrad_task *tasks, int n, void *stream);

/* The first-layer dgrad of TWO chains that read the same rows (the two prediction heads on the backbone features,
 * PointNet2.py:128-129) as ONE contraction: dx = dY_a W_a + dY_b W_b over K = 2 cout -- one launch, one prologue and epilogue,
 * no read-modify-write of dx in between (pn2_mlp_chain_bwd_f32 with dx = NULL has run for both chains: their layers[0]
 * hold y, stats with the backward coefficients and weight; dza / dzb = the gradient rows each call left in its scratch_a,
 * i.e. the gradient with respect to relu(bn(y)) of that layer).  la / lb: cin, cout (a multiple of 16), has_bn, relu equal;
 * rows large enough for the 128-row tiles.  Linked heads: la->in_stats / in_relu / in_partial as for the chains (x = the
 * producer's rows), and the producer's BatchNorm-backward sums are left behind by the epilogue.  `precision` and the
 * PN2_CHAIN_{X,STORE,DX}_BF16 flags as for pn2_mlp_chain_bwd_f32. */
int pn2_mlp_pair_dgrad_f32(int rows, const pn2_mlp_layer *la, const float *dza, const pn2_mlp_layer *lb, const float *dzb,
                           const float *x, int64_t ldx, float *dx, int64_t lddx, const pn2_segments *segments, int precision,
                           void *stream);

/* Feature propagation with the first convolution HOISTED in front of the interpolation   (blocks.py:194-215)
 *
 * Where a feature-propagation level has no skip connection (fp1: points1 = None, PointNet2.py:156) its first layer is
 * conv(interp(P)) with P the [B,S,D2] sampled rows.  Interpolation is linear and its three weights sum to one, so
 *     conv(interp(P)) = interp(conv(P)):
 * the contraction runs over the B*S sampled rows (a chain call of one bare conv layer, bias included) instead of the B*N
 * dense rows -- at the headline shape 1024 rows instead of 262 144 -- and what is left at full resolution is the
 * interpolation itself, which these two entry points fuse with the layer's train-mode BatchNorm:
 *
 * pn2_interp_bn_fwd_f32   y[r][:] = (q[i0]*w0 + q[i1]*w1) + q[i2]*w2 (the reference's operation order), the per-chunk
 *   (mean, M2) statistics partials from the same registers, then the finalize of the chain kernels: layer->stats gets the
 *   coefficient block(s), the running statistics advance.  q: [B][S][C] rows, idx / w: [rows][3] (three_nn), layer: cout = C
 *   (64, 128 or 256), has_bn = 1, gamma .. momentum, y [rows][C], stats [nseg][8][C].  The result is exactly a chain's
 *   PN2_CHAIN_LAZY_OUT output: the next chain links to (y, stats) through layers[0].in_stats.
 * pn2_interp_bn_bwd_f32   dout = gradient with respect to relu(bn(y)) (what the linked consumer's dgrad leaves; its
 *   BatchNorm-backward sums through layer->out_partial as between linked chains, NULL: reduced here).  Finalizes dgamma /
 *   dbeta (accumulated) and the backward coefficients, then scatters dZ(dout, y) -- rebuilt while the rows are read, never
 *   stored -- to dq [B][S][C] with the bucketed reduction of pn2_three_interpolate_grad_f32.
 * Dense side: B clouds of N rows (coff = row_cloud = NULL, rows = B*N), or ragged clouds ("Ragged clouds" below: coff
 * device [B + 1] row offsets, N = the longest cloud, rows = coff[B], row_cloud device [rows] = the cloud of every packed row).
 * Rounding: the same products summed in another order (~1e-7 relative); `segments` as for the chains, every segment made
 * of whole clouds.  rows_bf16 (bf16 mode with bfloat16 storage, PN2_CHAIN_STORE_BF16): y is written -- and y and dout are
 * read -- as __bf16 rows, so that the linked chain keeps its bfloat16 rows; q, dq, statistics and sums stay fp32.
 * training = 0 (eval mode, both *_bn_fwd entry points): the rows only; layer->stats gets ONE coefficient block from the running
 * statistics (segments ignored, as in the chains); fp32 rows.
 * workspace: pn2_interp_bn_workspace_bytes() for either direction. */
size_t pn2_interp_bn_workspace_bytes(int B, long long rows, int S, int C, int nseg);
int pn2_interp_bn_fwd_f32(const float *q, const int32_t *idx, const float *w, const int32_t *coff, const int32_t *row_cloud,
                          int B, int N, int S, long long rows, const pn2_mlp_layer *layer, const pn2_segments *segments,
                          int training, int rows_bf16, int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
int pn2_interp_bn_bwd_f32(const float *dout, const int32_t *idx, const float *w, const int32_t *coff, int B, int N, int S,
                          long long rows, const pn2_mlp_layer *layer, float *dq, const pn2_segments *segments,
                          int rows_bf16, void *workspace, size_t workspace_bytes, void *stream);

/* Set abstraction with the first convolution HOISTED in front of the grouping   (blocks.py:74-98, pointnet2_utils.py:156-161)
 *
 * The first layer of a set-abstraction level is conv([xyz_j - c_s, feats_j]) over the B*S*K grouped rows.  The conv is
 * linear and a grouped row is a gather, so its feature share is computed ONCE PER SOURCE POINT, gf = W_f feats + bias over
 * the B*N source rows (a chain call of one bare conv layer) -- N instead of S*K rows per cloud, e.g. 100 instead of 1600 at
 * the second level of a raster -- and the grouped rows of the layer are gf[idx] + W_x (xyz[idx] - c), the coordinate share
 * being three multiply-adds per channel on the same centred differences the reference forms.  The [rows][3 + D] grouped
 * tensor and its odd-width contraction (67 / 131 / 259 input channels) no longer exist.
 *
 * pn2_group_bn_fwd_f32   y[(b,s,k)][:] = gf[b][idx[b][s][k]][:] + W_x (xyz[b][idx] - new_xyz[b][s]), the (mean, M2) statistics
 *   partials from the same registers, then the chain kernels' finalize: exactly a chain's PN2_CHAIN_LAZY_OUT output, which
 *   the chain of the remaining layers (with its max over K) links to.  gf [B][N][C]; xyz strided as everywhere; wx = the
 *   coordinate columns of the layer's weight, element (c, d) at wx[c*ldw + d]; layer: cout = C (32, 64, 128 or 256), has_bn,
 *   gamma .. momentum, y [B*S*K][C], stats; K <= 64.
 * pn2_group_bn_bwd_f32   dout = gradient with respect to relu(bn(y)); BatchNorm-backward sums through layer->out_partial or
 *   reduced here; dgamma / dbeta accumulated; dZ(dout, y) is rebuilt row by row and scattered to dgf [B][N][C] (overwritten),
 *   the coordinate weights' gradient sum dZ (x) (xyz - c) is written to dwx[c*lddw + d].  No gradient for the coordinates.
 * `segments`: whole clouds per segment.  workspace: pn2_group_bn_workspace_bytes() for either direction. */
size_t pn2_group_bn_workspace_bytes(int B, int S, int K, int C, int nseg);
int pn2_group_bn_fwd_f32(const float *gf, const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz,
                         const int32_t *idx, const float *wx, int64_t ldw, int B, int N, int S, int K,
                         const pn2_mlp_layer *layer, const pn2_segments *segments, int training, int32_t *status, void *workspace,
                         size_t workspace_bytes, void *stream);
int pn2_group_bn_bwd_f32(const float *dout, const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz,
                         const int32_t *idx, int B, int N, int S, int K, const pn2_mlp_layer *layer, float *dgf, float *dwx,
                         int64_t lddw, const pn2_segments *segments, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * PointTransformerV3 serialized patch attention   replaces Modules/PointTransformerV3/blocks.py:384-437 and :457-488
 *
 * pn2_ptv3_pad_unpad_i64 -- get_padding_and_inverse (:384-437).  Clouds of n_i points are padded to whole patches of
 *   patch_size (clouds not longer than one patch are left alone); off / offpad: device [B + 1] prefix sums of n_i and of the
 *   padded sizes, cu_off: device [B + 1] prefix counts of patches per cloud (the caller has the cloud sizes on the host --
 *   the reference indexes `offset` on the host as well).  Outputs (device): pad [n_pad] int64 = for every padded position the
 *   serialized position it reads (the tail of a cloud's last patch repeats the positions one patch earlier, :409-420),
 *   unpad [n] int64 = padded position of every serialized position, cu_seqlens [patches + 1] int32.
 *
 * pn2_ptv3_patch_attention_f32 -- the non-flash branch of SerializedAttention.forward (:457-488):
 *   qkv rows [3][heads][head_dim] (row stride ld); order [n_rows] int64 = row of qkv behind every padded position (the
 *   reference's `qkv[order]` gather, fused) or NULL; every patch_size consecutive positions form a patch;
 *   out [n_rows][heads * head_dim] = softmax((q * scale) k^T) v per patch and head -- the K x K scores are never written.
 *   head_dim must be 16 (every stage of the repository's configuration), patch_size <= 1024, n_rows % patch_size == 0.
 *   precision: PN2_PRECISION_F32 (exact fp32 MFMA) or PN2_PRECISION_BF16 (bfloat16 operands, fp32 accumulation and softmax). */
int pn2_ptv3_pad_unpad_i64(const int64_t *off, const int64_t *offpad, const int64_t *cu_off, int B, int patch_size, int64_t n_pad,
                           int64_t *pad, int64_t *unpad, int32_t *cu_seqlens, void *stream);
int pn2_ptv3_patch_attention_f32(const float *qkv, int64_t ld, const int64_t *order, int64_t n_rows, int patch_size, int heads,
                                 int head_dim, float scale, float *out, int precision, void *stream);
/* Training (blocks.py:457-488 under autograd).
 * pn2_ptv3_patch_attention_lse_f32 -- the forward above that also stores lse [n_rows][heads] = log sum_k exp(s_k) of every
 *   (padded position, head): what the backward rebuilds the probabilities from (lse NULL: exactly the call above).
 * pn2_ptv3_patch_attention_bwd_f32 -- out, lse: what that forward returned; dout [n_rows][heads * head_dim] -> dqkv_rows
 *   [n_rows][3 * heads * head_dim] = the gradient w.r.t. the GATHERED rows qkv[order] (every element written).  A row of qkv
 *   read by several padded positions collects them in the caller (index_add over `order`).  Two launches (dq; dk + dv), one
 *   workgroup per (patch, head), exact fp32 MFMA, the K x K matrices rebuilt block by block and never written. */
int pn2_ptv3_patch_attention_lse_f32(const float *qkv, int64_t ld, const int64_t *order, int64_t n_rows, int patch_size, int heads,
                                     int head_dim, float scale, float *out, float *lse, int precision, void *stream);
int pn2_ptv3_patch_attention_bwd_f32(const float *qkv, int64_t ld, const int64_t *order, int64_t n_rows, int patch_size, int heads,
                                     int head_dim, float scale, const float *out, const float *lse, const float *dout,
                                     float *dqkv_rows, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Closest-cylinder projection       replaces Modules/Projection.py:19-114 (closest_cylinder_cuda_batch; duplicated at
 *                                   PreProcessing/LabelGenerationCuda.py:20-110)
 *   points [N] rows of point_stride floats (xyz first); cylinders: start [M,3], axis_unit [M,3], axis_length [M],
 *   radius [M] (Projection.py:126-132 prepares them), ids [M] int32 or NULL (then the cylinder index is returned).
 *   out_id [N] int32 = ID of the cylinder with the smallest distance (first minimum), out_dist [N] or NULL that
 *   distance, out_offset [N,3] = projection point - point; with move_points_to_mantle the non-perpendicular case is
 *   moved to the nearer end of the diameter segment (:90-105).  One launch for the whole cloud (the reference batches
 *   1024 points against [1024, M, 3] temporaries).
 */
int pn2_cylinder_project_f32(const float *points, int64_t point_stride, int N, const float *start, const float *axis_unit,
                             const float *axis_length, const float *radius, const int32_t *ids, int M,
                             int move_points_to_mantle, int32_t *out_id, float *out_dist, float *out_offset, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Raster data path                  replaces the host loops of Modules/Pipeline/ModelPredicting.py:98-163
 *                                   (rasterize_clouds) and Modules/DataLoading/RasterizedTreeSet.py:201-268, 390-459
 *                                   (__getitem__ box masks, collate_fn_streaming zero padding)
 *   bounds: per axis the ascending float32 lower and upper bounds of the boxes, laid out
 *     [lo_x (nx) | hi_x (nx) | lo_y (ny) | hi_y (ny) | lo_z (nz) | hi_z (nz)]; a point is inside box k of an axis iff
 *     lo[k] <= p < hi[k] (the dataset's float32 test).
 *   pn2_raster_ranges_f32: ranges [N][6] int32 = {first_x, last_x, first_y, last_y, first_z, last_z} (empty when
 *     last < first), count [N] = number of boxes the point falls into.
 *   pn2_raster_keys: offset [N] = exclusive prefix sum of count; keys [sum(count)] int64 = box_id * N + point_id with
 *     box_id = (kx * ny + ky) * nz + kz: sorting them yields the rasters in the reference's order, ascending ids inside.
 *   pn2_raster_pack_f32: from the sorted point ids to the network input of a whole tree: cloud_table [C][4] int32 =
 *     {padded rows before this raster, padded length N_j, first entry in sorted_ids, real length}; writes the flat
 *     channel-first zero-padded buffers ("Ragged clouds" above) xyz_cf [3 * rows], feats_cf [F * rows] and the padding
 *     mask [rows] (1 byte each).
 */
int pn2_raster_ranges_f32(const float *points, int64_t point_stride, int N, const float *bounds, int nx, int ny, int nz,
                          int32_t *ranges, int32_t *count, void *stream);
int pn2_raster_keys(const int32_t *ranges, const int64_t *offset, int N, int ny, int nz, int64_t *keys, void *stream);
int pn2_raster_pack_f32(const float *points, int64_t point_stride, const float *feats, int64_t feat_stride, int F,
                        const int64_t *sorted_ids, const int32_t *cloud_table, int C, int n_max, float *xyz_cf,
                        float *feats_cf, uint8_t *masks_pad, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Measurement hook (bench.py): with profiling enabled every kernel launch of the library is bracketed by two
 * HIP events on its launch stream.  pn2_prof_collect synchronises on them, aggregates by (kernel name,
 * algorithmic bytes, flops) of a launch, writes NUL-separated names and per-group totals, clears the records and
 * returns the number of groups.  Not meant to be left on in production (it creates two events per launch).
 */
void pn2_prof_enable(int on);
int pn2_prof_collect(char *names_buf, size_t names_cap, double *total_ms, long long *calls, double *bytes,
                     double *flops, int max_groups);

/* ---------------------------------------------------------------------------------------------------
 * Point-cloud serialization        replaces Modules/PointTransformerV3/serialization (default.py:8-40 encode / decode,
 *   z_order.py:86-125, hilbert.py:91-302), the first stage of Point.serialization (PointTransformerV3/blocks.py:98-150).
 *   grid_coord [N][3] int32 (element strides gs between points, gc between coordinates), batch [N] int64 or NULL,
 *   1 <= depth <= 16 (the reference's own limit, blocks.py:126); only the low `depth` bits of a coordinate count.
 *   out_codes [n_orders][N] int64: one row per requested order, code = batch << 3 depth | curve key.
 *   decode: order PN2_ORDER_Z or PN2_ORDER_HILBERT (like the reference); out_grid [N][3] int64, out_batch [N] or NULL. */
#define PN2_ORDER_Z 0
#define PN2_ORDER_Z_TRANS 1
#define PN2_ORDER_HILBERT 2
#define PN2_ORDER_HILBERT_TRANS 3
int pn2_serialize_encode_i64(const int32_t *grid_coord, int64_t gs, int64_t gc, const int64_t *batch, long long N,
                             int depth, const int32_t *orders, int n_orders, int64_t *out_codes, void *stream);
int pn2_serialize_decode_i64(const int64_t *codes, long long N, int depth, int order, int64_t *out_grid,
                             int64_t *out_batch, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * PointTransformerV3 conditional positional encoding   replaces the spconv.SubMConv3d(channels, channels, kernel_size=3,
 *   bias=True) in front of every Block, Modules/PointTransformerV3/blocks.py:561-568, on the voxels of Point.sparsify
 *   (:153-190: indices = [batch, grid_coord]).  spconv is absent here; the operator built is its definition -- the dense
 *   cross-correlation conv3d(padding = 1) of the voxel grid evaluated at the active voxels, inactive voxels contributing zeros:
 *       out[i] = bias + sum_{d in {-1,0,1}^3} W[d] feat[j(i, d)],   j(i, d) = the active voxel of i's cloud at grid_coord[i] + d.
 *
 * pn2_ptv3_subm_neighbors_i32   nbr [N][27] int32 = j(i, d) or -1, offset index d = (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1) for
 *   grid_coord = (x, y, z).  batch: device int64 [N] cloud ids 0..65535 (NULL: one cloud); grid_coord: device int32 [N][3],
 *   0 <= coordinate <= 65533 (a voxel outside gets no neighbours and PN2_STATUS_BAD_INDEX); duplicate voxels are represented by
 *   their lowest index.  The table is what the Blocks of one stage share (spconv's indice_key).
 *   workspace: pn2_ptv3_subm_workspace_bytes(N) (the hash table; contents irrelevant on entry).
 * pn2_ptv3_subm_conv_f32        feat [N][C_in] (row stride ldf, a multiple of 4), weight [27][C_in][C_out] (offset-major: the
 *   mirror permutes spconv's [C_out][3][3][3][C_in] parameter), bias [C_out] or NULL -> out [N][C_out] (row stride ldo).
 *   C_in a multiple of 16, C_out of 32.  fp32 (v_mfma_f32_32x32x2_f32).
 * kernel_size 3 (the CPE: 27 offsets) or 5 (the stem of Embedding, blocks.py:783-791: 125 offsets, index
 *   (dx + 2) * 25 + (dy + 2) * 5 + (dz + 2); nbr [N][125], weight [125][C_in][C_out]); a C_in that is not a multiple of 16 (the stem's
 *   input features) is zero-padded by the caller.
 * weight_bf16 (NULL: fp32): a __bf16 copy of the weights laid out [27][C_out][C_in] selects the bf16 mode for kernel_size 3 and
 *   C_in >= 64: operands rounded to bfloat16, v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 rows in and out (the throughput
 *   mode with its own tolerance, as PN2_PRECISION_BF16 of the attention); narrower layers and the stem stay fp32. */
size_t pn2_ptv3_subm_workspace_bytes(int N);
int pn2_ptv3_subm_neighbors_i32(const int64_t *batch, const int32_t *grid_coord, int N, int kernel_size, int32_t *nbr,
                                void *workspace, size_t workspace_bytes, int32_t *status, void *stream);
int pn2_ptv3_subm_conv_f32(const float *feat, int64_t ldf, const int32_t *nbr, int kernel_size, const float *weight,
                           const void *weight_bf16, const float *bias, int N, int Cin, int Cout, float *out, int64_t ldo,
                           void *stream);
/* Training (spconv's SubMConv3d backward).
 * pn2_ptv3_subm_wgrad_f32   dweight [k^3][C_in][C_out] (offset-major, like `weight` above; every element written)
 *       = sum_i feat[j(i, d)]^T dout[i]: a gathered split-K contraction on v_mfma_f32_32x32x2_f32 whose row ranges leave slabs in
 *   the workspace (pn2_ptv3_subm_wgrad_workspace_bytes) that a second launch sums in fixed order (deterministic).  C_in, C_out
 *   multiples of 4.
 * The input gradient is the forward kernel on the mirrored stencil: for distinct voxels j(i, d) = j' <=> j(j', -d) = i, so
 *   dfeat = pn2_ptv3_subm_conv_f32(dout, nbr, W') with W'[d][co][ci] = W[-d][ci][co] (the mirror builds W'); the bias
 *   gradient is a column sum. */
size_t pn2_ptv3_subm_wgrad_workspace_bytes(int N, int kernel_size, int Cin, int Cout);
int pn2_ptv3_subm_wgrad_f32(const float *feat, int64_t ldf, const int32_t *nbr, int kernel_size, const float *dout, int64_t ldo,
                            int N, int Cin, int Cout, float *dweight, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * LayerNorm over the channels of point rows    replaces the nn.LayerNorm `norm_layer` of every PointTransformerV3 Block,
 *   Modules/PointTransformerV3/blocks.py:551-597, on [rows][C] fp32 rows (row strides multiples of 4, 16-byte aligned bases),
 *   C in {32, 64, 128, 256, 512} (pn2_layer_norm_supported): C / 4 (<= 64) lanes own a row, two-pass statistics in registers.
 * pn2_layer_norm_fwd_f32   y = (x - mean) * rstd * gamma + beta (gamma / beta NULL: 1 / 0); mean, rstd [rows] are written when
 *   both are given (what the backward needs), both NULL at inference.
 * pn2_layer_norm_bwd_f32   dx (every element written), dgamma / dbeta [C] (overwritten; NULL: not wanted) from dy, x and the
 *   forward's mean / rstd; the parameter gradients are per-workgroup partials in the workspace summed in fixed order by a second
 *   launch (deterministic).  workspace: pn2_layer_norm_bwd_workspace_bytes(rows, C). */
int pn2_layer_norm_supported(int C);
int pn2_layer_norm_fwd_f32(const float *x, int64_t ldx, const float *gamma, const float *beta, float eps, int64_t rows, int C,
                           float *y, int64_t ldy, float *mean, float *rstd, void *stream);
size_t pn2_layer_norm_bwd_workspace_bytes(int64_t rows, int C);
int pn2_layer_norm_bwd_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *mean, const float *rstd,
                           const float *gamma, int64_t rows, int C, float *dx, int64_t lddx, float *dgamma, float *dbeta,
                           void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_HIP_H */
