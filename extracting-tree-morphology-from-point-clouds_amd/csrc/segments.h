// Row segments and BatchNorm coefficient blocks: the small shared vocabulary of the chain kernels (mlp_tile.h, mlp.hip,
// chain_coop.hip) and of the interpolation backward that rebuilds dZ from (dA, Y) (three_nn.hip).
#pragma once
#include "pn2_common.h"

namespace {

// rows of a layer's coefficient block (each `C` floats): see pn2_mlp_layer.stats in pn2_hip.h
enum { ST_MEAN = 0, ST_VAR = 1, ST_INVSTD = 2, ST_SCALE = 3, ST_BETA = 4, ST_A = 5, ST_B = 6, ST_ROWS = 8 };

// Row SEGMENTS (whole-tree execution): the rows of a chain are the concatenation of nseg mini-batches, each with its
// OWN train-mode BatchNorm statistics (the reference runs them as separate forward passes, PointNet2.py:238-306).
// Every kernel that needs a row's statistics is launched over row BLOCKS that never straddle a segment boundary:
// block i of a launch covers rows [row_off[s] + (i - blk_off[s]) * R, ...) of the segment s with
// blk_off[s] <= i < blk_off[s+1] (R = the launch's block size: GEMM tile, reduction block, ...).  The table travels BY
// VALUE in the kernel arguments (kernarg segment: scalar loads), so there is nothing to build or upload on the device.
// nseg == 1 is the ordinary batch: block i covers rows [i * R, ...).
constexpr int kMaxSegs = PN2_MAX_SEGMENTS;
struct SegTable {
    int nseg;
    int row_off[kMaxSegs + 1];
    int blk_off[kMaxSegs + 1];
};
struct RowBlock {
    int seg, row0, row_end;
};
// wave-uniform: blockIdx -> (segment, first row, end of the segment)
__device__ __forceinline__ RowBlock row_block(const SegTable& st, int blk, int R) {
    int lo = 0, hi = st.nseg;
    while (hi - lo > 1) {  // largest s with blk_off[s] <= blk
        const int mid = (lo + hi) >> 1;
        if (st.blk_off[mid] <= blk) lo = mid; else hi = mid;
    }
    return RowBlock{lo, st.row_off[lo] + (blk - st.blk_off[lo]) * R, st.row_off[lo + 1]};
}
__device__ __forceinline__ int seg_of_row(const SegTable& st, int row) {
    int lo = 0, hi = st.nseg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (st.row_off[mid] <= row) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace
