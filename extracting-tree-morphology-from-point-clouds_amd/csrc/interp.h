// Interpolation backward with the gradient rows rebuilt on the fly: the entry points that hoist a feature-propagation level's
// first convolution in front of the interpolation (mlp.hip: pn2_interp_bn_*) hand the bucketed reduction of three_nn.hip the
// gradient with respect to relu(bn(Y)) plus (Y, coefficient blocks) instead of a materialised dZ.
#pragma once
#include "pn2_common.h"

namespace pn2 {
namespace interp {

struct DySource {
    const float* y;            // [rows][D] pre-BatchNorm rows (row stride D)
    const float* coef;         // [nseg][8][D] coefficient blocks with the backward coefficients a, b filled in
    int relu;
    int nseg;                  // row segments of the packed rows (1: one block serves every row)
    const int32_t* row_off;    // host, nseg + 1 ascending offsets (ignored for nseg <= 1)
    int rows_bf16;             // dout and y are __bf16 rows (bf16 mode with bfloat16 storage)
};

size_t grad_workspace_bytes(int B, long long rows, int S);
// dpoints2[b][idx[r][k]][:] += w[r][k] * g[r][:] with g = dout rows (dy == nullptr) or dZ(dout, y) (dy given, D % 64 == 0,
// out_stride == D).  coff: ragged clouds (nullptr: B clouds of N rows).  Always the bucketed path when dy is given.
int grad(const float* dout, int64_t out_stride, int64_t out_offset, const int32_t* idx, const float* w, int B, int N, int S,
         int D, float* dpoints2, void* workspace, size_t workspace_bytes, hipStream_t stream, const int* coff, long long rows,
         const DySource* dy);

}  // namespace interp
}  // namespace pn2
