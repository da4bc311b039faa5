// Cell structure of a cloud, shared by the kernels that order points by locality (fps.hip builds it) and those that search
// by cell (ball_query.hip): every axis of the cloud's bounding box is cut into 16, cells are numbered along the Morton
// curve.  The box travels as order-preserving uint32 encodings {max x, max y, max z, max ~x, max ~y, max ~z, -, -}.
#pragma once
#include "pn2_common.h"

namespace pn2 {

constexpr int kCellBits = 4;                    // per axis: 16 x 16 x 16 cells of the cloud's bounding box
constexpr int kCells = 1 << (3 * kCellBits);

__device__ __forceinline__ unsigned ord_enc(float f) {   // order-preserving float -> u32
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

__device__ __forceinline__ unsigned spread3(unsigned v) {   // bit i -> bit 3 i (good for 8 bits)
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}

// Morton number of the cell of a point: every axis of the cloud's box is cut into 16 (non-finite values land in cell 0 / 15).
struct CellGrid {
    float lo[3], scale[3];
};
__device__ __forceinline__ CellGrid cell_grid(const unsigned* box) {
    CellGrid c;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.lo[a] = ord_dec(~box[3 + a]);
        const float ext = ord_dec(box[a]) - c.lo[a];
        c.scale[a] = ext > 0.0f ? (float)(1 << kCellBits) / ext : 0.0f;
    }
    return c;
}
__device__ __forceinline__ unsigned cell_of(const CellGrid& c, float x, float y, float z) {
    const float top = (float)((1 << kCellBits) - 1);
    const unsigned cx = (unsigned)fminf(fmaxf((x - c.lo[0]) * c.scale[0], 0.0f), top);
    const unsigned cy = (unsigned)fminf(fmaxf((y - c.lo[1]) * c.scale[1], 0.0f), top);
    const unsigned cz = (unsigned)fminf(fmaxf((z - c.lo[2]) * c.scale[2], 0.0f), top);
    return spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2);
}

// cell number along one axis (monotone non-decreasing in v: the cells of an interval are the cells of its end points)
__device__ __forceinline__ int cell_axis(const CellGrid& c, int a, float v) {
    const float top = (float)((1 << kCellBits) - 1);
    return (int)fminf(fmaxf((v - c.lo[a]) * c.scale[a], 0.0f), top);
}

}  // namespace pn2
