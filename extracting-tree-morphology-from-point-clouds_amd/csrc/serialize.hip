// Point-cloud serialization for gfx950 -- the space-filling-curve codes of Modules/PointTransformerV3/serialization
// (default.py:8-40 encode / decode, z_order.py:86-125 xyz2key / key2xyz, hilbert.py:91-198 encode, :201-302 decode), the
// first stage of the PointTransformerV3 backbone (blocks.py:98-150 Point.serialization).
//
// The reference builds a z-order key from two 256-entry lookup tables per axis and a Hilbert key by running Skilling's
// transform on [N, 3, depth] arrays of single BITS (3 * depth dependent tensor passes over 48 bytes per point).  Both are
// integer functions of the low `depth` bits of the three grid coordinates; here one thread computes every requested order
// of a point from registers:
//   z        bit i of x, y, z goes to bit 3 i + 2, 3 i + 1, 3 i of the code                         (z_order.py:45-55)
//   hilbert  Skilling, "Programming the Hilbert curve" (2004), axes -> transpose: from the top bit down, for every axis
//            either invert the lower bits of axis 0 (bit set) or exchange the lower bits of axis 0 and the axis where they
//            differ (bit clear)                                                                       (hilbert.py:157-177);
//            the three words are then interleaved axis 0 first (:180) and that 3 * depth-bit string is Gray-DEcoded by a
//            prefix xor from its top bit (:183, gray2binary :66-88)
//   *-trans  the same with the first two coordinates exchanged                                       (default.py:13-18)
//   batch    code |= batch << 3 depth                                                               (default.py:21-23)
// decode is the inverse walk.  8 (decode: 24) bytes out per point and order, 12 (+8) in: HBM-bound.
#include "pn2_common.h"

namespace {

constexpr int kBlock = 256;
using u64 = unsigned long long;

// bit i of v (i < 21) -> bit 3 i
__device__ __forceinline__ u64 spread3_64(u64 v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__device__ __forceinline__ u64 compact3_64(u64 v) {
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
    v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
    v = (v | (v >> 8)) & 0x1F0000FF0000FFull;
    v = (v | (v >> 16)) & 0x1F00000000FFFFull;
    v = (v | (v >> 32)) & 0x1FFFFFull;
    return v;
}
__device__ __forceinline__ u64 interleave3(unsigned a, unsigned b, unsigned c) {
    return (spread3_64(a) << 2) | (spread3_64(b) << 1) | spread3_64(c);
}

__device__ __forceinline__ u64 hilbert_code(unsigned x0, unsigned x1, unsigned x2, int depth) {
    unsigned X[3] = {x0, x1, x2};
    for (unsigned q = 1u << (depth - 1); q > 0; q >>= 1) {
        const unsigned p = q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & q) {
                X[0] ^= p;
            } else {
                const unsigned t = (X[0] ^ X[i]) & p;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    u64 h = interleave3(X[0], X[1], X[2]);
    h ^= h >> 1;
    h ^= h >> 2;
    h ^= h >> 4;
    h ^= h >> 8;
    h ^= h >> 16;
    h ^= h >> 32;
    return h;
}

__device__ __forceinline__ void hilbert_point(u64 h, int depth, unsigned& x0, unsigned& x1, unsigned& x2) {
    h ^= h >> 1;   // binary -> Gray (hilbert.py:257, binary2gray :44-63)
    unsigned X[3] = {(unsigned)compact3_64(h >> 2), (unsigned)compact3_64(h >> 1), (unsigned)compact3_64(h)};
    for (unsigned q = 1; q < (1u << depth); q <<= 1) {   // from the lowest bit up, axes backwards (hilbert.py:264-285)
        const unsigned p = q - 1;
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (X[i] & q) {
                X[0] ^= p;
            } else {
                const unsigned t = (X[0] ^ X[i]) & p;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    x0 = X[0], x1 = X[1], x2 = X[2];
}

struct Orders {
    int n;
    int code[8];   // PN2_ORDER_*
};

__global__ __launch_bounds__(kBlock) void serialize_encode_kernel(const int32_t* __restrict__ grid, int64_t gs, int64_t gc,
                                                                  const int64_t* __restrict__ batch, long long N, int depth,
                                                                  const Orders ord, int64_t* __restrict__ out) {
    const unsigned mask = depth >= 32 ? 0xFFFFFFFFu : ((1u << depth) - 1u);
    for (long long n = (long long)blockIdx.x * kBlock + threadIdx.x; n < N; n += (long long)gridDim.x * kBlock) {
        const int32_t* g = grid + n * gs;
        const unsigned x = (unsigned)g[0] & mask, y = (unsigned)g[gc] & mask, z = (unsigned)g[2 * gc] & mask;
        const u64 hi = batch ? ((u64)batch[n] << (3 * depth)) : 0ull;
        for (int o = 0; o < ord.n; ++o) {
            const int kind = ord.code[o];
            const unsigned a = (kind & 1) ? y : x, b = (kind & 1) ? x : y;   // "-trans": columns 0 and 1 exchanged
            const u64 c = (kind & 2) ? hilbert_code(a, b, z, depth) : interleave3(a, b, z);
            out[(long long)o * N + n] = (int64_t)(hi | c);
        }
    }
}

__global__ __launch_bounds__(kBlock) void serialize_decode_kernel(const int64_t* __restrict__ code, long long N, int depth,
                                                                  int hilbert, int64_t* __restrict__ grid,
                                                                  int64_t* __restrict__ batch) {
    for (long long n = (long long)blockIdx.x * kBlock + threadIdx.x; n < N; n += (long long)gridDim.x * kBlock) {
        const u64 c = (u64)code[n];
        const u64 key = c & ((1ull << (3 * depth)) - 1ull);
        unsigned x, y, z;
        if (hilbert) {
            hilbert_point(key, depth, x, y, z);
        } else {
            x = (unsigned)compact3_64(key >> 2), y = (unsigned)compact3_64(key >> 1), z = (unsigned)compact3_64(key);
        }
        grid[3 * n] = x, grid[3 * n + 1] = y, grid[3 * n + 2] = z;
        if (batch) batch[n] = code[n] >> (3 * depth);   // arithmetic shift, like the reference's
    }
}

inline unsigned grid_for(long long total) {
    long long g = (total + kBlock - 1) / kBlock;
    return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

}  // namespace

extern "C" int pn2_serialize_encode_i64(const int32_t* grid_coord, int64_t gs, int64_t gc, const int64_t* batch, long long N,
                                        int depth, const int32_t* orders, int n_orders, int64_t* out_codes, void* stream) {
    if (!grid_coord || !orders || !out_codes || N <= 0 || depth < 1 || depth > 16 || n_orders < 1 || n_orders > 8)
        return PN2_E_BADARG;
    Orders ord{};
    ord.n = n_orders;
    for (int i = 0; i < n_orders; ++i) {
        if (orders[i] < 0 || orders[i] > 3) return PN2_E_BADARG;
        ord.code[i] = orders[i];
    }
    PN2_LAUNCH("serialize_encode", (double)N * (12.0 + (batch ? 8.0 : 0.0) + 8.0 * n_orders), 0, serialize_encode_kernel,
               dim3(grid_for(N)), dim3(kBlock), (hipStream_t)stream, grid_coord, gs, gc, batch, N, depth, ord, out_codes);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_serialize_decode_i64(const int64_t* codes, long long N, int depth, int order, int64_t* out_grid,
                                        int64_t* out_batch, void* stream) {
    if (!codes || !out_grid || N <= 0 || depth < 1 || depth > 16 || (order != 0 && order != 2)) return PN2_E_BADARG;
    PN2_LAUNCH("serialize_decode", (double)N * (8.0 + 24.0 + (out_batch ? 8.0 : 0.0)), 0, serialize_decode_kernel,
               dim3(grid_for(N)), dim3(kBlock), (hipStream_t)stream, codes, N, depth, order == 2, out_grid, out_batch);
    PN2_LAUNCH_CHECK();
    return 0;
}
