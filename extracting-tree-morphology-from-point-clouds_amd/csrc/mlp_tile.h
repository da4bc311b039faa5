// The LDS-tiled MFMA tile body shared by the one-tile-per-workgroup GEMM kernels (mlp.hip) and the cooperative chain kernels
// (chain_coop.hip): operand staging with the BatchNorm / ReLU transforms, the epilogues, coherent tile I/O.  See the header of
// mlp.hip for the design.  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <stdlib.h>

#include <cstdlib>

#include "pn2_common.h"
#include "segments.h"

namespace {

constexpr int BK = 16, NT = 256;
#ifndef PN2_DGRAD_OCC
#define PN2_DGRAD_OCC 2
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

enum { TR_PLAIN = 0, TR_BNRELU = 1, TR_DY = 2, TR_DYP = 3 };   // TR_DYP: TR_DY whose dz is the max-pool's gradient, rebuilt per element

// Diagnostic build (tools/build_diag.sh, -DPN2_GEMM_DIAG): wave 0 of every workgroup of the one-tile-per-workgroup kernels
// stamps the shader clock at its phases into g_gemm_diag[linear workgroup][slot] (slot 0: 100 MHz wall clock at entry, 1..4:
// cycle counter after prepare / after the first tile is staged / after the K loop / after the epilogue, 5: wall clock at the
// end, 6: hardware id); tools/diag_gemm.py reads them back through pn2_gemm_diag_read().
#ifdef PN2_GEMM_DIAG
__device__ unsigned long long g_gemm_diag[16384][8];
#define PN2_GEMM_STAMP(SLOT)                                                                                           \
    do {                                                                                                               \
        if (threadIdx.x == 0) {                                                                                        \
            const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                        \
            if (wg < 16384) {                                                                                          \
                if ((SLOT) == 1) {                                                                                     \
                    g_gemm_diag[wg][0] = wall_clock64();                                                               \
                    g_gemm_diag[wg][6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |                 \
                                         ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
                }                                                                                                      \
                g_gemm_diag[wg][SLOT] = clock64();                                                                     \
                if ((SLOT) == 4) g_gemm_diag[wg][5] = wall_clock64();                                                  \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
#else
#define PN2_GEMM_STAMP(SLOT) do { } while (0)
#endif
enum { EPI_FWD = 0, EPI_STORE = 1, EPI_SLAB = 2 };
// the one-segment table of the cooperative kernels: nothing to search, nothing indexed dynamically
struct OneSeg {
    int rows;
};
__device__ __forceinline__ RowBlock row_block(const OneSeg& s, int blk, int R) { return RowBlock{0, blk * R, s.rows}; }

struct Operand {
    const float* p;   // X, Y or dZ: [rows][cols], cols (= channels) contiguous
    const float* q;   // TR_DY: the layer's pre-BN output Y
    long long ld, ldq;
    int rows, cols;
    const float* coef;  // coefficient block [ST_ROWS][cstride] of the BatchNorm involved, or null
    int cstride;
    int relu;
    int p16, q16;       // bf16 mode with bfloat16 STORAGE: p / q point to __bf16 rows (ld / ldq still count elements)
    // TR_DYP: the layer's output is max-pooled over groups of 2^pool_shift consecutive rows; p = the POOLED gradient
    // [rows >> pool_shift][cols] (ld = cols), parg8 = the arg-max row of every (group, channel) as ONE BYTE, same shape:
    //   dz[r][c] = (parg8[r >> shift][c] == (r & mask)) ? p[r >> shift][c] : 0
    // -- the dense scattered tensor (31/32 zeros at 32 rows per group) is never written or read
    const unsigned char* parg8;
    int pool_shift;
};

// Branch-free tile loads: the address is clamped into the matrix and the value masked afterwards, so that all of
// a thread's loads for a K-tile issue back to back and are waited for once (a branch per load made the compiler
// wait for each load separately: 8-12 dependent memory round trips per K-tile).
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* base, long long ld, int r, int c, int nrows, int ncols) {
    const int rc = r < nrows ? r : nrows - 1;
    if (VEC) {  // ncols % 4 == 0 and c % 4 == 0: c < ncols implies c + 3 < ncols
        const int cc = c < ncols ? c : ncols - 4;
        return *(const float4*)(base + (long long)rc * ld + cc);
    }
    const float* p = base + (long long)rc * ld;
    const int last = ncols - 1;
    float4 v;
    v.x = p[c < last ? c : last];
    v.y = p[c + 1 < last ? c + 1 : last];
    v.z = p[c + 2 < last ? c + 2 : last];
    v.w = p[c + 3 < last ? c + 3 : last];
    return v;
}

// The same load from rows stored as bfloat16 (bf16 mode, PN2_CHAIN_STORE_BF16): four values = 8 bytes, widened exactly.
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
template <bool VEC>
__device__ __forceinline__ float4 ld4h(const float* base_, long long ld, int r, int c, int nrows, int ncols) {
    const unsigned short* base = (const unsigned short*)base_;
    const int rc = r < nrows ? r : nrows - 1;
    if (VEC) {
        const int cc = c < ncols ? c : ncols - 4;
        const uint2 u = *(const uint2*)(base + (long long)rc * ld + cc);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xFFFF0000u));
    }
    const unsigned short* p = base + (long long)rc * ld;
    const int last = ncols - 1;
    return make_float4(bf16_to_f32(p[c < last ? c : last]), bf16_to_f32(p[c + 1 < last ? c + 1 : last]),
                       bf16_to_f32(p[c + 2 < last ? c + 2 : last]), bf16_to_f32(p[c + 3 < last ? c + 3 : last]));
}

// Y16 / DX16: the activation rows read / the input-gradient rows written are bfloat16 (bf16 mode with bfloat16 storage)
__device__ __forceinline__ float4 ld_row4(const float* base, long long elem, bool h16) {
    if (h16) {
        const uint2 u = *(const uint2*)((const unsigned short*)base + elem);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xFFFF0000u));
    }
    return *(const float4*)(base + elem);
}
__device__ __forceinline__ void st_row4(float* base, long long elem, float4 v, bool h16) {
    if (h16) {
        using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
        *(bf16x4*)((unsigned short*)base + elem) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    } else {
        *(float4*)(base + elem) = v;
    }
}

// COHERENT tile I/O for the cooperative chain kernels (chain_coop.hip part below): rows written by OTHER workgroups of
// the same launch are stored write-through (sc1) and read with L2-bypassing loads (sc1) -- every XCD has its own L2, a plain
// load may hit a stale line there (tools/ubench/gridbar.hip: control mode 3 reads garbage, modes 0/1 are exact).  Wide loads
// with the sc1 bit come from the buffer intrinsics (aux bit 4); the 32-bit byte offset limits a matrix to 4 GB.
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xFFFFFFFFu, 0x00020000);   // raw dwords, no bounds (gfx9 word 3)
}
constexpr int kAuxSc1 = 16;
template <bool VEC>
__device__ __forceinline__ float4 ld4_coh(const float* base, long long ld, int r, int c, int nrows, int ncols) {
    const int rc = r < nrows ? r : nrows - 1;
    const __amdgpu_buffer_rsrc_t rs = coh_rsrc(base);
    if (VEC) {
        const int cc = c < ncols ? c : ncols - 4;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)rc * ld + cc) * 4), 0, kAuxSc1));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    const int last = ncols - 1;
    const long long row = (long long)rc * ld;
    float4 v;
    v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((row + (c < last ? c : last)) * 4), 0, kAuxSc1));
    v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((row + (c + 1 < last ? c + 1 : last)) * 4), 0, kAuxSc1));
    v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((row + (c + 2 < last ? c + 2 : last)) * 4), 0, kAuxSc1));
    v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((row + (c + 3 < last ? c + 3 : last)) * 4), 0, kAuxSc1));
    return v;
}
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One operand's staging state for a TILE x 32 (T layout: global [outer][k]) or 32 x TILE (D layout: global
// [k][outer]) tile: TILE/32 float4 per thread (+ as many for the second source) and the BatchNorm coefficients
// of the thread's 4 channels in registers.  The channel is always the contiguous global index: k for the T
// layout (reloaded per K-tile), the outer index for the D layout (loaded once).
// IMG16: the LDS image is bfloat16, [outer][k] with a row pitch of 48 bytes (16 values + 16 bytes of padding): the matrix-core
// operand of v_mfma_f32_32x32x16_bf16 -- row r, eight consecutive k -- is then ONE conflict-free 16-byte read instead of eight
// 4-byte reads and eight conversions out of the fp32 [k][outer] image (which made the bf16 mode LDS-bound).
constexpr int kPitch16 = 48;
// ... and for a D-layout operand (global [k][outer], outer contiguous: both operands of a weight gradient, the weights of a dgrad)
// the image stays K-MAJOR -- [k][outer] bfloat16 rows, written with one 8-byte store per 8-byte load instead of four scattered
// 2-byte stores (which fall on 4 of the 64 banks: 8-way conflicts) -- and the matrix-core operand comes out of it with gfx950's
// transposing read, ds_read_b64_tr_b16 (two per operand).  Row pitch = tile + 32 values: 320 bytes at a 128-tile, i.e. four
// consecutive k rows start 64 bytes apart modulo the 256-byte bank row -- the 4 x 16 blocks of a half-wavefront's transposed
// read cover all 64 banks once.
#ifndef PN2_TR_IMAGE
#define PN2_TR_IMAGE 1
#endif
constexpr bool kTrImage = PN2_TR_IMAGE != 0;
template <int TILE> constexpr int kPitchTR = 2 * (TILE + 32);
using s16x4v = __attribute__((ext_vector_type(4))) short;
__device__ __forceinline__ bf16x8 tr_fragment(const char* image, int pitch, int outer0, int lane) {
    // lane (r = lane & 31, h = lane >> 5) gets [outer0 + r][k = 8 h .. 8 h + 7]: per 16-lane group a 4 (k) x 16 (outer) block,
    // lane 4 q + p of the group addresses row k0 + q, columns 4 p .. 4 p + 3
    const int li = lane & 15, q = li >> 2, pp = li & 3, grp = (lane >> 4) & 1, h = lane >> 5;
    const char* a = image + (8 * h + q) * pitch + 2 * (outer0 + 16 * grp + 4 * pp);
    using lds_ptr = __attribute__((address_space(3))) s16x4v*;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * pitch));
    using s16x8v = __attribute__((ext_vector_type(8))) short;
    const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
#ifndef PN2_PAIRED_STORE
#define PN2_PAIRED_STORE 0
#endif
constexpr bool kPairedStore = PN2_PAIRED_STORE != 0;   // bfloat16 C as 4-byte column pairs (measured slower than 2-byte stores: off)
template <bool T_LAYOUT, int KIND, int TILE, bool VEC, int THREADS = NT, bool COH = false, bool IMG16 = false, bool SRC16 = false>
struct Stager {
    static constexpr int NP = TILE * BK / (4 * THREADS);   // 16-byte loads per thread per K-tile
    static constexpr int KT = BK / 4;        // T layout: threads along k per row
    static constexpr int RPP = THREADS / KT; // T layout: rows per pass
    static constexpr int OQ = TILE / 4;      // D layout: threads per k-row
    static constexpr int KPP = THREADS / OQ; // D layout: k-rows per pass
    static constexpr int LD = TILE + 4;
    float4 v[NP], y[NP];
    unsigned parg[KIND == TR_DYP ? NP : 1];   // TR_DYP: the arg-max rows (one byte each) of the pass's four channels
    float cm[4], cs[4], cb[4], ca[4], cq[4];
    int tid;  // thread index inside the 256-thread team that stages this tile

    __device__ __forceinline__ int chan0(int o0, int k0) const {
        return T_LAYOUT ? k0 + 4 * (tid % KT) : o0 + 4 * (tid % OQ);
    }
    __device__ __forceinline__ int row_of(int p, int o0, int k0) const {
        return T_LAYOUT ? o0 + RPP * p + (tid / KT) : k0 + KPP * p + (tid / OQ);
    }
    __device__ __forceinline__ void load_coefs(const Operand& o, int c0) {
        if (KIND == TR_PLAIN) return;
        if (VEC) {  // cols % 4 == 0: one 16-byte load per coefficient row (the block is [8][cols], 16-byte aligned)
            const int cc = c0 < o.cols ? c0 : o.cols - 4;
            const float4 m4 = *(const float4*)(o.coef + ST_MEAN * o.cstride + cc);
            const float4 s4 = *(const float4*)(o.coef + ST_SCALE * o.cstride + cc);
            const float4 b4 = *(const float4*)(o.coef + ST_BETA * o.cstride + cc);
            cm[0] = m4.x; cm[1] = m4.y; cm[2] = m4.z; cm[3] = m4.w;
            cs[0] = s4.x; cs[1] = s4.y; cs[2] = s4.z; cs[3] = s4.w;
            cb[0] = b4.x; cb[1] = b4.y; cb[2] = b4.z; cb[3] = b4.w;
            if (KIND == TR_DY || KIND == TR_DYP) {
                const float4 a4 = *(const float4*)(o.coef + ST_A * o.cstride + cc);
                const float4 q4 = *(const float4*)(o.coef + ST_B * o.cstride + cc);
                ca[0] = a4.x; ca[1] = a4.y; ca[2] = a4.z; ca[3] = a4.w;
                cq[0] = q4.x; cq[1] = q4.y; cq[2] = q4.z; cq[3] = q4.w;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = c0 + j < o.cols ? c0 + j : o.cols - 1;
            cm[j] = o.coef[ST_MEAN * o.cstride + ch];
            cs[j] = o.coef[ST_SCALE * o.cstride + ch];
            cb[j] = o.coef[ST_BETA * o.cstride + ch];
            if (KIND == TR_DY || KIND == TR_DYP) {
                ca[j] = o.coef[ST_A * o.cstride + ch];
                cq[j] = o.coef[ST_B * o.cstride + ch];
            }
        }
    }
    __device__ __forceinline__ void prepare(const Operand& o, int o0) {
        if (!T_LAYOUT) load_coefs(o, chan0(o0, 0));
    }
    __device__ __forceinline__ void fetch(const Operand& o, int o0, int k0) {
        const int c = chan0(o0, k0);
        if (T_LAYOUT) load_coefs(o, c);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = row_of(p, o0, k0);
            if (COH && KIND != TR_PLAIN) {   // dZ / Y rows another workgroup of this launch may have written
                v[p] = ld4_coh<VEC>(o.p, o.ld, r, c, o.rows, o.cols);
                if (KIND == TR_DY) y[p] = ld4_coh<VEC>(o.q, o.ldq, r, c, o.rows, o.cols);
            } else if (SRC16) {   // bf16 mode, bfloat16 storage: both sources are __bf16 rows (compile-time: a branch between
                                  // the loads of a K-tile would serialise their round trips)
                v[p] = ld4h<VEC>(o.p, o.ld, r, c, o.rows, o.cols);
                if (KIND == TR_DY) y[p] = ld4h<VEC>(o.q, o.ldq, r, c, o.rows, o.cols);
            } else if (KIND == TR_DYP) {
                const int rr = r < o.rows ? r : o.rows - 1, gr = rr >> o.pool_shift;
                v[p] = ld4<VEC>(o.p, o.ld, gr, c, gr + 1, o.cols);
                if (VEC) {
                    const int cc = c < o.cols ? c : o.cols - 4;
                    parg[p] = *(const unsigned*)(o.parg8 + (long long)gr * o.cols + cc);
                } else {
                    const unsigned char* a = o.parg8 + (long long)gr * o.cols;
                    const int last = o.cols - 1;
                    parg[p] = (unsigned)a[c < last ? c : last] | ((unsigned)a[c + 1 < last ? c + 1 : last] << 8) |
                              ((unsigned)a[c + 2 < last ? c + 2 : last] << 16) | ((unsigned)a[c + 3 < last ? c + 3 : last] << 24);
                }
                y[p] = ld4<VEC>(o.q, o.ldq, r, c, o.rows, o.cols);
            } else {
                v[p] = ld4<VEC>(o.p, o.ld, r, c, o.rows, o.cols);
                if (KIND == TR_DY) y[p] = ld4<VEC>(o.q, o.ldq, r, c, o.rows, o.cols);
            }
        }
    }
    __device__ __forceinline__ float xf(float val, float yy, int j, int relu) const {
        if (KIND == TR_PLAIN) return val;
        if (KIND == TR_BNRELU) {
            const float t = __builtin_fmaf(val - cm[j], cs[j], cb[j]);
            return relu ? fmaxf(t, 0.0f) : t;
        }
        const float t = __builtin_fmaf(yy - cm[j], cs[j], cb[j]);
        const float dz = (!relu || t > 0.0f) ? val : 0.0f;
        return cs[j] * (dz - ca[j] - (yy - cm[j]) * cq[j]);
    }
    // stage pass p of the fetched tile into the LDS image (transform applied here, once per element)
    __device__ __forceinline__ void commit_pass(const Operand& o, float* S, int o0, int k0, int p) {
        const int c = chan0(o0, k0);
        const int r = row_of(p, o0, k0);
        const float e[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
        const float yy[4] = {y[p].x, y[p].y, y[p].z, y[p].w};
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // elements outside the matrix must be exactly zero: they pad the contraction
            const bool in = r < o.rows && c + j < o.cols;
            const float ej = KIND == TR_DYP ? (((parg[p] >> (8 * j)) & 0xFFu) == (unsigned)(r & ((1 << o.pool_shift) - 1)) ? e[j] : 0.0f) : e[j];
            w[j] = in ? xf(ej, (KIND == TR_DY || KIND == TR_DYP) ? yy[j] : 0.0f, j, o.relu) : 0.0f;
        }
        if (IMG16) {
            char* S16 = (char*)S;
            if (T_LAYOUT) {   // four consecutive k of one row: one 8-byte store
                const int m = RPP * p + (tid / KT), k = 4 * (tid % KT);
                using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
                *(bf16x4*)(S16 + m * kPitch16 + 2 * k) = bf16x4{(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
            } else if (kTrImage && TILE == 128) {   // one k of four consecutive rows: the K-major image, one 8-byte store
                const int k = KPP * p + (tid / OQ), m = 4 * (tid % OQ);
                using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
                *(bf16x4*)(S16 + k * kPitchTR<TILE> + 2 * m) = bf16x4{(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
            } else {          // one k of four consecutive rows
                const int k = KPP * p + (tid / OQ), m = 4 * (tid % OQ);
#pragma unroll
                for (int j = 0; j < 4; ++j) *(__bf16*)(S16 + (m + j) * kPitch16 + 2 * k) = (__bf16)w[j];
            }
        } else if (T_LAYOUT) {
            const int m = RPP * p + (tid / KT), k = 4 * (tid % KT);
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(k + j) * LD + m] = w[j];
        } else {
            const int k = KPP * p + (tid / OQ), m = 4 * (tid % OQ);
            *(float4*)(S + k * LD + m) = make_float4(w[0], w[1], w[2], w[3]);
        }
    }
    __device__ __forceinline__ void commit(const Operand& o, float* S, int o0, int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) commit_pass(o, S, o0, k0, p);
    }
};

struct GemmArgs {
    Operand A, B;
    int M, N, K;          // C[M][N] = sum_k A[m][k] B[k][n]
    float* C;             // EPI_FWD: Y [M][ldc]; EPI_STORE: [M][ldc]; EPI_SLAB: [split][M][ldc]
    long long ldc;
    const float* bias;    // EPI_FWD, may be null
    float* partial;       // EPI_FWD: [ceil(M/64)][2][N] (mean, M2) per 64-row chunk; EPI_STORE: (s1, s2); or null
    int k_per_split;      // EPI_SLAB: K range per blockIdx.z
    // EPI_STORE with partial: the output is dZ of the PREVIOUS layer; its BatchNorm-backward column sums are taken
    // here from the accumulators (needs that layer's Y and coefficient block)
    const float* ey;
    long long ldey;
    const float* ecoef;
    int erelu;
    long long pstride;    // partial layout: 0 = [chunk][2][N] (row-major); else channel-major, column c at c * pstride + 2 * chunk
    int c16, ey16;        // bf16 mode with bfloat16 storage: C (and what ACC reads of it) / ey are __bf16 rows
    int precision;        // host side only: PN2_PRECISION_* of the 128-tile contraction (travels with the call, no global state)
    int accumulate;       // EPI_STORE: C += result (PN2_CHAIN_ACCUMULATE_DX): the accumulators START from C -- the loads
                          // travel with the first K-tile's instead of forming a read-modify-write chain in the epilogue
    // DUAL kernels (the dgrad of two heads on the same input, dX = dY_a W_a + dY_b W_b as ONE contraction over K = 2 cout):
    // from contraction index ksplit on, the operands come from the second head -- same shapes, strides and ReLU flag
    const float *A2p, *A2q, *A2coef, *B2p;
    int ksplit;
};

// TILE = 128: four waves own 64 x 64 each (2 x 2 MFMA accumulators); TILE = 64: 32 x 32 each (one accumulator),
// for problems too small to fill the chip with 128-tiles.
// Epilogue shared by the GEMM kernels: the wave owns NI x NI 32x32 accumulators whose top-left element is
// (row0, col0); chunk_rows = rows covered by one wave (= one statistics chunk).
// FULL: the wave's sub-tile lies inside the matrix -- no per-element bounds checks (64 predicated stores otherwise)
template <int EPI, int NI, bool FULL, bool COH = false, bool C16 = false, bool E16 = false>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& g, f32x16 (&acc)[NI][NI], int row0, int col0, int chunk_rows,
                                              int lane, int split, long long chunk) {
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = row0, wm = 0, WT = chunk_rows, n0 = col0, wn = 0;   // names used by the body below
    // statistics partials: element (chunk, which, col) at chunk * pchunk + which * pwhich + col * pcol --
    // row-major [chunk][2][N] or channel-major (GemmArgs::pstride)
    const long long pchunk = g.pstride ? 2 : 2ll * g.N, pcol = g.pstride ? g.pstride : 1, pwhich = g.pstride ? 1 : g.N;
    (void)wm; (void)wn;
    // ---- epilogue.  C/D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* C = g.C;
    if (EPI == EPI_SLAB) C += (long long)split * g.M * g.ldc;
    const int rbase = m0 + wm * WT + 4 * half;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WT + 32 * j + l31;
        const bool cok = FULL || col < g.N;
        const float bias = (EPI == EPI_FWD && g.bias && cok) ? g.bias[col] : 0.0f;
        float sum = 0.0f;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2);
                const float val = acc[i][j][r] + bias;
                acc[i][j][r] = val;
                if (FULL || row < g.M) {
                    if (cok) {
                        if (COH && EPI != EPI_SLAB) st_coh(&C[(long long)row * g.ldc + col], val);
                        else if (C16 && EPI != EPI_SLAB) {
                            if (!FULL || !kPairedStore) ((__bf16*)C)[(long long)row * g.ldc + col] = (__bf16)val;   // (else: paired stores below)
                        }
                        else C[(long long)row * g.ldc + col] = val;
                    }
                    sum += val;
                    ++cnt;
                }
            }
        if (C16 && EPI != EPI_SLAB && FULL && kPairedStore) {
            // bfloat16 rows, two columns per store: lanes l and l ^ 1 hold adjacent columns of the same 16 rows; the even lane
            // writes both columns of the even row of a row pair, the odd lane both columns of the odd row -- 4-byte stores,
            // half as many of them as 2-byte ones
            const bool odd = lane & 1;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float mine = odd ? acc[i][j][r + 1] : acc[i][j][r];
                    const float send = odd ? acc[i][j][r] : acc[i][j][r + 1];
                    const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
                    const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2) + (odd ? 1 : 0);
                    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
                    const bf16x2 pr = odd ? bf16x2{(__bf16)recv, (__bf16)mine} : bf16x2{(__bf16)mine, (__bf16)recv};
                    *(bf16x2*)((__bf16*)C + (long long)row * g.ldc + (col & ~1)) = pr;
                }
        }
        // per-(row chunk of WT rows, column) partials; the other half-wave holds the other rows of the column
        if (EPI == EPI_FWD && g.partial) {
            sum += __shfl_xor(sum, 32, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mean = cnt > 0 ? sum / (float)cnt : 0.0f;
            float m2 = 0.0f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2);
                    const float d = acc[i][j][r] - mean;
                    if (FULL || row < g.M) m2 += d * d;
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (half == 0 && cok && m0 + wm * WT < g.M) {
                // one address formula for both layouts (a branch here costs the dgrad kernel 50 us: measured)
                float* pp = g.partial + chunk * pchunk + (long long)col * pcol;
                if (COH) {
                    st_coh(pp, mean);
                    st_coh(pp + pwhich, m2);
                } else {
                    pp[0] = mean;
                    pp[pwhich] = m2;
                }
            }
        }
        if (EPI == EPI_STORE && g.partial) {
            // BatchNorm-backward sums of the layer whose dZ this tile is: s1 = sum mask*dz, s2 = sum mask*dz*xhat
            const int cc = cok ? col : 0;
            const float mean = g.ecoef[ST_MEAN * g.N + cc], sc = g.ecoef[ST_SCALE * g.N + cc];
            const float bt = g.ecoef[ST_BETA * g.N + cc], invstd = g.ecoef[ST_INVSTD * g.N + cc];
            float s1 = 0.0f, s2 = 0.0f;
            float eyv[NI][16];
            if (E16 && FULL) {   // bfloat16 rows, two columns per 4-byte load, the halves exchanged between lanes l and l ^ 1
                const bool odd = lane & 1;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2) + (odd ? 1 : 0);
                        const unsigned u = *(const unsigned*)((const unsigned short*)g.ey + (long long)row * g.ldey + (cc & ~1));
                        const float lo = __uint_as_float(u << 16), hi = __uint_as_float(u & 0xFFFF0000u);
                        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? lo : hi), 0xB1, 0xF, 0xF, false));
                        eyv[i][r] = odd ? recv : lo;        // even lane: its own row r; odd lane: the partner loaded row r
                        eyv[i][r + 1] = odd ? hi : recv;
                    }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2);
                    const int rr = (FULL || row < g.M) ? row : g.M - 1;
                    const float yy = (E16 && FULL) ? eyv[i][r]
                                     : E16 ? bf16_to_f32(((const unsigned short*)g.ey)[(long long)rr * g.ldey + cc])
                                           : g.ey[(long long)rr * g.ldey + cc];
                    const float t = __builtin_fmaf(yy - mean, sc, bt);
                    const float dzh = ((FULL || row < g.M) && (!g.erelu || t > 0.0f)) ? acc[i][j][r] : 0.0f;
                    s1 += dzh;
                    s2 += dzh * ((yy - mean) * invstd);
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (half == 0 && cok && m0 + wm * WT < g.M) {
                float* pp = g.partial + chunk * pchunk + (long long)col * pcol;
                if (COH) {
                    st_coh(pp, s1);
                    st_coh(pp + pwhich, s2);
                } else {
                    pp[0] = s1;
                    pp[pwhich] = s2;
                }
            }
        }
    }
}

template <int EPI, int NI, bool COH = false, bool C16 = false, bool E16 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[NI][NI], int row0, int col0, int chunk_rows,
                                              int lane, int split, long long chunk) {
    if (row0 + 32 * NI <= g.M && col0 + 32 * NI <= g.N)
        gemm_epilogue_impl<EPI, NI, true, COH, C16, E16>(g, acc, row0, col0, chunk_rows, lane, split, chunk);
    else
        gemm_epilogue_impl<EPI, NI, false, COH, C16, E16>(g, acc, row0, col0, chunk_rows, lane, split, chunk);
}

// TEAMS = 4 (64-tiles only): the block holds four 256-thread teams that each contract a quarter of K into their own
// accumulators and LDS buffers; the quarters are summed through LDS at the end.  Deep levels have GEMMs with a few
// hundred rows and K up to 768: a handful of workgroups whose serial K loop is pure latency -- four teams put four
// times the loads in flight and give every SIMD four waves to interleave.
// BF16 (throughput mode, 128-tiles only): the operands are rounded to bfloat16 (round to nearest even) on their way from
// the fp32 LDS image into the matrix core and multiplied by v_mfma_f32_32x32x16_bf16 -- one instruction per K-tile and
// accumulator instead of eight fp32 ones, fp32 accumulation, everything else (staging, transforms, epilogues, what is
// stored in HBM) unchanged.  The contraction then costs 1/8 of the matrix-pipe time and the kernel is bound by HBM.
// ACC (EPI_STORE only): the accumulators start from C (GemmArgs::accumulate) -- a variant of its own: even a never-taken
// branch around those 64 loads costs every other dgrad launch ~10 us.
// The tile body.  BX / BY / BZ: the tile's coordinates (the launch's blockIdx for the one-tile-per-workgroup kernels; the
// cooperative chain kernels walk tile lists).  COH: the A operand and the outputs cross workgroups of ONE launch (coherent
// loads, write-through stores, see ld4_coh); the teams' early exit becomes a fall-through (every thread reaches the caller's
// barrier).  `lds`: TEAMS * 4 * BK * (TILE + 4) floats.
// NTT: threads per K-team -- 256 (four wavefronts, 2 x 2 over the tile), or 64 for the 32 x 32 tile of the smallest problems:
// ONE wavefront per team, so that a layer of a few hundred rows still spreads over a few hundred compute units (a 64-tile
// pins 2 * 64 * 64 * K flops to one unit's matrix cores: 1.7 us per K-tile with four teams on it, tools/diag_coop.py).
// ST (bf16 mode with bfloat16 storage, pn2_hip.h PN2_CHAIN_STORE_BF16): bit 0 the A operand's rows, bit 1 the B operand's rows,
// bit 2 C (and what ACC reads of it), bit 3 the epilogue's ey rows are __bf16.
template <bool A_T, int A_KIND, bool B_T, int B_KIND, int EPI, int TILE, bool VEC, int TEAMS, bool BF16, bool ACC, bool COH, int NTT, int ST,
          bool DUAL = false, class TAB = SegTable>
__device__ __forceinline__ void gemm_body(const GemmArgs& g0, const TAB& st, const int BX, const int BY, const int BZ,
                                          float* __restrict__ lds) {
    static_assert(NTT == NT || (NTT == 64 && TILE == 32 && TEAMS > 1), "single-wavefront teams are for the 32-tile");
    static_assert(!BF16 || TILE * kPitch16 <= BK * (TILE + 4) * 4, "the bfloat16 image fits the fp32 image's buffer");
    constexpr int LD = TILE + 4, WT = NTT == 64 ? TILE : TILE / 2, NI = WT / 32;
    // Which rows does this workgroup own?  Row tiles (forward, dgrad) and reduction ranges (wgrad) never straddle a
    // segment: the operands' row limit, the output's row limit and the BatchNorm coefficient blocks are those of the
    // block's segment (st.nseg == 1: the whole matrix, coefficient block 0).
    GemmArgs g = g0;
    int m0 = BX * TILE, k_begin = 0, k_end = g.K, seg;
    if (EPI == EPI_SLAB) {
        const RowBlock rb = row_block(st, BZ, g.k_per_split);
        seg = rb.seg;
        k_begin = rb.row0;
        k_end = rb.row0 + g.k_per_split < rb.row_end ? rb.row0 + g.k_per_split : rb.row_end;
        g.A.rows = g.B.rows = k_end;        // direct layout: rows are the contraction index
    } else {
        const RowBlock rb = row_block(st, BX, TILE);
        seg = rb.seg;
        m0 = rb.row0;
        g.A.rows = g.M = rb.row_end;
    }
    if (g.A.coef) g.A.coef += (long long)seg * ST_ROWS * g.A.cstride;
    if (DUAL && g.A2coef) g.A2coef += (long long)seg * ST_ROWS * g.A.cstride;
    if (g.B.coef) g.B.coef += (long long)seg * ST_ROWS * g.B.cstride;
    if (EPI == EPI_STORE && g.ecoef) g.ecoef += (long long)seg * ST_ROWS * g.N;
    static_assert(TEAMS == 1 || TILE == 64 || TILE == 32, "teams are for the small-problem tiles");
    // one LDS object (it is re-used as the teams' reduction buffer): [team][A|B][buffer][BK * LD]
    const int team = (int)threadIdx.x / NTT, tid = (int)threadIdx.x % NTT;
    float(*As)[BK * LD] = (float(*)[BK * LD])(lds + (team * 4 + 0) * BK * LD);
    float(*Bs)[BK * LD] = (float(*)[BK * LD])(lds + (team * 4 + 2) * BK * LD);

    const int n0 = BY * TILE;
    int nk = (k_end - k_begin + BK - 1) / BK;
    if (TEAMS > 1) {  // every team runs the same number of K-tiles (loads beyond K are zero-filled)
        nk = (nk + TEAMS - 1) / TEAMS;
        k_begin += team * nk * BK;
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if (ACC && EPI == EPI_STORE && (ST & 4) && m0 + wm * WT + 32 * NI <= g.M && n0 + wn * WT + 32 * NI <= g.N) {
        // bfloat16 C, full sub-tile: two columns per 4-byte load, halves exchanged between lanes l and l ^ 1
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = n0 + wn * WT + 32 * j + l31;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = m0 + wm * WT + 4 * half + 32 * i + (r & 3) + 8 * (r >> 2) + (odd ? 1 : 0);
                    const unsigned u = *(const unsigned*)((const unsigned short*)g.C + (long long)row * g.ldc + (col & ~1));
                    const float lo = __uint_as_float(u << 16), hi = __uint_as_float(u & 0xFFFF0000u);
                    const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? lo : hi), 0xB1, 0xF, 0xF, false));
                    acc[i][j][r] = odd ? recv : lo;
                    acc[i][j][r + 1] = odd ? hi : recv;
                }
            }
    } else if (ACC && EPI == EPI_STORE && team == 0) {   // uniform per wavefront; the other teams add their partial tiles later
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = n0 + wn * WT + 32 * j + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WT + 4 * half + 32 * i + (r & 3) + 8 * (r >> 2);
                    if (row < g.M && col < g.N)
                        acc[i][j][r] = (ST & 4) ? bf16_to_f32(((const unsigned short*)g.C)[(long long)row * g.ldc + col])
                                                : g.C[(long long)row * g.ldc + col];
                }
            }
    }

    static_assert(ST == 0 || BF16, "bfloat16 storage belongs to the bf16 mode");
    Stager<A_T, A_KIND, TILE, VEC, NTT, COH, BF16, (ST & 1) != 0> sa;
    Stager<B_T, B_KIND, TILE, VEC, NTT, false, BF16, (ST & 2) != 0> sb;
    sa.tid = tid;
    sb.tid = tid;
    sa.prepare(g.A, m0);
    sb.prepare(g.B, n0);
    PN2_GEMM_STAMP(1);
    if (nk > 0) {
        sa.fetch(g.A, m0, k_begin);
        sb.fetch(g.B, n0, k_begin);
        sa.commit(g.A, As[0], m0, k_begin);
        sb.commit(g.B, Bs[0], n0, k_begin);
    }
    __syncthreads();
    PN2_GEMM_STAMP(2);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        int knext = k_begin + (kt + 1) * BK;
        Operand da_ = g.A, db_ = g.B;        // DUAL: the operands the NEXT tile is read from -- the second head's from ksplit on
        if (DUAL && knext >= g.ksplit) {
            da_.p = g.A2p, da_.q = g.A2q, da_.coef = g.A2coef, db_.p = g.B2p;
            knext -= g.ksplit;
        }
        const Operand& na = DUAL ? da_ : g.A;
        const Operand& nb = DUAL ? db_ : g.B;
        if (kt + 1 < nk) {
            sa.fetch(na, m0, knext);
            sb.fetch(nb, n0, knext);
        }
        const float* a = As[cur] + wm * WT + l31;
        const float* b = Bs[cur] + wn * WT + l31;
        (void)a;
        (void)b;
        // The K-tile's MFMAs are issued in NP groups; after each group one pass of the NEXT tile is staged into the
        // other LDS buffer, so the staging VALU/LDS work sits in the shadow of the (asynchronous, 64-cycle) MFMAs
        // instead of forming a separate phase during which this wave's matrix pipe idles.
        constexpr int NP = TILE * BK / (4 * NTT), KQ = BK / NP;
        if constexpr (BF16) {
            static_assert(!BF16 || BK == 16, "one 32x32x16 MFMA consumes a whole K-tile");
            // lane (r = lane & 31, h = lane >> 5) supplies A[row r][k = 8h + e] and B[k = 8h + e][col r], e = 0..7: eight
            // consecutive values of row r of the bfloat16 image
            bf16x8 af[NI], bf[NI];
            const char* a16 = (const char*)As[cur] + (wm * WT + l31) * kPitch16 + 16 * half;
            const char* b16 = (const char*)Bs[cur] + (wn * WT + l31) * kPitch16 + 16 * half;
            constexpr bool A_TR = kTrImage && !A_T && TILE == 128, B_TR = kTrImage && !B_T && TILE == 128;
            static_assert(BK * kPitchTR<TILE> <= BK * (TILE + 4) * 4, "the K-major image fits the fp32 image's buffer");
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                af[i] = A_TR ? tr_fragment((const char*)As[cur], kPitchTR<TILE>, wm * WT + 32 * i, lane)
                             : *(const bf16x8*)(a16 + 32 * i * kPitch16);
                bf[i] = B_TR ? tr_fragment((const char*)Bs[cur], kPitchTR<TILE>, wn * WT + 32 * i, lane)
                             : *(const bf16x8*)(b16 + 32 * i * kPitch16);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (kt + 1 < nk) {
                sa.commit(na, As[cur ^ 1], m0, knext);
                sb.commit(nb, Bs[cur ^ 1], n0, knext);
            }
        } else
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int kk = q * KQ; kk < (q + 1) * KQ; kk += 2) {
                const int ko = (kk + half) * LD;
                float av[NI], bv[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    av[i] = a[ko + 32 * i];
                    bv[i] = b[ko + 32 * i];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nk) {
                sa.commit_pass(na, As[cur ^ 1], m0, knext, q);
                sb.commit_pass(nb, Bs[cur ^ 1], n0, knext, q);
            }
        }
        __syncthreads();
    }
    PN2_GEMM_STAMP(3);

    if (TEAMS > 1) {
        // sum the teams' partial tiles through LDS (the staging buffers are free now); team 0 runs the epilogue
        float* red = lds;   // (TEAMS - 1) x NT x 16 floats = 48 KiB of the 68 KiB
        static_assert(TEAMS == 1 || (TEAMS - 1) * 16 * NTT <= TEAMS * 4 * BK * LD, "reduction buffer must fit");
        if (team > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((team - 1) * 16 + r) * NTT + tid] = acc[0][0][r];
        }
        __syncthreads();
        if (team == 0) {
#pragma unroll
            for (int t = 0; t < TEAMS - 1; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += red[(t * 16 + r) * NTT + tid];
        }
    }
    if (TEAMS == 1 || team == 0)
        gemm_epilogue<EPI, NI, COH, (ST & 4) != 0, (ST & 8) != 0>(g, acc, m0 + wm * WT, n0 + wn * WT, WT, lane, BZ,
                                                                  (long long)BX * (TILE / WT) + wm);
    PN2_GEMM_STAMP(4);
}

// One tile per workgroup: the launch grid is the tile grid.
template <bool A_T, int A_KIND, bool B_T, int B_KIND, int EPI, int TILE, bool VEC, int TEAMS, bool BF16 = false, bool ACC = false,
          int NTT = NT, int ST = 0, bool DUAL = false>
__global__ __launch_bounds__(NTT * TEAMS, (NTT == 64 ? 4 : TEAMS > 1 ? 1 : (EPI == EPI_STORE ? PN2_DGRAD_OCC : 3))) void gemm_kernel(const GemmArgs g0,
                                                                                                                        const SegTable st) {
    __shared__ __attribute__((aligned(16))) float lds[TEAMS * 4 * BK * (TILE + 4)];
    gemm_body<A_T, A_KIND, B_T, B_KIND, EPI, TILE, VEC, TEAMS, BF16, ACC, false, NTT, ST, DUAL>(g0, st, (int)blockIdx.x, (int)blockIdx.y,
                                                                                      (int)blockIdx.z, lds);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// vector (16 B) staging is legal when rows start 16-byte aligned and the contiguous extent is a multiple of 4
inline int vec_ok(const Operand& o) {
    return (o.ld % 4 == 0) && (o.cols % 4 == 0) && aligned16(o.p) && (!o.q || (o.ldq % 4 == 0 && aligned16(o.q)));
}

Operand plain(const float* p, long long ld, int rows, int cols) {
    Operand o{};
    o.p = p;
    o.ld = ld;
    o.rows = rows;
    o.cols = cols;
    return o;
}

// activation source of a layer: raw rows (first layer) or the previous layer's Y seen through its BN+ReLU
struct Act {
    const float* p;
    long long ld;
    const float* coef;  // null -> plain
    int relu;
    int h16;            // the rows are stored as bfloat16 (bf16 mode with bfloat16 storage)
};

Operand act_operand(const Act& a, int rows, int cols) {
    Operand o = plain(a.p, a.ld, rows, cols);
    o.coef = a.coef;
    o.cstride = cols;
    o.relu = a.relu;
    o.p16 = a.h16;
    return o;
}

}  // namespace
