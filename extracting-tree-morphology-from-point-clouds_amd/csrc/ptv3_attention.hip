// PointTransformerV3 serialized patch attention for gfx950 (second stage after serialize.hip).
// Replaces Modules/PointTransformerV3/blocks.py:384-437 (get_padding_and_inverse: pad / unpad / cu_seqlens index maps) and the
// non-flash attention branch of SerializedAttention.forward, :457-488:
//     q, k, v = qkv[order].reshape(-1, K, 3, H, D)...;  attn = softmax((q * scale) @ k^T);  feat = (attn @ v) -> [N', K, H * D]
// (the repository's configuration: enable_flash = False, no RPE, patch_size 1024, D = C / H = 16 in every stage,
// PointTransformerV3.py:268-286).
//
// One workgroup per (patch, head): the patch's K <= 1024 keys and values (16 floats each) are staged ONCE in LDS (2 x 68 KB of the
// 160 KB), every wavefront owns 64 queries whose scaled rows stay in registers, and the K x K score matrix is never written:
//   pass 1   S^T = K Q^T per 16-key block on the matrix cores -> running row maxima (exact softmax: max first, like torch)
//   pass 2   S^T again, p = exp(s - max), row sums, and O += P V on the matrix cores
// Computing the TRANSPOSED scores is what makes pass 2 free of data movement: the accumulator layout of a 16 x 16 MFMA (lane
// holds 4 rows of one column) is exactly the A-operand layout of the next one when the contraction runs over those 4 rows, so
// the probabilities go from one MFMA's result registers straight into the next MFMA's operand registers.
//   fp32 mode: v_mfma_f32_16x16x4_f32 (exact fp32 products and sums); bf16 mode: operands rounded to bfloat16,
//   v_mfma_f32_16x16x16_bf16 (one instruction per 16 x 16 x 16 block instead of four), fp32 accumulation, softmax in fp32.
#include "pn2_common.h"

namespace {

constexpr int AT = 1024;           // threads per workgroup: 16 wavefronts x 64 queries = 1024 queries
constexpr int AD = 16;             // head dimension
constexpr int ALD = AD + 1;        // LDS row pitch (floats): conflict-free operand reads
constexpr int AKMAX = 1024;        // longest patch
using f32x4 = __attribute__((ext_vector_type(4))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;

__device__ __forceinline__ short bf16_bits(float x) {   // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}

// pad / unpad / cu_seqlens (blocks.py:384-437).  off, offpad: [B + 1] prefix sums of the cloud sizes / padded cloud sizes.
//   unpad[off_i + j] = offpad_i + j;   pad[offpad_i + t] = off_i + (t < n_i ? t : t - K)   (the tail of the last patch repeats the
//   points one patch earlier);   cu_seqlens = every K-th padded position of every cloud, then the padded total.
__global__ __launch_bounds__(256) void ptv3_pad_kernel(const long long* __restrict__ off, const long long* __restrict__ offpad, int B,
                                                       int K, long long* __restrict__ pad, long long* __restrict__ unpad,
                                                       int* __restrict__ cu, const long long* __restrict__ cu_off) {
    const long long n_pad = offpad[B], n = off[B];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n_pad; p += stride) {
        int lo = 0, hi = B;                       // cloud of padded position p
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offpad[mid] <= p) lo = mid; else hi = mid;
        }
        const long long t = p - offpad[lo], ni = off[lo + 1] - off[lo];
        pad[p] = off[lo] + (t < ni ? t : t - K);
        if (t < ni) unpad[off[lo] + t] = p;
        if (t % K == 0) cu[cu_off[lo] + t / K] = (int)p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cu[cu_off[B]] = (int)n_pad;
    (void)n;
}

template <bool BF16>
__global__ __launch_bounds__(AT, 1) void ptv3_attention_kernel(const float* __restrict__ qkv, long long ld,
                                                               const long long* __restrict__ order, int K, int H, float scale,
                                                               float* __restrict__ out) {
    __shared__ float sK[AKMAX * ALD], sV[AKMAX * ALD];
    const int patch = blockIdx.x / H, h = blockIdx.x - patch * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = H * AD;
    const long long row0 = (long long)patch * K;
    // ---- stage keys and values (rows beyond K: zeros, masked below)
    const int KP = (K + 15) & ~15;
    for (int e = tid; e < KP * 4; e += AT) {            // 4 float4 per row of 16
        const int r = e >> 2, c4 = (e & 3) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (r < K) {
            const long long src = order ? order[row0 + r] : row0 + r;
            const float* p = qkv + src * ld + h * AD + c4;
            kv = *(const float4*)(p + C);
            vv = *(const float4*)(p + 2 * C);
        }
        float* dk = sK + r * ALD + c4;
        float* dv = sV + r * ALD + c4;
        dk[0] = kv.x, dk[1] = kv.y, dk[2] = kv.z, dk[3] = kv.w;
        dv[0] = vv.x, dv[1] = vv.y, dv[2] = vv.z, dv[3] = vv.w;
    }
    // ---- this wavefront's queries: 4 blocks of 16, operand layout (q = lane % 16, d = 4 * step + lane / 16), scaled like the
    //      reference scales q before the product
    const int l16 = lane & 15, g = lane >> 4;
    float qf[4][4];
    s16x4 qb[4];
    const int qbase = wave * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = qbase + 16 * j + l16;
        const long long src = q < K ? (order ? order[row0 + q] : row0 + q) : 0;
        const float* p = qkv + src * ld + h * AD;
        if (BF16) {
            const float4 v4 = q < K ? *(const float4*)(p + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            qb[j] = s16x4{bf16_bits(v4.x * scale), bf16_bits(v4.y * scale), bf16_bits(v4.z * scale), bf16_bits(v4.w * scale)};
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[j][s] = q < K ? p[4 * s + g] * scale : 0.0f;
        }
    }
    __syncthreads();
    if (qbase >= K) return;   // (a short patch: this wavefront has no queries; no barrier follows)
    const int nkb = KP / 16;
    // scores of one 16-key block for query block j: lane holds S^T[key = 4 g + i][q = lane % 16], i = 0..3
    auto scores = [&](int kb, int j, const float (&kf)[4], const s16x4& kbv) -> f32x4 {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (BF16) {
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kbv, qb[j], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[j][s], acc, 0, 0, 0);
        }
        // keys beyond the patch do not take part
        if (16 * kb + 16 > K) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (16 * kb + 4 * g + i >= K) acc[i] = -__builtin_inff();
        }
        return acc;
    };
    auto load_k = [&](int kb, float (&kf)[4], s16x4& kbv) {
        const float* kr = sK + (16 * kb + l16) * ALD;
        if (BF16) {
            kbv = s16x4{bf16_bits(kr[4 * g]), bf16_bits(kr[4 * g + 1]), bf16_bits(kr[4 * g + 2]), bf16_bits(kr[4 * g + 3])};
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = kr[4 * s + g];
        }
    };
    // ---- pass 1: row maxima
    float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int kb = 0; kb < nkb; ++kb) {
        float kf[4];
        s16x4 kbv;
        load_k(kb, kf, kbv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s = scores(kb, j, kf, kbv);
            m[j] = fmaxf(m[j], fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = fmaxf(m[j], __shfl_xor(m[j], 16, 64));
        m[j] = fmaxf(m[j], __shfl_xor(m[j], 32, 64));
    }
    // ---- pass 2: p = exp(s - max), row sums, O += P V
    float lsum[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
        float kf[4];
        s16x4 kbv;
        load_k(kb, kf, kbv);
        // values of the block, operand layout (d = lane % 16, key = 4 g + i)
        float vf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = sV[(16 * kb + 4 * g + i) * ALD + l16];
        const s16x4 vb = s16x4{bf16_bits(vf[0]), bf16_bits(vf[1]), bf16_bits(vf[2]), bf16_bits(vf[3])};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s = scores(kb, j, kf, kbv);
            float p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[i] = __expf(s[i] - m[j]);
                lsum[j] += p[i];
            }
            if (BF16) {
                const s16x4 pb = s16x4{bf16_bits(p[0]), bf16_bits(p[1]), bf16_bits(p[2]), bf16_bits(p[3])};
                o[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pb, vb, o[j], 0, 0, 0);
            } else {
                // the contraction over this block's 16 keys in four steps: step i takes keys {4 g + i}, the very registers p[i]
#pragma unroll
                for (int i = 0; i < 4; ++i) o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[i], vf[i], o[j], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store: o[j][i] = O[q = 4 g + i][d = lane % 16]; the row sums sit in lanes with lane % 16 = q
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lsum[j] += __shfl_xor(lsum[j], 16, 64);
        lsum[j] += __shfl_xor(lsum[j], 32, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float l = __shfl(lsum[j], 4 * g + i, 64);
            const int q = qbase + 16 * j + 4 * g + i;
            if (q < K) out[(row0 + q) * C + h * AD + l16] = o[j][i] / l;
        }
    }
}

}  // namespace

extern "C" int pn2_ptv3_pad_unpad_i64(const int64_t* off, const int64_t* offpad, const int64_t* cu_off, int B, int patch_size,
                                      int64_t n_pad, int64_t* pad, int64_t* unpad, int32_t* cu_seqlens, void* stream) {
    if (!off || !offpad || !cu_off || !pad || !unpad || !cu_seqlens || B <= 0 || patch_size <= 0 || n_pad < 0) return PN2_E_BADARG;
    long long blocks = (n_pad + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    PN2_LAUNCH("ptv3_pad", 24.0 * (double)n_pad, 0, ptv3_pad_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream,
               (const long long*)off, (const long long*)offpad, B, patch_size, (long long*)pad, (long long*)unpad, cu_seqlens,
               (const long long*)cu_off);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_ptv3_patch_attention_f32(const float* qkv, int64_t ld, const int64_t* order, int64_t n_rows, int patch_size,
                                            int heads, int head_dim, float scale, float* out, int precision, void* stream) {
    if (!qkv || !out || n_rows <= 0 || patch_size <= 0 || heads <= 0) return PN2_E_BADARG;
    if (head_dim != AD || patch_size > AKMAX || n_rows % patch_size) return PN2_E_BADARG;   // the repository's configuration
    if (ld % 4 || ((uintptr_t)qkv & 15) || ld < 3ll * heads * head_dim) return PN2_E_BADARG;
    if (precision != PN2_PRECISION_F32 && precision != PN2_PRECISION_BF16) return PN2_E_BADARG;
    const long long patches = n_rows / patch_size;
    if (patches * heads > 0x7FFFFFFFll) return PN2_E_BADARG;
    const double kk = (double)patch_size * patch_size;
    const double flops = (double)patches * heads * 4.0 * kk * AD;                       // Q K^T and P V once each
    const double bytes = 4.0 * (double)n_rows * heads * AD * 4.0;                        // q, k, v read, out written
    const dim3 grid((unsigned)(patches * heads)), block(AT);
    if (precision == PN2_PRECISION_BF16)
        PN2_LAUNCH("ptv3_attention_bf16", bytes, flops, (ptv3_attention_kernel<true>), grid, block, (hipStream_t)stream, qkv, (long long)ld,
                   (const long long*)order, patch_size, heads, scale, out);
    else
        PN2_LAUNCH("ptv3_attention", bytes, flops, (ptv3_attention_kernel<false>), grid, block, (hipStream_t)stream, qkv, (long long)ld,
                   (const long long*)order, patch_size, heads, scale, out);
    PN2_LAUNCH_CHECK();
    return 0;
}
