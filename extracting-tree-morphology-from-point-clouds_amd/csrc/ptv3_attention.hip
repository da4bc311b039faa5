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
                                                               float* __restrict__ out, float* __restrict__ lse) {
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
        // (training: the row's log-sum-exp, from which the backward kernels rebuild the probabilities)
        if (lse && g == 0 && qbase + 16 * j + l16 < K) lse[(row0 + qbase + 16 * j + l16) * H + h] = m[j] + __logf(lsum[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float l = __shfl(lsum[j], 4 * g + i, 64);
            const int q = qbase + 16 * j + 4 * g + i;
            if (q < K) out[(row0 + q) * C + h * AD + l16] = o[j][i] / l;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the patch attention (training).  With qs = q * scale, S = qs K^T, P = softmax(S) = exp(S - lse), O = P V:
//     delta_i = dO_i . O_i      dP = dO V^T      dS = P o (dP - delta)      dqs = dS K      dK = dS^T qs      dV = P^T dO
// Two launches, one workgroup per (patch, head) each, built like the forward kernel (one operand set staged in LDS once, the
// other in the owning wavefront's registers, probabilities straight from one MFMA's result registers into the next one's
// operand registers) -- the K x K matrices are rebuilt block by block, never written:
//   ROLE 0  a wavefront owns 64 QUERIES (qs, dO, lse, delta in registers), keys / values stream from LDS; S^T and dP^T blocks
//           (key rows, query columns) -> dS^T -> dqs += dS K
//   ROLE 1  a wavefront owns 64 KEYS (K, V in registers), qs / dO / lse / delta stream from LDS; S and dP blocks (query rows,
//           key columns) -> dV += P^T dO, dK += dS^T qs
// Exact fp32 MFMA (v_mfma_f32_16x16x4_f32) in both precisions of the forward.  The gradients go to the PADDED positions
// (dg [n_rows][3 C], rows like qkv's): a row of qkv that several padded positions read (the repeated tail of a cloud's last
// patch) collects them in the caller (index_add over `order`).
template <int ROLE>
__global__ __launch_bounds__(AT, 1) void ptv3_attention_bwd_kernel(const float* __restrict__ qkv, long long ld,
                                                                   const long long* __restrict__ order, int K, int H, float scale,
                                                                   const float* __restrict__ out, const float* __restrict__ lse,
                                                                   const float* __restrict__ dout, float* __restrict__ dg) {
    __shared__ float sA[AKMAX * ALD], sB[AKMAX * ALD];   // ROLE 0: K, V rows; ROLE 1: qs, dO rows
    __shared__ float sL[AKMAX], sD[AKMAX];               // ROLE 1: lse, delta of every query
    const int patch = blockIdx.x / H, h = blockIdx.x - patch * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = H * AD;
    const long long row0 = (long long)patch * K;
    const int KP = (K + 15) & ~15;
    for (int e = tid; e < KP * 4; e += AT) {
        const int r = e >> 2, c4 = (e & 3) * 4;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
        float part = 0.0f;
        if (r < K) {
            const long long src = order ? order[row0 + r] : row0 + r;
            const float* p = qkv + src * ld + h * AD + c4;
            if (ROLE == 0) {
                a4 = *(const float4*)(p + C);
                b4 = *(const float4*)(p + 2 * C);
            } else {
                a4 = *(const float4*)p;
                a4.x *= scale, a4.y *= scale, a4.z *= scale, a4.w *= scale;
                b4 = *(const float4*)(dout + (row0 + r) * C + h * AD + c4);
                const float4 o4 = *(const float4*)(out + (row0 + r) * C + h * AD + c4);
                part = (b4.x * o4.x + b4.y * o4.y) + (b4.z * o4.z + b4.w * o4.w);
            }
        }
        float* da = sA + r * ALD + c4;
        float* db = sB + r * ALD + c4;
        da[0] = a4.x, da[1] = a4.y, da[2] = a4.z, da[3] = a4.w;
        db[0] = b4.x, db[1] = b4.y, db[2] = b4.z, db[3] = b4.w;
        if (ROLE == 1) {   // the row's four quarters sit in four neighbouring lanes (KP * 4 is a multiple of 64: whole wavefronts)
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            if ((e & 3) == 0) {
                sD[r] = part;
                sL[r] = r < K ? lse[(row0 + r) * H + h] : __builtin_inff();   // rows beyond the patch: p = exp(-inf) = 0
            }
        }
    }
    // ---- the rows this wavefront owns: 4 blocks of 16, B-operand layout (row = lane % 16, d = 4 * step + lane / 16)
    const int l16 = lane & 15, g = lane >> 4;
    float uf[4][4], wf[4][4];   // ROLE 0: qs, dO;  ROLE 1: K, V
    float lq[4], dq_[4];        // ROLE 0: lse and delta of query lane % 16 of block j
    const int base = wave * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = base + 16 * j + l16;
        const bool in = r < K;
        const long long src = in ? (order ? order[row0 + r] : row0 + r) : 0;
        const float* p = qkv + src * ld + h * AD;
        float part = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (ROLE == 0) {
                uf[j][s] = in ? p[4 * s + g] * scale : 0.0f;
                wf[j][s] = in ? dout[(row0 + r) * C + h * AD + 4 * s + g] : 0.0f;
                part += in ? wf[j][s] * out[(row0 + r) * C + h * AD + 4 * s + g] : 0.0f;
            } else {
                uf[j][s] = in ? p[C + 4 * s + g] : 0.0f;
                wf[j][s] = in ? p[2 * C + 4 * s + g] : 0.0f;
            }
        }
        if (ROLE == 0) {
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            dq_[j] = part;
            lq[j] = in ? lse[(row0 + r) * H + h] : __builtin_inff();
        }
    }
    __syncthreads();
    if (base >= K) return;   // (a short patch: this wavefront owns nothing; no barrier follows)
    const int nb = KP / 16;
    f32x4 acc0[4], acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nb; ++b) {
        // streamed block: A-operand layout (row = lane % 16, d = 4 * step + lane / 16) and contraction layout (row = 4 g + i, d = lane % 16)
        float ar[4], br[4], ac[4], bc[4], ls[4], de[4];
        const float* pa = sA + (16 * b + l16) * ALD;
        const float* pb = sB + (16 * b + l16) * ALD;
#pragma unroll
        for (int s = 0; s < 4; ++s) ar[s] = pa[4 * s + g], br[s] = pb[4 * s + g];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ac[i] = sA[(16 * b + 4 * g + i) * ALD + l16];
            bc[i] = sB[(16 * b + 4 * g + i) * ALD + l16];
            if (ROLE == 1) ls[i] = sL[16 * b + 4 * g + i], de[i] = sD[16 * b + 4 * g + i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // lane holds element (streamed row 4 g + i, owned row lane % 16) of the score block and of dP's
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[s], uf[j][s], sc, 0, 0, 0);                       // K qs^T | qs K^T
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ROLE == 0 ? br[s] : br[s], wf[j][s], dp, 0, 0, 0);   // V dO^T | dO V^T
            }
            float pr[4], ds[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool live = ROLE == 0 ? (16 * b + 4 * g + i < K) : true;   // (ROLE 1: rows beyond the patch carry lse = inf)
                pr[i] = live ? __expf(sc[i] - (ROLE == 0 ? lq[j] : ls[i])) : 0.0f;
                ds[i] = pr[i] * (dp[i] - (ROLE == 0 ? dq_[j] : de[i]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ROLE == 0) {
                    acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[i], ac[i], acc0[j], 0, 0, 0);   // dqs += dS K
                } else {
                    acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[i], ac[i], acc0[j], 0, 0, 0);   // dK += dS^T qs
                    acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[i], bc[i], acc1[j], 0, 0, 0);   // dV += P^T dO
                }
            }
        }
    }
    // ---- store: acc[j][i] = gradient of owned row 4 g + i of block j, column d = lane % 16
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = base + 16 * j + 4 * g + i;
            if (r >= K) continue;
            float* d = dg + (row0 + r) * 3 * C + h * AD + l16;
            if (ROLE == 0) {
                d[0] = acc0[j][i] * scale;
            } else {
                d[C] = acc0[j][i];
                d[2 * C] = acc1[j][i];
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

static int attention_args_ok(const float* qkv, int64_t ld, int64_t n_rows, int patch_size, int heads, int head_dim) {
    if (!qkv || n_rows <= 0 || patch_size <= 0 || heads <= 0) return 0;
    if (head_dim != AD || patch_size > AKMAX || n_rows % patch_size) return 0;   // the repository's configuration
    if (ld % 4 || ((uintptr_t)qkv & 15) || ld < 3ll * heads * head_dim) return 0;
    return (n_rows / patch_size) * heads <= 0x7FFFFFFFll;
}

extern "C" int pn2_ptv3_patch_attention_lse_f32(const float* qkv, int64_t ld, const int64_t* order, int64_t n_rows, int patch_size,
                                                int heads, int head_dim, float scale, float* out, float* lse, int precision,
                                                void* stream) {
    if (!out || !attention_args_ok(qkv, ld, n_rows, patch_size, heads, head_dim)) return PN2_E_BADARG;
    if (precision != PN2_PRECISION_F32 && precision != PN2_PRECISION_BF16) return PN2_E_BADARG;
    const long long patches = n_rows / patch_size;
    const double kk = (double)patch_size * patch_size;
    const double flops = (double)patches * heads * 4.0 * kk * AD;                       // Q K^T and P V once each
    const double bytes = 4.0 * (double)n_rows * heads * AD * 4.0;                        // q, k, v read, out written
    const dim3 grid((unsigned)(patches * heads)), block(AT);
    if (precision == PN2_PRECISION_BF16)
        PN2_LAUNCH("ptv3_attention_bf16", bytes, flops, (ptv3_attention_kernel<true>), grid, block, (hipStream_t)stream, qkv, (long long)ld,
                   (const long long*)order, patch_size, heads, scale, out, lse);
    else
        PN2_LAUNCH("ptv3_attention", bytes, flops, (ptv3_attention_kernel<false>), grid, block, (hipStream_t)stream, qkv, (long long)ld,
                   (const long long*)order, patch_size, heads, scale, out, lse);
    PN2_LAUNCH_CHECK();
    return 0;
}

extern "C" int pn2_ptv3_patch_attention_f32(const float* qkv, int64_t ld, const int64_t* order, int64_t n_rows, int patch_size,
                                            int heads, int head_dim, float scale, float* out, int precision, void* stream) {
    return pn2_ptv3_patch_attention_lse_f32(qkv, ld, order, n_rows, patch_size, heads, head_dim, scale, out, nullptr, precision, stream);
}

extern "C" int pn2_ptv3_patch_attention_bwd_f32(const float* qkv, int64_t ld, const int64_t* order, int64_t n_rows, int patch_size,
                                                int heads, int head_dim, float scale, const float* out, const float* lse,
                                                const float* dout, float* dqkv_rows, void* stream) {
    if (!out || !lse || !dout || !dqkv_rows || !attention_args_ok(qkv, ld, n_rows, patch_size, heads, head_dim)) return PN2_E_BADARG;
    if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15)) return PN2_E_BADARG;
    const long long patches = n_rows / patch_size;
    const double kk = (double)patch_size * patch_size;
    const double bytes = 4.0 * (double)n_rows * heads * AD * 6.0;
    const dim3 grid((unsigned)(patches * heads)), block(AT);
    PN2_LAUNCH("ptv3_attention_bwd_q", bytes, (double)patches * heads * 6.0 * kk * AD, (ptv3_attention_bwd_kernel<0>), grid, block,
               (hipStream_t)stream, qkv, (long long)ld, (const long long*)order, patch_size, heads, scale, out, lse, dout, dqkv_rows);
    PN2_LAUNCH("ptv3_attention_bwd_kv", bytes, (double)patches * heads * 8.0 * kk * AD, (ptv3_attention_bwd_kernel<1>), grid, block,
               (hipStream_t)stream, qkv, (long long)ld, (const long long*)order, patch_size, heads, scale, out, lse, dout, dqkv_rows);
    PN2_LAUNCH_CHECK();
    return 0;
}
