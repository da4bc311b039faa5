// Micro-benchmark (not part of the library): ONE kernel for the backward of a 128 -> 128 conv + BatchNorm + ReLU layer --
// dX = dY W (dgrad) and dW = dY^T X (wgrad) from a single staging of the dY tile -- against the library's two GEMM launches
// (130-165 us + 113 us at 262144 rows).  Persistent workgroups of 4 wavefronts, W resident in LDS, 32 rows per step.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/bwd_pair.hip -o gpurun_out/bwd_pair && gpurun_out/bwd_pair
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int C = 128, RT = 32, NT = 512, LDT = C + 1, LDW = C + 4;   // NT: 4 MFMA wavefronts + 4 staging wavefronts

struct Coef {   // per channel
    const float *mean, *scale, *beta, *a, *b;
};

// dY = scale * (mask ? dz : 0 - a - (y - mean) * b), mask = relu(bn(y)) > 0
__device__ __forceinline__ float tr_dy(float dz, float y, float mean, float sc, float bt, float a, float q) {
    const float t = __builtin_fmaf(y - mean, sc, bt);
    const float d = t > 0.0f ? dz : 0.0f;
    return sc * (d - a - (y - mean) * q);
}

__global__ __launch_bounds__(NT, 1) void bwd_pair_kernel(const float* __restrict__ dz, const float* __restrict__ y, Coef cy,
                                                         const float* __restrict__ xraw, Coef cx, const float* __restrict__ W,
                                                         int rows, float* __restrict__ dx, float* __restrict__ slab) {
    extern __shared__ float lds[];
    float* Ws = lds;                          // [C co][LDW]
    float* dYs = Ws + C * LDW;                // [2][RT][LDT]   dY rows
    float* Xs = dYs + 2 * RT * LDT;           // [2][RT][LDT]   RAW rows of the previous layer (activated on read)
    // wavefronts 0-3 multiply (consumers), 4-7 stage the next tile (producers): the SIMD interleaves one of each, so the
    // loads / transforms / LDS writes run in the shadow of the other wavefront's MFMAs without any instruction scheduling
    const bool producer = threadIdx.x >= 256;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int e = threadIdx.x; e < C * C / 4; e += NT) {
        const int co = e / (C / 4), c4 = (e % (C / 4)) * 4;
        *(float4*)(Ws + co * LDW + c4) = *(const float4*)(W + co * C + c4);
    }
    // staging (producers): thread = one COLUMN of 16 rows (rows rh, rh + 2, ...): a wavefront's loads are 256 contiguous bytes
    // of a row and its LDS writes hit 64 different banks (a float4 per lane means a 4-way bank conflict on every write)
    const int col = tid & 127, rh = tid >> 7;
    const float ym = cy.mean[col], ys = cy.scale[col], yb = cy.beta[col], ya = cy.a[col], yq = cy.b[col];
    const int steps_total = rows / RT;
    const int per = (steps_total + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per, t1 = min(t0 + per, steps_total);
    float rdz[16], ry[16], rx[16];      // the tile that is staged during this step (t + 1)
    float ndz[16], ny[16], nx[16];      // the one after it (t + 2): its loads have a whole step to arrive
    auto fetch = [&](int t) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const size_t o = ((size_t)t * RT + rh + 2 * p) * C + col;
            ndz[p] = dz[o];
            ny[p] = y[o];
            nx[p] = xraw[o];
        }
    };
    auto rotate = [&]() {
#pragma unroll
        for (int p = 0; p < 16; ++p) rdz[p] = ndz[p], ry[p] = ny[p], rx[p] = nx[p];
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            dYs[(buf * RT + rh + 2 * p) * LDT + col] = tr_dy(rdz[p], ry[p], ym, ys, yb, ya, yq);
            Xs[(buf * RT + rh + 2 * p) * LDT + col] = rx[p];
        }
    };
    // wgrad tile of this wave: co in [64 (wave >> 1), +64), ci in [64 (wave & 1), +64); dgrad tile: ci in [32 wave, +32)
    const int wco = 64 * (wave >> 1), wci = 64 * (wave & 1), dci = 32 * wave;
    float xm[2], xs[2], xb[2];   // X activation coefficients of the lane's wgrad B columns
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = wci + 32 * j + l31;
        xm[j] = cx.mean[ci], xs[j] = cx.scale[ci], xb[j] = cx.beta[ci];
    }
    f32x16 accw[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[i][j][r] = 0.0f;
    if (t0 < t1) {
        if (producer) {
            fetch(t0);
            rotate();
        }
        __syncthreads();   // W in LDS
        if (producer) {
            commit(0);
            fetch(t0 + 1 < t1 ? t0 + 1 : t1 - 1);
            rotate();
        }
    }
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        // LDS-only barrier: __syncthreads() would also wait for the dX stores and the prefetch loads in flight (vmcnt 0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (producer) {
#ifndef P_NOFETCH
            fetch(t + 2 < t1 ? t + 2 : t1 - 1);
#endif
#ifndef P_NOCOMMIT
            commit(buf ^ 1);       // tile t + 1 (in registers since the previous step)
#endif
            rotate();
            continue;
        }
        // ---- dgrad: dX[32 rows][32 ci of this wave] = dY[32][128] W[128][ci]
        f32x16 accd;
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[r] = 0.0f;
        const float* a = dYs + (buf * RT + l31) * LDT + half;    // A[m = row][k = co]
        const float* b = Ws + half * LDW + dci + l31;            // B[k = co][n = ci]
        // ---- wgrad: dW[co][ci] += dY^T[co][row] X[row][ci], K = 32 rows -- interleaved with the dgrad chain: the four
        // independent accumulators of a wgrad step fill the wait of the dependent dgrad MFMAs
        const float* at = dYs + (buf * RT + half) * LDT + wco + l31;   // A[m = co][k = row]
        const float* bt = Xs + (buf * RT + half) * LDT + wci + l31;    // B[k = row][n = ci] (raw: activated here)
#pragma unroll
        for (int s16 = 0; s16 < 16; ++s16) {
            const int k = 2 * s16;
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = at[k * LDT + 32 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = fmaxf(__builtin_fmaf(bt[k * LDT + 32 * j] - xm[j], xs[j], xb[j]), 0.0f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kd = 8 * s16 + 2 * q;
                accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kd], b[kd * LDW], accd, 0, 0, 0);
                accw[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 1], bv[q & 1], accw[q >> 1][q & 1], 0, 0, 0);
            }
        }
        // ---- dX rows out
        float* o = dx + ((size_t)t * RT + 4 * half) * C + dci + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * C] = accd[r];
    }
    if (producer) return;
    float* sl = slab + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sl[(size_t)(wco + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half) * C + wci + 32 * j + l31] = accw[i][j][r];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 262144;
    const int nblk = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<float> hdz((size_t)rows * C), hy((size_t)rows * C), hx((size_t)rows * C), hW(C * C), coef(8 * C);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto& v : hdz) v = rnd();
    for (auto& v : hy) v = rnd();
    for (auto& v : hx) v = rnd();
    for (auto& v : hW) v = rnd() * 0.1f;
    for (int c = 0; c < C; ++c) {
        coef[c] = rnd() * 0.1f; coef[C + c] = 1.0f + rnd() * 0.2f; coef[2 * C + c] = rnd() * 0.1f; coef[3 * C + c] = rnd() * 0.01f;
        coef[4 * C + c] = rnd() * 0.01f; coef[5 * C + c] = rnd() * 0.1f; coef[6 * C + c] = 1.0f + rnd() * 0.2f; coef[7 * C + c] = rnd() * 0.1f;
    }
    float *ddz, *dy, *dxr, *dW, *dcoef, *ddx, *dslab;
    CK(hipMalloc(&ddz, hdz.size() * 4)); CK(hipMalloc(&dy, hy.size() * 4)); CK(hipMalloc(&dxr, hx.size() * 4));
    CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dcoef, coef.size() * 4)); CK(hipMalloc(&ddx, hdz.size() * 4));
    CK(hipMalloc(&dslab, (size_t)nblk * C * C * 4));
    CK(hipMemcpy(ddz, hdz.data(), hdz.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dxr, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcoef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    Coef cy{dcoef, dcoef + C, dcoef + 2 * C, dcoef + 3 * C, dcoef + 4 * C}, cx{dcoef + 5 * C, dcoef + 6 * C, dcoef + 7 * C, nullptr, nullptr};
    const size_t lds = (size_t)(C * LDW + 4 * RT * LDT) * sizeof(float);
    CK(hipFuncSetAttribute((const void*)bwd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("rows %d, %d workgroups, %zu bytes of LDS\n", rows, nblk, lds);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(bwd_pair_kernel, dim3(nblk), dim3(NT), lds, 0, ddz, dy, cy, dxr, cx, dW, rows, ddx, dslab);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(bwd_pair_kernel, dim3(nblk), dim3(NT), lds, 0, ddz, dy, cy, dxr, cx, dW, rows, ddx, dslab);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3;
    printf("fused dgrad + wgrad: %.1f us  (%.1f TFLOP/s of 157.3 fp32 MFMA; library: 130-165 + 113 us)\n", us, 2.0 * 2.0 * rows * C * C / us * 1e-6);
    // ---- check a few entries against a float64 host computation
    std::vector<float> gdx((size_t)rows * C), gslab((size_t)nblk * C * C);
    CK(hipMemcpy(gdx.data(), ddx, gdx.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gslab.data(), dslab, gslab.size() * 4, hipMemcpyDeviceToHost));
    auto dyv = [&](size_t r, int co) {
        const double yy = hy[r * C + co], t = (yy - coef[co]) * coef[C + co] + coef[2 * C + co];
        const double d = t > 0 ? hdz[r * C + co] : 0.0;
        return coef[C + co] * (d - coef[3 * C + co] - (yy - coef[co]) * coef[4 * C + co]);
    };
    double worst = 0;
    for (int s = 0; s < 64; ++s) {
        const size_t r = (size_t)rand() % rows; const int ci = rand() % C;
        double ref = 0; for (int co = 0; co < C; ++co) ref += dyv(r, co) * hW[co * C + ci];
        worst = fmax(worst, fabs(ref - gdx[r * C + ci]));
    }
    printf("dX max abs error on 64 samples: %.3g\n", worst);
    const int chk = rows <= 8192 ? 16 : 0;
    worst = 0;
    for (int s = 0; s < chk; ++s) {
        const int co = rand() % C, ci = rand() % C;
        double ref = 0;
        for (size_t r = 0; r < (size_t)rows; ++r) {
            const double t = (hx[r * C + ci] - coef[5 * C + ci]) * coef[6 * C + ci] + coef[7 * C + ci];
            ref += dyv(r, co) * (t > 0 ? t : 0);
        }
        double got = 0; for (int b = 0; b < nblk; ++b) got += gslab[(size_t)b * C * C + co * C + ci];
        worst = fmax(worst, fabs(ref - got) / (fabs(ref) + 1e-3));
    }
    if (chk) printf("dW max relative error on %d entries: %.3g\n", chk, worst);
    return 0;
}
