// Micro-benchmark: what does a grid-wide barrier WITH data exchange cost inside one persistent kernel, by flavour of the
// stores / loads that carry the exchanged data?  G co-resident workgroups run L "layers": write a tile, barrier (arrival
// counter, one polling lane), read the tile ANOTHER workgroup wrote (most likely on another XCD) and check it.  The same
// addresses are rewritten every second layer, so a stale line in the reader's L2 / L1 shows up as an error count.
//   mode 0  write-through stores (sc1) ........ counter ... L2-bypassing loads (sc1)
//   mode 1  write-through stores (sc1) ........ counter ... acquire fence (buffer_inv sc1), plain loads
//   mode 2  plain stores, release fence (wbl2)  counter ... acquire fence, plain loads          (the textbook form)
//   mode 3  plain stores, NO fence ............ counter ... plain loads   (expected to FAIL across XCDs: control)
// Reports microseconds per layer and the number of wrong elements read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void layers(float* buf0, float* buf1, int tile_floats, int L, unsigned* counter, unsigned* errors,
                                              unsigned spin_limit) {
    const int G = gridDim.x, me = blockIdx.x, t = threadIdx.x;
    __shared__ int dead;
    if (t == 0) dead = 0;
    __syncthreads();
    unsigned bad = 0;
    for (int l = 0; l < L; ++l) {
        float* buf = (l & 1) ? buf1 : buf0;
        float* mine = buf + (size_t)me * tile_floats;
        for (int i = t; i < tile_floats; i += 256) {
            const float v = (float)(l * 1000 + me) + (float)i * (1.0f / 65536.0f);
            if (MODE <= 1) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[i] = v;
        }
        if (MODE == 2) __threadfence();
        __builtin_amdgcn_s_waitcnt(0);   // every store of this thread has been acknowledged
        __syncthreads();
        if (t == 0) {
            atomicAdd(counter, 1u);
            const unsigned want = (unsigned)(l + 1) * (unsigned)G;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > spin_limit) { dead = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (dead) { if (t == 0) atomicAdd(errors + 1, 1u); return; }
        if (MODE == 1 || MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int other = (me + 1 + 37 * l) % G;
        const float* theirs = buf + (size_t)other * tile_floats;
        for (int i = t; i < tile_floats; i += 256) {
            const float want = (float)(l * 1000 + other) + (float)i * (1.0f / 65536.0f);
            const float got = MODE == 0 ? __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : theirs[i];
            bad += got != want;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>
void run(int G, int tile_floats, int L, float* b0, float* b1, unsigned* counter, unsigned* errors) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0, ms0 = 0;
    unsigned herr[2] = {0, 0};
    for (int w = 0; w < 3; ++w) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(errors, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(layers<MODE>, dim3(G), dim3(256), 0, 0, b0, b1, tile_floats, L, counter, errors, 1u << 22);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(herr, errors, 8, hipMemcpyDeviceToHost));
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(layers<MODE>, dim3(G), dim3(256), 0, 0, b0, b1, tile_floats, 1, counter, errors, 1u << 22);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
    }
    printf("mode %d  G=%3d  tile %6d floats: %.2f us per layer (%d layers %.1f us, 1 layer %.1f us)  wrong=%u dead=%u\n", MODE, G,
           tile_floats, (ms - ms0) * 1e3 / (L - 1), L, ms * 1e3, ms0 * 1e3, herr[0], herr[1]);
}

int main() {
    float *b0, *b1; unsigned *counter, *errors;
    CK(hipMalloc(&b0, 64 << 20)); CK(hipMalloc(&b1, 64 << 20)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&errors, 8));
    const int L = 65;
    for (int G : {16, 64, 256})
        for (int n : {256, 4096, 16384}) {
            run<0>(G, n, L, b0, b1, counter, errors);
            run<1>(G, n, L, b0, b1, counter, errors);
            run<2>(G, n, L, b0, b1, counter, errors);
            run<3>(G, n, L, b0, b1, counter, errors);
        }
    return 0;
}
