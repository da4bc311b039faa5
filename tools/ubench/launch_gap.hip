// Micro-benchmark: what does a dependent kernel boundary cost on this runtime, and what does the "last workgroup does the
// tail" pattern cost instead?  A: producer kernel (G workgroups write a small tile each) + separate one-workgroup consumer
// kernel, back to back in one stream.  B: the same producer with the consumer's work folded in behind a device-scope fence +
// arrival counter.  Reports microseconds per (producer + consumer) pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void producer(float* out, int n_per_block, float v) {
    float* o = out + (size_t)blockIdx.x * n_per_block;
    for (int i = threadIdx.x; i < n_per_block; i += blockDim.x) o[i] = v + i;
}
__global__ void consumer(const float* in, int blocks, int n_per_block, float* res) {
    float a = 0.f;
    for (int b = threadIdx.x; b < blocks; b += blockDim.x) a += in[(size_t)b * n_per_block];
    __shared__ float s[256];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) res[0] = s[0];
}
__global__ void fused(float* out, int n_per_block, float v, unsigned* counter, float* res) {
    float* o = out + (size_t)blockIdx.x * n_per_block;
    for (int i = threadIdx.x; i < n_per_block; i += blockDim.x) o[i] = v + i;
    __shared__ int last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) *counter = 0;
    float a = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) a += __hip_atomic_load(out + (size_t)b * n_per_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ float s[256];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) res[0] = s[0];
}
int main() {
    float *buf, *res; unsigned* counter;
    CK(hipMalloc(&buf, 64 << 20)); CK(hipMalloc(&res, 64)); CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    for (int G : {8, 32, 128, 512}) for (int n : {256, 4096, 32768}) {
        if ((size_t)G * n * 4 > (64u << 20)) continue;
        float ms_a, ms_b;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) { producer<<<G, 256>>>(buf, n, (float)i); consumer<<<1, 256>>>(buf, G, n, res); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_a, e0, e1));
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) fused<<<G, 256>>>(buf, n, (float)i, counter, res);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_b, e0, e1));
        }
        printf("G=%4d workgroups x %6d floats: two launches %.2f us, one launch with last-workgroup tail %.2f us\n", G, n, ms_a / reps * 1e3, ms_b / reps * 1e3);
    }
    return 0;
}
