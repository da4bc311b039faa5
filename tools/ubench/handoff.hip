// Micro-benchmark (tuning aid, not part of the library): round-trip latency of an 8-byte granule hand-off
// between two workgroups on the SAME XCD vs DIFFERENT XCDs, for several store/load flavours.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
using u64 = unsigned long long;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
template <int ST> __device__ __forceinline__ void st(u64* p, u64 v) {
    if (ST == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store sc1
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           // plain store
}
template <int LD> __device__ __forceinline__ u64 ld(u64* p) {
    if (LD == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // global_load sc1
    if (LD == 1) return __builtin_nontemporal_load(p);                                       // nt
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);             // plain load
}

template <int ST, int LD>
__global__ void pingpong(u64* g, unsigned* table, unsigned* arrived, int want_same, int iters, u64* out) {
    if (threadIdx.x != 0) return;
    const unsigned me = blockIdx.x, n = gridDim.x;
    table[me] = xcc_id() + 1;
    __threadfence();
    atomicAdd(arrived, 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
        if (++spins > (1u << 24)) { if (me == 0) out[0] = ~0ull; return; }
    }
    const unsigned x0 = __hip_atomic_load(&table[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned partner = 0;
    for (unsigned b = 1; b < n; ++b) {
        const unsigned xb = __hip_atomic_load(&table[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((xb == x0) == (want_same != 0)) { partner = b; break; }
    }
    if (partner == 0) { if (me == 0) out[0] = ~0ull - 1; return; }
    if (me != 0 && me != partner) return;
    u64 bad = 0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= iters; ++i) {
        if (me == 0) {
            st<ST>(g, (u64)i);
            unsigned s = 0;
            while (ld<LD>(g + 16) != (u64)i) { if (++s > (1u << 22)) { bad = 1; break; } }
        } else {
            unsigned s = 0;
            while (ld<LD>(g) != (u64)i) { if (++s > (1u << 22)) { bad = 1; break; } }
            st<ST>(g + 16, (u64)i);
        }
        if (bad) break;
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (me == 0) { out[0] = (t1 - t0) / (u64)iters; out[1] = bad; out[2] = x0; out[3] = partner; }
}

extern "C" int run_pingpong(int st_mode, int ld_mode, int want_same, int iters, u64* result4) {
    u64 *g, *out; unsigned *table, *arrived;
    hipMalloc(&g, 4096); hipMalloc(&out, 64); hipMalloc(&table, 4096); hipMalloc(&arrived, 64);
    hipMemset(g, 0, 4096); hipMemset(out, 0, 64); hipMemset(table, 0, 4096); hipMemset(arrived, 0, 64);
    #define CASE(S, L) if (st_mode == S && ld_mode == L) hipLaunchKernelGGL((pingpong<S, L>), dim3(64), dim3(64), 0, 0, g, table, arrived, want_same, iters, out);
    CASE(0,0) CASE(0,1) CASE(0,2) CASE(1,0) CASE(1,1) CASE(1,2)
    hipDeviceSynchronize();
    hipMemcpy(result4, out, 32, hipMemcpyDeviceToHost);
    hipFree(g); hipFree(out); hipFree(table); hipFree(arrived);
    return 0;
}
