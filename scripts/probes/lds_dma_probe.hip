// Compile-only probe (round 6): does hipcc keep an LDS-direct prefetch (global_load_lds_dwordx4 into a second static __shared__ array) in flight across LDS reads of the first array
// and across a workgroup barrier?  hipcc --offload-arch=gfx950 -O3 -c lds_dma_probe.hip -save-temps; grep "global_load_lds\|s_waitcnt\|s_barrier" *.s
// Findings: reads of the other array carry no vmcnt wait (alias-aware LDS-DMA tracking); __syncthreads() and __builtin_amdgcn_fence(.., "workgroup", "local") both wait vmcnt(0);
// an asm barrier "s_waitcnt lgkmcnt(0); s_barrier" with a memory clobber does not. DESIGN.md 9.
#include <hip/hip_runtime.h>
__shared__ float4 buf0[1600];
__shared__ float4 buf1[1600];
__shared__ float s_x[256];
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(256) void k(const float4* __restrict__ a, float4* __restrict__ out, int n) {
    const int t = threadIdx.x;
    for (int i = 0; i < 6; ++i) {
        const float4* src = a + (size_t)(blockIdx.x * 1600 + i * 256 + t);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(buf0 + i * 256 + (t & ~63)), 16, 0, 0);
    }
    __syncthreads();
    for (int i = 0; i < 6; ++i) {
        const float4* src = a + (size_t)((blockIdx.x + 1) * 1600 + i * 256 + t);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(buf1 + i * 256 + (t & ~63)), 16, 0, 0);
    }
    s_x[t] = (float)t * 0.5f;
    lds_barrier();
    float4 acc = make_float4(s_x[(t + 1) & 255], 0, 0, 0);
    for (int r = 0; r < n; ++r) {
        const float4 v = buf0[(t * 7 + r * 13) % 1600];
        acc.x += v.x * v.y; acc.y += v.z; acc.z += v.w; acc.w += v.x;
    }
    __syncthreads();
    const float4 w = buf1[(t * 5) % 1600];
    out[blockIdx.x * 256 + t] = make_float4(acc.x + w.x, acc.y + w.y, acc.z + w.z, acc.w + w.w);
}
