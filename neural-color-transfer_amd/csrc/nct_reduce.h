// nct_reduce.h — deterministic reductions of fp64 values shared by the colour-stage kernels (k_colorsolve.hip, k_s1.hip). Device code only.
#pragma once
#include <hip/hip_runtime.h>

// ---------------------------------------------------------------- deterministic block reduction of NQ doubles
// Fixed 256-wide tree s[t] += s[t + off], off = 128 … 1 (the order the oracle mirrors), evaluated with two barriers instead of nine: the two cross-wave
// steps go through LDS, the six steps inside the first wave are lane shifts (a lane t < off adds the value lane t + off held BEFORE the step, exactly as
// the array form does; what lanes >= off compute is never used).
template <int NQ>
__device__ __forceinline__ void tree256(double (&v)[NQ], double* __restrict__ s_red /*[128 * NQ]*/) {
    const int t = threadIdx.x;
    if (t >= 128) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_red[q * 128 + t - 128] = v[q];
    }
    __syncthreads();
    if (t < 128) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] += s_red[q * 128 + t];            // off = 128
    }
    __syncthreads();
    if (t >= 64 && t < 128) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_red[q * 128 + t - 64] = v[q];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double x = v[q] + s_red[q * 128 + t];                          // off = 64
            x += __shfl_down(x, 32); x += __shfl_down(x, 16); x += __shfl_down(x, 8);
            x += __shfl_down(x, 4); x += __shfl_down(x, 2); x += __shfl_down(x, 1);
            v[q] = x;                                                      // lane 0 holds the sum
        }
    }
}
template <int NQ>
__device__ __forceinline__ void block_reduce_store(double (&v)[NQ], double* __restrict__ partial /*[nblocks][NQ]*/, int slot = -1 /* logical block (default: blockIdx.x) */) {
    __shared__ double s_red[128 * NQ];
    tree256<NQ>(v, s_red);
    if (threadIdx.x == 0) {
        const size_t b = slot >= 0 ? (size_t)slot : (size_t)blockIdx.x;
#pragma unroll
        for (int q = 0; q < NQ; ++q) partial[b * NQ + q] = v[q];
    }
}
// sum partial[nb][NQ] in a fixed order (single block of 256 threads): thread t adds its partials b = t, t + 256, … in ascending order, then the same tree
template <int NQ>
__device__ __forceinline__ void final_reduce(const double* __restrict__ partial, int nb, double (&out)[NQ]) {
    __shared__ double s_fin[128 * NQ];
    __shared__ double s_out[NQ];
    const int t = threadIdx.x;
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
    for (int b = t; b < nb; b += 256)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] += partial[(size_t)b * NQ + q];
    tree256<NQ>(acc, s_fin);
    if (t == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_out[q] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = s_out[q];
    __syncthreads();
}

