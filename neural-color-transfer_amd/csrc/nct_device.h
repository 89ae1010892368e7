// nct_device.h — device-side primitives shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// NNF element packing (reference: GeneralizedPatchMatch.cu:24-34)
__device__ __forceinline__ uint32_t xy_pack(int x, int y) { return ((uint32_t)y << 12) | (uint32_t)x; }
__device__ __forceinline__ int nnf_x(uint32_t v) { return (int)(v & 0xFFFu); }
__device__ __forceinline__ int nnf_y(uint32_t v) { return (int)((v >> 12) & 0xFFFu); }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Counter-based RNG (replaces the reference's per-column cuRAND stream, GeneralizedPatchMatch.cu:54-66).
// u in (0,1] like curand_uniform; keyed by (seed, query pixel, iteration, search step, axis).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ float rand_u01(uint32_t seed, int ax, int ay, int iter, int step, int axis) {
    uint32_t ctr = (uint32_t)(1 + iter * 64 + step * 2 + axis);
    uint32_t h = mix32(seed ^ mix32((uint32_t)(ay * 4096 + ax) + 0x9E3779B9u * ctr));
    return (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
}

// 16-lane sum within one DPP row (a 16-lane group = one query / one pixel). Rotate-and-add with
// row_ror 8,4,2,1: every lane ends with the same value, bitwise equal to an xor butterfly because fp add is
// commutative and each stage's partial sums are periodic in the rotated distance.
template <int CTRL> __device__ __forceinline__ float dpp_rot(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float x) {
    x = x + dpp_rot<0x128>(x);   // row_ror:8
    x = x + dpp_rot<0x124>(x);   // row_ror:4
    x = x + dpp_rot<0x122>(x);   // row_ror:2
    x = x + dpp_rot<0x121>(x);   // row_ror:1
    return x;
}

__device__ __forceinline__ float dot4_acc(const float4 a, const float4 b, float acc) {
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    acc = __builtin_fmaf(a.w, b.w, acc);
    return acc;
}
