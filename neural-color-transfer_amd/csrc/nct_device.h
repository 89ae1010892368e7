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

// The same sum with EIGHT lanes per group, two of the sixteen partial sums per lane: the lane that holds partials j and j + 8 (j < 8) adds
// them itself (= the row_ror:8 step: p[j] + p[j+8]); the remaining steps pair j with j^4, then j^2, then j^1, which are lane^1 and lane^2
// (quad permutes) and "the other quad" (row_half_mirror: lane^7 — all four lanes of a quad hold the same bits by then, IEEE addition
// being commutative) when lane = pm_lane8(j) below. Same binary tree, same result bit for bit.
__device__ __forceinline__ float half8_sum(float p_lo, float p_hi) {
    float x = p_lo + p_hi;
    x = x + dpp_rot<0xB1>(x);    // quad_perm:[1,0,3,2]
    x = x + dpp_rot<0x4E>(x);    // quad_perm:[2,3,0,1]
    x = x + dpp_rot<0x141>(x);   // row_half_mirror
    return x;
}
// partial-sum index j (0..7) served by lane l (0..7) of an 8-lane group: bit 2 of j <-> bit 0 of l, bit 1 <-> bit 1, bit 0 <-> bit 2
__device__ __forceinline__ int pm_chunk8(int l) { return ((l & 1) << 2) | (l & 2) | ((l >> 2) & 1); }

__device__ __forceinline__ float dot4_acc(const float4 a, const float4 b, float acc) {
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    acc = __builtin_fmaf(a.w, b.w, acc);
    return acc;
}
