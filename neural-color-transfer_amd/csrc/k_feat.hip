// k_feat.hip — layout transposes, per-pixel L2 normalisation (N1) and feature_distance (B2 tail).
// Reference: `norm` GeneralizedPatchMatch.cu:237-283 (7 library launches + 2 D2H copies fused into one kernel +
// an optional tiny min/max pass); `feature_distance` :833-855.
// Roofline: HBM streaming (read C*4 B + write C*4 B per pixel). One 16-lane DPP row per pixel, float4 per lane.
#include "nct_internal.h"
#include "nct_device.h"
#include <hip/hip_fp16.h>

// ---------------------------------------------------------------- CHW <-> HWC
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    // src is [rows][cols] row-major, dst is [cols][rows]
    __shared__ float tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = src[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = bx + j, r = by + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][j];
    }
}

int nctk_chw_to_hwc(nct_ctx* ctx, hipStream_t s, const float* src, float* dst, int C, int HW) {
    dim3 grid(cdiv(HW, 32), cdiv(C, 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, src, dst, C, HW);
    NCT_LAUNCH_CHECK();
    return 0;
}
int nctk_hwc_to_chw(nct_ctx* ctx, hipStream_t s, const float* src, float* dst, int C, int HW) {
    dim3 grid(cdiv(C, 32), cdiv(HW, 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, src, dst, HW, C);
    NCT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- N1 normalise
// Summation order (must match oracle/orc_nnf.c): lane v of the 16-lane row owns float4 chunks v, v+16, …;
// one fmaf chain per lane; 16-lane butterfly.
// dst_h (nullable): the same values rounded to fp16 (round-to-nearest-even), HWC — the candidate tiles of the opt-in reduced-precision mode
// (NCT_FLAG_FEAT16, k_patchmatch.hip NCT_PM_FP16: fp32 accumulate, results differ from the fp32 path; there is no fp16 prefilter in the product).
__global__ __launch_bounds__(256) void k_normalize(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dis_out,
                                                   int C, int HW, uint2* __restrict__ dst_h) {
    int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    int v = threadIdx.x & 15;
    bool live = pix < HW;
    int p = live ? pix : HW - 1;
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)p * C);
    int nchunk = C >> 2;
    float acc = 0.f;
    for (int j = v; j < nchunk; j += 16) { float4 x = s4[j]; acc = dot4_acc(x, x, acc); }
    float d = sqrtf(row16_sum(acc));
    if (!live) return;
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)p * C);
    for (int j = v; j < nchunk; j += 16) {
        float4 x = s4[j];
        const float4 y = make_float4(x.x / d, x.y / d, x.z / d, x.w / d);
        d4[j] = y;
        if (dst_h) {
            const __half2 lo = __floats2half2_rn(y.x, y.y), hi = __floats2half2_rn(y.z, y.w);
            uint2 pk; pk.x = *reinterpret_cast<const unsigned*>(&lo); pk.y = *reinterpret_cast<const unsigned*>(&hi);
            dst_h[(size_t)p * nchunk + j] = pk;
        }
    }
    if (dis_out && v == 0) dis_out[p] = d;
}

__global__ void k_minmax(const float* __restrict__ dis, int n, unsigned int* __restrict__ mm) {
    // dis >= 0, so IEEE bit patterns order like unsigned ints; min/max are order independent => exact
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned int b = __float_as_uint(dis[i]);
        lo = b < lo ? b : lo; hi = b > hi ? b : hi;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned int l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}

__global__ void k_response(const float* __restrict__ dis, float* __restrict__ resp, int n, const unsigned int* __restrict__ mm) {
    float mn = __uint_as_float(mm[0]), mx = __uint_as_float(mm[1]);
    float sc = 1.0f / (mx - mn);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) resp[i] = (dis[i] + (-mn)) * sc;
}

int nctk_normalize(nct_ctx* ctx, hipStream_t s, const float* src_hwc, float* dst_hwc, float* resp, int C, int HW, void* dst_h16) {
    NCT_REQUIRE(C > 0 && (C & 3) == 0, "normalize: C=%d must be a positive multiple of 4", C);
    if (!resp) {
        hipLaunchKernelGGL(k_normalize, dim3(cdiv(HW, 16)), dim3(256), 0, s, src_hwc, dst_hwc, (float*)nullptr, C, HW, (uint2*)dst_h16);
        NCT_LAUNCH_CHECK();
        return 0;
    }
    DevBuf<float> dis(ctx, HW);
    DevBuf<unsigned int> mm(ctx, 2);
    if (!dis.ok() || !mm.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemsetD32Async((hipDeviceptr_t)(unsigned int*)mm, (int)0xFFFFFFFFu, 1, s));
    NCT_HIP(hipMemsetD32Async((hipDeviceptr_t)((unsigned int*)mm + 1), 0, 1, s));
    hipLaunchKernelGGL(k_normalize, dim3(cdiv(HW, 16)), dim3(256), 0, s, src_hwc, dst_hwc, (float*)dis, C, HW, (uint2*)dst_h16);
    NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax, dim3(256), dim3(256), 0, s, (const float*)dis, HW, (unsigned int*)mm);
    NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_response, dim3(cdiv(HW, 256)), dim3(256), 0, s, (const float*)dis, resp, HW, (const unsigned int*)mm);
    NCT_LAUNCH_CHECK();
    return 0;                            // dis/mm return to the arena (recycled in stream order)
}

// ---------------------------------------------------------------- feature_distance
__global__ __launch_bounds__(256) void k_feature_distance(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ err, int C, int HW) {
    int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    int v = threadIdx.x & 15;
    bool live = pix < HW;
    int p = live ? pix : HW - 1;
    const float4* a4 = reinterpret_cast<const float4*>(a + (size_t)p * C);
    const float4* b4 = reinterpret_cast<const float4*>(b + (size_t)p * C);
    int nchunk = C >> 2;
    float acc = 0.f;
    for (int j = v; j < nchunk; j += 16) acc = dot4_acc(a4[j], b4[j], acc);
    float sum = row16_sum(acc);
    if (live && v == 0) err[p] = -sum;
}

int nctk_feature_distance(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, float* err, int C, int HW) {
    NCT_REQUIRE(C > 0 && (C & 3) == 0, "feature_distance: C=%d must be a positive multiple of 4", C);
    hipLaunchKernelGGL(k_feature_distance, dim3(cdiv(HW, 16)), dim3(256), 0, s, a_hwc, b_hwc, err, C, HW);
    NCT_LAUNCH_CHECK();
    return 0;
}
