// k_nnf.hip — NNF initialisation (scaled identity) and nearest-neighbour upsampling with offset scaling (N2).
// Reference: init_Ann_kernel GeneralizedPatchMatch.cu:527-544; upSample_kernel :546-580 (+ the temp buffer and
// D2D copy of main.cu:238-250, which disappear here: the upsample writes straight into the destination).
// Trivial bandwidth kernels: 4 B written per query.
#include "nct_internal.h"
#include "nct_device.h"

__global__ void k_nnf_init(uint32_t* __restrict__ nnf, int ah, int aw, int bh, int bw) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ah * aw) return;
    int ay = i / aw, ax = i - ay * aw;
    // same float expression as the reference: float(ax) / float(aw-1) * (bw-1)
    int bx = min((int)((float)ax / (float)(aw - 1) * (float)(bw - 1)), bw - 1);
    int by = min((int)((float)ay / (float)(ah - 1) * (float)(bh - 1)), bh - 1);
    nnf[i] = xy_pack(bx, by);
}

__global__ void k_nnf_upsample(const uint32_t* __restrict__ half, uint32_t* __restrict__ nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ah * aw) return;
    int ay = i / aw, ax = i - ay * aw;
    float aw_ratio = (float)aw / (float)aw_half;
    float ah_ratio = (float)ah / (float)ah_half;
    // (ax+0.5)/ratio is evaluated in double in the reference (int + double literal)
    int ax_half = (int)(((double)ax + 0.5) / (double)aw_ratio);
    int ay_half = (int)(((double)ay + 0.5) / (double)ah_ratio);
    ax_half = clampi(ax_half, 0, aw_half - 1);
    ay_half = clampi(ay_half, 0, ah_half - 1);
    uint32_t v = half[ay_half * aw_half + ax_half];
    int bx_half = nnf_x(v), by_half = nnf_y(v);
    // ax + (bx_half-ax_half)*ratio is float arithmetic, the trailing +0.5 promotes to double
    float fx = (float)ax + (float)(bx_half - ax_half) * aw_ratio;
    float fy = (float)ay + (float)(by_half - ay_half) * ah_ratio;
    int bx = (int)((double)fx + 0.5);
    int by = (int)((double)fy + 0.5);
    bx = clampi(bx, 0, bw - 1);
    by = clampi(by, 0, bh - 1);
    nnf[i] = xy_pack(bx, by);
}

int nctk_nnf_init(nct_ctx* ctx, hipStream_t s, uint32_t* nnf, int ah, int aw, int bh, int bw) {
    NCT_REQUIRE(ah >= 2 && aw >= 2 && bh >= 1 && bw >= 1 && ah < 4096 && aw < 4096 && bh < 4096 && bw < 4096,
                "nnf_init: dims out of range (%dx%d -> %dx%d); NNF coordinates are 12-bit", ah, aw, bh, bw);
    hipLaunchKernelGGL(k_nnf_init, dim3(cdiv(ah * aw, 256)), dim3(256), 0, s, nnf, ah, aw, bh, bw);
    NCT_LAUNCH_CHECK();
    return 0;
}

int nctk_nnf_upsample(nct_ctx* ctx, hipStream_t s, const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half) {
    NCT_REQUIRE(ah >= 1 && aw >= 1 && ah_half >= 1 && aw_half >= 1 && ah < 4096 && aw < 4096 && bh < 4096 && bw < 4096,
                "nnf_upsample: dims out of range");
    hipLaunchKernelGGL(k_nnf_upsample, dim3(cdiv(ah * aw, 256)), dim3(256), 0, s, nnf_half, nnf, ah, aw, bh, bw, ah_half, aw_half);
    NCT_LAUNCH_CHECK();
    return 0;
}
