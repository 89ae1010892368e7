// nct_dev.cpp — device-pointer variants of the seams of transfer_color_single_bds (main.cu:204-316): the same launchers the fused per-pair path uses
// (nctk_*), on buffers that STAY in HBM between calls — what an integrator replacing single seams of the reference needs (its own code keeps Ndata_C1,
// ann_device, … on the device across these calls, main.cu:238-326). Host-pointer variants (nct_api.cpp) pay H2D + D2H + a synchronise per call.
// All calls are enqueued on the context's stream in call order and return without waiting; nct_dev_download / nct_synchronize wait.
// Layout: features channel-last (HWC) fp32 — nct_chw_to_hwc_dev converts a Caffe blob once; NNFs u32 (y << 12) | x; images u8 BGR HWC.
#include "nct_internal.h"

#define CTX_ENTER() do { if (!ctx) return NCT_ERR_INVALID; NCT_HIP(hipSetDevice(ctx->device)); } while (0)

extern "C" {

int nct_dev_alloc(nct_ctx* ctx, size_t bytes, void** out) {
    CTX_ENTER();
    NCT_REQUIRE(out && bytes > 0, "dev_alloc: bad arguments");
    *out = ctx->alloc(bytes);                       // context arena: cached blocks, reused in stream order
    return *out ? NCT_OK : NCT_ERR_HIP;
}
int nct_dev_free(nct_ctx* ctx, void* p) {
    CTX_ENTER();
    if (!p) return NCT_OK;
    bool mine = false;
    for (const auto& b : ctx->blocks) if (b.p == p && b.used) { mine = true; break; }
    NCT_REQUIRE(mine, "dev_free: %p is not a live nct_dev_alloc block of this context", p);      // a foreign or already freed pointer is an error, not a silent no-op
    ctx->release(p);
    return NCT_OK;
}
int nct_dev_upload(nct_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    CTX_ENTER();
    NCT_REQUIRE(dst_dev && src_host, "dev_upload: null pointer");
    NCT_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));     // the host buffer may be reused on return
    return NCT_OK;
}
int nct_dev_download(nct_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    CTX_ENTER();
    NCT_REQUIRE(dst_host && src_dev, "dev_download: null pointer");
    NCT_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    return NCT_OK;
}

int nct_chw_to_hwc_dev(nct_ctx* ctx, const float* src_chw, float* dst_hwc, int C, int H, int W) {
    CTX_ENTER();
    NCT_REQUIRE(src_chw && dst_hwc && C > 0 && H > 0 && W > 0, "chw_to_hwc_dev: bad arguments");
    return nctk_chw_to_hwc(ctx, ctx->stream, src_chw, dst_hwc, C, H * W);
}
int nct_hwc_to_chw_dev(nct_ctx* ctx, const float* src_hwc, float* dst_chw, int C, int H, int W) {
    CTX_ENTER();
    NCT_REQUIRE(src_hwc && dst_chw && C > 0 && H > 0 && W > 0, "hwc_to_chw_dev: bad arguments");
    return nctk_hwc_to_chw(ctx, ctx->stream, src_hwc, dst_chw, C, H * W);
}

// V2: Classifier::Predict on a device image; taps in Caffe's CHW layout (d_taps_chw[t] nullable)
int nct_vgg19_features_dev(nct_ctx* ctx, const uint8_t* d_bgr, int h, int w, int stride, int deepest_tap, float* const* d_taps_chw, int* dims) {
    CTX_ENTER();
    NCT_REQUIRE(d_bgr && h > 0 && w > 0 && stride >= 3 * w, "vgg19_features_dev: bad image arguments");
    return nctk_vgg19_forward(ctx, ctx->stream, d_bgr, h, w, stride, deepest_tap, d_taps_chw, dims);
}

int nct_vgg19_features_hwc_dev(nct_ctx* ctx, const uint8_t* d_bgr, int h, int w, int stride, int deepest_tap, float* const* d_taps_chw, float* const* d_taps_hwc, int* dims) {
    CTX_ENTER();
    NCT_REQUIRE(d_bgr && h > 0 && w > 0 && stride >= 3 * w, "vgg19_features_hwc_dev: bad image arguments");
    return nctk_vgg19_forward(ctx, ctx->stream, d_bgr, h, w, stride, deepest_tap, d_taps_chw, dims, d_taps_hwc);
}

// N1: norm (main.cu:265,274,313)
int nct_feat_normalize_dev(nct_ctx* ctx, const float* src_hwc, float* dst_hwc, float* resp, int C, int H, int W) {
    CTX_ENTER();
    NCT_REQUIRE(src_hwc && dst_hwc && C > 0 && (C & 3) == 0 && H > 0 && W > 0, "feat_normalize_dev: C must be a positive multiple of 4");
    return nctk_normalize(ctx, ctx->stream, src_hwc, dst_hwc, resp, C, H * W);
}

// N2: init_Ann_kernel / upSample_kernel (main.cu:232-250)
int nct_nnf_init_dev(nct_ctx* ctx, uint32_t* nnf, int ah, int aw, int bh, int bw) {
    CTX_ENTER();
    NCT_REQUIRE(nnf && ah >= 2 && aw >= 2 && bh >= 1 && bw >= 1 && ah < 4096 && aw < 4096 && bh < 4096 && bw < 4096, "nnf_init_dev: dims out of range");
    return nctk_nnf_init(ctx, ctx->stream, nnf, ah, aw, bh, bw);
}
int nct_nnf_upsample_dev(nct_ctx* ctx, const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half) {
    CTX_ENTER();
    NCT_REQUIRE(nnf_half && nnf && nnf_half != nnf && ah >= 1 && aw >= 1 && bh >= 1 && bw >= 1 && ah_half >= 1 && aw_half >= 1, "nnf_upsample_dev: bad arguments");
    NCT_REQUIRE(ah < 4096 && aw < 4096 && bh < 4096 && bw < 4096, "nnf_upsample_dev: dims out of range (the NNF word holds 12 bits per coordinate)");
    return nctk_nnf_upsample(ctx, ctx->stream, nnf_half, nnf, ah, aw, bh, bw, ah_half, aw_half);
}

// P1: patchmatch_single (main.cu:283-284). One field, or both fields of a level in the same launches (what the pipeline runs).
int nct_patchmatch_dev(nct_ctx* ctx, const float* a_hwc, const float* b_hwc, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max,
                       uint32_t seed, uint32_t* nnf, float* dist) {
    CTX_ENTER();
    NCT_REQUIRE(a_hwc && b_hwc && nnf && dist, "patchmatch_dev: null pointer");
    NCT_REQUIRE(patch == 3, "patchmatch_dev: patch must be 3 (Config.h:70), got %d", patch);
    return nctk_patchmatch(ctx, ctx->stream, a_hwc, b_hwc, C, ah, aw, bh, bw, iters, rs_max, seed, nnf, dist, nullptr);
}
int nct_patchmatch_bidir_dev(nct_ctx* ctx, const float* a_hwc, const float* b_hwc, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max,
                             uint32_t seed_ab, uint32_t seed_ba, uint32_t* ann, float* annd, uint32_t* bnn, float* bnnd, int unit_norm) {
    CTX_ENTER();
    NCT_REQUIRE(a_hwc && b_hwc && ann && annd && bnn && bnnd, "patchmatch_bidir_dev: null pointer");
    NCT_REQUIRE(patch == 3, "patchmatch_bidir_dev: patch must be 3 (Config.h:70), got %d", patch);
    return nctk_patchmatch_bidir(ctx, ctx->stream, a_hwc, b_hwc, nullptr, nullptr, C, ah, aw, bh, bw, iters, rs_max, seed_ab, seed_ba, ann, annd, bnn, bnnd,
                                 unit_norm ? NCT_PM_ROWREJECT : NCT_PM_PLAIN, nullptr);
}

// B2 / B1: avg_vote_bds_a/_b/avg_vote_bds, feature_distance, reconstruct_bds (main.cu:291-316)
int nct_bds_vote_features_dev(nct_ctx* ctx, const uint32_t* ann, const uint32_t* bnn, const float* pin_hwc, float* pout_hwc, float* pw, int C, int ah, int aw, int bh, int bw,
                              int patch, float w_coherence, float w_complete) {
    CTX_ENTER();
    NCT_REQUIRE(ann && bnn && pin_hwc && pout_hwc, "bds_vote_features_dev: null pointer");
    NCT_REQUIRE(patch == 3, "bds_vote_features_dev: patch must be 3, got %d", patch);
    return nctk_bds_vote_features(ctx, ctx->stream, ann, bnn, pin_hwc, pout_hwc, pw, C, ah, aw, bh, bw, w_coherence, w_complete);
}
int nct_bds_vote_image_dev(nct_ctx* ctx, const uint8_t* b_bgr, const uint32_t* ann, const uint32_t* bnn, int ah, int aw, int bh, int bw, int patch,
                           double w_coherence, double w_complete, uint8_t* out_bgr) {
    CTX_ENTER();
    NCT_REQUIRE(b_bgr && ann && bnn && out_bgr, "bds_vote_image_dev: null pointer");
    NCT_REQUIRE(patch == 3, "bds_vote_image_dev: patch must be 3, got %d", patch);
    return nctk_bds_vote_image(ctx, ctx->stream, b_bgr, ann, bnn, ah, aw, bh, bw, w_coherence, w_complete, out_bgr);
}
int nct_feature_distance_dev(nct_ctx* ctx, const float* a_hwc, const float* b_hwc, float* err, int C, int H, int W) {
    CTX_ENTER();
    NCT_REQUIRE(a_hwc && b_hwc && err && C > 0 && (C & 3) == 0 && H > 0 && W > 0, "feature_distance_dev: bad arguments");
    return nctk_feature_distance(ctx, ctx->stream, a_hwc, b_hwc, err, C, H * W);
}

}  // extern "C"
