// nct_color.cpp — C-ABI entry points of the colour stage (host-pointer variants used by tests and by integrators that
// want to replace a single seam of transfer_color_single_bds; the fused per-pair path lives in nct_pipeline.cpp).
#include "nct_internal.h"
#include <cstring>

#define CTX_ENTER() do { if (!ctx) return NCT_ERR_INVALID; NCT_HIP(hipSetDevice(ctx->device)); } while (0)
#define H2D(dst, src, bytes) NCT_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, ctx->stream))
#define D2H(dst, src, bytes) NCT_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, ctx->stream))
#define SYNC() NCT_HIP(hipStreamSynchronize(ctx->stream))
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

extern "C" {

void nct_params_default(nct_params* p) {
    if (!p) return;
    // Config::Config() (ColorTransfer/Config.h:58-72) — NOT the values quoted in the help strings (main.cu:40-43)
    p->bds_weight = 2.0; p->eps = 0.60; p->nonlocal_weight = 2.0; p->local_weight = 0.125; p->wls_lambda_init = 0.024;
    p->cluster_num = 10; p->k_num = 8; p->patch_size = 3; p->wls_alpha = 1.2;
    p->pm_iters = 10; p->seed = 1;
    p->levels = 5; p->flags = 0;
}

int nct_bgr2lab_u8(nct_ctx* ctx, const uint8_t* bgr, size_t npix, uint8_t* lab) {
    CTX_ENTER();
    NCT_REQUIRE(bgr && lab && npix > 0, "bgr2lab: bad arguments");
    DevBuf<uint8_t> a(ctx, npix * 3), b(ctx, npix * 3);
    if (!a.ok() || !b.ok()) return NCT_ERR_HIP;
    H2D(a, bgr, npix * 3);
    RC(nctk_bgr2lab(ctx, ctx->stream, a, b, npix));
    D2H(lab, b, npix * 3); SYNC();
    return NCT_OK;
}
int nct_lab2bgr_u8(nct_ctx* ctx, const uint8_t* lab, size_t npix, uint8_t* bgr) { return nct_lab2bgr_u8_form(ctx, lab, npix, bgr, NCT_LAB2BGR_PIECEWISE); }
int nct_lab2bgr_u8_form(nct_ctx* ctx, const uint8_t* lab, size_t npix, uint8_t* bgr, int form) {
    CTX_ENTER();
    NCT_REQUIRE(bgr && lab && npix > 0, "lab2bgr: bad arguments");
    NCT_REQUIRE(form == NCT_LAB2BGR_CUBE || form == NCT_LAB2BGR_PIECEWISE, "lab2bgr: unknown form %d", form);
    DevBuf<uint8_t> a(ctx, npix * 3), b(ctx, npix * 3);
    if (!a.ok() || !b.ok()) return NCT_ERR_HIP;
    H2D(a, lab, npix * 3);
    RC(nctk_lab2bgr(ctx, ctx->stream, a, b, npix, form));
    D2H(bgr, b, npix * 3); SYNC();
    return NCT_OK;
}
int nct_resize_u8c3(nct_ctx* ctx, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    CTX_ENTER();
    NCT_REQUIRE(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, "resize_u8c3: bad arguments");
    DevBuf<uint8_t> a(ctx, (size_t)sh * sw * 3), b(ctx, (size_t)dh * dw * 3);
    if (!a.ok() || !b.ok()) return NCT_ERR_HIP;
    H2D(a, src, (size_t)sh * sw * 3);
    RC(nctk_resize_u8c3(ctx, ctx->stream, a, sh, sw, b, dh, dw));
    D2H(dst, b, (size_t)dh * dw * 3); SYNC();
    return NCT_OK;
}
int nct_resize_f64c3(nct_ctx* ctx, const double* src, int sh, int sw, double* dst, int dh, int dw) {
    CTX_ENTER();
    NCT_REQUIRE(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, "resize_f64c3: bad arguments");
    DevBuf<double> a(ctx, (size_t)sh * sw * 3), b(ctx, (size_t)dh * dw * 3);
    if (!a.ok() || !b.ok()) return NCT_ERR_HIP;
    H2D(a, src, sizeof(double) * sh * sw * 3);
    RC(nctk_resize_f64c3(ctx, ctx->stream, a, sh, sw, b, dh, dw));
    D2H(dst, b, sizeof(double) * dh * dw * 3); SYNC();
    return NCT_OK;
}

int nct_cluster_features(nct_ctx* ctx, const float* feat_chw, int C, int h, int w, int K, int iters, uint64_t seed, int* labels, int* nlabels) {
    CTX_ENTER();
    NCT_REQUIRE(feat_chw && labels && nlabels && h > 0 && w > 0 && (C & 3) == 0, "cluster_features: bad arguments");
    const int n = h * w;
    DevBuf<float> t(ctx, (size_t)C * n), f(ctx, (size_t)C * n), fn(ctx, (size_t)C * n);
    DevBuf<int> lab(ctx, n), nl(ctx, 1);
    if (!t.ok() || !f.ok() || !fn.ok() || !lab.ok() || !nl.ok()) return NCT_ERR_HIP;
    H2D(t, feat_chw, sizeof(float) * C * n);
    RC(nctk_chw_to_hwc(ctx, ctx->stream, t, f, C, n));
    RC(nctk_normalize(ctx, ctx->stream, f, fn, nullptr, C, n));          // main.cu:139-165 (per-pixel L2 normalise, HWC)
    RC(nctk_kmeans_labels(ctx, ctx->stream, fn, n, C, K, iters, seed, lab, nl));
    D2H(labels, lab, sizeof(int) * n); D2H(nlabels, nl, sizeof(int)); SYNC();
    return NCT_OK;
}

int nct_knn_graph(nct_ctx* ctx, const uint8_t* lab_u8, int h, int w, const int* labels, int lh, int lw, int nlabels, int samples, int k,
                  int* knn_id, double* knn_w) {
    CTX_ENTER();
    NCT_REQUIRE(lab_u8 && labels && knn_id && knn_w && h > 0 && w > 0 && lh > 0 && lw > 0 && samples > 0, "knn_graph: bad arguments");
    NCT_REQUIRE(k == 8, "knn_graph: only k=8 is supported (Config.h:68), got %d", k);
    const int n = h * w;
    DevBuf<uint8_t> l(ctx, (size_t)n * 3);
    DevBuf<int> lb(ctx, (size_t)lh * lw), id(ctx, (size_t)n * 8);
    DevBuf<double> kw(ctx, (size_t)n * 8);
    if (!l.ok() || !lb.ok() || !id.ok() || !kw.ok()) return NCT_ERR_HIP;
    H2D(l, lab_u8, (size_t)n * 3); H2D(lb, labels, sizeof(int) * lh * lw);
    RC(nctk_knn_graph(ctx, ctx->stream, l, h, w, lb, lh, lw, nlabels, nullptr, samples, id, kw));
    D2H(knn_id, id, sizeof(int) * n * 8); D2H(knn_w, kw, sizeof(double) * n * 8); SYNC();
    return NCT_OK;
}

int nct_local_color_transfer(nct_ctx* ctx, const float* err, const uint8_t* s_bgr_level, const uint8_t* g_bgr_level, const uint8_t* s_bgr_full,
                             const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W, const nct_params* prm,
                             uint8_t* out_bgr_full, nct_color_stages* stages) {
    CTX_ENTER();
    NCT_REQUIRE(err && s_bgr_level && g_bgr_level && s_bgr_full && knn_id && knn_w && prm && out_bgr_full, "local_color_transfer: null pointer");
    NCT_REQUIRE(h > 0 && w > 0 && H >= h && W >= w && layer >= 0 && layer <= 4, "local_color_transfer: bad geometry");
    const size_t n = (size_t)h * w, N = (size_t)H * W;
    DevBuf<float> derr(ctx, n);
    DevBuf<uint8_t> sl(ctx, n * 3), gl(ctx, n * 3), sf(ctx, N * 3), slab(ctx, n * 3), glab(ctx, n * 3), sflab(ctx, N * 3), olab(ctx, N * 3), obgr(ctx, N * 3);
    DevBuf<int> id(ctx, n * 8);
    DevBuf<double> kw(ctx, n * 8);
    if (!derr.ok() || !sl.ok() || !gl.ok() || !sf.ok() || !slab.ok() || !glab.ok() || !sflab.ok() || !olab.ok() || !obgr.ok() || !id.ok() || !kw.ok()) return NCT_ERR_HIP;
    H2D(derr, err, sizeof(float) * n); H2D(sl, s_bgr_level, n * 3); H2D(gl, g_bgr_level, n * 3); H2D(sf, s_bgr_full, N * 3);
    H2D(id, knn_id, sizeof(int) * n * 8); H2D(kw, knn_w, sizeof(double) * n * 8);
    RC(nctk_bgr2lab(ctx, ctx->stream, sl, slab, n));          // main.cu:351-352
    RC(nctk_bgr2lab(ctx, ctx->stream, gl, glab, n));          // main.cu:370-371
    RC(nctk_bgr2lab(ctx, ctx->stream, sf, sflab, N));         // ColorTransfer.h:58
    nct_color_params cp{prm->eps, prm->nonlocal_weight, prm->local_weight, prm->wls_lambda_init, prm->wls_alpha, (double)prm->k_num};
    nct_color_debug dbg{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (stages) { dbg.ab_local = stages->ab_local; dbg.ab_nonlocal = stages->ab_nonlocal; dbg.ab_up = stages->ab_up; dbg.rough = stages->roughness;
                  dbg.ab_wls = stages->ab_wls; dbg.cg_iters = stages->cg_iters; dbg.wls_iters = stages->wls_iters; }
    RC(nctk_local_color_transfer(ctx, ctx->stream, derr, slab, glab, sflab, id, kw, layer, h, w, H, W, cp, olab, stages ? &dbg : nullptr));
    RC(nctk_lab2bgr(ctx, ctx->stream, olab, obgr, N, (prm->flags & NCT_FLAG_LAB2BGR_CUBE) ? 1 : 0));        // ColorTransfer.cpp:1469
    D2H(out_bgr_full, obgr, N * 3); SYNC();
    return NCT_OK;
}

}  // extern "C"
