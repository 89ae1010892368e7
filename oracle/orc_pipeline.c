/* oracle/orc_pipeline.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the per-pair hot loop, transfer_color_single_bds (main.cu:47-454), composed from the per-stage
 * restatements in this directory. Same order of operations as the reference:
 *   features of S (conv5_1) and R (all taps)                  main.cu:94,102
 *   image pyramids by progressive bilinear resize             main.cu:104-108
 *   k-means of S's normalised conv5_1                         main.cu:139-168
 *   per level: NNF init/upsample, normalise, PatchMatch x2, image BDS vote, feature BDS vote, normalise,
 *   matching error, Lab, kNN graph, local colour transfer, re-predict      main.cu:179-428
 * Divergences are those of the stage files (schedule/RNG of PatchMatch, vote order, k-means seed, canonical solver order)
 * plus: the re-predict stops at the tap the next level needs (result-identical).
 */
#include "orc_common.h"
#include <stdio.h>

void orc_vgg19_features(const uint8_t* bgr, int H, int W, const float* const* weights, const float* const* biases, int deepest_tap, float* const* taps, int* dims);
void orc_feat_normalize(const float* src_chw, float* dst_chw, float* resp, int C, int H, int W);
void orc_nnf_init(uint32_t* nnf, int ah, int aw, int bh, int bw);
void orc_nnf_upsample(const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half);
void orc_patchmatch(const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max, uint32_t seed, uint32_t* nnf, float* dist);
int orc_patchmatch_inplace(const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max, uint32_t seed, int schedule, uint32_t* nnf, float* dist);
int orc_get_pm_schedule(void);      /* orc_nnf_inplace.c: 0 = the product's Jacobi schedule (default), 1 / 2 = the reference's in-place schedule under two legal interleavings */
void orc_feature_distance(const float* a_chw, const float* b_chw, float* err, int C, int H, int W);
void orc_bds_vote_features(const uint32_t* ann, const uint32_t* bnn, const float* pin, float* pout, float* pw_out, int C, int ah, int aw, int bh, int bw, int patch, float wCohen, float wComplete);
void orc_bds_vote_image(const uint8_t* a, int ah, int aw, const uint8_t* b, int bh, int bw, const uint32_t* ann, const uint32_t* bnn, int patch, double wCohen, double wComplete, uint8_t* out);
void orc_resize_u8c3(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
void orc_bgr2lab_u8(const uint8_t* src, size_t npix, uint8_t* dst);
void orc_set_lab2bgr_form(int form); int orc_get_lab2bgr_form(void);
void orc_u8_to_f64_scaled(const uint8_t* src, size_t n, double* dst);
int orc_kmeans_labels(const float* feat, int n, int C, int K, int iters, uint64_t seed, int* labels);
void orc_knn_graph(const double* lab, int h, int w, const int* labels, int lh, int lw, int nlabels, int samples, int k, int* knn_id, double* knn_w);
typedef struct { double eps, nonlocal_weight, local_weight, wls_lambda_init, wls_alpha, k_num; } orc_color_params;
int orc_local_color_transfer(const float* err, const uint8_t* s_bgr_level, const uint8_t* g_bgr_level, const uint8_t* s_bgr_full,
                             const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W, const orc_color_params* prm,
                             uint8_t* out_bgr_full, const void* st, int s2_exact);

typedef struct {
    double bds_weight, eps, nonlocal_weight, local_weight, wls_lambda_init;
    int cluster_num, k_num, patch_size;
    double wls_alpha;
    int pm_iters;
    uint32_t seed;
    int levels;         /* pyramid levels to run, coarse -> fine (5 = full loop; 1 = "L=5 only", BASELINE config 1) */
    uint32_t flags;     /* only bit 8 (NCT_FLAG_LAB2BGR_CUBE) is honoured; the product's reduced-precision / profiling switches are ignored */
} orc_params;       /* same layout as nct_params (include/nct.h) */

/* per-level intermediates for level-wise validation; same layout as nct_pair_levels (include/nct.h); every pointer nullable */
typedef struct {
    uint32_t* ann[5]; uint32_t* bnn[5];
    float* annd[5]; float* bnnd[5];
    uint8_t* guide[5];
    float* err[5];
    uint8_t* result[5];
} orc_pair_levels;

static const int kTapC[5] = {64, 128, 256, 512, 512};

/* level_out (nullable): receives the 5 intermediate full-resolution results, [5][H*W*3]. Returns 0 on success. */
int orc_process_pair_levels(const uint8_t* src, int H, int W, const uint8_t* ref, int RH, int RW, const float* const* weights, const float* const* biases,
                            const orc_params* prm, uint8_t* out, uint8_t* level_out, int s2_exact, const orc_pair_levels* lv) {
    const int nlevels = prm->levels >= 1 && prm->levels <= 5 ? prm->levels : 5;
    const int form_before = orc_get_lab2bgr_form();
    if (prm->flags & 8u) orc_set_lab2bgr_form(1);
    int ah[5], aw[5], bh[5], bw[5];
    { int h = H, w = W, h2 = RH, w2 = RW;
      for (int t = 0; t < 5; ++t) { ah[4 - t] = h; aw[4 - t] = w; bh[4 - t] = h2; bw[4 - t] = w2; h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; h2 = (h2 - 1) / 2 + 1; w2 = (w2 - 1) / 2 + 1; } }
    int maxLen = W > H ? W : H; if (RW > maxLen) maxLen = RW; if (RH > maxLen) maxLen = RH;
    const int rs_range[5] = {maxLen / 16, maxLen / 32, maxLen / 64, 32, 32};
    const size_t N = (size_t)H * W, NR = (size_t)RH * RW;
    /* pyramids */
    uint8_t* simg[5]; uint8_t* rimg[5];
    simg[4] = (uint8_t*)src; rimg[4] = (uint8_t*)ref;
    for (int l = 3; l >= 0; --l) {
        simg[l] = (uint8_t*)malloc((size_t)ah[l] * aw[l] * 3); rimg[l] = (uint8_t*)malloc((size_t)bh[l] * bw[l] * 3);
        orc_resize_u8c3(simg[l + 1], ah[l + 1], aw[l + 1], simg[l], ah[l], aw[l]);
        orc_resize_u8c3(rimg[l + 1], bh[l + 1], bw[l + 1], rimg[l], bh[l], bw[l]);
    }
    /* features */
    float* rtap[5]; float* stap[5] = {NULL, NULL, NULL, NULL, NULL};
    for (int t = 0; t < 5; ++t) rtap[t] = (float*)malloc(sizeof(float) * kTapC[t] * (size_t)bh[4 - t] * bw[4 - t]);
    orc_vgg19_features(ref, RH, RW, weights, biases, 5, rtap, NULL);
    float* sfeat = (float*)malloc(sizeof(float) * 64 * N);
    stap[4] = sfeat;
    orc_vgg19_features(src, H, W, weights, biases, 5, stap, NULL);
    float* na = (float*)malloc(sizeof(float) * 64 * N); float* nb = (float*)malloc(sizeof(float) * 64 * NR);
    float* voted = (float*)malloc(sizeof(float) * 64 * N); float* nvoted = (float*)malloc(sizeof(float) * 64 * N);
    /* k-means on normalised conv5_1 (HWC) */
    int n0 = ah[0] * aw[0];
    orc_feat_normalize(sfeat, na, NULL, 512, ah[0], aw[0]);
    float* hwc = (float*)malloc(sizeof(float) * 512 * (size_t)n0);
    for (int c = 0; c < 512; ++c) for (int i = 0; i < n0; ++i) hwc[(size_t)i * 512 + c] = na[(size_t)c * n0 + i];
    int* labels = (int*)malloc(sizeof(int) * n0);
    int nlabels = orc_kmeans_labels(hwc, n0, 512, prm->cluster_num, 11, (uint64_t)prm->seed, labels);
    free(hwc);
    uint32_t* ann = (uint32_t*)malloc(sizeof(uint32_t) * N); uint32_t* bnn = (uint32_t*)malloc(sizeof(uint32_t) * NR);
    uint32_t* annp = (uint32_t*)malloc(sizeof(uint32_t) * N); uint32_t* bnnp = (uint32_t*)malloc(sizeof(uint32_t) * NR);
    float* annd = (float*)malloc(sizeof(float) * N); float* bnnd = (float*)malloc(sizeof(float) * NR); float* err = (float*)malloc(sizeof(float) * N);
    uint8_t* guide = (uint8_t*)malloc(N * 3); uint8_t* slab = (uint8_t*)malloc(N * 3);
    double* labd = (double*)malloc(sizeof(double) * N * 3);
    int* knn_id = (int*)malloc(sizeof(int) * N * 8); double* knn_w = (double*)malloc(sizeof(double) * N * 8);
    orc_color_params cp = {prm->eps, prm->nonlocal_weight, prm->local_weight, prm->wls_lambda_init, prm->wls_alpha, (double)prm->k_num};
    int rc = 0;
    for (int l = 0; l < nlevels && rc == 0; ++l) {
        const int C = kTapC[4 - l];
        if (l == 0) { orc_nnf_init(ann, ah[0], aw[0], bh[0], bw[0]); orc_nnf_init(bnn, bh[0], bw[0], ah[0], aw[0]); }
        else {
            memcpy(annp, ann, sizeof(uint32_t) * ah[l - 1] * aw[l - 1]); memcpy(bnnp, bnn, sizeof(uint32_t) * bh[l - 1] * bw[l - 1]);
            orc_nnf_upsample(annp, ann, ah[l], aw[l], bh[l], bw[l], ah[l - 1], aw[l - 1]);
            orc_nnf_upsample(bnnp, bnn, bh[l], bw[l], ah[l], aw[l], bh[l - 1], bw[l - 1]);
        }
        if (l > 0) orc_feat_normalize(sfeat, na, NULL, C, ah[l], aw[l]);
        orc_feat_normalize(rtap[4 - l], nb, NULL, C, bh[l], bw[l]);
        const uint32_t seed_ab = prm->seed ^ (0x9E3779B9u * (uint32_t)(2 * l + 1)), seed_ba = prm->seed ^ (0x9E3779B9u * (uint32_t)(2 * l + 2));
        if (orc_get_pm_schedule() == 0) {
            orc_patchmatch(na, nb, C, ah[l], aw[l], bh[l], bw[l], 3, prm->pm_iters, rs_range[l], seed_ab, ann, annd);
            orc_patchmatch(nb, na, C, bh[l], bw[l], ah[l], aw[l], 3, prm->pm_iters, rs_range[l], seed_ba, bnn, bnnd);
        } else {      /* schedule experiment (tests/golden/gen_pm_inplace_band.py): the reference's own in-place, thread-ordered schedule */
            orc_patchmatch_inplace(na, nb, C, ah[l], aw[l], bh[l], bw[l], 3, prm->pm_iters, rs_range[l], seed_ab, orc_get_pm_schedule(), ann, annd);
            orc_patchmatch_inplace(nb, na, C, bh[l], bw[l], ah[l], aw[l], 3, prm->pm_iters, rs_range[l], seed_ba, orc_get_pm_schedule(), bnn, bnnd);
        }
        if (lv) {
            if (lv->ann[l]) memcpy(lv->ann[l], ann, sizeof(uint32_t) * ah[l] * aw[l]);
            if (lv->bnn[l]) memcpy(lv->bnn[l], bnn, sizeof(uint32_t) * bh[l] * bw[l]);
            if (lv->annd[l]) memcpy(lv->annd[l], annd, sizeof(float) * ah[l] * aw[l]);
            if (lv->bnnd[l]) memcpy(lv->bnnd[l], bnnd, sizeof(float) * bh[l] * bw[l]);
        }
        orc_bds_vote_image(simg[l], ah[l], aw[l], rimg[l], bh[l], bw[l], ann, bnn, 3, 1.0, prm->bds_weight, guide);
        orc_bds_vote_features(ann, bnn, rtap[4 - l], voted, NULL, C, ah[l], aw[l], bh[l], bw[l], 3, 1.f, (float)prm->bds_weight);
        orc_feat_normalize(voted, nvoted, NULL, C, ah[l], aw[l]);
        orc_feature_distance(na, nvoted, err, C, ah[l], aw[l]);
        if (lv) {
            if (lv->guide[l]) memcpy(lv->guide[l], guide, (size_t)ah[l] * aw[l] * 3);
            if (lv->err[l]) memcpy(lv->err[l], err, sizeof(float) * ah[l] * aw[l]);
        }
        orc_bgr2lab_u8(simg[l], (size_t)ah[l] * aw[l], slab);
        orc_u8_to_f64_scaled(slab, (size_t)ah[l] * aw[l] * 3, labd);
        orc_knn_graph(labd, ah[l], aw[l], labels, ah[0], aw[0], nlabels, 1 << l, 8, knn_id, knn_w);
        rc = orc_local_color_transfer(err, simg[l], guide, src, knn_id, knn_w, l, ah[l], aw[l], H, W, &cp, out, NULL, s2_exact);
        if (level_out) memcpy(level_out + (size_t)l * N * 3, out, N * 3);
        if (lv && lv->result[l]) memcpy(lv->result[l], out, N * 3);
        if (l < nlevels - 1) {
            const int tap = 4 - l;
            float* taps[5] = {NULL, NULL, NULL, NULL, NULL};
            taps[tap - 1] = sfeat;
            orc_vgg19_features(out, H, W, weights, biases, tap, taps, NULL);
        }
    }
    for (int l = 0; l < 4; ++l) { free(simg[l]); free(rimg[l]); }
    for (int t = 0; t < 5; ++t) free(rtap[t]);
    free(sfeat); free(na); free(nb); free(voted); free(nvoted); free(labels); free(ann); free(bnn); free(annp); free(bnnp);
    free(annd); free(bnnd); free(err); free(guide); free(slab); free(labd); free(knn_id); free(knn_w);
    orc_set_lab2bgr_form(form_before);
    return rc;
}

int orc_process_pair(const uint8_t* src, int H, int W, const uint8_t* ref, int RH, int RW, const float* const* weights, const float* const* biases,
                     const orc_params* prm, uint8_t* out, uint8_t* level_out, int s2_exact) {
    return orc_process_pair_levels(src, H, W, ref, RH, RW, weights, biases, prm, out, level_out, s2_exact, NULL);
}
