// nct_pipeline.cpp — the per-pair hot loop: the MI355X counterpart of transfer_color_single_bds (main.cu:47-454).
// Everything between "two BGR images in" and "one BGR image out" stays on the device: no per-level cudaMalloc/Free churn
// (main.cu:238-257,297-326), no D2H of NNFs/error maps (main.cu:286-289,318), no host-side BDS vote (main.cu:291),
// no CSR ping-pong for the solvers. S features are recomputed from the intermediate result only up to the tap the
// next level needs (1115 instead of 2297 GFLOP per 700x700 pair, identical values).
#include "nct_internal.h"
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <cmath>

struct pair_state {
    uint8_t *src = nullptr, *ref = nullptr, *out = nullptr;    // device BGR images
    int sh = 0, sw = 0, rh = 0, rw = 0;
};
static pair_state* pair_of(nct_ctx* ctx) {
    if (!ctx->pair) ctx->pair = new pair_state();
    return (pair_state*)ctx->pair;
}
// the images live in the context arena like every other device buffer (no hipMalloc/hipFree — device-wide synchronisation points —
// between the pairs of other contexts in flight on the same GPU)
void nct_pair_free(nct_ctx* ctx) {
    if (!ctx->pair) return;
    pair_state* p = (pair_state*)ctx->pair;
    if (p->src) ctx->release(p->src);
    if (p->ref) ctx->release(p->ref);
    if (p->out) ctx->release(p->out);
    delete p; ctx->pair = nullptr;
}

static const int kTapC[5] = {64, 128, 256, 512, 512};       // tap 1 (conv1_1) … tap 5 (conv5_1)

// stage tags of the event marks (nct_ctx::mark): tag = stage * 8 + level; a mark closes the stage it names
enum { ST_OTHER = 0, ST_VGG, ST_CLUSTER, ST_PM, ST_VOTE, ST_KNN, ST_COLOR, ST_NONLOCAL, ST_WLS };
static thread_local int g_tm_level = 0;          // pyramid level whose colour stage is being enqueued (per host thread = per context)
int nct_stage_tag_nonlocal() { return ST_NONLOCAL * 8 + g_tm_level; }
int nct_stage_tag_wls() { return ST_WLS * 8 + g_tm_level; }
int nct_stage_tag_color() { return ST_COLOR * 8 + g_tm_level; }
#define MARK(stage, level) do { int rcm_ = ctx->mark(s, (stage) * 8 + (level)); if (rcm_) return rcm_; } while (0)

static int read_marks(nct_ctx* ctx, nct_pair_timing* t) {
    double* acc[9] = {&t->other_ms, &t->vgg_ms, &t->cluster_ms, &t->patchmatch_ms, &t->vote_ms, &t->knn_ms, &t->color_ms, &t->nonlocal_ms, &t->wls_ms};
    for (size_t i = 1; i < ctx->tm_tags.size(); ++i) {
        float ms = 0.f;
        NCT_HIP(hipEventElapsedTime(&ms, ctx->tm_events[i - 1], ctx->tm_events[i]));
        const int stage = ctx->tm_tags[i] >> 3, level = ctx->tm_tags[i] & 7;
        if (stage >= 0 && stage < 9) *acc[stage] += ms;
        if (level < 5) {
            if (stage == ST_PM) t->pm_level_ms[level] += ms;
            else if (stage == ST_VOTE) t->vote_level_ms[level] += ms;
            else if (stage == ST_NONLOCAL) t->nonlocal_level_ms[level] += ms;
            else if (stage == ST_WLS) t->wls_level_ms[level] += ms;
        }
    }
    if (getenv("NCT_HOST_TRACE") && ctx->tm_host.size() == ctx->tm_tags.size()) {
        static const char* names[9] = {"other", "vgg", "cluster", "pm", "vote", "knn", "color", "nonlocal", "wls"};
        for (size_t i = 1; i < ctx->tm_tags.size(); ++i) {
            float ms = 0.f; (void)hipEventElapsedTime(&ms, ctx->tm_events[0], ctx->tm_events[i]);
            fprintf(stderr, "nct mark %-8s L%d  host %8.3f ms  gpu %8.3f ms\n", names[(ctx->tm_tags[i] >> 3) % 9], ctx->tm_tags[i] & 7, (ctx->tm_host[i] - ctx->tm_host[0]) / 1000.0, ms);
        }
    }
    t->color_ms += t->nonlocal_ms + t->wls_ms;       // color_ms is the whole stage; the two solves are also reported on their own
    return 0;
}

// run the whole L=5->1 loop on device-resident images
static int process_resident(nct_ctx* ctx, const nct_params* prm, nct_pair_timing* timing, const nct_pair_levels* lv) {
    pair_state* P = (pair_state*)ctx->pair;
    if (!P || !P->src || !P->ref) return ctx->fail(NCT_ERR_STATE, "process: no pair uploaded");
    hipStream_t s = ctx->stream;
    const int H = P->sh, W = P->sw, RH = P->rh, RW = P->rw;
    NCT_REQUIRE(prm->patch_size == 3 && prm->k_num == 8, "process: patch_size must be 3 and k_num 8 (Config.h:68-70)");
    NCT_REQUIRE(prm->cluster_num >= 1 && prm->cluster_num <= 16, "process: cluster_num out of range");
    NCT_REQUIRE(prm->levels >= 1 && prm->levels <= 5, "process: levels must be in [1, 5] (got %d)", prm->levels);
    if (timing) memset(timing, 0, sizeof *timing);
    auto wall0 = std::chrono::steady_clock::now();
    ctx->tm_on = timing != nullptr; ctx->tm_tags.clear(); ctx->tm_host.clear();
    ctx->kt_on = timing != nullptr && (prm->flags & NCT_FLAG_TIME_KERNELS) != 0; ctx->kt_ids.clear();
    struct TmOff { nct_ctx* c; ~TmOff() { c->tm_on = false; c->kt_on = false; } } tm_off{ctx};
    const int nlevels = prm->levels;
    const bool feat16 = (prm->flags & NCT_FLAG_FEAT16) != 0;
    ctx->wls_split = (prm->flags & NCT_FLAG_LATENCY) ? 1 : 0;
    const bool count = timing && (prm->flags & NCT_FLAG_COUNT_EVALS);
    if (count) {
        if (!ctx->d_counter) NCT_HIP(hipMalloc(&ctx->d_counter, 32 * sizeof(unsigned long long)));
        NCT_HIP(hipMemsetAsync(ctx->d_counter, 0, 32 * sizeof(unsigned long long), s));
    }
    MARK(ST_OTHER, 0);

    // level geometry, coarse -> fine (level 0 = conv5_1)
    int ah[5], aw[5], bh[5], bw[5];
    { int h = H, w = W, h2 = RH, w2 = RW;
      for (int t = 0; t < 5; ++t) { ah[4 - t] = h; aw[4 - t] = w; bh[4 - t] = h2; bw[4 - t] = w2; h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; h2 = (h2 - 1) / 2 + 1; w2 = (w2 - 1) / 2 + 1; } }
    const int maxLen = std::max(std::max(W, H), std::max(RW, RH));
    const int rs_range[5] = {maxLen / 16, maxLen / 32, maxLen / 64, 32, 32};                   // main.cu:77-83
    const size_t N = (size_t)H * W;

    // ---- S in Lab (ColorTransfer ctor, ColorTransfer.h:54-75) and image pyramids (main.cu:104-108)
    DevBuf<uint8_t> s_lab_full(ctx, N * 3);
    if (!s_lab_full.ok()) return NCT_ERR_HIP;
    int rc = nctk_bgr2lab(ctx, s, P->src, s_lab_full, N); if (rc) return rc;
    std::vector<DevBuf<uint8_t>*> spyr(5, nullptr), rpyr(5, nullptr);
    struct Cleanup { std::vector<DevBuf<uint8_t>*>&a, &b; ~Cleanup() { for (auto* p : a) delete p; for (auto* p : b) delete p; } } cleanup{spyr, rpyr};
    const uint8_t* simg[5]; const uint8_t* rimg[5];
    simg[4] = P->src; rimg[4] = P->ref;
    for (int l = 3; l >= 0; --l) {
        spyr[l] = new DevBuf<uint8_t>(ctx, (size_t)ah[l] * aw[l] * 3); rpyr[l] = new DevBuf<uint8_t>(ctx, (size_t)bh[l] * bw[l] * 3);
        if (!spyr[l]->ok() || !rpyr[l]->ok()) return NCT_ERR_HIP;
        rc = nctk_resize_u8c3(ctx, s, simg[l + 1], ah[l + 1], aw[l + 1], *spyr[l], ah[l], aw[l]); if (rc) return rc;
        rc = nctk_resize_u8c3(ctx, s, rimg[l + 1], bh[l + 1], bw[l + 1], *rpyr[l], bh[l], bw[l]); if (rc) return rc;
        simg[l] = *spyr[l]; rimg[l] = *rpyr[l];
    }
    MARK(ST_OTHER, 0);

    // ---- VGG19: R once (all five taps kept, HWC), S to conv5_1 (main.cu:94,102)
    DevBuf<float> sfeat(ctx, (size_t)64 * N);   // S features of the current level, channel-last (largest: H x W x 64)
    if (!sfeat.ok()) return NCT_ERR_HIP;
    std::vector<DevBuf<float>*> rfeat(5, nullptr);     // R features, un-normalised, HWC, indexed by level
    struct Cleanup2 { std::vector<DevBuf<float>*>& a; ~Cleanup2() { for (auto* p : a) delete p; } } cleanup2{rfeat};
    {
        // the five taps of R arrive channel-last straight from their conv layers' epilogues (round 4: no CHW -> HWC transpose pass)
        float* taps_hwc[5];
        for (int t = 0; t < 5; ++t) {
            const int l = 4 - t;
            rfeat[l] = new DevBuf<float>(ctx, (size_t)kTapC[t] * bh[l] * bw[l]); if (!rfeat[l]->ok()) return NCT_ERR_HIP;
            taps_hwc[t] = *rfeat[l];
        }
        // R and S together: conv5_1 of both images is one launch (two grids of 124 workgroups at 700 x 700 would each leave half the chip idle)
        float* staps_hwc[5] = {nullptr, nullptr, nullptr, nullptr, sfeat};
        rc = nctk_vgg19_forward_pair(ctx, s, P->ref, RH, RW, RW * 3, taps_hwc, P->src, H, W, W * 3, staps_hwc); if (rc) return rc;
    }
    MARK(ST_VGG, 0);

    // ---- C1: cluster the coarsest S features (main.cu:139-168)
    DevBuf<int> labels(ctx, (size_t)ah[0] * aw[0]), nlab_dev(ctx, 1);
    DevBuf<float> na(ctx, (size_t)64 * N), nb(ctx, (size_t)64 * (size_t)RH * RW), voted(ctx, (size_t)64 * N), nvoted(ctx, (size_t)64 * N);
    // fp16 shadow maps of the normalised features: the candidate tiles of the opt-in reduced-precision mode (NCT_FLAG_FEAT16)
    DevBuf<uint16_t> na_h(ctx, feat16 ? (size_t)64 * N : 8), nb_h(ctx, feat16 ? (size_t)64 * (size_t)RH * RW : 8);
    if (!labels.ok() || !nlab_dev.ok() || !na.ok() || !nb.ok() || !voted.ok() || !nvoted.ok() || !na_h.ok() || !nb_h.ok()) return NCT_ERR_HIP;
    rc = nctk_normalize(ctx, s, sfeat, na, nullptr, 512, ah[0] * aw[0], feat16 ? (uint16_t*)na_h : nullptr); if (rc) return rc;
    rc = nctk_kmeans_labels(ctx, s, na, ah[0] * aw[0], 512, prm->cluster_num, 11, (uint64_t)prm->seed, labels, nlab_dev); if (rc) return rc;
    // the number of labels (1 if k-means degenerated, else K) stays on the device: reading it back would stall the host — and with it
    // the enqueueing of everything below — until the VGG forwards and k-means have finished
    MARK(ST_CLUSTER, 0);

    // ---- K1 for the levels that run (nct_params.levels) on the side stream: the kNN graph of a level depends only on the level image of S and on the
    // labels (main.cu:351-359), not on the correspondence, so it overlaps with PatchMatch / votes / solvers of the main stream
    // (whose many small launches leave most CUs idle). Scratch released meanwhile stays reserved until the join (nct_internal.h).
    std::vector<DevBuf<uint8_t>*> slab(5, nullptr);
    std::vector<DevBuf<int>*> knn_ids(5, nullptr);
    std::vector<DevBuf<double>*> knn_ws(5, nullptr);
    std::vector<nct_s1_graph_bufs*> s1g(5, nullptr);       // the graph-only part of S1's system (reverse adjacency, hub block table: k_s1.hip), built behind each graph
    struct Cleanup3 { std::vector<DevBuf<uint8_t>*>& a; std::vector<DevBuf<int>*>& b; std::vector<DevBuf<double>*>& c; std::vector<nct_s1_graph_bufs*>& d; nct_ctx* ctx;
                      ~Cleanup3() { (void)hipStreamSynchronize(ctx->stream2); ctx->defer_release = false; ctx->flush_deferred();
                                    for (auto* p : a) delete p; for (auto* p : b) delete p; for (auto* p : c) delete p; for (auto* p : d) delete p; } } cleanup3{slab, knn_ids, knn_ws, s1g, ctx};
    // enqueued from inside the level loop, AFTER the coarsest level's correspondence work has been submitted: the side stream's ~200
    // small packets would otherwise sit in front of the main stream's and the main stream starts the level loop ~2.6 ms late
    auto enqueue_knn = [&]() -> int {
        // arena blocks are recycled in stream order: the side stream may reuse blocks the main stream released up to this point, so it
        // starts behind everything enqueued on the main stream so far
        hipStream_t s2 = ctx->stream2;
        NCT_HIP(hipEventRecord(ctx->ev_fork, s));
        NCT_HIP(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
        for (int l = 0; l < nlevels; ++l) {
            const size_t npx = (size_t)ah[l] * aw[l];
            slab[l] = new DevBuf<uint8_t>(ctx, npx * 3); knn_ids[l] = new DevBuf<int>(ctx, npx * 8); knn_ws[l] = new DevBuf<double>(ctx, npx * 8);
            s1g[l] = new nct_s1_graph_bufs(ctx, (int)npx);
            if (!slab[l]->ok() || !knn_ids[l]->ok() || !knn_ws[l]->ok() || !s1g[l]->ok()) return NCT_ERR_HIP;
        }
        int rc2 = 0;
        ctx->defer_release = true;
        for (int l = 0; l < nlevels && rc2 == 0; ++l) {      // only the levels that run (nct_params.levels)
            rc2 = nctk_bgr2lab(ctx, s2, simg[l], *slab[l], (size_t)ah[l] * aw[l]);
            if (rc2 == 0) rc2 = nctk_knn_graph(ctx, s2, *slab[l], ah[l], aw[l], labels, ah[0], aw[0], 0, nlab_dev, 1 << l, *knn_ids[l], *knn_ws[l]);
            // S1's reverse adjacency and hub block table depend on the graph alone: built here, off the main stream; the block count lands in page-locked memory
            // before ev_level[l] completes, so the host can size (or skip) the level's hub passes without a synchronisation
            if (rc2 == 0) rc2 = nctk_s1_graph_build(ctx, s2, *knn_ids[l], *knn_ws[l], sqrt(prm->nonlocal_weight / (double)prm->k_num), s1g[l]->view(-1, -1), ctx->s1_hub_blocks() + 2 * l);
            if (rc2 == 0 && hipEventRecord(ctx->ev_level[l], s2) != hipSuccess) rc2 = ctx->fail(NCT_ERR_HIP, "hipEventRecord failed");
        }
        ctx->defer_release = false;
        return rc2;
    };

    // ---- level loop (main.cu:179-428)
    DevBuf<uint32_t> ann(ctx, N), bnn(ctx, (size_t)RH * RW), ann_prev(ctx, N), bnn_prev(ctx, (size_t)RH * RW);
    DevBuf<float> annd(ctx, N), bnnd(ctx, (size_t)RH * RW), err(ctx, N);
    DevBuf<uint8_t> guide(ctx, N * 3), g_lab_l(ctx, N * 3), out_lab(ctx, N * 3);
    if (!ann.ok() || !bnn.ok() || !ann_prev.ok() || !bnn_prev.ok() || !annd.ok() || !bnnd.ok() || !err.ok() || !guide.ok() ||
        !g_lab_l.ok() || !out_lab.ok()) return NCT_ERR_HIP;
    if (!P->out) { P->out = (uint8_t*)ctx->alloc(N * 3); if (!P->out) return NCT_ERR_HIP; }
    auto d2h = [&](void* dst, const void* src, size_t bytes) -> int {
        if (dst) NCT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        return 0;
    };
    if (lv) { rc = d2h(lv->labels, labels, sizeof(int) * (size_t)ah[0] * aw[0]); if (rc) return rc; }
    nct_color_params cp{prm->eps, prm->nonlocal_weight, prm->local_weight, prm->wls_lambda_init, prm->wls_alpha, (double)prm->k_num};

    for (int l = 0; l < nlevels; ++l) {
        const int C = kTapC[4 - l];
        const int na_px = ah[l] * aw[l], nb_px = bh[l] * bw[l];
        // NNF init / upsample (main.cu:230-251)
        if (l == 0) {
            rc = nctk_nnf_init(ctx, s, ann, ah[0], aw[0], bh[0], bw[0]); if (rc) return rc;
            rc = nctk_nnf_init(ctx, s, bnn, bh[0], bw[0], ah[0], aw[0]); if (rc) return rc;
        } else {
            NCT_HIP(hipMemcpyAsync(ann_prev, ann, sizeof(uint32_t) * ah[l - 1] * aw[l - 1], hipMemcpyDeviceToDevice, s));
            NCT_HIP(hipMemcpyAsync(bnn_prev, bnn, sizeof(uint32_t) * bh[l - 1] * bw[l - 1], hipMemcpyDeviceToDevice, s));
            rc = nctk_nnf_upsample(ctx, s, ann_prev, ann, ah[l], aw[l], bh[l], bw[l], ah[l - 1], aw[l - 1]); if (rc) return rc;
            rc = nctk_nnf_upsample(ctx, s, bnn_prev, bnn, bh[l], bw[l], ah[l], aw[l], bh[l - 1], bw[l - 1]); if (rc) return rc;
        }
        // normalise (main.cu:259-275), PatchMatch both directions (main.cu:283-284)
        if (l > 0) { rc = nctk_normalize(ctx, s, sfeat, na, nullptr, C, na_px, feat16 ? (uint16_t*)na_h : nullptr); if (rc) return rc; }
        rc = nctk_normalize(ctx, s, *rfeat[l], nb, nullptr, C, nb_px, feat16 ? (uint16_t*)nb_h : nullptr); if (rc) return rc;
        MARK(ST_OTHER, l);
        const uint32_t seed_ab = prm->seed ^ (0x9E3779B9u * (uint32_t)(2 * l + 1)), seed_ba = prm->seed ^ (0x9E3779B9u * (uint32_t)(2 * l + 2));
        // na, nb are unit vectors: the row-wise rejection is exact (and worth a third of the finest level: 15.7 vs 24.1 ms with NCT_PM_PLAIN). The fp16 tiles pay from C = 128 on (11-37 % per level); the C = 64 level is
        // latency bound, not byte bound (fp16 tiles: 15.8 vs 16.0 ms), and stays fp32
#ifndef NCT_PIPE_PM_EXACT_MODE
#define NCT_PIPE_PM_EXACT_MODE NCT_PM_ROWREJECT
#endif
        const int pm_mode = (feat16 && C >= 256) ? NCT_PM_FP16 : NCT_PIPE_PM_EXACT_MODE;
        rc = nctk_patchmatch_bidir(ctx, s, na, nb, (const uint16_t*)na_h, (const uint16_t*)nb_h, C, ah[l], aw[l], bh[l], bw[l], prm->pm_iters, rs_range[l], seed_ab, seed_ba,
                                   ann, annd, bnn, bnnd, pm_mode, count ? ctx->d_counter + 4 * l : nullptr); if (rc) return rc;
        MARK(ST_PM, l);
        if (timing) timing->pm_level_launches[l] = (ctx->pm_persist && 4 * prm->pm_iters <= 250) ? 1 : 1 + 4 * prm->pm_iters;
        if (lv) {
            rc = d2h(lv->ann[l], ann, sizeof(uint32_t) * na_px); if (rc) return rc;
            rc = d2h(lv->bnn[l], bnn, sizeof(uint32_t) * nb_px); if (rc) return rc;
            rc = d2h(lv->annd[l], annd, sizeof(float) * na_px); if (rc) return rc;
            rc = d2h(lv->bnnd[l], bnnd, sizeof(float) * nb_px); if (rc) return rc;
        }
        // BDS votes: guidance image (main.cu:291) and features + matching error (main.cu:303-318)
        rc = nctk_bds_vote_both(ctx, s, rimg[l], *rfeat[l], ann, bnn, C, ah[l], aw[l], bh[l], bw[l], 1.0, prm->bds_weight, guide, voted); if (rc) return rc;
        rc = nctk_normalize(ctx, s, voted, nvoted, nullptr, C, na_px); if (rc) return rc;
        rc = nctk_feature_distance(ctx, s, na, nvoted, err, C, na_px); if (rc) return rc;
        MARK(ST_VOTE, l);
        if (lv) {
            rc = d2h(lv->guide[l], guide, (size_t)na_px * 3); if (rc) return rc;
            rc = d2h(lv->err[l], err, sizeof(float) * na_px); if (rc) return rc;
        }
        // kNN graph in Lab (main.cu:351-359): computed on the side stream; join once before its first use
        rc = nctk_bgr2lab(ctx, s, guide, g_lab_l, na_px); if (rc) return rc;
        if (l == 0) { rc = enqueue_knn(); if (rc) return rc; }
        NCT_HIP(hipStreamWaitEvent(s, ctx->ev_level[l], 0));          // level l's graph only: the fine levels keep overlapping
        if (l == nlevels - 1) ctx->flush_deferred();
        const uint8_t* s_lab_l = *slab[l]; const int* knn_id = *knn_ids[l]; const double* knn_w = *knn_ws[l];
        MARK(ST_KNN, l);
        // local colour transfer (main.cu:368-380)
        g_tm_level = l;
        nct_color_debug dbg{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int wls_it[6] = {0, 0, 0, 0, 0, 0};
        dbg.wls_iters = wls_it;
        const nct_color_stages* cs = lv ? lv->color[l] : nullptr;
        if (cs) { dbg.ab_local = cs->ab_local; dbg.ab_nonlocal = cs->ab_nonlocal; dbg.ab_up = cs->ab_up; dbg.rough = cs->roughness; dbg.ab_wls = cs->ab_wls; dbg.cg_iters = cs->cg_iters; }
        // what the host knows about level l's hub blocks right now: the count, if the side stream has passed ev_level[l] (always, from the second level on: the host
        // has just waited for the previous level's WLS solve); else -1 and the hub pass is launched on the device-side count. The result does not depend on it.
        int hub_hint = -1, sup_hint = -1;
        // the coarsest level's graph is built while the host is still far ahead of the GPU (the VGG forwards are running), so its count has not arrived when the host gets
        // here. Default: wait for that one event — the GPU has the level's correspondence work queued meanwhile and the solve's 200 launches are enqueued faster than they
        // execute. NCT_S1_HUB_WAIT=0 (no host wait inside a pair): the level's 101 hub passes (+ 101 second-level passes) are launched blind on grids sized by
        // the level and exit on the device-side count. Measured (profiles/round6_ab.md): 77.9 vs 78.7 ms per single pair, 15.44-15.49 vs 15.31-15.38 pairs/s with four in flight.
        if (ctx->s1_hub_hint && ctx->s1_hub_wait && l == 0) (void)hipEventSynchronize(ctx->ev_level[0]);
        if (ctx->s1_hub_hint && hipEventQuery(ctx->ev_level[l]) == hipSuccess) { hub_hint = *(volatile int*)(ctx->s1_hub_blocks() + 2 * l); sup_hint = *(volatile int*)(ctx->s1_hub_blocks() + 2 * l + 1); }
        (void)hipGetLastError();                                     // hipEventQuery's hipErrorNotReady is not an error
        ctx->s1_hub_blocks_last[l] = hub_hint;
        const nct_s1_graph s1graph = s1g[l]->view(hub_hint, sup_hint);
        rc = nctk_local_color_transfer(ctx, s, err, s_lab_l, g_lab_l, s_lab_full, knn_id, knn_w, l, ah[l], aw[l], H, W, cp, out_lab, (timing || cs) ? &dbg : nullptr, &s1graph); if (rc) return rc;
        if (cs && cs->wls_iters) for (int q = 0; q < 6; ++q) cs->wls_iters[q] = wls_it[q];
        rc = nctk_lab2bgr(ctx, s, out_lab, P->out, N, (prm->flags & NCT_FLAG_LAB2BGR_CUBE) ? 1 : 0); if (rc) return rc;
        if (timing) { timing->wls_iters[l] = *std::max_element(wls_it, wls_it + 6); }
        MARK(ST_COLOR, l);
        if (lv) { rc = d2h(lv->result[l], P->out, N * 3); if (rc) return rc; }
        // re-predict: S features of the next level from the intermediate result (main.cu:424-427)
        if (l < nlevels - 1) {
            const int tap = 4 - l;                         // next level uses tap (5 - (l+1))
            float* taps_hwc[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            taps_hwc[tap - 1] = sfeat;
            rc = nctk_vgg19_forward(ctx, s, P->out, H, W, W * 3, tap, nullptr, nullptr, taps_hwc); if (rc) return rc;
            MARK(ST_VGG, l);
        }
    }
    // the side stream's kNN graphs (one per level that ran) finish before their buffers go back (Cleanup3 synchronises stream2)
    NCT_HIP(hipStreamSynchronize(s));
    rc = nctk_pm_check(ctx); if (rc) return rc;
    if (timing) {
        timing->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        rc = read_marks(ctx, timing); if (rc) return rc;
        for (size_t i = 0; i < ctx->kt_ids.size(); ++i) {               // NCT_FLAG_TIME_KERNELS: average the samples per kernel
            float ms = 0.f;
            NCT_HIP(hipEventElapsedTime(&ms, ctx->kt_events[2 * i], ctx->kt_events[2 * i + 1]));
            const int id = ctx->kt_ids[i];
            if (id >= 0 && id < 10) { timing->kernel_us[id] += 1e3 * ms; timing->kernel_samples[id] += 1; }
        }
        for (int id = 0; id < 10; ++id) if (timing->kernel_samples[id]) timing->kernel_us[id] /= timing->kernel_samples[id];
        if (count) {
            unsigned long long h[32];
            NCT_HIP(hipMemcpy(h, ctx->d_counter, sizeof h, hipMemcpyDeviceToHost));
            for (int l = 0; l < 5; ++l) { timing->pm_level_evals[l] = h[4 * l]; timing->pm_level_accepted[l] = h[4 * l + 1]; }
        }
    }
    return NCT_OK;
}

extern "C" {

int nct_pair_upload(nct_ctx* ctx, const uint8_t* src_bgr, int sh, int sw, const uint8_t* ref_bgr, int rh, int rw) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(src_bgr && ref_bgr, "pair_upload: null image");
    // the coarsest pyramid level (four ceil-halvings) must be at least 2x2 (init_Ann_kernel scales by (bw-1)/(aw-1)): side >= 17
    NCT_REQUIRE(sh >= 17 && sw >= 17 && rh >= 17 && rw >= 17 && sh <= 4000 && sw <= 4000 && rh <= 4000 && rw <= 4000,
                "pair_upload: image sides must be in [17, 4000] (got %dx%d and %dx%d)", sw, sh, rw, rh);
    pair_state* P = pair_of(ctx);
    if (P->src) { ctx->release(P->src); P->src = nullptr; }
    if (P->ref) { ctx->release(P->ref); P->ref = nullptr; }
    if (P->out) { ctx->release(P->out); P->out = nullptr; }
    P->src = (uint8_t*)ctx->alloc((size_t)sh * sw * 3);
    P->ref = (uint8_t*)ctx->alloc((size_t)rh * rw * 3);
    if (!P->src || !P->ref) return NCT_ERR_HIP;
    NCT_HIP(hipMemcpyAsync(P->src, src_bgr, (size_t)sh * sw * 3, hipMemcpyHostToDevice, ctx->stream));
    NCT_HIP(hipMemcpyAsync(P->ref, ref_bgr, (size_t)rh * rw * 3, hipMemcpyHostToDevice, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    P->sh = sh; P->sw = sw; P->rh = rh; P->rw = rw;
    return NCT_OK;
}

int nct_pair_run(nct_ctx* ctx, const nct_params* prm, nct_pair_timing* timing) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(prm, "pair_run: null params");
    return process_resident(ctx, prm, timing, nullptr);
}

int nct_pair_run_levels(nct_ctx* ctx, const nct_params* prm, nct_pair_timing* timing, const nct_pair_levels* levels) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(prm, "pair_run_levels: null params");
    return process_resident(ctx, prm, timing, levels);
}

int nct_pair_download(nct_ctx* ctx, uint8_t* out_bgr) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    pair_state* P = (pair_state*)ctx->pair;
    if (!P || !P->out) return ctx->fail(NCT_ERR_STATE, "pair_download: no result (call nct_pair_run first)");
    NCT_REQUIRE(out_bgr, "pair_download: null pointer");
    NCT_HIP(hipMemcpyAsync(out_bgr, P->out, (size_t)P->sh * P->sw * 3, hipMemcpyDeviceToHost, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    return NCT_OK;
}

int nct_process_pair(nct_ctx* ctx, const uint8_t* src_bgr, int sh, int sw, const uint8_t* ref_bgr, int rh, int rw, const nct_params* prm,
                     uint8_t* out_bgr, nct_pair_timing* timing) {
    int rc = nct_pair_upload(ctx, src_bgr, sh, sw, ref_bgr, rh, rw); if (rc) return rc;
    rc = nct_pair_run(ctx, prm, timing); if (rc) return rc;
    return nct_pair_download(ctx, out_bgr);
}

}  // extern "C"
