// nct_api.cpp — C-ABI entry points (host-pointer variants) + context / arena management.
// Every function here is a thin marshalling layer: upload, call the device launcher (nctk_*), download.
#include <atomic>
#include "nct_internal.h"
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <mutex>

static std::string g_create_err;
void nct_vgg_free(nct_ctx* ctx);   // nct_vgg.cpp

void* nct_ctx::alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < blocks.size(); ++i)
        if (!blocks[i].used && blocks[i].bytes >= bytes && (best < 0 || blocks[i].bytes < blocks[best].bytes)) best = (int)i;
    if (best >= 0 && blocks[best].bytes <= bytes * 2 + (1u << 20)) { blocks[best].used = true; return blocks[best].p; }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // free cached-but-unused blocks and retry once
        for (auto& b : blocks) if (!b.used && b.p) { (void)hipFree(b.p); bytes_allocated -= b.bytes; b.p = nullptr; b.bytes = 0; }
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { fail(NCT_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    }
    bytes_allocated += bytes;
    for (auto& b : blocks) if (!b.p) { b = {p, bytes, true}; return p; }
    blocks.push_back({p, bytes, true});
    return p;
}
int nct_ctx::mark(hipStream_t s, int tag) {
    if (!tm_on) return 0;
    const size_t i = tm_tags.size();
    if (i >= tm_events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return fail(NCT_ERR_HIP, "hipEventCreate failed");
        tm_events.push_back(e);
    }
    if (hipEventRecord(tm_events[i], s) != hipSuccess) return fail(NCT_ERR_HIP, "hipEventRecord failed");
    tm_tags.push_back(tag);
    tm_host.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
    return 0;
}
int nct_ctx::kt_begin(hipStream_t s, int id) {
    if (!kt_on) return 0;
    const size_t i = 2 * kt_ids.size();
    while (kt_events.size() < i + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return fail(NCT_ERR_HIP, "hipEventCreate failed");
        kt_events.push_back(e);
    }
    if (hipEventRecord(kt_events[i], s) != hipSuccess) return fail(NCT_ERR_HIP, "hipEventRecord failed");
    kt_ids.push_back(id);
    return 0;
}
int nct_ctx::kt_end(hipStream_t s) {
    if (!kt_on || kt_ids.empty()) return 0;
    if (hipEventRecord(kt_events[2 * kt_ids.size() - 1], s) != hipSuccess) return fail(NCT_ERR_HIP, "hipEventRecord failed");
    return 0;
}
void nct_ctx::release(void* p) {
    if (defer_release) { deferred.push_back(p); return; }
    for (auto& b : blocks) if (b.p == p) { b.used = false; return; }
}

extern "C" {

int nct_version(void) { return NCT_VERSION; }

int nct_create(int device, nct_ctx** out) {
    if (!out) { g_create_err = "nct_create: out is NULL"; return NCT_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_err = std::string("nct_create: no usable HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                       "); libnct has no CPU fallback";
        return NCT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { g_create_err = "nct_create: device index out of range"; return NCT_ERR_INVALID; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return NCT_ERR_HIP; }
    nct_ctx* c = new nct_ctx();
    c->device = device;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream_wls, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_wls_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_wls_join, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) {
        g_create_err = std::string("stream/event creation: ") + hipGetErrorString(e);
        delete c; return NCT_ERR_HIP;
    }
    for (int l = 0; l < 5; ++l)
        if ((e = hipEventCreateWithFlags(&c->ev_level[l], hipEventDisableTiming)) != hipSuccess) { g_create_err = std::string("event creation: ") + hipGetErrorString(e); delete c; return NCT_ERR_HIP; }
    for (int l = 0; l < 4; ++l)
        if ((e = hipEventCreateWithFlags(&c->ev_poll[l], hipEventDisableTiming)) != hipSuccess) { g_create_err = std::string("event creation: ") + hipGetErrorString(e); delete c; return NCT_ERR_HIP; }
    if ((e = hipHostMalloc(&c->pinned, 4096 + 64, hipHostMallocDefault)) != hipSuccess) { g_create_err = std::string("hipHostMalloc: ") + hipGetErrorString(e); delete c; return NCT_ERR_HIP; }
    if (const char* g = getenv("NCT_WLS_GRAPH")) c->wls_graph = atoi(g);
    if (const char* g = getenv("NCT_S2_LINES")) c->wls_lines = atoi(g) != 0;
    if (const char* f = getenv("NCT_WLS_FORECAST")) { const int v = atoi(f); if (v == 0 || v == 1) c->wls_forecast = v; }
    if (const char* r = getenv("NCT_WLS_RTOL")) { const double v = atof(r); if (v > 0 && v < 1) c->wls_rtol = v; }
    if (const char* f = getenv("NCT_CONV_POOL_FUSE")) { const int v = atoi(f); if (v == 0 || v == 1) c->conv_pool_fuse = v; }
    { static std::atomic<int> next_home{0}; c->home_xcd = next_home.fetch_add(1) & 7; }
    if (const char* q = getenv("NCT_CONV_PAIR")) { const int v = atoi(q); if (v == 0 || v == 1) c->conv_pair = v; }
    if (const char* q = getenv("NCT_KNN_RUNS")) { const int v = atoi(q); if (v == 0 || v == 1) c->knn_runs = v; }
    if (const char* q = getenv("NCT_PM_PERSIST")) { const int v = atoi(q); if (v == 0 || v == 1) c->pm_persist = v; }
    if (const char* q = getenv("NCT_PM_PERSIST_WGS")) { const int v = atoi(q); if (v > 0) c->pm_persist_wgs = v; }
    if (const char* q = getenv("NCT_S1_HUB_HINT")) { const int v = atoi(q); if (v == 0 || v == 1) c->s1_hub_hint = v; }
    if (const char* q = getenv("NCT_S1_HUB_WAIT")) { const int v = atoi(q); if (v == 0 || v == 1) c->s1_hub_wait = v; }
    if (const char* m = getenv("NCT_WLS_MAXIT")) { const int v = atoi(m); if (v > 0) c->wls_maxit = v; }
    *out = c;
    return NCT_OK;
}

int nct_ctx_counter(nct_ctx* ctx, int which, int64_t* out) {
    if (!ctx || !out) return NCT_ERR_INVALID;
    switch (which) {
        case NCT_CTR_ARENA_BYTES: *out = (int64_t)ctx->bytes_allocated; return NCT_OK;
        case NCT_CTR_S1_HUB_BLOCKS_L0: case NCT_CTR_S1_HUB_BLOCKS_L0 + 1: case NCT_CTR_S1_HUB_BLOCKS_L0 + 2: case NCT_CTR_S1_HUB_BLOCKS_L0 + 3: case NCT_CTR_S1_HUB_BLOCKS_L0 + 4:
            *out = ctx->s1_hub_blocks_last[which - NCT_CTR_S1_HUB_BLOCKS_L0]; return NCT_OK;
    }
    return ctx->fail(NCT_ERR_INVALID, "nct_ctx_counter: unknown counter %d", which);
}

int nct_device_count(int* count) {
    if (!count) return NCT_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { *count = 0; g_create_err = "nct_device_count: no HIP device (no CPU fallback)"; return NCT_ERR_NO_DEVICE; }
    *count = n;
    return NCT_OK;
}
int nct_device_pci_bus_id(int device, char* buf, int buflen) {
    if (!buf || buflen < 13) return NCT_ERR_INVALID;
    hipError_t e = hipDeviceGetPCIBusId(buf, buflen, device);
    if (e != hipSuccess) { g_create_err = std::string("hipDeviceGetPCIBusId: ") + hipGetErrorString(e); return NCT_ERR_HIP; }
    for (char* p = buf; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells the address in lower case
    return NCT_OK;
}

void nct_destroy(nct_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    nct_vgg_free(ctx);
    nct_cvt_free(ctx);
    nct_pair_free(ctx);
    for (auto& b : ctx->blocks) if (b.p) (void)hipFree(b.p);
    if (ctx->bench_a) (void)hipFree(ctx->bench_a);
    if (ctx->bench_b) (void)hipFree(ctx->bench_b);
    if (ctx->bench_ah16) (void)hipFree(ctx->bench_ah16);
    if (ctx->bench_bh16) (void)hipFree(ctx->bench_bh16);
    for (hipEvent_t e : ctx->tm_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->kt_events) (void)hipEventDestroy(e);
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    if (ctx->d_pm_err) (void)hipFree(ctx->d_pm_err);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (int l = 0; l < 5; ++l) if (ctx->ev_level[l]) (void)hipEventDestroy(ctx->ev_level[l]);
    for (int l = 0; l < 4; ++l) if (ctx->ev_poll[l]) (void)hipEventDestroy(ctx->ev_poll[l]);
    if (ctx->stream_wls) (void)hipStreamDestroy(ctx->stream_wls);
    if (ctx->ev_wls_fork) (void)hipEventDestroy(ctx->ev_wls_fork);
    if (ctx->ev_wls_join) (void)hipEventDestroy(ctx->ev_wls_join);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    delete ctx;
}

const char* nct_last_error(const nct_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int nct_device_name(nct_ctx* ctx, char* buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return NCT_ERR_INVALID;
    hipDeviceProp_t p;
    NCT_HIP(hipGetDeviceProperties(&p, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return NCT_OK;
}

int nct_synchronize(nct_ctx* ctx) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_HIP(hipDeviceSynchronize());
    return nctk_pm_check(ctx);
}

#define CTX_ENTER() do { if (!ctx) return NCT_ERR_INVALID; NCT_HIP(hipSetDevice(ctx->device)); } while (0)
#define H2D(dst, src, bytes) NCT_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, ctx->stream))
#define D2H(dst, src, bytes) NCT_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, ctx->stream))
#define SYNC() NCT_HIP(hipStreamSynchronize(ctx->stream))
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// upload a CHW host tensor and return it in HWC device layout
static int upload_hwc(nct_ctx* ctx, const float* chw, float* d_tmp, float* d_hwc, int C, int HW) {
    H2D(d_tmp, chw, sizeof(float) * (size_t)C * HW);
    return nctk_chw_to_hwc(ctx, ctx->stream, d_tmp, d_hwc, C, HW);
}

int nct_feat_normalize(nct_ctx* ctx, const float* src_chw, float* dst_chw, float* resp, int C, int H, int W) {
    CTX_ENTER();
    NCT_REQUIRE(src_chw && dst_chw && H > 0 && W > 0, "feat_normalize: null pointer or empty image");
    const int HW = H * W; const size_t n = (size_t)C * HW;
    DevBuf<float> t(ctx, n), x(ctx, n), y(ctx, n), r(ctx, HW);
    if (!t.ok() || !x.ok() || !y.ok() || !r.ok()) return NCT_ERR_HIP;
    RC(upload_hwc(ctx, src_chw, t, x, C, HW));
    RC(nctk_normalize(ctx, ctx->stream, x, y, resp ? (float*)r : nullptr, C, HW));
    RC(nctk_hwc_to_chw(ctx, ctx->stream, y, t, C, HW));
    D2H(dst_chw, t, sizeof(float) * n);
    if (resp) D2H(resp, r, sizeof(float) * HW);
    SYNC();
    return NCT_OK;
}

int nct_nnf_init(nct_ctx* ctx, uint32_t* nnf, int ah, int aw, int bh, int bw) {
    CTX_ENTER();
    NCT_REQUIRE(nnf, "nnf_init: null pointer");
    NCT_REQUIRE(ah > 0 && aw > 0, "nnf_init: empty query image");
    DevBuf<uint32_t> d(ctx, (size_t)ah * aw);
    if (!d.ok()) return NCT_ERR_HIP;
    RC(nctk_nnf_init(ctx, ctx->stream, d, ah, aw, bh, bw));
    D2H(nnf, d, sizeof(uint32_t) * (size_t)ah * aw);
    SYNC();
    return NCT_OK;
}

int nct_nnf_upsample(nct_ctx* ctx, const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half) {
    CTX_ENTER();
    NCT_REQUIRE(nnf_half && nnf, "nnf_upsample: null pointer");
    NCT_REQUIRE(ah > 0 && aw > 0 && ah_half > 0 && aw_half > 0, "nnf_upsample: empty image");
    DevBuf<uint32_t> h(ctx, (size_t)ah_half * aw_half), d(ctx, (size_t)ah * aw);
    if (!h.ok() || !d.ok()) return NCT_ERR_HIP;
    H2D(h, nnf_half, sizeof(uint32_t) * (size_t)ah_half * aw_half);
    RC(nctk_nnf_upsample(ctx, ctx->stream, h, d, ah, aw, bh, bw, ah_half, aw_half));
    D2H(nnf, d, sizeof(uint32_t) * (size_t)ah * aw);
    SYNC();
    return NCT_OK;
}

int nct_patchmatch(nct_ctx* ctx, const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw,
                   int patch, int iters, int rs_max, uint32_t seed, uint32_t* nnf, float* dist) {
    CTX_ENTER();
    NCT_REQUIRE(a_chw && b_chw && nnf && dist, "patchmatch: null pointer");
    NCT_REQUIRE(patch == 3, "patchmatch: only patch=3 is supported (Config.h:70), got %d", patch);
    NCT_REQUIRE(ah > 0 && aw > 0 && bh > 0 && bw > 0, "patchmatch: empty image");
    const size_t na = (size_t)ah * aw, nb = (size_t)bh * bw;
    DevBuf<float> t(ctx, (size_t)C * (na > nb ? na : nb)), A(ctx, C * na), B(ctx, C * nb), d(ctx, na);
    DevBuf<uint32_t> n(ctx, na);
    if (!t.ok() || !A.ok() || !B.ok() || !d.ok() || !n.ok()) return NCT_ERR_HIP;
    RC(upload_hwc(ctx, a_chw, t, A, C, (int)na));
    RC(upload_hwc(ctx, b_chw, t, B, C, (int)nb));
    H2D(n, nnf, sizeof(uint32_t) * na);
    RC(nctk_patchmatch(ctx, ctx->stream, A, B, C, ah, aw, bh, bw, iters, rs_max, seed, n, d, nullptr));
    D2H(nnf, n, sizeof(uint32_t) * na);
    D2H(dist, d, sizeof(float) * na);
    RC(nctk_pm_check(ctx));
    SYNC();
    return NCT_OK;
}

int nct_bds_vote_features(nct_ctx* ctx, const uint32_t* ann, const uint32_t* bnn, const float* pin_chw, float* pout_chw, float* pw,
                          int C, int ah, int aw, int bh, int bw, int patch, float w_coherence, float w_complete) {
    CTX_ENTER();
    NCT_REQUIRE(ann && bnn && pin_chw && pout_chw, "bds_vote_features: null pointer");
    NCT_REQUIRE(patch == 3, "bds_vote_features: only patch=3 is supported, got %d", patch);
    NCT_REQUIRE(ah > 0 && aw > 0 && bh > 0 && bw > 0, "bds_vote_features: empty image");
    const size_t na = (size_t)ah * aw, nb = (size_t)bh * bw;
    DevBuf<float> t(ctx, (size_t)C * (na > nb ? na : nb)), P(ctx, C * nb), O(ctx, C * na), w(ctx, na);
    DevBuf<uint32_t> da(ctx, na), db(ctx, nb);
    if (!t.ok() || !P.ok() || !O.ok() || !w.ok() || !da.ok() || !db.ok()) return NCT_ERR_HIP;
    RC(upload_hwc(ctx, pin_chw, t, P, C, (int)nb));
    H2D(da, ann, sizeof(uint32_t) * na);
    H2D(db, bnn, sizeof(uint32_t) * nb);
    RC(nctk_bds_vote_features(ctx, ctx->stream, da, db, P, O, w, C, ah, aw, bh, bw, w_coherence, w_complete));
    RC(nctk_hwc_to_chw(ctx, ctx->stream, O, t, C, (int)na));
    D2H(pout_chw, t, sizeof(float) * C * na);
    if (pw) D2H(pw, w, sizeof(float) * na);
    SYNC();
    return NCT_OK;
}

int nct_feature_distance(nct_ctx* ctx, const float* a_chw, const float* b_chw, float* err, int C, int H, int W) {
    CTX_ENTER();
    NCT_REQUIRE(a_chw && b_chw && err && H > 0 && W > 0, "feature_distance: null pointer or empty image");
    const size_t n = (size_t)H * W;
    DevBuf<float> t(ctx, C * n), A(ctx, C * n), B(ctx, C * n), e(ctx, n);
    if (!t.ok() || !A.ok() || !B.ok() || !e.ok()) return NCT_ERR_HIP;
    RC(upload_hwc(ctx, a_chw, t, A, C, (int)n));
    RC(upload_hwc(ctx, b_chw, t, B, C, (int)n));
    RC(nctk_feature_distance(ctx, ctx->stream, A, B, e, C, (int)n));
    D2H(err, e, sizeof(float) * n);
    SYNC();
    return NCT_OK;
}

int nct_bds_vote_image(nct_ctx* ctx, const uint8_t* a_bgr, int ah, int aw, const uint8_t* b_bgr, int bh, int bw,
                       const uint32_t* ann, const uint32_t* bnn, int patch, double w_coherence, double w_complete, uint8_t* out_bgr) {
    CTX_ENTER();
    (void)a_bgr;   // reconstruct_bds only uses a's size (GeneralizedPatchMatch.cu:124-131)
    NCT_REQUIRE(b_bgr && ann && bnn && out_bgr, "bds_vote_image: null pointer");
    NCT_REQUIRE(patch == 3, "bds_vote_image: only patch=3 is supported, got %d", patch);
    NCT_REQUIRE(ah > 0 && aw > 0 && bh > 0 && bw > 0, "bds_vote_image: empty image");
    const size_t na = (size_t)ah * aw, nb = (size_t)bh * bw;
    DevBuf<uint8_t> b(ctx, nb * 3), o(ctx, na * 3);
    DevBuf<uint32_t> da(ctx, na), db(ctx, nb);
    if (!b.ok() || !o.ok() || !da.ok() || !db.ok()) return NCT_ERR_HIP;
    H2D(b, b_bgr, nb * 3);
    H2D(da, ann, sizeof(uint32_t) * na);
    H2D(db, bnn, sizeof(uint32_t) * nb);
    RC(nctk_bds_vote_image(ctx, ctx->stream, b, da, db, ah, aw, bh, bw, w_coherence, w_complete, o));
    D2H(out_bgr, o, na * 3);
    SYNC();
    return NCT_OK;
}

// ---------------------------------------------------------------- measurement hooks
int nct_pm_bench_setup(nct_ctx* ctx, const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw) {
    CTX_ENTER();
    NCT_REQUIRE(a_chw && b_chw, "pm_bench_setup: null pointer");
    const size_t na = (size_t)ah * aw, nb = (size_t)bh * bw;
    if (ctx->bench_a) { (void)hipFree(ctx->bench_a); ctx->bench_a = nullptr; }
    if (ctx->bench_b) { (void)hipFree(ctx->bench_b); ctx->bench_b = nullptr; }
    if (ctx->bench_ah16) { (void)hipFree(ctx->bench_ah16); ctx->bench_ah16 = nullptr; }
    if (ctx->bench_bh16) { (void)hipFree(ctx->bench_bh16); ctx->bench_bh16 = nullptr; }
    NCT_HIP(hipMalloc(&ctx->bench_a, sizeof(float) * C * na));
    NCT_HIP(hipMalloc(&ctx->bench_b, sizeof(float) * C * nb));
    NCT_HIP(hipMalloc(&ctx->bench_ah16, 2 * (size_t)C * na));
    NCT_HIP(hipMalloc(&ctx->bench_bh16, 2 * (size_t)C * nb));
    if (!ctx->d_counter) NCT_HIP(hipMalloc(&ctx->d_counter, 32 * sizeof(unsigned long long)));
    {
        DevBuf<float> t(ctx, (size_t)C * (na > nb ? na : nb)), x(ctx, (size_t)C * (na > nb ? na : nb));
        if (!t.ok() || !x.ok()) return NCT_ERR_HIP;
        RC(upload_hwc(ctx, a_chw, t, x, C, (int)na));
        RC(nctk_normalize(ctx, ctx->stream, x, ctx->bench_a, nullptr, C, (int)na, ctx->bench_ah16));
        RC(upload_hwc(ctx, b_chw, t, x, C, (int)nb));
        RC(nctk_normalize(ctx, ctx->stream, x, ctx->bench_b, nullptr, C, (int)nb, ctx->bench_bh16));
        SYNC();
    }
    ctx->bench_C = C; ctx->bench_ah = ah; ctx->bench_aw = aw; ctx->bench_bh = bh; ctx->bench_bw = bw;
    return NCT_OK;
}

int nct_pm_bench_run(nct_ctx* ctx, int iters, int rs_max, uint32_t seed, float* kernel_ms, uint64_t* evals, uint32_t* nnf_out, float* dist_out) {
    CTX_ENTER();
    if (!ctx->bench_a) return ctx->fail(NCT_ERR_STATE, "pm_bench_run: call nct_pm_bench_setup first");
    const int C = ctx->bench_C, ah = ctx->bench_ah, aw = ctx->bench_aw, bh = ctx->bench_bh, bw = ctx->bench_bw;
    const size_t na = (size_t)ah * aw;
    DevBuf<uint32_t> n(ctx, na);
    DevBuf<float> d(ctx, na);
    if (!n.ok() || !d.ok()) return NCT_ERR_HIP;
    RC(nctk_nnf_init(ctx, ctx->stream, n, ah, aw, bh, bw));
    unsigned long long* counter = evals ? ctx->d_counter : nullptr;
    if (counter) NCT_HIP(hipMemsetAsync(counter, 0, 4 * sizeof(unsigned long long), ctx->stream));
    NCT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    RC(nctk_patchmatch(ctx, ctx->stream, ctx->bench_a, ctx->bench_b, C, ah, aw, bh, bw, iters, rs_max, seed, n, d, counter));
    NCT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NCT_HIP(hipEventSynchronize(ctx->ev1));
    RC(nctk_pm_check(ctx));
    float ms = 0.f;
    NCT_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (kernel_ms) *kernel_ms = ms;
    if (evals) { unsigned long long h = 0; NCT_HIP(hipMemcpy(&h, counter, sizeof h, hipMemcpyDeviceToHost)); *evals = (uint64_t)h; }
    if (nnf_out) NCT_HIP(hipMemcpy(nnf_out, n, sizeof(uint32_t) * na, hipMemcpyDeviceToHost));
    if (dist_out) NCT_HIP(hipMemcpy(dist_out, d, sizeof(float) * na, hipMemcpyDeviceToHost));
    return NCT_OK;
}

int nct_pm_bench_run_bidir(nct_ctx* ctx, int iters, int rs_max, uint32_t seed, int pm_mode, float* kernel_ms, uint64_t* counters, uint32_t* ann_out, float* annd_out,
                           uint32_t* bnn_out, float* bnnd_out) {
    CTX_ENTER();
    if (!ctx->bench_a) return ctx->fail(NCT_ERR_STATE, "pm_bench_run_bidir: call nct_pm_bench_setup first");
    NCT_REQUIRE(pm_mode >= NCT_PM_PLAIN && pm_mode <= NCT_PM_FP16, "pm_bench_run_bidir: pm_mode must be 0 (fp32), 1 (fp32 + row rejection) or 2 (fp16)");
    const int C = ctx->bench_C, ah = ctx->bench_ah, aw = ctx->bench_aw, bh = ctx->bench_bh, bw = ctx->bench_bw;
    const size_t na = (size_t)ah * aw, nb = (size_t)bh * bw;
    DevBuf<uint32_t> an(ctx, na), bn(ctx, nb);
    DevBuf<float> ad(ctx, na), bd(ctx, nb);
    if (!an.ok() || !bn.ok() || !ad.ok() || !bd.ok()) return NCT_ERR_HIP;
    RC(nctk_nnf_init(ctx, ctx->stream, an, ah, aw, bh, bw));
    RC(nctk_nnf_init(ctx, ctx->stream, bn, bh, bw, ah, aw));
    unsigned long long* counter = counters ? ctx->d_counter : nullptr;
    if (counter) NCT_HIP(hipMemsetAsync(counter, 0, 4 * sizeof(unsigned long long), ctx->stream));
    NCT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    RC(nctk_patchmatch_bidir(ctx, ctx->stream, ctx->bench_a, ctx->bench_b, ctx->bench_ah16, ctx->bench_bh16, C, ah, aw, bh, bw, iters, rs_max, seed, seed ^ 0x5bd1e995u,
                             an, ad, bn, bd, pm_mode, counter));
    NCT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NCT_HIP(hipEventSynchronize(ctx->ev1));
    RC(nctk_pm_check(ctx));
    float ms = 0.f;
    NCT_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (kernel_ms) *kernel_ms = ms;
    if (counters) { unsigned long long h[2] = {0, 0}; NCT_HIP(hipMemcpy(h, counter, sizeof h, hipMemcpyDeviceToHost)); for (int i = 0; i < 2; ++i) counters[i] = (uint64_t)h[i]; }
    if (ann_out) NCT_HIP(hipMemcpy(ann_out, an, sizeof(uint32_t) * na, hipMemcpyDeviceToHost));
    if (annd_out) NCT_HIP(hipMemcpy(annd_out, ad, sizeof(float) * na, hipMemcpyDeviceToHost));
    if (bnn_out) NCT_HIP(hipMemcpy(bnn_out, bn, sizeof(uint32_t) * nb, hipMemcpyDeviceToHost));
    if (bnnd_out) NCT_HIP(hipMemcpy(bnnd_out, bd, sizeof(float) * nb, hipMemcpyDeviceToHost));
    return NCT_OK;
}

}  // extern "C"
