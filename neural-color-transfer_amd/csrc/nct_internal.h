// nct_internal.h — context, device-memory arena and launch helpers shared by the libnct translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>
#include "../../include/nct.h"

struct nct_block { void* p; size_t bytes; bool used; };

struct nct_ctx {
    int device = 0;
    hipStream_t stream = nullptr;     // main stream (S->R direction, VGG, colour stage)
    hipStream_t stream2 = nullptr;    // second stream (R->S direction runs concurrently)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_level[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // side-stream completion of level l's kNN graph
    hipEvent_t ev_poll[4] = {nullptr, nullptr, nullptr, nullptr};   // completion of the in-flight solver-state read-backs (k_wls_mg.hip: two per half-solve)
    hipStream_t stream_wls = nullptr;  // helper stream of the split WLS solve (NCT_FLAG_LATENCY)
    hipEvent_t ev_wls_fork = nullptr, ev_wls_join = nullptr;
    void* pinned = nullptr;                        // 4 KB of page-locked host memory for those read-backs (+ 64 B: s1_hub_blocks)
    std::string err;
    std::vector<nct_block> blocks;    // cached device allocations, reused across calls and pairs
    size_t bytes_allocated = 0;
    // measurement fixture (nct_pm_bench_*)
    float *bench_a = nullptr, *bench_b = nullptr; void *bench_ah16 = nullptr, *bench_bh16 = nullptr; int bench_C = 0, bench_ah = 0, bench_aw = 0, bench_bh = 0, bench_bw = 0;
    // opaque sub-states owned by other translation units
    void* vgg = nullptr;              // struct vgg_weights* (nct_vgg.cpp)
    void* cvt = nullptr;              // struct cvt_dev* (k_cvt.hip): colour-conversion LUTs on the device
    void* pair = nullptr;             // struct pair_state* (nct_pipeline.cpp): device-resident source/reference/result images
    unsigned long long* d_counter = nullptr;   // device counters of the pm kernels (NCT_FLAG_COUNT_EVALS): [0] distance evaluations performed, [1] accepted candidates;
                                                // 4 slots per pyramid level in pair runs (nct_pipeline.cpp reads [4 l] and [4 l + 1])
    int home_xcd = 0;                           // the XCD this context's single-XCD launches aim at (k_s1.hip: small S1 levels); contexts of a process count round-robin (nct_create)
    int pm_persist = 0;                         // PatchMatch: one persistent launch per pyramid level (k_pm_level) instead of 1 + 4 iters launches (env NCT_PM_PERSIST)
    int pm_persist_wgs = 0;                     // workgroups of that launch (0: CUs x occupancy; env NCT_PM_PERSIST_WGS, experiments)
    uint32_t* d_pm_err = nullptr;               // device word the persistent kernel's watchdog sets; read by nctk_pm_check at the synchronisation points
    unsigned pm_attr_mask = 0;                 // k_pm_step instantiations whose dynamic-LDS opt-in has been set on this context's device
    // stage clock: events recorded on the main stream at stage boundaries, read once after the pair's final synchronise
    // (no host syncs in between: see nct_pair_timing in nct.h)
    int wls_split = 0;                          // NCT_FLAG_LATENCY of the running pair: a- and b-half of the WLS solve on two streams
    int wls_forecast = 1;                       // size the PCG iteration batches by the host's convergence forecast (k_wls_mg.hip: pcg_part); NCT_WLS_FORECAST=0: fixed batches
    int wls_lines = 1;                          // block step of alternating line solves on the finest S2 level (k_wls_mg.hip: k_mg_block; oracle: mg_block_step). NCT_S2_LINES=0: the cycle without it (other arithmetic; comparison only)
    int wls_graph = 0;                          // experiment hook (env NCT_WLS_GRAPH=1): replay the PCG iteration batch as a HIP graph
    int conv_pool_fuse = -1;                    // VGG: 2x2 max-pool inside the conv epilogue: -1 = where the tile shape fits the map (nctk_conv3x3_pool_fits), 0 never, 1 always (NCT_CONV_POOL_FUSE; tests)
    double wls_rtol = 3e-8;                     // relative residual at which the WLS solve stops. The loosest tolerance at which the 8-bit result of every level equals the EXACT solve's on the
                                                // 700x700, mixed and 1000x1000 fixtures: 1e-7 with the point smoother of rounds 3-5a (1e-6: 55.4 / 50.1 dB), 3e-8 with the block step (the same residual norm
                                                // leaves more low-frequency error: 5e-8 differs in 53 bytes on the mixed pair; profiles/round5_wls_rtol_sweep.json). Experiment hook: env NCT_WLS_RTOL
    int wls_maxit = 5000;                       // iteration budget of the WLS solve (test hook: env NCT_WLS_MAXIT)
    bool tm_on = false;
    std::vector<hipEvent_t> tm_events;          // pool, reused across pairs
    std::vector<double> tm_host;                // host clock (us) at mark i: NCT_HOST_TRACE=1 prints it beside the GPU clock (how far the host runs ahead)
    std::vector<int> tm_tags;                   // tag of mark i = the stage that ENDS at event i
    int mark(hipStream_t s, int tag);           // nct_api.cpp; no-op unless tm_on
    // kernel clock (NCT_FLAG_TIME_KERNELS): event pairs around single launches of the full-resolution colour-solver kernels; sample i = events 2i, 2i+1, id kt_ids[i]
    int conv_pair = 1;                          // conv5_1 of the source and the reference in one launch (k_vgg.hip: nctk_conv3x3_pair); NCT_CONV_PAIR=0: two launches
    int* s1_hub_blocks() { return (int*)((char*)pinned + 4096); }   // [5][2] hub block and super-block count of each pyramid level's kNN graph (k_s1.hip), written by the side stream behind the WLS solver's 4 KB
    long long s1_hub_blocks_last[5] = {0, 0, 0, 0, 0};            // the counts the last pair's solves were launched with (-1: not known when the solve was enqueued); nct_ctx_counter
    int knn_runs = -1;                          // kNN search form: -1 = one search per (cluster, colour) run where runs average > 2.5 entries, decided on the device; 0 / 1 = NCT_KNN_RUNS (tests)
    int s1_hub_hint = 1;                        // use the host-side hub block counts (NCT_S1_HUB_HINT=0: always launch the hub pass — the conservative path, for tests)
    int s1_hub_wait = 1;                        // level 0 only: the host waits for the coarsest graph's event instead of launching 101 + 101 hub passes blind (NCT_S1_HUB_WAIT=0: no host wait inside a pair;
                                                // measured in round 6, profiles/round6_ab.md: waiting is 0.8 ms faster per single pair and +0.5-1 % with four in flight)
    bool kt_on = false;
    std::vector<hipEvent_t> kt_events; std::vector<int> kt_ids;
    int kt_begin(hipStream_t s, int id);        // nct_api.cpp; no-ops unless kt_on
    int kt_end(hipStream_t s);

    int fail(int code, const char* fmt, ...) {
        char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf; return code;
    }
    void* alloc(size_t bytes);        // never returns null on success; sets err and returns null on failure
    void release(void* p);
    // Arena blocks are recycled in stream order, which is only safe on ONE stream. While work is being enqueued on the side
    // stream, set defer_release: blocks released meanwhile stay reserved until flush_deferred() is called after the main
    // stream has been made to wait on the side stream's completion event.
    bool defer_release = false;
    std::vector<void*> deferred;
    void flush_deferred() { for (void* p : deferred) for (auto& b : blocks) if (b.p == p) { b.used = false; break; } deferred.clear(); }
};

// RAII scratch buffer from the context arena
template <typename T> struct DevBuf {
    nct_ctx* c; T* p;
    DevBuf(nct_ctx* ctx, size_t n) : c(ctx), p((T*)ctx->alloc(n * sizeof(T))) {}
    ~DevBuf() { if (p) c->release(p); }
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    operator T*() const { return p; }
    bool ok() const { return p != nullptr; }
};

#define NCT_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return ctx->fail(NCT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NCT_LAUNCH_CHECK() NCT_HIP(hipGetLastError())
#define NCT_REQUIRE(cond, ...) do { if (!(cond)) return ctx->fail(NCT_ERR_INVALID, __VA_ARGS__); } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device-side launchers (all on device pointers, features channel-last HWC fp32) ----
// k_feat.hip
int nctk_chw_to_hwc(nct_ctx* ctx, hipStream_t s, const float* src, float* dst, int C, int HW);
int nctk_hwc_to_chw(nct_ctx* ctx, hipStream_t s, const float* src, float* dst, int C, int HW);
int nctk_normalize(nct_ctx* ctx, hipStream_t s, const float* src_hwc, float* dst_hwc, float* resp /*nullable*/, int C, int HW,
                   void* dst_h16 = nullptr /* nullable: fp16 (round-to-nearest) shadow copy of dst, same HWC layout */);
int nctk_feature_distance(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, float* err, int C, int HW);
// k_patchmatch.hip: after a stream synchronise — has a persistent PatchMatch level's watchdog fired since the last check? (NCT_ERR_HIP then; no-op without pm_persist)
int nctk_pm_check(nct_ctx* ctx);
// k_nnf.hip
int nctk_nnf_init(nct_ctx* ctx, hipStream_t s, uint32_t* nnf, int ah, int aw, int bh, int bw);
int nctk_nnf_upsample(nct_ctx* ctx, hipStream_t s, const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half);
// k_patchmatch.hip — nnf in/out, dist out; tmp buffers come from the arena
int nctk_patchmatch(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, int C, int ah, int aw, int bh, int bw,
                    int iters, int rs_max, uint32_t seed, uint32_t* nnf, float* dist, unsigned long long* eval_counter /*nullable*/);
// k_vgg.hip / nct_vgg.cpp
int nctk_conv3x3_pair(nct_ctx* ctx, hipStream_t s, const float* in1, int H1, int W1, const float* in2, int H2, int W2, const float* wp, const float* bias,
                      float* out1, float* out2, int Cin, int Cout, int relu, float* hwc1, float* hwc2);
// both images to conv5_1 (channel-last taps only), the last layer for both in one launch
int nctk_vgg19_forward_pair(nct_ctx* ctx, hipStream_t s, const uint8_t* d_bgr1, int H1, int W1, int stride1, float* const* taps_hwc1,
                            const uint8_t* d_bgr2, int H2, int W2, int stride2, float* const* taps_hwc2);
int nctk_vgg19_forward(nct_ctx* ctx, hipStream_t s, const uint8_t* d_bgr, int H, int W, int stride, int deepest_tap, float* const* d_taps_chw, int* dims,
                       float* const* d_taps_hwc = nullptr /* the same taps channel-last, written by the tap layers' epilogues */);
void nct_vgg_free(nct_ctx* ctx);
// k_cvt.hip
int nctk_bgr2lab(nct_ctx* ctx, hipStream_t s, const uint8_t* src, uint8_t* dst, size_t npix);
int nctk_lab2bgr(nct_ctx* ctx, hipStream_t s, const uint8_t* src, uint8_t* dst, size_t npix, int form = 0 /* 0 = piecewise form (default), 1 = plain-cube form: k_cvt.hip */);
int nctk_resize_u8c3(nct_ctx* ctx, hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
int nctk_resize_f64c3(nct_ctx* ctx, hipStream_t s, const double* src, int sh, int sw, double* dst, int dh, int dw);
void nct_cvt_free(nct_ctx* ctx);
void nct_pair_free(nct_ctx* ctx);   // nct_pipeline.cpp
int nct_stage_tag_nonlocal(); int nct_stage_tag_wls(); int nct_stage_tag_color();   // event-mark tags of the colour stage (nct_pipeline.cpp)
// k_cluster.hip
int nctk_kmeans_labels(nct_ctx* ctx, hipStream_t s, const float* feat_hwc_norm, int n, int C, int K, int iters, uint64_t seed, int* labels, int* nlabels_dev);
int nctk_knn_graph(nct_ctx* ctx, hipStream_t s, const uint8_t* lab_u8, int h, int w, const int* labels, int lh, int lw, int nlabels, const int* nlabels_dev /*nullable: overrides nlabels*/, int samples,
                   int* knn_id, double* knn_w);
// k_s1.hip — S1, the nonlocal truncated CG. The graph-only part of its system depends on the level's kNN graph alone and is built once per level (in the pipeline: on the
// side stream, right behind the graph); nseg_hint is what the HOST knows about the number of hub blocks when it enqueues the solve (-1: nothing, the hub pass is launched anyway)
#define NCT_S1_SEG 64                        // in-edges of a pixel summed one by one; every further block of this many is summed by a tree in the hub pass
struct nct_s1_graph {
    int n;
    double* iw2;                             // [n][8] squared nonlocal weight of every out-edge
    unsigned long long* starts;              // [n + 1] low half: start of the pixel's first block (<= 64 in-edges) in c_src / c_w; high half: index of its first hub block
    int* c_src; double* c_w;                 // compact arrays of the first blocks
    int* rev_start;                          // [n + 1] start of the pixel's in-edges in the target-sorted edge list
    int* rev_src; double* rev_w;             // that list by position (only entries behind a first block are written and read)
    int* seg_tgt; int* seg_e0;               // hub block table: target pixel, position of the block's first edge
    double* hub_part;                        // [blocks][6] block sums of the current operator pass
    int* sup_start;                          // [n + 1] index of the pixel's first SUPER-block (64 hub blocks; only pixels with more than 64 hub blocks have any); sup_start[n] = their number
    int* sup_b0;                             // [super-blocks] index of the super-block's first hub block (its last: min(b0 + 64, the pixel's last block))
    double* sup_part;                        // [super-blocks][6] super-block sums of the current operator pass
    int nseg_hint, nsup_hint;                // what the host knows about the two counts (-1: nothing yet)
};
struct nct_s1_graph_bufs {
    int n;
    DevBuf<double> iw2, c_w, rev_w, hub_part, sup_part; DevBuf<unsigned long long> starts; DevBuf<int> c_src, rev_start, rev_src, seg_tgt, seg_e0, sup_start, sup_b0;
    nct_s1_graph_bufs(nct_ctx* c, int n_) : n(n_), iw2(c, (size_t)8 * n_), c_w(c, (size_t)8 * n_), rev_w(c, (size_t)8 * n_), hub_part(c, ((size_t)n_ / 8 + 1) * 6), sup_part(c, ((size_t)n_ / 256 + 2) * 6),
                                            starts(c, (size_t)n_ + 1), c_src(c, (size_t)8 * n_), rev_start(c, (size_t)n_ + 1), rev_src(c, (size_t)8 * n_), seg_tgt(c, (size_t)n_ / 8 + 1),
                                            seg_e0(c, (size_t)n_ / 8 + 1), sup_start(c, (size_t)n_ + 1), sup_b0(c, (size_t)n_ / 256 + 2) {}
    bool ok() const { return iw2.ok() && c_w.ok() && rev_w.ok() && hub_part.ok() && sup_part.ok() && starts.ok() && c_src.ok() && rev_start.ok() && rev_src.ok() && seg_tgt.ok() && seg_e0.ok() && sup_start.ok() && sup_b0.ok(); }
    nct_s1_graph view(int hint, int hint2) const { return nct_s1_graph{n, iw2, starts, c_src, c_w, rev_start, rev_src, rev_w, seg_tgt, seg_e0, hub_part, sup_start, sup_b0, sup_part, hint, hint2}; }
};
int nctk_s1_graph_build(nct_ctx* ctx, hipStream_t s, const int* knn_id, const double* knn_w, double nonlocalWeight, const nct_s1_graph& g, int* nseg_pinned /*nullable: [2] hub blocks, super-blocks*/);
int nctk_s1_solve(nct_ctx* ctx, hipStream_t s, const nct_s1_graph& g, const int* knn_id, const double* weight, float dWeight, const uint8_t* s_lab_level,
                  const uint8_t* g_lab_level, const double* gx, const double* gy, int layer, int h, int w, double* x, int* cg_iters_host);
// k_colorsolve.hip
struct nct_color_params { double eps, nonlocal_weight, local_weight, wls_lambda_init, wls_alpha, k_num; };
struct nct_color_debug { double *ab_local, *ab_nonlocal, *ab_up, *rough, *ab_wls; int* cg_iters; int* wls_iters; };   // host pointers, all nullable
int nctk_local_color_transfer(nct_ctx* ctx, hipStream_t s, const float* err, const uint8_t* s_lab_level, const uint8_t* g_lab_level,
                              const uint8_t* s_lab_full, const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W,
                              const nct_color_params& prm, uint8_t* out_lab_full, const nct_color_debug* dbg,
                              const nct_s1_graph* graph = nullptr /* the level's prebuilt graph part of S1; null: built inside, on s */);
// k_wls_mg.hip
int nctk_wls_solve_mg(nct_ctx* ctx, hipStream_t s, double* X, const double* rough, const double* wx, const double* wy, int H, int W,
                      double rtol, int* iters_out);
// pm_mode: how candidate distances are evaluated (k_patchmatch.hip)
enum { NCT_PM_PLAIN = 0,      // fp32 candidate tiles, no rejection (any features)
       NCT_PM_ROWREJECT = 1,  // unit-norm features: exact row-wise early rejection (same NNF and distances as PLAIN) — the pipeline's default
       NCT_PM_FP16 = 2 };     // opt-in reduced-precision mode: fp16 candidate tiles (fp32 accumulate); results differ from fp32
int nctk_patchmatch_bidir(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, const void* a_h16, const void* b_h16, int C, int ah, int aw, int bh, int bw,
                          int iters, int rs_max, uint32_t seed_ab, uint32_t seed_ba, uint32_t* ann, float* annd, uint32_t* bnn, float* bnnd,
                          int pm_mode, unsigned long long* counters /*nullable, 4 slots*/);
// k_vote.hip
int nctk_bds_vote_features(nct_ctx* ctx, hipStream_t s, const uint32_t* ann, const uint32_t* bnn, const float* pin_hwc, float* pout_hwc, float* pw /*nullable*/,
                           int C, int ah, int aw, int bh, int bw, float w_coh, float w_comp);
int nctk_bds_vote_image(nct_ctx* ctx, hipStream_t s, const uint8_t* b_bgr, const uint32_t* ann, const uint32_t* bnn,
                        int ah, int aw, int bh, int bw, double w_coh, double w_comp, uint8_t* out_bgr);
int nctk_bds_vote_both(nct_ctx* ctx, hipStream_t s, const uint8_t* b_bgr, const float* pin_hwc, const uint32_t* ann, const uint32_t* bnn, int C,
                       int ah, int aw, int bh, int bw, double w_coh, double w_comp, uint8_t* out_bgr, float* pout_hwc);
