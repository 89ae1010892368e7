/* include/nct.h — C ABI of libnct.so: the MI355X-native hot path of Neural-Color-Transfer.
 *
 * The reference (hmmlillian/Neural-Color-Transfer) has no plugin/FFI layer: the path is reached through
 * ordinary C++ calls and kernel launches inside `transfer_color_single_bds`
 * (code/windows/neural_color_transfer/source/main.cu:47-454). Each entry point below replaces one seam of
 * that function; the seam is cited as `main.cu:<line>` (+ the kernel/function it launches).
 *
 * Conventions
 *  - plain C, no C++/torch types. Every function returns 0 on success, <0 on error (nct_status);
 *    `nct_last_error(ctx)` returns a human-readable message for the last failure on that context.
 *  - a context is bound to ONE GPU and is NOT thread-safe; use one context per worker thread / process.
 *    The context owns every device allocation (cached arena, reused across calls and pairs).
 *  - "host" entry points take host pointers and are synchronous. Feature tensors are CHW fp32 exactly like the
 *    reference kernels' arguments (`float* a1`, Caffe blob layout). The `*_dev` variants (section "device-pointer seams"
 *    below) work on device pointers in the library's internal channel-last (HWC) layout and do not synchronise.
 *  - NNF element = uint32 `(y << 12) | x` (GeneralizedPatchMatch.cu:24-34), row-major (ah x aw).
 *  - there is NO CPU fallback: without a usable HIP device nct_create fails with NCT_ERR_NO_DEVICE.
 */
#ifndef NCT_H
#define NCT_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a struct of this header changes layout or an entry point changes meaning (round 6: 110 — nct_model_layer; nct_pair_timing grew in round 5 without a
 * bump). A caller checks `nct_version() == NCT_VERSION` before it passes any struct: the CLI and the python binding do. */
#define NCT_VERSION 110

typedef enum {
    NCT_OK = 0,
    NCT_ERR_NO_DEVICE = -1,     /* no HIP device / hipInit failed (fail loudly, no fallback) */
    NCT_ERR_INVALID = -2,       /* bad argument (null pointer, size out of range, unsupported C/patch) */
    NCT_ERR_HIP = -3,           /* a HIP runtime call or kernel launch failed */
    NCT_ERR_IO = -4,            /* file not found / unreadable / malformed */
    NCT_ERR_STATE = -5          /* call order (e.g. features before weights are loaded) */
} nct_status;

typedef struct nct_ctx nct_ctx;

/* ---- context (replaces cudaSetDevice/cudaDeviceReset + the two Classifier objects, main.cu:562-584) ---- */
int nct_create(int device, nct_ctx** out);
void nct_destroy(nct_ctx* ctx);
const char* nct_last_error(const nct_ctx* ctx);     /* ctx may be NULL: returns the last create() failure */
int nct_version(void);
int nct_device_name(nct_ctx* ctx, char* buf, int buflen);
/* no context needed: number of visible HIP devices, and a device's PCI address "dddd:bb:dd.f" — the key of /sys/bus/pci/devices/<addr>/numa_node and
 * local_cpulist, which the CLI pins a GPU's worker threads to ("worker thread pinned to the GPU's NUMA node") */
int nct_device_count(int* count);
int nct_device_pci_bus_id(int device, char* buf, int buflen);
int nct_synchronize(nct_ctx* ctx);
/* counters of a context (what library code must not print): NCT_CTR_ARENA_BYTES = device memory held by the context's arena; NCT_CTR_S1_HUB_BLOCKS_L0 + l (l = 0 coarsest … 4) =
 * the number of in-edge blocks beyond a pixel's first 64 in level l's kNN graph of the LAST pair, as the host knew it when it enqueued that level's nonlocal solve
 * (0 on the synthetic pairs; thousands on natural photographs, where groups of one colour make kNN hubs — k_s1.hip; -1: not known in time, the hub pass was launched anyway). */
enum { NCT_CTR_ARENA_BYTES = 0, NCT_CTR_S1_HUB_BLOCKS_L0 = 1 };
int nct_ctx_counter(nct_ctx* ctx, int which, int64_t* out);

/* ---- N1: feature L2 normalisation — `norm` (GeneralizedPatchMatch.cu:237-283), called main.cu:265,274,313.
 * dst = src / sqrt(sum_c src^2) per pixel (no epsilon, like the reference). resp (nullable, H*W) receives the
 * min-max normalised response map (|x| - min)/(max - min). C must be a multiple of 4. */
int nct_feat_normalize(nct_ctx* ctx, const float* src_chw, float* dst_chw, float* resp, int C, int H, int W);

/* ---- N2: NNF initialisation / upsampling — init_Ann_kernel (:527-544), upSample_kernel (:546-580);
 * launched main.cu:232-233 and :240-250. */
int nct_nnf_init(nct_ctx* ctx, uint32_t* nnf, int ah, int aw, int bh, int bw);
int nct_nnf_upsample(nct_ctx* ctx, const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half);

/* ---- P1: PatchMatch — patchmatch_single (:677-831), launched main.cu:283-284 (S->R) and (R->S).
 * a/b: L2-normalised CHW features of the query / candidate image. nnf: in/out. dist: out (ah*aw), the
 * negative mean cosine of the best match. patch must be 3 (Config.h:70), C a multiple of 4.
 * Deterministic: Jacobi step per (iteration, jump), counter-based RNG keyed by `seed`. */
int nct_patchmatch(nct_ctx* ctx, const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw,
                   int patch, int iters, int rs_max, uint32_t seed, uint32_t* nnf, float* dist);

/* ---- B2: feature-domain BDS vote + matching error — avg_vote_bds_a/_b/avg_vote_bds (:1074-1202),
 * feature_distance (:833-855); launched main.cu:303-316.
 * pin: UN-normalised R features (C,bh,bw). pout: voted features (C,ah,aw). pw (nullable): accumulated weights. */
int nct_bds_vote_features(nct_ctx* ctx, const uint32_t* ann, const uint32_t* bnn, const float* pin_chw, float* pout_chw, float* pw,
                          int C, int ah, int aw, int bh, int bw, int patch, float w_coherence, float w_complete);
int nct_feature_distance(nct_ctx* ctx, const float* a_chw, const float* b_chw, float* err, int C, int H, int W);

/* ---- B1: image-domain BDS vote — reconstruct_bds (GeneralizedPatchMatch.cu:122-235), called main.cu:291.
 * a, b: u8 BGR HWC level images of S and R; out: guidance image G (ah x aw x 3). */
int nct_bds_vote_image(nct_ctx* ctx, const uint8_t* a_bgr, int ah, int aw, const uint8_t* b_bgr, int bh, int bw,
                       const uint32_t* ann, const uint32_t* bnn, int patch, double w_coherence, double w_complete, uint8_t* out_bgr);

/* ---- V1: VGG19 weights — Classifier::Classifier (Classifier.cpp:5-42) -> Net::CopyTrainedLayersFrom
 * (code/src/caffe/net.cpp:760-813). Reads a .caffemodel (protobuf wire format, V1 `layers`=2 as in the Oxford file, or
 * V2 `layer`=100), matches conv layers BY NAME, ignores unknown layers, fails on a shape mismatch. Only conv1_1..conv5_1
 * (13 layers) are required/uploaded. _load_raw takes Caffe-layout arrays [Cout][Cin][3][3] + [Cout] in net order. */
int nct_vgg19_load_caffemodel(nct_ctx* ctx, const char* path);
int nct_vgg19_load_raw(nct_ctx* ctx, const float* const* weights, const float* const* biases, int nlayers);
/* A driver with several contexts (the CLI's -gpus N -inflight K; main.cu:581-582 builds two Nets from the file): parse the 575 MB file ONCE per process into a
 * host-side model (52 MB: the 13 needed layers), upload it ONCE per GPU (nct_vgg19_load_model), and let the other contexts of that GPU use the same read-only
 * device copy (nct_vgg19_share_weights; both contexts must live on the same device; the copy is freed with its last user). nct_model_* need no context;
 * nct_model_last_error() returns the message of the last failed parse on the calling thread. nct_vgg19_weights_info: identity (device address of conv1_1's
 * packed weights), size in bytes and number of contexts sharing this context's copy. */
typedef struct nct_model nct_model;
int nct_model_parse_caffemodel(const char* path, nct_model** out);
void nct_model_free(nct_model* m);
/* read access to a parsed model (host memory, valid until nct_model_free): layer 0..12 = conv1_1 .. conv5_1 in net order; weights [cout][cin][3][3], bias [cout].
 * What Net::CopyTrainedLayersFrom leaves in the layer's blobs (net.cpp:776-792) — lets a caller (and the CPU tests) check an ingest without a device. */
int nct_model_layer(const nct_model* m, int layer, const float** weights, const float** bias, int* cout, int* cin);
const char* nct_model_last_error(void);
int nct_vgg19_load_model(nct_ctx* ctx, const nct_model* m);
int nct_vgg19_share_weights(nct_ctx* ctx, nct_ctx* from);
int nct_vgg19_weights_info(nct_ctx* ctx, uint64_t* id, size_t* bytes, int* sharers);
/* Classifier::Classifier builds its Net from <model_dir>/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt (Classifier.cpp:16; main.cu:575-577). This library's
 * topology is built in, so the file is only checked (protobuf text format, `layer` or V1 `layers` messages): conv1_1..conv5_1 in order with the built-in
 * channel counts, 3x3 / pad 1 / stride 1, each followed by an in-place ReLU, 2x2 / stride-2 MAX pools after conv1_2, 2_2, 3_4, 4_4. Anything else ->
 * NCT_ERR_IO with a message (ctx nullable: then through nct_model_last_error). Layers behind relu5_1 are not looked at. */
int nct_vgg19_check_prototxt(nct_ctx* ctx, const char* path);

/* ---- V2/R1: VGG19 features — Classifier::Predict (Classifier.cpp:59-143), called main.cu:94,102,426.
 * bgr: u8 BGR HWC image (row stride in bytes). Runs preprocess (mean subtraction, Classifier.cpp:211-275) and the net
 * up to tap `deepest_tap` (1 = conv1_1 … 5 = conv5_1; the reference always runs to pool5, SURVEY quirk 9 — the
 * result is identical). taps_chw[t] (nullable entries) receives the post-ReLU blob of tap t+1 in Caffe's CHW layout;
 * dims (nullable, int[15]) receives {C,H,W} per tap. */
int nct_vgg19_features(nct_ctx* ctx, const uint8_t* bgr, int h, int w, int stride, int deepest_tap, float* const* taps_chw, int* dims);

/* single layers (unit parity against Caffe's known-answer tests): 3x3 pad-1 stride-1 conv + bias (+ ReLU) on f32 MFMA
 * (conv_layer.cpp:8-40; Cout % 64 == 0), 2x2/2 MAX pool with ceil-mode size (pooling_layer.cpp:90-93,147-165). */
int nct_conv3x3_relu(nct_ctx* ctx, const float* in_chw, int Cin, int H, int W, const float* weights, const float* bias, int Cout, float* out_chw, int relu);
int nct_maxpool2x2(nct_ctx* ctx, const float* in_chw, int C, int H, int W, float* out_chw);

/* ---- parameters: `Config` (ColorTransfer/Config.h:55-98) + the constants hard-coded in transfer_color_single_bds
 * (main.cu:64-68: iter = 10). nct_params_default fills Config::Config()'s values (bds 2.0, eps 0.60, nl 2.0, l 0.125,
 * w 0.024, clusters 10, k 8, patch 3, alpha 1.2) — the help strings at main.cu:40-43 quote different numbers; the
 * constructor is what runs. `seed` keys the counter-based RNG (PatchMatch random search, k-means centres). */
typedef struct nct_params {
    double bds_weight, eps, nonlocal_weight, local_weight, wls_lambda_init;
    int cluster_num, k_num, patch_size;
    double wls_alpha;
    int pm_iters;
    uint32_t seed;
    int levels;         /* pyramid levels to run, coarse -> fine: 5 = the reference's full L=5..1 loop (main.cu:179); 1 = "L=5 only"
                           (BASELINE config 1): the result is the full-resolution image after the coarsest level's colour transfer */
    uint32_t flags;     /* NCT_FLAG_* (extensions; 0 = reference behaviour) */
} nct_params;
#define NCT_FLAG_FEAT16      1u   /* opt-in reduced precision: PatchMatch reads fp16 candidate feature tiles (fp32 accumulate). Halves the
                                     dominant kernel's bytes; the result is NOT bit-identical to the fp32 path (report PSNR against it) */
#define NCT_FLAG_COUNT_EVALS 2u   /* profiling: count PatchMatch evaluations per level on the device (nct_pair_timing.pm_level_evals …) */
#define NCT_FLAG_LATENCY     4u   /* one pair in flight on this GPU: spend extra launches on its latency — the a- and b-halves of each WLS solve
                                     run concurrently on two streams. Same result bit for bit; with several pairs in flight it only adds launches */
#define NCT_FLAG_LAB2BGR_CUBE 8u  /* CV_Lab2BGR in the older plain-cube form of OpenCV's Lab2RGB_f (no linear branch, no clipping) instead of the default piecewise
                                     form — see nct_lab2bgr_u8_form for which one the reference's own result images show */
#define NCT_FLAG_TIME_KERNELS 16u /* profiling: HIP events around single launches of the finest level's colour-solver kernels (nct_pair_timing.kernel_us); the
                                     extra events perturb the pair a little: use a run of its own */
void nct_params_default(nct_params* p);

/* ---- A1 + third-party (OpenCV 2.4.10) arithmetic used on the path: cvtColor(CV_BGR2Lab / CV_Lab2BGR) on 8U
 * (main.cu:352,371; ColorTransfer.h:58; ColorTransfer.cpp:1469) and cv::resize(INTER_LINEAR) on 8UC3 / 64FC3
 * (main.cu:106-107; ColorTransfer.cpp:462-463; includes cv::resize's silent INTER_AREA switch for an exact 2x shrink). */
int nct_bgr2lab_u8(nct_ctx* ctx, const uint8_t* bgr, size_t npix, uint8_t* lab);
int nct_lab2bgr_u8(nct_ctx* ctx, const uint8_t* lab, size_t npix, uint8_t* bgr);      /* = form NCT_LAB2BGR_PIECEWISE */
/* The two forms of Lab2RGB_f in OpenCV's history (the sources of the pinned 2.4.10 cannot be inspected here — SURVEY App. A):
 * NCT_LAB2BGR_PIECEWISE (default): CIE's linear branch below L* = 8 / f = 6/29 and linear RGB clipped to [0, 1]; NCT_LAB2BGR_CUBE: fY = (L+16)/116,
 * fX = fY + a/500, fZ = fY - b/200, all cubed, linear RGB NOT clipped (the inverse-gamma spline extrapolates, the final cast saturates).
 * The default is decided by the reference's own artefacts: the cube form cannot produce a pixel darker than (9, 9, 9) on the grey axis nor (0, 0, 0) at
 * all (Y >= (16/116)^3 > 0), yet demo/example/res/in0_tar0_2.00.png holds 415 pixels with all channels < 5 incl. (0, 0, 0) and every result image with
 * dark regions has pixels like (0, 2, 0) = the linear branch at L_u8 = 1..2 (tests/golden/demo_res_dark_stats.json). */
#define NCT_LAB2BGR_PIECEWISE 0
#define NCT_LAB2BGR_CUBE 1
int nct_lab2bgr_u8_form(nct_ctx* ctx, const uint8_t* lab, size_t npix, uint8_t* bgr, int form);
int nct_resize_u8c3(nct_ctx* ctx, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
int nct_resize_f64c3(nct_ctx* ctx, const double* src, int sh, int sw, double* dst, int dh, int dw);

/* ---- C1: k-means labels of the coarsest S features — main.cu:139-168 -> ColorTransfer::clusterFeastures
 * (ColorTransfer.cpp:355-395; cvflann k-means, branching K, `iters` Lloyd steps). feat_chw: UN-normalised conv5_1 of S;
 * labels: h*w ints in [0, *nlabels). */
int nct_cluster_features(nct_ctx* ctx, const float* feat_chw, int C, int h, int w, int K, int iters, uint64_t seed, int* labels, int* nlabels);

/* ---- K1: kNN graph in Lab — ColorTransfer::findKnns (ColorTransfer.cpp:397-423), called main.cu:359.
 * lab_u8: level image in 8-bit Lab (HWC); labels: coarsest-level cluster labels (lh x lw); samples = 2^level.
 * knn_id / knn_w: [h*w][k] neighbour ids and weights exp(1 - d/3); k must be 8. */
int nct_knn_graph(nct_ctx* ctx, const uint8_t* lab_u8, int h, int w, const int* labels, int lh, int lw, int nlabels, int samples, int k,
                  int* knn_id, double* knn_w);

/* ---- T1,T2,S1,U1,S2,A1: build_accumTable_downsample x2 + transfer_color_downsample + getRes
 * (ColorTransfer.cpp:425-455,1180-1478), called main.cu:368-380. err: matching-error map of the level (h x w);
 * s_bgr_level / g_bgr_level: level images of S and the BDS guidance G; s_bgr_full: S at full resolution (H x W);
 * layer = 0 (coarsest) … 4 (finest). out: recoloured S (H x W x 3, BGR). `stages` (nullable) receives intermediate
 * coefficient maps for stage-wise validation: arrays are [2][pixels][3] doubles (a-part then b-part). */
typedef struct nct_color_stages {
    double* ab_local;      /* [2][h*w][3]  after the local statistics (T1)            */
    double* ab_nonlocal;   /* [2][h*w][3]  after the truncated CG (S1)                */
    double* ab_up;         /* [2][H*W][3]  after bilinear upsampling (U1)             */
    double* roughness;     /* [H*W]                                                   */
    double* ab_wls;        /* [2][H*W][3]  after WLS smoothing (S2)                   */
    int* cg_iters;         /* [3]  CG iterations executed per Lab channel             */
    int* wls_iters;        /* [6]  PCG iterations per right-hand side                 */
} nct_color_stages;
int nct_local_color_transfer(nct_ctx* ctx, const float* err, const uint8_t* s_bgr_level, const uint8_t* g_bgr_level, const uint8_t* s_bgr_full,
                             const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W, const nct_params* prm,
                             uint8_t* out_bgr_full, nct_color_stages* stages);

/* ---- D3: the per-pair hot loop — transfer_color_single_bds (main.cu:47-454): VGG19 features of S and R, k-means of
 * S's conv5_1, then for L = 5..1: NNF init/upsample, normalise, PatchMatch both ways, BDS votes, matching error, kNN
 * graph, local colour transfer, re-predict. Images are 8-bit BGR, tightly packed HWC. out has the size of src.
 * `timing` (nullable) receives per-stage device milliseconds under the reference's own stage names (main.cu:331,453;
 * ColorTransfer.cpp:1373,1434), taken from events on the stream (no extra host synchronisation).
 * nct_process_pair = nct_pair_upload + nct_pair_run + nct_pair_download; the split form lets a caller keep inputs
 * resident in HBM (bench.py times nct_pair_run only and reports the PCIe-inclusive rate separately). */
typedef struct nct_pair_timing {
    double total_ms;                    /* host wall time of nct_pair_run */
    /* device time between stage-boundary events on the pair's main stream; read once after the pair has finished, so asking for
       timing does not add host synchronisation (the kNN graphs run on a side stream: knn_ms is only what the main stream waited) */
    double vgg_ms, cluster_ms, patchmatch_ms, vote_ms, knn_ms, color_ms, other_ms;
    double nonlocal_ms, wls_ms;         /* parts of color_ms: S1 = "Nonlocal Solve Time" (ColorTransfer.cpp:1373), S2 = "WLS Solve Time" (:1434) */
    int wls_iters[5];                   /* PCG iterations of the WLS solve per level */
    /* PatchMatch per pyramid level (0 = coarsest): kernel time of the level's launches (init + pm_iters*4 steps, both directions per launch) */
    double pm_level_ms[5];
    int pm_level_launches[5];
    double vote_level_ms[5], nonlocal_level_ms[5], wls_level_ms[5];   /* the same split per level, for the reference's per-level log lines */
    unsigned long long pm_level_evals[5], pm_level_accepted[5];   /* NCT_FLAG_COUNT_EVALS only, else 0: distance evaluations, accepted candidates */
    /* NCT_FLAG_TIME_KERNELS only, else 0: average microseconds per launch (HIP events on the pair's stream around kernel_samples[i] single launches) of the
       colour-solver kernels at full resolution — index NCT_KT_*: the operator pass, the one-workgroup scalar step and the vector pass of the finest level's truncated CG (S1), and the WLS
       PCG's finest V-cycle legs, operator + dot products, vector update, and everything below the finest level of one V-cycle */
    double kernel_us[10];
    int kernel_samples[10];
} nct_pair_timing;
enum { NCT_KT_S1_APPLY = 0, NCT_KT_S1_SCALARS = 1, NCT_KT_S1_UPDATE = 2, NCT_KT_WLS_DOWN = 3, NCT_KT_WLS_UP = 4, NCT_KT_WLS_APPLY = 5, NCT_KT_WLS_UPDATE = 6, NCT_KT_WLS_COARSE = 7,
       NCT_KT_WLS_BLOCK_PRE = 8, NCT_KT_WLS_BLOCK_POST = 9 /* the block step of alternating line solves in front of the finest down leg / behind the finest up leg */ };
/* per-level intermediates for level-wise validation (all pointers nullable; level 0 = coarsest … 4 = finest; arrays have the level's
 * size ah*aw / bh*bw except `result`, the full-resolution intermediate result after that level, H*W*3 like level_out of the oracle) */
typedef struct nct_pair_levels {
    uint32_t* ann[5]; uint32_t* bnn[5];
    float* annd[5]; float* bnnd[5];
    uint8_t* guide[5];
    float* err[5];
    uint8_t* result[5];
    const nct_color_stages* color[5];   /* the colour stage's coefficient maps of the level (T1 / S1 / U1 / S2), as nct_local_color_transfer returns them */
    int* labels;                        /* [ah0*aw0] k-means labels of S's deepest features (the kNN graphs' clusters); level 0's size */
} nct_pair_levels;
int nct_process_pair(nct_ctx* ctx, const uint8_t* src_bgr, int sh, int sw, const uint8_t* ref_bgr, int rh, int rw, const nct_params* prm,
                     uint8_t* out_bgr, nct_pair_timing* timing);
int nct_pair_upload(nct_ctx* ctx, const uint8_t* src_bgr, int sh, int sw, const uint8_t* ref_bgr, int rh, int rw);
int nct_pair_run(nct_ctx* ctx, const nct_params* prm, nct_pair_timing* timing);
int nct_pair_run_levels(nct_ctx* ctx, const nct_params* prm, nct_pair_timing* timing, const nct_pair_levels* levels);   /* + host copies of the intermediates */
int nct_pair_download(nct_ctx* ctx, uint8_t* out_bgr);

/* ---- device-pointer seams: the same operations on buffers that stay in HBM between calls (main.cu:204-316 keeps Ndata_C1, ann_device, ... on the device
 * across these kernels; an integrator replacing single seams should not pay H2D + D2H + a synchronise per call). Buffers come from the context's arena
 * (nct_dev_alloc / nct_dev_free; any device pointer of the context's GPU works as an operand). Calls are enqueued on the context's stream in call order and return
 * immediately; nct_dev_download and nct_synchronize wait. Arena blocks are recycled in the order of THAT stream: use them from another stream (or free them while your own
 * kernels still read them) only after nct_synchronize. nct_dev_free of a pointer that is not a live nct_dev_alloc block of the context returns NCT_ERR_INVALID. Features: channel-last HWC fp32 (nct_chw_to_hwc_dev converts a Caffe blob once); `unit_norm` = 1
 * tells nct_patchmatch_bidir_dev that both maps hold unit vectors (output of nct_feat_normalize_dev), which enables the exact row-wise rejection.
 * Chained like the reference's level loop they reproduce nct_pair_run_levels bit for bit (tests/test_gpu_pipeline.py::test_dev_seams_chain_equals_pipeline). */
int nct_dev_alloc(nct_ctx* ctx, size_t bytes, void** out);
int nct_dev_free(nct_ctx* ctx, void* p);
int nct_dev_upload(nct_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int nct_dev_download(nct_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int nct_chw_to_hwc_dev(nct_ctx* ctx, const float* src_chw, float* dst_hwc, int C, int H, int W);
int nct_hwc_to_chw_dev(nct_ctx* ctx, const float* src_hwc, float* dst_chw, int C, int H, int W);
int nct_vgg19_features_dev(nct_ctx* ctx, const uint8_t* d_bgr, int h, int w, int stride, int deepest_tap, float* const* d_taps_chw, int* dims);   /* main.cu:94,102,426 */
/* the same forward with the taps ALSO / INSTEAD channel-last (d_taps_hwc[t], nullable like d_taps_chw[t], which may itself be NULL): the tap layer's own epilogue writes the HWC map
 * the correspondence seams below read, so an integrator who does not need Caffe's planar blobs skips nct_chw_to_hwc_dev altogether */
int nct_vgg19_features_hwc_dev(nct_ctx* ctx, const uint8_t* d_bgr, int h, int w, int stride, int deepest_tap, float* const* d_taps_chw, float* const* d_taps_hwc, int* dims);
int nct_feat_normalize_dev(nct_ctx* ctx, const float* src_hwc, float* dst_hwc, float* resp, int C, int H, int W);                                    /* main.cu:265,274,313 */
int nct_nnf_init_dev(nct_ctx* ctx, uint32_t* nnf, int ah, int aw, int bh, int bw);                                                                    /* main.cu:232-233 */
int nct_nnf_upsample_dev(nct_ctx* ctx, const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half);           /* main.cu:240-250 */
int nct_patchmatch_dev(nct_ctx* ctx, const float* a_hwc, const float* b_hwc, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max,
                       uint32_t seed, uint32_t* nnf, float* dist);                                                                                     /* main.cu:283 */
int nct_patchmatch_bidir_dev(nct_ctx* ctx, const float* a_hwc, const float* b_hwc, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max,
                             uint32_t seed_ab, uint32_t seed_ba, uint32_t* ann, float* annd, uint32_t* bnn, float* bnnd, int unit_norm);              /* main.cu:283-284 */
int nct_bds_vote_features_dev(nct_ctx* ctx, const uint32_t* ann, const uint32_t* bnn, const float* pin_hwc, float* pout_hwc, float* pw, int C, int ah, int aw,
                              int bh, int bw, int patch, float w_coherence, float w_complete);                                                         /* main.cu:303-311 */
int nct_bds_vote_image_dev(nct_ctx* ctx, const uint8_t* b_bgr, const uint32_t* ann, const uint32_t* bnn, int ah, int aw, int bh, int bw, int patch,
                           double w_coherence, double w_complete, uint8_t* out_bgr);                                                                   /* main.cu:291 */
int nct_feature_distance_dev(nct_ctx* ctx, const float* a_hwc, const float* b_hwc, float* err, int C, int H, int W);                                  /* main.cu:316 */

/* ---- measurement hooks (bench.py / rocprof): device-resident PatchMatch on synthetic features ----
 * nct_pm_bench_setup uploads + normalises two CHW feature maps once; nct_pm_bench_run re-initialises the NNF
 * (scaled identity) and runs one full nct_patchmatch pass (init-dist + iters*4 Jacobi steps) entirely on the
 * device, returning the kernel time of that pass measured with HIP events on the library's own stream and the
 * number of distance evaluations actually performed (device counter). */
int nct_pm_bench_setup(nct_ctx* ctx, const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw);
int nct_pm_bench_run(nct_ctx* ctx, int iters, int rs_max, uint32_t seed, float* kernel_ms, uint64_t* evals, uint32_t* nnf_out, float* dist_out);
/* the pipeline's form of the same pass: both directions fused per launch; pm_mode 0 = fp32 tiles, 1 = fp32 tiles + exact row-wise
 * rejection for unit-norm features (the pipeline's default), 2 = fp16 tiles (NCT_FLAG_FEAT16). counters (nullable, 2 x uint64):
 * evaluations, accepted candidates. */
int nct_pm_bench_run_bidir(nct_ctx* ctx, int iters, int rs_max, uint32_t seed, int pm_mode, float* kernel_ms, uint64_t* counters, uint32_t* ann_out, float* annd_out,
                           uint32_t* bnn_out, float* bnnd_out);

#ifdef __cplusplus
}
#endif
#endif /* NCT_H */
