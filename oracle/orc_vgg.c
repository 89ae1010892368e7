/* oracle/orc_vgg.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the VGG19 feature extractor the reference runs through Caffe (V2 in SURVEY §8a):
 *   Classifier::Preprocess      Classifier.cpp:211-275  (u8 BGR -> float, minus mean (103.939,116.779,123.68), planar)
 *   ConvolutionLayer::Forward   code/src/caffe/layers/conv_layer.cpp:8-40, base_conv_layer.cpp:257-283 (im2col GEMM,
 *                               M=Cout, N=H*W, K=Cin*9 with K index = ci*9 + ky*3 + kx: util/im2col.cpp:19-56), bias add
 *   ReLULayer::Forward          layers/relu_layer.cpp:14-17  (max(x,0), in place => the tapped blobs are post-ReLU)
 *   PoolingLayer::Forward (MAX) layers/pooling_layer.cpp:127-179; ceil-mode output size pooling_layer.cpp:90-93
 *   net topology                demo/model/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt (16 conv, pools after 1_2/2_2/3_4/4_4)
 *
 * Pinned by: Caffe's own known-answer tests restated in tests/test_oracle_vgg.py (3x5 max-pool
 * test_pooling_layer.cpp:56-99, ceil shapes) and a torch-CPU conv2d/max_pool2d(ceil_mode) cross-check.
 *
 * The fp32 accumulation order is FIXED (so the GPU's f32 MFMA chain can reproduce it bit-for-bit):
 * one fmaf chain per output over k = ci*9 + ky*3 + kx ascending (Caffe's im2col K order), starting from 0,
 * out-of-image taps contribute fmaf(0,w,acc) == acc; then `+ bias`, then ReLU.
 * Blobs are CHW fp32 like Caffe's.
 */
#include "orc_common.h"

#define ORC_VGG_NCONV 16
static const int k_vgg_cin[ORC_VGG_NCONV]  = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512};
static const int k_vgg_cout[ORC_VGG_NCONV] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512};
/* pool AFTER conv index (0-based): conv1_2=1, conv2_2=3, conv3_4=7, conv4_4=11 */
static const int k_vgg_pool_after[ORC_VGG_NCONV] = {0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0};
/* taps: conv1_1=0, conv2_1=2, conv3_1=4, conv4_1=8, conv5_1=12 */
static const int k_vgg_tap_conv[5] = {0, 2, 4, 8, 12};

int orc_vgg_layer_cin(int i) { return k_vgg_cin[i]; }
int orc_vgg_layer_cout(int i) { return k_vgg_cout[i]; }

/* Classifier::Preprocess — planar float(v) - float(mean_c) */
void orc_vgg_preprocess(const uint8_t* bgr, int H, int W, float* out_chw) {
    const float mean[3] = {(float)103.939, (float)116.779, (float)123.68};
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < H * W; ++i) out_chw[(size_t)c * H * W + i] = (float)bgr[(size_t)i * 3 + c] - mean[c];
}

__attribute__((target_clones("arch=haswell", "default")))
void orc_conv3x3(const float* in, int Cin, int H, int W, const float* wgt /*[Cout][Cin][3][3]*/, const float* bias,
                 int Cout, float* out, int relu) {
    /* zero-padded copy so the inner x loop is branch-free; fmaf(0,w,acc) == acc exactly */
    const int Wp = W + 2, Hp = H + 2;
    float* pad = (float*)calloc((size_t)Cin * Hp * Wp, sizeof(float));
    for (int c = 0; c < Cin; ++c)
        for (int y = 0; y < H; ++y) memcpy(pad + ((size_t)c * Hp + y + 1) * Wp + 1, in + ((size_t)c * H + y) * W, sizeof(float) * W);
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int y = 0; y < H; ++y) {
            float acc[4096];
            for (int x = 0; x < W; ++x) acc[x] = 0.f;
            for (int ci = 0; ci < Cin; ++ci)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const float w = wgt[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
                        const float* row = pad + ((size_t)ci * Hp + y + ky) * Wp + kx;
                        for (int x = 0; x < W; ++x) acc[x] = fmaf(row[x], w, acc[x]);
                    }
            float* o = out + ((size_t)co * H + y) * W;
            const float b = bias[co];
            for (int x = 0; x < W; ++x) {
                float v = acc[x] + b;
                if (relu) v = v > 0.f ? v : 0.f;          /* std::max(x, 0) */
                o[x] = v;
            }
        }
    free(pad);
}

static int pool_out(int n) { return (int)ceil((double)(n - 2) / 2.0) + 1; }    /* pooling_layer.cpp:90-93, pad 0 */
int orc_pool_out_size(int n) { return pool_out(n); }

/* MAX pool 2x2 stride 2, ceil mode, window clipped to the image (pooling_layer.cpp:147-165) */
void orc_maxpool2x2(const float* in, int C, int H, int W, float* out) {
    const int Ho = pool_out(H), Wo = pool_out(W);
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int py = 0; py < Ho; ++py)
            for (int px = 0; px < Wo; ++px) {
                int hs = py * 2, ws = px * 2;
                int he = hs + 2 < H ? hs + 2 : H, we = ws + 2 < W ? ws + 2 : W;
                float m = -FLT_MAX;
                for (int y = hs; y < he; ++y)
                    for (int x = ws; x < we; ++x) { float v = in[((size_t)c * H + y) * W + x]; if (v > m) m = v; }
                out[((size_t)c * Ho + py) * Wo + px] = m;
            }
}

/* Generic stride/kernel max pool used only to restate Caffe's own known-answer test (kernel 2, stride 1). */
void orc_maxpool_generic(const float* in, int C, int H, int W, int k, int s, float* out, int* Ho_out, int* Wo_out) {
    const int Ho = (int)ceil((double)(H - k) / s) + 1, Wo = (int)ceil((double)(W - k) / s) + 1;
    for (int c = 0; c < C; ++c)
        for (int py = 0; py < Ho; ++py)
            for (int px = 0; px < Wo; ++px) {
                int hs = py * s, ws = px * s;
                int he = hs + k < H ? hs + k : H, we = ws + k < W ? ws + k : W;
                float m = -FLT_MAX;
                for (int y = hs; y < he; ++y)
                    for (int x = ws; x < we; ++x) { float v = in[((size_t)c * H + y) * W + x]; if (v > m) m = v; }
                out[((size_t)c * Ho + py) * Wo + px] = m;
            }
    *Ho_out = Ho; *Wo_out = Wo;
}

/* VGG19 forward up to tap `deepest_tap` (1 = conv1_1 … 5 = conv5_1). weights[i]/biases[i]: conv i in net order.
 * taps[t] (nullable) receives the post-ReLU blob of tap t+1 in CHW; dims[t] = {C,H,W}. */
void orc_vgg19_features(const uint8_t* bgr, int H, int W, const float* const* weights, const float* const* biases,
                        int deepest_tap, float* const* taps, int* dims /*[5][3]*/) {
    int h = H, w = W;
    float* cur = (float*)malloc(sizeof(float) * 3 * (size_t)H * W);
    orc_vgg_preprocess(bgr, H, W, cur);
    int last_conv = k_vgg_tap_conv[deepest_tap - 1];
    for (int i = 0; i <= last_conv; ++i) {
        float* nxt = (float*)malloc(sizeof(float) * (size_t)k_vgg_cout[i] * h * w);
        orc_conv3x3(cur, k_vgg_cin[i], h, w, weights[i], biases[i], k_vgg_cout[i], nxt, 1);
        free(cur); cur = nxt;
        for (int t = 0; t < 5; ++t)
            if (k_vgg_tap_conv[t] == i) {
                if (dims) { dims[t * 3 + 0] = k_vgg_cout[i]; dims[t * 3 + 1] = h; dims[t * 3 + 2] = w; }
                if (taps && taps[t]) memcpy(taps[t], cur, sizeof(float) * (size_t)k_vgg_cout[i] * h * w);
            }
        if (k_vgg_pool_after[i] && i < last_conv) {
            int ho = pool_out(h), wo = pool_out(w);
            float* p = (float*)malloc(sizeof(float) * (size_t)k_vgg_cout[i] * ho * wo);
            orc_maxpool2x2(cur, k_vgg_cout[i], h, w, p);
            free(cur); cur = p; h = ho; w = wo;
        }
    }
    free(cur);
}
