// nct_vgg.cpp — VGG19 weights (caffemodel ingest, V1) and the feature forward (V2/R1) on top of k_vgg.hip.
// Reference: Classifier::Classifier (Classifier.cpp:5-42) -> Net::CopyTrainedLayersFrom (code/src/caffe/net.cpp:760-813):
// layers are matched BY NAME, unknown source layers (fc6..8, relu, pool, …) are ignored, a shape mismatch is fatal;
// wire schema code/src/caffe/proto/caffe.proto (NetParameter :64-100, V1LayerParameter :1287-1335, LayerParameter
// :311-329, BlobProto :6-22). Classifier::Predict (Classifier.cpp:59-143) = preprocess + forward + return named blobs.
// The 575 MB file is parsed ONCE per context (the reference builds two Nets from it, main.cu:581-582).
#include "nct_internal.h"
#include <cstring>
#include <cstdlib>
#include <memory>
#include <cctype>
#include <map>
#include <cerrno>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

int nctk_conv3x3(nct_ctx*, hipStream_t, const float* in, const float* wp, const float* bias, float* out, int Cin, int Cout, int H, int W, int relu, int pool, float* out_hwc = nullptr);
bool nctk_conv3x3_pool_fits(int H, int W);
int nctk_maxpool2x2(nct_ctx*, hipStream_t, const float* in, float* out, int C, int H, int W);
int nctk_vgg_preprocess(nct_ctx*, hipStream_t, const uint8_t* bgr, int stride, float* out, int H, int W);
int nctk_pack_weights(nct_ctx*, hipStream_t, const float* w, float* wp, int Cout, int Cin, int Cin_pad);

static const int NCONV = 16;
static const char* const kConvName[NCONV] = {"conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4",
                                             "conv4_1", "conv4_2", "conv4_3", "conv4_4", "conv5_1", "conv5_2", "conv5_3", "conv5_4"};
static const int kCin[NCONV]  = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512};
static const int kCout[NCONV] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512};
static const bool kPoolAfter[NCONV] = {false, true, false, true, false, false, false, true, false, false, false, true, false, false, false, false};
static const int kTapConv[5] = {0, 2, 4, 8, 12};     // conv1_1, conv2_1, conv3_1, conv4_1, conv5_1
static const int kNeeded = 13;                       // conv5_2..conv5_4 are never needed

// Device copy of the packed weights. Read-only after loading, so every context on the same GPU can use ONE copy (nct_vgg19_share_weights: the CLI
// with -inflight K keeps one 52 MB copy per GPU instead of K; the reference keeps two Nets per process, main.cu:581-582). Freed with its last user.
struct vgg_weights {
    int device = 0;
    float* wp[NCONV] = {nullptr};     // packed [Cin_pad*9][Cout]
    float* bias[NCONV] = {nullptr};
    size_t bytes = 0;
    bool loaded = false;
    ~vgg_weights() {
        int cur = 0; (void)hipGetDevice(&cur); (void)hipSetDevice(device);
        for (int i = 0; i < NCONV; ++i) { if (wp[i]) (void)hipFree(wp[i]); if (bias[i]) (void)hipFree(bias[i]); }
        (void)hipSetDevice(cur);
    }
};
struct vgg_holder { std::shared_ptr<vgg_weights> w; };

static vgg_weights* vgg_of(nct_ctx* ctx) {
    if (!ctx->vgg) { auto* h = new vgg_holder(); h->w = std::make_shared<vgg_weights>(); h->w->device = ctx->device; ctx->vgg = h; }
    return ((vgg_holder*)ctx->vgg)->w.get();
}
// a context about to (re)load weights must not write into a copy other contexts read: detach to a fresh one
static vgg_weights* vgg_own(nct_ctx* ctx) {
    if (ctx->vgg && ((vgg_holder*)ctx->vgg)->w.use_count() > 1) { delete (vgg_holder*)ctx->vgg; ctx->vgg = nullptr; }
    return vgg_of(ctx);
}

static int upload_layer(nct_ctx* ctx, int i, const float* w, const float* b) {
    vgg_weights* v = vgg_of(ctx);
    const int cin_pad = (kCin[i] + 1) & ~1;
    const size_t nw = (size_t)kCout[i] * kCin[i] * 9;
    if (!v->wp[i]) { NCT_HIP(hipMalloc(&v->wp[i], sizeof(float) * (size_t)cin_pad * 9 * kCout[i])); v->bytes += sizeof(float) * (size_t)cin_pad * 9 * kCout[i]; }
    if (!v->bias[i]) { NCT_HIP(hipMalloc(&v->bias[i], sizeof(float) * kCout[i])); v->bytes += sizeof(float) * kCout[i]; }
    DevBuf<float> tmp(ctx, nw);
    if (!tmp.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemcpyAsync(tmp, w, sizeof(float) * nw, hipMemcpyHostToDevice, ctx->stream));
    int rc = nctk_pack_weights(ctx, ctx->stream, tmp, v->wp[i], kCout[i], kCin[i], cin_pad);
    if (rc) return rc;
    NCT_HIP(hipMemcpyAsync(v->bias[i], b, sizeof(float) * kCout[i], hipMemcpyHostToDevice, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------- minimal protobuf wire reader
struct PB { const uint8_t* p; const uint8_t* end; bool ok = true;
    bool more() const { return ok && p < end; }
    uint64_t varint() { uint64_t v = 0; int sh = 0; while (p < end && sh < 64) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; } ok = false; return 0; }
    bool tag(uint32_t& field, uint32_t& wt) { if (!more()) return false; uint64_t t = varint(); field = (uint32_t)(t >> 3); wt = (uint32_t)(t & 7); return ok; }
    PB sub() { uint64_t n = varint(); if (!ok || n > (uint64_t)(end - p)) { ok = false; return PB{p, p}; } PB s{p, p + n}; p += n; return s; }
    void skip(uint32_t wt) {
        switch (wt) { case 0: varint(); break; case 1: if (end - p < 8) ok = false; else p += 8; break;
                      case 2: sub(); break; case 5: if (end - p < 4) ok = false; else p += 4; break; default: ok = false; }
    }
};

struct BlobView { int64_t dims[4] = {0, 0, 0, 0}; int ndim = 0; std::vector<float> data; };

static bool parse_blob(PB b, BlobView& out) {
    int64_t legacy[4] = {0, 0, 0, 0}; bool has_legacy = false;
    uint32_t f, wt;
    while (b.tag(f, wt)) {
        if (f >= 1 && f <= 4 && wt == 0) { legacy[f - 1] = (int64_t)b.varint(); has_legacy = true; }
        else if (f == 5 && wt == 2) { PB d = b.sub(); size_t n = (size_t)(d.end - d.p) / 4; size_t o = out.data.size(); out.data.resize(o + n); memcpy(out.data.data() + o, d.p, n * 4); }
        else if (f == 5 && wt == 5) { if (b.end - b.p < 4) return false; float v; memcpy(&v, b.p, 4); b.p += 4; out.data.push_back(v); }
        else if (f == 8 && wt == 1) { if (b.end - b.p < 8) return false; double v; memcpy(&v, b.p, 8); b.p += 8; out.data.push_back((float)v); }      // double_data, not packed
        else if (f == 8 && wt == 2) { PB d = b.sub(); size_t n = (size_t)(d.end - d.p) / 8; for (size_t i = 0; i < n; ++i) { double v; memcpy(&v, d.p + 8 * i, 8); out.data.push_back((float)v); } }
        else if (f == 7 && wt == 2) {   // BlobShape { repeated int64 dim = 1 [packed] }
            PB s = b.sub(); uint32_t f2, w2; out.ndim = 0;
            while (s.tag(f2, w2)) {
                if (f2 == 1 && w2 == 2) { PB d = s.sub(); while (d.more() && out.ndim < 4) out.dims[out.ndim++] = (int64_t)d.varint(); }
                else if (f2 == 1 && w2 == 0) { if (out.ndim < 4) out.dims[out.ndim++] = (int64_t)s.varint(); else s.varint(); }
                else s.skip(w2);
            }
        } else b.skip(wt);
    }
    if (out.ndim == 0 && has_legacy) { out.ndim = 4; for (int i = 0; i < 4; ++i) out.dims[i] = legacy[i]; }
    return b.ok;
}

static int64_t blob_count(const BlobView& b) { int64_t n = 1; for (int i = 0; i < b.ndim; ++i) n *= b.dims[i]; return b.ndim ? n : 0; }

// Host copy of the 13 needed conv layers (52 MB): what a process parses ONCE from the 575 MB caffemodel and then uploads to every GPU it drives.
struct nct_model {
    std::vector<float> w[NCONV], b[NCONV];
    std::string err;
};
static thread_local std::string g_model_err;

// parses `path` into m; returns NCT_OK or an error code with the message in `err`
static int parse_caffemodel(const char* path, nct_model& m, std::string& err) {
    char buf[512];
    auto fail = [&](int code, const char* fmt, auto... a) { snprintf(buf, sizeof buf, fmt, a...); err = buf; return code; };
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(NCT_ERR_IO, "cannot open caffemodel '%s': %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); return fail(NCT_ERR_IO, "cannot stat caffemodel '%s'", path); }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(NCT_ERR_IO, "mmap of '%s' failed: %s", path, strerror(errno));
    bool found[NCONV] = {false};
    int rc = NCT_OK;
    PB net{(const uint8_t*)map, (const uint8_t*)map + st.st_size};
    uint32_t f, wt;
    while (rc == NCT_OK && net.tag(f, wt)) {
        const bool v1 = (f == 2 && wt == 2), v2 = (f == 100 && wt == 2);   // NetParameter.layers (V1) / .layer (V2)
        if (!v1 && !v2) { net.skip(wt); continue; }
        PB layer = net.sub();
        const uint32_t name_field = v1 ? 4 : 1, blobs_field = v1 ? 6 : 7;
        std::string name; std::vector<PB> blobs;
        uint32_t lf, lw;
        while (layer.tag(lf, lw)) {
            if (lf == name_field && lw == 2) { PB s = layer.sub(); name.assign((const char*)s.p, (size_t)(s.end - s.p)); }
            else if (lf == blobs_field && lw == 2) blobs.push_back(layer.sub());
            else layer.skip(lw);
        }
        if (!layer.ok) { rc = fail(NCT_ERR_IO, "malformed layer message in '%s'", path); break; }
        int idx = -1;
        for (int i = 0; i < NCONV; ++i) if (name == kConvName[i]) idx = i;
        if (idx < 0 || blobs.empty()) continue;          // unknown source layer: ignored (net.cpp:770-773)
        if (idx >= kNeeded) { found[idx] = true; continue; }
        if (blobs.size() < 2) { rc = fail(NCT_ERR_IO, "layer %s has %zu blobs, expected weights + bias", name.c_str(), blobs.size()); break; }
        BlobView w, b;
        if (!parse_blob(blobs[0], w) || !parse_blob(blobs[1], b)) { rc = fail(NCT_ERR_IO, "malformed blob in layer %s", name.c_str()); break; }
        const int64_t nw = (int64_t)kCout[idx] * kCin[idx] * 9;
        const bool wshape = w.ndim == 4 && w.dims[0] == kCout[idx] && w.dims[1] == kCin[idx] && w.dims[2] == 3 && w.dims[3] == 3;
        if (!wshape || (int64_t)w.data.size() != nw)      // shape mismatch is fatal (net.cpp:780-791)
            { rc = fail(NCT_ERR_IO, "layer %s: weight shape mismatch (got %lldx%lldx%lldx%lld, %zu values; expected %dx%dx3x3)", name.c_str(),
                        (long long)w.dims[0], (long long)w.dims[1], (long long)w.dims[2], (long long)w.dims[3], w.data.size(), kCout[idx], kCin[idx]); break; }
        if (blob_count(b) != kCout[idx] || (int64_t)b.data.size() != kCout[idx])
            { rc = fail(NCT_ERR_IO, "layer %s: bias shape mismatch (%zu values, expected %d)", name.c_str(), b.data.size(), kCout[idx]); break; }
        m.w[idx] = std::move(w.data); m.b[idx] = std::move(b.data);
        found[idx] = true;
    }
    if (rc == NCT_OK && !net.ok) rc = fail(NCT_ERR_IO, "malformed NetParameter in '%s'", path);
    munmap(map, (size_t)st.st_size);
    if (rc != NCT_OK) return rc;
    for (int i = 0; i < kNeeded; ++i)
        if (!found[i]) return fail(NCT_ERR_IO, "caffemodel '%s' has no weights for layer %s", path, kConvName[i]);
    return NCT_OK;
}

// ---------------------------------------------------------------- deploy prototxt (protobuf text format) -> topology check
// Classifier::Classifier builds the Net from <model_dir>/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt (Classifier.cpp:16, main.cu:575-577). The topology
// is built into k_vgg.hip, so the file is only CHECKED: a model directory whose prototxt describes another network must not be accepted silently.
namespace {
struct TxtLayer { std::string name, type, pool; std::vector<std::string> bottom, top; long num_output = -1, kernel = -1, pad = 0, stride = 1;
                  long kernel_h = -1, kernel_w = -1, pad_h = -1, pad_w = -1, stride_h = -1, stride_w = -1; };     // the _h/_w spellings of caffe.proto (ConvolutionParameter 11-14, PoolingParameter 5-10)
struct TxtTok { const char* p; const char* end; bool ok = true;
    void ws() { while (p < end) { if (*p == '#') { while (p < end && *p != '\n') ++p; } else if (isspace((unsigned char)*p)) ++p; else break; } }
    bool ident(std::string& out) { ws(); const char* s = p; while (p < end && (isalnum((unsigned char)*p) || *p == '_' || *p == '.' || *p == '-' || *p == '+')) ++p; out.assign(s, p); return p > s; }
    bool lit(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    bool value(std::string& out) {      // "string" | identifier/number
        ws();
        if (p < end && (*p == '"' || *p == '\'')) { const char q = *p++; const char* s = p; while (p < end && *p != q) ++p; if (p >= end) return false; out.assign(s, p); ++p; return true; }
        return ident(out);
    }
};
// parses `field: value` / `field { ... }` pairs of one message body until '}' (or the end of the file at depth 0)
bool parse_msg(TxtTok& t, int depth, TxtLayer* L, std::vector<TxtLayer>* layers, const std::string& path_in_layer) {
    if (depth > 32) return false;              // model directories are user input: bounded recursion on deeply nested braces
    for (;;) {
        t.ws();
        if (t.p >= t.end) return depth == 0;
        if (*t.p == '}') { if (depth == 0) return false; ++t.p; return true; }
        std::string key;
        if (!t.ident(key)) return false;
        const bool colon = t.lit(':');
        t.ws();
        if (t.p < t.end && (*t.p == '{' || *t.p == '<')) {
            ++t.p;
            if (depth == 0 && (key == "layer" || key == "layers")) {
                TxtLayer nl;
                if (!parse_msg(t, 1, &nl, nullptr, "")) return false;
                layers->push_back(nl);
            } else if (!parse_msg(t, depth + 1, L, nullptr, path_in_layer.empty() ? key : path_in_layer + "." + key)) return false;
            continue;
        }
        if (!colon) return false;
        std::string v;
        if (!t.value(v)) return false;
        if (L && depth >= 1) {
            const std::string full = path_in_layer.empty() ? key : path_in_layer + "." + key;
            if (full == "name") L->name = v; else if (full == "type") L->type = v;
            else if (full == "bottom") L->bottom.push_back(v); else if (full == "top") L->top.push_back(v);
            else if (full == "convolution_param.num_output") L->num_output = atol(v.c_str());
            else if (full == "convolution_param.kernel_size" || full == "pooling_param.kernel_size") L->kernel = atol(v.c_str());
            else if (full == "convolution_param.pad") L->pad = atol(v.c_str());
            else if (full == "convolution_param.stride" || full == "pooling_param.stride") L->stride = atol(v.c_str());
            else if (full == "pooling_param.pool") L->pool = v;
            else if (full == "convolution_param.kernel_h" || full == "pooling_param.kernel_h") L->kernel_h = atol(v.c_str());
            else if (full == "convolution_param.kernel_w" || full == "pooling_param.kernel_w") L->kernel_w = atol(v.c_str());
            else if (full == "convolution_param.pad_h" || full == "pooling_param.pad_h") L->pad_h = atol(v.c_str());
            else if (full == "convolution_param.pad_w" || full == "pooling_param.pad_w") L->pad_w = atol(v.c_str());
            else if (full == "convolution_param.stride_h" || full == "pooling_param.stride_h") L->stride_h = atol(v.c_str());
            else if (full == "convolution_param.stride_w" || full == "pooling_param.stride_w") L->stride_w = atol(v.c_str());
        }
    }
}
std::string lower(std::string s) { for (auto& c : s) c = (char)tolower((unsigned char)c); return s; }
}  // namespace

static int check_prototxt(const char* path, std::string& err) {
    char buf[512];
    auto fail = [&](const char* fmt, auto... a) { snprintf(buf, sizeof buf, fmt, a...); err = buf; return (int)NCT_ERR_IO; };
    FILE* f = fopen(path, "rb");
    if (!f) return fail("cannot open prototxt '%s': %s", path, strerror(errno));
    std::string txt; char chunk[65536]; size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) { txt.append(chunk, n); if (txt.size() > (64u << 20)) break; }
    fclose(f);
    TxtTok t{txt.data(), txt.data() + txt.size()};
    std::vector<TxtLayer> layers;
    if (!parse_msg(t, 0, nullptr, &layers, "")) return fail("prototxt '%s': not protobuf text format (near byte %zu)", path, (size_t)(t.p - txt.data()));
    // walk the data path: every conv of the built-in topology up to conv5_1 must be there, in order, 3x3 / pad 1 / stride 1 with the built-in
    // channel count, followed by an in-place ReLU; a 2x2 / stride-2 MAX pool exactly after conv1_2, conv2_2, conv3_4, conv4_4
    int ci = 0; std::string cur = "data"; bool expect_relu = false, expect_pool = false;
    for (TxtLayer L : layers) {
        if (ci >= kNeeded && !expect_relu) break;
        const std::string ty = lower(L.type);
        // square kernels / pads / strides spelled per axis (kernel_h == kernel_w ...) are the same layer; unequal ones are not this network
        if (L.kernel_h >= 0 || L.kernel_w >= 0) { if (L.kernel_h != L.kernel_w) return fail("prototxt '%s': %s has a non-square kernel", path, L.name.c_str()); L.kernel = L.kernel_h; }
        if (L.pad_h >= 0 || L.pad_w >= 0) { if (L.pad_h != L.pad_w) return fail("prototxt '%s': %s has unequal pads", path, L.name.c_str()); L.pad = L.pad_h; }
        if (L.stride_h >= 0 || L.stride_w >= 0) { if (L.stride_h != L.stride_w) return fail("prototxt '%s': %s has unequal strides", path, L.name.c_str()); L.stride = L.stride_h; }
        if (ty == "input" || ty == "data" || ty == "memorydata" || ty == "dummydata" || ty == "5" || ty == "29") {
            // the input declared as a layer (the reference's own Caffe ships input_layer.cpp; the demo file uses the legacy `input:` fields): its top is the data path's start
            if (ci != 0 || L.top.empty()) return fail("prototxt '%s': input layer '%s' inside the network", path, L.name.c_str());
            cur = L.top[0];
            continue;
        }
        if (ty == "convolution" || ty == "4") {
            if (expect_relu) return fail("prototxt '%s': %s is not followed by a ReLU", path, kConvName[ci - 1]);
            if (expect_pool) return fail("prototxt '%s': no 2x2 max pool after %s", path, kConvName[ci - 1]);
            if (L.name != kConvName[ci]) return fail("prototxt '%s': conv layer %d is '%s', expected '%s'", path, ci + 1, L.name.c_str(), kConvName[ci]);
            if (L.num_output != kCout[ci] || L.kernel != 3 || L.pad != 1 || L.stride != 1)
                return fail("prototxt '%s': %s is num_output %ld kernel %ld pad %ld stride %ld, expected %d / 3 / 1 / 1", path, L.name.c_str(), L.num_output, L.kernel, L.pad, L.stride, kCout[ci]);
            if (L.bottom.size() != 1 || L.bottom[0] != cur || L.top.size() != 1) return fail("prototxt '%s': %s does not consume '%s'", path, L.name.c_str(), cur.c_str());
            cur = L.top[0]; expect_relu = true; expect_pool = kPoolAfter[ci]; ++ci;
        } else if (ty == "relu" || ty == "18") {
            if (!expect_relu || L.bottom.size() != 1 || L.bottom[0] != cur || L.top.size() != 1)
                return fail("prototxt '%s': unexpected ReLU '%s'", path, L.name.c_str());
            cur = L.top[0];                        // in place (top == bottom, the shipped file) or into a blob of its own: the same function
            expect_relu = false;
        } else if (ty == "pooling" || ty == "17") {
            if (!expect_pool || expect_relu) return fail("prototxt '%s': unexpected pooling layer '%s'", path, L.name.c_str());
            if ((lower(L.pool) != "max" && L.pool != "0" && !L.pool.empty()) || L.kernel != 2 || L.stride != 2 || L.bottom.size() != 1 || L.bottom[0] != cur || L.top.size() != 1)
                return fail("prototxt '%s': %s is not a 2x2 / stride-2 MAX pool of '%s'", path, L.name.c_str(), cur.c_str());
            cur = L.top[0]; expect_pool = false;
        } else return fail("prototxt '%s': layer '%s' of type '%s' in front of conv5_1 — not the VGG19 this library implements", path, L.name.c_str(), L.type.c_str());
    }
    if (ci < kNeeded || expect_relu) return fail("prototxt '%s': the network ends before relu5_1 (%d of %d conv layers)", path, ci, kNeeded);
    return NCT_OK;
}

extern "C" {

int nct_vgg19_load_raw(nct_ctx* ctx, const float* const* weights, const float* const* biases, int nlayers) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(weights && biases && nlayers >= kNeeded && nlayers <= NCONV, "vgg19_load_raw: need >= %d conv layers (conv1_1..conv5_1)", kNeeded);
    vgg_own(ctx)->loaded = false;
    for (int i = 0; i < kNeeded; ++i) {
        NCT_REQUIRE(weights[i] && biases[i], "vgg19_load_raw: layer %s missing", kConvName[i]);
        int rc = upload_layer(ctx, i, weights[i], biases[i]);
        if (rc) return rc;
    }
    vgg_of(ctx)->loaded = true;
    return NCT_OK;
}

int nct_model_parse_caffemodel(const char* path, nct_model** out) {
    if (!path || !out) { g_model_err = "nct_model_parse_caffemodel: null argument"; return NCT_ERR_INVALID; }
    nct_model* m = new nct_model();
    const int rc = parse_caffemodel(path, *m, g_model_err);
    if (rc != NCT_OK) { delete m; *out = nullptr; return rc; }
    *out = m;
    return NCT_OK;
}
void nct_model_free(nct_model* m) { delete m; }
int nct_model_layer(const nct_model* m, int layer, const float** weights, const float** bias, int* cout, int* cin) {
    if (!m || layer < 0 || layer >= kNeeded || m->w[layer].empty()) { g_model_err = "nct_model_layer: no such layer in the parsed model"; return NCT_ERR_INVALID; }
    if (weights) *weights = m->w[layer].data();
    if (bias) *bias = m->b[layer].data();
    if (cout) *cout = kCout[layer];
    if (cin) *cin = kCin[layer];
    return NCT_OK;
}
const char* nct_model_last_error(void) { return g_model_err.c_str(); }

int nct_vgg19_load_model(nct_ctx* ctx, const nct_model* m) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(m, "vgg19_load_model: null model");
    vgg_own(ctx)->loaded = false;
    for (int i = 0; i < kNeeded; ++i) { int rc = upload_layer(ctx, i, m->w[i].data(), m->b[i].data()); if (rc) return rc; }
    vgg_of(ctx)->loaded = true;
    return NCT_OK;
}

int nct_vgg19_load_caffemodel(nct_ctx* ctx, const char* path) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(path, "vgg19_load_caffemodel: null path");
    nct_model m; std::string err;
    const int rc = parse_caffemodel(path, m, err);
    if (rc != NCT_OK) return ctx->fail(rc, "%s", err.c_str());
    return nct_vgg19_load_model(ctx, &m);
}

int nct_vgg19_share_weights(nct_ctx* ctx, nct_ctx* from) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_REQUIRE(from && from != ctx, "vgg19_share_weights: need another context");
    NCT_REQUIRE(from->device == ctx->device, "vgg19_share_weights: contexts live on different GPUs (%d vs %d): load the weights once per GPU", ctx->device, from->device);
    vgg_holder* src = (vgg_holder*)from->vgg;
    if (!src || !src->w || !src->w->loaded) return ctx->fail(NCT_ERR_STATE, "vgg19_share_weights: the source context has no weights loaded");
    if (!ctx->vgg) ctx->vgg = new vgg_holder();
    ((vgg_holder*)ctx->vgg)->w = src->w;           // read-only after loading: no synchronisation needed between the sharers
    return NCT_OK;
}

int nct_vgg19_weights_info(nct_ctx* ctx, uint64_t* id, size_t* bytes, int* sharers) {
    if (!ctx) return NCT_ERR_INVALID;
    vgg_holder* h = (vgg_holder*)ctx->vgg;
    if (!h || !h->w || !h->w->loaded) return ctx->fail(NCT_ERR_STATE, "vgg19_weights_info: no weights loaded");
    if (id) *id = (uint64_t)(uintptr_t)h->w->wp[0];
    if (bytes) *bytes = h->w->bytes;
    if (sharers) *sharers = (int)h->w.use_count();
    return NCT_OK;
}

int nct_vgg19_check_prototxt(nct_ctx* ctx, const char* path) {
    std::string err;
    if (!path) { if (ctx) ctx->fail(NCT_ERR_INVALID, "vgg19_check_prototxt: null path"); else g_model_err = "vgg19_check_prototxt: null path"; return NCT_ERR_INVALID; }
    const int rc = check_prototxt(path, err);
    if (rc != NCT_OK) { if (ctx) ctx->fail(rc, "%s", err.c_str()); else g_model_err = err; }
    return rc;
}

}  // extern "C"

void nct_vgg_free(nct_ctx* ctx) {
    if (!ctx || !ctx->vgg) return;
    delete (vgg_holder*)ctx->vgg;          // the device copy goes with its last holder
    ctx->vgg = nullptr;
}

// ---------------------------------------------------------------- forward (device): taps in CHW
// d_bgr: device u8 BGR HWC. d_taps[t] (nullable, caller-owned device buffers of C*h*w floats) receive tap t+1.
// dims[t] = {C,h,w} is filled for every tap <= deepest_tap.
// d_taps: Caffe's planar CHW maps of the taps (entries nullable); d_taps_hwc (nullable array, entries nullable): the same taps channel-last, written by the tap layer's own
// epilogue (no transpose pass). A tap asked for only channel-last is still written planar into the ping-pong buffer when a further layer reads it; the deepest tap is then not.
// stop_into (nullable): the forward stops in FRONT of its last conv layer and leaves that layer's input (the pooled map of the layer before) there; *stop_h / *stop_w = its size
static int vgg19_forward_impl(nct_ctx* ctx, hipStream_t s, const uint8_t* d_bgr, int H, int W, int stride, int deepest_tap, float* const* d_taps, int* dims, float* const* d_taps_hwc,
                              float* stop_into, int* stop_h, int* stop_w) {
    vgg_weights* v = ctx->vgg ? ((vgg_holder*)ctx->vgg)->w.get() : nullptr;
    if (!v || !v->loaded) return ctx->fail(NCT_ERR_STATE, "vgg19: weights not loaded (nct_vgg19_load_caffemodel / _load_raw)");
    NCT_REQUIRE(deepest_tap >= 1 && deepest_tap <= 5, "vgg19: deepest_tap must be 1..5");
    NCT_REQUIRE(H >= 2 && W >= 2 && H < 4096 && W < 4096, "vgg19: image size %dx%d out of range", W, H);
    const size_t big = (size_t)64 * H * W;
    DevBuf<float> pp0(ctx, big), pp1(ctx, big);
    if (!pp0.ok() || !pp1.ok()) return NCT_ERR_HIP;
    float* pp[2] = {pp0, pp1};
    int h = H, w = W;
    int rc = nctk_vgg_preprocess(ctx, s, d_bgr, stride, pp[0], H, W);
    if (rc) return rc;
    const float* cur = pp[0];
    const int last = kTapConv[deepest_tap - 1];
    for (int i = 0; i <= last; ++i) {
        if (stop_into && i == last) { *stop_h = h; *stop_w = w; break; }
        int tap = -1;
        for (int t = 0; t < 5; ++t) if (kTapConv[t] == i) tap = t;
        float* dst;
        float* dst_hwc = (tap >= 0 && d_taps_hwc) ? d_taps_hwc[tap] : nullptr;
        if (tap >= 0 && d_taps && d_taps[tap]) dst = d_taps[tap];
        else if (i == last && dst_hwc) dst = nullptr;                        // the forward ends here and nobody asked for the planar map
        else dst = (cur == pp[0]) ? pp[1] : pp[0];
        const bool into_stop = stop_into && i == last - 1;                   // this layer's (pooled) output is what the caller keeps
        // a pooled layer is never a tap (taps are conv*_1): where the tile shape fits, the pool rides in the conv epilogue and the unpooled map is never written
        const bool pooled = kPoolAfter[i] && i < last;
        const bool fuse = pooled && tap < 0 && (ctx->conv_pool_fuse == 1 || (ctx->conv_pool_fuse < 0 && nctk_conv3x3_pool_fits(h, w)));
        if (into_stop && (fuse || !pooled)) dst = stop_into;
        rc = nctk_conv3x3(ctx, s, cur, v->wp[i], v->bias[i], dst, (kCin[i] + 1) & ~1, kCout[i], h, w, 1, fuse ? 1 : 0, dst_hwc);
        if (rc) return rc;
        cur = dst;
        if (tap >= 0 && dims) { dims[tap * 3 + 0] = kCout[i]; dims[tap * 3 + 1] = h; dims[tap * 3 + 2] = w; }
        if (fuse) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
        else if (pooled) {
            float* pd = into_stop ? stop_into : ((cur == pp[0]) ? pp[1] : pp[0]);
            rc = nctk_maxpool2x2(ctx, s, cur, pd, kCout[i], h, w);
            if (rc) return rc;
            cur = pd; h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
        }
    }
    return 0;
}
int nctk_vgg19_forward(nct_ctx* ctx, hipStream_t s, const uint8_t* d_bgr, int H, int W, int stride, int deepest_tap, float* const* d_taps, int* dims, float* const* d_taps_hwc) {
    return vgg19_forward_impl(ctx, s, d_bgr, H, W, stride, deepest_tap, d_taps, dims, d_taps_hwc, nullptr, nullptr, nullptr);
}
int nctk_vgg19_forward_pair(nct_ctx* ctx, hipStream_t s, const uint8_t* d_bgr1, int H1, int W1, int stride1, float* const* taps_hwc1,
                            const uint8_t* d_bgr2, int H2, int W2, int stride2, float* const* taps_hwc2) {
    NCT_REQUIRE(taps_hwc1 && taps_hwc2 && taps_hwc1[4] && taps_hwc2[4], "vgg19 pair: both conv5_1 maps are needed");
    auto pooled4 = [](int n) { for (int k = 0; k < 4; ++k) n = (n - 1) / 2 + 1; return n; };
    DevBuf<float> p1(ctx, (size_t)512 * pooled4(H1) * pooled4(W1)), p2(ctx, (size_t)512 * pooled4(H2) * pooled4(W2));
    if (!p1.ok() || !p2.ok()) return NCT_ERR_HIP;
    int h1 = 0, w1 = 0, h2 = 0, w2 = 0;
    int rc = vgg19_forward_impl(ctx, s, d_bgr1, H1, W1, stride1, 5, nullptr, nullptr, taps_hwc1, p1, &h1, &w1); if (rc) return rc;
    rc = vgg19_forward_impl(ctx, s, d_bgr2, H2, W2, stride2, 5, nullptr, nullptr, taps_hwc2, p2, &h2, &w2); if (rc) return rc;
    vgg_weights* v = ((vgg_holder*)ctx->vgg)->w.get();
    const int i = kTapConv[4];
    return nctk_conv3x3_pair(ctx, s, p1, h1, w1, p2, h2, w2, v->wp[i], v->bias[i], nullptr, nullptr, (kCin[i] + 1) & ~1, kCout[i], 1, taps_hwc1[4], taps_hwc2[4]);
}

extern "C" {

int nct_vgg19_features(nct_ctx* ctx, const uint8_t* bgr, int h, int w, int stride, int deepest_tap, float* const* taps_chw, int* dims) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(bgr && h > 0 && w > 0 && stride >= 3 * w, "vgg19_features: bad image arguments");
    NCT_REQUIRE(deepest_tap >= 1 && deepest_tap <= 5, "vgg19_features: deepest_tap must be 1..5");
    DevBuf<uint8_t> img(ctx, (size_t)stride * h);
    if (!img.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemcpyAsync(img, bgr, (size_t)stride * h, hipMemcpyHostToDevice, ctx->stream));
    // tap buffers
    int hh = h, ww = w; size_t sizes[5]; int td[5][3];
    for (int t = 0; t < 5; ++t) { const int C = kCout[kTapConv[t]]; td[t][0] = C; td[t][1] = hh; td[t][2] = ww; sizes[t] = (size_t)C * hh * ww; hh = (hh - 1) / 2 + 1; ww = (ww - 1) / 2 + 1; }
    float* d_taps[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<DevBuf<float>*> bufs;
    int rc = 0;
    for (int t = 0; t < deepest_tap && rc == 0; ++t)
        if (taps_chw && taps_chw[t]) { auto* b = new DevBuf<float>(ctx, sizes[t]); bufs.push_back(b); if (!b->ok()) rc = NCT_ERR_HIP; d_taps[t] = *b; }
    int ldims[15] = {0};
    if (rc == 0) rc = nctk_vgg19_forward(ctx, ctx->stream, img, h, w, stride, deepest_tap, d_taps, ldims);
    if (rc == 0)
        for (int t = 0; t < deepest_tap; ++t)
            if (d_taps[t]) { hipError_t e = hipMemcpyAsync(taps_chw[t], d_taps[t], sizeof(float) * sizes[t], hipMemcpyDeviceToHost, ctx->stream);
                             if (e != hipSuccess) { rc = ctx->fail(NCT_ERR_HIP, "D2H of tap %d failed: %s", t + 1, hipGetErrorString(e)); break; } }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == 0 && e != hipSuccess) rc = ctx->fail(NCT_ERR_HIP, "vgg19_features: %s", hipGetErrorString(e));
    for (auto* b : bufs) delete b;
    if (dims) for (int t = 0; t < 5; ++t) for (int k = 0; k < 3; ++k) dims[t * 3 + k] = t < deepest_tap ? td[t][k] : 0;
    return rc;
}

// single-layer entry points (unit parity tests against Caffe's known answers / the oracle)
int nct_conv3x3_relu(nct_ctx* ctx, const float* in_chw, int Cin, int H, int W, const float* weights /*[Cout][Cin][3][3]*/, const float* bias,
                     int Cout, float* out_chw, int relu) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(in_chw && weights && bias && out_chw, "conv3x3_relu: null pointer");
    NCT_REQUIRE(Cin >= 1 && (Cout & 63) == 0 && H >= 2 && W >= 2, "conv3x3_relu: need Cout %% 64 == 0, H,W >= 2");
    const int cin_pad = (Cin + 1) & ~1;
    const size_t hw = (size_t)H * W;
    DevBuf<float> din(ctx, cin_pad * hw), dw(ctx, (size_t)Cout * Cin * 9), dwp(ctx, (size_t)cin_pad * 9 * Cout), db(ctx, Cout), dout(ctx, Cout * hw);
    if (!din.ok() || !dw.ok() || !dwp.ok() || !db.ok() || !dout.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemsetAsync(din, 0, sizeof(float) * cin_pad * hw, ctx->stream));
    NCT_HIP(hipMemcpyAsync(din, in_chw, sizeof(float) * Cin * hw, hipMemcpyHostToDevice, ctx->stream));
    NCT_HIP(hipMemcpyAsync(dw, weights, sizeof(float) * (size_t)Cout * Cin * 9, hipMemcpyHostToDevice, ctx->stream));
    NCT_HIP(hipMemcpyAsync(db, bias, sizeof(float) * Cout, hipMemcpyHostToDevice, ctx->stream));
    int rc = nctk_pack_weights(ctx, ctx->stream, dw, dwp, Cout, Cin, cin_pad);
    if (rc) return rc;
    rc = nctk_conv3x3(ctx, ctx->stream, din, dwp, db, dout, cin_pad, Cout, H, W, relu, 0);
    if (rc) return rc;
    NCT_HIP(hipMemcpyAsync(out_chw, dout, sizeof(float) * Cout * hw, hipMemcpyDeviceToHost, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    return NCT_OK;
}

int nct_maxpool2x2(nct_ctx* ctx, const float* in_chw, int C, int H, int W, float* out_chw) {
    if (!ctx) return NCT_ERR_INVALID;
    NCT_HIP(hipSetDevice(ctx->device));
    NCT_REQUIRE(in_chw && out_chw && C >= 1 && H >= 2 && W >= 2, "maxpool2x2: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    DevBuf<float> din(ctx, (size_t)C * H * W), dout(ctx, (size_t)C * Ho * Wo);
    if (!din.ok() || !dout.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemcpyAsync(din, in_chw, sizeof(float) * C * H * W, hipMemcpyHostToDevice, ctx->stream));
    int rc = nctk_maxpool2x2(ctx, ctx->stream, din, dout, C, H, W);
    if (rc) return rc;
    NCT_HIP(hipMemcpyAsync(out_chw, dout, sizeof(float) * C * Ho * Wo, hipMemcpyDeviceToHost, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    return NCT_OK;
}

}  // extern "C"
