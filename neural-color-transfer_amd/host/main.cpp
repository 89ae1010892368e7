// neural_color_transfer — the reference's console driver re-created on top of libnct (C ABI, include/nct.h).
// Mirrors: get_input / main (main.cu:29-44, 546-590), transfer_single (main.cu:456-543), Utility::CmdLine
// (CmdLine.h:58-69,132-147; CmdLine.cpp:21-56,93-109). Same flags, same pairs.txt, same output names, same log lines.
// Differences: portable path handling ('/' and '\\'), in-repo PNG + JPEG (baseline, progressive) decoders instead of cv::imread
// (output is PNG like the reference), a missing pairs.txt is an error instead of a NULL dereference (main.cu:463-471), plus the
// extensions `-gpus N` (pairs sharded over N GPUs, one context per worker thread), `-inflight K`, `-io T` (shared decode/encode pool: the GPU workers never
// touch zlib), `-pin` (threads on the GPU's NUMA node), weights parsed once per process and held once per GPU, `-seed`, `-levels L` (BASELINE
// config 1: "L=5 only" = -levels 1), `-resume 1` (skip pairs whose output exists; <out>/status.jsonl gets one JSON line per pair)
// and `-feat16 1` (reduced-precision PatchMatch features; not bit-identical).
#include <sys/stat.h>
#include <sys/wait.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include "nct.h"
#include "png_io.h"
#include "jpeg_io.h"
#include "affinity.h"
#include "rccl_sync.h"

namespace {
constexpr int MAX_SIZE = 1000;                 // Config.h:5

struct Param { std::string flag, comment; enum { STR, INT, DBL } kind; void* dst; };
// Utility::CmdLine (CmdLine.h:58-69,132-147; CmdLine.cpp:21-56,93-109) re-created; pinned against the reference's own parser compiled unmodified
// (oracle/ref_cmdline.cpp -> tests/golden/cmdline_ref.json -> tests/test_cli.py). The rules, all the reference's:
//  * a token is an "argument" when it starts with '-' or '/' (Parameter::IsArg) — "-5" and "/x" included; any other token is a positional file (collected, unused);
//  * "-h" "-?" "-help" (and the '/' forms) print the help and end the run; an argument that names no parameter prints "Unrecognized parameter: …", the help, and ends it;
//  * the token after a parameter is its value unless it is missing, empty or itself an argument (TParm::Parse) — then the parameter keeps its value and the
//    token is looked at again; numbers are read with operator>> (so "12abc" is 12, "abc" is 0, "3.7" for -g is 3).
// Two portable extensions, each a case the reference ends with "Unrecognized parameter" (recorded as such in the fixture): a value of the form -<digit|.>… is taken
// as a negative number by numeric parameters, and a value that starts with '/' but names no parameter is taken as a unix path by the string parameters (the reference
// is a Windows tool: "-m /data/models" cannot be passed to it at all). A string value keeps its blanks (operator>> into a std::string stops at the first one).
struct CmdLine {
    std::vector<Param> params;
    int files = 0;
    void add(const char* flag, std::string& v, const char* c) { params.push_back({flag, c, Param::STR, &v}); }
    void add(const char* flag, int& v, const char* c) { params.push_back({flag, c, Param::INT, &v}); }
    void add(const char* flag, double& v, const char* c) { params.push_back({flag, c, Param::DBL, &v}); }
    static bool is_arg(const char* a) { return a && (a[0] == '-' || a[0] == '/'); }                       // Parameter::IsArg, CmdLine.h:68-70
    static bool is_help(const std::string& a) { return a == "h" || a == "?" || a == "help"; }              // CmdLine.cpp:93-100
    bool names_a_parameter(const char* a) const {
        const std::string n(a + 1);
        if (is_help(n)) return true;
        for (const auto& p : params) if (n == p.flag) return true;
        return false;
    }
    bool is_value(const Param& p, const char* v) const {                                                    // TParm::Parse's test, CmdLine.h:133-136, + the two extensions
        if (!v || !*v) return false;
        if (!is_arg(v)) return true;
        if (p.kind != Param::STR) return v[0] == '-' && ((v[1] >= '0' && v[1] <= '9') || v[1] == '.');
        return v[0] == '/' && !names_a_parameter(v);
    }
    void help(const char* prog) const {
        std::cout << "Running: " << prog << std::endl;
        for (const auto& p : params) {                                          // TParm::Print, CmdLine.h:140-142
            std::cout << "-" << p.flag << ": " << "(default=";
            if (p.kind == Param::STR) std::cout << *(const std::string*)p.dst;
            else if (p.kind == Param::INT) std::cout << *(const int*)p.dst;
            else std::cout << *(const double*)p.dst;
            std::cout << ") " << p.comment << std::endl;
        }
    }
    bool parse(int argc, char** argv, int first = 1) {
        int i = first;
        while (i < argc) {
            if (!is_arg(argv[i])) { ++files; ++i; continue; }              // positional "files" are collected and never used (CmdLine.cpp:26-29)
            const std::string a(argv[i] + 1);
            if (is_help(a)) { help(argv[0]); return false; }
            bool done = false;
            for (const auto& p : params)
                if (a == p.flag) {
                    if (i + 1 < argc && is_value(p, argv[i + 1])) {
                        if (p.kind == Param::STR) *(std::string*)p.dst = argv[i + 1];
                        else { std::istringstream is(argv[i + 1]); if (p.kind == Param::INT) is >> *(int*)p.dst; else is >> *(double*)p.dst; }
                        ++i;
                    }
                    ++i; done = true; break;
                }
            if (!done) { std::cout << "Unrecognized parameter: " << argv[i] << std::endl << std::endl; help(argv[0]); return false; }
        }
        return true;
    }
};

std::string stem(const std::string& path) {          // main.cu:524-531 (find_last_of on both separators, strip the extension)
    const size_t pos = path.find_last_of("\\/") + 1;
    const size_t dot = path.find_last_of('.');
    return path.substr(pos, dot == std::string::npos || dot < pos ? std::string::npos : dot - pos);
}

struct Pair { std::string cnt, stl; float bds; };
std::mutex g_print;

struct Config { std::string input_dir, output_dir, model_dir; nct_params prm; bool resume = false, vis = false; int rank = 0, world = 1; };

// ---- ENABLE_VIS debug outputs (Config.h:8) behind the runtime flag -vis 1: per pyramid level the flow maps of both NNFs (reconstruct_flow,
// GeneralizedPatchMatch.cu:337-353), the level images tCnt / tStl (main.cu:343-347), the matching-error heat map (getHeat,
// ColorTransfer.cpp:1127-1178 on the min-max normalised error, :1318-1338) — under the reference's file names <pre>_aFlow_<l>.png … — plus
// the BDS guidance image and the intermediate result of the level (guide_<l>, result_<l>: not dumped by the reference, but what its
// refine_* images are for); the linear colour coefficients after each stage of the level as aVis / bVis images and the source recoloured by
// them (aVis_init, bVis_init, refine_init: local statistics sampled with x / samples, ColorTransfer.cpp:1268-1300; aVis_nonlocal, bVis_nonlocal,
// refine_nonlocal: after the nonlocal solve and the bilinear upsampling, :1384-1413; aVis, bVis: after the WLS solve, :1450-1463); the
// clusters as <pre>_cluster_small.png and per level as knn_<l>.png (visualizeClusterRandom / findKnns, :222-246, :336-351 — with a hashed
// palette: the reference's 260-entry RandomColorList is a data table of its Config.h); patchVis_<l>: per level pixel a 3-wide, 6-high cell
// with the (border-clipped) 3x3 patch of the guidance image above that of the level image, the windows of the local statistics (:1190-1221).
// <pre> = the output file's stem.
void heat(double v, uint8_t* bgr) {
    v = !(v >= 0) ? 0 : (v > 1 ? 1 : v);          // NaN -> 0 as well
    double dr, dg, db;
    if (v < 0.1242) { db = 0.504 + ((1. - 0.504) / 0.1242) * v; dg = dr = 0.; }
    else if (v < 0.3747) { db = 1.; dr = 0.; dg = (v - 0.1242) * (1. / (0.3747 - 0.1242)); }
    else if (v < 0.6253) { db = (0.6253 - v) * (1. / (0.6253 - 0.3747)); dg = 1.; dr = (v - 0.3747) * (1. / (0.6253 - 0.3747)); }
    else if (v < 0.8758) { db = 0.; dr = 1.; dg = (0.8758 - v) * (1. / (0.8758 - 0.6253)); }
    else { db = 0.; dg = 0.; dr = 1. - (v - 0.8758) * ((1. - 0.504) / (1. - 0.8758)); }
    auto q = [](double d) { const int i = (int)(255 * d); return (uint8_t)(i > 255 ? 255 : i); };
    bgr[0] = q(db); bgr[1] = q(dg); bgr[2] = q(dr);
}
bool run_with_vis(nct_ctx* ctx, const ImageBGR& cnt, const ImageBGR& stl, const nct_params& prm, const std::string& pre, uint8_t* out, nct_pair_timing* tm, std::string& err) {
    int ah[5], aw[5], bh[5], bw[5];
    { int h = cnt.h, w = cnt.w, h2 = stl.h, w2 = stl.w;
      for (int t = 0; t < 5; ++t) { ah[4 - t] = h; aw[4 - t] = w; bh[4 - t] = h2; bw[4 - t] = w2; h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; h2 = (h2 - 1) / 2 + 1; w2 = (w2 - 1) / 2 + 1; } }
    std::vector<std::vector<uint32_t>> ann(5), bnn(5);
    std::vector<std::vector<uint8_t>> guide(5), result(5), simg(5), rimg(5);
    std::vector<std::vector<float>> errm(5);
    std::vector<std::vector<double>> ab_local(5), ab_up(5), ab_wls(5);
    std::vector<int> labels((size_t)ah[0] * aw[0]);
    nct_color_stages cs[5]; memset(cs, 0, sizeof cs);
    const size_t N = (size_t)cnt.h * cnt.w;
    nct_pair_levels lv; memset(&lv, 0, sizeof lv);
    lv.labels = labels.data();
    for (int l = 0; l < prm.levels; ++l) {
        ab_local[l].resize((size_t)6 * ah[l] * aw[l]); ab_up[l].resize(6 * N); ab_wls[l].resize(6 * N);
        cs[l].ab_local = ab_local[l].data(); cs[l].ab_up = ab_up[l].data(); cs[l].ab_wls = ab_wls[l].data();
        lv.color[l] = &cs[l];
        ann[l].resize((size_t)ah[l] * aw[l]); bnn[l].resize((size_t)bh[l] * bw[l]); guide[l].resize((size_t)ah[l] * aw[l] * 3);
        errm[l].resize((size_t)ah[l] * aw[l]); result[l].resize((size_t)cnt.h * cnt.w * 3);
        lv.ann[l] = ann[l].data(); lv.bnn[l] = bnn[l].data(); lv.guide[l] = guide[l].data(); lv.err[l] = errm[l].data(); lv.result[l] = result[l].data();
    }
    if (nct_pair_upload(ctx, cnt.px.data(), cnt.h, cnt.w, stl.px.data(), stl.h, stl.w) != NCT_OK || nct_pair_run_levels(ctx, &prm, tm, &lv) != NCT_OK ||
        nct_pair_download(ctx, out) != NCT_OK) { err = nct_last_error(ctx); return false; }
    // level images: the progressive bilinear pyramid of main.cu:104-108
    simg[4] = cnt.px; rimg[4] = stl.px;
    for (int l = 3; l >= 0; --l) {
        simg[l].resize((size_t)ah[l] * aw[l] * 3); rimg[l].resize((size_t)bh[l] * bw[l] * 3);
        if (nct_resize_u8c3(ctx, simg[l + 1].data(), ah[l + 1], aw[l + 1], simg[l].data(), ah[l], aw[l]) != NCT_OK ||
            nct_resize_u8c3(ctx, rimg[l + 1].data(), bh[l + 1], bw[l + 1], rimg[l].data(), bh[l], bw[l]) != NCT_OK) { err = nct_last_error(ctx); return false; }
    }
    auto save = [&](const char* what, int l, const uint8_t* px, int h, int w) {
        char name[1200]; snprintf(name, sizeof name, "%s_%s_%d.png", pre.c_str(), what, l);
        std::string e; return pngio::write(name, px, h, w, e);
    };
    // coefficient images: a -> int(a * 50), b -> int(b * 255 + 127), clamped to a byte (the clamp in double first: the cast of an
    // out-of-range double is undefined); recoloured source: clamp(lab / 255 * a + b, 0, 1) -> 8 bit (convertTo, round half to even) -> BGR
    std::vector<uint8_t> lab(N * 3);
    if (nct_bgr2lab_u8(ctx, cnt.px.data(), N, lab.data()) != NCT_OK) { err = nct_last_error(ctx); return false; }
    auto coef_images = [&](const char* tag, int l, const double* ab, int h, int w, int samples) {
        const double* a = ab; const double* b = ab + (size_t)3 * h * w;
        std::vector<uint8_t> av(N * 3), bv(N * 3), rl(N * 3), rb(N * 3);
        for (int y = 0; y < cnt.h; ++y)
            for (int x = 0; x < cnt.w; ++x) {
                const size_t i = (size_t)y * cnt.w + x, j = (size_t)(y / samples) * w + x / samples;
                for (int c = 0; c < 3; ++c) {
                    const double ac = a[3 * j + c], bc = b[3 * j + c];
                    auto byte = [](double v) { return (uint8_t)(int)(v != v ? 0. : (v < 0. ? 0. : (v > 255. ? 255. : v))); };
                    av[3 * i + c] = byte(ac * 50); bv[3 * i + c] = byte(bc * 255 + 127);
                    double v = lab[3 * i + c] / 255.0 * ac + bc;
                    v = v > 0.0 ? v : 0.0; v = v < 1.0 ? v : 1.0;
                    rl[3 * i + c] = (uint8_t)nearbyint(v * 255.0);
                }
            }
        char na[40], nb[40], nr[40];
        snprintf(na, sizeof na, "aVis%s", tag); snprintf(nb, sizeof nb, "bVis%s", tag); snprintf(nr, sizeof nr, "refine%s", tag);
        if (!save(na, l, av.data(), cnt.h, cnt.w) || !save(nb, l, bv.data(), cnt.h, cnt.w)) return false;
        if (!tag[0]) return true;                                        // the recoloured source after the WLS solve is result_<l>
        return nct_lab2bgr_u8(ctx, rl.data(), N, rb.data()) == NCT_OK && save(nr, l, rb.data(), cnt.h, cnt.w);
    };
    auto palette = [](int label, uint8_t* bgr) {
        uint32_t hsh = (uint32_t)(label + 1) * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        bgr[0] = (uint8_t)(64 + (hsh & 0xBF)); bgr[1] = (uint8_t)(64 + ((hsh >> 8) & 0xBF)); bgr[2] = (uint8_t)(64 + ((hsh >> 16) & 0xBF));
    };
    auto cluster_image = [&](int h, int w, int samples) {
        std::vector<uint8_t> im((size_t)h * w * 3);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int ly = std::min(y / samples, ah[0] - 1), lx = std::min(x / samples, aw[0] - 1);
                palette(labels[(size_t)ly * aw[0] + lx], &im[((size_t)y * w + x) * 3]);
            }
        return im;
    };
    { const auto im = cluster_image(ah[0], aw[0], 1);
      std::string e; if (!pngio::write((pre + "_cluster_small.png").c_str(), im.data(), ah[0], aw[0], e)) { err = "cannot write the -vis images"; return false; } }
    auto patch_image = [&](const uint8_t* stl_px, const uint8_t* cnt_px, int h, int w) {
        const int ps = 3;
        std::vector<uint8_t> im((size_t)h * ps * 2 * w * ps * 3, 0);
        const size_t pitch = (size_t)w * ps * 3;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int sx0 = std::max(x - 1, 0), sy0 = std::max(y - 1, 0), ex = std::min(x + 2, w), ey = std::min(y + 2, h);
                for (int sy = sy0; sy < ey; ++sy)
                    for (int sx = sx0; sx < ex; ++sx) {
                        memcpy(&im[(size_t)(y * ps * 2 + sy - sy0) * pitch + (size_t)(x * ps + sx - sx0) * 3], &stl_px[((size_t)sy * w + sx) * 3], 3);
                        memcpy(&im[(size_t)(y * ps * 2 + ps + sy - sy0) * pitch + (size_t)(x * ps + sx - sx0) * 3], &cnt_px[((size_t)sy * w + sx) * 3], 3);
                    }
            }
        return im;
    };
    for (int l = 0; l < prm.levels; ++l) {
        if (!save("patchVis", l, patch_image(guide[l].data(), simg[l].data(), ah[l], aw[l]).data(), ah[l] * 6, aw[l] * 3)) { err = "cannot write the -vis images"; return false; }
        if (!coef_images("_init", l, ab_local[l].data(), ah[l], aw[l], 1 << (4 - l)) || !coef_images("_nonlocal", l, ab_up[l].data(), cnt.h, cnt.w, 1) ||
            !coef_images("", l, ab_wls[l].data(), cnt.h, cnt.w, 1) || !save("knn", l, cluster_image(ah[l], aw[l], 1 << l).data(), ah[l], aw[l])) {
            err = "cannot write the -vis images"; return false; }
        auto flow = [&](const std::vector<uint32_t>& nn, int h, int w, int oh, int ow) {
            std::vector<uint8_t> f((size_t)h * w * 3);
            for (size_t i = 0; i < (size_t)h * w; ++i) {
                const int xb = (int)(nn[i] & 0xFFFu), yb = (int)((nn[i] >> 12) & 0xFFFu);
                f[3 * i] = (uint8_t)(255 * ((float)xb / ow)); f[3 * i + 1] = 0; f[3 * i + 2] = (uint8_t)(255 * ((float)yb / oh));
            }
            return f;
        };
        const auto fa = flow(ann[l], ah[l], aw[l], bh[l], bw[l]), fb = flow(bnn[l], bh[l], bw[l], ah[l], aw[l]);
        float mn = errm[l][0], mx = errm[l][0];
        for (float e : errm[l]) { mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
        std::vector<uint8_t> hm((size_t)ah[l] * aw[l] * 3);
        // a constant error map normalises to 0 (cv::normalize's min-max of a flat image), not 0/0
        for (size_t i = 0; i < errm[l].size(); ++i) heat(mx > mn ? ((double)errm[l][i] - mn) / ((double)mx - mn) : 0.0, &hm[3 * i]);
        if (!save("aFlow", l, fa.data(), ah[l], aw[l]) || !save("bFlow", l, fb.data(), bh[l], bw[l]) || !save("tCnt", l, simg[l].data(), ah[l], aw[l]) ||
            !save("tStl", l, rimg[l].data(), bh[l], bw[l]) || !save("errMap", l, hm.data(), ah[l], aw[l]) || !save("guide", l, guide[l].data(), ah[l], aw[l]) ||
            !save("result", l, result[l].data(), cnt.h, cnt.w)) { err = "cannot write the -vis images"; return false; }
    }
    return true;
}
std::mutex g_status;

std::string json_escape(const std::string& s) {
    std::string o;
    for (char c : s) { if (c == '"' || c == '\\') { o += '\\'; o += c; } else if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o += c; }
    return o;
}
// one JSON line per pair in <output_dir>/status.jsonl (batch bookkeeping for -resume; absent in the reference)
void write_status(const Config& cfg, size_t index, const char* status, const std::string& cnt, const std::string& stl, double bds, const std::string& out, double sec,
                  const std::string& msg) {
    std::lock_guard<std::mutex> g(g_status);
    // one process per GPU (-world N): every rank appends to its own file, status.<rank>.jsonl — appends of different processes to one file could interleave
    FILE* f = fopen((cfg.output_dir + (cfg.world > 1 ? "/status." + std::to_string(cfg.rank) + ".jsonl" : std::string("/status.jsonl"))).c_str(), "a");
    if (!f) return;
    fprintf(f, "{\"pair\": %zu, \"content\": \"%s\", \"style\": \"%s\", \"bds\": %.6g, \"status\": \"%s\", \"output\": \"%s\", \"seconds\": %.4f, \"message\": \"%s\"}\n",
            index, json_escape(cnt).c_str(), json_escape(stl).c_str(), bds, status, json_escape(out).c_str(), sec, json_escape(msg).c_str());
    fclose(f);
}

// shrink so that the longer side is <= MAX_SIZE, int truncation as in main.cu:500-522
bool shrink(nct_ctx* ctx, ImageBGR& img) {
    if (img.w <= MAX_SIZE && img.h <= MAX_SIZE) return true;
    int cw = MAX_SIZE, ch = (int)(cw / (float)img.w * img.h);
    if (img.w < img.h) { ch = MAX_SIZE; cw = (int)(ch / (float)img.h * img.w); }
    ImageBGR out; out.h = ch; out.w = cw; out.px.resize((size_t)ch * cw * 3);
    if (nct_resize_u8c3(ctx, img.px.data(), img.h, img.w, out.px.data(), ch, cw) != NCT_OK) return false;
    img = std::move(out);
    return true;
}

// ---- one pair = three stages, so that the GPU workers never wait on zlib (the CLI lost up to 30 % to PNG work):
//   load  (I/O pool)   : resume check, decode both images (PNG / JPEG)
//   run   (GPU worker) : shrink to MAX_SIZE on the GPU, nct_process_pair, the reference's log lines
//   store (I/O pool)   : PNG-encode the result (zlib level 3), status line
// With `-io 0` a GPU worker runs all three itself (the round-2 behaviour).
struct Job {
    size_t index = 0; Pair p; std::string name, log, err;
    ImageBGR cnt, stl; std::vector<uint8_t> out;
    std::chrono::steady_clock::time_point t0;
    enum { LOADED, SKIPPED, FAILED, DONE } state = LOADED;
    template <typename... A> void say(const char* fmt, A... a) { char line[1200]; snprintf(line, sizeof line, fmt, a...); log += line; }
    double secs() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

void finish(const Config& cfg, Job& j) {           // status line + the pair's log block, printed in one piece
    const char* st = j.state == Job::DONE ? "done" : (j.state == Job::SKIPPED ? "skipped" : "error");
    write_status(cfg, j.index, st, j.p.cnt, j.p.stl, j.p.bds, j.name, j.state == Job::SKIPPED ? 0.0 : j.secs(), j.state == Job::SKIPPED ? "output exists" : j.err);
    std::lock_guard<std::mutex> g(g_print); fputs(j.log.c_str(), stdout); fflush(stdout);
}

void load_pair(const Config& cfg, Job& j) {
    j.t0 = std::chrono::steady_clock::now();
    j.log += "-----------------***********************----------------------\n";
    j.say("Content: %s, style: %s, BDS weight: %f.\n", j.p.cnt.c_str(), j.p.stl.c_str(), (double)j.p.bds);
    const std::string cntStr = cfg.input_dir + "/" + j.p.cnt, stlStr = cfg.input_dir + "/" + j.p.stl;
    char name[1024];
    snprintf(name, sizeof name, "%s/%s_%s_%2.2f.png", cfg.output_dir.c_str(), stem(cntStr).c_str(), stem(stlStr).c_str(), (double)j.p.bds);   // main.cu:524-537
    j.name = name;
    if (cfg.resume && pngio::looks_complete(j.name)) {           // a truncated file (killed run, full disk) is redone, not skipped
        j.say("Skipping (-resume): %s exists.\n\n", name);
        j.state = Job::SKIPPED; return;
    }
    std::string err;
    if (!imgio::read(cntStr, j.cnt, err)) { j.say("Error: Fail reading content image: %s\n", cntStr.c_str()); j.err = "cannot read content image: " + err; j.state = Job::FAILED; return; }
    j.say("\n**Read content file: %s, w = %d, h = %d\n", cntStr.c_str(), j.cnt.w, j.cnt.h);
    if (!imgio::read(stlStr, j.stl, err)) { j.say("Error: Fail reading style image: %s\n", stlStr.c_str()); j.err = "cannot read style image: " + err; j.state = Job::FAILED; return; }
    j.say("Read style file: %s, w = %d, h = %d\n", stlStr.c_str(), j.stl.w, j.stl.h);
}

void run_pair(nct_ctx* ctx, const Config& cfg, Job& j) {
    if (!shrink(ctx, j.cnt) || !shrink(ctx, j.stl)) { j.say("Error: resize failed: %s\n", nct_last_error(ctx)); j.err = nct_last_error(ctx); j.state = Job::FAILED; return; }
    nct_params prm = cfg.prm;
    prm.bds_weight = j.p.bds;                                   // the per-line weight overrides -bds (main.cu:475)
    j.out.resize((size_t)j.cnt.h * j.cnt.w * 3);
    nct_pair_timing tm;                                         // stage times come from stream events: asking for them adds no host synchronisation
    if (cfg.vis) {
        std::string pre(j.name); pre.resize(pre.size() - 4);    // the output file's stem
        std::string err;
        if (!run_with_vis(ctx, j.cnt, j.stl, prm, pre, j.out.data(), &tm, err)) { j.say("Error: %s\n", err.c_str()); j.err = err; j.state = Job::FAILED; return; }
    } else {
        const int rc = nct_process_pair(ctx, j.cnt.px.data(), j.cnt.h, j.cnt.w, j.stl.px.data(), j.stl.h, j.stl.w, &prm, j.out.data(), &tm);
        if (rc != NCT_OK) { j.say("Error: %s\n", nct_last_error(ctx)); j.err = nct_last_error(ctx); j.state = Job::FAILED; return; }
    }
    // the reference's per-level lines (main.cu:331; ColorTransfer.cpp:1373,1434), then its total (main.cu:453)
    for (int l = 0; l < prm.levels; ++l) {
        j.say("Patch Match Time: %lf sec.\n", (tm.pm_level_ms[l] + tm.vote_level_ms[l]) * 1e-3);
        j.say("Nonlocal Solve Time: %lf\n", tm.nonlocal_level_ms[l] * 1e-3);
        j.say("WLS Solve Time: %lf\n", tm.wls_level_ms[l] * 1e-3);
    }
    j.say("VGG19 Time: %lf sec.\n", tm.vgg_ms * 1e-3);
    j.say("**Finished Time: %lf sec.\n", tm.total_ms * 1e-3);
    j.stl.px.clear(); j.stl.px.shrink_to_fit();
    j.cnt.px.clear(); j.cnt.px.shrink_to_fit();                 // the store stage needs only cnt.h / cnt.w
}

void store_pair(Job& j) {
    std::string err;
    if (!pngio::write(j.name, j.out.data(), j.cnt.h, j.cnt.w, err)) { j.say("Error: cannot write %s: %s\n", j.name.c_str(), err.c_str()); j.err = "cannot write output: " + err; j.state = Job::FAILED; return; }
    j.say("Final output file: %s.\n\n", j.name.c_str());
    j.state = Job::DONE;
}

// Bounded hand-over between the I/O pool and the GPU workers. `ready` holds decoded pairs (at most `cap`: the decoders stay a little ahead of the GPUs, not a
// whole batch), `results` finished ones waiting for the PNG encoder (the same bound: a worker blocks rather than pile up results if zlib falls behind).
// Which pairs.txt lines this process runs. One process (-world 1): all of them, in order. One process per GPU (-world N -rank r; what `-procs N` forks): line i belongs to
// rank i mod N — or, with -steal 1, to whichever rank draws it: a counter in <output>/.tickets, advanced under an fcntl lock, hands the lines out in order to whoever is
// free (BASELINE config 5, mixed sizes; north_star's "work-stealing" — a file lock rather than RCCL: the ranks exchange one integer per pair, and a lock file also works
// between ranks that were started by hand on different devices, with no rendezvous). Which rank runs a pair has no influence on its result.
struct Tickets {
    size_t total = 0, next = 0; int rank = 0, world = 1, fd = -1;
    long draw() {                                                // next global line index, -1 when there is none left for this process
        if (fd >= 0) {
            struct flock lk; memset(&lk, 0, sizeof lk); lk.l_type = F_WRLCK; lk.l_whence = SEEK_SET;
            if (fcntl(fd, F_SETLKW, &lk) != 0) return -1;
            unsigned long long v = 0;
            if (pread(fd, &v, sizeof v, 0) != (ssize_t)sizeof v) v = 0;
            const long got = v < total ? (long)v : -1;
            if (got >= 0) { ++v; if (pwrite(fd, &v, sizeof v, 0) != (ssize_t)sizeof v) { /* the lock is released below; the next reader sees the old value and redoes the line: harmless */ } }
            lk.l_type = F_UNLCK; (void)fcntl(fd, F_SETLK, &lk);
            return got;
        }
        while (next < total && (int)(next % (size_t)world) != rank) ++next;
        return next < total ? (long)next++ : -1;
    }
};

// Bounded hand-over between the I/O pool and the GPU workers. `ready` holds decoded pairs (at most `cap`: the decoders stay a little ahead of the GPUs, not a
// whole batch), `results` finished ones waiting for the PNG encoder (the same bound: a worker blocks rather than pile up results if zlib falls behind).
struct Pipeline {
    std::mutex m; std::condition_variable cv;
    std::deque<std::unique_ptr<Job>> ready, results;
    Tickets tickets;
    size_t cap = 4, taken = 0, loading = 0, finished = 0;
    bool exhausted = false;                                      // the ticket source has nothing left for this process
    // decoded pixels waiting for a GPU worker: images are queued BEFORE the GPU-side shrink to MAX_SIZE and a decoder accepts up to 64 MP (192 MB), so the queue is
    // bounded by bytes as well as by count — a new load starts only while the decoded backlog is below byte_cap (or nothing at all is queued or loading)
    // Loads in flight are charged too: a load reserves `load_estimate` bytes when it starts — the largest decoded pair seen so far, at least two 1000 x 1000
    // images — and is corrected to its real size when it lands in `ready`, so several I/O threads cannot all pass the test while the backlog is still being decoded.
    size_t ready_bytes = 0, byte_cap = (size_t)1 << 30, loading_bytes = 0, load_estimate = (size_t)6 << 20;
    bool may_load() const { return !exhausted && ready.size() + loading < cap && (ready_bytes + loading_bytes < byte_cap || ready.size() + loading == 0); }
    bool loads_done() const { return exhausted && loading == 0; }
    bool all_done() const { return exhausted && finished == taken; }
};

void io_thread(Pipeline& P, const Config& cfg, const std::vector<Pair>& pairs) {
    for (;;) {
        std::unique_ptr<Job> j; bool store = false; size_t idx = 0, reserved = 0;
        {
            std::unique_lock<std::mutex> lk(P.m);
            P.cv.wait(lk, [&] { return !P.results.empty() || P.may_load() || P.all_done(); });
            if (!P.results.empty()) { j = std::move(P.results.front()); P.results.pop_front(); store = true; }       // encoding first: it frees memory and unblocks workers
            else if (P.may_load()) {
                const long t = P.tickets.draw();
                if (t < 0) { P.exhausted = true; lk.unlock(); P.cv.notify_all(); continue; }
                idx = (size_t)t; ++P.taken; ++P.loading; reserved = P.load_estimate; P.loading_bytes += reserved;
            }
            else return;                                                                                                 // every ticket of this process is finished
        }
        P.cv.notify_all();
        if (store) {
            store_pair(*j); finish(cfg, *j);
            { std::lock_guard<std::mutex> lk(P.m); ++P.finished; }
        } else {
            j.reset(new Job()); j->index = idx; j->p = pairs[idx];
            load_pair(cfg, *j);
            const bool go = j->state == Job::LOADED;
            if (!go) finish(cfg, *j);
            std::lock_guard<std::mutex> lk(P.m);
            --P.loading; P.loading_bytes -= reserved;
            if (go) {
                const size_t b = j->cnt.px.size() + j->stl.px.size();
                P.ready_bytes += b; P.load_estimate = std::max(P.load_estimate, b);
                P.ready.push_back(std::move(j));
            } else ++P.finished;
        }
        P.cv.notify_all();
    }
}

void gpu_worker(Pipeline& P, nct_ctx* ctx, const Config& cfg) {
    for (;;) {
        std::unique_ptr<Job> j;
        {
            std::unique_lock<std::mutex> lk(P.m);
            P.cv.wait(lk, [&] { return !P.ready.empty() || P.loads_done(); });
            if (P.ready.empty()) return;
            j = std::move(P.ready.front()); P.ready.pop_front();
            P.ready_bytes -= j->cnt.px.size() + j->stl.px.size();
        }
        P.cv.notify_all();
        run_pair(ctx, cfg, *j);
        if (j->state == Job::FAILED) {
            finish(cfg, *j);
            { std::lock_guard<std::mutex> lk(P.m); ++P.finished; }
        } else {
            std::unique_lock<std::mutex> lk(P.m);
            P.cv.wait(lk, [&] { return P.results.size() < P.cap; });
            P.results.push_back(std::move(j));
        }
        P.cv.notify_all();
    }
}
}  // namespace

int main(int argc, char** argv) {
    if (argc == 4 && !strcmp(argv[1], "--png-roundtrip")) {       // codec self-test hook (no GPU): decode argv[2] (PNG or JPEG), re-encode to argv[3]
        ImageBGR im; std::string err;
        if (!imgio::read(argv[2], im, err)) { printf("Error: %s: %s\n", argv[2], err.c_str()); return 1; }
        if (!pngio::write(argv[3], im.px.data(), im.h, im.w, err)) { printf("Error: %s: %s\n", argv[3], err.c_str()); return 1; }
        printf("%d %d\n", im.w, im.h);
        return 0;
    }
    if (argc == 3 && !strcmp(argv[1], "--gpu-locality")) {        // affinity self-test hook (no GPU): NUMA node and CPU list of the PCI device argv[2]
        const affinity::GpuLocality g = affinity::gpu_locality(argv[2]);
        printf("%d %s\n", g.numa_node, affinity::cpus_to_string(g.cpus).c_str());
        return 0;
    }
    if (argc == 3 && !strcmp(argv[1], "--check-prototxt")) {      // deploy-prototxt self-test hook (no GPU)
        if (nct_vgg19_check_prototxt(nullptr, argv[2]) != NCT_OK) { printf("Error: %s\n", nct_model_last_error()); return 1; }
        printf("ok\n");
        return 0;
    }
    if (nct_version() != NCT_VERSION) { printf("Error: libnct is version %d, this driver was built against %d.\n", nct_version(), NCT_VERSION); return -1; }
    CmdLine cl;
    Config cfg;
    nct_params_default(&cfg.prm);
    int gpu = 0, ngpus = 1, seed = 1, inflight = 1, levels = 5, resume = 0, feat16 = 0, vis = 0, io = -1, pin = 1, world = 1, rank = 0, steal = 0, procs = 0, rccl = 0;
    cl.add("m", cfg.model_dir, "Directory of network models.");
    cl.add("i", cfg.input_dir, "Input directory of content and style images and pairs.txt.");
    cl.add("o", cfg.output_dir, "Output directory of result images.");
    // the reference's comment strings (main.cu:32-43); its "(default: ...)" remarks quote other numbers than Config::Config() sets
    // (Config.h:58-72) — the leading "(default=...)" that CmdLine prints (CmdLine.h:140-142) is the value that really applies
    cl.add("g", gpu, "GPU ID (default: 0).");
    cl.add("bds", cfg.prm.bds_weight, "Weight of reverse color in BDS voting (default: 2.0).");
    cl.add("eps", cfg.prm.eps, "Eps is used to avoid dividing zero (default: 0.6 with range in [0-255]).");
    cl.add("nl", cfg.prm.nonlocal_weight, "Weight of nonlocal constraint (default: 0.4.");
    cl.add("l", cfg.prm.local_weight, "Weight of local constraitn (default: 0.001).");
    cl.add("w", cfg.prm.wls_lambda_init, "Initial value of WLS weight (default: 0.0234375).");
    cl.add("gpus", ngpus, "[extension] number of GPUs to shard pairs.txt over, starting at -g (default: 1).");
    cl.add("inflight", inflight, "[extension] pairs in flight per GPU, one context + host thread each (default: 1; 2-4 raises throughput ~20 %).");
    cl.add("io", io, "[extension] threads of the shared decode/encode pool (default -1: two per GPU, at most the machine's; 0: every GPU worker does its own file I/O).");
    cl.add("pin", pin, "[extension] 1 = pin each GPU's worker and I/O threads to the CPUs of the GPU's NUMA node (sysfs local_cpulist); 0 = leave the scheduler alone.");
    cl.add("seed", seed, "[extension] seed of the counter-based RNG (default: 1).");
    cl.add("levels", levels, "[extension] pyramid levels to run, coarse to fine: 5 = the full L=5..1 loop, 1 = L=5 only.");
    cl.add("resume", resume, "[extension] 1 = skip pairs whose output file exists and is a complete PNG; every pair appends a JSON line to <output>/status.jsonl.");
    cl.add("vis", vis, "[extension] 1 = the reference's ENABLE_VIS dumps per level (flow maps, level images, error heat map, coefficient and cluster images) next to the output.");
    cl.add("procs", procs, "[extension] N > 0: fork N processes, one per GPU (-g, -g + 1, ...): process r runs with -rank r -world N on its own device, HIP runtime and status.<r>.jsonl (the process-per-GPU shape; -gpus N keeps all GPUs in one process).");
    cl.add("world", world, "[extension] number of cooperating processes that share this pairs.txt and output directory (default 1); set by -procs, or by hand with -rank.");
    cl.add("rank", rank, "[extension] this process's rank in [0, world): it runs the pairs.txt lines i with i mod world = rank (or the ones it draws, -steal 1).");
    cl.add("rccl", rccl, "[extension] 1 = the ranks of -procs / -world form an RCCL communicator (one rank per GPU, over xGMI on a node): a start barrier, and the job's time = MAX over ranks and its pair count = SUM over ranks by all-reduce, printed by rank 0. Nothing of a pair's data crosses GPUs; without RCCL (or with 0, the default) the ranks simply run.");
    cl.add("steal", steal, "[extension] 1 = with -world > 1, lines are drawn from a shared counter (<output>/.tickets under a file lock) by whichever rank is free, instead of i mod world: mixed-size batches.");
    cl.add("feat16", feat16, "[extension] 1 = fp16 PatchMatch feature tiles (fp32 accumulate); not bit-identical to the default (about 45 dB against it).");
    // parser self-test hook (no GPU): `--parse-only <args…>` parses the rest like a normal run and prints what main would go on with, in the format of
    // oracle/ref_cmdline.cpp (the reference's own parser): tests/test_cli.py compares the two on the vectors of tests/golden/cmdline_ref.json
    const bool parse_only = argc >= 2 && !strcmp(argv[1], "--parse-only");
    const bool parsed = cl.parse(argc, argv, parse_only ? 2 : 1);
    if (parse_only) {
        std::cout << std::flush;
        printf("@@RESULT rc=%d\n", parsed ? 1 : 0);
        printf("m=%s\ni=%s\no=%s\ng=%d\n", cfg.model_dir.c_str(), cfg.input_dir.c_str(), cfg.output_dir.c_str(), gpu);
        printf("bds=%.17g\neps=%.17g\nnl=%.17g\nl=%.17g\nw=%.17g\n", cfg.prm.bds_weight, cfg.prm.eps, cfg.prm.nonlocal_weight, cfg.prm.local_weight, cfg.prm.wls_lambda_init);
        printf("files=%d\n", cl.files);
        return 0;
    }
    if (!parsed) return -1;
    if (world < 1 || rank < 0 || rank >= world) { printf("Error: -rank %d is not in [0, -world %d).\n", rank, world); return -1; }
    mkdir(cfg.output_dir.c_str(), 0777);                                    // main.cu:458
    uint64_t run_token = getenv("NCT_RUN_TOKEN") ? strtoull(getenv("NCT_RUN_TOKEN"), nullptr, 0) : 0;      // hand-started ranks of one run share it (and remove <output>/.rccl_id between runs)
    const std::string tickets_path = cfg.output_dir + "/.tickets";
    if (procs > 0) {
        // one process per GPU: fork BEFORE anything touches the HIP runtime (a forked HIP context is unusable), every child goes on as rank r of `procs` on device -g + r
        if (world != 1) { printf("Error: -procs and -world are exclusive (-procs sets -world for its children).\n"); return -1; }
        if (steal) { unlink(tickets_path.c_str()); }                         // a fresh counter for this run
        run_token = ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();      // children are forks: they inherit it; an id file of another run carries another token
        fflush(stdout);
        std::vector<pid_t> kids;
        int my = -1;
        for (int r = 0; r < procs; ++r) {
            const pid_t k = fork();
            if (k < 0) { printf("Error: fork failed.\n"); return -1; }
            if (k == 0) { my = r; break; }
            kids.push_back(k);
        }
        if (my < 0) {                                                       // the parent only waits: exit status = the worst child's
            const auto t0 = std::chrono::steady_clock::now();
            int worst = 0;
            for (pid_t k : kids) { int st = 0; if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) worst = -1; }
            printf("All %d process(es) finished in %.3f sec%s.\n", procs, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), worst ? " (at least one failed)" : "");
            return worst;
        }
        world = procs; rank = my; gpu += my; ngpus = 1;
    }
    cfg.rank = rank; cfg.world = world;
    cfg.prm.seed = (uint32_t)seed;
    cfg.prm.levels = levels < 1 ? 1 : (levels > 5 ? 5 : levels);
    if (feat16) cfg.prm.flags |= NCT_FLAG_FEAT16;
    if (inflight <= 1) cfg.prm.flags |= NCT_FLAG_LATENCY;          // one pair at a time per GPU: split WLS solves (same result, -3 ms per 700x700 pair)
    cfg.resume = resume != 0;
    cfg.vis = vis != 0;
    if (ngpus < 1) ngpus = 1;
    if (inflight < 1) inflight = 1;
    if (inflight > 8) inflight = 8;
    const int nworkers = ngpus * inflight;
    const int hw = (int)std::thread::hardware_concurrency();
    if (io < 0) io = std::min(2 * ngpus, hw > 0 ? hw : 2 * ngpus);
    if (io > 64) io = 64;

    const std::string pairsFile = cfg.input_dir + "/pairs.txt";
    FILE* fp = fopen(pairsFile.c_str(), "r");
    if (!fp) { printf("Error: File %s does not exist in the input directory.\n", pairsFile.c_str()); return -1; }
    std::vector<Pair> pairs;
    char a[260], b[260]; float w = 0.f;
    while (fscanf(fp, "%259s %259s %f\n", a, b, &w) == 3) pairs.push_back({a, b, w});
    fclose(fp);

    // model: <model_dir>/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt + VGG_ILSVRC_19_layers.caffemodel (main.cu:575-580; '\\' or '/' accepted in model_dir).
    // The topology is built into the library, so the prototxt is only checked: a directory that describes another network is refused, a missing file is noted.
    const std::string proto = cfg.model_dir + "/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt", model = cfg.model_dir + "/vgg19/VGG_ILSVRC_19_layers.caffemodel";
    struct stat pst;
    if (stat(proto.c_str(), &pst) == 0) {
        if (nct_vgg19_check_prototxt(nullptr, proto.c_str()) != NCT_OK) { printf("Error: %s\n", nct_model_last_error()); return -1; }
    } else printf("Note: %s not found; using the library's built-in VGG19 topology (conv1_1 ... relu5_1).\n", proto.c_str());

    // test hook: NCT_DEVICE_OVERRIDE=d runs every logical GPU of -gpus N on HIP device d (the N > 1 host path on a 1-GPU box)
    const char* ovr = getenv("NCT_DEVICE_OVERRIDE");
    auto device_of = [&](int g) { return ovr && *ovr ? atoi(ovr) : gpu + g; };
    // one context (streams, arena) per worker; worker j runs on GPU j mod G, so -inflight K gives every GPU K independent pairs whose
    // launch-latency-bound phases (coarse pyramid levels, solver reductions) overlap with the other pairs' heavy kernels.
    // Weights: the 575 MB caffemodel is parsed ONCE per process, uploaded ONCE per device, and every other context of that device shares the read-only copy
    // (the reference parses it twice, main.cu:581-582; round 2 of this CLI parsed and uploaded it once per worker).
    std::vector<nct_ctx*> ctxs(nworkers, nullptr);
    nct_model* host_model = nullptr;
    if (nct_model_parse_caffemodel(model.c_str(), &host_model) != NCT_OK) { printf("Error: %s\n", nct_model_last_error()); return -1; }
    int uploads = 0;
    for (int j = 0; j < nworkers; ++j) {
        const int g = j % ngpus, dev = device_of(g);
        if (nct_create(dev, &ctxs[j]) != NCT_OK) { printf("Error: %s\n", nct_last_error(nullptr)); return -1; }
        if (j < ngpus) { char name[256]; nct_device_name(ctxs[j], name, sizeof name); printf("Set device %d: %s.\n", dev, name); }
        int owner = -1;
        for (int k = 0; k < j; ++k) if (device_of(k % ngpus) == dev) { owner = k; break; }
        const int rc = owner < 0 ? (++uploads, nct_vgg19_load_model(ctxs[j], host_model)) : nct_vgg19_share_weights(ctxs[j], ctxs[owner]);
        if (rc != NCT_OK) { printf("Error: %s\n", nct_last_error(ctxs[j])); return -1; }
    }
    nct_model_free(host_model);
    { size_t wb = 0; nct_vgg19_weights_info(ctxs[0], nullptr, &wb, nullptr);
      printf("VGG19 weights: parsed once, %d device cop%s of %.1f MB shared by %d context(s).\n", uploads, uploads == 1 ? "y" : "ies", wb / 1e6, nworkers); }

    // NUMA placement: the CPUs next to each GPU (sysfs), for its workers and its share of the I/O pool
    std::vector<affinity::GpuLocality> loc(ngpus);
    if (pin)
        for (int g = 0; g < ngpus; ++g) {
            char addr[32];
            if (nct_device_pci_bus_id(device_of(g), addr, sizeof addr) != NCT_OK) continue;
            loc[g] = affinity::gpu_locality(addr);
            if (!loc[g].cpus.empty()) printf("GPU %d (%s): NUMA node %d, host threads pinned to CPUs %s.\n", device_of(g), addr, loc[g].numa_node, affinity::cpus_to_string(loc[g].cpus).c_str());
        }

    // -rccl 1: one communicator over the ranks of this run (one process per GPU), used for the start barrier and for the two reductions at the end — never on a pair's data path
    rccl_sync::Group rg;
    const std::string rccl_id_path = cfg.output_dir + "/.rccl_id";
    if (rccl) {
        if (ngpus != 1) printf("Note: -rccl 1 is for one process per GPU (-procs / -world); this process drives %d GPUs and joins with its first.\n", ngpus);
        if (!rg.init(world, rank, device_of(0), rccl_id_path, run_token) || !rg.barrier())
            printf("Note: -rccl 1: no RCCL group (%s); rank %d goes on without the barrier.\n", rg.why().c_str(), rank);
        else printf("RCCL: rank %d of %d joined, start barrier passed.\n", rank, world);
    }
    const auto t0 = std::chrono::steady_clock::now();
    // pairs are independent and of mixed sizes: every worker takes the next decoded pair (work stealing inside the node, BASELINE config 5);
    // which worker runs a pair has no influence on its result
    std::vector<std::thread> threads;
    Pipeline P; P.cap = (size_t)std::max(2, 2 * nworkers);
    P.tickets.total = pairs.size(); P.tickets.rank = rank; P.tickets.world = world;
    if (world > 1 && steal) {
        P.tickets.fd = open(tickets_path.c_str(), O_RDWR | O_CREAT, 0644);     // 8 bytes: the next line to hand out (absent or short = 0). A hand-started set of ranks removes it between runs.
        if (P.tickets.fd < 0) { printf("Error: cannot open %s for -steal.\n", tickets_path.c_str()); return -1; }
    }
    if (const char* e = getenv("NCT_IO_READY_MB")) P.byte_cap = (size_t)std::max(0L, atol(e)) << 20;      // test hook: decoded backlog allowed in front of the GPU workers (default 1 GiB)
    std::mutex next_m;
    size_t mine = 0;
    if (io > 0) {
        for (int t = 0; t < io; ++t) threads.emplace_back([&, t] { if (pin) affinity::pin_current_thread(loc[t % ngpus].cpus); io_thread(P, cfg, pairs); });
        for (int j = 0; j < nworkers; ++j) threads.emplace_back([&, j] { if (pin) affinity::pin_current_thread(loc[j % ngpus].cpus); gpu_worker(P, ctxs[j], cfg); });
    } else {
        for (int j = 0; j < nworkers; ++j)
            threads.emplace_back([&, j] {
                if (pin) affinity::pin_current_thread(loc[j % ngpus].cpus);
                for (;;) {
                    long t; { std::lock_guard<std::mutex> lk(next_m); t = P.tickets.draw(); if (t >= 0) ++mine; }
                    if (t < 0) break;
                    const size_t i = (size_t)t;
                    Job job; job.index = i; job.p = pairs[i];
                    load_pair(cfg, job);
                    if (job.state == Job::LOADED) run_pair(ctxs[j], cfg, job);
                    if (job.state == Job::LOADED) store_pair(job);
                    finish(cfg, job);
                }
            });
    }
    for (auto& t : threads) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t done = io > 0 ? P.taken : mine;
    if (P.tickets.fd >= 0) close(P.tickets.fd);
    if (world > 1) printf("Rank %d of %d: ", rank, world);
    printf("Processed %zu pair(s) on %d GPU(s), %d in flight each, %d I/O thread(s), in %.3f sec (%.3f pairs/sec).\n", done, ngpus, inflight, io, sec, done == 0 ? 0.0 : done / sec);
    if (rg.ok()) {
        double sec_max = sec, total = (double)done;
        const bool r1 = rg.reduce(sec, rccl_sync::kMax, &sec_max), r2 = rg.reduce((double)done, rccl_sync::kSum, &total);
        if (r1 && r2) { if (rank == 0) printf("All %d rank(s) over RCCL: %.0f pair(s) in %.3f sec = MAX over ranks (%.3f pairs/sec).\n", world, total, sec_max, sec_max > 0 ? total / sec_max : 0.0); }
        else printf("Note: RCCL reduction failed (%s).\n", rg.why().c_str());
        rg.finish(rccl_id_path);
    }
    for (auto* c : ctxs) nct_destroy(c);
    return 0;
}
