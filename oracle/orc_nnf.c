/* oracle/orc_nnf.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md). "parity unpinned" (no reference vectors exist).
 *
 * CPU restatement of the reference's correspondence sub-path:
 *   N1  norm                 GeneralizedPatchMatch.cu:237-283
 *   N2  init_Ann_kernel      GeneralizedPatchMatch.cu:527-544
 *       upSample_kernel      GeneralizedPatchMatch.cu:546-580
 *   P1  patchmatch_single    GeneralizedPatchMatch.cu:677-831
 *       dist_compute_single  GeneralizedPatchMatch.cu:355-405
 *       improve_guess_single GeneralizedPatchMatch.cu:505-515
 *   B2  feature_distance     GeneralizedPatchMatch.cu:833-855
 *
 * Feature tensors at this boundary are CHW fp32 exactly like the reference kernels' a1/b1 arguments.
 *
 * Two documented divergences from the reference (SPEC.md):
 *  (1) schedule: the reference kernel is one racy launch (both __syncthreads commented out, :801,:827).
 *      Here every (iteration, jump) step is a Jacobi step on a double-buffered NNF: all queries read the
 *      previous step's NNF/dist, write the next one. Random search is fused into the jump==1 step.
 *  (2) fp32 summation order of the channel dot product is FIXED so a GPU can reproduce it bit-for-bit:
 *      16 "virtual lanes"; lane v owns the float4 channel chunks j = v, v+16, v+32 …; one fmaf chain per
 *      lane over (valid taps in dy-outer/dx-inner order) x (its chunks ascending) x (4 components);
 *      then a 16-lane xor butterfly (8,4,2,1). The reference sums channels sequentially per tap
 *      (pixel_sum1 -= a*b); the value differs only by fp32 rounding.
 */
#include "orc_common.h"
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* thread control for the test harness: the GPU boxes expose 256 host cores and libgomp's default team (one thread
 * per core, spinning) turns the thousands of tiny parallel regions of the solvers into a 600x slowdown. */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- layout helper: CHW -> HWC with channel padding to a multiple of 4 (zeros) ---- */
static float* chw_to_hwc(const float* src, int C, int H, int W, int* Cpad_out) {
    int Cp = (C + 3) & ~3;
    float* dst = (float*)calloc((size_t)H * W * Cp, sizeof(float));
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H * W; ++i) dst[(size_t)i * Cp + c] = src[(size_t)c * H * W + i];
    *Cpad_out = Cp;
    return dst;
}

static inline float butterfly16(float acc[16]) {
    for (int off = 8; off >= 1; off >>= 1) {
        float t[16];
        for (int v = 0; v < 16; ++v) t[v] = acc[v] + acc[v ^ off];
        memcpy(acc, t, sizeof(t));
    }
    return acc[0];
}

/* canonical dot product of two HWC pixel vectors accumulated INTO acc[16] (fmaf chains) */
static inline void dot_accum16(float acc[16], const float* a, const float* b, int Cp) {
    int nchunk = Cp >> 2;
    for (int v = 0; v < 16; ++v)
        for (int j = v; j < nchunk; j += 16)
            for (int k = 0; k < 4; ++k) acc[v] = fmaf(a[4 * j + k], b[4 * j + k], acc[v]);
}

/* N1 — norm(): dst = src / sqrt(sum_c src^2); optional response map (dis - min) * (1/(max-min)).
 * No epsilon (0/0 -> NaN like the reference, :276-277). */
void orc_feat_normalize(const float* src_chw, float* dst_chw, float* resp, int C, int H, int W) {
    int Cp; float* s = chw_to_hwc(src_chw, C, H, W, &Cp);
    int n = H * W;
    float* dis = (float*)malloc(sizeof(float) * n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float acc[16] = {0};
        dot_accum16(acc, s + (size_t)i * Cp, s + (size_t)i * Cp, Cp);
        float d = sqrtf(butterfly16(acc));           /* caffe_gpu_powx(sum, 0.5) */
        dis[i] = d;
        for (int c = 0; c < C; ++c) dst_chw[(size_t)c * n + i] = src_chw[(size_t)c * n + i] / d;   /* caffe_gpu_div */
    }
    if (resp) {
        float mn = dis[0], mx = dis[0];
        for (int i = 1; i < n; ++i) { if (dis[i] < mn) mn = dis[i]; if (dis[i] > mx) mx = dis[i]; }
        float sc = 1.0f / (mx - mn);
        for (int i = 0; i < n; ++i) resp[i] = (dis[i] + (-mn)) * sc;   /* add_scalar(-minv) ; scal(1/(max-min)) */
    }
    free(dis); free(s);
}

/* N2 — init_Ann_kernel */
void orc_nnf_init(uint32_t* nnf, int ah, int aw, int bh, int bw) {
    for (int ay = 0; ay < ah; ++ay)
        for (int ax = 0; ax < aw; ++ax) {
            int bx = (int)((float)ax / (float)(aw - 1) * (bw - 1));
            int by = (int)((float)ay / (float)(ah - 1) * (bh - 1));
            if (bx > bw - 1) bx = bw - 1;
            if (by > bh - 1) by = bh - 1;
            nnf[ay * aw + ax] = orc_xy_to_int(bx, by);
        }
}

/* N2 — upSample_kernel. Note the double-precision intermediates: (ax+0.5) is int+double, divided by a float
 * ratio promoted to double; (bx_half-ax_half)*aw_ratio is int*float -> float, then ax + that + 0.5 in double. */
void orc_nnf_upsample(const uint32_t* nnf_half, uint32_t* nnf, int ah, int aw, int bh, int bw, int ah_half, int aw_half) {
    float aw_ratio = (float)aw / (float)aw_half;
    float ah_ratio = (float)ah / (float)ah_half;
    for (int ay = 0; ay < ah; ++ay)
        for (int ax = 0; ax < aw; ++ax) {
            int ax_half = (int)((ax + 0.5) / aw_ratio);
            int ay_half = (int)((ay + 0.5) / ah_ratio);
            ax_half = orc_clamp(ax_half, aw_half - 1, 0);
            ay_half = orc_clamp(ay_half, ah_half - 1, 0);
            uint32_t v = nnf_half[ay_half * aw_half + ax_half];
            int bx_half = orc_int_to_x(v), by_half = orc_int_to_y(v);
            int bx = (int)(ax + (bx_half - ax_half) * aw_ratio + 0.5);
            int by = (int)(ay + (by_half - ay_half) * ah_ratio + 0.5);
            bx = orc_clamp(bx, bw - 1, 0);
            by = orc_clamp(by, bh - 1, 0);
            nnf[ay * aw + ax] = orc_xy_to_int(bx, by);
        }
}

/* dist_compute_single with weight = 1, flag_constraint = 0 (main.cu:212-214; dist_constraint is dead). */
static float patch_dist(const float* A, const float* B, int Cp, int ah, int aw, int bh, int bw,
                        int ax, int ay, int bx, int by, int patch, float cutoff) {
    float acc[16] = {0};
    float pixel_no = 0;
    int r = patch / 2;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            if ((ay + dy) < ah && (ay + dy) >= 0 && (ax + dx) < aw && (ax + dx) >= 0 &&
                (by + dy) < bh && (by + dy) >= 0 && (bx + dx) < bw && (bx + dx) >= 0) {
                dot_accum16(acc, A + ((size_t)(ay + dy) * aw + (ax + dx)) * Cp, B + ((size_t)(by + dy) * bw + (bx + dx)) * Cp, Cp);
                pixel_no += 1;
            }
        }
    float pixel_dist;
    if (pixel_no == 0) pixel_dist = 1;
    else pixel_dist = (0.0f + 1.0f * (-butterfly16(acc))) / pixel_no;   /* (pixel_sum + weight*pixel_sum1)/pixel_no */
    if (pixel_dist >= cutoff) return cutoff;
    return pixel_dist;
}

static inline void improve(const float* A, const float* B, int Cp, int ah, int aw, int bh, int bw, int ax, int ay,
                           int* xbest, int* ybest, float* dbest, int xp, int yp, int patch, float rr) {
    float d = patch_dist(A, B, Cp, ah, aw, bh, bw, ax, ay, xp, yp, patch, *dbest);
    if (d + rr < *dbest) { *xbest = xp; *ybest = yp; *dbest = d; }
}

/* diagnostic: number of distance evaluations of the last orc_patchmatch call (for the algorithmic-bytes model) */
static long long g_last_evals = 0;
long long orc_patchmatch_last_evals(void) { return g_last_evals; }

/* P1 — patchmatch_single under the Jacobi schedule described in the header.
 * a_chw/b_chw: L2-normalised features (C,ah,aw)/(C,bh,bw). nnf: in/out (ah*aw). dist: out (ah*aw). */
void orc_patchmatch(const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw,
                    int patch, int iters, int rs_max, uint32_t seed, uint32_t* nnf, float* dist) {
    int Cp, Cp2;
    float* A = chw_to_hwc(a_chw, C, ah, aw, &Cp);
    float* B = chw_to_hwc(b_chw, C, bh, bw, &Cp2);
    int n = ah * aw;
    uint32_t* nnf_in = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t* nnf_out = (uint32_t*)malloc(sizeof(uint32_t) * n);
    float* d_in = (float*)malloc(sizeof(float) * n);
    float* d_out = (float*)malloc(sizeof(float) * n);
    memcpy(nnf_in, nnf, sizeof(uint32_t) * n);
    long long evals = 0;

    /* annd = dist(current) with no cutoff (:712-714) */
#pragma omp parallel for schedule(dynamic, 4) reduction(+:evals)
    for (int ay = 0; ay < ah; ++ay)
        for (int ax = 0; ax < aw; ++ax) {
            uint32_t v = nnf_in[ay * aw + ax];
            d_in[ay * aw + ax] = patch_dist(A, B, Cp, ah, aw, bh, bw, ax, ay, orc_int_to_x(v), orc_int_to_y(v), patch, (float)INT32_MAX);
            evals++;
        }

    for (int iter = 0; iter < iters; ++iter) {
        for (int jump = 8; jump > 0; jump /= 2) {
#pragma omp parallel for schedule(dynamic, 4) reduction(+:evals)
            for (int ay = 0; ay < ah; ++ay)
                for (int ax = 0; ax < aw; ++ax) {
                    uint32_t v = nnf_in[ay * aw + ax];
                    int xbest = orc_int_to_x(v), ybest = orc_int_to_y(v);
                    float dbest = d_in[ay * aw + ax];
                    int xp, yp; uint32_t vp;
                    if ((ax - jump) < aw && (ax - jump) >= 0) {            /* left  (:725-739) */
                        vp = nnf_in[ay * aw + ax - jump];
                        xp = orc_int_to_x(vp) + jump; yp = orc_int_to_y(vp);
                        if (yp >= 0 && yp < bh && xp >= 0 && xp < bw) { improve(A, B, Cp, ah, aw, bh, bw, ax, ay, &xbest, &ybest, &dbest, xp, yp, patch, 0); evals++; }
                    }
                    if ((ax + jump) < aw) {                                 /* right (:743-758) */
                        vp = nnf_in[ay * aw + ax + jump];
                        xp = orc_int_to_x(vp) - jump; yp = orc_int_to_y(vp);
                        if (yp >= 0 && yp < bh && xp >= 0 && xp < bw) { improve(A, B, Cp, ah, aw, bh, bw, ax, ay, &xbest, &ybest, &dbest, xp, yp, patch, 0); evals++; }
                    }
                    if ((ay - jump) < ah && (ay - jump) >= 0) {            /* up    (:762-778) */
                        vp = nnf_in[(ay - jump) * aw + ax];
                        xp = orc_int_to_x(vp); yp = orc_int_to_y(vp) + jump;
                        if (yp >= 0 && yp < bh && xp >= 0 && xp < bw) { improve(A, B, Cp, ah, aw, bh, bw, ax, ay, &xbest, &ybest, &dbest, xp, yp, patch, 0); evals++; }
                    }
                    if ((ay + jump) < ah) {                                 /* down  (:780-796) */
                        vp = nnf_in[(ay + jump) * aw + ax];
                        xp = orc_int_to_x(vp); yp = orc_int_to_y(vp) - jump;
                        if (yp >= 0 && yp < bh && xp >= 0 && xp < bw) { improve(A, B, Cp, ah, aw, bh, bw, ax, ay, &xbest, &ybest, &dbest, xp, yp, patch, 0); evals++; }
                    }
                    if (jump == 1) {                                        /* random search (:804-820) */
                        int rs_start = rs_max;
                        int mx = bw > bh ? bw : bh;
                        if (rs_start > mx) rs_start = mx;
                        int step = 0;
                        for (int mag = rs_start; mag >= 1; mag /= 2, ++step) {
                            int xmin = xbest - mag > 0 ? xbest - mag : 0, xmax = xbest + mag + 1 < bw ? xbest + mag + 1 : bw;
                            int ymin = ybest - mag > 0 ? ybest - mag : 0, ymax = ybest + mag + 1 < bh ? ybest + mag + 1 : bh;
                            xp = xmin + (int)(orc_rand_u01(seed, ax, ay, iter, step, 0) * (xmax - xmin)) % (xmax - xmin);
                            yp = ymin + (int)(orc_rand_u01(seed, ax, ay, iter, step, 1) * (ymax - ymin)) % (ymax - ymin);
                            improve(A, B, Cp, ah, aw, bh, bw, ax, ay, &xbest, &ybest, &dbest, xp, yp, patch, FLT_MIN); evals++;
                        }
                    }
                    nnf_out[ay * aw + ax] = orc_xy_to_int(xbest, ybest);
                    d_out[ay * aw + ax] = dbest;
                }
            { uint32_t* t = nnf_in; nnf_in = nnf_out; nnf_out = t; }
            { float* t = d_in; d_in = d_out; d_out = t; }
        }
    }
    memcpy(nnf, nnf_in, sizeof(uint32_t) * n);
    memcpy(dist, d_in, sizeof(float) * n);
    g_last_evals = evals;
    free(nnf_in); free(nnf_out); free(d_in); free(d_out); free(A); free(B);
}

/* B2 — feature_distance: err(p) = -<a(p), b(p)> (canonical summation order, see header). */
void orc_feature_distance(const float* a_chw, const float* b_chw, float* err, int C, int H, int W) {
    int Cp, Cp2;
    float* A = chw_to_hwc(a_chw, C, H, W, &Cp);
    float* B = chw_to_hwc(b_chw, C, H, W, &Cp2);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < H * W; ++i) {
        float acc[16] = {0};
        dot_accum16(acc, A + (size_t)i * Cp, B + (size_t)i * Cp, Cp);
        err[i] = -butterfly16(acc);
    }
    free(A); free(B);
}
