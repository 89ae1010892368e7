/* oracle/orc_cvt.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md). "parity unpinned".
 *
 * Restatement of the OpenCV 2.4.10 calls the reference makes on the hot path. OpenCV is an
 * external dependency that is NOT under /root/reference (NuGet pin: code/windows/libcaffe/packages.config
 * "OpenCV 2.4.10") and is not installed here, so these follow OpenCV's published 8-bit algorithms
 * (modules/imgproc/src/color.cpp RGB2Lab_b / Lab2RGB_b / Lab2RGB_f, imgwarp.cpp resize) as documented in SURVEY
 * Appendix A. Call sites in the reference:
 *   cvtColor(BGR2Lab) 8U   main.cu:352,371; ColorTransfer.h:58
 *   cvtColor(Lab2BGR) 8U   ColorTransfer.cpp:1469
 *   resize INTER_LINEAR 8UC3  main.cu:106-107,509,521;  64FC3  ColorTransfer.cpp:462-463
 *   convertTo(CV_64F, 1/255), convertTo(CV_8U, 255)  main.cu:355-356,375; ColorTransfer.cpp:1468
 */
#include "orc_common.h"

static inline int cv_round(double v) { return (int)lrint(v); }            /* cvRound: round-half-to-even */
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline uint16_t sat_u16f(float v) { int i = cv_round(v); return (uint16_t)(i < 0 ? 0 : (i > 65535 ? 65535 : i)); }

/* ---------------------------------------------------------------- BGR <-> Lab, 8-bit */
enum { LAB_SHIFT = 12, LAB_SHIFT2 = 15, GAMMA_SHIFT = 3, LAB_CBRT_TAB_SIZE_B = 256 * 3 / 2 * (1 << GAMMA_SHIFT), GAMMA_TAB_SIZE = 1024 };
static uint16_t sRGBGammaTab_b[256];
static uint16_t LabCbrtTab_b[LAB_CBRT_TAB_SIZE_B];
static float sRGBInvGammaTab[GAMMA_TAB_SIZE * 4];
static int lab_coeffs_b[9];
static float lab2rgb_coeffs[9];
static int tabs_ready = 0;

#define CV_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static void spline_build(const float* f, int n, float* tab) {
    float cn = 0;
    tab[0] = tab[1] = 0.f;
    for (int i = 1; i < n - 1; i++) {
        float t = 3 * (f[i + 1] - 2 * f[i] + f[i - 1]);
        float l = 1 / (4 - tab[(i - 1) * 4]);
        tab[i * 4] = l; tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
    }
    for (int i = n - 1; i >= 0; i--) {
        float c = tab[i * 4 + 1] - tab[i * 4] * cn;
        float b = f[i + 1] - f[i] - (cn + c * 2) * 0.3333333333333333f;
        float d = (cn - c) * 0.3333333333333333f;
        tab[i * 4] = f[i]; tab[i * 4 + 1] = b; tab[i * 4 + 2] = c; tab[i * 4 + 3] = d;
        cn = c;
    }
}
static inline float spline_interp(float x, const float* tab, int n) {
    int ix = (int)floorf(x);
    ix = ix < 0 ? 0 : (ix > n - 1 ? n - 1 : ix);
    x -= ix;
    tab += ix * 4;
    return ((tab[3] * x + tab[2]) * x + tab[1]) * x + tab[0];
}

static void init_tabs(void) {
    if (tabs_ready) return;
    for (int i = 0; i < 256; i++) {
        float x = i * (1.f / 255.f);
        sRGBGammaTab_b[i] = sat_u16f(255.f * (1 << GAMMA_SHIFT) * (x <= 0.04045f ? x * (1.f / 12.92f) : (float)pow((double)(x + 0.055) * (1. / 1.055), 2.4)));
    }
    for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; i++) {
        float x = i * (1.f / (255.f * (1 << GAMMA_SHIFT)));
        LabCbrtTab_b[i] = sat_u16f((1 << LAB_SHIFT2) * (x < 0.008856f ? x * 7.787f + 0.13793103448275862f : cbrtf(x)));
    }
    static float g[GAMMA_TAB_SIZE + 1];
    for (int i = 0; i <= GAMMA_TAB_SIZE; i++) {
        float x = i * (1.f / GAMMA_TAB_SIZE);
        g[i] = x <= 0.0031308 ? x * 12.92f : (float)(1.055 * pow((double)x, 1. / 2.4) - 0.055);
    }
    spline_build(g, GAMMA_TAB_SIZE, sRGBInvGammaTab);
    /* sRGB -> XYZ (D65), rows scaled by 1/whitepoint; blueIdx = 0 for BGR input (coefficient columns swapped) */
    static const float s2x[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f, 0.072169f, 0.019334f, 0.119193f, 0.950227f};
    static const float wp[3] = {0.950456f, 1.f, 1.088754f};
    const float scale[3] = {(1 << LAB_SHIFT) / wp[0], (float)(1 << LAB_SHIFT), (1 << LAB_SHIFT) / wp[2]};
    for (int i = 0; i < 3; i++) {
        lab_coeffs_b[i * 3 + 2] = cv_round(s2x[i * 3] * scale[i]);       /* R coefficient sits at index blueIdx^2 = 2 */
        lab_coeffs_b[i * 3 + 1] = cv_round(s2x[i * 3 + 1] * scale[i]);
        lab_coeffs_b[i * 3 + 0] = cv_round(s2x[i * 3 + 2] * scale[i]);   /* B coefficient at index blueIdx = 0 */
    }
    static const float x2s[9] = {3.240479f, -1.53715f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
    for (int i = 0; i < 3; i++) {
        lab2rgb_coeffs[i + 2 * 3] = x2s[i] * wp[i];        /* row producing R -> output slot blueIdx^2 = 2 */
        lab2rgb_coeffs[i + 3] = x2s[i + 3] * wp[i];
        lab2rgb_coeffs[i + 0 * 3] = x2s[i + 6] * wp[i];    /* row producing B -> output slot blueIdx = 0 */
    }
    tabs_ready = 1;
}

void orc_cvt_tables(uint16_t* gamma256, uint16_t* cbrt_tab, float* inv_gamma_spline, int* coeffs9, float* lab2rgb9) {
    init_tabs();
    if (gamma256) memcpy(gamma256, sRGBGammaTab_b, sizeof sRGBGammaTab_b);
    if (cbrt_tab) memcpy(cbrt_tab, LabCbrtTab_b, sizeof LabCbrtTab_b);
    if (inv_gamma_spline) memcpy(inv_gamma_spline, sRGBInvGammaTab, sizeof sRGBInvGammaTab);
    if (coeffs9) memcpy(coeffs9, lab_coeffs_b, sizeof lab_coeffs_b);
    if (lab2rgb9) memcpy(lab2rgb9, lab2rgb_coeffs, sizeof lab2rgb_coeffs);
}
int orc_cvt_cbrt_tab_size(void) { return LAB_CBRT_TAB_SIZE_B; }

void orc_bgr2lab_u8(const uint8_t* src, size_t npix, uint8_t* dst) {
    init_tabs();
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    const int* C = lab_coeffs_b;
    for (size_t i = 0; i < npix; ++i) {
        int R = sRGBGammaTab_b[src[i * 3 + 0]], G = sRGBGammaTab_b[src[i * 3 + 1]], B = sRGBGammaTab_b[src[i * 3 + 2]];   /* "R" = first channel */
        int fX = LabCbrtTab_b[CV_DESCALE(R * C[0] + G * C[1] + B * C[2], LAB_SHIFT)];
        int fY = LabCbrtTab_b[CV_DESCALE(R * C[3] + G * C[4] + B * C[5], LAB_SHIFT)];
        int fZ = LabCbrtTab_b[CV_DESCALE(R * C[6] + G * C[7] + B * C[8], LAB_SHIFT)];
        int L = CV_DESCALE(Lscale * fY + Lshift, LAB_SHIFT2);
        int a = CV_DESCALE(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
        int b = CV_DESCALE(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
        dst[i * 3 + 0] = sat_u8(L); dst[i * 3 + 1] = sat_u8(a); dst[i * 3 + 2] = sat_u8(b);
    }
}

/* CV_Lab2BGR on 8U (ColorTransfer.cpp:1469) = Lab2RGB_b around the float Lab2RGB_f. Two forms of Lab2RGB_f exist in OpenCV's history; the sources of the
 * reference's pinned 2.4.10 cannot be read here (parity unpinned), but its artefacts can (tests/golden/demo_res_dark_stats.json):
 *   form 0 (DEFAULT): piecewise — CIE linear branch for L* <= 8 and f <= 6/29, linear RGB clipped to [0, 1]. The reference's demo results contain (0,0,0)
 *           and (0,2,0)-like pixels that only this form produces;
 *   form 1: plain cubes — fY = (L+16)*(1/116), fX = fY + a*0.002f, fZ = fY - b*0.005f, each cubed; linear RGB is NOT clipped: splineInterpolate clamps only
 *           the table index, so out-of-gamut values run along the first / last cubic, and saturate_cast<uchar>(v*255) does the clamping.
 * They coincide for in-gamut colours with L* > 8, fX > 6/29, fZ > 6/29. */
static int g_lab2bgr_form = 0;
void orc_set_lab2bgr_form(int form) { g_lab2bgr_form = form ? 1 : 0; }
int orc_get_lab2bgr_form(void) { return g_lab2bgr_form; }
void orc_lab2bgr_u8_form(const uint8_t* src, size_t npix, uint8_t* dst, int form) {
    init_tabs();
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    const float* C = lab2rgb_coeffs;
    for (size_t i = 0; i < npix; ++i) {
        float li = src[i * 3] * (100.f / 255.f), ai = (float)(src[i * 3 + 1] - 128), bi = (float)(src[i * 3 + 2] - 128);
        float x, y, z;
        if (form == 1) {
            float fy = (li + 16.f) * (1.f / 116.f);
            x = fy + ai * 0.002f; z = fy - bi * 0.005f;
            y = fy * fy * fy; x = x * x * x; z = z * z * z;
        } else {
            float fy;
            if (li <= lThresh) { y = li / 903.3f; fy = 7.787f * y + 16.0f / 116.0f; }
            else { fy = (li + 16.0f) / 116.0f; y = fy * fy * fy; }
            float fxz[2] = {ai / 500.0f + fy, fy - bi / 200.0f};
            for (int j = 0; j < 2; j++)
                if (fxz[j] <= fThresh) fxz[j] = (fxz[j] - 16.0f / 116.0f) / 7.787f;
                else fxz[j] = fxz[j] * fxz[j] * fxz[j];
            x = fxz[0]; z = fxz[1];
        }
        for (int k = 0; k < 3; ++k) {
            float v = C[k * 3] * x + C[k * 3 + 1] * y + C[k * 3 + 2] * z;
            if (form == 0) v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
            v = spline_interp(v * (float)GAMMA_TAB_SIZE, sRGBInvGammaTab, GAMMA_TAB_SIZE) * 255.f;
            v = v < -1.f ? -1.f : (v > 256.f ? 256.f : v);          /* keeps the int conversion defined; saturate_cast clamps anyway */
            dst[i * 3 + k] = sat_u8(cv_round(v));
        }
    }
}
void orc_lab2bgr_u8(const uint8_t* src, size_t npix, uint8_t* dst) { orc_lab2bgr_u8_form(src, npix, dst, g_lab2bgr_form); }

/* ---------------------------------------------------------------- resize (INTER_LINEAR semantics of cv::resize) */
static void linear_coeffs(int ssize, int dsize, int* ofs, float* alpha, int* xmax_out) {
    double scale = (double)ssize / dsize;
    int xmax = dsize;
    for (int dx = 0; dx < dsize; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= ssize) { if (dx < xmax) xmax = dx; if (sx >= ssize - 1) { fx = 0; sx = ssize - 1; } }
        ofs[dx] = sx; alpha[dx * 2] = 1.f - fx; alpha[dx * 2 + 1] = fx;
    }
    *xmax_out = xmax;
}

void orc_resize_u8c3(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    if (sh == dh && sw == dw) { memcpy(dst, src, (size_t)sh * sw * 3); return; }
    if (sw == dw * 2 && sh == dh * 2) {          /* cv::resize switches INTER_LINEAR -> INTER_AREA for exact 2x shrink */
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < 3; ++c) {
                    const uint8_t* p = src + ((size_t)(2 * y) * sw + 2 * x) * 3 + c;
                    dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((p[0] + p[3] + p[(size_t)sw * 3] + p[(size_t)sw * 3 + 3] + 2) >> 2);
                }
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw); int* yofs = (int*)malloc(sizeof(int) * dh);
    float* fa = (float*)malloc(sizeof(float) * 2 * dw); float* fb = (float*)malloc(sizeof(float) * 2 * dh);
    int xmax, ymax;
    linear_coeffs(sw, dw, xofs, fa, &xmax);
    linear_coeffs(sh, dh, yofs, fb, &ymax);
    int* rows = (int*)malloc(sizeof(int) * 2 * (size_t)dw * 3);
    for (int dy = 0; dy < dh; ++dy) {
        int sy = yofs[dy];
        int b0 = (short)cv_round(fb[dy * 2] * 2048.f), b1 = (short)cv_round(fb[dy * 2 + 1] * 2048.f);
        for (int k = 0; k < 2; ++k) {
            int syk = sy + k; if (syk > sh - 1) syk = sh - 1;                    /* vertical border replicate (weight is 0 there) */
            const uint8_t* S = src + (size_t)syk * sw * 3;
            int* D = rows + (size_t)k * dw * 3;
            for (int dx = 0; dx < dw; ++dx) {
                int sx = xofs[dx];
                int a0 = (short)cv_round(fa[dx * 2] * 2048.f), a1 = (short)cv_round(fa[dx * 2 + 1] * 2048.f);
                for (int c = 0; c < 3; ++c)
                    D[dx * 3 + c] = dx < xmax ? S[sx * 3 + c] * a0 + S[(sx + 1) * 3 + c] * a1 : S[sx * 3 + c] * 2048;
            }
        }
        const int* S0 = rows; const int* S1 = rows + (size_t)dw * 3;
        for (int x = 0; x < dw * 3; ++x)
            dst[(size_t)dy * dw * 3 + x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(fa); free(fb); free(rows);
}

void orc_resize_f64c3(const double* src, int sh, int sw, double* dst, int dh, int dw) {
    if (sh == dh && sw == dw) { memcpy(dst, src, sizeof(double) * (size_t)sh * sw * 3); return; }
    int* xofs = (int*)malloc(sizeof(int) * dw); int* yofs = (int*)malloc(sizeof(int) * dh);
    float* fa = (float*)malloc(sizeof(float) * 2 * dw); float* fb = (float*)malloc(sizeof(float) * 2 * dh);
    int xmax, ymax;
    linear_coeffs(sw, dw, xofs, fa, &xmax);
    linear_coeffs(sh, dh, yofs, fb, &ymax);
    double* rows = (double*)malloc(sizeof(double) * 2 * (size_t)dw * 3);
    for (int dy = 0; dy < dh; ++dy) {
        int sy = yofs[dy];
        for (int k = 0; k < 2; ++k) {
            int syk = sy + k; if (syk > sh - 1) syk = sh - 1;
            const double* S = src + (size_t)syk * sw * 3;
            double* D = rows + (size_t)k * dw * 3;
            for (int dx = 0; dx < dw; ++dx) {
                int sx = xofs[dx];
                for (int c = 0; c < 3; ++c)
                    D[dx * 3 + c] = dx < xmax ? S[sx * 3 + c] * fa[dx * 2] + S[(sx + 1) * 3 + c] * fa[dx * 2 + 1] : S[sx * 3 + c] * 1.0;
            }
        }
        const double* S0 = rows; const double* S1 = rows + (size_t)dw * 3;
        double b0 = fb[dy * 2], b1 = fb[dy * 2 + 1];
        for (int x = 0; x < dw * 3; ++x) dst[(size_t)dy * dw * 3 + x] = S0[x] * b0 + S1[x] * b1;
    }
    free(xofs); free(yofs); free(fa); free(fb); free(rows);
}

/* Mat::convertTo(CV_64F, 1/255): double(v) * (1/255) */
void orc_u8_to_f64_scaled(const uint8_t* src, size_t n, double* dst) {
    const double s = 1.0 / 255.0;
    for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i] * s;
}
