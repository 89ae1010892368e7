/* oracle/orc_wls_mg.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * S2 (ColorTransfer.cpp:951-1125): canonical-order restatement of the product's multigrid-preconditioned CG for the WLS
 * system (diag(r) + L_g) x = r x0 with 6 right-hand sides. The reference solves this system exactly (MKL PARDISO);
 * this solver converges to 1e-8 relative residual and is cross-checked against the exact solve (banded Cholesky / PARDISO
 * fixture) in tests/test_oracle_color.py. It exists in this exact arithmetic order so that the 8-bit result of the GPU
 * path can be compared bit-for-bit (see orc_color_canon.c for why that matters: S1 downstream is chaotic).
 * Hierarchy: 2x2 aggregation (coarse data term = sum of the 4 fine ones, coarse edge = sum of the crossing fine edges), built in
 * fp64; V(2,2) cycle, Chebyshev-weighted Jacobi (omega 0.5808 then 2.6437), 60 Jacobi sweeps (0.8) on the coarsest grid — the cycle runs in fp32 on rounded copies
 * of the level operators (it is only the preconditioner; the CG recurrences, the operator and every dot product stay fp64);
 * two-stage 256-wide tree reductions. */
#include "orc_common.h"
#include <stdio.h>

void orc_wls_system(const double* lab, int H, int W, double lamda, double alpha, const double* roughness, double* diag, double* wx, double* wy);

#define NQ 6
/* two Chebyshev-weighted Jacobi sweeps per leg (k_wls_mg.hip: NCT_MG_W1 / NCT_MG_W2); fdinv = (float)(W1 / diag), the second sweep and the
 * coarsest grid scale it in fp32 exactly as the kernels do */
typedef struct { int H, W, n; double *r, *wx, *wy, *diag; float *fdiag, *fdinv, *fwx, *fwy, *b, *x, *x2; } lvl_t;   /* fdinv = (float)(omega / diag) */

static void tree256(double* s) { for (int off = 128; off >= 1; off >>= 1) for (int t = 0; t < off; ++t) s[t] += s[t + off]; }
static void canon_sum(const double* v, int n, int nq, double* out) {
    const int nb = (n + 255) / 256;
    double* partial = (double*)malloc(sizeof(double) * (size_t)nb * nq);
    double s[256];
    for (int b = 0; b < nb; ++b)
        for (int q = 0; q < nq; ++q) {
            for (int t = 0; t < 256; ++t) { int i = b * 256 + t; s[t] = i < n ? v[(size_t)i * nq + q] : 0.0; }
            tree256(s);
            partial[(size_t)b * nq + q] = s[0];
        }
    for (int q = 0; q < nq; ++q) {
        for (int t = 0; t < 256; ++t) { double acc = 0.0; for (int b = t; b < nb; b += 256) acc += partial[(size_t)b * nq + q]; s[t] = acc; }
        tree256(s);
        out[q] = s[0];
    }
    free(partial);
}

/* y = M v at pixel i; VAL(j,q) is an expression giving v_j[q] */
#define LVL_OP(L, i, VAL, y) do { \
    const int W_ = (L)->W, H_ = (L)->H; const int r_ = (i) / W_, c_ = (i) - r_ * W_; const double d_ = (L)->diag[i]; \
    for (int q = 0; q < NQ; ++q) (y)[q] = d_ * VAL((i), q); \
    if (c_ + 1 < W_) { const double w_ = (L)->wx[i]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) + 1, q); } \
    if (c_ > 0) { const double w_ = (L)->wx[(i) - 1]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) - 1, q); } \
    if (r_ + 1 < H_) { const double w_ = (L)->wy[i]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) + W_, q); } \
    if (r_ > 0) { const double w_ = (L)->wy[(i) - W_]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) - W_, q); } } while (0)

/* fp32 stencil of a level; VAL(j,q) is a float expression giving v_j[q]; every operation is a float operation */
#define LVL_OPF(L, i, VAL, y) do { \
    const int W_ = (L)->W, H_ = (L)->H; const int r_ = (i) / W_, c_ = (i) - r_ * W_; const float d_ = (L)->fdiag[i]; \
    for (int q = 0; q < NQ; ++q) (y)[q] = d_ * VAL((i), q); \
    if (c_ + 1 < W_) { const float w_ = (L)->fwx[i]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) + 1, q); } \
    if (c_ > 0) { const float w_ = (L)->fwx[(i) - 1]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) - 1, q); } \
    if (r_ + 1 < H_) { const float w_ = (L)->fwy[i]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) + W_, q); } \
    if (r_ > 0) { const float w_ = (L)->fwy[(i) - W_]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) - W_, q); } } while (0)

/* Smoother: NS damped-Jacobi sweeps per leg with Chebyshev weights (k_wls_mg.hip: MG_NS, MG_W[]); fdinv = (float)(W[0] / diag), sweep k scales it in fp32 by
 * (float)(W[k] / W[0]) exactly as the kernels do. orc_set_mg_smoother selects one of the shipped / experimental sets (design experiments; the default is the product's). */
static int MG_NS = 3;
static double MG_W[8] = {0.5346, 0.9677, 5.0974};
static float mg_rk(int k) { return (float)(MG_W[k] / MG_W[0]); }
void orc_set_mg_smoother(int ns) {
    static const double w2[2] = {0.5808, 2.6437}, w3[3] = {0.5346, 0.9677, 5.0974}, w4[4] = {0.5193, 0.7153, 1.5340, 8.0502};
    const double* w = ns == 2 ? w2 : (ns == 4 ? w4 : w3);
    MG_NS = ns == 2 || ns == 4 ? ns : 3;
    for (int k = 0; k < MG_NS; ++k) MG_W[k] = w[k];
}
int orc_get_mg_smoother(void) { return MG_NS; }

/* one Jacobi sweep on level L: out = in + (rhs - M in) * (fdinv * rk); in == NULL means "from zero": out = rhs * fdinv (rk = 1). rhs: fp64 residual (rounded on load) at level 0 */
static void mg_sweep(const lvl_t* L, const double* r0, const float* in, float* out, float rk) {
#define BVS(j, q) (r0 ? (float)r0[(size_t)(j) * NQ + (q)] : L->b[(size_t)(j) * NQ + (q)])
    if (!in) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < L->n; ++i) for (int q = 0; q < NQ; ++q) out[(size_t)i * NQ + q] = BVS(i, q) * L->fdinv[i];
        return;
    }
#define INV(j, q) (in[(size_t)(j) * NQ + (q)])
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L->n; ++i) {
        float y[NQ]; LVL_OPF(L, i, INV, y);
        const float d = rk == 1.0f ? L->fdinv[i] : L->fdinv[i] * rk;
        for (int q = 0; q < NQ; ++q) { const float t = BVS(i, q) - y[q]; const float u = t * d; out[(size_t)i * NQ + q] = in[(size_t)i * NQ + q] + u; }
    }
#undef INV
}

/* z = lv[0].x (fp32) for the fp64 residual r0 (rounded to fp32 on load) */
static void vcycle(lvl_t* lv, int nl, const double* r0) {
    for (int l = 0; l < nl - 1; ++l) {
        lvl_t* L = &lv[l]; lvl_t* C = &lv[l + 1];
        const double* rr = l == 0 ? r0 : NULL;
        /* pre-smoothing from zero: NS sweeps, the result in L->x */
        float* cur = (MG_NS & 1) ? L->x : L->x2; float* oth = (MG_NS & 1) ? L->x2 : L->x;
        mg_sweep(L, rr, NULL, cur, 1.0f);
        for (int k = 1; k < MG_NS; ++k) { mg_sweep(L, rr, cur, oth, mg_rk(k)); float* t = cur; cur = oth; oth = t; }
        /* cur == L->x */
#define BV(j, q) (l == 0 ? (float)r0[(size_t)(j) * NQ + (q)] : L->b[(size_t)(j) * NQ + (q)])
#define XV(j, q) (L->x[(size_t)(j) * NQ + (q)])
#pragma omp parallel for schedule(static)
        for (int I = 0; I < C->n; ++I) {
            const int Y = I / C->W, X = I - Y * C->W;
            float acc[NQ] = {0, 0, 0, 0, 0, 0};
            for (int t = 0; t < 4; ++t) {
                const int y = 2 * Y + (t >> 1), xx = 2 * X + (t & 1);
                if (y < L->H && xx < L->W) {
                    const int i = y * L->W + xx;
                    float yv[NQ]; LVL_OPF(L, i, XV, yv);
                    for (int q = 0; q < NQ; ++q) { const float t2 = BV(i, q) - yv[q]; acc[q] += t2; }
                }
            }
            for (int q = 0; q < NQ; ++q) C->b[(size_t)I * NQ + q] = acc[q];
        }
#undef XV
#undef BV
    }
    {   /* coarsest: 60 Jacobi sweeps from zero */
        lvl_t* L = &lv[nl - 1];
        float* cur = L->x; float* nxt = L->x2;
        memset(cur, 0, sizeof(float) * (size_t)L->n * NQ);
        const float r0c = (float)(0.8 / MG_W[0]);
        for (int s = 0; s < 60; ++s) {
#define CV(j, q) (cur[(size_t)(j) * NQ + (q)])
            for (int i = 0; i < L->n; ++i) {
                float y[NQ]; LVL_OPF(L, i, CV, y);
                const float d = L->fdinv[i] * r0c;
                for (int q = 0; q < NQ; ++q) { const float t = L->b[(size_t)i * NQ + q] - y[q]; const float u = t * d; nxt[(size_t)i * NQ + q] = cur[(size_t)i * NQ + q] + u; }
            }
#undef CV
            float* t = cur; cur = nxt; nxt = t;
        }
        /* 60 is even: the result is back in L->x */
    }
    for (int l = nl - 2; l >= 0; --l) {
        lvl_t* L = &lv[l]; lvl_t* C = &lv[l + 1];
        const int Wc = C->W;
        const double* rr = l == 0 ? r0 : NULL;
        /* xe = x + e_coarse(parent), then NS sweeps; the result ends in L->x */
        float* cur = (MG_NS & 1) ? L->x2 : L->x; float* oth = (MG_NS & 1) ? L->x : L->x2;
        if (cur != L->x) {
#pragma omp parallel for schedule(static)
            for (int j = 0; j < L->n; ++j) for (int q = 0; q < NQ; ++q) cur[(size_t)j * NQ + q] = L->x[(size_t)j * NQ + q] + C->x[(size_t)(((j / L->W) >> 1) * Wc + ((j % L->W) >> 1)) * NQ + q];
        } else {
#pragma omp parallel for schedule(static)
            for (int j = 0; j < L->n; ++j) for (int q = 0; q < NQ; ++q) L->x[(size_t)j * NQ + q] = L->x[(size_t)j * NQ + q] + C->x[(size_t)(((j / L->W) >> 1) * Wc + ((j % L->W) >> 1)) * NQ + q];
        }
        for (int k = 0; k < MG_NS; ++k) { mg_sweep(L, rr, cur, oth, k == 0 ? 1.0f : mg_rk(k)); float* t = cur; cur = oth; oth = t; }
    }
}
#undef BVS

/* diagnostic: iteration counts (max over the right-hand sides) of the solves since the last reset */
static int g_wls_log[64], g_wls_log_n = 0;
int orc_wls_log(int* out, int reset) { const int n = g_wls_log_n; if (out) memcpy(out, g_wls_log, sizeof(int) * (size_t)n); if (reset) g_wls_log_n = 0; return n; }

/* a,b: full-res [N][3] in (x0) / out. iters_out[6] nullable. Returns max iterations, or -1 if not converged. */
int orc_wls_solve_mg(double* a, double* b, const double* lab, int H, int W, double lamda, double alpha, const double* roughness, double rtol, int* iters_out) {
    lvl_t lv[16]; int nl = 0;
    { int h = H, w = W;
      for (;;) {
          lvl_t* L = &lv[nl]; L->H = h; L->W = w; L->n = h * w;
          L->r = (double*)malloc(sizeof(double) * L->n); L->wx = (double*)malloc(sizeof(double) * L->n); L->wy = (double*)malloc(sizeof(double) * L->n);
          L->diag = (double*)malloc(sizeof(double) * L->n);
          L->fdiag = (float*)malloc(sizeof(float) * L->n); L->fdinv = (float*)malloc(sizeof(float) * L->n); L->fwx = (float*)malloc(sizeof(float) * L->n); L->fwy = (float*)malloc(sizeof(float) * L->n);
          L->b = (float*)malloc(sizeof(float) * (size_t)L->n * NQ); L->x = (float*)malloc(sizeof(float) * (size_t)L->n * NQ); L->x2 = (float*)malloc(sizeof(float) * (size_t)L->n * NQ);
          ++nl;
          if (L->n <= 64 || (h <= 8 && w <= 8) || nl >= 16) break;
          h = (h + 1) / 2; w = (w + 1) / 2;
      } }
    { double* dtmp = (double*)malloc(sizeof(double) * lv[0].n);
      orc_wls_system(lab, H, W, lamda, alpha, roughness, dtmp, lv[0].wx, lv[0].wy);
      memcpy(lv[0].r, roughness, sizeof(double) * lv[0].n); free(dtmp); }
    for (int l = 0; l < nl; ++l) {
        lvl_t* L = &lv[l];
        if (l > 0) {
            lvl_t* F = &lv[l - 1];
            for (int I = 0; I < L->n; ++I) {
                const int Y = I / L->W, X = I - Y * L->W, y0 = 2 * Y, x0 = 2 * X;
                const int x1ok = x0 + 1 < F->W, y1ok = y0 + 1 < F->H;
                double rs = F->r[y0 * F->W + x0];
                if (x1ok) rs += F->r[y0 * F->W + x0 + 1];
                if (y1ok) rs += F->r[(y0 + 1) * F->W + x0];
                if (x1ok && y1ok) rs += F->r[(y0 + 1) * F->W + x0 + 1];
                double ex = 0.0, ey = 0.0;
                if (x0 + 2 < F->W) { ex = F->wx[y0 * F->W + x0 + 1]; if (y1ok) ex += F->wx[(y0 + 1) * F->W + x0 + 1]; }
                if (y0 + 2 < F->H) { ey = F->wy[(y0 + 1) * F->W + x0]; if (x1ok) ey += F->wy[(y0 + 1) * F->W + x0 + 1]; }
                L->r[I] = rs; L->wx[I] = ex; L->wy[I] = ey;
            }
        }
        for (int i = 0; i < L->n; ++i) {
            const int y = i / L->W, x = i - y * L->W;
            double a00 = 0.0;
            a00 += L->r[i];
            if (x + 1 < L->W) a00 += L->wx[i];
            if (x > 0) a00 += L->wx[i - 1];
            if (y + 1 < L->H) a00 += L->wy[i];
            if (y > 0) a00 += L->wy[i - L->W];
            L->diag[i] = a00;
            L->fdiag[i] = (float)a00; L->fdinv[i] = (float)(MG_W[0] / a00); L->fwx[i] = (float)L->wx[i]; L->fwy[i] = (float)L->wy[i];
        }
    }
    lvl_t* F = &lv[0];
    const int n = F->n;
    double* x6 = (double*)malloc(sizeof(double) * (size_t)n * NQ); double* r = (double*)malloc(sizeof(double) * (size_t)n * NQ);
    double* p = (double*)calloc((size_t)n * NQ, sizeof(double)); double* sv = (double*)calloc((size_t)n * NQ, sizeof(double));
    double* w = (double*)malloc(sizeof(double) * (size_t)n * NQ);
    double* acc = (double*)malloc(sizeof(double) * (size_t)n * 18);
    const double rtol2 = rtol * rtol;
    double gam_old[6] = {0}, alp_old[6] = {0}, bb[6], s[18];
    int active[6], iters[6] = {0, 0, 0, 0, 0, 0};
#define X0(j, q) ((q) < 3 ? a[(size_t)(j) * 3 + (q)] : b[(size_t)(j) * 3 + (q) - 3])
    for (int i = 0; i < n; ++i) {
        double y[NQ]; LVL_OP(F, i, X0, y);
        const double rg = F->r[i];
        for (int q = 0; q < NQ; ++q) {
            const double x0 = X0(i, q), bq = rg * x0, rv = bq - y[q];
            x6[(size_t)i * NQ + q] = x0; r[(size_t)i * NQ + q] = rv;
            acc[(size_t)i * 12 + q] = rv * rv; acc[(size_t)i * 12 + 6 + q] = bq * bq;
        }
    }
#undef X0
    canon_sum(acc, n, 12, s);
    int any = 0;
    for (int q = 0; q < 6; ++q) { bb[q] = s[6 + q]; active[q] = s[q] > rtol2 * s[6 + q]; any |= active[q]; }
    /* single-reduction (Chronopoulos-Gear) PCG, operation for operation as k_cg_apply / k_cg_fin / k_cg_update */
    int it = 0;
    const int maxit = 5000;
    while (any && it < maxit) {
        const int first = it == 0;
        vcycle(lv, nl, r);                                             /* u = F->x (fp32), widened exactly below */
#define UV(j, q) ((double)F->x[(size_t)(j) * NQ + (q)])
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            double y[NQ]; LVL_OP(F, i, UV, y);
            for (int q = 0; q < NQ; ++q) {
                const double u = UV(i, q), rv = r[(size_t)i * NQ + q];
                w[(size_t)i * NQ + q] = y[q];
                acc[(size_t)i * 18 + q] = rv * u; acc[(size_t)i * 18 + 6 + q] = y[q] * u; acc[(size_t)i * 18 + 12 + q] = rv * rv;
            }
        }
        canon_sum(acc, n, 18, s);                                      /* gamma [0,6), delta [6,12), rho [12,18) */
        double al[6], be[6]; int act[6];
        any = 0;
        for (int q = 0; q < 6; ++q) {
            const double gam = s[q], del = s[6 + q], rho = s[12 + q];
            act[q] = active[q] && rho > rtol2 * bb[q];
            be[q] = first ? 0.0 : gam / gam_old[q];
            al[q] = first ? gam / del : gam / (del - be[q] * gam / alp_old[q]);
        }
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < NQ; ++q) {
                if (!act[q]) continue;
                const size_t j = (size_t)i * NQ + q;
                const double zv = UV(i, q), wv = w[j];
                const double pn = first ? zv : zv + be[q] * p[j];
                const double sn = first ? wv : wv + be[q] * sv[j];
                p[j] = pn; sv[j] = sn;
                x6[j] += al[q] * pn;
                r[j] -= al[q] * sn;
            }
#undef UV
        for (int q = 0; q < 6; ++q) { if (act[q]) { gam_old[q] = s[q]; alp_old[q] = al[q]; iters[q]++; } active[q] = act[q]; any |= act[q]; }
        ++it;
    }
    for (int i = 0; i < n; ++i) for (int q = 0; q < NQ; ++q) { if (q < 3) a[(size_t)i * 3 + q] = x6[(size_t)i * NQ + q]; else b[(size_t)i * 3 + q - 3] = x6[(size_t)i * NQ + q]; }
    if (iters_out) memcpy(iters_out, iters, sizeof iters);
    int mx = 0; for (int q = 0; q < 6; ++q) if (iters[q] > mx) mx = iters[q];
    if (g_wls_log_n < 64) g_wls_log[g_wls_log_n++] = mx;
    for (int l = 0; l < nl; ++l) { free(lv[l].r); free(lv[l].wx); free(lv[l].wy); free(lv[l].diag); free(lv[l].fdiag); free(lv[l].fdinv); free(lv[l].fwx); free(lv[l].fwy); free(lv[l].b); free(lv[l].x); free(lv[l].x2); }
    free(x6); free(r); free(p); free(sv); free(w); free(acc);
    return any ? -1 : mx;
}
