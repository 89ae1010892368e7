/* oracle/orc_wls_mg.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * S2 (ColorTransfer.cpp:951-1125): canonical-order restatement of the product's multigrid-preconditioned CG for the WLS
 * system (diag(r) + L_g) x = r x0 with 6 right-hand sides. The reference solves this system exactly (MKL PARDISO,
 * SparseSolver_CPU.cpp:104-286); this solver converges to 1e-7 relative residual and is cross-checked against the exact solve
 * (banded Cholesky / PARDISO fixture) in tests/test_oracle_color.py. It exists in this exact arithmetic order so that the
 * 8-bit result of the GPU path can be compared bit-for-bit (see orc_color_canon.c for why that matters: S1 downstream is chaotic).
 *
 * Hierarchy (round 4; rounds 1-3: 2x2 aggregation with piecewise-constant transfer): VERTEX-CENTRED coarsening — coarse point
 * (Y, X) IS fine point (2Y, 2X) — with OPERATOR-DEPENDENT interpolation (the black-box multigrid of Alcouffe/Brandt/Dendy/Painter):
 *   fine points on a coarse grid line interpolate from their two coarse neighbours with weights from the stencil collapsed across
 *   the line (w_W / (d - w_N - w_S) ...), fine points in the middle of a coarse cell solve their own equation for the 8 neighbours;
 * restriction = transpose, coarse operators = Galerkin products P^T A P: symmetric 9-point stencils, stored as the diagonal and the
 * four forward couplings (+x, +y, +x+y, -x+y), all as w = -A(i,j) so that level 0 keeps the 5-point (diag, wx, wy) form.
 * Built in fp64; the V(NS,NS) cycle runs in fp32 on rounded copies (it is only the preconditioner; the CG recurrences, the
 * operator and every dot product stay fp64). Smoother: NS Chebyshev-weighted Jacobi sweeps per leg on the "safe" diagonal
 * dt = max(d, (d + sum |w|) / 2) (Gershgorin: lambda_max(dt^-1 A) <= 2 also where a Galerkin stencil has couplings of the wrong
 * sign; dt = d for M-matrix rows, i.e. everywhere on level 0); 60 Jacobi sweeps (0.8) on the coarsest grid; two-stage 256-wide
 * tree reductions. */
#include "orc_common.h"
#include <stdio.h>

void orc_wls_system(const double* lab, int H, int W, double lamda, double alpha, const double* roughness, double* diag, double* wx, double* wy);

#define NQ 6
/* w-notation: (A v)_i = d_i v_i - sum_k w_ik v_k ; forward couplings wE = w(i, i+1), wS = w(i, i+W), wSE = w(i, i+W+1), wSW = w(i, i+W-1),
 * zero where the neighbour does not exist. Level 0: wSE = wSW = NULL (5-point). pa / pb: interpolation weights of the transfer to the next
 * coarser level — (even y, odd x): to the W / E coarse point; (odd y, even x): to the N / S one; (odd, odd): pa = 1 / d. */
typedef struct { int H, W, n, nine; double *d, *wE, *wS, *wSE, *wSW, *pa, *pb;
                 float *fd, *fdinv, *fE, *fS, *fSE, *fSW, *fpst, *b, *x, *x2;
                 float *lxm, *lxp, *lym, *lyp; } lvl_t;   /* fdinv = (float)(omega_0 / dt); fpst: the columns of P towards the next coarser level as
                                                                                           3x3 blocks, [n_coarse][9], (float) of mg_pstencil's values: THE transfer weights of the cycle */

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

/* the 8 couplings of pixel (r, c) in the order E, W, S, N, SE, SW, NE, NW (0 where the neighbour does not exist); fp64 and fp32 forms */
#define COUP8(TYPE, L, E_, S_, SE_, SW_, r, c, w) do { \
    const int W_ = (L)->W, H_ = (L)->H, i_ = (r) * W_ + (c); const int xr = (c) + 1 < W_, xl = (c) > 0, yd = (r) + 1 < H_, yu = (r) > 0; \
    (w)[0] = xr ? (L)->E_[i_] : (TYPE)0; (w)[1] = xl ? (L)->E_[i_ - 1] : (TYPE)0; (w)[2] = yd ? (L)->S_[i_] : (TYPE)0; (w)[3] = yu ? (L)->S_[i_ - W_] : (TYPE)0; \
    if ((L)->nine) { (w)[4] = (xr && yd) ? (L)->SE_[i_] : (TYPE)0; (w)[5] = (xl && yd) ? (L)->SW_[i_] : (TYPE)0; \
                     (w)[6] = (xr && yu) ? (L)->SW_[i_ - W_ + 1] : (TYPE)0; (w)[7] = (xl && yu) ? (L)->SE_[i_ - W_ - 1] : (TYPE)0; } \
    else { (w)[4] = (w)[5] = (w)[6] = (w)[7] = (TYPE)0; } } while (0)
static const int OFFY[8] = {0, 0, 1, -1, 1, 1, -1, -1}, OFFX[8] = {1, -1, 0, 0, 1, -1, 1, -1};
static int nbr_ok(const lvl_t* L, int r, int c, int k) { const int y = r + OFFY[k], x = c + OFFX[k]; return y >= 0 && y < L->H && x >= 0 && x < L->W && (k < 4 || L->nine); }

/* fp64 fine-level operator (5-point), neighbour order +x, -x, +y, -y */
#define LVL_OP(L, i, VAL, y) do { \
    const int W_ = (L)->W, H_ = (L)->H; const int r_ = (i) / W_, c_ = (i) - r_ * W_; const double d_ = (L)->d[i]; \
    for (int q = 0; q < NQ; ++q) (y)[q] = d_ * VAL((i), q); \
    if (c_ + 1 < W_) { const double w_ = (L)->wE[i]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) + 1, q); } \
    if (c_ > 0) { const double w_ = (L)->wE[(i) - 1]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) - 1, q); } \
    if (r_ + 1 < H_) { const double w_ = (L)->wS[i]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) + W_, q); } \
    if (r_ > 0) { const double w_ = (L)->wS[(i) - W_]; for (int q = 0; q < NQ; ++q) (y)[q] -= w_ * VAL((i) - W_, q); } } while (0)

/* fp32 stencil of a level: y = d v_i - sum_k w_k v_k over the existing neighbours in the order E, W, S, N, SE, SW, NE, NW; every operation is a float operation */
static void opf(const lvl_t* L, int i, const float* v, float* y) {
    const int W = L->W, r = i / W, c = i - r * W;
    float w[8]; COUP8(float, L, fE, fS, fSE, fSW, r, c, w);
    const float d = L->fd[i];
    for (int q = 0; q < NQ; ++q) y[q] = d * v[(size_t)i * NQ + q];
    for (int k = 0; k < 8; ++k)
        if (nbr_ok(L, r, c, k)) { const int j = i + OFFY[k] * W + OFFX[k]; for (int q = 0; q < NQ; ++q) y[q] -= w[k] * v[(size_t)j * NQ + q]; }
}

/* Smoother: NS damped-Jacobi sweeps per leg with Chebyshev weights (k_wls_mg.hip: MG_NS, MG_W[]); fdinv = (float)(W[0] / dt), sweep k scales it in fp32 by
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
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L->n; ++i) {
        float y[NQ]; opf(L, i, in, y);
        const float d = rk == 1.0f ? L->fdinv[i] : L->fdinv[i] * rk;
        for (int q = 0; q < NQ; ++q) { const float t = BVS(i, q) - y[q]; const float u = t * d; out[(size_t)i * NQ + q] = in[(size_t)i * NQ + q] + u; }
    }
}

/* restriction R = P^T: coarse point I at fine point f = (2Y, 2X):  b_c(I) = sum over the 3x3 block around f, row-major, of fpst[I][k] * res(f + off_k)  (points outside the grid skipped;
 * the centre weight is 1) */
static void mg_restrict(const lvl_t* L, lvl_t* C, const float* res) {
    const int W = L->W, H = L->H;
#pragma omp parallel for schedule(static)
    for (int I = 0; I < C->n; ++I) {
        const int Y = I / C->W, X = I - Y * C->W, r = 2 * Y, c = 2 * X;
        const float* ps = L->fpst + (size_t)I * 9;
        float acc[NQ] = {0, 0, 0, 0, 0, 0};
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int y = r + dy, x = c + dx;
                if (y < 0 || y >= H || x < 0 || x >= W) continue;
                const float p = ps[(dy + 1) * 3 + dx + 1];
                for (int q = 0; q < NQ; ++q) acc[q] += p * res[(size_t)(y * W + x) * NQ + q];
            }
        for (int q = 0; q < NQ; ++q) C->b[(size_t)I * NQ + q] = acc[q];
    }
}
/* prolongation e = P ec: a fine point combines its existing coarse parents in the order NW, NE, SW, SE (a point on a coarse grid line has two of them: W, E or N, S; a coarse
 * point copies): e = ((p0 e0 + p1 e1) + p2 e2) + p3 e3, weight of parent J for fine point i = fpst[J][i - 2J] */
static void mg_prolong(const lvl_t* L, const lvl_t* C, const float* ec, float* e) {
    const int W = L->W, Wc = C->W, Hc = C->H;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L->n; ++i) {
        const int r = i / W, c = i - r * W;
        const int Y0 = r >> 1, X0 = c >> 1, ny = (r & 1) ? 2 : 1, nx = (c & 1) ? 2 : 1;        /* parents: rows Y0 .. Y0 + ny - 1, columns X0 .. X0 + nx - 1 */
        float acc[NQ]; int first = 1;
        for (int jy = 0; jy < ny; ++jy)
            for (int jx = 0; jx < nx; ++jx) {
                const int Y = Y0 + jy, X = X0 + jx;
                if (Y >= Hc || X >= Wc) continue;
                const int J = Y * Wc + X;
                if (ny == 1 && nx == 1) { for (int q = 0; q < NQ; ++q) acc[q] = ec[(size_t)J * NQ + q]; first = 0; continue; }
                const float p = L->fpst[(size_t)J * 9 + (r - 2 * Y + 1) * 3 + (c - 2 * X + 1)];
                for (int q = 0; q < NQ; ++q) { const float t = p * ec[(size_t)J * NQ + q]; acc[q] = first ? t : acc[q] + t; }
                first = 0;
            }
        for (int q = 0; q < NQ; ++q) e[(size_t)i * NQ + q] = acc[q];
    }
}

/* ---- Round 5: a BLOCK STEP on the FINEST level, first thing of the pre-smoother and last thing of the post-smoother (orc_set_mg_lines(0) / the product's NCT_S2_LINES=0
 * select the cycle without it: rounds 1-5a, kept for comparison). The grid is cut into fixed blocks of LINE_BX x LINE_BY pixels (aligned at 0); inside a block, for a residual r:
 *     pre :  e1 = Lx^-1 r,  e2 = Ly^-1 (Sy e1),   x += LINE_OM (e1 + e2)        post (the adjoint):  e1 = Ly^-1 r,  e2 = Lx^-1 (Sx e1),  x += LINE_OM (e1 + e2)
 * Lx / Ly = the tridiagonal matrices of the level's diagonal and the x / y couplings INSIDE the block, Sy / Sx = the y / x couplings inside the block (Sy e1 = r - A_block e1
 * in exact arithmetic: an alternating-direction block solve). Point Jacobi cannot smooth along the strong couplings of a photograph's flat runs (12-37 % of the neighbour
 * pairs carry the largest weight, next to edges 10^4 weaker); line solves can (PCG iterations -25 % on the synthetic pair, -40 ... -50 % on photographs).
 * The Thomas factors come from mg_lines_setup in fp64 and are rounded once: lm[i] = w(i-1, i) / p(i-1) (0 at the start of a line), lp[i] = 1 / p(i), p(i) = d(i) - w(i-1, i) lm[i];
 * the solves are fp32:   forward y(0) = r(0), y(i) = r(i) + lm(i) y(i-1);   backward e(last) = y(last) lp(last), e(i) = (y(i) + w(i, i+1) e(i+1)) lp(i). */
#define LINE_BX 32
#define LINE_BY 16
#define LINE_OM 0.9f
static int MG_LINES = 1;
void orc_set_mg_lines(int on) { MG_LINES = on ? 1 : 0; }
int orc_get_mg_lines(void) { return MG_LINES; }
static void mg_lines_setup(lvl_t* L) {
    const int W = L->W, H = L->H; const size_t n = (size_t)L->n;
    L->lxm = (float*)malloc(sizeof(float) * n); L->lxp = (float*)malloc(sizeof(float) * n); L->lym = (float*)malloc(sizeof(float) * n); L->lyp = (float*)malloc(sizeof(float) * n);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < H; ++r)
        for (int c0 = 0; c0 < W; c0 += LINE_BX) {
            double p = 0.0;
            for (int c = c0; c < W && c < c0 + LINE_BX; ++c) {
                const int i = r * W + c;
                if (c == c0) { p = L->d[i]; L->lxm[i] = 0.0f; }
                else { const double w = L->wE[i - 1], m = w / p; p = L->d[i] - w * m; L->lxm[i] = (float)m; }
                L->lxp[i] = (float)(1.0 / p);
            }
        }
#pragma omp parallel for schedule(static)
    for (int c = 0; c < W; ++c)
        for (int r0 = 0; r0 < H; r0 += LINE_BY) {
            double p = 0.0;
            for (int r = r0; r < H && r < r0 + LINE_BY; ++r) {
                const int i = r * W + c;
                if (r == r0) { p = L->d[i]; L->lym[i] = 0.0f; }
                else { const double w = L->wS[i - W], m = w / p; p = L->d[i] - w * m; L->lym[i] = (float)m; }
                L->lyp[i] = (float)(1.0 / p);
            }
        }
}
/* out = in + LINE_OM * B (rhs - M in)   (in == NULL: from zero, out = LINE_OM * B rhs);  pre: x lines first, else y lines first. out must not alias in. */
static void mg_block_step(const lvl_t* L, const double* r0, const float* in, float* out, int pre) {
    const int W = L->W, H = L->H, nbx = (W + LINE_BX - 1) / LINE_BX, nby = (H + LINE_BY - 1) / LINE_BY;
#pragma omp parallel for schedule(dynamic, 4)
    for (int blk = 0; blk < nbx * nby; ++blk) {
        const int x0 = (blk % nbx) * LINE_BX, y0 = (blk / nbx) * LINE_BY;
        const int bw = x0 + LINE_BX <= W ? LINE_BX : W - x0, bh = y0 + LINE_BY <= H ? LINE_BY : H - y0;
        float res[LINE_BY][LINE_BX], e1[LINE_BY][LINE_BX], e2[LINE_BY][LINE_BX];
        for (int q = 0; q < NQ; ++q) {
            for (int y = 0; y < bh; ++y)
                for (int x = 0; x < bw; ++x) {
                    const int i = (y0 + y) * W + x0 + x;
                    const float bq = r0 ? (float)r0[(size_t)i * NQ + q] : L->b[(size_t)i * NQ + q];
                    if (!in) { res[y][x] = bq; continue; }
                    /* this right-hand side's fp32 stencil in opf's order (E, W, S, N; level 0 is 5-point) */
                    float yv = L->fd[i] * in[(size_t)i * NQ + q];
                    if (x0 + x + 1 < W) yv -= L->fE[i] * in[(size_t)(i + 1) * NQ + q];
                    if (x0 + x > 0) yv -= L->fE[i - 1] * in[(size_t)(i - 1) * NQ + q];
                    if (y0 + y + 1 < H) yv -= L->fS[i] * in[(size_t)(i + W) * NQ + q];
                    if (y0 + y > 0) yv -= L->fS[i - W] * in[(size_t)(i - W) * NQ + q];
                    res[y][x] = bq - yv;
                }
            for (int stage = 0; stage < 2; ++stage) {
                const int xl = (stage == 0) == (pre != 0);                 /* this stage solves x lines */
                float (*rr)[LINE_BX] = stage == 0 ? res : e2;              /* stage 1's right-hand side is built in e2 and solved in place */
                float (*ee)[LINE_BX] = stage == 0 ? e1 : e2;
                if (stage == 1)
                    for (int y = 0; y < bh; ++y)
                        for (int x = 0; x < bw; ++x) {
                            const int i = (y0 + y) * W + x0 + x;
                            float acc = 0.0f;                                /* what the first stage's line solves left out, inside the block: forward neighbour, then backward */
                            if (xl) { if (x + 1 < bw) acc += L->fE[i] * e1[y][x + 1]; if (x > 0) acc += L->fE[i - 1] * e1[y][x - 1]; }     /* x lines now: Sx e1 (the first stage solved y lines) */
                            else    { if (y + 1 < bh) acc += L->fS[i] * e1[y + 1][x]; if (y > 0) acc += L->fS[i - W] * e1[y - 1][x]; }     /* y lines now: Sy e1 */
                            e2[y][x] = acc;
                        }
                if (xl) {
                    for (int y = 0; y < bh; ++y) {
                        const int i0 = (y0 + y) * W + x0;
                        float t = rr[y][0]; ee[y][0] = t;
                        for (int x = 1; x < bw; ++x) { t = rr[y][x] + L->lxm[i0 + x] * t; ee[y][x] = t; }
                        t = ee[y][bw - 1] * L->lxp[i0 + bw - 1]; ee[y][bw - 1] = t;
                        for (int x = bw - 2; x >= 0; --x) { t = (ee[y][x] + L->fE[i0 + x] * t) * L->lxp[i0 + x]; ee[y][x] = t; }
                    }
                } else {
                    for (int x = 0; x < bw; ++x) {
                        const int i0 = y0 * W + x0 + x;
                        float t = rr[0][x]; ee[0][x] = t;
                        for (int y = 1; y < bh; ++y) { t = rr[y][x] + L->lym[i0 + y * W] * t; ee[y][x] = t; }
                        t = ee[bh - 1][x] * L->lyp[i0 + (bh - 1) * W]; ee[bh - 1][x] = t;
                        for (int y = bh - 2; y >= 0; --y) { t = (ee[y][x] + L->fS[i0 + y * W] * t) * L->lyp[i0 + y * W]; ee[y][x] = t; }
                    }
                }
            }
            for (int y = 0; y < bh; ++y)
                for (int x = 0; x < bw; ++x) {
                    const size_t j = (size_t)((y0 + y) * W + x0 + x) * NQ + q;
                    const float u = (e1[y][x] + e2[y][x]) * LINE_OM;
                    out[j] = in ? in[j] + u : u;
                }
        }
    }
}

/* z = lv[0].x (fp32) for the fp64 residual r0 (rounded to fp32 on load). scr1: [n0][NQ] float scratch */
static void vcycle(lvl_t* lv, int nl, const double* r0, float* scr1) {
    for (int l = 0; l < nl - 1; ++l) {
        lvl_t* L = &lv[l]; lvl_t* C = &lv[l + 1];
        const double* rr = l == 0 ? r0 : NULL;
        /* pre-smoothing from zero: NS sweeps, the result in L->x */
        float* cur = (MG_NS & 1) ? L->x : L->x2; float* oth = (MG_NS & 1) ? L->x2 : L->x;
        if (l == 0 && MG_LINES) {                           /* block step from zero, then NS regular sweeps: the result in L->x again */
            float* t = cur; cur = oth; oth = t;
            mg_block_step(L, rr, NULL, cur, 1);
            for (int k = 0; k < MG_NS; ++k) { mg_sweep(L, rr, cur, oth, k == 0 ? 1.0f : mg_rk(k)); t = cur; cur = oth; oth = t; }
        } else {
        mg_sweep(L, rr, NULL, cur, 1.0f);
        for (int k = 1; k < MG_NS; ++k) { mg_sweep(L, rr, cur, oth, mg_rk(k)); float* t = cur; cur = oth; oth = t; }
        }
        /* cur == L->x ; residual, then restriction */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < L->n; ++i) {
            float yv[NQ]; opf(L, i, L->x, yv);
            for (int q = 0; q < NQ; ++q) scr1[(size_t)i * NQ + q] = (l == 0 ? (float)r0[(size_t)i * NQ + q] : L->b[(size_t)i * NQ + q]) - yv[q];
        }
        mg_restrict(L, C, scr1);
    }
    {   /* coarsest: 60 Jacobi sweeps from zero */
        lvl_t* L = &lv[nl - 1];
        float* cur = L->x; float* nxt = L->x2;
        memset(cur, 0, sizeof(float) * (size_t)L->n * NQ);
        const float r0c = (float)(0.8 / MG_W[0]);
        for (int s = 0; s < 60; ++s) {
            for (int i = 0; i < L->n; ++i) {
                float y[NQ]; opf(L, i, cur, y);
                const float d = L->fdinv[i] * r0c;
                for (int q = 0; q < NQ; ++q) { const float t = L->b[(size_t)i * NQ + q] - y[q]; const float u = t * d; nxt[(size_t)i * NQ + q] = cur[(size_t)i * NQ + q] + u; }
            }
            float* t = cur; cur = nxt; nxt = t;
        }
        /* 60 is even: the result is back in L->x */
    }
    for (int l = nl - 2; l >= 0; --l) {
        lvl_t* L = &lv[l]; lvl_t* C = &lv[l + 1];
        const double* rr = l == 0 ? r0 : NULL;
        /* xe = x + P e_coarse, then NS sweeps; the result ends in L->x */
        mg_prolong(L, C, C->x, scr1);
        float* cur = (MG_NS & 1) ? L->x2 : L->x; float* oth = (MG_NS & 1) ? L->x : L->x2;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < L->n; ++j) for (int q = 0; q < NQ; ++q) cur[(size_t)j * NQ + q] = L->x[(size_t)j * NQ + q] + scr1[(size_t)j * NQ + q];
        for (int k = 0; k < MG_NS; ++k) { mg_sweep(L, rr, cur, oth, k == 0 ? 1.0f : mg_rk(k)); float* t = cur; cur = oth; oth = t; }
        if (l == 0 && MG_LINES) {                           /* cur == L->x: the mirrored block step, back into L->x through scr1 */
            mg_block_step(L, rr, L->x, scr1, 0);
            memcpy(L->x, scr1, sizeof(float) * (size_t)L->n * NQ);
        }
    }
}
#undef BVS

/* ---- hierarchy construction (fp64) */
/* interpolation weights of level L (k_mg_weights): collapse the stencil across the coarse grid line */
static void mg_weights(lvl_t* L) {
    const int W = L->W;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L->n; ++i) {
        const int r = i / W, c = i - r * W;
        double w[8]; COUP8(double, L, wE, wS, wSE, wSW, r, c, w);
        const double d = L->d[i];
        double pa = 0.0, pb = 0.0;
        /* den <= 0 cannot happen on level 0 (den = r + the two weights along the line) and did not on any Galerkin level seen; should a stencil with couplings of the
         * wrong sign produce it, the point falls back to plain averaging of its existing coarse neighbours instead of dividing by it */
        if (!(r & 1) && (c & 1)) { const double den = (d - w[2]) - w[3]; const int e2 = c + 1 < W;
                                   if (den > 0.0) { pa = ((w[1] + w[5]) + w[7]) / den; pb = ((w[0] + w[4]) + w[6]) / den; } else { pa = e2 ? 0.5 : 1.0; pb = e2 ? 0.5 : 0.0; } }     /* W: W+SW+NW ; E: E+SE+NE */
        else if ((r & 1) && !(c & 1)) { const double den = (d - w[0]) - w[1]; const int e2 = r + 1 < L->H;
                                        if (den > 0.0) { pa = ((w[3] + w[6]) + w[7]) / den; pb = ((w[2] + w[4]) + w[5]) / den; } else { pa = e2 ? 0.5 : 1.0; pb = e2 ? 0.5 : 0.0; } } /* N: N+NE+NW ; S: S+SE+SW */
        else if ((r & 1) && (c & 1)) pa = 1.0 / d;
        L->pa[i] = pa; L->pb[i] = pb;
    }
}
/* the 3x3 block of column I of P around fine point (2Y, 2X): pst[(dy+1)*3 + dx+1], 0 outside the grid (k_mg_pstencil) */
static void mg_pstencil(const lvl_t* L, int Y, int X, double* pst) {
    const int W = L->W, H = L->H, r = 2 * Y, c = 2 * X, f = r * W + c;
    for (int k = 0; k < 9; ++k) pst[k] = 0.0;
    pst[4] = 1.0;
    if (c > 0) pst[3] = L->pb[f - 1];
    if (c + 1 < W) pst[5] = L->pa[f + 1];
    if (r > 0) pst[1] = L->pb[f - W];
    if (r + 1 < H) pst[7] = L->pa[f + W];
    for (int dy = -1; dy <= 1; dy += 2)
        for (int dx = -1; dx <= 1; dx += 2) {
            const int y = r + dy, x = c + dx;
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
            double w[8]; COUP8(double, L, wE, wS, wSE, wSW, y, x, w);
            /* couplings of the centre (y, x) towards (-dy, -dx) [the coarse point], (-dy, 0) [the line point (r, x)], (0, -dx) [the line point (y, c)] */
            const double wdiag = dy < 0 ? (dx < 0 ? w[4] : w[5]) : (dx < 0 ? w[6] : w[7]);
            const double wvert = dy < 0 ? w[2] : w[3];
            const double whor = dx < 0 ? w[0] : w[1];
            pst[(dy + 1) * 3 + dx + 1] = ((wdiag + wvert * pst[3 + dx + 1]) + whor * pst[(dy + 1) * 3 + 1]) * L->pa[y * W + x];
        }
}
/* Galerkin product (k_mg_galerkin): A_c(I, J) = sum over i in block(I), row-major, of pst_I(i) * (A pst_J)(i),  (A u)(i) = d_i u_i - sum_k w_ik u_k, k = E, W, S, N, SE, SW, NE, NW */
static double galerkin_entry(const lvl_t* L, int Y, int X, const double* pI, int YJ, int XJ, const double* pJ) {
    const int W = L->W, H = L->H;
    double acc = 0.0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int y = 2 * Y + dy, x = 2 * X + dx;
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
            const double pi = pI[(dy + 1) * 3 + dx + 1];
            const int ry = y - 2 * YJ, rx = x - 2 * XJ;                       /* position of fine point i relative to block J */
            double w[8]; COUP8(double, L, wE, wS, wSE, wSW, y, x, w);
            double au = (ry >= -1 && ry <= 1 && rx >= -1 && rx <= 1) ? L->d[y * W + x] * pJ[(ry + 1) * 3 + rx + 1] : 0.0;
            for (int k = 0; k < 8; ++k) {
                if (!nbr_ok(L, y, x, k)) continue;
                const int ky = ry + OFFY[k], kx = rx + OFFX[k];
                if (ky < -1 || ky > 1 || kx < -1 || kx > 1) continue;
                au -= w[k] * pJ[(ky + 1) * 3 + kx + 1];
            }
            acc += pi * au;
        }
    return acc;
}
static void mg_galerkin(lvl_t* L, lvl_t* C) {
    const int Wc = C->W, Hc = C->H;
    double* pst = (double*)malloc(sizeof(double) * 9 * (size_t)C->n);
#pragma omp parallel for schedule(static)
    for (int I = 0; I < C->n; ++I) { mg_pstencil(L, I / Wc, I % Wc, pst + (size_t)I * 9); for (int k = 0; k < 9; ++k) L->fpst[(size_t)I * 9 + k] = (float)pst[(size_t)I * 9 + k]; }
#pragma omp parallel for schedule(static)
    for (int I = 0; I < C->n; ++I) {
        const int Y = I / Wc, X = I - Y * Wc;
        const double* pI = pst + (size_t)I * 9;
        C->d[I] = galerkin_entry(L, Y, X, pI, Y, X, pI);
        C->wE[I] = X + 1 < Wc ? -galerkin_entry(L, Y, X, pI, Y, X + 1, pst + (size_t)(I + 1) * 9) : 0.0;
        C->wS[I] = Y + 1 < Hc ? -galerkin_entry(L, Y, X, pI, Y + 1, X, pst + (size_t)(I + Wc) * 9) : 0.0;
        C->wSE[I] = (X + 1 < Wc && Y + 1 < Hc) ? -galerkin_entry(L, Y, X, pI, Y + 1, X + 1, pst + (size_t)(I + Wc + 1) * 9) : 0.0;
        C->wSW[I] = (X > 0 && Y + 1 < Hc) ? -galerkin_entry(L, Y, X, pI, Y + 1, X - 1, pst + (size_t)(I + Wc - 1) * 9) : 0.0;
    }
    free(pst);
}
/* fp32 copies + the safe smoother diagonal (k_mg_finish) */
static void mg_finish(lvl_t* L) {
    const int W = L->W;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L->n; ++i) {
        const int r = i / W, c = i - r * W;
        const double d = L->d[i];
        double dt = d;
        if (L->nine) {
            double w[8]; COUP8(double, L, wE, wS, wSE, wSW, r, c, w);
            double s = fabs(d); for (int k = 0; k < 8; ++k) s += fabs(w[k]);
            const double h = 0.5 * s; dt = h > d ? h : d;
        }
        L->fd[i] = (float)d; L->fdinv[i] = (float)(MG_W[0] / dt); L->fE[i] = (float)L->wE[i]; L->fS[i] = (float)L->wS[i];
        if (L->nine) { L->fSE[i] = (float)L->wSE[i]; L->fSW[i] = (float)L->wSW[i]; }
    }
}

/* diagnostic: iteration counts (max over the right-hand sides) of the solves since the last reset */
static int g_wls_log[64], g_wls_log_n = 0;
int orc_wls_log(int* out, int reset) { const int n = g_wls_log_n; if (out) memcpy(out, g_wls_log, sizeof(int) * (size_t)n); if (reset) g_wls_log_n = 0; return n; }

static int mg_build(lvl_t* lv, const double* lab, int H, int W, double lamda, double alpha, const double* roughness) {
    int nl = 0;
    { int h = H, w = W;
      for (;;) {
          lvl_t* L = &lv[nl]; memset(L, 0, sizeof *L); L->H = h; L->W = w; L->n = h * w; L->nine = nl > 0;
          const size_t n = (size_t)L->n;
          L->d = (double*)malloc(sizeof(double) * n); L->wE = (double*)malloc(sizeof(double) * n); L->wS = (double*)malloc(sizeof(double) * n);
          L->pa = (double*)calloc(n, sizeof(double)); L->pb = (double*)calloc(n, sizeof(double));
          L->fd = (float*)malloc(sizeof(float) * n); L->fdinv = (float*)malloc(sizeof(float) * n); L->fE = (float*)malloc(sizeof(float) * n); L->fS = (float*)malloc(sizeof(float) * n);
          L->fpst = (float*)calloc(9 * ((size_t)((h + 1) / 2) * ((w + 1) / 2)), sizeof(float));
          if (L->nine) { L->wSE = (double*)malloc(sizeof(double) * n); L->wSW = (double*)malloc(sizeof(double) * n); L->fSE = (float*)malloc(sizeof(float) * n); L->fSW = (float*)malloc(sizeof(float) * n); }
          L->b = (float*)malloc(sizeof(float) * n * NQ); L->x = (float*)malloc(sizeof(float) * n * NQ); L->x2 = (float*)malloc(sizeof(float) * n * NQ);
          ++nl;
          if (L->n <= 64 || (h <= 8 && w <= 8) || nl >= 16) break;
          h = (h + 1) / 2; w = (w + 1) / 2;
      } }
    orc_wls_system(lab, H, W, lamda, alpha, roughness, lv[0].d, lv[0].wE, lv[0].wS);      /* d = r + the 4 weights, accumulated in the order r, +x, -x, +y, -y */
    for (int l = 0; l < nl; ++l) {
        if (l > 0) mg_galerkin(&lv[l - 1], &lv[l]);       /* also rounds level l-1's transfer weights into lv[l-1].fpst */
        if (l + 1 < nl) mg_weights(&lv[l]);
        mg_finish(&lv[l]);
    }
    if (MG_LINES) mg_lines_setup(&lv[0]);
    return nl;
}
static void mg_free(lvl_t* lv, int nl) {
    for (int l = 0; l < nl; ++l) { lvl_t* L = &lv[l]; free(L->d); free(L->wE); free(L->wS); free(L->wSE); free(L->wSW); free(L->pa); free(L->pb); free(L->fd); free(L->fdinv); free(L->fE); free(L->fS);
                                   free(L->fSE); free(L->fSW); free(L->fpst); free(L->b); free(L->x); free(L->x2); free(L->lxm); free(L->lxp); free(L->lym); free(L->lyp); }
}

/* Property-test hooks (tests/test_oracle_color.py): z = Vcycle(r) for nv vectors of [n][6] doubles on the hierarchy of this system, and the level operators' statistics.
 * The cycle must be a SYMMETRIC positive definite linear map (R = P^T, the post-smoother the adjoint of the pre-smoother) for PCG to be a valid solver. */
int orc_wls_vcycle_apply(const double* lab, int H, int W, double lamda, double alpha, const double* roughness, const double* r_in, double* z_out, int nv) {
    lvl_t lv[16];
    const int nl = mg_build(lv, lab, H, W, lamda, alpha, roughness);
    const int n = lv[0].n;
    float* scr1 = (float*)malloc(sizeof(float) * (size_t)n * NQ);
    for (int v = 0; v < nv; ++v) {
        vcycle(lv, nl, r_in + (size_t)v * n * NQ, scr1);
        for (size_t j = 0; j < (size_t)n * NQ; ++j) z_out[(size_t)v * n * NQ + j] = (double)lv[0].x[j];
    }
    free(scr1); mg_free(lv, nl);
    return nl;
}
/* per level (up to 16): [n, min d, min over rows of (d - sum of couplings) = the coarse "data term", number of couplings of the wrong sign (w < 0), max dt / d] */
int orc_wls_hierarchy_stats(const double* lab, int H, int W, double lamda, double alpha, const double* roughness, double* out /*[16][5]*/) {
    lvl_t lv[16];
    const int nl = mg_build(lv, lab, H, W, lamda, alpha, roughness);
    for (int l = 0; l < nl; ++l) {
        const lvl_t* L = &lv[l];
        double dmin = 1e300, rmin = 1e300, ratio = 1.0; double neg = 0;
        for (int i = 0; i < L->n; ++i) {
            const int r = i / L->W, c = i - r * L->W;
            double w[8]; COUP8(double, L, wE, wS, wSE, wSW, r, c, w);
            double s = 0, sa = fabs(L->d[i]);
            for (int k = 0; k < 8; ++k) { s += w[k]; sa += fabs(w[k]); if (w[k] < 0) neg += 1; }
            if (L->d[i] < dmin) dmin = L->d[i];
            if (L->d[i] - s < rmin) rmin = L->d[i] - s;
            const double dt = 0.5 * sa > L->d[i] ? 0.5 * sa : L->d[i];
            if (dt / L->d[i] > ratio) ratio = dt / L->d[i];
        }
        out[l * 5] = L->n; out[l * 5 + 1] = dmin; out[l * 5 + 2] = rmin; out[l * 5 + 3] = neg; out[l * 5 + 4] = ratio;
    }
    mg_free(lv, nl);
    return nl;
}

/* a,b: full-res [N][3] in (x0) / out. iters_out[6] nullable. Returns max iterations, or -1 if not converged. */
int orc_wls_solve_mg(double* a, double* b, const double* lab, int H, int W, double lamda, double alpha, const double* roughness, double rtol, int* iters_out) {
    lvl_t lv[16];
    const int nl = mg_build(lv, lab, H, W, lamda, alpha, roughness);
    lvl_t* F = &lv[0];
    const int n = F->n;
    float* scr1 = (float*)malloc(sizeof(float) * (size_t)n * NQ);
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
        const double rg = roughness[i];
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
        vcycle(lv, nl, r, scr1);                                 /* u = F->x (fp32), widened exactly below */
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
    mg_free(lv, nl);
    free(scr1); free(x6); free(r); free(p); free(sv); free(w); free(acc);
    return any ? -1 : mx;
}
