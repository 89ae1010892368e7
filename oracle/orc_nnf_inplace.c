/* oracle/orc_nnf_inplace.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md). "parity unpinned" (no reference vectors exist).
 *
 * P1 under the REFERENCE'S OWN SCHEDULE: a literal restatement of patchmatch_single (GeneralizedPatchMatch.cu:677-831) —
 * ONE in-place NNF/distance array that every query reads from and writes to (ann[] / annd[] written after every direction,
 * :741-742,:756-757,:775-776,:793-794,:822-823), the channel sum of dist_compute_single as the sequential
 * `pixel_sum1 -= a*b` over CHW features (:376-380), the kernel's launch geometry (24x24-thread blocks over a
 * (aw/24+1) x (ah/24+1) grid, main.cu:196-201 with GPU_GRID = 24), and the reference's seeding quirk: curand_init(i, 0, 0)
 * with i = the GLOBAL X INDEX only (:60-66), i.e. every query of a column draws the same uniform sequence.
 *
 * The reference kernel is racy (both __syncthreads commented out, :801,:827): its result depends on how the hardware
 * interleaves threads. A CPU cannot reproduce "the" result; it can execute the same arithmetic under LEGAL interleavings.
 * Two are provided, the two extremes of how tightly threads advance together:
 *   ORC_PM_SEQUENTIAL (1): thread-sequential. Blocks in blockIdx.y, blockIdx.x order, threads in threadIdx.y, threadIdx.x
 *       order; each thread runs the whole kernel body (all iterations) before the next one starts. Propagation only
 *       flows from finished threads to later ones (what a 1-SM, 1-thread-at-a-time GPU would do).
 *   ORC_PM_LOCKSTEP (2): all threads advance statement by statement together (an idealised machine in which every block is
 *       resident and all warps step in lock step): for each (iteration, jump, direction) every thread reads its neighbour's
 *       CURRENT entry of the shared array and then writes its own — a sweep in thread order over the in-place array, so
 *       a thread sees the same-step update of every thread that precedes it in the order (Gauss-Seidel), as the
 *       reference's un-synchronised reads do for warps that ran earlier.
 * Neither equals the product's schedule (ORC_PM_JACOBI = 0, orc_patchmatch in orc_nnf.c: double-buffered steps, counter RNG
 * per query, 16-lane fp32 tree). tests/golden/pm_inplace_band.json holds the energy statistics of both on three seeded
 * feature pairs; tests/test_gpu_correspondence.py asserts the product's NNF energy sits in the stated band around them
 * (VERDICT r2 "missing #2": SURVEY §8c G4).
 *
 * The only stand-in is the uniform generator (cuRAND XORWOW is not available here): a counter hash keyed by
 * (seed, global x index, draw number) — the column-shared-sequence structure of the reference is kept. */
#include "orc_common.h"
#include <stdio.h>

enum { ORC_PM_JACOBI = 0, ORC_PM_SEQUENTIAL = 1, ORC_PM_LOCKSTEP = 2 };
#define ORC_GPU_GRID 24     /* Config.h: GPU_GRID; main.cu:196-201 */

/* curand_uniform stand-in: u in (0, 1]; one stream per global x index (curand_init(i, 0, 0), :60-66) */
static inline float col_rand(uint32_t seed, int gx, uint32_t draw) {
    uint32_t h = orc_mix32(seed ^ orc_mix32((uint32_t)gx * 0x9E3779B9u + draw * 0x85EBCA6Bu + 0x27D4EB2Fu));
    return (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
}

/* dist_compute_single (:355-405), literally: CHW operands, taps dy-outer / dx-inner, channels innermost, sequential
 * subtraction of separately rounded products (-ffp-contract=off: the source has no fused operation), weight = energy_weight = 1.0
 * (main.cu:214), pixel_sum = 0. dist_single (:480-493) adds nothing with flag_constraint = 0 (main.cu:212). */
static float dist_single_lit(const float* a1, const float* b1, int channels, int a_rows, int a_cols, int b_rows, int b_cols,
                             int ax, int ay, int bx, int by, int patch_w, float cutoff) {
    float pixel_sum = 0, pixel_no = 0, pixel_dist = 0, pixel_sum1 = 0;
    const size_t a_slice = (size_t)a_rows * a_cols, b_slice = (size_t)b_rows * b_cols;
    const float weight = 1.0f;
    for (int dy = -patch_w / 2; dy <= patch_w / 2; dy++)
        for (int dx = -patch_w / 2; dx <= patch_w / 2; dx++) {
            if ((ay + dy) < a_rows && (ay + dy) >= 0 && (ax + dx) < a_cols && (ax + dx) >= 0 &&
                (by + dy) < b_rows && (by + dy) >= 0 && (bx + dx) < b_cols && (bx + dx) >= 0) {
                const float* pa = a1 + (size_t)(ay + dy) * a_cols + (ax + dx);
                const float* pb = b1 + (size_t)(by + dy) * b_cols + (bx + dx);
                for (int dc = 0; dc < channels; dc++) {
                    float dp_tmp = pa[dc * a_slice] * pb[dc * b_slice];
                    pixel_sum1 -= dp_tmp;
                }
                pixel_no += 1;
            }
        }
    if (pixel_no == 0) pixel_dist = 1;
    else pixel_dist = (pixel_sum + weight * pixel_sum1) / pixel_no;
    return pixel_dist >= cutoff ? cutoff : pixel_dist;
}

typedef struct { const float *a1, *b1; int ch, a_rows, a_cols, b_rows, b_cols, patch_w, rs_max; uint32_t seed; uint32_t* ann; float* annd; long long evals; } pm_env;

/* improve_guess_single (:505-515) */
static inline void improve_lit(pm_env* e, int ax, int ay, int* xbest, int* ybest, float* dbest, int xp, int yp, float rr) {
    float d = dist_single_lit(e->a1, e->b1, e->ch, e->a_rows, e->a_cols, e->b_rows, e->b_cols, ax, ay, xp, yp, e->patch_w, *dbest);
    e->evals++;
    if (d + rr < *dbest) { *xbest = xp; *ybest = yp; *dbest = d; }
}

/* the four propagation statements of one jump (:725-796); dir 0 left, 1 right, 2 up, 3 down. Exactly the reference's
 * conditions and write-backs: `left` writes the array unconditionally (:741-742), the others only inside their if. */
static inline void propagate_dir(pm_env* e, int ax, int ay, int jump, int dir, int* xbest, int* ybest, float* dbest) {
    const int a_cols = e->a_cols, a_rows = e->a_rows, b_cols = e->b_cols, b_rows = e->b_rows;
    uint32_t vp; int xp, yp;
    switch (dir) {
    case 0:
        if ((ax - jump) < a_cols && (ax - jump) >= 0) {
            vp = e->ann[ay * a_cols + ax - jump];
            xp = orc_int_to_x(vp) + jump; yp = orc_int_to_y(vp);
            if (yp >= 0 && yp < b_rows && xp >= 0 && xp < b_cols) improve_lit(e, ax, ay, xbest, ybest, dbest, xp, yp, 0);
        }
        e->ann[ay * a_cols + ax] = orc_xy_to_int(*xbest, *ybest);
        e->annd[ay * a_cols + ax] = *dbest;
        break;
    case 1:
        if ((ax + jump) < a_cols) {
            vp = e->ann[ay * a_cols + ax + jump];
            xp = orc_int_to_x(vp) - jump; yp = orc_int_to_y(vp);
            if (yp >= 0 && yp < b_rows && xp >= 0 && xp < b_cols) {
                improve_lit(e, ax, ay, xbest, ybest, dbest, xp, yp, 0);
                e->ann[ay * a_cols + ax] = orc_xy_to_int(*xbest, *ybest); e->annd[ay * a_cols + ax] = *dbest;
            }
        }
        break;
    case 2:
        if ((ay - jump) < a_rows && (ay - jump) >= 0) {
            vp = e->ann[(ay - jump) * a_cols + ax];
            xp = orc_int_to_x(vp); yp = orc_int_to_y(vp) + jump;
            if (yp >= 0 && yp < b_rows && xp >= 0 && xp < b_cols) {
                improve_lit(e, ax, ay, xbest, ybest, dbest, xp, yp, 0);
                e->ann[ay * a_cols + ax] = orc_xy_to_int(*xbest, *ybest); e->annd[ay * a_cols + ax] = *dbest;
            }
        }
        break;
    default:
        if ((ay + jump) < a_rows) {
            vp = e->ann[(ay + jump) * a_cols + ax];
            xp = orc_int_to_x(vp); yp = orc_int_to_y(vp) - jump;
            if (yp >= 0 && yp < b_rows && xp >= 0 && xp < b_cols) {
                improve_lit(e, ax, ay, xbest, ybest, dbest, xp, yp, 0);
                e->ann[ay * a_cols + ax] = orc_xy_to_int(*xbest, *ybest); e->annd[ay * a_cols + ax] = *dbest;
            }
        }
    }
}

/* random search of one iteration (:803-823); *draw = the thread's position in its column's uniform sequence */
static inline void random_search(pm_env* e, int ax, int ay, int* xbest, int* ybest, float* dbest, uint32_t* draw) {
    int rs_start = e->rs_max;
    const int mx = e->b_cols > e->b_rows ? e->b_cols : e->b_rows;
    if (rs_start > mx) rs_start = mx;
    for (int mag = rs_start; mag >= 1; mag /= 2) {
        int xmin = *xbest - mag > 0 ? *xbest - mag : 0, xmax = *xbest + mag + 1 < e->b_cols ? *xbest + mag + 1 : e->b_cols;
        int ymin = *ybest - mag > 0 ? *ybest - mag : 0, ymax = *ybest + mag + 1 < e->b_rows ? *ybest + mag + 1 : e->b_rows;
        int xp = xmin + (int)(col_rand(e->seed, ax, (*draw)++) * (xmax - xmin)) % (xmax - xmin);
        int yp = ymin + (int)(col_rand(e->seed, ax, (*draw)++) * (ymax - ymin)) % (ymax - ymin);
        improve_lit(e, ax, ay, xbest, ybest, dbest, xp, yp, FLT_MIN);
    }
    e->ann[ay * e->a_cols + ax] = orc_xy_to_int(*xbest, *ybest);
    e->annd[ay * e->a_cols + ax] = *dbest;
}

/* the kernel body of one thread, start to end (:677-831) */
static void thread_body(pm_env* e, int ax, int ay, int pm_iters) {
    const int a_cols = e->a_cols;
    uint32_t draw = 0;
    uint32_t v = e->ann[ay * a_cols + ax];
    int xbest = orc_int_to_x(v), ybest = orc_int_to_y(v);
    float dbest;
    e->annd[ay * a_cols + ax] = dist_single_lit(e->a1, e->b1, e->ch, e->a_rows, a_cols, e->b_rows, e->b_cols, ax, ay, xbest, ybest, e->patch_w, (float)INT32_MAX);
    e->evals++;
    for (int iter = 0; iter < pm_iters; iter++) {
        v = e->ann[ay * a_cols + ax];
        xbest = orc_int_to_x(v); ybest = orc_int_to_y(v);
        dbest = e->annd[ay * a_cols + ax];
        for (int jump = 8; jump > 0; jump /= 2)
            for (int dir = 0; dir < 4; ++dir) propagate_dir(e, ax, ay, jump, dir, &xbest, &ybest, &dbest);
        random_search(e, ax, ay, &xbest, &ybest, &dbest, &draw);
    }
}

static long long g_inplace_evals = 0;
long long orc_patchmatch_inplace_last_evals(void) { return g_inplace_evals; }

/* a_chw / b_chw: L2-normalised CHW features exactly as the reference kernel receives them (Ndata_C1 / Ndata_S1).
 * nnf in/out, dist out. schedule: ORC_PM_SEQUENTIAL or ORC_PM_LOCKSTEP. Single-threaded by nature. */
int orc_patchmatch_inplace(const float* a_chw, const float* b_chw, int C, int ah, int aw, int bh, int bw, int patch, int iters, int rs_max,
                           uint32_t seed, int schedule, uint32_t* nnf, float* dist) {
    if (schedule != ORC_PM_SEQUENTIAL && schedule != ORC_PM_LOCKSTEP) return -1;
    pm_env e = {a_chw, b_chw, C, ah, aw, bh, bw, patch, rs_max, seed, nnf, dist, 0};
    const int gx = aw / ORC_GPU_GRID + 1, gy = ah / ORC_GPU_GRID + 1;
    const int n = ah * aw;
    /* the launch's thread order: blocks row-major, threads row-major inside a block */
    int* order = (int*)malloc(sizeof(int) * (size_t)n);
    int m = 0;
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx)
            for (int ty = 0; ty < ORC_GPU_GRID; ++ty)
                for (int tx = 0; tx < ORC_GPU_GRID; ++tx) {
                    const int ax = bx * ORC_GPU_GRID + tx, ay = by * ORC_GPU_GRID + ty;
                    if (ax < aw && ay < ah) order[m++] = ay * aw + ax;       /* :695 */
                }
    if (schedule == ORC_PM_SEQUENTIAL) {
        /* annd of a thread that has not started yet is read by nobody before that thread writes it: propagation reads ann[] only */
        for (int t = 0; t < n; ++t) thread_body(&e, order[t] % aw, order[t] / aw, iters);
    } else {
        int* xb = (int*)malloc(sizeof(int) * (size_t)n); int* yb = (int*)malloc(sizeof(int) * (size_t)n);
        float* db = (float*)malloc(sizeof(float) * (size_t)n); uint32_t* draw = (uint32_t*)calloc((size_t)n, sizeof(uint32_t));
        for (int t = 0; t < n; ++t) {                                     /* :710-714 */
            const int q = order[t], ax = q % aw, ay = q / aw;
            const uint32_t v = nnf[q];
            dist[q] = dist_single_lit(a_chw, b_chw, C, ah, aw, bh, bw, ax, ay, orc_int_to_x(v), orc_int_to_y(v), patch, (float)INT32_MAX);
            e.evals++;
        }
        for (int iter = 0; iter < iters; ++iter) {
            for (int t = 0; t < n; ++t) { const int q = order[t]; xb[q] = orc_int_to_x(nnf[q]); yb[q] = orc_int_to_y(nnf[q]); db[q] = dist[q]; }   /* :718-721 */
            for (int jump = 8; jump > 0; jump /= 2)
                for (int dir = 0; dir < 4; ++dir)
                    for (int t = 0; t < n; ++t) { const int q = order[t]; propagate_dir(&e, q % aw, q / aw, jump, dir, &xb[q], &yb[q], &db[q]); }
            for (int t = 0; t < n; ++t) { const int q = order[t]; random_search(&e, q % aw, q / aw, &xb[q], &yb[q], &db[q], &draw[q]); }
        }
        free(xb); free(yb); free(db); free(draw);
    }
    g_inplace_evals = e.evals;
    free(order);
    return 0;
}

/* energy statistics of a distance field: mean and the 5/25/50/75/95 percentiles (sorted copy); out[6] */
static int cmp_f(const void* a, const void* b) { float x = *(const float*)a, y = *(const float*)b; return x < y ? -1 : (x > y ? 1 : 0); }
void orc_field_stats(const float* d, int n, double* out) {
    float* s = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(s, d, sizeof(float) * (size_t)n);
    qsort(s, (size_t)n, sizeof(float), cmp_f);
    double sum = 0; for (int i = 0; i < n; ++i) sum += d[i];
    out[0] = sum / n;
    const double p[5] = {0.05, 0.25, 0.50, 0.75, 0.95};
    for (int k = 0; k < 5; ++k) { int i = (int)(p[k] * (n - 1) + 0.5); out[1 + k] = s[i]; }
    free(s);
}

/* pipeline switch (orc_pipeline.c): which PatchMatch schedule orc_process_pair* runs. Default ORC_PM_JACOBI = the product's. */
static int g_pm_schedule = ORC_PM_JACOBI;
void orc_set_pm_schedule(int s) { g_pm_schedule = s; }
int orc_get_pm_schedule(void) { return g_pm_schedule; }
