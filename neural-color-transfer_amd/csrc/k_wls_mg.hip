// k_wls_mg.hip — S2: edge-aware WLS smoothing of the (a, b) coefficient maps at full resolution.
// Reference: ColorTransfer::solve_WLS_roughness_cpu (ColorTransfer.cpp:951-1125) assembles the 5-point SPD system
//   (diag(r) + L_g) x = r * x0,   g^2 = lamda / (|dL|^alpha + 1e-4),   6 right-hand sides (a,b x 3 Lab channels)
// and factorises it on the CPU with MKL PARDISO (SparseSolver_CPU.cpp:104-286), five times per pair (n = W*H = 490k @700^2).
//
// MI355X design: conjugate gradients preconditioned by one aggregation-multigrid V(2,2) cycle, all 6 right-hand sides in
// lock step; CG recurrences, operator and dot products in fp64, the cycle (a fixed linear preconditioner) in fp32.
//  * the hierarchy needs no Galerkin triple products: with 2x2 aggregates and piecewise-constant interpolation, P^T A P of a
//    weighted 5-point graph Laplacian + diagonal is again one (coarse data term = sum of the 4 fine ones, coarse edge = sum of
//    the fine edges crossing between the two aggregates);
//  * smoother = MG_NS Chebyshev-weighted Jacobi sweeps per leg (round 3: three — 0.53, 0.97, 5.10; rounds 1-2: two) — order independent, so the result is reproducible; one cycle is two tile-fused
//    launches per level (k_mg_down / k_mg_up, iterates exchanged through LDS) and ONE workgroup for all levels <= 512 pixels;
//  * vectors are planar [6][pixels]: on a regular 5-point stencil the neighbours of consecutive pixels are consecutive, so
//    every load/store of a wave is one fully coalesced segment per right-hand side;
//  * the Krylov part is single-reduction (Chronopoulos-Gear) PCG: 3 launches per iteration beside the cycle;
//  * every dot product is the same two-stage fixed-tree reduction as in k_colorsolve.hip (mirrored by the oracle);
//  * the host never drains the stream: convergence is polled one batch behind through page-locked memory, kernels enqueued
//    past convergence return on a device flag.
// Jacobi-PCG needed 2633/1391/701/359/357 iterations (rtol 1e-10) on the five levels of a 700x700 pair (profiles/r1b); this needs
// 63/47/34/27/27 (rtol 1e-6) at ~150 us each. Roofline: Infinity-Cache/HBM streaming at 700^2 and 350^2, launch latency below.
#include "nct_internal.h"
#include <chrono>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include "nct_device.h"
#include <vector>
#include <cstring>
#include <string>
#include <thread>
#include <cstdarg>

namespace {
// Smoother: two damped-Jacobi sweeps per leg with the weights of the degree-2 Chebyshev polynomial on [lambda_max / 20, lambda_max] of D^-1 M,
// lambda_max <= 2 for these diagonally dominant M-matrices: omega = 1 / (1.05 -+ 0.95 cos(pi/4)) = 0.5808, 2.6437 (first sweep, second sweep).
// Order independent like plain Jacobi, and the pre- and post-smoother stay adjoint (polynomials in the same operator commute), so the cycle
// is still a symmetric preconditioner. Against omega = 0.8 twice (round 1): 63/47/34/27/27 instead of 74/56/42/34/34 PCG iterations on the
// five solves of a 700x700 pair (-17 %) at the same cost per cycle; [lambda_max/4, lambda_max] (0.562, 1.39) gave -8 %, wider intervals
// than /20 nothing more (scripts: NCT_MG_W1 / NCT_MG_W2 builds). The coarsest grid keeps 60 sweeps at 0.8.
// Round 3: MG_NS = 3 sweeps per leg with the degree-3 weights on [lambda_max / 30, lambda_max] (0.5346, 0.9677, 5.0974): 60/41/29/22/22 instead of 71/53/39/31/31 iterations
// (-24 %) for legs that cost ~1.3x (halo 3 instead of 2) — and, at rtol 1e-7, the same 8-bit result. Degree 4 (0.5193, 0.7153, 1.5340, 8.0502 on [lambda_max / 40, lambda_max]):
// -35 % iterations, legs ~1.7x. NCT_MG_NS selects 2 / 3 / 4 at build time (the oracle mirrors it: orc_set_mg_smoother).
#ifndef NCT_MG_NS
#define NCT_MG_NS 3
#endif
constexpr int MG_NS = NCT_MG_NS;
static_assert(MG_NS >= 2 && MG_NS <= 4, "2, 3 or 4 smoothing sweeps per leg");
constexpr double MG_W[4] = {MG_NS == 2 ? 0.5808 : (MG_NS == 3 ? 0.5346 : 0.5193), MG_NS == 2 ? 2.6437 : (MG_NS == 3 ? 0.9677 : 0.7153), MG_NS == 3 ? 5.0974 : 1.5340, 8.0502};
constexpr double OMEGA = MG_W[0];
// weight of sweep k relative to the first (fdinv = omega_0 / diag is what the levels store); coarsest grid: 0.8
__host__ __device__ constexpr float mg_rk(int k) { return (float)(MG_W[k] / MG_W[0]); }
constexpr float MG_R0 = (float)(0.8 / MG_W[0]);
constexpr int NQMAX = 6;      // right-hand sides of a solve: 6 (a and b of the 3 Lab channels) or 3 + 3 on two streams (template parameter NQ)
#ifndef NCT_MG_TXB
#define NCT_MG_TXB 32
#define NCT_MG_TYB 16
#endif

// The PCG itself (and the hierarchy construction) is fp64; the V-cycle — a fixed linear preconditioner, whose accuracy does not
// limit the accuracy of the solution — runs in fp32 on rounded copies of the level operators: half the bytes on the two
// bandwidth-bound levels, same iteration counts (scripts/mg_convergence_experiments.py).
//
// Hierarchy (round 4; rounds 1-3: 2x2 aggregation with piecewise-constant transfer, 174 PCG iterations per 700x700 pair): VERTEX-CENTRED coarsening —
// coarse point (Y, X) IS fine point (2Y, 2X) — with OPERATOR-DEPENDENT interpolation (black-box multigrid, Alcouffe/Brandt/Dendy/Painter): a fine point on
// a coarse grid line interpolates from its two coarse neighbours with the weights of the stencil collapsed across the line, a fine point in the middle of
// a coarse cell solves its own equation for its 8 neighbours. Restriction = transpose, coarse operators = Galerkin products P^T A P: symmetric 9-point
// stencils kept as the diagonal and the four forward couplings (+x, +y, +x+y, -x+y), all in the form w = -A(i, j) so that level 0 keeps (diag, wx, wy).
// The edge-aware weights vary by 10^4 between neighbouring edges, which a piecewise-constant transfer cannot follow; this one halves the iteration count.
typedef float vf;
struct Lvl { int H, W, n, nine;                                  // nine: 9-point stencil (every level but the finest)
             double *d, *wE, *wS, *wSE, *wSW, *pa, *pb;          // fp64 operator: (A v)_i = d_i v_i - sum_k w_ik v_k; forward couplings, 0 where the neighbour does not exist; line weights
             vf *fd, *fdinv, *fE, *fS, *fSE, *fSW;               // fp32 copies; fdinv = (float)(omega_0 / dt), dt = the safe smoother diagonal (k_mg_finish)
             vf *fpst, *fpw;                                     // transfer to the next coarser level: fpst[I][9] = column I of P as a 3x3 block (restriction reads it),
                                                                 // fpw[4][n] = the same numbers per FINE point, one plane per parent NW, NE, SW, SE / W, E / N, S (prolongation)
             vf *b, *x, *x2;                                     // V-cycle vectors, planar [6][n]
             vf *lxm, *lxp, *lym, *lyp; };                       // finest level, block step: Thomas factors of the x / y lines cut at the blocks (k_mg_lines_setup)
// pa / pb (fp64, construction only) of a fine point on a coarse grid line = its weights from the W / E (even y, odd x) or N / S (odd y, even x) coarse point;
// of a cell centre (odd, odd): pa = 1 / d (the centre is eliminated exactly). k_mg_pstencil turns them into the columns of P.

// State of the 6 right-hand sides of the single-reduction (Chronopoulos-Gear) PCG, double buffered: the update kernel of iteration k
// reads st[k & 1] and (workgroup 0) writes st[(k + 1) & 1]. nactive = number of systems still iterating: the host polls it only every
// few iterations, and every kernel of an iteration enqueued past convergence returns at once when it is 0.
struct PState { double gam[6], alp[6], bb[6]; int active[6]; int iters[6]; int nactive; int seq; double rho[6]; };   // rho: r.r the last iteration saw (host-side convergence forecast only); seq: publication number of a host-visible copy (pcg_publish)

// Fixed 256-wide tree s[t] += s[t + off], off = 128 … 1 (the order the oracle mirrors), evaluated with two barriers instead of nine: the two
// cross-wave steps go through LDS, the six steps inside the first wave are lane shifts (a lane t < off adds the value lane t + off held BEFORE
// the step, exactly as the array form does; what lanes >= off compute is never used).
template <int NV>
__device__ __forceinline__ void mg_block_reduce(double (&v)[NV], double* __restrict__ partial, int lb = -1 /* logical block whose slot receives the sums (default: blockIdx.x) */) {
    __shared__ double s_red[128 * NV];
    const int t = threadIdx.x;
    if (t >= 128) {
#pragma unroll
        for (int q = 0; q < NV; ++q) s_red[q * 128 + t - 128] = v[q];
    }
    __syncthreads();
    if (t < 128) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] += s_red[q * 128 + t];            // off = 128
    }
    __syncthreads();
    if (t >= 64 && t < 128) {
#pragma unroll
        for (int q = 0; q < NV; ++q) s_red[q * 128 + t - 64] = v[q];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double x = v[q] + s_red[q * 128 + t];                          // off = 64
            x += __shfl_down(x, 32); x += __shfl_down(x, 16); x += __shfl_down(x, 8);
            x += __shfl_down(x, 4); x += __shfl_down(x, 2); x += __shfl_down(x, 1);
            if (t == 0) partial[(size_t)(lb < 0 ? (int)blockIdx.x : lb) * NV + q] = x;
        }
    }
}
template <int NV>
__device__ __forceinline__ void mg_final_reduce(const double* __restrict__ partial, int nb, double (&out)[NV]) {
    __shared__ double s_fin[256 * NV];
    const int t = threadIdx.x;
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0.0;
    for (int b = t; b < nb; b += 256)
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[q] += partial[(size_t)b * NV + q];
#pragma unroll
    for (int q = 0; q < NV; ++q) s_fin[q * 256 + t] = acc[q];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (t < off)
#pragma unroll
            for (int q = 0; q < NV; ++q) s_fin[q * 256 + t] += s_fin[q * 256 + t + off];
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) out[q] = s_fin[q * 256];
    __syncthreads();
}

// same fixed order over one group of NV values inside records of `stride` doubles
template <int NV>
__device__ __forceinline__ void mg_final_reduce_strided(const double* __restrict__ partial, int nb, int stride, double (&out)[NV]) {
    __shared__ double s_fin2[256 * NV];
    const int t = threadIdx.x;
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0.0;
    for (int b = t; b < nb; b += 256)
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[q] += partial[(size_t)b * stride + q];
#pragma unroll
    for (int q = 0; q < NV; ++q) s_fin2[q * 256 + t] = acc[q];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (t < off)
#pragma unroll
            for (int q = 0; q < NV; ++q) s_fin2[q * 256 + t] += s_fin2[q * 256 + t + off];
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) out[q] = s_fin2[q * 256];
    __syncthreads();
}

// y = M v at pixel i of the FINE level (5-point: diag*v - sum_w w*v_nbr, neighbour order +x, -x, +y, -y)
template <int NQ, typename F>
__device__ __forceinline__ void lvl_op(const Lvl& L, int i, F&& val /* val(j, q) */, double (&y)[NQ]) {
    const int W = L.W, H = L.H;
    const int r = i / W, c = i - r * W;
    const double d = L.d[i];
    // The four couplings and the four neighbour values of every right-hand side are requested TOGETHER (an absent neighbour reads the pixel itself and is not subtracted):
    // written as `if (exists) { w = wE[..]; y -= w * val(..) }` each direction was a branch of its own with two dependent global round trips inside (round 6, from the ISA).
    // Subtracted in the same order (+x, -x, +y, -y): same bits.
    const bool e0 = c + 1 < W, e1 = c > 0, e2 = r + 1 < H, e3 = r > 0;
    const int j0 = e0 ? i + 1 : i, j1 = e1 ? i - 1 : i, j2 = e2 ? i + W : i, j3 = e3 ? i - W : i;
    const double w0 = L.wE[i], w1 = L.wE[j1], w2 = L.wS[i], w3 = L.wS[j3];
    double v0[NQ], v1[NQ], v2[NQ], v3[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { y[q] = val(i, q); v0[q] = val(j0, q); v1[q] = val(j1, q); v2[q] = val(j2, q); v3[q] = val(j3, q); }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double t = d * y[q];
        t = e0 ? t - w0 * v0[q] : t;
        t = e1 ? t - w1 * v1[q] : t;
        t = e2 ? t - w2 * v2[q] : t;
        t = e3 ? t - w3 * v3[q] : t;
        y[q] = t;
    }
}

// ---- hierarchy construction (fp64; the oracle mirrors every expression: oracle/orc_wls_mg.c mg_weights / mg_pstencil / mg_galerkin / mg_finish)
// neighbour k of a pixel: E, W, S, N, SE, SW, NE, NW
__device__ __forceinline__ int nb_dy(int k) { return k == 2 || k == 4 || k == 5 ? 1 : (k == 3 || k == 6 || k == 7 ? -1 : 0); }
__device__ __forceinline__ int nb_dx(int k) { return k == 0 || k == 4 || k == 6 ? 1 : (k == 1 || k == 5 || k == 7 ? -1 : 0); }
// existence mask of the 8 neighbours of (r, c) on level L (diagonals only on 9-point levels)
__device__ __forceinline__ unsigned nb_mask(const Lvl& L, int r, int c) {
    const bool xr = c + 1 < L.W, xl = c > 0, yd = r + 1 < L.H, yu = r > 0;
    unsigned m = (xr ? 1u : 0u) | (xl ? 2u : 0u) | (yd ? 4u : 0u) | (yu ? 8u : 0u);
    if (L.nine) m |= (xr && yd ? 16u : 0u) | (xl && yd ? 32u : 0u) | (xr && yu ? 64u : 0u) | (xl && yu ? 128u : 0u);
    return m;
}
// the 8 couplings of pixel (r, c), 0 where the neighbour does not exist
__device__ __forceinline__ void coup8(const Lvl& L, int r, int c, double (&w)[8]) {
    const int W = L.W, i = r * W + c;
    const unsigned m = nb_mask(L, r, c);
    w[0] = (m & 1u) ? L.wE[i] : 0.0; w[1] = (m & 2u) ? L.wE[i - 1] : 0.0; w[2] = (m & 4u) ? L.wS[i] : 0.0; w[3] = (m & 8u) ? L.wS[i - W] : 0.0;
    w[4] = (m & 16u) ? L.wSE[i] : 0.0; w[5] = (m & 32u) ? L.wSW[i] : 0.0; w[6] = (m & 64u) ? L.wSW[i - W + 1] : 0.0; w[7] = (m & 128u) ? L.wSE[i - W - 1] : 0.0;
}
// level 0: diagonal of the reference's system (accumulated in its order: data term, +x, -x, +y, -y) and the fp32 copies
__global__ void k_mg_diag(Lvl L, const double* __restrict__ rough) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    const int y = i / L.W, x = i - y * L.W;
    double a00 = 0.0;
    a00 += rough[i];
    if (x + 1 < L.W) a00 += L.wE[i];
    if (x > 0) a00 += L.wE[i - 1];
    if (y + 1 < L.H) a00 += L.wS[i];
    if (y > 0) a00 += L.wS[i - L.W];
    L.d[i] = a00;
    L.fd[i] = (vf)a00; L.fdinv[i] = (vf)(OMEGA / a00); L.fE[i] = (vf)L.wE[i]; L.fS[i] = (vf)L.wS[i];
}
// interpolation weights towards the next coarser level: the stencil collapsed across the coarse grid line
__device__ __forceinline__ void mg_weights_at(const Lvl& L, int i) {
    const int r = i / L.W, c = i - r * L.W;
    double w[8]; coup8(L, r, c, w);
    const double d = L.d[i];
    double pa = 0.0, pb = 0.0;
    // (den <= 0 cannot happen on level 0 and did not on any Galerkin level seen; such a point would fall back to plain averaging of its existing coarse neighbours)
    if (!(r & 1) && (c & 1)) { const double den = (d - w[2]) - w[3]; const bool e2 = c + 1 < L.W;
                               if (den > 0.0) { pa = ((w[1] + w[5]) + w[7]) / den; pb = ((w[0] + w[4]) + w[6]) / den; } else { pa = e2 ? 0.5 : 1.0; pb = e2 ? 0.5 : 0.0; } }
    else if ((r & 1) && !(c & 1)) { const double den = (d - w[0]) - w[1]; const bool e2 = r + 1 < L.H;
                                    if (den > 0.0) { pa = ((w[3] + w[6]) + w[7]) / den; pb = ((w[2] + w[4]) + w[5]) / den; } else { pa = e2 ? 0.5 : 1.0; pb = e2 ? 0.5 : 0.0; } }
    else if ((r & 1) && (c & 1)) pa = 1.0 / d;
    L.pa[i] = pa; L.pb[i] = pb;
}
__global__ void k_mg_weights(Lvl L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    mg_weights_at(L, i);
}
// column I of P as a 3x3 block around fine point (2Y, 2X): pst[I*9 + (dy+1)*3 + dx+1]
__device__ __forceinline__ void mg_pstencil_at(const Lvl& L, const Lvl& C, double* pst_out, int I) {      // also L.fpst = (float) of it: THE transfer weights of the cycle
    const int Y = I / C.W, X = I - Y * C.W;
    const int W = L.W, H = L.H, r = 2 * Y, c = 2 * X, f = r * W + c;
    double pst[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) pst[k] = 0.0;
    pst[4] = 1.0;
    if (c > 0) pst[3] = L.pb[f - 1];
    if (c + 1 < W) pst[5] = L.pa[f + 1];
    if (r > 0) pst[1] = L.pb[f - W];
    if (r + 1 < H) pst[7] = L.pa[f + W];
#pragma unroll
    for (int dy = -1; dy <= 1; dy += 2)
#pragma unroll
        for (int dx = -1; dx <= 1; dx += 2) {
            const int y = r + dy, x = c + dx;
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
            double w[8]; coup8(L, y, x, w);
            const double wdiag = dy < 0 ? (dx < 0 ? w[4] : w[5]) : (dx < 0 ? w[6] : w[7]);
            const double wvert = dy < 0 ? w[2] : w[3];
            const double whor = dx < 0 ? w[0] : w[1];
            pst[(dy + 1) * 3 + dx + 1] = ((wdiag + wvert * pst[3 + dx + 1]) + whor * pst[(dy + 1) * 3 + 1]) * L.pa[y * W + x];
        }
#pragma unroll
    for (int k = 0; k < 9; ++k) { pst_out[(size_t)I * 9 + k] = pst[k]; L.fpst[(size_t)I * 9 + k] = (vf)pst[k]; }
}
__global__ void k_mg_pstencil(Lvl L, Lvl C, double* __restrict__ pst_out) {
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= C.n) return;
    mg_pstencil_at(L, C, pst_out, I);
}
// the same weights per fine point: plane s = parent s of the point in the order NW, NE, SW, SE (cell centre) / W, E / N, S (line points); 0 for absent parents and coarse points
__device__ __forceinline__ void mg_pweights_at(const Lvl& L, const Lvl& C, int i) {
    const int r = i / L.W, c = i - r * L.W;
    const int Y0 = r >> 1, X0 = c >> 1, ny = (r & 1) ? 2 : 1, nx = (c & 1) ? 2 : 1;
    vf pw[4] = {0.f, 0.f, 0.f, 0.f};
    if (ny * nx > 1) {
        int sl = 0;
        for (int jy = 0; jy < ny; ++jy)
            for (int jx = 0; jx < nx; ++jx, ++sl) {
                const int Y = Y0 + jy, X = X0 + jx;
                if (Y < C.H && X < C.W) pw[sl] = L.fpst[(size_t)(Y * C.W + X) * 9 + (r - 2 * Y + 1) * 3 + (c - 2 * X + 1)];
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) L.fpw[(size_t)k * L.n + i] = pw[k];
}
__global__ void k_mg_pweights(Lvl L, Lvl C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    mg_pweights_at(L, C, i);
}
// Galerkin product: A_c(I, J) = sum over the fine points i of block(I), row-major, of pst_I(i) * (A pst_J)(i) for J = I and its four forward neighbours
__device__ __forceinline__ void mg_galerkin_at(const Lvl& L, const Lvl& C, const double* pst, int I) {
    const int Wc = C.W, Hc = C.H, Y = I / Wc, X = I - Y * Wc;
    const int W = L.W, H = L.H;
    // J = (Y + JY[j], X + JX[j]): self, E, S, SE, SW
    const int JY[5] = {0, 0, 1, 1, 1}, JX[5] = {0, 1, 0, 1, -1};
    bool jok[5]; double acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { const int yj = Y + JY[j], xj = X + JX[j]; jok[j] = yj < Hc && xj >= 0 && xj < Wc; acc[j] = 0.0; }
    const double* pI = pst + (size_t)I * 9;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int y = 2 * Y + dy, x = 2 * X + dx;
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
            const double pi = pI[(dy + 1) * 3 + dx + 1];
            double w[8]; coup8(L, y, x, w);
            const unsigned m = nb_mask(L, y, x);
            const double di = L.d[y * W + x];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (!jok[j]) continue;
                const double* pJ = pst + (size_t)(I + JY[j] * Wc + JX[j]) * 9;
                const int ry = dy - 2 * JY[j], rx = dx - 2 * JX[j];                 // position of fine point i relative to block J
                double au = (ry >= -1 && ry <= 1 && rx >= -1 && rx <= 1) ? di * pJ[(ry + 1) * 3 + rx + 1] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (!(m & (1u << k))) continue;
                    const int ky = ry + nb_dy(k), kx = rx + nb_dx(k);
                    if (ky < -1 || ky > 1 || kx < -1 || kx > 1) continue;
                    au -= w[k] * pJ[(ky + 1) * 3 + kx + 1];
                }
                acc[j] += pi * au;
            }
        }
    C.d[I] = acc[0];
    C.wE[I] = jok[1] ? -acc[1] : 0.0;
    C.wS[I] = jok[2] ? -acc[2] : 0.0;
    C.wSE[I] = jok[3] ? -acc[3] : 0.0;
    C.wSW[I] = jok[4] ? -acc[4] : 0.0;
}
__global__ void k_mg_galerkin(Lvl L, Lvl C, const double* __restrict__ pst) {
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= C.n) return;
    mg_galerkin_at(L, C, pst, I);
}
// 9-point levels: fp32 copies and the safe smoother diagonal dt = max(d, (|d| + sum |w|) / 2) — Gershgorin keeps lambda_max(dt^-1 A) <= 2 where a Galerkin
// stencil has couplings of the wrong sign; dt = d on M-matrix rows
__device__ __forceinline__ void mg_finish_at(const Lvl& L, int i) {
    const int r = i / L.W, c = i - r * L.W;
    const double d = L.d[i];
    double w[8]; coup8(L, r, c, w);
    double s = fabs(d);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += fabs(w[k]);
    const double h = 0.5 * s, dt = h > d ? h : d;
    L.fd[i] = (vf)d; L.fdinv[i] = (vf)(OMEGA / dt); L.fE[i] = (vf)L.wE[i]; L.fS[i] = (vf)L.wS[i]; L.fSE[i] = (vf)L.wSE[i]; L.fSW[i] = (vf)L.wSW[i];
}
__global__ void k_mg_finish(Lvl L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    mg_finish_at(L, i);
}

// The small levels of the hierarchy in ONE launch: one 1024-thread workgroup walks levels l0 .. nl-1 (coarse level of at most MG_TAIL_N points each) through
// the same five stages with the same per-point functions; a stage boundary is a workgroup barrier instead of a kernel boundary (5 launches of ~5 us of dependent
// latency each per level before: 24 launches and ~0.12 ms per solve at 700x700). Needs pa / pb of level l0 - 1 (k_mg_weights) done.
#ifndef NCT_MG_TAIL_N
#define NCT_MG_TAIL_N 512
#endif
constexpr int MG_TAIL_N = NCT_MG_TAIL_N;
constexpr int MG_MAXL = 12;
struct LvlPack { Lvl lv[MG_MAXL]; };
__global__ __launch_bounds__(1024) void k_mg_setup_tail(LvlPack P, int l0, int nl, double* pst) {
    const int t = threadIdx.x;
    for (int l = l0; l < nl; ++l) {
        const Lvl& L = P.lv[l - 1]; const Lvl& C = P.lv[l];
        for (int I = t; I < C.n; I += 1024) mg_pstencil_at(L, C, pst, I);
        __syncthreads();
        for (int i = t; i < L.n; i += 1024) mg_pweights_at(L, C, i);
        for (int I = t; I < C.n; I += 1024) mg_galerkin_at(L, C, pst, I);
        __syncthreads();
        for (int i = t; i < C.n; i += 1024) mg_finish_at(C, i);
        if (l + 1 < nl) {
            for (int i = t; i < C.n; i += 1024) mg_weights_at(C, i);     // reads d and the couplings (k_mg_galerkin's output), not what mg_finish_at writes
        }
        __syncthreads();
    }
}

// ---- V-cycle (fp32, vectors planar [6][n])
// tile-fused legs: every intermediate iterate of a leg lives in LDS for a TX x TY fine tile plus a halo (recomputed by the
// neighbouring tiles with the same expressions, hence bit-identical) instead of making a round trip through global memory and a
// launch per sweep:
//   down: x1 = b*dinv ; x_{k+1} = x_k + (b - M x_k)*dinv*rk (MG_NS Chebyshev-weighted Jacobi sweeps from zero) ; res = b - M x ; coarse rhs = P^T res
//   up:   xe = x + P e_coarse ; MG_NS more sweeps
// One thread per pixel of the tile + halo: it loads ITS pixel's right-hand side, coefficients and inputs once (all loads of a leg are issued in the
// first phase), iterates are exchanged through two LDS arrays in ping-pong. Tiles start at even coordinates, so a tile owns the coarse points
// (even, even) inside it. The transfers reach one pixel further than the smoother on ONE side (the coarse point at the tile's left / top edge
// gathers from the line point in front of it; for an even sweep count the up leg's outermost column needs the coarse point behind it), so the
// thread grid has MG_NS + 1 halo pixels on that side and MG_NS on the other: (TX + 2 MG_NS + 1) x (TY + 2 MG_NS + 1) threads.
struct PxCoef { vf d, dinv, w[8]; unsigned ex; };    // diag, omega/dt, couplings E W S N SE SW NE NW (0 where absent), existence mask (diagonal bits on 9-point levels only)
template <bool NINE>
__device__ __forceinline__ PxCoef px_coef(const Lvl& L, int gy, int gx) {
    const int i = gy * L.W + gx, W = L.W;
    PxCoef c;
    const bool xr = gx + 1 < W, xl = gx > 0, yd = gy + 1 < L.H, yu = gy > 0;
    c.ex = (xr ? 1u : 0u) | (xl ? 2u : 0u) | (yd ? 4u : 0u) | (yu ? 8u : 0u);
    c.d = L.fd[i]; c.dinv = L.fdinv[i];
    c.w[0] = xr ? L.fE[i] : 0.f; c.w[1] = xl ? L.fE[i - 1] : 0.f; c.w[2] = yd ? L.fS[i] : 0.f; c.w[3] = yu ? L.fS[i - W] : 0.f;
    if (NINE) {
        c.ex |= (xr && yd ? 16u : 0u) | (xl && yd ? 32u : 0u) | (xr && yu ? 64u : 0u) | (xl && yu ? 128u : 0u);
        c.w[4] = (xr && yd) ? L.fSE[i] : 0.f; c.w[5] = (xl && yd) ? L.fSW[i] : 0.f; c.w[6] = (xr && yu) ? L.fSW[i - W + 1] : 0.f; c.w[7] = (xl && yu) ? L.fSE[i - W - 1] : 0.f;
    } else { c.w[4] = c.w[5] = c.w[6] = c.w[7] = 0.f; }
    return c;
}
template <int LW> __device__ __forceinline__ constexpr int lds_off(int k) {
    return k == 0 ? 1 : k == 1 ? -1 : k == 2 ? LW : k == 3 ? -LW : k == 4 ? LW + 1 : k == 5 ? LW - 1 : k == 6 ? -LW + 1 : -LW - 1;
}
// y = M v at the pixel stored at LDS position p of a grid with row pitch LW (neighbour order E, W, S, N, SE, SW, NE, NW). `own` = the thread's own value of v,
// which it wrote to s_v[.. + p] itself and still holds in registers
template <int NQ, int LW, int LN, bool NINE>
__device__ __forceinline__ void lds_op(const PxCoef& c, const vf* __restrict__ s_v, int p, const vf (&own)[NQ], vf (&y)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) y[q] = c.d * own[q];
#pragma unroll
    for (int k = 0; k < (NINE ? 8 : 4); ++k)
        if (c.ex & (1u << k)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) y[q] -= c.w[k] * s_v[q * LN + p + lds_off<LW>(k)];
        }
    // (round 6: every coupling is a branch of its own with its LDS round trip inside — 58-105 exec-mask branches per leg in the ISA. Reading the neighbours unconditionally
    //  and adding w * (exists ? v : 0), as mid_op below does, halves the branches but costs the 9-point legs 90 instead of 52 registers: 9.2 / 9.8 against 9.1 / 9.5 us; not kept here)
}
// XCD-aware tile order (round 4): consecutive workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2. A 2-D grid puts neighbouring tiles on different
// XCDs, so every halo pixel was fetched from the fabric again (k_mg_up: 131 MB per launch for 66 MB of compulsory bytes, 84 % L2 misses). The legs run on a 1-D grid and
// give each XCD a contiguous band of tile rows: neighbouring tiles meet in the same L2. Pure scheduling: which workgroup computes a tile has no influence on its values.
__device__ __forceinline__ int mg_tile_of_block(int bid, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
constexpr int mg_threads(int TX, int TY, bool X0 = false) { return ((TX + 2 * MG_NS + 1 + (X0 ? 2 : 0)) * (TY + 2 * MG_NS + 1 + (X0 ? 2 : 0)) + 63) / 64 * 64; }
// TB = type of this level's right-hand side in memory (vf everywhere since round 5: the finest level reads the fp32 copy of the PCG residual that k_cg_update writes next to the fp64 one).
// Sweep k produces its iterate on ring(k) = the thread grid shrunk by k from every side; the residual lives on ring(MG_NS) = the tile + one pixel to the left / top.
// Restriction R = P^T (oracle: mg_restrict): the thread of a coarse point sums fpst[I][k] * res over the 3x3 block around it, row-major.
// X0 (block step, NCT_S2_LINES): the leg starts from the iterate x1 instead of zero, so sweep 0 is a regular sweep too and every ring moves in by one: one more halo pixel per side.
template <int NQ, int TX, int TY, typename TB, bool NINE, bool X0 = false>
__global__ __launch_bounds__(mg_threads(TX, TY, X0)) void k_mg_down(const PState* __restrict__ st, Lvl F, const TB* __restrict__ b, vf* __restrict__ x, Lvl C, vf* __restrict__ bc, const vf* __restrict__ x1 = nullptr) {
    const int nact = st->nactive;   // checked below, behind the kernel's first loads: as the first statement it was a dependent scalar round trip in front of everything (round 6)
    constexpr int S0 = X0 ? 1 : 0;
    constexpr int HA = MG_NS + 1 + S0, HB = MG_NS + S0, LW = TX + HA + HB, LH = TY + HA + HB, LN = LW * LH;
    static_assert(LN <= mg_threads(TX, TY, X0) && mg_threads(TX, TY, X0) <= 1024, "one thread per pixel of the tile and its halo");
    __shared__ vf s_a[NQ * LN], s_b[NQ * LN];
    const int tiles_x = (F.W + TX - 1) / TX, tile = mg_tile_of_block(blockIdx.x, gridDim.x);
    const int x0 = (tile % tiles_x) * TX, y0 = (tile / tiles_x) * TY;
    const int p = threadIdx.x;
    const int ly = p / LW, lx = p - ly * LW;
    const int gy = y0 + ly - HA, gx = x0 + lx - HA;
    const bool valid = p < LN && gy >= 0 && gy < F.H && gx >= 0 && gx < F.W;
    auto ring = [&](int k) { return valid && lx >= k && lx <= LW - 1 - k && ly >= k && ly <= LH - 1 - k; };
    const bool interior = valid && lx >= HA && lx < HA + TX && ly >= HA && ly < HA + TY;
    const bool coarse = interior && !(gy & 1) && !(gx & 1);                    // this thread's pixel IS a coarse point: it gathers the restricted residual
    const int i = gy * F.W + gx;
    vf bq[NQ], xk[NQ], ps[9]; PxCoef c;
    if (valid) {
        c = px_coef<NINE>(F, gy, gx);
        if (coarse) {
            const vf* pp = F.fpst + (size_t)((gy >> 1) * C.W + (gx >> 1)) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) ps[k] = pp[k];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) { bq[q] = (vf)b[(size_t)q * F.n + i]; xk[q] = X0 ? x1[(size_t)q * F.n + i] : bq[q] * c.dinv; s_a[q * LN + p] = xk[q]; }     // sweep 0 (from zero) / the given iterate
    }
    if (nact == 0) return;                                   // a launch enqueued past convergence: nothing has been stored to global memory yet
    __syncthreads();
#pragma unroll
    for (int k = 1 - S0; k < MG_NS; ++k) {
        vf* src = ((k + S0) & 1) ? s_a : s_b; vf* dst = ((k + S0) & 1) ? s_b : s_a;
        if (ring(k + S0)) {
            vf y[NQ]; lds_op<NQ, LW, LN, NINE>(c, src, p, xk, y);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                xk[q] = xk[q] + (bq[q] - y[q]) * (k == 0 ? c.dinv : c.dinv * mg_rk(k));
                dst[q * LN + p] = xk[q];
                if (k == MG_NS - 1 && interior) x[(size_t)q * F.n + i] = xk[q];
            }
        }
        __syncthreads();
    }
    vf* xs = ((MG_NS + S0) & 1) ? s_a : s_b; vf* rs = ((MG_NS + S0) & 1) ? s_b : s_a;       // the smoothed iterate, and where the residual goes
    if (ring(MG_NS + S0)) {
        vf yv[NQ]; lds_op<NQ, LW, LN, NINE>(c, xs, p, xk, yv);
#pragma unroll
        for (int q = 0; q < NQ; ++q) rs[q * LN + p] = bq[q] - yv[q];
    }
    __syncthreads();
    if (coarse) {
        const bool xr = (c.ex & 1u) != 0, xl = (c.ex & 2u) != 0, yd = (c.ex & 4u) != 0, yu = (c.ex & 8u) != 0;
        const bool in[9] = {xl && yu, yu, xr && yu, xl, true, xr, xl && yd, yd, xr && yd};
        vf acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (in[k]) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] += ps[k] * rs[q * LN + p + (k / 3 - 1) * LW + (k % 3 - 1)];
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) bc[(size_t)q * C.n + (gy >> 1) * C.W + (gx >> 1)] = acc[q];
    }
}
// Prolongation (oracle: mg_prolong): coarse points copy e_c; every other point ((p0 e0 + p1 e1) + p2 e2) + p3 e3 over its existing coarse parents NW, NE, SW, SE (W, E / N, S on a line).
// xo must not alias x (neighbouring tiles still read x for their halo)
template <int NQ, int TX, int TY, typename TB, bool NINE>
__global__ __launch_bounds__(mg_threads(TX, TY)) void k_mg_up(const PState* __restrict__ st, Lvl L, const TB* __restrict__ b, const vf* __restrict__ x, int Wc, int nc,
                                                              const vf* __restrict__ ec, vf* __restrict__ xo) {
    const int nact = st->nactive;   // checked below, behind the kernel's first loads: as the first statement it was a dependent scalar round trip in front of everything (round 6)
    constexpr bool ODD = (MG_NS & 1) != 0;
    constexpr int HA = ODD ? MG_NS + 1 : MG_NS, HB = ODD ? MG_NS : MG_NS + 1, LW = TX + HA + HB, LH = TY + HA + HB, LN = LW * LH;
    constexpr int OL = ODD ? 1 : 0, OR = ODD ? 0 : 1;                          // xe lives on the grid minus its first (odd sweep count) / last (even) row and column
    __shared__ vf s_a[NQ * LN], s_b[NQ * LN];
    const int tiles_x = (L.W + TX - 1) / TX, tile = mg_tile_of_block(blockIdx.x, gridDim.x);
    const int x0 = (tile % tiles_x) * TX, y0 = (tile / tiles_x) * TY;
    const int p = threadIdx.x;
    const int ly = p / LW, lx = p - ly * LW;
    const int gy = y0 + ly - HA, gx = x0 + lx - HA;
    const bool valid = p < LN && gy >= 0 && gy < L.H && gx >= 0 && gx < L.W;
    auto ring = [&](int k) { return valid && lx >= OL + k && lx <= LW - 1 - OR - k && ly >= OL + k && ly <= LH - 1 - OR - k; };
    const bool oy = (gy & 1) != 0, ox = (gx & 1) != 0;
    const int i = gy * L.W + gx;
    vf bq[NQ], xk[NQ], pw[4]; PxCoef c;
    if (valid) {
        c = px_coef<NINE>(L, gy, gx);
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[k] = L.fpw[(size_t)k * L.n + i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { bq[q] = (vf)b[(size_t)q * L.n + i]; xk[q] = x[(size_t)q * L.n + i]; }
        if (!oy && !ox) {
            const int ip = (gy >> 1) * Wc + (gx >> 1);
#pragma unroll
            for (int q = 0; q < NQ; ++q) s_a[q * LN + p] = ec[(size_t)q * nc + ip];
        }
    }
    if (nact == 0) return;
    __syncthreads();
    if (ring(0)) {                                                             // xe = x + P e_coarse -> s_b (s_a keeps the coarse values the neighbours still read)
        const bool xr = (c.ex & 1u) != 0, yd = (c.ex & 4u) != 0;
        // parents in the order NW, NE, SW, SE (centre) / W, E (row line point) / N, S (column line point): LDS offsets and existence of the 2nd .. 4th
        const int o0 = -(oy ? LW : 0) - (ox ? 1 : 0);
        const int o1 = (oy && ox) ? -LW + 1 : (ox ? 1 : LW);
        const bool e1 = (oy && ox) ? xr : (ox ? xr : yd);
        const bool e2 = oy && ox && yd, e3 = oy && ox && yd && xr;
        // (the four parent values of every right-hand side in ONE LDS round trip — an absent parent reads the first one's position and is not added; the conditional
        //  reads were up to three more dependent round trips per right-hand side)
        const bool copy = !oy && !ox;
        const int a1 = e1 ? o1 : o0, a2 = e2 ? LW - 1 : o0, a3 = e3 ? LW + 1 : o0;
        vf v0[NQ], v1[NQ], v2[NQ], v3[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { v0[q] = s_a[q * LN + p + (copy ? 0 : o0)]; v1[q] = s_a[q * LN + p + (copy ? 0 : a1)]; v2[q] = s_a[q * LN + p + (copy ? 0 : a2)]; v3[q] = s_a[q * LN + p + (copy ? 0 : a3)]; }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            vf e = pw[0] * v0[q];
            e = e1 ? e + pw[1] * v1[q] : e;
            e = e2 ? e + pw[2] * v2[q] : e;
            e = e3 ? e + pw[3] * v3[q] : e;
            e = copy ? v0[q] : e;
            xk[q] = xk[q] + e; s_b[q * LN + p] = xk[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MG_NS; ++k) {
        vf* src = (k & 1) ? s_a : s_b; vf* dst = (k & 1) ? s_b : s_a;
        if (ring(k + 1)) {
            vf y[NQ]; lds_op<NQ, LW, LN, NINE>(c, src, p, xk, y);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                xk[q] = xk[q] + (bq[q] - y[q]) * (k == 0 ? c.dinv : c.dinv * mg_rk(k));
                if (k == MG_NS - 1) xo[(size_t)q * L.n + i] = xk[q]; else dst[q * LN + p] = xk[q];
            }
        }
        if (k < MG_NS - 1) __syncthreads();
    }
}
// ---- Round 5 (NCT_S2_LINES=0 switches it off; oracle: orc_set_mg_lines, mg_block_step): a BLOCK STEP on the finest level — first thing of the pre-smoother (from zero: the residual
// is the right-hand side, no halo at all), last thing of the post-smoother. The grid is cut into fixed LBX x LBY blocks (aligned at 0, independent of the legs' tiles); inside a block
//     pre:  e1 = Lx^-1 r, e2 = Ly^-1 (Sy e1), x += OM (e1 + e2)          post (the adjoint):  e1 = Ly^-1 r, e2 = Lx^-1 (Sx e1), x += OM (e1 + e2)
// with Lx / Ly the tridiagonal matrices of the diagonal and the x / y couplings inside the block, Sy / Sx the y / x couplings inside the block: an alternating-direction solve of the
// block's own system. Point Jacobi cannot smooth along the runs of exactly flat neighbour pairs a photograph has next to its edges (couplings 10^4 apart); the line solves can
// (PCG iterations 19 -> 13 on the synthetic pair, 35 -> 19 / 44 -> 23 on photographs). Thomas factors: fp64 at set-up, rounded once (k_mg_lines_setup); the
// solves are fp32 in the oracle's order:  y(0) = r(0), y(i) = r(i) + lm(i) y(i-1);  e(last) = y(last) lp(last), e(i) = (y(i) + w(i, i+1) e(i+1)) lp(i).
constexpr int LBX = 32, LBY = 16, LBP = LBX + 1, LBN = LBY * LBP + 1;      // block, padded row pitch and plane size in LDS
constexpr float LINE_OM = 0.9f;
// lm(i) = w(i-1, i) / p(i-1) (0 at the start of a line), lp(i) = 1 / p(i), p(i) = d(i) - w(i-1, i) lm(i): one thread per line piece
__global__ void k_mg_lines_setup(Lvl L) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = L.W, H = L.H, nbx = (W + LBX - 1) / LBX, nby = (H + LBY - 1) / LBY;
    if (t < H * nbx) {
        const int r = t / nbx, c0 = (t - r * nbx) * LBX;
        double p = 0.0;
        for (int c = c0; c < W && c < c0 + LBX; ++c) {
            const int i = r * W + c;
            if (c == c0) { p = L.d[i]; L.lxm[i] = 0.0f; }
            else { const double w = L.wE[i - 1], m = w / p; p = L.d[i] - w * m; L.lxm[i] = (vf)m; }
            L.lxp[i] = (vf)(1.0 / p);
        }
    } else if (t < H * nbx + W * nby) {
        const int u = t - H * nbx, c = u / nby, r0 = (u - c * nby) * LBY;
        double p = 0.0;
        for (int r = r0; r < H && r < r0 + LBY; ++r) {
            const int i = r * W + c;
            if (r == r0) { p = L.d[i]; L.lym[i] = 0.0f; }
            else { const double w = L.wS[i - W], m = w / p; p = L.d[i] - w * m; L.lym[i] = (vf)m; }
            L.lyp[i] = (vf)(1.0 / p);
        }
    }
}
// one line of at most LEN points at LDS stride STR: right-hand side in s_r, result to s_e (may be the same array); factors / forward couplings in s_m, s_p, s_w at the same positions
template <int LEN, int STR>
__device__ __forceinline__ void line_solve(const vf* __restrict__ s_r, vf* __restrict__ s_e, const vf* __restrict__ s_m, const vf* __restrict__ s_p, const vf* __restrict__ s_w, int len) {
    vf y[LEN];
    if (len == LEN) {                                   // every block but the ones at the right / bottom edge: no predicates, so all LDS reads are issued ahead of the dependent chain
        constexpr int CH = 8;                             // factors are fetched CH at a time, ahead of the CH dependent steps that use them
        static_assert(LEN % CH == 0, "line length");
#pragma unroll
        for (int k = 0; k < LEN; ++k) y[k] = s_r[k * STR];
        vf t = 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < LEN; k0 += CH) {
            vf m[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) m[j] = s_m[(k0 + j) * STR];
#pragma unroll
            for (int j = 0; j < CH; ++j) { const int k = k0 + j; if (k == 0) t = y[0]; else { t = y[k] + m[j] * t; y[k] = t; } }
        }
#pragma unroll
        for (int k0 = LEN - CH; k0 >= 0; k0 -= CH) {
            vf pp[CH], ww[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) { pp[j] = s_p[(k0 + j) * STR]; ww[j] = s_w[(k0 + j) * STR]; }
#pragma unroll
            for (int j = CH - 1; j >= 0; --j) { const int k = k0 + j; t = k == LEN - 1 ? y[k] * pp[j] : (y[k] + ww[j] * t) * pp[j]; y[k] = t; }
        }
#pragma unroll
        for (int k = 0; k < LEN; ++k) s_e[k * STR] = y[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < LEN; ++k) if (k < len) y[k] = s_r[k * STR];
    vf t = y[0];
#pragma unroll
    for (int k = 1; k < LEN; ++k) if (k < len) { t = y[k] + s_m[k * STR] * t; y[k] = t; }
#pragma unroll
    for (int k = LEN - 1; k >= 0; --k)
        if (k < len) {
            if (k == len - 1) t = y[k] * s_p[k * STR];
            else t = (y[k] + s_w[k * STR] * t) * s_p[k * STR];
            s_e[k * STR] = t;
        }
}
template <int NQ, bool POST>
__global__ __launch_bounds__(LBX * LBY, 8) void k_mg_block(const PState* __restrict__ st, Lvl L, const vf* __restrict__ b, const vf* __restrict__ xin, vf* __restrict__ xout) {
    const int nact = st->nactive;   // checked below, behind the kernel's first loads: as the first statement it was a dependent scalar round trip in front of everything (round 6)
    constexpr int XW = LBX + 2, XN = XW * (LBY + 2);
    static_assert(NQ * XN <= (NQ + 6) * LBN, "the halo tile of the iterate lives where e1 and the factors go afterwards");
    __shared__ vf s_all[(2 * NQ + 6) * LBN];
    vf* const s_r = s_all; vf* const s_e1 = s_all + NQ * LBN; vf* const s_x = s_e1;          // s_x (post only): dead once the residual is in registers
    vf* const s_xm = s_all + 2 * NQ * LBN; vf* const s_xp = s_xm + LBN; vf* const s_xw = s_xp + LBN; vf* const s_ym = s_xw + LBN; vf* const s_yp = s_ym + LBN; vf* const s_yw = s_yp + LBN;
    const int W = L.W, H = L.H, nbx = (W + LBX - 1) / LBX;
    const int blk = mg_tile_of_block(blockIdx.x, gridDim.x);
    const int x0 = (blk % nbx) * LBX, y0 = (blk / nbx) * LBY;
    const int bw = x0 + LBX <= W ? LBX : W - x0, bh = y0 + LBY <= H ? LBY : H - y0;
    const int p = threadIdx.x, ly = p / LBX, lx = p - ly * LBX, gy = y0 + ly, gx = x0 + lx;
    const bool valid = ly < bh && lx < bw;
    const int i = gy * W + gx, lp = ly * LBP + lx;
    vf own[NQ], bq[NQ];
    if (POST) {                                                                  // the iterate with one ring of halo (whatever block owns it)
        for (int h = p; h < XN; h += LBX * LBY) {
            const int hy = h / XW, hx = h - hy * XW, yy = y0 + hy - 1, xx = x0 + hx - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) s_x[q * XN + h] = xin[(size_t)q * L.n + yy * W + xx];
            }
        }
    }
    vf fE_own = 0.f, fS_own = 0.f, c_xm = 0.f, c_xp = 0.f, c_ym = 0.f, c_yp = 0.f;
    if (valid) {
        c_xm = L.lxm[i]; c_xp = L.lxp[i]; c_ym = L.lym[i]; c_yp = L.lyp[i]; fE_own = L.fE[i]; fS_own = L.fS[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) bq[q] = b[(size_t)q * L.n + i];
    }
    if (nact == 0) return;
    if (POST) {
        __syncthreads();
        if (valid) {                                                             // residual, the stencil in lds_op's order E, W, S, N (level 0 is 5-point)
            const vf d = L.fd[i];
            const bool xr = gx + 1 < W, xl = gx > 0, yd = gy + 1 < H, yu = gy > 0;
            const vf wW = xl ? L.fE[i - 1] : 0.f, wN = yu ? L.fS[i - W] : 0.f;
            const int hp = (ly + 1) * XW + lx + 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                own[q] = s_x[q * XN + hp];
                const vf vE = s_x[q * XN + hp + 1], vW = s_x[q * XN + hp - 1], vS = s_x[q * XN + hp + XW], vN = s_x[q * XN + hp - XW];   // all inside the halo tile; read together (lesson (x))
                vf y = d * own[q];
                y = xr ? y - fE_own * vE : y;
                y = xl ? y - wW * vW : y;
                y = yd ? y - fS_own * vS : y;
                y = yu ? y - wN * vN : y;
                bq[q] = bq[q] - y;
            }
        }
        __syncthreads();
    }
    if (valid) {
        s_xm[lp] = c_xm; s_xp[lp] = c_xp; s_ym[lp] = c_ym; s_yp[lp] = c_yp; s_xw[lp] = fE_own; s_yw[lp] = fS_own;
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_r[q * LBN + lp] = bq[q];
    }
    __syncthreads();
    // stage 0: x lines (pre) / y lines (post)
    if (!POST) { if (p < LBY * NQ) { const int row = p % LBY, q = p / LBY; if (row < bh) { const int o = row * LBP; line_solve<LBX, 1>(s_r + q * LBN + o, s_e1 + q * LBN + o, s_xm + o, s_xp + o, s_xw + o, bw); } } }
    else       { if (p < LBX * NQ) { const int col = p % LBX, q = p / LBX; if (col < bw) line_solve<LBY, LBP>(s_r + q * LBN + col, s_e1 + q * LBN + col, s_ym + col, s_yp + col, s_yw + col, bh); } }
    __syncthreads();
    // what the first stage's lines left out, inside the block: forward neighbour, then backward
    if (valid) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            vf acc = 0.0f;
            if (!POST) { if (ly + 1 < bh) acc += fS_own * s_e1[q * LBN + lp + LBP]; if (ly > 0) acc += s_yw[lp - LBP] * s_e1[q * LBN + lp - LBP]; }
            else       { if (lx + 1 < bw) acc += fE_own * s_e1[q * LBN + lp + 1];   if (lx > 0) acc += s_xw[lp - 1] * s_e1[q * LBN + lp - 1]; }
            s_r[q * LBN + lp] = acc;
        }
    }
    __syncthreads();
    // stage 1, in place in s_r
    if (!POST) { if (p < LBX * NQ) { const int col = p % LBX, q = p / LBX; if (col < bw) line_solve<LBY, LBP>(s_r + q * LBN + col, s_r + q * LBN + col, s_ym + col, s_yp + col, s_yw + col, bh); } }
    else       { if (p < LBY * NQ) { const int row = p % LBY, q = p / LBY; if (row < bh) { const int o = row * LBP; line_solve<LBX, 1>(s_r + q * LBN + o, s_r + q * LBN + o, s_xm + o, s_xp + o, s_xw + o, bw); } } }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const vf u = (s_e1[q * LBN + lp] + s_r[q * LBN + lp]) * LINE_OM;
            xout[(size_t)q * L.n + i] = POST ? own[q] + u : u;
        }
    }
}

// Middle + tail of the V-cycle in ONE launch, one 1024-thread workgroup PER RIGHT-HAND SIDE: the first fused level has <= 4096 pixels
// (44x44 at 700x700, 63x63 at 1000x1000), the deeper ones <= 1024 (22x22, 11x11, 6x6). The six systems share the operator but not a single
// value, so no workgroup ever waits for another and the whole sub-cycle needs only __syncthreads(). Everything a level needs for the way
// back up (right-hand side, pre-smoothed iterate, stencil coefficients of the deeper levels) stays in REGISTERS of the thread that owns
// the pixel; iterates are exchanged through whole-grid LDS arrays (no halos); coefficients and transfer weights come from global memory once
// per level (the first fused level re-reads its coefficients for the up leg) and those loads are issued at the start of the level. (A version
// that re-read coefficients and iterates from global memory in each of its barrier-separated phases was SLOWER than the launches it replaced
// — DESIGN.md §9 — a single workgroup has nothing to hide a global round trip with.)
// Same expressions and operation order as k_mg_down / k_mg_up (stencil E, W, S, N, SE, SW, NE, NW; restriction over the 3x3 block row-major;
// prolongation over the parents NW, NE, SW, SE), so the cycle is bit-identical to the tile-fused launches it replaces. lv[0] is the first fused
// level: its rhs lv[0].b was written by the restriction above it, its correction goes to lv[0].x2. The coarsest grid (n <= 64) is solved by
// `sweeps` damped-Jacobi sweeps from zero by one wave (one lane per unknown, the iterate in a register, neighbours through ds_bpermute).
// P0 = pixels per thread of the first fused level: 1 or 2 (<= 2048 pixels: 44x44 at 700x700; 63x63 at 1000x1000 stays on tile launches — four pixels per thread measured slower)
#ifndef NCT_MID_MAXP0
#define NCT_MID_MAXP0 2      // largest first fused level, in units of 1024 pixels (1 or 2): 2 = the 44x44 level of a 700x700 pair rides in k_mg_mid (same time as its two tile launches, 186 launches fewer per pair)
#endif
static_assert(NCT_MID_MAXP0 == 1 || NCT_MID_MAXP0 == 2, "k_mg_mid is instantiated for one or two pixels per thread at its first level (four measured slower: DESIGN.md 9)");
constexpr int MID_T = 1024, MID_P1 = 1, MID_N1 = MID_T * MID_P1, MID_LV = 5;
// largest level at depth d >= 1 (the first fused level: P0 * 1024); levels shrink ~4x per depth. The levels below the first park their 10 coefficients and 4 prolongation
// weights per pixel in LDS for the way back up (a workgroup has the CU to itself: 79 KB of the 160 KB), so only right-hand side and iterate stay in registers across the recursion
constexpr int mid_cap(int d) { return d == 1 ? 1024 : (d == 2 ? 256 : 64); }
constexpr int mid_stash_off(int d) { return d <= 1 ? 0 : mid_stash_off(d - 1) + 14 * mid_cap(d - 1); }
constexpr int MID_STASH = mid_stash_off(MID_LV);
constexpr int mid_ppt(int P0, int D) { return D == 0 ? P0 : (D == 1 ? MID_P1 : 1); }      // pixels per thread at depth D of the fused sub-cycle
struct MidPack { Lvl lv[MID_LV]; int nl; };
__device__ __forceinline__ vf mid_op(const PxCoef& c, const vf* __restrict__ s_v, int i, int W, vf* centre = nullptr) {
    const vf v0 = s_v[i];
    if (centre) *centre = v0;
    vf y = c.d * v0;
    // All eight neighbours in ONE LDS round trip: an absent one (image border) reads the pixel itself and contributes w * 0 with w = +0 (px_coef): y - (+0) = y for every y, -0
    // included — the bits of the skipped term. Written as `if (exists) y -= w * s_v[..]` every coupling was a branch of its own with its own LDS round trip: eight serialised
    // round trips per application, ~24 applications per call (round 6, from the ISA: k_mg_mid 36.5 -> 31.3 us).
    vf t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = s_v[(c.ex & (1u << k)) ? i + nb_dy(k) * W + nb_dx(k) : i];
#pragma unroll
    for (int k = 0; k < 8; ++k) y -= c.w[k] * ((c.ex & (1u << k)) ? t[k] : 0.0f);
    return y;
}
// restricted residual of coarse pixel I of the grid (Wc wide) below the fine grid (Wf x Hf): sum over the 3x3 block around fine point (2Y, 2X), row-major
__device__ __forceinline__ vf mid_restrict(const vf* __restrict__ s_res, const vf (&ps)[9], int I, int Wc, int Wf, int Hf) {
    const int Y = I / Wc, X = I - Y * Wc, r = 2 * Y, c = 2 * X;
    vf acc = 0.0f, t[9]; bool in[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {                            // nine reads in one LDS round trip (an absent fine point reads the centre and is not added)
        const int yy = r + k / 3 - 1, xx = c + k % 3 - 1;
        in[k] = yy >= 0 && yy < Hf && xx >= 0 && xx < Wf;
        t[k] = s_res[in[k] ? yy * Wf + xx : r * Wf + c];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) acc = in[k] ? acc + ps[k] * t[k] : acc;
    return acc;
}
// P e at fine pixel (gy, gx) from the child's correction s_c (Wc x Hc): the existing parents in the order NW, NE, SW, SE (W, E / N, S on a line)
__device__ __forceinline__ vf mid_prolong(const vf* __restrict__ s_c, const vf (&pw)[4], int gy, int gx, int Wc, int Hc) {
    const int Y0 = gy >> 1, X0 = gx >> 1;
    const bool oy = (gy & 1) != 0, ox = (gx & 1) != 0;
    const bool xr = X0 + 1 < Wc, yd = Y0 + 1 < Hc;
    // the (up to) four parents in ONE LDS round trip; an absent one reads the first parent and is not added. Terms and order as before:
    // centre: NW + NE + SW + SE; row line point (ox only): W + E; column line point (oy only): N + S
    const int i00 = Y0 * Wc + X0, i01 = xr ? i00 + 1 : i00, i10 = yd ? i00 + Wc : i00, i11 = (yd && xr) ? i00 + Wc + 1 : i00;
    const vf c00 = s_c[i00], c01 = s_c[i01], c10 = s_c[i10], c11 = s_c[i11];
    if (!oy && !ox) return c00;
    vf e = pw[0] * c00;
    if (oy && ox) {
        e = xr ? e + pw[1] * c01 : e;
        e = yd ? e + pw[2] * c10 : e;
        e = (yd && xr) ? e + pw[3] * c11 : e;
    } else if (ox) { e = xr ? e + pw[1] * c01 : e; }
    else { e = yd ? e + pw[1] * c10 : e; }
    return e;
}
// Level D of the fused sub-cycle. In: this level's right-hand side b[] in registers (pixel i = t + k * MID_T). Out: this level's correction
// in sC (LDS, n values) — or in global L.x2 for D == 0. sA / sB: exchange arrays (>= n); sC: the child's correction (n_child values), then this level's own.
template <int P0, int D>
__device__ __forceinline__ void mid_level(const MidPack& P, int q, int t, vf* __restrict__ sA, vf* __restrict__ sB, vf* __restrict__ sC, vf* __restrict__ sS,
                                          const vf (&breg)[mid_ppt(P0, D)], int sweeps) {
    constexpr int PPT = mid_ppt(P0, D);
    const Lvl& L = P.lv[D];
    const int n = L.n, W = L.W;
    if (D == P.nl - 1) {
        // ---- coarsest grid: wave 0, one lane per unknown
        if (t < n) sA[t] = breg[0];
        __syncthreads();
        if (t < 64) {
            const int i = t;
            const bool live = i < n;
            const int r = live ? i / W : 0, c = live ? i - r * W : 0;
            vf bq = 0.f, d = 0.f, dv = 0.f, w[8]; unsigned ex = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = 0.f;
            if (live) { const PxCoef pc = px_coef<true>(L, r, c); d = pc.d; dv = pc.dinv; ex = pc.ex; bq = sA[i];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = pc.w[k]; }
            vf x = 0.0f;
            for (int s = 0; s < sweeps; ++s) {
                vf y = d * x;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const vf xn = __shfl(x, (i + nb_dy(k) * W + nb_dx(k)) & 63);
                    if (ex & (1u << k)) y -= w[k] * xn;
                }
                if (live) x = x + (bq - y) * (dv * MG_R0);
            }
            if (live) { if (D == 0) L.x2[(size_t)q * n + i] = x; else sC[i] = x; }
        }
        __syncthreads();
        return;
    }
    if constexpr (D + 1 < MID_LV) {
        const Lvl& C = P.lv[D + 1];
        constexpr int CPT = mid_ppt(P0, D + 1);
        vf* __restrict__ stash = sS + mid_stash_off(D);       // D >= 1: this level's coefficients and prolongation weights, planar [14][n], parked for the way back up
        vf x[PPT];
        {
            PxCoef c[PPT]; vf ps[CPT][9];
            // ---- all loads of the level first: coefficients, the P columns of the coarse pixels this thread restricts to, (D >= 1) the prolongation weights
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) c[k] = px_coef<true>(L, i / W, i % W); }
#pragma unroll
            for (int k = 0; k < CPT; ++k) { const int I = t + k * MID_T; if (I < C.n) {
#pragma unroll
                for (int j = 0; j < 9; ++j) ps[k][j] = L.fpst[(size_t)I * 9 + j]; } }
            if constexpr (D >= 1) {
                if (t < n) {
                    vf pw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pw[j] = L.fpw[(size_t)j * n + t];
                    stash[t] = c[0].d; stash[n + t] = c[0].dinv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) stash[(2 + j) * n + t] = c[0].w[j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) stash[(10 + j) * n + t] = pw[j];
                }
            }
            // ---- down: MG_NS sweeps from zero (x_1 = b*dinv ; x_{s+1} = x_s + (b - M x_s)*dinv*rk(s)), iterates in ping-pong through sA / sB ; res = b - M x
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) sA[i] = breg[k] * c[k].dinv; }
            __syncthreads();
#pragma unroll
            for (int sw = 1; sw < MG_NS; ++sw) {
                vf* src = (sw & 1) ? sA : sB; vf* dst = (sw & 1) ? sB : sA;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int i = t + k * MID_T;
                    if (i < n) {
                        vf xs; const vf y = mid_op(c[k], src, i, W, &xs);
                        const vf xv = xs + (breg[k] - y) * (c[k].dinv * mg_rk(sw));
                        dst[i] = xv;
                        if (sw == MG_NS - 1) x[k] = xv;
                    }
                }
                __syncthreads();
            }
            {
                vf* xs = (MG_NS & 1) ? sA : sB; vf* rs = (MG_NS & 1) ? sB : sA;
#pragma unroll
                for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) rs[i] = breg[k] - mid_op(c[k], xs, i, W); }
            }
            __syncthreads();
            vf bc[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) { const int I = t + k * MID_T; bc[k] = I < C.n ? mid_restrict((MG_NS & 1) ? sB : sA, ps[k], I, C.W, W, L.H) : 0.f; }
            __syncthreads();                                      // the residual array is free again
            asm volatile("" ::: "memory");                        // nothing of this level but b and x stays in registers across the deeper levels
            mid_level<P0, D + 1>(P, q, t, sA, sB, sC, sS, bc, sweeps);    // its correction arrives in sC
        }
        // ---- up: xe = x + P e_child ; MG_NS more sweeps. Coefficients and weights come back from the stash (D >= 1) or from global memory again (D == 0)
        PxCoef c[PPT]; vf pw[PPT][4];
        if constexpr (D == 0) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) { c[k] = px_coef<true>(L, i / W, i % W);
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[k][j] = L.fpw[(size_t)j * n + i]; } }
        } else {
            if (t < n) {
                const int gy = t / W, gx = t - gy * W;
                const bool xr = gx + 1 < W, xl = gx > 0, yd = gy + 1 < L.H, yu = gy > 0;
                c[0].ex = (xr ? 1u : 0u) | (xl ? 2u : 0u) | (yd ? 4u : 0u) | (yu ? 8u : 0u) | (xr && yd ? 16u : 0u) | (xl && yd ? 32u : 0u) | (xr && yu ? 64u : 0u) | (xl && yu ? 128u : 0u);
                c[0].d = stash[t]; c[0].dinv = stash[n + t];
#pragma unroll
                for (int j = 0; j < 8; ++j) c[0].w[j] = stash[(2 + j) * n + t];
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[0][j] = stash[(10 + j) * n + t];
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = t + k * MID_T;
            if (i < n) sA[i] = x[k] + mid_prolong(sC, pw[k], i / W, i % W, C.W, C.H);
        }
        __syncthreads();                                      // (from here on every thread has finished reading the child's correction in sC)
#pragma unroll
        for (int sw = 0; sw < MG_NS; ++sw) {
            vf* src = (sw & 1) ? sB : sA; vf* dst = (sw & 1) ? sA : sB;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = t + k * MID_T;
                if (i < n) {
                    vf xs; const vf y = mid_op(c[k], src, i, W, &xs);
                    const vf v = xs + (breg[k] - y) * (sw == 0 ? c[k].dinv : c[k].dinv * mg_rk(sw));
                    if (sw < MG_NS - 1) dst[i] = v;
                    else if (D == 0) L.x2[(size_t)q * n + i] = v; else sC[i] = v;
                }
            }
            __syncthreads();
        }
    }
}
template <int P0>
__global__ __launch_bounds__(MID_T) void k_mg_mid(const PState* __restrict__ st, MidPack P, int sweeps) {
    const int nact = st->nactive;   // checked below, behind the kernel's first loads: as the first statement it was a dependent scalar round trip in front of everything (round 6)
    __shared__ vf sA[MID_T * P0], sB[MID_T * P0], sC[MID_N1], sS[MID_STASH];
    const int q = blockIdx.x, t = threadIdx.x;
    const Lvl& L0 = P.lv[0];
    vf b[P0];
#pragma unroll
    for (int k = 0; k < P0; ++k) { const int i = t + k * MID_T; b[k] = i < L0.n ? L0.b[(size_t)q * L0.n + i] : 0.f; }
    if (nact == 0) return;
    mid_level<P0, 0>(P, q, t, sA, sB, sC, sS, b, sweeps);
}

// ---- PCG pieces at the fine level

// x6 = interleave(X); r = rough*x0 - M x0 ; partial: rr, bb (12)
// The solver state the host polls is written by the kernel that produces it, straight into page-locked host memory (fine-grained, device-visible): fields, a
// system-scope fence, then the publication number the host spins on. No copy command and no event sit in the stream between two iterations (an in-order
// stream would hold the next kernel back until the copy engine has finished: a bubble per poll).
__device__ __forceinline__ void pcg_publish(PState* __restrict__ host_out, const PState& v, int seq) {
    PState o = v; o.seq = 0;
    *host_out = o;
    __threadfence_system();
    __hip_atomic_store(&host_out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int NQ>
__global__ __launch_bounds__(256) void k_pcg_start(Lvl L, const double* __restrict__ rough, const double* __restrict__ X /*[2][n][3]*/, double* __restrict__ x6, double* __restrict__ r, vf* __restrict__ rf, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc[2 * NQ];
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) acc[q] = 0.0;
    if (i < L.n) {
        auto xv = [&](int j, int q) { return X[((size_t)(q / 3) * L.n + j) * 3 + (q % 3)]; };
        double y[NQ]; lvl_op<NQ>(L, i, xv, y);
        const double rg = rough[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double x0 = xv(i, q);
            const double bq = rg * x0;
            const double rv = bq - y[q];
            x6[(size_t)q * L.n + i] = x0; r[(size_t)q * L.n + i] = rv; rf[(size_t)q * L.n + i] = (vf)rv;
            acc[q] = rv * rv; acc[NQ + q] = bq * bq;
        }
    }
    mg_block_reduce<2 * NQ>(acc, partial);
}
template <int NQ>
__global__ void k_pcg_start_fin(const double* __restrict__ partial, int nb, PState* __restrict__ st, double rtol2, PState* __restrict__ host_out, int seq) {
    double s[2 * NQ]; mg_final_reduce<2 * NQ>(partial, nb, s);
    if (threadIdx.x < NQ) { const int q = threadIdx.x; st->bb[q] = s[NQ + q]; st->gam[q] = 0; st->alp[q] = 0; st->iters[q] = 0; st->rho[q] = s[q];
                            st->active[q] = (s[q] > rtol2 * s[NQ + q]) ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) { int na = 0; for (int q = 0; q < NQ; ++q) na += st->active[q]; st->nactive = na; if (host_out) pcg_publish(host_out, *st, seq); }
}
// Single-reduction PCG (Chronopoulos & Gear): per iteration  u = M^-1 r (the V-cycle, fp32) ; w = A u ; gamma = r.u, delta = w.u,
// rho = r.r in ONE reduction ; beta = gamma/gamma_old, alpha = gamma / (delta - beta*gamma/alpha_old) ; p = u + beta p ; s = w + beta s
// (= A p by recurrence) ; x += alpha p ; r -= alpha s. Same iterates as textbook PCG in exact arithmetic and the same iteration counts
// in practice (scripts/cgcg_check.py), with 3 launches and one reduction per iteration instead of 7 and three.
// w = A u with u = the V-cycle output widened exactly; partial sums of gamma, delta, rho (18 per 256-pixel block)
template <int NQ>
__global__ __launch_bounds__(256) void k_cg_apply(const PState* __restrict__ st, Lvl L, const vf* __restrict__ z, const double* __restrict__ r,
                                                  double* __restrict__ w, double* __restrict__ partial) {
    const int nact = st->nactive;   // checked below, behind the kernel's first loads: as the first statement it was a dependent scalar round trip in front of everything (round 6)
    // XCD-aware block order (as the V-cycle legs): each XCD takes a contiguous eighth of the image, so the rows above and below a block's pixels are in its own L2
    // (109 MB of fabric traffic per launch for 70 MB of compulsory bytes before). The partial sums keep their LOGICAL block slot: the reduction order is unchanged.
    const int lb = mg_tile_of_block(blockIdx.x, gridDim.x);
    const int i = lb * 256 + threadIdx.x;
    double acc[3 * NQ];
#pragma unroll
    for (int q = 0; q < 3 * NQ; ++q) acc[q] = 0.0;
    auto uv = [&](int j, int q) { return (double)z[(size_t)q * L.n + j]; };
    double y[NQ], rq[NQ], uq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { y[q] = 0.0; rq[q] = 0.0; uq[q] = 0.0; }
    if (i < L.n) {
        lvl_op<NQ>(L, i, uv, y);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { rq[q] = r[(size_t)q * L.n + i]; uq[q] = uv(i, q); }
    }
    if (nact == 0) return;                                   // (every thread; before the first store)
    if (i < L.n) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double u = uq[q], rv = rq[q];
            w[(size_t)q * L.n + i] = y[q];
            acc[q] = rv * u; acc[NQ + q] = y[q] * u; acc[2 * NQ + q] = rv * rv;
        }
    }
    mg_block_reduce<3 * NQ>(acc, partial, lb);
}
// three workgroups: workgroup j reduces gamma (0), delta (1), rho (2) of all 6 systems in the fixed order
template <int NQ>
__global__ void k_cg_fin(const PState* __restrict__ st, const double* __restrict__ partial, int nb, double* __restrict__ sums) {
    if (st->nactive == 0) return;
    double s[NQ]; mg_final_reduce_strided<NQ>(partial + blockIdx.x * NQ, nb, 3 * NQ, s);
    if (threadIdx.x < NQ) sums[blockIdx.x * NQ + threadIdx.x] = s[threadIdx.x];
}
// scalars + all four vector recurrences; every workgroup derives the scalars itself, workgroup 0 publishes the next state
template <int NQ>
__global__ __launch_bounds__(256) void k_cg_update(int n, const PState* __restrict__ sc, PState* __restrict__ sn, const double* __restrict__ sums, double rtol2, int first,
                                                   const vf* __restrict__ z, const double* __restrict__ w, double* __restrict__ p, double* __restrict__ s,
                                                   double* __restrict__ x, double* __restrict__ r, vf* __restrict__ rf, PState* __restrict__ host_out, int seq) {
    if (sc->nactive == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) { *sn = *sc; if (host_out) pcg_publish(host_out, *sc, seq); } return; }
    double al[NQ], be[NQ]; bool act[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double gam = sums[q], del = sums[NQ + q], rho = sums[2 * NQ + q];
        act[q] = sc->active[q] != 0 && rho > rtol2 * sc->bb[q];
        be[q] = first ? 0.0 : gam / sc->gam[q];
        al[q] = first ? gam / del : gam / (del - be[q] * gam / sc->alp[q]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int na = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            sn->bb[q] = sc->bb[q];
            sn->gam[q] = act[q] ? sums[q] : sc->gam[q]; sn->alp[q] = act[q] ? al[q] : sc->alp[q];
            sn->iters[q] = sc->iters[q] + (act[q] ? 1 : 0); sn->active[q] = act[q] ? 1 : 0; na += act[q] ? 1 : 0;
            sn->rho[q] = sc->active[q] != 0 ? sums[2 * NQ + q] : sc->rho[q];
        }
        sn->nactive = na;
        if (host_out) pcg_publish(host_out, *sn, seq);
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // the operands of all active systems are requested first (36 loads in flight per thread instead of six per system in turn: the kernel is a pure stream), then the updates
    vf zq[NQ]; double wq[NQ], pq[NQ], sq[NQ], xq[NQ], rq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        zq[q] = 0.f; wq[q] = pq[q] = sq[q] = xq[q] = rq[q] = 0.0;
        if (act[q]) {
            const size_t j = (size_t)q * n + i;
            zq[q] = z[j]; wq[q] = w[j]; xq[q] = x[j]; rq[q] = r[j];
            if (!first) { pq[q] = p[j]; sq[q] = s[j]; }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!act[q]) continue;
        const size_t j = (size_t)q * n + i;
        const double zv = (double)zq[q], wv = wq[q];
        const double pn = first ? zv : zv + be[q] * pq[q];
        const double sv = first ? wv : wv + be[q] * sq[q];
        p[j] = pn; s[j] = sv;
        x[j] = xq[q] + al[q] * pn;
        const double rn = rq[q] - al[q] * sv;
        r[j] = rn; rf[j] = (vf)rn;                       // the V-cycle reads the residual rounded to fp32 (four kernels on the finest level): rounded once here, half the bytes there
    }
}
template <int NQ>
__global__ void k_pcg_finish(int n, const double* __restrict__ x6, double* __restrict__ X) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * NQ) return;
    const int q = i / n, px = i - q * n;
    X[((size_t)(q / 3) * n + px) * 3 + (q % 3)] = x6[i];
}
}  // namespace

#define LCHK() NCT_LAUNCH_CHECK()
#ifndef NCT_WLS_BATCH
#define NCT_WLS_BATCH 2      // iterations enqueued between two convergence polls (even: the state double buffer); 4: +1.0 ms of empty launches past convergence per pair, 6: +1.3
#endif

namespace {
// One PCG solve over NQ right-hand sides on its own stream. Everything it needs was allocated by the caller (the arena is not thread safe);
// `ctx` here is only an error sink, so that the second half of a split solve can run on a helper thread.
struct ErrSink {
    std::string err;
    int fail(int code, const char* fmt, ...) { char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); err = buf; return code; }
};
static std::atomic<int> g_seq{0};                         // publication numbers of the host-visible solver states: unique per process, so a slot never shows a stale match
struct PartBufs {
    double *x6, *r, *p, *sv, *w, *partial, *sums; PState* st;      // Krylov vectors [NQ][N], reduction scratch, double-buffered state
    vf* rf;                                                         // the residual rounded to fp32: what the V-cycle's finest level reads (written next to r by k_pcg_start / k_cg_update)
    std::vector<Lvl> lv;                                            // the shared operator hierarchy with THIS part's V-cycle vectors (b, x, x2)
    PState* hst; hipEvent_t ev[2];                                  // two page-locked read-back slots and their events
    const double* rough;                                            // the data term (right-hand side = rough * x0)
    nct_ctx* kt;                                                    // kernel clock (NCT_FLAG_TIME_KERNELS, unsplit solves only), else null
    int maxit, graph;
    bool trace;
    bool forecast;                                                  // size the batches by the convergence forecast (pcg_part)
    bool lines;                                                     // block step on the finest level (default; NCT_S2_LINES=0: without)
    int iters[NQMAX];
};
template <int NQ>
int pcg_part(ErrSink* ctx, hipStream_t s, double* X /* this part's [N][3] block(s) */, PartBufs& B, double rtol) {
    const std::vector<Lvl>& lv = B.lv;
    const int nl = (int)lv.size();
    const Lvl& F = lv[0];
    const int N = F.n, nb = cdiv(N, 256);
    double *x6 = B.x6, *r = B.r, *p = B.p, *sv = B.sv, *w = B.w, *partial = B.partial, *sums = B.sums;
    vf* rf = B.rf;
    PState* st = B.st;                                 // st[0] / st[1]; `cur` = the state the iteration being enqueued reads
    const PState* cur = st;
    const double rtol2 = rtol * rtol;
    hipLaunchKernelGGL(k_pcg_start<NQ>, dim3(nb), dim3(256), 0, s, F, B.rough, (const double*)X, (double*)x6, (double*)r, rf, (double*)partial); LCHK();
    const bool zero_copy = !B.graph;                       // the graph hook replays fixed kernel arguments: it keeps the copy + event form
    int seq_of_slot[2] = {0, 0};
    seq_of_slot[0] = ++g_seq;
    hipLaunchKernelGGL(k_pcg_start_fin<NQ>, dim3(1), dim3(256), 0, s, (const double*)partial, nb, st, rtol2, zero_copy ? &B.hst[0] : (PState*)nullptr, seq_of_slot[0]); LCHK();

    // z = Vcycle(r). Levels 0..nl-2 run the tile-fused down/up legs (2 launches per level), the coarsest grid one wave per right-hand side.
    // lv[l].x2 = where level l's correction ends up (the up leg cannot write in place: neighbouring tiles still read lv[l].x).
    // Tile shapes: bandwidth-bound levels use TXB x TYB tiles; below 100k pixels the legs are latency bound, so a 16x8 tile keeps the
    // dependent load chains short and spreads over more CUs.
    constexpr int TXB = NCT_MG_TXB, TYB = NCT_MG_TYB;
    auto down = [&](int l) {
        const dim3 gb(cdiv(lv[l].W, TXB) * cdiv(lv[l].H, TYB)), gs(cdiv(lv[l].W, 16) * cdiv(lv[l].H, 8));      // 1-D: mg_tile_of_block maps block -> tile
        if (l == 0 && B.lines) {
            // the leg from the block step's iterate (block_pre: x2, free until the up leg): one more halo pixel per side, so 32 x 14 tiles (41 x 23 = 943 threads)
            constexpr int TYL = TYB - 2;
            const dim3 gbl(cdiv(lv[0].W, TXB) * cdiv(lv[0].H, TYL));
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_down<NQ, TXB, TYL, vf, false, true>), gbl, dim3(mg_threads(TXB, TYL, true)), 0, s, cur, lv[l], (const vf*)rf, lv[l].x, lv[l + 1], lv[l + 1].b, (const vf*)lv[0].x2);
            else                   hipLaunchKernelGGL((k_mg_down<NQ, 16, 8, vf, false, true>), gs, dim3(mg_threads(16, 8, true)), 0, s, cur, lv[l], (const vf*)rf, lv[l].x, lv[l + 1], lv[l + 1].b, (const vf*)lv[0].x2);
        } else if (l == 0) {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_down<NQ, TXB, TYB, vf, false>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const vf*)rf, lv[l].x, lv[l + 1], lv[l + 1].b);
            else                   hipLaunchKernelGGL((k_mg_down<NQ, 16, 8, vf, false>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const vf*)rf, lv[l].x, lv[l + 1], lv[l + 1].b);
        } else {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_down<NQ, TXB, TYB, vf, true>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const vf*)lv[l].b, lv[l].x, lv[l + 1], lv[l + 1].b);
            else                   hipLaunchKernelGGL((k_mg_down<NQ, 16, 8, vf, true>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const vf*)lv[l].b, lv[l].x, lv[l + 1], lv[l + 1].b);
        }
    };
    // the block step (finest level): from zero into x2 in front of the down leg; mirrored, x2 -> x (the pre-smoothed iterate there is dead by then), behind the up leg
    auto block_step = [&](bool post) {
        if (!B.lines) return;
        const dim3 gl(cdiv(lv[0].W, LBX) * cdiv(lv[0].H, LBY));
        if (!post) hipLaunchKernelGGL((k_mg_block<NQ, false>), gl, dim3(LBX * LBY), 0, s, cur, lv[0], (const vf*)rf, (const vf*)nullptr, lv[0].x2);
        else       hipLaunchKernelGGL((k_mg_block<NQ, true>), gl, dim3(LBX * LBY), 0, s, cur, lv[0], (const vf*)rf, (const vf*)lv[0].x2, lv[0].x);
    };
    auto up = [&](int l, const vf* ec) {
        const dim3 gb(cdiv(lv[l].W, TXB) * cdiv(lv[l].H, TYB)), gs(cdiv(lv[l].W, 16) * cdiv(lv[l].H, 8));
        const int Wc = lv[l + 1].W, nc = lv[l + 1].n;
        if (l == 0) {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_up<NQ, TXB, TYB, vf, false>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const vf*)rf, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
            else                   hipLaunchKernelGGL((k_mg_up<NQ, 16, 8, vf, false>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const vf*)rf, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
        } else {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_up<NQ, TXB, TYB, vf, true>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const vf*)lv[l].b, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
            else                   hipLaunchKernelGGL((k_mg_up<NQ, 16, 8, vf, true>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const vf*)lv[l].b, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
        }
    };
    // tail0 = first level of the fused middle + tail (k_mg_mid): the deepest run of levels whose first has <= 4096 pixels and the others
    // <= MID_N1, at most MID_LV of them; never the fine level (its right-hand side is the fp64 PCG residual and its tiles fill the chip)
    int tail0 = nl - 1;
    auto mid_fits = [&](int first) {                     // levels first .. nl-1 as depths 0 .. of k_mg_mid
        if (nl - first > MID_LV) return false;
        for (int l = first; l < nl; ++l) { const int d = l - first; if (lv[l].n > (d == 0 ? NCT_MID_MAXP0 * MID_T : mid_cap(d))) return false; }
        return true;
    };
    while (tail0 > 1 && mid_fits(tail0 - 1)) --tail0;
    MidPack pack; memset(&pack, 0, sizeof pack); pack.nl = nl - tail0;
    for (int l = tail0; l < nl; ++l) pack.lv[l - tail0] = lv[l];
    bool ktime = false;                                   // this iteration's launches are timed one by one (NCT_FLAG_TIME_KERNELS)
    auto kt_b = [&](int id) -> int { if (ktime && B.kt->kt_begin(s, id)) return ctx->fail(NCT_ERR_HIP, "%s", B.kt->err.c_str()); return 0; };
    auto kt_e = [&]() -> int { if (ktime && B.kt->kt_end(s)) return ctx->fail(NCT_ERR_HIP, "%s", B.kt->err.c_str()); return 0; };
    auto vcycle = [&]() -> int {
        if (ktime) {
            int rc = 0;
            if (B.lines) { rc = kt_b(NCT_KT_WLS_BLOCK_PRE); if (rc) return rc; block_step(false); LCHK(); rc = kt_e(); if (rc) return rc; }
            rc = kt_b(NCT_KT_WLS_DOWN); if (rc) return rc; down(0); LCHK(); rc = kt_e(); if (rc) return rc;
            rc = kt_b(NCT_KT_WLS_COARSE); if (rc) return rc;
            for (int l = 1; l < tail0; ++l) { down(l); LCHK(); }
            if (lv[tail0].n <= MID_T) hipLaunchKernelGGL(k_mg_mid<1>, dim3(NQ), dim3(MID_T), 0, s, cur, pack, 60);
            else                      hipLaunchKernelGGL(k_mg_mid<2>, dim3(NQ), dim3(MID_T), 0, s, cur, pack, 60);
            LCHK();
            for (int l = tail0 - 1; l >= 1; --l) { up(l, lv[l + 1].x2); LCHK(); }
            rc = kt_e(); if (rc) return rc;
            rc = kt_b(NCT_KT_WLS_UP); if (rc) return rc; up(0, lv[1].x2); LCHK(); rc = kt_e(); if (rc) return rc;
            if (B.lines) { rc = kt_b(NCT_KT_WLS_BLOCK_POST); if (rc) return rc; block_step(true); LCHK(); rc = kt_e(); if (rc) return rc; }
            return 0;
        }
        block_step(false); LCHK();
        for (int l = 0; l < tail0; ++l) { down(l); LCHK(); }
        if (lv[tail0].n <= MID_T) hipLaunchKernelGGL(k_mg_mid<1>, dim3(NQ), dim3(MID_T), 0, s, cur, pack, 60);
        else                      hipLaunchKernelGGL(k_mg_mid<2>, dim3(NQ), dim3(MID_T), 0, s, cur, pack, 60);
        LCHK();
        for (int l = tail0 - 1; l >= 0; --l) { up(l, lv[l + 1].x2); LCHK(); }
        block_step(true); LCHK();
        return 0;
    };
    const vf* z = B.lines ? lv[0].x : lv[0].x2;
    // Convergence is polled without draining the stream: after every batch of `batch` iterations the solver state is copied to
    // page-locked host memory and an event is recorded; the host then enqueues the NEXT batch before it waits for that event, so
    // the GPU always has a batch queued. The batch enqueued past convergence costs only empty launches (nactive == 0).
    const int maxit = B.maxit, batch = NCT_WLS_BATCH;
    PState* hst = B.hst;                                  // two slots of page-locked memory
    PState* pub_to = nullptr; int pub_seq = 0;           // set for the LAST iteration of a batch: its update kernel publishes the state
    auto iteration = [&](int it) -> int {
        cur = st + (it & 1);
        PState* nxt = st + ((it + 1) & 1);
        ktime = B.kt != nullptr && it >= 2 && it < 6 && tail0 >= 1;
        int rc = vcycle(); if (rc) return rc;
        rc = kt_b(NCT_KT_WLS_APPLY); if (rc) return rc;
        hipLaunchKernelGGL(k_cg_apply<NQ>, dim3(nb), dim3(256), 0, s, cur, F, z, (const double*)r, (double*)w, (double*)partial); LCHK();
        rc = kt_e(); if (rc) return rc;
        hipLaunchKernelGGL(k_cg_fin<NQ>, dim3(3), dim3(256), 0, s, cur, (const double*)partial, nb, (double*)sums); LCHK();
        rc = kt_b(NCT_KT_WLS_UPDATE); if (rc) return rc;
        hipLaunchKernelGGL(k_cg_update<NQ>, dim3(nb), dim3(256), 0, s, N, cur, nxt, (const double*)sums, rtol2, it == 0 ? 1 : 0, z, (const double*)w,
                           (double*)p, (double*)sv, (double*)x6, (double*)r, rf, pub_to, pub_seq); LCHK();
        rc = kt_e(); if (rc) return rc;
        cur = nxt;
        return 0;
    };
    auto snapshot = [&](int slot) -> int {               // copy + event form (graph hook only)
        NCT_HIP(hipMemcpyAsync(&hst[slot], cur, sizeof(PState), hipMemcpyDeviceToHost, s));
        NCT_HIP(hipEventRecord(B.ev[slot], s));
        return 0;
    };
    auto wait_published = [&](int slot) -> int {           // spin on the publication number (the kernels of at least one batch are queued behind it)
        const int want_seq = seq_of_slot[slot];
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1; __atomic_load_n(&hst[slot].seq, __ATOMIC_ACQUIRE) != want_seq; ++spins) {
            if (spins > 4096) std::this_thread::yield();
            if ((spins & 0xFFFFu) == 0) {
                const hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady) return ctx->fail(NCT_ERR_HIP, "WLS MG-PCG: stream error while polling the solver state: %s", hipGetErrorString(q));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return ctx->fail(NCT_ERR_HIP, "WLS MG-PCG: the solver state was not published within 60 s");
            }
        }
        return 0;
    };
    // Snapshot k is taken behind the k-th enqueue (k = 0: behind the start kernel), into slot k & 1; the host reads them in order, one per turn, after it
    // has enqueued the next batch. How MANY iterations that batch gets is a forecast: from the last two snapshots it has read, the host knows r.r of every
    // system still iterating and its decay per iteration, hence the iterations still needed (the one that finds r.r below the threshold included); what is
    // already in flight is subtracted. A forecast of 0 enqueues nothing and just waits for the in-flight snapshot — if the solve is not done then (the decay
    // slowed down), the GPU idles for one host round trip (~20 us), about what ONE needless iteration costs; without the forecast every solve ran 2-4 of
    // those (22-44 empty launches: 19 iterations of 110 per 700x700 pair). The forecast never changes what an iteration computes.
    int it = 0; bool done = false;
    int issued = 0, seen = -1;                             // snapshot numbers
    int its_at_snapshot[2] = {0, 0};                       // iterations enqueued when snapshot (k & 1) was taken
    if (!zero_copy) { int rc = snapshot(0); if (rc) return rc; }   // state after the start kernel (x0 may already solve the system); zero-copy: k_pcg_start_fin published it
    PState fin; memset(&fin, 0, sizeof fin);
    double prev_rho[6]; int prev_its = -1;
    int remaining_after_seen = -1;                         // forecast; -1 = none yet
    int its_seen = 0;
    // Experiment hook (NCT_WLS_GRAPH=1): from the second batch on, the iteration batch (its kernels, arguments and the
    // state double-buffering repeat exactly) is captured once and replayed as a HIP graph instead of being enqueued kernel by kernel.
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    struct GraphCleanup { hipGraph_t& g; hipGraphExec_t& e; ~GraphCleanup() { if (e) (void)hipGraphExecDestroy(e); if (g) (void)hipGraphDestroy(g); } } gcleanup{graph, gexec};
    const bool forecast = B.forecast && !B.graph;
    bool forecast_on = forecast;
    while (true) {
        int want = batch;
        if (forecast_on && remaining_after_seen >= 0) {
            const int inflight = it - its_seen;
            want = remaining_after_seen - inflight;
            if (want > batch) want = batch;
            if (want < 0) want = 0;
            if (want == 0 && issued == seen) { want = batch; forecast_on = false; }   // nothing in flight and not converged: the forecast was short (the decay
                                                                                       // flattened) — the GPU just idled for a round trip; fixed batches for the rest of this solve
        }
        if (it + want > maxit) want = maxit - it;
        const bool enqueued = want > 0;
        if (enqueued && B.graph && it >= batch && want == batch) {
            if (!gexec) {
                NCT_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int rc = 0;
                for (int k = 0; k < batch && !rc; ++k) rc = iteration(it + k);
                hipError_t e = hipStreamEndCapture(s, &graph);
                if (rc) return rc;
                NCT_HIP(e);
                NCT_HIP(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
            }
            NCT_HIP(hipGraphLaunch(gexec, s));
            it += batch;
            cur = st + (it & 1);
        } else
        if (enqueued) {
            for (int k = 0; k < want; ++k, ++it) {
                if (zero_copy && k == want - 1) { seq_of_slot[(issued + 1) & 1] = ++g_seq; pub_to = &hst[(issued + 1) & 1]; pub_seq = seq_of_slot[(issued + 1) & 1]; }
                int rc = iteration(it); pub_to = nullptr; if (rc) return rc;
            }
        }
        if (enqueued) { ++issued; its_at_snapshot[issued & 1] = it; if (!zero_copy) { int rc = snapshot(issued & 1); if (rc) return rc; } }
        if (seen == issued) break;                         // nothing left to read: the iteration budget is spent
        ++seen;
        if (zero_copy) { int rc = wait_published(seen & 1); if (rc) return rc; } else NCT_HIP(hipEventSynchronize(B.ev[seen & 1]));
        fin = hst[seen & 1];
        its_seen = its_at_snapshot[seen & 1];
        if (fin.nactive == 0) { done = true; break; }
        if (it >= maxit && seen == issued) break;
        if (forecast) {
            // iterations still needed after this snapshot: the slowest system's m = ceil(log(threshold / rho) / log(decay per iteration)), decay from the
            // previous snapshot read; no forecast (full batches) while there is no history or the residual does not decay
            const int ridx = its_seen > 0 ? its_seen - 1 : 0;   // fin.rho = r_ridx . r_ridx (the start kernel and iteration 0 both see r_0)
            int need = -1; bool ok = prev_its >= 0 && ridx > prev_its;
            if (ok) {
                need = 1;
                for (int q = 0; q < NQ; ++q) {
                    if (!fin.active[q]) continue;
                    const double thr = rtol2 * fin.bb[q], rho = fin.rho[q];
                    if (!(rho > 0.0) || !(prev_rho[q] > 0.0) || !(thr > 0.0)) { ok = false; break; }
                    const double decay = pow(rho / prev_rho[q], 1.0 / (double)(ridx - prev_its));
                    if (!(decay < 0.95)) { ok = false; break; }
                    // fin.rho = r.r the LAST iteration saw, i.e. one update older than the residual now: the next iteration sees rho * decay
                    const double m = rho * decay <= thr ? 1.0 : 1.0 + ceil(log(thr / (rho * decay)) / log(decay));
                    if (m > need) need = (int)(m < 1e6 ? m : 1e6);
                }
            }
            remaining_after_seen = ok ? need : -1;
            for (int q = 0; q < NQ; ++q) prev_rho[q] = fin.rho[q];
            prev_its = ridx;
        }
    }
    // iterations enqueued after `fin` was taken leave the state untouched (nactive == 0), so fin is final
    if (!done) return ctx->fail(NCT_ERR_HIP, "WLS MG-PCG did not converge in %d iterations", maxit);
    hipLaunchKernelGGL(k_pcg_finish<NQ>, dim3(cdiv(N * NQ, 256)), dim3(256), 0, s, N, (const double*)x6, X); LCHK();
    for (int q = 0; q < NQ; ++q) B.iters[q] = fin.iters[q];
    if (B.trace) {                                          // NCT_WLS_TRACE=1: iterations enqueued against iterations that did something, snapshots read
        int mx = 0; for (int q = 0; q < NQ; ++q) mx = fin.iters[q] > mx ? fin.iters[q] : mx;
        fprintf(stderr, "nct wls: %d x %d, %d iterations enqueued, %d needed (+1 that finds every system converged), %d snapshots read\n", F.W, F.H, it, mx, seen + 1);
    }
    return 0;
}
}  // namespace

// X: [2][N][3] in (x0) / out. rough, wx, wy: fine-level data term and edge weights (wx[i] = edge (i,i+1), wy[i] = edge (i,i+W)).
// ctx->wls_split (NCT_FLAG_LATENCY): the a-half and the b-half (3 right-hand sides each, same operator, no shared value) are solved
// concurrently — the second on ctx->stream2 from a helper thread — so that one half's latency-bound coarse legs hide behind the other's
// bandwidth-bound 700x700 kernels. Every right-hand side sees exactly the same arithmetic as in the 6-wide solve (per-system reductions,
// scalars and convergence tests): the result is bit-identical. It doubles the launches, so it is for ONE pair in flight.
int nctk_wls_solve_mg(nct_ctx* ctx, hipStream_t s, double* X, const double* rough, const double* wx, const double* wy, int H, int W,
                      double rtol, int* iters_out /*host[6], nullable*/) {
    const bool split = ctx->wls_split != 0;
    const int nq0 = split ? 3 : 6;
    // ---- hierarchy
    std::vector<Lvl> lv;
    std::vector<void*> owned;
    struct Cleanup { nct_ctx* c; std::vector<void*>& v; ~Cleanup() { for (void* q : v) c->release(q); } } cleanup{ctx, owned};
    auto newd = [&](size_t n) -> double* { void* q = ctx->alloc(n * sizeof(double)); if (q) owned.push_back(q); return (double*)q; };
    auto newf = [&](size_t n) -> vf* { void* q = ctx->alloc(n * sizeof(vf)); if (q) owned.push_back(q); return (vf*)q; };
    {
        int h = H, w = W;
        for (int l = 0;; ++l) {
            Lvl L; memset(&L, 0, sizeof L); L.H = h; L.W = w; L.n = h * w; L.nine = l > 0 ? 1 : 0;
            const bool last = L.n <= 64 || (h <= 8 && w <= 8) || l >= 15;
            if (l == 0) { L.wE = (double*)wx; L.wS = (double*)wy; }
            else { L.wE = newd(L.n); L.wS = newd(L.n); L.wSE = newd(L.n); L.wSW = newd(L.n); L.fSE = newf(L.n); L.fSW = newf(L.n);
                   if (!L.wE || !L.wS || !L.wSE || !L.wSW || !L.fSE || !L.fSW) return NCT_ERR_HIP; }
            L.d = newd(L.n); L.fd = newf(L.n); L.fdinv = newf(L.n); L.fE = newf(L.n); L.fS = newf(L.n);
            if (!last) { L.pa = newd(L.n); L.pb = newd(L.n); }
            if (!last) { L.fpst = newf((size_t)((h + 1) / 2) * ((w + 1) / 2) * 9); L.fpw = newf((size_t)L.n * 4); }
            L.b = l == 0 ? nullptr : newf((size_t)L.n * nq0); L.x = newf((size_t)L.n * nq0); L.x2 = newf((size_t)L.n * nq0);
            if (l == 0 && ctx->wls_lines) { L.lxm = newf(L.n); L.lxp = newf(L.n); L.lym = newf(L.n); L.lyp = newf(L.n); if (!L.lxm || !L.lxp || !L.lym || !L.lyp) return NCT_ERR_HIP; }
            if (!L.d || !L.fd || !L.fdinv || !L.fE || !L.fS || (!last && (!L.pa || !L.pb || !L.fpst || !L.fpw)) || (l > 0 && !L.b) || !L.x || !L.x2) return NCT_ERR_HIP;
            lv.push_back(L);
            if (last) break;
            h = (h + 1) / 2; w = (w + 1) / 2;
        }
        if (lv.back().n > 64 || lv.size() < 2) return ctx->fail(NCT_ERR_INVALID, "wls: unsupported grid %dx%d (coarsest level %d)", W, H, lv.back().n);
    }
    const int nl = (int)lv.size();
    {
        double* pst = newd((size_t)lv[1].n * 9);                  // columns of P as 3x3 blocks, reused level by level
        if (!pst) return NCT_ERR_HIP;
        hipLaunchKernelGGL(k_mg_diag, dim3(cdiv(lv[0].n, 256)), dim3(256), 0, s, lv[0], rough); LCHK();
        if (ctx->wls_lines) { hipLaunchKernelGGL(k_mg_lines_setup, dim3(cdiv(lv[0].H * cdiv(lv[0].W, LBX) + lv[0].W * cdiv(lv[0].H, LBY), 128)), dim3(128), 0, s, lv[0]); LCHK(); }
        int l_tail = nl;                                        // first level built by k_mg_setup_tail
        for (int l = 1; l < nl; ++l) if (lv[l].n <= MG_TAIL_N) { l_tail = l; break; }
        if (nl > MG_MAXL) l_tail = nl;
        for (int l = 0; l < l_tail; ++l) {
            const dim3 g(cdiv(lv[l].n, 256));
            if (l > 0) {
                const dim3 gc(cdiv(lv[l].n, 128));
                hipLaunchKernelGGL(k_mg_pstencil, gc, dim3(128), 0, s, lv[l - 1], lv[l], pst); LCHK();
                hipLaunchKernelGGL(k_mg_pweights, dim3(cdiv(lv[l - 1].n, 256)), dim3(256), 0, s, lv[l - 1], lv[l]); LCHK();
                hipLaunchKernelGGL(k_mg_galerkin, gc, dim3(128), 0, s, lv[l - 1], lv[l], (const double*)pst); LCHK();
                hipLaunchKernelGGL(k_mg_finish, g, dim3(256), 0, s, lv[l]); LCHK();
            }
            if (l + 1 < nl) { hipLaunchKernelGGL(k_mg_weights, g, dim3(256), 0, s, lv[l]); LCHK(); }
        }
        if (l_tail < nl) {
            LvlPack P;
            for (int l = 0; l < nl; ++l) P.lv[l] = lv[l];
            hipLaunchKernelGGL(k_mg_setup_tail, dim3(1), dim3(1024), 0, s, P, l_tail, nl, pst); LCHK();
        }
    }
    const int N = lv[0].n, nb = cdiv(N, 256);
    static_assert(4 * sizeof(PState) <= 4096, "pinned read-back area too small");
    // ---- per-part buffers (both parts' before anything runs: the helper thread must not touch the arena)
    PartBufs part[2];
    const int nparts = split ? 2 : 1;
    for (int h = 0; h < nparts; ++h) {
        PartBufs& B = part[h];
        const size_t v = (size_t)N * nq0;
        B.rf = newf(v); if (!B.rf) return NCT_ERR_HIP;
        B.x6 = newd(v); B.r = newd(v); B.p = newd(v); B.sv = newd(v); B.w = newd(v); B.partial = newd((size_t)nb * 3 * nq0); B.sums = newd(3 * nq0);
        B.st = (PState*)ctx->alloc(2 * sizeof(PState)); if (B.st) owned.push_back(B.st);
        if (!B.x6 || !B.r || !B.p || !B.sv || !B.w || !B.partial || !B.sums || !B.st) return NCT_ERR_HIP;
        B.lv = lv;
        if (h == 1) for (int l = 0; l < nl; ++l) {             // the second part's own V-cycle vectors
            Lvl& L = B.lv[l];
            L.b = l == 0 ? nullptr : newf((size_t)L.n * nq0); L.x = newf((size_t)L.n * nq0); L.x2 = newf((size_t)L.n * nq0);
            if ((l > 0 && !L.b) || !L.x || !L.x2) return NCT_ERR_HIP;
        }
        B.hst = (PState*)ctx->pinned + 2 * h; B.ev[0] = ctx->ev_poll[2 * h]; B.ev[1] = ctx->ev_poll[2 * h + 1];
        B.maxit = ctx->wls_maxit; B.graph = ctx->wls_graph; B.forecast = ctx->wls_forecast != 0; B.lines = ctx->wls_lines != 0; B.trace = getenv("NCT_WLS_TRACE") != nullptr; B.rough = rough; B.kt = (ctx->kt_on && !split && !ctx->wls_graph) ? ctx : nullptr;   /* events recorded inside a stream capture cannot be read back: no kernel clock under NCT_WLS_GRAPH */
        memset(B.iters, 0, sizeof B.iters);
    }
    if (!split) {
        ErrSink sink;
        const int rc = pcg_part<6>(&sink, s, X, part[0], rtol);
        if (rc) return ctx->fail(rc, "%s", sink.err.c_str());
        if (iters_out) for (int q = 0; q < 6; ++q) iters_out[q] = part[0].iters[q];
        return 0;
    }
    // fork: the helper stream starts when the hierarchy (and everything before it on s) is complete; join: s waits for the helper's last kernel
    hipStream_t s2 = ctx->stream_wls;
    NCT_HIP(hipEventRecord(ctx->ev_wls_fork, s));
    NCT_HIP(hipStreamWaitEvent(s2, ctx->ev_wls_fork, 0));
    ErrSink sink_a, sink_b;
    int rc_b = 0;
    const int dev = ctx->device;
    std::thread helper([&] {
        if (hipSetDevice(dev) != hipSuccess) { rc_b = sink_b.fail(NCT_ERR_HIP, "hipSetDevice failed on the WLS helper thread"); return; }
        rc_b = pcg_part<3>(&sink_b, s2, X + (size_t)3 * N, part[1], rtol);
    });
    const int rc_a = pcg_part<3>(&sink_a, s, X, part[0], rtol);
    helper.join();
    hipError_t e1 = hipEventRecord(ctx->ev_wls_join, s2), e2 = hipStreamWaitEvent(s, ctx->ev_wls_join, 0);     // also on errors: the arena blocks return in s's order
    if (rc_a) return ctx->fail(rc_a, "%s", sink_a.err.c_str());
    if (rc_b) return ctx->fail(rc_b, "%s", sink_b.err.c_str());
    NCT_HIP(e1); NCT_HIP(e2);
    if (iters_out) for (int q = 0; q < 3; ++q) { iters_out[q] = part[0].iters[q]; iters_out[3 + q] = part[1].iters[q]; }
    return 0;
}
