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
#define NCT_MG_TXB 48
#define NCT_MG_TYB 8
#endif

// The PCG itself (and the hierarchy construction) is fp64; the V-cycle — a fixed linear preconditioner, whose accuracy does not
// limit the accuracy of the solution — runs in fp32 on rounded copies of the level operators: half the bytes on the two
// bandwidth-bound levels, same iteration counts (scripts/mg_convergence_experiments.py).
typedef float vf;
struct Lvl { int H, W, n; double *r, *wx, *wy, *diag;      // fp64 operator: data term, edge weights, diagonal
             vf *fdiag, *fdinv, *fwx, *fwy;                // fp32 copies; fdinv = (float)(omega / diag)
             vf *b, *x, *x2; };                            // V-cycle vectors, planar [6][n]

// State of the 6 right-hand sides of the single-reduction (Chronopoulos-Gear) PCG, double buffered: the update kernel of iteration k
// reads st[k & 1] and (workgroup 0) writes st[(k + 1) & 1]. nactive = number of systems still iterating: the host polls it only every
// few iterations, and every kernel of an iteration enqueued past convergence returns at once when it is 0.
struct PState { double gam[6], alp[6], bb[6]; int active[6]; int iters[6]; int nactive; };

// Fixed 256-wide tree s[t] += s[t + off], off = 128 … 1 (the order the oracle mirrors), evaluated with two barriers instead of nine: the two
// cross-wave steps go through LDS, the six steps inside the first wave are lane shifts (a lane t < off adds the value lane t + off held BEFORE
// the step, exactly as the array form does; what lanes >= off compute is never used).
template <int NV>
__device__ __forceinline__ void mg_block_reduce(double (&v)[NV], double* __restrict__ partial) {
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
            if (t == 0) partial[(size_t)blockIdx.x * NV + q] = x;
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

// y = M v at pixel i of a level (diag*v - sum_w w*v_nbr, neighbour order +x, -x, +y, -y)
template <int NQ, typename F>
__device__ __forceinline__ void lvl_op(const Lvl& L, int i, F&& val /* val(j, q) */, double (&y)[NQ]) {
    const int W = L.W, H = L.H;
    const int r = i / W, c = i - r * W;
    const double d = L.diag[i];
#pragma unroll
    for (int q = 0; q < NQ; ++q) y[q] = d * val(i, q);
    if (c + 1 < W) { const double w = L.wx[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= w * val(i + 1, q); }
    if (c > 0) { const double w = L.wx[i - 1];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= w * val(i - 1, q); }
    if (r + 1 < H) { const double w = L.wy[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= w * val(i + W, q); }
    if (r > 0) { const double w = L.wy[i - W];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= w * val(i - W, q); }
}

// ---- hierarchy construction
__global__ void k_mg_coarsen(Lvl F, Lvl C) {
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= C.n) return;
    const int Y = I / C.W, X = I - Y * C.W;
    const int y0 = 2 * Y, x0 = 2 * X;
    const bool x1ok = x0 + 1 < F.W, y1ok = y0 + 1 < F.H;
    double rs = F.r[y0 * F.W + x0];
    if (x1ok) rs += F.r[y0 * F.W + x0 + 1];
    if (y1ok) rs += F.r[(y0 + 1) * F.W + x0];
    if (x1ok && y1ok) rs += F.r[(y0 + 1) * F.W + x0 + 1];
    double ex = 0.0, ey = 0.0;
    if (x0 + 2 < F.W) {                       // fine edges (x0+1 -> x0+2) of both rows
        ex = F.wx[y0 * F.W + x0 + 1];
        if (y1ok) ex += F.wx[(y0 + 1) * F.W + x0 + 1];
    }
    if (y0 + 2 < F.H) {
        ey = F.wy[(y0 + 1) * F.W + x0];
        if (x1ok) ey += F.wy[(y0 + 1) * F.W + x0 + 1];
    }
    C.r[I] = rs; C.wx[I] = ex; C.wy[I] = ey;
}
__global__ void k_mg_diag(Lvl L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    const int y = i / L.W, x = i - y * L.W;
    double a00 = 0.0;
    a00 += L.r[i];
    if (x + 1 < L.W) a00 += L.wx[i];
    if (x > 0) a00 += L.wx[i - 1];
    if (y + 1 < L.H) a00 += L.wy[i];
    if (y > 0) a00 += L.wy[i - L.W];
    L.diag[i] = a00;
    L.fdiag[i] = (vf)a00; L.fdinv[i] = (vf)(OMEGA / a00); L.fwx[i] = (vf)L.wx[i]; L.fwy[i] = (vf)L.wy[i];
}

// ---- V-cycle (fp32, vectors planar [6][n])
// tile-fused legs: every intermediate iterate of a leg lives in LDS for a TX x TY fine tile plus a halo (recomputed by the
// neighbouring tiles with the same expressions, hence bit-identical) instead of making a round trip through global memory and a
// launch per sweep:
//   down: x1 = b*dinv ; x = x1 + (b - M x1)*dinv (two damped-Jacobi sweeps from zero) ; coarse rhs = sum over the 2x2 aggregate
//         of (b - M x), fine pixels in the order (0,0),(0,1),(1,0),(1,1)
//   up:   xe = x + e_coarse(parent) ; x2 = xe + (b - M xe)*dinv ; xo = x2 + (b - M x2)*dinv
// One thread per pixel of the tile + 2-pixel halo: it loads ITS pixel's right-hand side, coefficients and inputs once (all loads of
// a leg are issued in the first phase), the first iterate is exchanged through LDS on the halo-2 grid, the second on the halo-1
// grid, the result is produced on the tile. (Recomputing the first iterate at the 5 stencil points from global memory instead
// cost 40-75 loads per thread and ~28 us per leg at 700x700.)
struct PxCoef { vf d, dinv, w0, w1, w2, w3; bool r, l, dn, up; };    // diag, omega/diag, weights to +x, -x, +y, -y and their existence
__device__ __forceinline__ PxCoef px_coef(const Lvl& L, int gy, int gx) {
    const int i = gy * L.W + gx;
    PxCoef c;
    c.r = gx + 1 < L.W; c.l = gx > 0; c.dn = gy + 1 < L.H; c.up = gy > 0;
    c.d = L.fdiag[i]; c.dinv = L.fdinv[i];
    c.w0 = c.r ? L.fwx[i] : 0.f; c.w1 = c.l ? L.fwx[i - 1] : 0.f; c.w2 = c.dn ? L.fwy[i] : 0.f; c.w3 = c.up ? L.fwy[i - L.W] : 0.f;
    return c;
}
// y = M v at the pixel stored at LDS position p of a grid with row pitch LW (same operation order as lvl_op: +x, -x, +y, -y). `own` = the thread's own value of v,
// which it wrote to s_v[.. + p] itself and still holds in registers (a fifth of the stencil's LDS reads)
template <int NQ, int LW, int LN>
__device__ __forceinline__ void lds_op(const PxCoef& c, const vf* __restrict__ s_v, int p, const vf (&own)[NQ], vf (&y)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) y[q] = c.d * own[q];
    if (c.r) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= c.w0 * s_v[q * LN + p + 1]; }
    if (c.l) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= c.w1 * s_v[q * LN + p - 1]; }
    if (c.dn) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= c.w2 * s_v[q * LN + p + LW]; }
    if (c.up) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] -= c.w3 * s_v[q * LN + p - LW]; }
}
constexpr int mg_threads(int TX, int TY) { return ((TX + 2 * MG_NS) * (TY + 2 * MG_NS) + 63) / 64 * 64; }
// TB = type of this level's right-hand side in memory: double at level 0 (the PCG residual, rounded on load), vf below.
// One thread per pixel of the tile + MG_NS-pixel halo. Sweep k produces its iterate on the tile + (MG_NS - k)-pixel halo from the previous one (exchanged through LDS,
// two arrays in ping-pong); halo pixels are recomputed by the neighbouring tiles with the same expressions, hence bit-identical.
template <int NQ, int TX, int TY, typename TB>
__global__ __launch_bounds__(mg_threads(TX, TY)) void k_mg_down(const PState* __restrict__ st, Lvl F, const TB* __restrict__ b, vf* __restrict__ x, Lvl C, vf* __restrict__ bc) {
    if (st->nactive == 0) return;
    constexpr int HL = MG_NS, LW = TX + 2 * HL, LH = TY + 2 * HL, LN = LW * LH;
    __shared__ vf s_a[NQ * LN], s_b[NQ * LN];
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int p = threadIdx.x;
    const int ly = p / LW, lx = p - ly * LW;
    const int gy = y0 + ly - HL, gx = x0 + lx - HL;
    const bool valid = p < LN && gy >= 0 && gy < F.H && gx >= 0 && gx < F.W;
    auto ring = [&](int k) { return valid && lx >= k && lx <= TX + 2 * HL - 1 - k && ly >= k && ly <= TY + 2 * HL - 1 - k; };   // tile + (HL - k)-pixel halo
    const bool interior = ring(HL);
    const int i = gy * F.W + gx;
    vf bq[NQ], xk[NQ]; PxCoef c;
    if (valid) {
        c = px_coef(F, gy, gx);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { bq[q] = (vf)b[(size_t)q * F.n + i]; xk[q] = bq[q] * c.dinv; s_a[q * LN + p] = xk[q]; }     // sweep 0 (from zero)
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < MG_NS; ++k) {
        vf* src = (k & 1) ? s_a : s_b; vf* dst = (k & 1) ? s_b : s_a;
        if (ring(k)) {
            vf y[NQ]; lds_op<NQ, LW, LN>(c, src, p, xk, y);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                xk[q] = xk[q] + (bq[q] - y[q]) * (c.dinv * mg_rk(k));
                dst[q * LN + p] = xk[q];
                if (k == MG_NS - 1 && interior) x[(size_t)q * F.n + i] = xk[q];
            }
        }
        __syncthreads();
    }
    vf* xs = (MG_NS & 1) ? s_a : s_b; vf* rs = (MG_NS & 1) ? s_b : s_a;       // the smoothed iterate (on the tile + 1) and where the residual goes
    if (interior) {
        vf yv[NQ]; lds_op<NQ, LW, LN>(c, xs, p, xk, yv);
#pragma unroll
        for (int q = 0; q < NQ; ++q) rs[q * LN + p] = bq[q] - yv[q];
    }
    __syncthreads();
    if (p < (TX / 2) * (TY / 2)) {
        const int cy = p / (TX / 2), cx = p - cy * (TX / 2);
        const int Y = y0 / 2 + cy, X = x0 / 2 + cx;
        if (Y < C.H && X < C.W) {
            vf acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int yy = 2 * Y + (t >> 1), xx = 2 * X + (t & 1);
                if (yy < F.H && xx < F.W) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] += rs[q * LN + (yy - y0 + HL) * LW + (xx - x0 + HL)];
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) bc[(size_t)q * C.n + Y * C.W + X] = acc[q];
        }
    }
}
// xo must not alias x (neighbouring tiles still read x for their halo)
template <int NQ, int TX, int TY, typename TB>
__global__ __launch_bounds__(mg_threads(TX, TY)) void k_mg_up(const PState* __restrict__ st, Lvl L, const TB* __restrict__ b, const vf* __restrict__ x, int Wc, int nc,
                                                              const vf* __restrict__ ec, vf* __restrict__ xo) {
    if (st->nactive == 0) return;
    constexpr int HL = MG_NS, LW = TX + 2 * HL, LH = TY + 2 * HL, LN = LW * LH;
    __shared__ vf s_a[NQ * LN], s_b[NQ * LN];
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int p = threadIdx.x;
    const int ly = p / LW, lx = p - ly * LW;
    const int gy = y0 + ly - HL, gx = x0 + lx - HL;
    const bool valid = p < LN && gy >= 0 && gy < L.H && gx >= 0 && gx < L.W;
    auto ring = [&](int k) { return valid && lx >= k && lx <= TX + 2 * HL - 1 - k && ly >= k && ly <= TY + 2 * HL - 1 - k; };
    const int i = gy * L.W + gx;
    vf bq[NQ], xk[NQ]; PxCoef c;
    if (valid) {
        c = px_coef(L, gy, gx);
        const int ip = (gy >> 1) * Wc + (gx >> 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { bq[q] = (vf)b[(size_t)q * L.n + i]; xk[q] = x[(size_t)q * L.n + i] + ec[(size_t)q * nc + ip]; s_a[q * LN + p] = xk[q]; }   // xe = x + e_coarse(parent)
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MG_NS; ++k) {
        vf* src = (k & 1) ? s_b : s_a; vf* dst = (k & 1) ? s_a : s_b;
        if (ring(k + 1)) {
            vf y[NQ]; lds_op<NQ, LW, LN>(c, src, p, xk, y);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                xk[q] = xk[q] + (bq[q] - y[q]) * (k == 0 ? c.dinv : c.dinv * mg_rk(k));
                if (k == MG_NS - 1) xo[(size_t)q * L.n + i] = xk[q]; else dst[q * LN + p] = xk[q];
            }
        }
        if (k < MG_NS - 1) __syncthreads();
    }
}
// Middle + tail of the V-cycle in ONE launch, one 1024-thread workgroup PER RIGHT-HAND SIDE: the first fused level has <= 4096 pixels
// (44x44 at 700x700, 63x63 at 1000x1000), the deeper ones <= 1024 (22x22, 11x11, 6x6). The six systems share the operator but not a single
// value, so no workgroup ever waits for another and the whole sub-cycle needs only __syncthreads(). Everything a level needs for the way
// back up (right-hand side, pre-smoothed iterate, stencil coefficients of the deeper levels) stays in REGISTERS of the thread that owns
// the pixel; iterates are exchanged through whole-grid LDS arrays (no halos); coefficients come from global memory once per level (the
// first fused level re-reads them for the up leg) and all those loads are issued at the start. A first version that re-read
// coefficients and iterates from global memory in each of its 35 barrier-separated phases was SLOWER than the launches it replaced
//  — a single workgroup has nothing to hide a global round trip with.
// Replaces round 1's single-workgroup tail (<= 512 pixels, all six systems in one workgroup: 12.5 us) plus the k_mg_down / k_mg_up
// launches of the 44x44 level. Same expressions and operation order (+x, -x, +y, -y; children (0,0),(0,1),(1,0),(1,1)) as k_mg_down /
// k_mg_up, so the cycle is bit-identical. lv[0] is the first fused level: its rhs lv[0].b was written by the restriction above it, its
// correction goes to lv[0].x2. The coarsest grid (n <= 64) is solved by `sweeps` damped-Jacobi sweeps from zero by one wave (one lane per
// unknown, the iterate in a register, neighbours through ds_bpermute).
// Pixels per thread of the first / second fused level. 8 / 2 would take the 88x88 level of a 700x700 pair into the launch as well (two
// launches fewer per cycle), but 8 pixels x 7 coefficient registers do not fit the 128 VGPRs of a 1024-thread workgroup next to the deeper
// levels' state: 59 (NCT_MID_LDS0: right-hand side and iterate of depth 0 in LDS) to 92 spilled registers, +14 us per call, 40.5 vs 37.1 ms
// of WLS per pair.
#ifndef NCT_MID_P0
#define NCT_MID_P0 4
#define NCT_MID_P1 1
#endif
#ifndef NCT_MID_LDS0
#define NCT_MID_LDS0 0      // 1: depth 0 keeps its right-hand side and pre-smoothed iterate in LDS instead of registers
#endif
constexpr int MID_T = 1024, MID_P0 = NCT_MID_P0, MID_P1 = NCT_MID_P1, MID_N0 = MID_T * MID_P0, MID_N1 = MID_T * MID_P1, MID_N2 = MID_T, MID_LV = MID_P0 > 4 ? 7 : 6;
constexpr int mid_ppt(int D) { return D == 0 ? MID_P0 : (D == 1 ? MID_P1 : 1); }      // pixels per thread at depth D of the fused sub-cycle
struct MidPack { Lvl lv[MID_LV]; int nl; };
struct MidCoef { vf d, dinv, w0, w1, w2, w3; unsigned flags; };      // flags: 1 = +x exists, 2 = -x, 4 = +y, 8 = -y
__device__ __forceinline__ MidCoef mid_coef(const Lvl& L, int i) {
    const int gy = i / L.W, gx = i - gy * L.W;
    MidCoef c;
    const bool r = gx + 1 < L.W, l = gx > 0, dn = gy + 1 < L.H, up = gy > 0;
    c.flags = (r ? 1u : 0u) | (l ? 2u : 0u) | (dn ? 4u : 0u) | (up ? 8u : 0u);
    c.d = L.fdiag[i]; c.dinv = L.fdinv[i];
    c.w0 = r ? L.fwx[i] : 0.f; c.w1 = l ? L.fwx[i - 1] : 0.f; c.w2 = dn ? L.fwy[i] : 0.f; c.w3 = up ? L.fwy[i - L.W] : 0.f;
    return c;
}
__device__ __forceinline__ vf mid_op(const MidCoef& c, const vf* __restrict__ s_v, int i, int W, vf* centre = nullptr) {
    const vf v0 = s_v[i];
    if (centre) *centre = v0;
    vf y = c.d * v0;
    if (c.flags & 1u) y -= c.w0 * s_v[i + 1];
    if (c.flags & 2u) y -= c.w1 * s_v[i - 1];
    if (c.flags & 4u) y -= c.w2 * s_v[i + W];
    if (c.flags & 8u) y -= c.w3 * s_v[i - W];
    return y;
}
// restricted residual of coarse pixel I: children of the fine grid (Wf x Hf) in the order (0,0),(0,1),(1,0),(1,1)
__device__ __forceinline__ vf mid_restrict(const vf* __restrict__ s_res, int I, int Wc, int Wf, int Hf) {
    const int Y = I / Wc, X = I - Y * Wc;
    vf acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = 2 * Y + (k >> 1), xx = 2 * X + (k & 1);
        if (yy < Hf && xx < Wf) acc += s_res[yy * Wf + xx];
    }
    return acc;
}
// Level D of the fused sub-cycle. In: this level's right-hand side b[] in registers (pixel i = t + k * MID_T). Out: this level's correction
// in s_out (LDS, n values) — or in global L.x2 for D == 0. sA / sB: exchange arrays (>= n); sC: the child's correction (n_child values).
template <int D>
__device__ __forceinline__ void mid_level(const MidPack& P, int q, int t, vf* __restrict__ sA, vf* __restrict__ sB, vf* __restrict__ sC,
                                          vf* __restrict__ sb0, vf* __restrict__ sx0, const vf (&breg)[mid_ppt(D)], int sweeps) {
    constexpr int PPT = mid_ppt(D);
    constexpr bool INLDS = D == 0 && NCT_MID_LDS0 != 0;
    const Lvl& L = P.lv[D];
    const int n = L.n, W = L.W;
    auto bval = [&](int k, int i) -> vf { if constexpr (INLDS) return sb0[i]; else return breg[k]; };
    if (D == P.nl - 1) {
        // ---- coarsest grid: wave 0, one lane per unknown (same code as the single-workgroup tail of round 1)
        if (t < n) sA[t] = bval(0, t);
        __syncthreads();
        if (t < 64) {
            const int i = t, H = L.H;
            const bool live = i < n;
            const int r = live ? i / W : 0, c = live ? i - r * W : 0;
            vf bq = 0, d = 0, dv = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            const bool has_r = live && c + 1 < W, has_l = live && c > 0, has_d = live && r + 1 < H, has_u = live && r > 0;
            if (live) {
                bq = sA[i]; d = L.fdiag[i]; dv = L.fdinv[i];
                if (has_r) w0 = L.fwx[i];
                if (has_l) w1 = L.fwx[i - 1];
                if (has_d) w2 = L.fwy[i];
                if (has_u) w3 = L.fwy[i - W];
            }
            vf x = 0.0f;
            for (int s = 0; s < sweeps; ++s) {
                const vf xr = __shfl(x, (i + 1) & 63), xl = __shfl(x, (i - 1) & 63), xd = __shfl(x, (i + W) & 63), xu = __shfl(x, (i - W) & 63);
                vf y = d * x;
                if (has_r) y -= w0 * xr;
                if (has_l) y -= w1 * xl;
                if (has_d) y -= w2 * xd;
                if (has_u) y -= w3 * xu;
                if (live) x = x + (bq - y) * (dv * MG_R0);
            }
            if (live) { if (D == 0) L.x2[(size_t)q * n + i] = x; else sC[i] = x; }
        }
        __syncthreads();
        return;
    }
    if constexpr (D + 1 < MID_LV) {
        const Lvl& C = P.lv[D + 1];
        MidCoef c[PPT]; vf x[PPT];
        // ---- down: MG_NS sweeps from zero (x_1 = b*dinv ; x_{s+1} = x_s + (b - M x_s)*dinv*rk(s)), iterates in ping-pong through sA / sB ; res = b - M x
#pragma unroll
        for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) { c[k] = mid_coef(L, i); sA[i] = bval(k, i) * c[k].dinv; } }
        __syncthreads();
#pragma unroll
        for (int sw = 1; sw < MG_NS; ++sw) {
            vf* src = (sw & 1) ? sA : sB; vf* dst = (sw & 1) ? sB : sA;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = t + k * MID_T;
                if (i < n) {
                    vf xs; const vf y = mid_op(c[k], src, i, W, &xs);
                    const vf xv = xs + (bval(k, i) - y) * (c[k].dinv * mg_rk(sw));
                    dst[i] = xv;
                    if (sw == MG_NS - 1) { if constexpr (INLDS) sx0[i] = xv; else x[k] = xv; }
                }
            }
            __syncthreads();
        }
        {
            vf* xs = (MG_NS & 1) ? sA : sB; vf* rs = (MG_NS & 1) ? sB : sA;
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) rs[i] = bval(k, i) - mid_op(c[k], xs, i, W); }
        }
        __syncthreads();
        constexpr int CPT = mid_ppt(D + 1);
        vf bc[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) { const int I = t + k * MID_T; bc[k] = I < C.n ? mid_restrict((MG_NS & 1) ? sB : sA, I, C.W, W, L.H) : 0.f; }
        __syncthreads();                                      // the residual array is free again
        mid_level<D + 1>(P, q, t, sA, sB, sC, sb0, sx0, bc, sweeps);    // its correction arrives in sC
        // ---- up: xe = x + e_coarse(parent) ; x2 = xe + (b - M xe)*dinv ; xo = x2 + (b - M x2)*dinv
        if constexpr (D == 0) {                                // the first fused level does not keep its coefficients across the deeper levels
            asm volatile("" ::: "memory");                     // (a real reload: without the clobber the first loads' registers stay live)
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const int i = t + k * MID_T; if (i < n) c[k] = mid_coef(L, i); }
        }
        // (the iterates xe and x2 of the own pixel are read back from the exchange arrays, where the operator reads them anyway)
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = t + k * MID_T;
            if (i < n) {
                const int gy = i / W, gx = i - gy * W;
                vf xk; if constexpr (INLDS) xk = sx0[i]; else xk = x[k];
                sA[i] = xk + sC[(gy >> 1) * C.W + (gx >> 1)];
            }
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
                    const vf v = xs + (bval(k, i) - y) * (sw == 0 ? c[k].dinv : c[k].dinv * mg_rk(sw));
                    if (sw < MG_NS - 1) dst[i] = v;
                    else if (D == 0) L.x2[(size_t)q * n + i] = v; else sC[i] = v;
                }
            }
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(MID_T) void k_mg_mid(const PState* __restrict__ st, MidPack P, int sweeps) {
    if (st->nactive == 0) return;
    __shared__ vf sA[MID_N0], sB[MID_N0], sC[MID_N1];
    const int q = blockIdx.x, t = threadIdx.x;
    const Lvl& L0 = P.lv[0];
    vf b[MID_P0];
#if NCT_MID_LDS0
    __shared__ vf sb0[MID_N0], sx0[MID_N0];
#pragma unroll
    for (int k = 0; k < MID_P0; ++k) { const int i = t + k * MID_T; b[k] = 0.f; if (i < L0.n) sb0[i] = L0.b[(size_t)q * L0.n + i]; }
    mid_level<0>(P, q, t, sA, sB, sC, sb0, sx0, b, sweeps);         // each thread reads back only what it wrote: no barrier needed
#else
#pragma unroll
    for (int k = 0; k < MID_P0; ++k) { const int i = t + k * MID_T; b[k] = i < L0.n ? L0.b[(size_t)q * L0.n + i] : 0.f; }
    mid_level<0>(P, q, t, sA, sB, sC, nullptr, nullptr, b, sweeps);
#endif
}

// ---- PCG pieces at the fine level

// x6 = interleave(X); r = rough*x0 - M x0 ; partial: rr, bb (12)
template <int NQ>
__global__ __launch_bounds__(256) void k_pcg_start(Lvl L, const double* __restrict__ X /*[2][n][3]*/, double* __restrict__ x6, double* __restrict__ r, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc[2 * NQ];
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) acc[q] = 0.0;
    if (i < L.n) {
        auto xv = [&](int j, int q) { return X[((size_t)(q / 3) * L.n + j) * 3 + (q % 3)]; };
        double y[NQ]; lvl_op<NQ>(L, i, xv, y);
        const double rg = L.r[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double x0 = xv(i, q);
            const double bq = rg * x0;
            const double rv = bq - y[q];
            x6[(size_t)q * L.n + i] = x0; r[(size_t)q * L.n + i] = rv;
            acc[q] = rv * rv; acc[NQ + q] = bq * bq;
        }
    }
    mg_block_reduce<2 * NQ>(acc, partial);
}
template <int NQ>
__global__ void k_pcg_start_fin(const double* __restrict__ partial, int nb, PState* __restrict__ st, double rtol2) {
    double s[2 * NQ]; mg_final_reduce<2 * NQ>(partial, nb, s);
    if (threadIdx.x < NQ) { const int q = threadIdx.x; st->bb[q] = s[NQ + q]; st->gam[q] = 0; st->alp[q] = 0; st->iters[q] = 0;
                            st->active[q] = (s[q] > rtol2 * s[NQ + q]) ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) { int na = 0; for (int q = 0; q < NQ; ++q) na += st->active[q]; st->nactive = na; }
}
// Single-reduction PCG (Chronopoulos & Gear): per iteration  u = M^-1 r (the V-cycle, fp32) ; w = A u ; gamma = r.u, delta = w.u,
// rho = r.r in ONE reduction ; beta = gamma/gamma_old, alpha = gamma / (delta - beta*gamma/alpha_old) ; p = u + beta p ; s = w + beta s
// (= A p by recurrence) ; x += alpha p ; r -= alpha s. Same iterates as textbook PCG in exact arithmetic and the same iteration counts
// in practice (scripts/cgcg_check.py), with 3 launches and one reduction per iteration instead of 7 and three.
// w = A u with u = the V-cycle output widened exactly; partial sums of gamma, delta, rho (18 per 256-pixel block)
template <int NQ>
__global__ __launch_bounds__(256) void k_cg_apply(const PState* __restrict__ st, Lvl L, const vf* __restrict__ z, const double* __restrict__ r,
                                                  double* __restrict__ w, double* __restrict__ partial) {
    if (st->nactive == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc[3 * NQ];
#pragma unroll
    for (int q = 0; q < 3 * NQ; ++q) acc[q] = 0.0;
    if (i < L.n) {
        auto uv = [&](int j, int q) { return (double)z[(size_t)q * L.n + j]; };
        double y[NQ]; lvl_op<NQ>(L, i, uv, y);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double u = uv(i, q), rv = r[(size_t)q * L.n + i];
            w[(size_t)q * L.n + i] = y[q];
            acc[q] = rv * u; acc[NQ + q] = y[q] * u; acc[2 * NQ + q] = rv * rv;
        }
    }
    mg_block_reduce<3 * NQ>(acc, partial);
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
                                                   double* __restrict__ x, double* __restrict__ r) {
    if (sc->nactive == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) *sn = *sc; return; }
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
        }
        sn->nactive = na;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!act[q]) continue;
        const size_t j = (size_t)q * n + i;
        const double zv = (double)z[j], wv = w[j];
        const double pn = first ? zv : zv + be[q] * p[j];
        const double sv = first ? wv : wv + be[q] * s[j];
        p[j] = pn; s[j] = sv;
        x[j] += al[q] * pn;
        r[j] -= al[q] * sv;
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
struct PartBufs {
    double *x6, *r, *p, *sv, *w, *partial, *sums; PState* st;      // Krylov vectors [NQ][N], reduction scratch, double-buffered state
    std::vector<Lvl> lv;                                            // the shared operator hierarchy with THIS part's V-cycle vectors (b, x, x2)
    PState* hst; hipEvent_t ev[2];                                  // two page-locked read-back slots and their events
    int maxit, graph;
    int iters[NQMAX];
};
template <int NQ>
int pcg_part(ErrSink* ctx, hipStream_t s, double* X /* this part's [N][3] block(s) */, PartBufs& B, double rtol) {
    const std::vector<Lvl>& lv = B.lv;
    const int nl = (int)lv.size();
    const Lvl& F = lv[0];
    const int N = F.n, nb = cdiv(N, 256);
    double *x6 = B.x6, *r = B.r, *p = B.p, *sv = B.sv, *w = B.w, *partial = B.partial, *sums = B.sums;
    PState* st = B.st;                                 // st[0] / st[1]; `cur` = the state the iteration being enqueued reads
    const PState* cur = st;
    const double rtol2 = rtol * rtol;
    hipLaunchKernelGGL(k_pcg_start<NQ>, dim3(nb), dim3(256), 0, s, F, (const double*)X, (double*)x6, (double*)r, (double*)partial); LCHK();
    hipLaunchKernelGGL(k_pcg_start_fin<NQ>, dim3(1), dim3(256), 0, s, (const double*)partial, nb, st, rtol2); LCHK();

    // z = Vcycle(r). Levels 0..nl-2 run the tile-fused down/up legs (2 launches per level), the coarsest grid one 6-wave kernel.
    // res[l] = where level l's correction ends up (the up leg cannot write in place: neighbouring tiles still read lv[l].x).
    // The deepest levels run in k_mg_mid, one workgroup per right-hand side (round 1's tail — ONE workgroup for all six systems — had to stop
    // at 512 pixels: with the 44x44 level it took 157 us per cycle).
    // tail0 = first level of the fused middle + tail (k_mg_mid): the deepest run of levels whose first has <= MID_N0 pixels and the others
    // <= MID_N1, at most MID_LV of them; never the fine level (its right-hand side is the fp64 PCG residual and its tiles fill the chip)
    int tail0 = nl - 1;
    auto mid_fits = [&](int first) {                     // levels first .. nl-1 as depths 0 .. of k_mg_mid
        if (nl - first > MID_LV) return false;
        for (int l = first; l < nl; ++l) { const int d = l - first; if (lv[l].n > (d == 0 ? MID_N0 : (d == 1 ? MID_N1 : MID_N2))) return false; }
        return true;
    };
    while (tail0 > 1 && mid_fits(tail0 - 1)) --tail0;
    MidPack pack; memset(&pack, 0, sizeof pack); pack.nl = nl - tail0;
    for (int l = tail0; l < nl; ++l) pack.lv[l - tail0] = lv[l];
    // Tile shapes: bandwidth-bound levels use TXB x TYB tiles; below 100k pixels the legs are latency bound, so a 16x8 tile whose
    // haloed footprint (18x10) fits one pass of the 256 threads keeps the dependent load chains short and spreads over more CUs.
    constexpr int TXB = NCT_MG_TXB, TYB = NCT_MG_TYB;
    auto down = [&](int l) {
        const dim3 gb(cdiv(lv[l].W, TXB), cdiv(lv[l].H, TYB)), gs(cdiv(lv[l].W, 16), cdiv(lv[l].H, 8));
        if (l == 0) {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_down<NQ, TXB, TYB, double>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const double*)r, lv[l].x, lv[l + 1], lv[l + 1].b);
            else                   hipLaunchKernelGGL((k_mg_down<NQ, 16, 8, double>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const double*)r, lv[l].x, lv[l + 1], lv[l + 1].b);
        } else {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_down<NQ, TXB, TYB, vf>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const vf*)lv[l].b, lv[l].x, lv[l + 1], lv[l + 1].b);
            else                   hipLaunchKernelGGL((k_mg_down<NQ, 16, 8, vf>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const vf*)lv[l].b, lv[l].x, lv[l + 1], lv[l + 1].b);
        }
    };
    auto up = [&](int l, const vf* ec) {
        const dim3 gb(cdiv(lv[l].W, TXB), cdiv(lv[l].H, TYB)), gs(cdiv(lv[l].W, 16), cdiv(lv[l].H, 8));
        const int Wc = lv[l + 1].W, nc = lv[l + 1].n;
        if (l == 0) {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_up<NQ, TXB, TYB, double>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const double*)r, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
            else                   hipLaunchKernelGGL((k_mg_up<NQ, 16, 8, double>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const double*)r, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
        } else {
            if (lv[l].n >= 100000) hipLaunchKernelGGL((k_mg_up<NQ, TXB, TYB, vf>), gb, dim3(mg_threads(TXB, TYB)), 0, s, cur, lv[l], (const vf*)lv[l].b, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
            else                   hipLaunchKernelGGL((k_mg_up<NQ, 16, 8, vf>), gs, dim3(mg_threads(16, 8)), 0, s, cur, lv[l], (const vf*)lv[l].b, (const vf*)lv[l].x, Wc, nc, ec, lv[l].x2);
        }
    };
    auto vcycle = [&]() -> int {
        for (int l = 0; l < tail0; ++l) { down(l); LCHK(); }
        hipLaunchKernelGGL(k_mg_mid, dim3(NQ), dim3(MID_T), 0, s, cur, pack, 60); LCHK();
        for (int l = tail0 - 1; l >= 0; --l) { up(l, lv[l + 1].x2); LCHK(); }
        return 0;
    };
    const vf* z = lv[0].x2;
    // Convergence is polled without draining the stream: after every batch of `batch` iterations the solver state is copied to
    // page-locked host memory and an event is recorded; the host then enqueues the NEXT batch before it waits for that event, so
    // the GPU always has a batch queued. The batch enqueued past convergence costs only empty launches (nactive == 0).
    const int maxit = B.maxit, batch = NCT_WLS_BATCH;
    PState* hst = B.hst;                                  // two slots of page-locked memory
    auto iteration = [&](int it) -> int {
        cur = st + (it & 1);
        PState* nxt = st + ((it + 1) & 1);
        int rc = vcycle(); if (rc) return rc;
        hipLaunchKernelGGL(k_cg_apply<NQ>, dim3(nb), dim3(256), 0, s, cur, F, z, (const double*)r, (double*)w, (double*)partial); LCHK();
        hipLaunchKernelGGL(k_cg_fin<NQ>, dim3(3), dim3(256), 0, s, cur, (const double*)partial, nb, (double*)sums); LCHK();
        hipLaunchKernelGGL(k_cg_update<NQ>, dim3(nb), dim3(256), 0, s, N, cur, nxt, (const double*)sums, rtol2, it == 0 ? 1 : 0, z, (const double*)w,
                           (double*)p, (double*)sv, (double*)x6, (double*)r); LCHK();
        cur = nxt;
        return 0;
    };
    auto snapshot = [&](int slot) -> int {
        NCT_HIP(hipMemcpyAsync(&hst[slot], cur, sizeof(PState), hipMemcpyDeviceToHost, s));
        NCT_HIP(hipEventRecord(B.ev[slot], s));
        return 0;
    };
    int it = 0, slot = 0; bool done = false;
    { int rc = snapshot(slot); if (rc) return rc; }        // state after the start kernel (x0 may already solve the system)
    PState fin; memset(&fin, 0, sizeof fin);
    // Experiment hook (NCT_WLS_GRAPH=1): from the second batch on, the iteration batch (its kernels, arguments and the
    // state double-buffering repeat exactly) is captured once and replayed as a HIP graph instead of being enqueued kernel by kernel.
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    struct GraphCleanup { hipGraph_t& g; hipGraphExec_t& e; ~GraphCleanup() { if (e) (void)hipGraphExecDestroy(e); if (g) (void)hipGraphDestroy(g); } } gcleanup{graph, gexec};
    while (true) {
        const bool enqueued = it < maxit;
        if (enqueued && B.graph && it >= batch) {
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
        if (enqueued) { for (int k = 0; k < batch; ++k, ++it) { int rc = iteration(it); if (rc) return rc; } }
        { int rc = snapshot(slot ^ 1); if (rc) return rc; }
        NCT_HIP(hipEventSynchronize(B.ev[slot]));  // the snapshot taken BEFORE the batch just enqueued
        fin = hst[slot];
        if (fin.nactive == 0) { done = true; break; }
        if (!enqueued) break;                              // that was the snapshot behind the last batch: the iteration budget is spent
        slot ^= 1;
    }
    // iterations enqueued after `fin` was taken leave the state untouched (nactive == 0), so fin is final
    if (!done) return ctx->fail(NCT_ERR_HIP, "WLS MG-PCG did not converge in %d iterations", maxit);
    hipLaunchKernelGGL(k_pcg_finish<NQ>, dim3(cdiv(N * NQ, 256)), dim3(256), 0, s, N, (const double*)x6, X); LCHK();
    for (int q = 0; q < NQ; ++q) B.iters[q] = fin.iters[q];
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
            Lvl L; memset(&L, 0, sizeof L); L.H = h; L.W = w; L.n = h * w;
            if (l == 0) { L.r = (double*)rough; L.wx = (double*)wx; L.wy = (double*)wy; }
            else { L.r = newd(L.n); L.wx = newd(L.n); L.wy = newd(L.n); }
            L.diag = newd(L.n); L.fdiag = newf(L.n); L.fdinv = newf(L.n); L.fwx = newf(L.n); L.fwy = newf(L.n);
            L.b = l == 0 ? nullptr : newf((size_t)L.n * nq0); L.x = newf((size_t)L.n * nq0); L.x2 = newf((size_t)L.n * nq0);
            if (!L.r || !L.wx || !L.wy || !L.diag || !L.fdiag || !L.fdinv || !L.fwx || !L.fwy || (l > 0 && !L.b) || !L.x || !L.x2) return NCT_ERR_HIP;
            lv.push_back(L);
            if (L.n <= 64 || (h <= 8 && w <= 8) || lv.size() >= 16) break;
            h = (h + 1) / 2; w = (w + 1) / 2;
        }
        if (lv.back().n > 64 || lv.size() < 2) return ctx->fail(NCT_ERR_INVALID, "wls: unsupported grid %dx%d (coarsest level %d)", W, H, lv.back().n);
    }
    const int nl = (int)lv.size();
    for (int l = 0; l < nl; ++l) {
        if (l > 0) { hipLaunchKernelGGL(k_mg_coarsen, dim3(cdiv(lv[l].n, 256)), dim3(256), 0, s, lv[l - 1], lv[l]); LCHK(); }
        hipLaunchKernelGGL(k_mg_diag, dim3(cdiv(lv[l].n, 256)), dim3(256), 0, s, lv[l]); LCHK();
    }
    const int N = lv[0].n, nb = cdiv(N, 256);
    static_assert(4 * sizeof(PState) <= 4096, "pinned read-back area too small");
    // ---- per-part buffers (both parts' before anything runs: the helper thread must not touch the arena)
    PartBufs part[2];
    const int nparts = split ? 2 : 1;
    for (int h = 0; h < nparts; ++h) {
        PartBufs& B = part[h];
        const size_t v = (size_t)N * nq0;
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
        B.maxit = ctx->wls_maxit; B.graph = ctx->wls_graph;
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
