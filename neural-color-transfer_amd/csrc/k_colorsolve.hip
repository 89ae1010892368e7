// k_colorsolve.hip — local colour transfer on the GPU: T1 local statistics, T2 confidence weights, S1 nonlocal
// least squares (truncated CG), U1 upsample + roughness, S2 edge-aware WLS smoothing (PCG), A1 apply.
// Reference: ColorTransfer::transfer_color_downsample ColorTransfer.cpp:1180-1478 and what it calls:
//   stats loop :1194-1265 (+ build_accumTable_downsample :425-455), weights :1302-1357,
//   solve_nonlocal_downsample_gpu_gradient :548-949 -> solve_ls_cg_gpu SparseSolver_GPU.cu:3-198,
//   upsample_color_coefficients_bilinear :457-490, solve_WLS_roughness_cpu :951-1125 -> PARDISO SparseSolver_CPU.cpp:104-286.
//
// MI355X design (the reference assembles CSR on the host, ships it to the GPU three times, forms A^T A with SpGEMM and
// factorises a 490k x 490k matrix with PARDISO on the CPU at every level):
//  * nothing leaves the device. All vectors are fp64, interleaved [pixel][3 Lab channels], a-part then b-part;
//  * S1 is matrix free: A has <= 2 non-zeros per row, so A^T A is (a) a 2x2 data block per pixel and channel,
//    (b) twice the 5-point graph Laplacian with weights g^2 and (c) the kNN graph Laplacian. The out-edges come from the
//    [n][8] kNN table, the in-edges from a radix-sorted reverse adjacency. The three Lab channels run in lock step
//    with their own CG scalars; it is the SAME truncated, un-preconditioned recurrence started from the local-stats
//    guess and stopped by the 50/100 iteration cap — the iterate count is part of the result;
//  * S2: the 5-point SPD system is solved for its 6 right-hand sides in lock step by preconditioned CG
//    (the reference's direct solve is exact, so any converged solver is result-equivalent);
//  * all reductions are two-stage with fixed-shape trees => run-to-run deterministic.
// Roofline: HBM streaming of a handful of fp64 vectors per iteration (latency/launch bound at the coarse levels).
#include "nct_internal.h"
#include "nct_device.h"
#include "nct_detmath.h"
#include <cstring>
#include <cstdio>
#include <rocprim/device/device_radix_sort.hpp>   // rocPRIM directly (no CUB-compatibility layer)

#define LAB_D(u) ((double)(u) * (1.0 / 255.0))      // Mat::convertTo(CV_64F, 1/255)

// ---------------------------------------------------------------- deterministic block reduction of NQ doubles
// Fixed 256-wide tree s[t] += s[t + off], off = 128 … 1 (the order the oracle mirrors), evaluated with two barriers instead of nine: the two cross-wave
// steps go through LDS, the six steps inside the first wave are lane shifts (a lane t < off adds the value lane t + off held BEFORE the step, exactly as
// the array form does; what lanes >= off compute is never used).
template <int NQ>
__device__ __forceinline__ void tree256(double (&v)[NQ], double* __restrict__ s_red /*[128 * NQ]*/) {
    const int t = threadIdx.x;
    if (t >= 128) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_red[q * 128 + t - 128] = v[q];
    }
    __syncthreads();
    if (t < 128) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] += s_red[q * 128 + t];            // off = 128
    }
    __syncthreads();
    if (t >= 64 && t < 128) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_red[q * 128 + t - 64] = v[q];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double x = v[q] + s_red[q * 128 + t];                          // off = 64
            x += __shfl_down(x, 32); x += __shfl_down(x, 16); x += __shfl_down(x, 8);
            x += __shfl_down(x, 4); x += __shfl_down(x, 2); x += __shfl_down(x, 1);
            v[q] = x;                                                      // lane 0 holds the sum
        }
    }
}
template <int NQ>
__device__ __forceinline__ void block_reduce_store(double (&v)[NQ], double* __restrict__ partial /*[nblocks][NQ]*/) {
    __shared__ double s_red[128 * NQ];
    tree256<NQ>(v, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) partial[(size_t)blockIdx.x * NQ + q] = v[q];
    }
}
// sum partial[nb][NQ] in a fixed order (single block of 256 threads): thread t adds its partials b = t, t + 256, … in ascending order, then the same tree
template <int NQ>
__device__ __forceinline__ void final_reduce(const double* __restrict__ partial, int nb, double (&out)[NQ]) {
    __shared__ double s_fin[128 * NQ];
    __shared__ double s_out[NQ];
    const int t = threadIdx.x;
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
    for (int b = t; b < nb; b += 256)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] += partial[(size_t)b * NQ + q];
    tree256<NQ>(acc, s_fin);
    if (t == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_out[q] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = s_out[q];
    __syncthreads();
}

// ================================================================= T1 local statistics
__global__ void k_local_stats(const uint8_t* __restrict__ cnt, const uint8_t* __restrict__ stl, int h, int w, double eps,
                              double* __restrict__ a, double* __restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const int sx = max(x - 1, 0), sy = max(y - 1, 0), ex = min(x + 2, w), ey = min(y + 2, h);
    const int cSum = (ex - sx) * (ey - sy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        long long cs = 0, cs2 = 0, ss = 0, ss2 = 0;
        for (int yy = sy; yy < ey; ++yy)
            for (int xx = sx; xx < ex; ++xx) {
                const int cv = cnt[((size_t)yy * w + xx) * 3 + c], sv = stl[((size_t)yy * w + xx) * 3 + c];
                cs += cv; cs2 += cv * cv; ss += sv; ss2 += sv * sv;
            }
        const double cm = (double)cs / (double)cSum;
        double cvr = (double)cs2 / (double)cSum - cm * cm; cvr = cvr > 0.0 ? cvr : 0.0;
        double csd = sqrt(cvr); csd = csd > 0.0 ? csd : 0.0;
        const double sm = (double)ss / (double)cSum;
        double svr = (double)ss2 / (double)cSum - sm * sm; svr = svr > 0.0 ? svr : 0.0;
        double ssd = sqrt(svr); ssd = ssd > 0.0 ? ssd : 0.0;
        const double av = ssd / (csd + eps);
        a[(size_t)i * 3 + c] = av;
        b[(size_t)i * 3 + c] = (sm - cm * av) * (1.0 / 255.0);
    }
}

// ================================================================= T2 weights
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__global__ void k_minmax_f(const float* __restrict__ v, int n, unsigned* __restrict__ mm) {
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const unsigned o = f2ord(v[i]); lo = min(lo, o); hi = max(hi, o); }
    for (int off = 32; off >= 1; off >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, off)); hi = max(hi, (unsigned)__shfl_xor((int)hi, off)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}
__global__ void k_err_weight(const float* __restrict__ err, int n, const unsigned* __restrict__ mm, double* __restrict__ weight) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mn = (double)ord2f(mm[0]), mx = (double)ord2f(mm[1]);
    const double e = ((double)err[i] - mn) / (mx - mn);
    const double wv = 1.0 - e;
    weight[i] = wv > 1e-6 ? wv : 1e-6;
}

// gradient weights g = sqrt(lamda / (|dL|^alpha + 1e-4)) of the L channel of an 8-bit Lab image (compute_gradientMat)
__global__ void k_gradient_weights(const uint8_t* __restrict__ lab, int h, int w, double lamda, double alpha, double* __restrict__ gx, double* __restrict__ gy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const double val = LAB_D(lab[(size_t)i * 3]);
    double vx = 0.0, vy = 0.0;
    if (x + 1 < w) { const double g = LAB_D(lab[(size_t)(i + 1) * 3]) - val; vx = sqrt(lamda / (nct_pow(fabs(g), alpha) + 0.0001)); }
    if (y + 1 < h) { const double g = LAB_D(lab[(size_t)(i + w) * 3]) - val; vy = sqrt(lamda / (nct_pow(fabs(g), alpha) + 0.0001)); }
    gx[i] = vx; gy[i] = vy;
}

// ================================================================= S1 nonlocal least squares
struct S1Sys {
    int n, h, w;
    const double *daa, *dab, *dbb;          // [n][3]: (dw s)^2, (dw s) dw, dw^2
    const double *gx, *gy;                  // [n]
    const int* knn_id; const double* iw2;   // [n][8]
    const int* rev_start; const int* rev_src; const double* rev_w;   // reverse adjacency (in-edges sorted by target, then by edge id src*8+ki): source pixel, iw2 of the edge
};
struct CGState { double r0[6], r1[6], va[6], vb[6]; int active[6]; int iters[6]; };

__global__ void k_s1_setup(int n, const double* __restrict__ weight, float dWeight, const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref,
                           const double* __restrict__ knn_w, double nonlocalWeight,
                           double* __restrict__ daa, double* __restrict__ dab, double* __restrict__ dbb, double* __restrict__ rhs /*[2][n][3]*/,
                           double* __restrict__ iw2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dw = sqrt(weight[i]) * (double)sqrtf(dWeight);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v0 = dw * LAB_D(src[(size_t)i * 3 + c]);
        const double rb = dw * LAB_D(ref[(size_t)i * 3 + c]);
        daa[(size_t)i * 3 + c] = v0 * v0; dab[(size_t)i * 3 + c] = v0 * dw; dbb[(size_t)i * 3 + c] = dw * dw;
        rhs[(size_t)i * 3 + c] = v0 * rb; rhs[(size_t)(n + i) * 3 + c] = dw * rb;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const double iw = sqrt(knn_w[(size_t)i * 8 + k]) * nonlocalWeight; iw2[(size_t)i * 8 + k] = iw * iw; }
}

// y = Op(p) at pixel i (live = i < n); must be called by every thread of a 256-thread workgroup whose threads own consecutive
// pixels. The in-degree of the kNN graph is mild on average (8) but uneven (p99 19, max 37 at 700x700: scripts/knn_indegree.py),
// and an in-edge is a dependent random 48-byte gather: with one thread walking its own list a wave waits for its longest list
// (a uniform-degree graph runs the whole stage 20 % faster: scripts/s1_locality_probe.py). So the gathers are shared: the in-edges of
// a workgroup's 256 consecutive pixels are ONE contiguous range of the target-sorted edge arrays; the threads fetch it edge-parallel
// into LDS in chunks of S1_CHUNK edges, then every thread adds ITS edges from LDS in edge order — the per-pixel operation order
// (local, out-edges, in-edges ascending) and therefore every bit of the result is unchanged.
#ifndef NCT_S1_CHUNK
#define NCT_S1_CHUNK 1024
#endif
constexpr int S1_CHUNK = NCT_S1_CHUNK;
template <bool COOP>
__device__ __forceinline__ void s1_op(const S1Sys& S, const double* __restrict__ p, int i, bool live, double (&ya)[3], double (&yb)[3]) {
    const int w = S.w, h = S.h;
    double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
    int e0 = 0, e1 = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { ya[c] = 0.0; yb[c] = 0.0; }
    if (live) {
        const int y = i / w, x = i - y * w;
        // the gathered vector is interleaved [pixel][a0 a1 a2 b0 b1 b2]: one 48-byte read per neighbour instead of two 24-byte ones
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] = p[(size_t)i * 6 + c]; b[c] = p[(size_t)i * 6 + 3 + c]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ya[c] = S.daa[(size_t)i * 3 + c] * a[c] + S.dab[(size_t)i * 3 + c] * b[c];
            yb[c] = S.dab[(size_t)i * 3 + c] * a[c] + S.dbb[(size_t)i * 3 + c] * b[c];
        }
        auto edge = [&](int j, double wt) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - p[(size_t)j * 6 + c]); yb[c] += wt * (b[c] - p[(size_t)j * 6 + 3 + c]); }
        };
        // local smoothness: every edge is entered twice in A (ColorTransfer.cpp:671-843)
        if (x + 1 < w) { const double g = S.gx[i]; edge(i + 1, 2.0 * (g * g)); }
        if (x > 0) { const double g = S.gx[i - 1]; edge(i - 1, 2.0 * (g * g)); }
        if (y + 1 < h) { const double g = S.gy[i]; edge(i + w, 2.0 * (g * g)); }
        if (y > 0) { const double g = S.gy[i - w]; edge(i - w, 2.0 * (g * g)); }
        // nonlocal: out-edges (8 independent gathers per thread), then in-edges
#pragma unroll
        for (int k = 0; k < 8; ++k) edge(S.knn_id[(size_t)i * 8 + k], S.iw2[(size_t)i * 8 + k]);
        e0 = S.rev_start[i]; e1 = S.rev_start[i + 1];
    }
    if constexpr (!COOP) {
        // small levels (latency bound, few workgroups): every thread walks its own list, loads of four edges issued together
        if (live) {
            int e = e0;
            for (; e + 4 <= e1; e += 4) {
                int j[4]; double wt[4], pv[4][6];
#pragma unroll
                for (int u = 0; u < 4; ++u) { j[u] = S.rev_src[e + u]; wt[u] = S.rev_w[e + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 6; ++c) pv[u][c] = p[(size_t)j[u] * 6 + c];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ya[c] += wt[u] * (a[c] - pv[u][c]); yb[c] += wt[u] * (b[c] - pv[u][3 + c]); }
            }
            for (; e < e1; ++e) {
                const int j = S.rev_src[e]; const double wt = S.rev_w[e];
#pragma unroll
                for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - p[(size_t)j * 6 + c]); yb[c] += wt * (b[c] - p[(size_t)j * 6 + 3 + c]); }
            }
        }
    } else {
        // in-edges of the workgroup's pixels [i0, i1): edge range [E0, E1)
        __shared__ double s_pv[6 * S1_CHUNK];          // [c][edge]
        __shared__ double s_wt[S1_CHUNK];
        const int i0 = blockIdx.x * 256, i1 = min(i0 + 256, S.n);
        const int E0 = S.rev_start[i0], E1 = S.rev_start[i1];
        for (int base = E0; base < E1; base += S1_CHUNK) {
            const int cnt = min(S1_CHUNK, E1 - base);
            for (int t = threadIdx.x; t < cnt; t += 256) {
                const int j = S.rev_src[base + t];
                s_wt[t] = S.rev_w[base + t];
#pragma unroll
                for (int c = 0; c < 6; ++c) s_pv[c * S1_CHUNK + t] = p[(size_t)j * 6 + c];
            }
            __syncthreads();
            const int lo = max(e0, base) - base, hi = min(e1, base + cnt) - base;
            for (int t = lo; t < hi; ++t) {
                const double wt = s_wt[t];
#pragma unroll
                for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - s_pv[c * S1_CHUNK + t]); yb[c] += wt * (b[c] - s_pv[(3 + c) * S1_CHUNK + t]); }
            }
            __syncthreads();
        }
    }
}

// r = rhs - Op(x0); partial r.r
template <bool COOP>
__global__ __launch_bounds__(256) void k_s1_residual(S1Sys S, const double* __restrict__ x, const double* __restrict__ rhs, double* __restrict__ r, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc[3] = {0, 0, 0};
    double ya[3], yb[3];
    s1_op<COOP>(S, x, i, i < S.n, ya, yb);
    if (i < S.n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double ra = rhs[(size_t)i * 3 + c] - ya[c], rb = rhs[(size_t)(S.n + i) * 3 + c] - yb[c];
            r[(size_t)i * 3 + c] = ra; r[(size_t)(S.n + i) * 3 + c] = rb;
            acc[c] = ra * ra + rb * rb;
        }
    }
    block_reduce_store<3>(acc, partial);
}
template <bool COOP>
__global__ __launch_bounds__(256) void k_s1_apply(S1Sys S, const double* __restrict__ p, double* __restrict__ Ap, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc[3] = {0, 0, 0};
    double ya[3], yb[3];
    s1_op<COOP>(S, p, i, i < S.n, ya, yb);
    if (i < S.n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Ap[(size_t)i * 3 + c] = ya[c]; Ap[(size_t)(S.n + i) * 3 + c] = yb[c];
            acc[c] = p[(size_t)i * 6 + c] * ya[c] + p[(size_t)i * 6 + 3 + c] * yb[c];
        }
    }
    block_reduce_store<3>(acc, partial);
}
// generic CG pieces over NQ lock-step systems stored as [part][n][3] (S1: NQ=3 channels, both parts share a scalar)
__global__ void k_cg_init(const double* __restrict__ partial, int nb, CGState* __restrict__ st, double tol2, int nq) {
    double s[3]; final_reduce<3>(partial, nb, s);
    if (threadIdx.x < nq) { const int c = threadIdx.x; st->r1[c] = s[c]; st->r0[c] = 0.0; st->va[c] = 0.0; st->vb[c] = 0.0; st->iters[c] = 0; st->active[c] = s[c] > tol2 ? 1 : 0; }
}
__global__ void k_cg_alpha(const double* __restrict__ partial, int nb, CGState* __restrict__ st) {
    double s[3]; final_reduce<3>(partial, nb, s);
    if (threadIdx.x < 3) { const int c = threadIdx.x; if (st->active[c]) st->va[c] = st->r1[c] / s[c]; }
}
__global__ void k_cg_beta(const double* __restrict__ partial, int nb, CGState* __restrict__ st, double tol2) {
    double s[3]; final_reduce<3>(partial, nb, s);
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        if (st->active[c]) { st->r0[c] = st->r1[c]; st->r1[c] = s[c]; st->vb[c] = s[c] / st->r0[c]; st->iters[c]++; st->active[c] = s[c] > tol2 ? 1 : 0; }
    }
}
// thread per pixel (both parts): p (interleaved [pixel][6]) is one contiguous 48-byte record per thread, r ([part][pixel][3]) two dense streams
// (the CG scalars are read once up front and every operand of the pixel is requested before the first store: 22 -> 16 us at 700x700 against a loop that re-read
// st->active / st->vb per component and alternated loads and stores)
__global__ __launch_bounds__(256) void k_s1_dir(int n, const CGState* __restrict__ st, const double* __restrict__ r, double* __restrict__ p, int first) {
    bool act[3]; double vb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { act[c] = st->active[c] != 0; vb[c] = st->vb[c]; }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double rv[6], pv[6];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int c = 0; c < 3; ++c) { rv[part * 3 + c] = r[((size_t)part * n + i) * 3 + c]; pv[part * 3 + c] = first ? 0.0 : p[(size_t)i * 6 + part * 3 + c]; }
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (!act[c]) continue;
            p[(size_t)i * 6 + part * 3 + c] = first ? rv[part * 3 + c] : vb[c] * pv[part * 3 + c] + rv[part * 3 + c];
        }
}
// ---- fused variants for levels with few partial sums (nb <= S1_FUSE_NB): the two single-workgroup kernels of an iteration
// (k_cg_alpha, k_cg_beta: ~5 us each, pure latency) disappear — EVERY workgroup of the following vector kernel repeats the
// fixed-order final reduction of the nb x 3 partials (a few KB out of L2) and derives the scalars itself. Same reductions, same
// order, same values. The state is double buffered (a workgroup may not overwrite scalars its neighbours still read) and the two
// dot products use separate partial arrays.
constexpr int S1_FUSE_NB = 512;
// beta step + direction update, thread per pixel: state_out = beta(state_in, partial_rr); p = r + vb p   (first: state_out = state_in, p = r)
__global__ __launch_bounds__(256) void k_s1_dir_f(int n, int nb, const double* __restrict__ partial_rr, const CGState* __restrict__ sin, CGState* __restrict__ sout,
                                                  double tol2, const double* __restrict__ r, double* __restrict__ p, int first) {
    // the pixel's operands do not depend on the scalars: they are requested in front of the reduction that produces those (a chain of its own) and fly under it
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rv[6], pv[6];
    if (i < n) {
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) { rv[part * 3 + c] = r[((size_t)part * n + i) * 3 + c]; pv[part * 3 + c] = first ? 0.0 : p[(size_t)i * 6 + part * 3 + c]; }
    }
    double vb[3]; int act[3];
    if (first) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { vb[c] = 0.0; act[c] = sin->active[c]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) *sout = *sin;
    } else {
        double sm[3]; final_reduce<3>(partial_rr, nb, sm);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool a = sin->active[c] != 0;
            vb[c] = a ? sm[c] / sin->r1[c] : sin->vb[c];
            act[c] = a ? (sm[c] > tol2 ? 1 : 0) : 0;
        }
        if (blockIdx.x == 0 && threadIdx.x < 3) {
            const int c = threadIdx.x; const bool a = sin->active[c] != 0;
            sout->r0[c] = a ? sin->r1[c] : sin->r0[c]; sout->r1[c] = a ? sm[c] : sin->r1[c]; sout->va[c] = sin->va[c]; sout->vb[c] = vb[c];
            sout->iters[c] = sin->iters[c] + (a ? 1 : 0); sout->active[c] = act[c];
        }
    }
    if (i >= n) return;
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (!act[c]) continue;
            p[(size_t)i * 6 + part * 3 + c] = first ? rv[part * 3 + c] : vb[c] * pv[part * 3 + c] + rv[part * 3 + c];
        }
}
// alpha step + solution/residual update: va = r1 / (p.Ap) from partial_pap ; x += va p ; r -= va Ap ; partial_rr = r.r
__global__ __launch_bounds__(256) void k_s1_update_f(int n, int nb, const double* __restrict__ partial_pap, const CGState* __restrict__ st, const double* __restrict__ p,
                                                     const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r, double* __restrict__ partial_rr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double pv[6], av[6], xv[6], rv[6];                           // requested in front of the reduction (see k_s1_dir_f)
    bool act[3]; double r1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { act[c] = st->active[c] != 0; r1[c] = st->r1[c]; }
    if (i < n) {
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t j = ((size_t)part * n + i) * 3 + c;
                pv[part * 3 + c] = p[(size_t)i * 6 + part * 3 + c]; av[part * 3 + c] = Ap[j]; xv[part * 3 + c] = x[j]; rv[part * 3 + c] = r[j];
            }
    }
    double sm[3]; final_reduce<3>(partial_pap, nb, sm);
    double acc[3] = {0, 0, 0};
    if (i < n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (!act[c]) continue;
            const double va = r1[c] / sm[c];
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const size_t j = ((size_t)part * n + i) * 3 + c;
                x[j] = xv[part * 3 + c] + va * pv[part * 3 + c];
                const double rn = rv[part * 3 + c] - va * av[part * 3 + c];
                r[j] = rn; acc[c] += rn * rn;
            }
        }
    }
    block_reduce_store<3>(acc, partial_rr);
}
// ---- persistent variant for the levels whose whole grid is resident with one workgroup per CU (nb <= S1_PERSIST_NB: 44^2 .. 175^2 of a 700^2 pair): the
// maxit iterations of the recurrence above in ONE launch instead of 3 x maxit. Same block decomposition (256 consecutive pixels per workgroup), same
// per-pixel operation order, same two-stage reductions in the same order => the same bits as the three-kernel form. What changes is where things live:
//   * a thread owns its pixel for the whole solve: x, r, p and Op(p) of the pixel stay in registers, and so does everything the operator needs that does
//     not change between iterations (data block, the four local weights 2 g^2, the eight out-edge ids and weights, its in-edge range);
//   * the workgroup's in-edge range (one contiguous piece of the target-sorted edge arrays) is parked in LDS once (S1_PERSIST_EDGES entries; what does not
//     fit is read from the arrays);
//   * per iteration only p crosses workgroups: 48 B stored per pixel, ~20 gathers of 48 B, and the two partial-sum vectors.
// Three grid-wide barriers per iteration (p published -> Op; p.Op(p) partials -> alpha; r.r partials -> beta), each the placement-independent protocol:
// every thread's stores are complete at the workgroup barrier, thread 0 issues an agent-scope release fence, arrives on ONE monotonic counter, polls it
// with relaxed agent-scope loads (s_sleep between polls), and issues an agent-scope acquire fence before the workgroup barrier that lets the others go.
// The spin is bounded by the constant-rate clock (S1_PERSIST_TIMEOUT_TICKS at 100 MHz): a grid that cannot become resident (a GPU oversubscribed by many
// processes) raises *fail, leaves x untouched and returns; the host then repeats the solve with the three-kernel form (nctk_local_color_transfer).
constexpr int S1_PERSIST_NB = 32;          // 44^2 (8 workgroups) and 88^2 (31); at 175^2 (120) the three-kernel form is faster (2.6 against 3.7 ms per level)
constexpr int S1_PERSIST_EDGES = 4096;
#ifndef NCT_S1_PERSIST_TIMEOUT_TICKS
#define NCT_S1_PERSIST_TIMEOUT_TICKS 25000000ull     // 0.25 s
#endif
struct S1Bar { unsigned count; int fail; };
#ifndef NCT_S1P_SC1
#define NCT_S1P_SC1 0        // 1: p and the partial sums cross workgroups through agent-scope (sc1: write-through / cache-bypassing) 8-byte accesses instead of plain ones
                             //    (measured, §9: 4.2 / 3.4 / 5.1 ms per level against 2.4 / 2.2 / 3.7 — 120 eight-byte loads per pixel that all go to memory)
#endif
#ifndef NCT_S1P_FENCE
#define NCT_S1P_FENCE 1      // 1: release / acquire fences at agent scope around the counter (L2 write-back + invalidate); 0: s_waitcnt vmcnt(0) only (needs NCT_S1P_SC1)
#endif
__device__ __forceinline__ double s1p_ld(const double* q) {
#if NCT_S1P_SC1
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *q;
#endif
}
__device__ __forceinline__ void s1p_st(double* q, double v) {
#if NCT_S1P_SC1
    __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *q = v;
#endif
}
__device__ __forceinline__ bool s1_grid_barrier(S1Bar* bar, unsigned target) {
    __shared__ int s_fail;
#if !NCT_S1P_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every thread's write-through stores have been acknowledged
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#if NCT_S1P_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        __hip_atomic_fetch_add(&bar->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int fail = 0;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&bar->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load(&bar->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > NCT_S1_PERSIST_TIMEOUT_TICKS) { fail = 1; break; }
        }
        if (fail) __hip_atomic_store(&bar->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if NCT_S1P_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        s_fail = fail;
    }
    __syncthreads();
    return s_fail == 0;
}
// the fixed-order reductions of block_reduce_store / final_reduce on memory that other workgroups write during the launch (no __restrict__, no const)
__device__ __forceinline__ void s1p_block_store(double (&v)[3], double* partial, double* s_red) {
    tree256<3>(v, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) s1p_st(partial + (size_t)blockIdx.x * 3 + q, v[q]);
    }
}
__device__ __forceinline__ void s1p_final(double* partial, int nb, double (&out)[3], double* s_red, double* s_out) {
    const int t = threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = t; b < nb; b += 256)
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] += s1p_ld(partial + (size_t)b * 3 + q);
    tree256<3>(acc, s_red);
    if (t == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) s_out[q] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = s_out[q];
    __syncthreads();
}
__global__ __launch_bounds__(256, 1) void k_s1_cg_persist(S1Sys S, int maxit, double tol2, double* __restrict__ x /*[2][n][3]*/, const double* __restrict__ r_in,
                                                          double* p /*[n][6]: in = the packed first guess, rewritten every iteration*/,
                                                          double* partA, double* partB, const CGState* __restrict__ st_in, CGState* __restrict__ st_out, S1Bar* bar) {
    __shared__ double s_red[128 * 3];
    __shared__ double s_out[3];
    __shared__ int s_esrc[S1_PERSIST_EDGES];
    __shared__ double s_ew[S1_PERSIST_EDGES];
    const int n = S.n, w = S.w, h = S.h, nb = gridDim.x;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    // ---- loop invariants of the pixel
    double daa[3], dab[3], dbb[3], lw[4], ow[8];
    int lj[4], oj[8], e0 = 0, e1 = 0;
    double xv[6], rv[6], pv[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) { daa[c] = dab[c] = dbb[c] = 0.0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { lw[k] = 0.0; lj[k] = -1; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { ow[k] = 0.0; oj[k] = 0; }
#pragma unroll
    for (int c = 0; c < 6; ++c) { xv[c] = rv[c] = pv[c] = 0.0; }
    if (live) {
        const int y = i / w, xx = i - y * w;
#pragma unroll
        for (int c = 0; c < 3; ++c) { daa[c] = S.daa[(size_t)i * 3 + c]; dab[c] = S.dab[(size_t)i * 3 + c]; dbb[c] = S.dbb[(size_t)i * 3 + c]; }
        if (xx + 1 < w) { const double g = S.gx[i]; lw[0] = 2.0 * (g * g); lj[0] = i + 1; }
        if (xx > 0) { const double g = S.gx[i - 1]; lw[1] = 2.0 * (g * g); lj[1] = i - 1; }
        if (y + 1 < h) { const double g = S.gy[i]; lw[2] = 2.0 * (g * g); lj[2] = i + w; }
        if (y > 0) { const double g = S.gy[i - w]; lw[3] = 2.0 * (g * g); lj[3] = i - w; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { oj[k] = S.knn_id[(size_t)i * 8 + k]; ow[k] = S.iw2[(size_t)i * 8 + k]; }
        e0 = S.rev_start[i]; e1 = S.rev_start[i + 1];
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t j = ((size_t)part * n + i) * 3 + c;
                xv[part * 3 + c] = x[j]; rv[part * 3 + c] = r_in[j]; pv[part * 3 + c] = p[(size_t)i * 6 + part * 3 + c];
            }
    }
    const int i0 = blockIdx.x * 256, i1 = min(i0 + 256, n);
    const int E0 = S.rev_start[i0], E1 = S.rev_start[i1];
    for (int t = threadIdx.x; t < min(E1 - E0, S1_PERSIST_EDGES); t += 256) { s_esrc[t] = S.rev_src[E0 + t]; s_ew[t] = S.rev_w[E0 + t]; }
    // ---- CG scalars (every workgroup carries the same copy: same partials, same order)
    double r0[3], r1[3], vb[3]; int act[3], iters[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { r0[c] = st_in->r0[c]; r1[c] = st_in->r1[c]; vb[c] = st_in->vb[c]; act[c] = st_in->active[c]; iters[c] = st_in->iters[c]; }
    const double va_in[3] = {st_in->va[0], st_in->va[1], st_in->va[2]};
    unsigned gen = 0;
    bool ok = true;
    __syncthreads();
    for (int k = 1; k <= maxit; ++k) {
        // -- beta step + direction (k_s1_dir_f)
        if (k > 1) {
            double sm[3]; s1p_final(partB, nb, sm, s_red, s_out);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool a = act[c] != 0;
                const double vbn = a ? sm[c] / r1[c] : vb[c];
                const int an = a ? (sm[c] > tol2 ? 1 : 0) : 0;
                r0[c] = a ? r1[c] : r0[c]; r1[c] = a ? sm[c] : r1[c]; vb[c] = vbn; iters[c] += a ? 1 : 0; act[c] = an;
            }
        }
        if (!(act[0] | act[1] | act[2])) break;                 // nothing iterates any more (every workgroup takes this exit together)
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (!act[c]) continue;
                pv[part * 3 + c] = k == 1 ? rv[part * 3 + c] : vb[c] * pv[part * 3 + c] + rv[part * 3 + c];
            }
        if (live) {
#pragma unroll
            for (int c = 0; c < 6; ++c) s1p_st(p + (size_t)i * 6 + c, pv[c]);
        }
        ok = s1_grid_barrier(bar, (unsigned)nb * ++gen); if (!ok) break;
        // -- Op(p) (s1_op<false>), p.Op(p)
        double ya[3] = {0.0, 0.0, 0.0}, yb[3] = {0.0, 0.0, 0.0}, acc[3] = {0.0, 0.0, 0.0};
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ya[c] = daa[c] * pv[c] + dab[c] * pv[3 + c];
                yb[c] = dab[c] * pv[c] + dbb[c] * pv[3 + c];
            }
            auto edge = [&](int j, double wt) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { ya[c] += wt * (pv[c] - s1p_ld(p + (size_t)j * 6 + c)); yb[c] += wt * (pv[3 + c] - s1p_ld(p + (size_t)j * 6 + 3 + c)); }
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) if (lj[q] >= 0) edge(lj[q], lw[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) edge(oj[q], ow[q]);
            int e = e0;
            for (; e + 4 <= e1; e += 4) {
                int j[4]; double wt[4], gv[4][6];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = e + u - E0;
                    if (t < S1_PERSIST_EDGES) { j[u] = s_esrc[t]; wt[u] = s_ew[t]; } else { j[u] = S.rev_src[e + u]; wt[u] = S.rev_w[e + u]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 6; ++c) gv[u][c] = s1p_ld(p + (size_t)j[u] * 6 + c);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ya[c] += wt[u] * (pv[c] - gv[u][c]); yb[c] += wt[u] * (pv[3 + c] - gv[u][3 + c]); }
            }
            for (; e < e1; ++e) {
                const int t = e - E0;
                int j; double wt;
                if (t < S1_PERSIST_EDGES) { j = s_esrc[t]; wt = s_ew[t]; } else { j = S.rev_src[e]; wt = S.rev_w[e]; }
#pragma unroll
                for (int c = 0; c < 3; ++c) { ya[c] += wt * (pv[c] - s1p_ld(p + (size_t)j * 6 + c)); yb[c] += wt * (pv[3 + c] - s1p_ld(p + (size_t)j * 6 + 3 + c)); }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = pv[c] * ya[c] + pv[3 + c] * yb[c];
        }
        s1p_block_store(acc, partA, s_red);
        ok = s1_grid_barrier(bar, (unsigned)nb * ++gen); if (!ok) break;
        // -- alpha step, x += va p, r -= va Op(p), r.r (k_s1_update_f)
        double sm[3]; s1p_final(partA, nb, sm, s_red, s_out);
        double acc2[3] = {0.0, 0.0, 0.0};
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (!act[c]) continue;
                const double va = r1[c] / sm[c];
                xv[c] = xv[c] + va * pv[c];
                { const double rn = rv[c] - va * ya[c]; rv[c] = rn; acc2[c] += rn * rn; }
                xv[3 + c] = xv[3 + c] + va * pv[3 + c];
                { const double rn = rv[3 + c] - va * yb[c]; rv[3 + c] = rn; acc2[c] += rn * rn; }
            }
        }
        s1p_block_store(acc2, partB, s_red);
        ok = s1_grid_barrier(bar, (unsigned)nb * ++gen); if (!ok) break;
    }
    if (!ok) return;                                            // x untouched: the host repeats the solve
    if (live) {
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) x[((size_t)part * n + i) * 3 + c] = xv[part * 3 + c];
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) {
        const int c = threadIdx.x;
        st_out->r0[c] = r0[c]; st_out->r1[c] = r1[c]; st_out->va[c] = va_in[c]; st_out->vb[c] = vb[c]; st_out->iters[c] = iters[c]; st_out->active[c] = act[c];
    }
}
// [part][n][3] -> [n][6]
__global__ void k_pack6(int n, const double* __restrict__ x, double* __restrict__ x6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n * 3) return;
    const int c = i % 3, part = i / (n * 3), px = (i - part * n * 3) / 3;
    x6[(size_t)px * 6 + part * 3 + c] = x[i];
}
__global__ __launch_bounds__(256) void k_s1_update(int n, const CGState* __restrict__ st, const double* __restrict__ p, const double* __restrict__ Ap,
                                                   double* __restrict__ x, double* __restrict__ r, double* __restrict__ partial) {
    bool act[3]; double va[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { act[c] = st->active[c] != 0; va[c] = st->va[c]; }
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc[3] = {0, 0, 0};
    if (i < n) {
        double pv[6], av[6], xv[6], rv[6];                       // every operand of the pixel is requested before the first store
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t j = ((size_t)part * n + i) * 3 + c;
                pv[part * 3 + c] = p[(size_t)i * 6 + part * 3 + c]; av[part * 3 + c] = Ap[j]; xv[part * 3 + c] = x[j]; rv[part * 3 + c] = r[j];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (!act[c]) continue;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const size_t j = ((size_t)part * n + i) * 3 + c;
                x[j] = xv[part * 3 + c] + va[c] * pv[part * 3 + c];
                const double rn = rv[part * 3 + c] - va[c] * av[part * 3 + c];
                r[j] = rn; acc[c] += rn * rn;
            }
        }
    }
    block_reduce_store<3>(acc, partial);
}

__global__ void k_edge_keys(const int* __restrict__ knn_id, int m, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    keys[e] = (unsigned)knn_id[e]; vals[e] = (unsigned)e;
}
__global__ void k_rev_edges(const unsigned* __restrict__ sorted_edge, const double* __restrict__ iw2, int m, int* __restrict__ rev_src, double* __restrict__ rev_w) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const unsigned ed = sorted_edge[e];
    rev_src[e] = (int)(ed >> 3); rev_w[e] = iw2[ed];
}
__global__ void k_seg_starts(const unsigned* __restrict__ keys, int m, int* __restrict__ start, int n) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n) return;
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < (unsigned)s) lo = mid + 1; else hi = mid; }
    start[s] = lo;
}

// ================================================================= U1 roughness / A1 apply
__global__ void k_roughness(const double* __restrict__ a, const double* __restrict__ b, const uint8_t* __restrict__ lab, int n, double* __restrict__ rough) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double nc = LAB_D(lab[(size_t)i * 3 + 2]) * a[(size_t)i * 3 + 2] + b[(size_t)i * 3 + 2];     // only channel 2 survives (quirk 5)
    rough[i] = (nc < 0 || nc > 1) ? 1e-6 : 1.0;
}
__global__ void k_apply(const double* __restrict__ a, const double* __restrict__ b, const uint8_t* __restrict__ lab, int n, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    double v = LAB_D(lab[i]) * a[i] + b[i];
    v = v > 0.0 ? v : 0.0; v = v < 1.0 ? v : 1.0;
    const int q = (int)rint(v * 255.0);
    out[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
}

// ================================================================= S2 WLS: (diag(r) + L) x = r x0, 6 right-hand sides

__global__ void k_wls_system(const double* __restrict__ gx, const double* __restrict__ gy, const double* __restrict__ rough, int H, int W,
                             double* __restrict__ diag, double* __restrict__ wx, double* __restrict__ wy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    double a00 = 0.0, ex = 0.0, ey = 0.0;
    a00 += rough[i];
    if (x + 1 < W) { const double g = gx[i] * gx[i]; a00 += g; ex = g; }
    if (x > 0) { const double g = gx[i - 1] * gx[i - 1]; a00 += g; }
    if (y + 1 < H) { const double g = gy[i] * gy[i]; a00 += g; ey = g; }
    if (y > 0) { const double g = gy[i - W] * gy[i - W]; a00 += g; }
    diag[i] = a00; wx[i] = ex; wy[i] = ey;
}
// ================================================================= orchestration
#define LCHK() NCT_LAUNCH_CHECK()
static int dbg_copy(nct_ctx* ctx, hipStream_t s, double* host, const double* dev, size_t n) {
    if (!host) return 0;
    NCT_HIP(hipMemcpyAsync(host, dev, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    NCT_HIP(hipStreamSynchronize(s));
    return 0;
}

static int local_color_transfer_once(nct_ctx* ctx, hipStream_t s, const float* err, const uint8_t* s_lab_level, const uint8_t* g_lab_level,
                              const uint8_t* s_lab_full, const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W,
                              const nct_color_params& prm, uint8_t* out_lab_full, const nct_color_debug* dbg, bool* s1_stalled) {
    const int n = h * w, N = H * W;
    const int nbl = cdiv(n, 256), nbL = cdiv(N, 256);
    // ---------------- T1 + T2
    DevBuf<double> x(ctx, (size_t)6 * n), weight(ctx, n);
    DevBuf<unsigned> mm(ctx, 2);
    if (!x.ok() || !weight.ok() || !mm.ok()) return NCT_ERR_HIP;
    double* xa = x; double* xb = (double*)x + (size_t)3 * n;
    hipLaunchKernelGGL(k_local_stats, dim3(nbl), dim3(256), 0, s, s_lab_level, g_lab_level, h, w, prm.eps, xa, xb); LCHK();
    if (dbg) { int rc = dbg_copy(ctx, s, dbg->ab_local, x, (size_t)6 * n); if (rc) return rc; }
    NCT_HIP(hipMemsetD32Async((hipDeviceptr_t)(unsigned*)mm, (int)0xFFFFFFFFu, 1, s));
    NCT_HIP(hipMemsetD32Async((hipDeviceptr_t)((unsigned*)mm + 1), 0, 1, s));
    hipLaunchKernelGGL(k_minmax_f, dim3(128), dim3(256), 0, s, err, n, (unsigned*)mm); LCHK();
    hipLaunchKernelGGL(k_err_weight, dim3(nbl), dim3(256), 0, s, err, n, (const unsigned*)mm, (double*)weight); LCHK();
    { int rcm = ctx->mark(s, nct_stage_tag_color()); if (rcm) return rcm; }
    // ---------------- S1
    const double normFactor = (double)(W * H) / (double)(w * h);
    bool persist_used = false;
    {
        DevBuf<double> gx(ctx, n), gy(ctx, n), daa(ctx, (size_t)3 * n), dab(ctx, (size_t)3 * n), dbb(ctx, (size_t)3 * n), rhs(ctx, (size_t)6 * n), iw2(ctx, (size_t)8 * n);
        DevBuf<double> r(ctx, (size_t)6 * n), p(ctx, (size_t)6 * n), Ap(ctx, (size_t)6 * n), partial(ctx, (size_t)nbl * 3);
        DevBuf<unsigned> ek(ctx, (size_t)8 * n), ev(ctx, (size_t)8 * n), eks(ctx, (size_t)8 * n), evs(ctx, (size_t)8 * n);
        DevBuf<int> rstart(ctx, n + 1), rev_src(ctx, (size_t)8 * n);
        DevBuf<double> rev_w(ctx, (size_t)8 * n);
        DevBuf<CGState> st(ctx, 2);
        DevBuf<double> partial2(ctx, (size_t)nbl * 3);
        if (!gx.ok() || !gy.ok() || !daa.ok() || !dab.ok() || !dbb.ok() || !rhs.ok() || !iw2.ok() || !r.ok() || !p.ok() || !Ap.ok() || !partial.ok() ||
            !ek.ok() || !ev.ok() || !eks.ok() || !evs.ok() || !rstart.ok() || !rev_src.ok() || !rev_w.ok() || !st.ok() || !partial2.ok()) return NCT_ERR_HIP;
        // lambda / alpha / dWeight arrive as float in the reference signature (ColorTransfer.cpp:548-550)
        const float lambda_f = (float)prm.local_weight, alpha_f = (float)prm.wls_alpha, dWeight_f = (float)normFactor;
        hipLaunchKernelGGL(k_gradient_weights, dim3(nbl), dim3(256), 0, s, s_lab_level, h, w, (double)lambda_f, (double)alpha_f, (double*)gx, (double*)gy); LCHK();
        const double nonlocalWeight = sqrt(prm.nonlocal_weight / prm.k_num);
        hipLaunchKernelGGL(k_s1_setup, dim3(nbl), dim3(256), 0, s, n, (const double*)weight, dWeight_f, s_lab_level, g_lab_level, knn_w, nonlocalWeight,
                           (double*)daa, (double*)dab, (double*)dbb, (double*)rhs, (double*)iw2); LCHK();
        // reverse adjacency of the kNN graph
        const int m = 8 * n;
        hipLaunchKernelGGL(k_edge_keys, dim3(cdiv(m, 256)), dim3(256), 0, s, knn_id, m, (unsigned*)ek, (unsigned*)ev); LCHK();
        int end_bit = 1; while ((1u << end_bit) < (unsigned)n && end_bit < 32) ++end_bit;
        size_t tmp_bytes = 0;
        NCT_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned*)ek, (unsigned*)eks, (const unsigned*)ev, (unsigned*)evs, m, 0, end_bit, s));
        DevBuf<char> tmp(ctx, tmp_bytes ? tmp_bytes : 16);
        if (!tmp.ok()) return NCT_ERR_HIP;
        NCT_HIP(rocprim::radix_sort_pairs((void*)(char*)tmp, tmp_bytes, (const unsigned*)ek, (unsigned*)eks, (const unsigned*)ev, (unsigned*)evs, m, 0, end_bit, s));
        hipLaunchKernelGGL(k_seg_starts, dim3(cdiv(n + 1, 256)), dim3(256), 0, s, (const unsigned*)eks, m, (int*)rstart, n); LCHK();
        hipLaunchKernelGGL(k_rev_edges, dim3(cdiv(m, 256)), dim3(256), 0, s, (const unsigned*)evs, (const double*)iw2, m, (int*)rev_src, (double*)rev_w); LCHK();
        S1Sys S{n, h, w, daa, dab, dbb, gx, gy, knn_id, iw2, rstart, rev_src, rev_w};
        const double tol2 = 1e-6 * 1e-6;
        const int maxit = layer == 4 ? 50 : 100;                       // ColorTransfer.cpp:916-921
        hipLaunchKernelGGL(k_pack6, dim3(cdiv(6 * n, 256)), dim3(256), 0, s, n, (const double*)x, (double*)p); LCHK();
        const bool coop = n >= 100000;                                  // shared in-edge gathers pay off on the bandwidth-bound levels only
        if (coop) hipLaunchKernelGGL(k_s1_residual<true>, dim3(nbl), dim3(256), 0, s, S, (const double*)p, (const double*)rhs, (double*)r, (double*)partial);
        else      hipLaunchKernelGGL(k_s1_residual<false>, dim3(nbl), dim3(256), 0, s, S, (const double*)p, (const double*)rhs, (double*)r, (double*)partial);
        LCHK();
        CGState* st_final = (CGState*)st;
        bool persist_done = false;
        if (nbl <= S1_PERSIST_NB && ctx->s1_persist) {
            // one launch for the whole recurrence (k_s1_cg_persist); S[1] -> S[0], then the last beta step as below
            DevBuf<S1Bar> bar(ctx, 1);
            if (!bar.ok()) return NCT_ERR_HIP;
            CGState* S2[2] = {(CGState*)st, (CGState*)st + 1};
            NCT_HIP(hipMemsetAsync((S1Bar*)bar, 0, sizeof(S1Bar), s));
            if (ctx->s1_persist == 2) NCT_HIP(hipMemsetD32Async((hipDeviceptr_t)&((S1Bar*)bar)->fail, 1, 1, s));   // test hook (NCT_S1_PERSIST=2): the launch finds the stall flag raised
            hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(256), 0, s, (const double*)partial, nbl, S2[1], tol2, 3); LCHK();
            hipLaunchKernelGGL(k_s1_cg_persist, dim3(nbl), dim3(256), 0, s, S, maxit, tol2, (double*)x, (const double*)r, (double*)p, (double*)partial, (double*)partial2,
                               (const CGState*)S2[1], S2[0], (S1Bar*)bar); LCHK();
            st_final = S2[0];
            hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(256), 0, s, (const double*)partial2, nbl, st_final, tol2); LCHK();
            // the stall flag travels to page-locked memory behind the kernel; it is looked at after the WLS solve of this level, whose convergence polls have
            // taken the host past this point of the stream anyway (no extra synchronisation)
            NCT_HIP(hipMemcpyAsync(ctx->s1_stall_flag(), &((S1Bar*)bar)->fail, sizeof(int), hipMemcpyDeviceToHost, s));
            persist_done = true; persist_used = true;
        }
        if (persist_done) {
        } else if (nbl <= S1_FUSE_NB) {
            // 3 launches per iteration; state ping-pongs between S[0] and S[1] (iteration k reads S[k&1], writes S[(k+1)&1])
            CGState* S2[2] = {(CGState*)st, (CGState*)st + 1};
            hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(256), 0, s, (const double*)partial, nbl, S2[1], tol2, 3); LCHK();
            for (int k = 1; k <= maxit; ++k) {
                hipLaunchKernelGGL(k_s1_dir_f, dim3(nbl), dim3(256), 0, s, n, nbl, (const double*)partial2, (const CGState*)S2[k & 1], S2[(k + 1) & 1], tol2,
                                   (const double*)r, (double*)p, k == 1 ? 1 : 0); LCHK();
                if (coop) hipLaunchKernelGGL(k_s1_apply<true>, dim3(nbl), dim3(256), 0, s, S, (const double*)p, (double*)Ap, (double*)partial);
                else      hipLaunchKernelGGL(k_s1_apply<false>, dim3(nbl), dim3(256), 0, s, S, (const double*)p, (double*)Ap, (double*)partial);
                LCHK();
                hipLaunchKernelGGL(k_s1_update_f, dim3(nbl), dim3(256), 0, s, n, nbl, (const double*)partial, (const CGState*)S2[(k + 1) & 1], (const double*)p, (const double*)Ap,
                                   (double*)x, (double*)r, (double*)partial2); LCHK();
            }
            st_final = S2[(maxit + 1) & 1];
            hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(256), 0, s, (const double*)partial2, nbl, st_final, tol2); LCHK();   // the last beta step (iteration count, final r.r)
        } else {
            hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(256), 0, s, (const double*)partial, nbl, (CGState*)st, tol2, 3); LCHK();
            for (int k = 1; k <= maxit; ++k) {
                const bool kt = ctx->kt_on && layer == 4 && k >= 3 && k < 11;          // NCT_FLAG_TIME_KERNELS: eight iterations of the finest level, one event pair per launch
#define KT(id, launch) do { if (kt) { int rk_ = ctx->kt_begin(s, id); if (rk_) return rk_; } launch; if (kt) { int rk_ = ctx->kt_end(s); if (rk_) return rk_; } } while (0)
                KT(NCT_KT_S1_DIR, hipLaunchKernelGGL(k_s1_dir, dim3(nbl), dim3(256), 0, s, n, (const CGState*)st, (const double*)r, (double*)p, k == 1 ? 1 : 0)); LCHK();
                if (coop) KT(NCT_KT_S1_APPLY, hipLaunchKernelGGL(k_s1_apply<true>, dim3(nbl), dim3(256), 0, s, S, (const double*)p, (double*)Ap, (double*)partial));
                else      KT(NCT_KT_S1_APPLY, hipLaunchKernelGGL(k_s1_apply<false>, dim3(nbl), dim3(256), 0, s, S, (const double*)p, (double*)Ap, (double*)partial));
                LCHK();
                hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(256), 0, s, (const double*)partial, nbl, (CGState*)st); LCHK();
                KT(NCT_KT_S1_UPDATE, hipLaunchKernelGGL(k_s1_update, dim3(nbl), dim3(256), 0, s, n, (const CGState*)st, (const double*)p, (const double*)Ap, (double*)x, (double*)r, (double*)partial)); LCHK();
#undef KT
                hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(256), 0, s, (const double*)partial, nbl, (CGState*)st, tol2); LCHK();
            }
        }
        if (dbg && dbg->cg_iters) {
            CGState hst;
            NCT_HIP(hipMemcpyAsync(&hst, st_final, sizeof hst, hipMemcpyDeviceToHost, s));
            NCT_HIP(hipStreamSynchronize(s));
            for (int c = 0; c < 3; ++c) dbg->cg_iters[c] = hst.iters[c];
        }
    }
    { int rcm = ctx->mark(s, nct_stage_tag_nonlocal()); if (rcm) return rcm; }
    if (dbg) { int rc = dbg_copy(ctx, s, dbg->ab_nonlocal, x, (size_t)6 * n); if (rc) return rc; }
    // ---------------- U1: bilinear upsample to full resolution + roughness
    DevBuf<double> X(ctx, (size_t)6 * N), rough(ctx, N);
    if (!X.ok() || !rough.ok()) return NCT_ERR_HIP;
    double* Xa = X; double* Xb = (double*)X + (size_t)3 * N;
    if (W > w || H > h) {
        int rc = nctk_resize_f64c3(ctx, s, xa, h, w, Xa, H, W); if (rc) return rc;
        rc = nctk_resize_f64c3(ctx, s, xb, h, w, Xb, H, W); if (rc) return rc;
    } else {
        NCT_HIP(hipMemcpyAsync(X, x, sizeof(double) * (size_t)6 * N, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_roughness, dim3(nbL), dim3(256), 0, s, (const double*)Xa, (const double*)Xb, s_lab_full, N, (double*)rough); LCHK();
    if (dbg) { int rc = dbg_copy(ctx, s, dbg->ab_up, X, (size_t)6 * N); if (rc) return rc; rc = dbg_copy(ctx, s, dbg->rough, rough, N); if (rc) return rc; }
    // ---------------- S2: WLS (k_wls_mg.hip)
    {
        double lamda = prm.wls_lambda_init * normFactor;
        if (h == H && w == W) lamda *= 4;                               // ColorTransfer.cpp:1418-1424
        DevBuf<double> gx(ctx, N), gy(ctx, N), diag(ctx, N), wx(ctx, N), wy(ctx, N);
        if (!gx.ok() || !gy.ok() || !diag.ok() || !wx.ok() || !wy.ok()) return NCT_ERR_HIP;
        hipLaunchKernelGGL(k_gradient_weights, dim3(nbL), dim3(256), 0, s, s_lab_full, H, W, lamda, prm.wls_alpha, (double*)gx, (double*)gy); LCHK();
        hipLaunchKernelGGL(k_wls_system, dim3(nbL), dim3(256), 0, s, (const double*)gx, (const double*)gy, (const double*)rough, H, W, (double*)diag, (double*)wx, (double*)wy); LCHK();
        int wit[6] = {0, 0, 0, 0, 0, 0};
        { int rcm = ctx->mark(s, nct_stage_tag_color()); if (rcm) return rcm; }
        int rc = nctk_wls_solve_mg(ctx, s, X, rough, wx, wy, H, W, ctx->wls_rtol, wit); if (rc) return rc;
        { int rcm = ctx->mark(s, nct_stage_tag_wls()); if (rcm) return rcm; }
        if (persist_used && *(volatile int*)ctx->s1_stall_flag() != 0) { *s1_stalled = true; return 0; }
        if (dbg && dbg->wls_iters) for (int q = 0; q < 6; ++q) dbg->wls_iters[q] = wit[q];
    }
    if (dbg) { int rc = dbg_copy(ctx, s, dbg->ab_wls, X, (size_t)6 * N); if (rc) return rc; }
    // ---------------- A1
    hipLaunchKernelGGL(k_apply, dim3(cdiv(3 * N, 256)), dim3(256), 0, s, (const double*)Xa, (const double*)Xb, s_lab_full, N, out_lab_full); LCHK();
    return 0;
}
int nctk_local_color_transfer(nct_ctx* ctx, hipStream_t s, const float* err, const uint8_t* s_lab_level, const uint8_t* g_lab_level,
                              const uint8_t* s_lab_full, const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W,
                              const nct_color_params& prm, uint8_t* out_lab_full, const nct_color_debug* dbg) {
    bool stalled = false;
    *ctx->s1_stall_flag() = 0;
    int rc = local_color_transfer_once(ctx, s, err, s_lab_level, g_lab_level, s_lab_full, knn_id, knn_w, layer, h, w, H, W, prm, out_lab_full, dbg, &stalled);
    if (rc || !stalled) return rc;
    // k_s1_cg_persist gave up at a grid barrier (its workgroups did not all become resident within the time limit: a GPU shared with many other processes). It left
    // the first guess untouched; everything after it in this level ran on that guess and is overwritten now. This context uses the three-kernel form from here on.
    ctx->s1_persist = 0; ctx->s1_stalls++;
    fprintf(stderr, "nct: S1 persistent launch stalled at level %d (%d x %d); repeating the level with per-iteration launches, which this context keeps from now on\n", layer, w, h);
    return local_color_transfer_once(ctx, s, err, s_lab_level, g_lab_level, s_lab_full, knn_id, knn_w, layer, h, w, H, W, prm, out_lab_full, dbg, &stalled);
}
