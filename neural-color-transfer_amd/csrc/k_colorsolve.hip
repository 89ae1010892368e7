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
#include "nct_reduce.h"
#include <cstring>
#include <cstdio>
#include <rocprim/device/device_radix_sort.hpp>   // rocPRIM directly (no CUB-compatibility layer)

#define LAB_D(u) ((double)(u) * (1.0 / 255.0))      // Mat::convertTo(CV_64F, 1/255)

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
    // a NaN (the matching error of a dead feature pixel: `norm` divides by a zero norm, GeneralizedPatchMatch.cu:276-277) takes part in neither extreme, exactly like
    // the reference's `if (err2 < minDist)` / `if (err2 > maxDist)` comparisons (ColorTransfer.cpp:1311-1320); its weight then becomes 1e-6 in k_err_weight, which is what
    // the reference's max() — the Windows macro (stdafx.h includes windows.h): ((a) > (b)) ? (a) : (b) — makes of a NaN
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const float f = v[i]; if (f == f) { const unsigned o = f2ord(f); lo = min(lo, o); hi = max(hi, o); } }
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

int nctk_local_color_transfer(nct_ctx* ctx, hipStream_t s, const float* err, const uint8_t* s_lab_level, const uint8_t* g_lab_level,
                              const uint8_t* s_lab_full, const int* knn_id, const double* knn_w, int layer, int h, int w, int H, int W,
                              const nct_color_params& prm, uint8_t* out_lab_full, const nct_color_debug* dbg, const nct_s1_graph* graph) {
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
    // ---------------- S1 (k_s1.hip)
    const double normFactor = (double)(W * H) / (double)(w * h);
    {
        DevBuf<double> gx(ctx, n), gy(ctx, n);
        if (!gx.ok() || !gy.ok()) return NCT_ERR_HIP;
        // lambda / alpha / dWeight arrive as float in the reference signature (ColorTransfer.cpp:548-550)
        const float lambda_f = (float)prm.local_weight, alpha_f = (float)prm.wls_alpha, dWeight_f = (float)normFactor;
        hipLaunchKernelGGL(k_gradient_weights, dim3(nbl), dim3(256), 0, s, s_lab_level, h, w, (double)lambda_f, (double)alpha_f, (double*)gx, (double*)gy); LCHK();
        int rc1;
        if (graph) {
            rc1 = nctk_s1_solve(ctx, s, *graph, knn_id, weight, dWeight_f, s_lab_level, g_lab_level, gx, gy, layer, h, w, x, dbg ? dbg->cg_iters : nullptr);
        } else {
            // no prebuilt graph part (the host entry point nct_local_color_transfer): build it here, on this stream; the host does not know the hub block count
            nct_s1_graph_bufs gb(ctx, n);
            if (!gb.ok()) return NCT_ERR_HIP;
            const nct_s1_graph g = gb.view(-1, -1);
            rc1 = nctk_s1_graph_build(ctx, s, knn_id, knn_w, sqrt(prm.nonlocal_weight / prm.k_num), g, nullptr);
            if (rc1 == 0) rc1 = nctk_s1_solve(ctx, s, g, knn_id, weight, dWeight_f, s_lab_level, g_lab_level, gx, gy, layer, h, w, x, dbg ? dbg->cg_iters : nullptr);
        }
        if (rc1) return rc1;
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
        if (dbg && dbg->wls_iters) for (int q = 0; q < 6; ++q) dbg->wls_iters[q] = wit[q];
    }
    if (dbg) { int rc = dbg_copy(ctx, s, dbg->ab_wls, X, (size_t)6 * N); if (rc) return rc; }
    // ---------------- A1
    hipLaunchKernelGGL(k_apply, dim3(cdiv(3 * N, 256)), dim3(256), 0, s, (const double*)Xa, (const double*)Xb, s_lab_full, N, out_lab_full); LCHK();
    return 0;
}
