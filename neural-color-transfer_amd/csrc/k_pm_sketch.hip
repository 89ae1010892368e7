// k_pm_sketch.hip — P1 helper (round 6): an EXACT low-dimensional upper bound on a random-search candidate's patch similarity, so that a far sample that cannot
// win is rejected from 32-byte records per pixel instead of from the first 768 / 1536-byte row of its feature tile.
// Reference: the random search of patchmatch_single (GeneralizedPatchMatch.cu:790-826) evaluates every sample in full; the NNF and the distances stay the same bits here
// because a sample is only ever dropped when it provably loses (below), exactly like the row rejection of k_patchmatch.hip.
//
// The bound. For ANY matrix P (K x C) with orthonormal rows and any vectors a, b:  a.b = (P a).(P b) + res_a . res_b  <=  y_a . y_b + rho_a rho_b,
// y = P x, res = x - P^T y, rho = |res| (Cauchy-Schwarz on the residuals; no unit-norm assumption). A sketch record is [y_0 .. y_6, rho]: the bound of a tap is the plain
// dot product of two 8-float records, the bound of a 3x3 patch the dot product of 72 + 72 floats — 288 bytes per candidate against 2304 (C = 64) / 4608 (C = 128).
// P = the K = 7 dominant principal directions of the level's pooled feature vectors (every 4th pixel of both maps): 94-99 % of the energy of conv1_1 / conv2_1 vectors
// (post-ReLU vectors live near a low-dimensional cone), so rho ~ 0.1-0.2 and the bound sits 0.2-0.3 above the true 9-tap sum — tight enough to reject 97 / 92 / 75 % of
// the radius-32 / 16 / 8 samples of a converging field (the first-row test: 91 / 78 / 50 %; on the demo photographs with flat regions the row test rejects nothing and
// the sketch 80-90 %: scripts/pm_sketch_probe.py).
// Exactness does NOT depend on the quality of P — any orthonormal P gives a valid bound; the subspace iteration only decides how MANY samples are rejected. What it
// depends on is (a) orthonormality: the kernel measures the defect max |P P^T - I| of the fp32 rows it stores and zeroes P when it exceeds 1e-5 (then y = 0, rho = |x|,
// the bound is >= the Cauchy-Schwarz bound of 9 taps and rejects nothing); (b) rounding: y is computed in fp32 (|error| < 1e-5 for |x| <= 1), rho from the residual of the
// ROUNDED y, inflated by 1e-5 relative + 3e-5 absolute, so y_a.y_b + rho_a rho_b over-estimates a.b up to < 6e-5 per tap; the consumer rejects only when
// bound + 2e-3 < need (k_patchmatch.hip), 2e-3 > 9 x 6e-5 + the 1e-4 fp32 accumulation margin of the canonical sum + the 1e-4 of the existing full-sum rule.
// NaN feature vectors (dead pixels: `norm` has no epsilon) are left out of the covariance; their records are NaN, a NaN bound compares false and rejects nothing.
#include "nct_internal.h"
#include "nct_device.h"

namespace {
constexpr int SK_K = 7;          // principal directions
constexpr int SK_F = 8;          // floats per record: y[0..6], rho
constexpr int SK_SUB = 4;        // every 4th pixel of each map enters the covariance
constexpr int SK_G = 256;        // workgroups of the covariance pass (fixed: partial sums are added in a fixed order)
constexpr int SK_CH = 32;        // pixels staged per round of the covariance pass
constexpr int SK_NIT = 8;        // subspace iterations (energy within 1e-4 of the exact eigenvectors' after 8: scripts/pm_sketch_probe.py)

// partial[g][C][C] = sum over the workgroup's share of the sampled pixels of x x^T. Thread t owns the (C/16) x (C/16) block (t / 16, t % 16).
template <int C>
__global__ __launch_bounds__(256) void k_pm_cov(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float* __restrict__ partial) {
    constexpr int C4 = C / 4, R = C / 16, R4 = R / 4;
    __shared__ float4 s_px[SK_CH * C4];
    const int t = threadIdx.x, bi = t >> 4, bj = t & 15;
    const int nsa = (na + SK_SUB - 1) / SK_SUB, nsb = (nb + SK_SUB - 1) / SK_SUB, ns = nsa + nsb;
    const int per = (ns + SK_G - 1) / SK_G, lo = blockIdx.x * per, hi = min(ns, lo + per);
    float acc[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < R; ++c) acc[r][c] = 0.f;
    for (int base = lo; base < hi; base += SK_CH) {
        for (int e = t; e < SK_CH * C4; e += 256) {
            const int p = e / C4, j = e - p * C4, si = base + p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (si < hi) {
                const float* src = si < nsa ? a + (size_t)si * SK_SUB * C : b + (size_t)(si - nsa) * SK_SUB * C;
                v = reinterpret_cast<const float4*>(src)[j];
                v.x = v.x == v.x ? v.x : 0.f; v.y = v.y == v.y ? v.y : 0.f; v.z = v.z == v.z ? v.z : 0.f; v.w = v.w == v.w ? v.w : 0.f;
            }
            s_px[e] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < SK_CH; ++p) {
            float ai[R], aj[R];
#pragma unroll
            for (int m = 0; m < R4; ++m) {
                const float4 u = s_px[p * C4 + bi * R4 + m], w = s_px[p * C4 + bj * R4 + m];
                ai[4 * m] = u.x; ai[4 * m + 1] = u.y; ai[4 * m + 2] = u.z; ai[4 * m + 3] = u.w;
                aj[4 * m] = w.x; aj[4 * m + 1] = w.y; aj[4 * m + 2] = w.z; aj[4 * m + 3] = w.w;
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < R; ++c) acc[r][c] = __builtin_fmaf(ai[r], aj[c], acc[r][c]);
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < R; ++c) out[(bi * R + r) * C + bj * R + c] = acc[r][c];
}

template <int C>
__global__ void k_pm_cov_reduce(const float* __restrict__ partial, float* __restrict__ cov, int ns) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * C) return;
    double s = 0.0;
    for (int g = 0; g < SK_G; ++g) s += (double)partial[(size_t)g * C * C + e];
    cov[e] = (float)(s / (double)(ns > 0 ? ns : 1));
}

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// One workgroup: K-dimensional subspace iteration on the C x C covariance (fp64 iterates, modified Gram-Schmidt every second multiplication and twice at the end),
// P[K][C] in fp32, the orthonormality defect of the STORED rows in info[1]; defect > 1e-5 (or not finite): P = 0, info[0] = 0 — the bound then rejects nothing.
template <int C>
__global__ __launch_bounds__(256) void k_pm_pca(const float* __restrict__ cov, float* __restrict__ P, float* __restrict__ info) {
    constexpr int K = SK_K, M = C / 64;
    extern __shared__ double s_dyn[];
    double* s_y = s_dyn; double* s_z = s_y + C * K; float* s_cov = reinterpret_cast<float*>(s_z + C * K);
    const int t = threadIdx.x;
    for (int e = t; e < C * C; e += 256) s_cov[e] = cov[e];
    for (int e = t; e < C * K; e += 256) { const int c = e / K, k = e - c * K; s_z[e] = (double)cosf((float)((c + 1) * (k + 1)) * 0.37f) + (k == 0 ? 1.0 : 0.0); }
    __syncthreads();
    // modified Gram-Schmidt of the columns of s_z into s_y, by wave 0 (lane l owns rows l, l + 64)
    auto mgs = [&]() {
        if (t < 64) {
            double z[K][M];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int m = 0; m < M; ++m) z[k][m] = s_z[(t + 64 * m) * K + k];
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if (j < k) {
                        double d = 0.0;
#pragma unroll
                        for (int m = 0; m < M; ++m) d += z[k][m] * z[j][m];
                        d = wave_sum_d(d);
#pragma unroll
                        for (int m = 0; m < M; ++m) z[k][m] -= d * z[j][m];
                    }
                }
                double q = 0.0;
#pragma unroll
                for (int m = 0; m < M; ++m) q += z[k][m] * z[k][m];
                q = wave_sum_d(q);
                if (!(q > 1e-280)) {                          // rank-deficient data (a constant image): any unit vector will do; the defect test below guards the rest
#pragma unroll
                    for (int m = 0; m < M; ++m) z[k][m] = (t + 64 * m) == (k * 9 + 1) % C ? 1.0 : 0.0;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (j < k) {
                            double d = 0.0;
#pragma unroll
                            for (int m = 0; m < M; ++m) d += z[k][m] * z[j][m];
                            d = wave_sum_d(d);
#pragma unroll
                            for (int m = 0; m < M; ++m) z[k][m] -= d * z[j][m];
                        }
                    }
                    q = 0.0;
#pragma unroll
                    for (int m = 0; m < M; ++m) q += z[k][m] * z[k][m];
                    q = wave_sum_d(q);
                }
                const double inv = 1.0 / sqrt(q);
#pragma unroll
                for (int m = 0; m < M; ++m) z[k][m] *= inv;
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int m = 0; m < M; ++m) s_y[(t + 64 * m) * K + k] = z[k][m];
        }
    };
    mgs();
    __syncthreads();
    for (int it = 1; it <= SK_NIT; ++it) {
        for (int e = t; e < C * K; e += 256) {
            const int c = e / K, k = e - c * K;
            double d = 0.0;
            for (int j = 0; j < C; ++j) d += (double)s_cov[j * C + c] * s_y[j * K + k];      // the matrix is symmetric bit for bit: column reads, no bank conflicts
            s_z[e] = d;
        }
        __syncthreads();
        if ((it & 1) == 0) mgs();
        else for (int e = t; e < C * K; e += 256) s_y[e] = s_z[e];
        __syncthreads();
    }
    for (int e = t; e < C * K; e += 256) s_z[e] = s_y[e];          // once more: the second pass removes what the first one's rounding left
    __syncthreads();
    mgs();
    __syncthreads();
    // the stored fp32 rows and their defect
    if (t < 64) {
        float pf[K][M];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int m = 0; m < M; ++m) pf[k][m] = (float)s_y[(t + 64 * m) * K + k];
        double defect = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (j <= k) {
                    double d = 0.0;
#pragma unroll
                    for (int m = 0; m < M; ++m) d += (double)pf[k][m] * (double)pf[j][m];
                    d = wave_sum_d(d) - (j == k ? 1.0 : 0.0);
                    d = d < 0 ? -d : d;
                    defect = (d > defect || !(d == d)) ? d : defect;
                }
            }
        const bool ok = defect <= 1e-5;                       // false for NaN
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int m = 0; m < M; ++m) P[k * C + t + 64 * m] = ok ? pf[k][m] : 0.f;
        if (t == 0) { info[0] = ok ? 1.f : 0.f; info[1] = (float)defect; }
    }
}

// sketch records [n][8] of one channel-last map: 16 lanes per pixel (lane v owns float4 chunks v, v + 16, ...)
template <int C>
__global__ __launch_bounds__(256) void k_pm_sketch_build(const float* __restrict__ map, int n, const float* __restrict__ P, float* __restrict__ out) {
    constexpr int K = SK_K, C4 = C / 4, M = C / 64;
    __shared__ float4 s_P[K * C4];
    for (int e = threadIdx.x; e < K * C4; e += 256) s_P[e] = reinterpret_cast<const float4*>(P)[e];
    __syncthreads();
    const int grp = threadIdx.x >> 4, v = threadIdx.x & 15;
    const int npass = (n + 15) / 16;
    for (int pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const int px = pass * 16 + grp;
        const int pc = px < n ? px : n - 1;                      // all lanes stay in the row reductions
        float4 a4[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a4[m] = reinterpret_cast<const float4*>(map)[(size_t)pc * C4 + v + 16 * m];
        float y[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float p = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) p = dot4_acc(a4[m], s_P[k * C4 + v + 16 * m], p);
            y[k] = row16_sum(p);
        }
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float4 r = a4[m];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float4 pk = s_P[k * C4 + v + 16 * m];
                r.x = __builtin_fmaf(-y[k], pk.x, r.x); r.y = __builtin_fmaf(-y[k], pk.y, r.y); r.z = __builtin_fmaf(-y[k], pk.z, r.z); r.w = __builtin_fmaf(-y[k], pk.w, r.w);
            }
            q = dot4_acc(r, r, q);
        }
        q = row16_sum(q);
        const float rho = __builtin_sqrtf(q) * (1.0f + 1e-5f) + 3e-5f;
        float o = rho;
#pragma unroll
        for (int k = 0; k < K; ++k) o = v == k ? y[k] : o;
        if (px < n && v < SK_F) out[(size_t)px * SK_F + v] = o;
    }
}

template <int C>
int sketch_run(nct_ctx* ctx, hipStream_t s, const float* a_hwc, int na, const float* b_hwc, int nb, float* skA, float* skB) {
    DevBuf<float> partial(ctx, (size_t)SK_G * C * C), cov(ctx, (size_t)C * C), P(ctx, (size_t)SK_K * C), info(ctx, 2);
    if (!partial.ok() || !cov.ok() || !P.ok() || !info.ok()) return NCT_ERR_HIP;
    const int ns = (na + SK_SUB - 1) / SK_SUB + (nb + SK_SUB - 1) / SK_SUB;
    hipLaunchKernelGGL(k_pm_cov<C>, dim3(SK_G), dim3(256), 0, s, a_hwc, na, b_hwc, nb, (float*)partial); NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pm_cov_reduce<C>, dim3(cdiv(C * C, 256)), dim3(256), 0, s, (const float*)partial, (float*)cov, ns); NCT_LAUNCH_CHECK();
    const size_t lds = (size_t)2 * C * SK_K * sizeof(double) + (size_t)C * C * sizeof(float);
    if (lds > 65536) NCT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pm_pca<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per device; a host-side call
    hipLaunchKernelGGL(k_pm_pca<C>, dim3(1), dim3(256), lds, s, (const float*)cov, (float*)P, (float*)info); NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pm_sketch_build<C>, dim3(min(cdiv(na, 16), 4096)), dim3(256), 0, s, a_hwc, na, (const float*)P, skA); NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pm_sketch_build<C>, dim3(min(cdiv(nb, 16), 4096)), dim3(256), 0, s, b_hwc, nb, (const float*)P, skB); NCT_LAUNCH_CHECK();
    return 0;
}
}  // namespace

// skA / skB: [na][8] / [nb][8] floats. C = 64 or 128 (the levels whose random search is byte bound); a_hwc / b_hwc: the natural channel-last maps (not the lane-interleaved copies)
int nctk_pm_sketch(nct_ctx* ctx, hipStream_t s, const float* a_hwc, int na, const float* b_hwc, int nb, int C, float* skA, float* skB) {
    NCT_REQUIRE(C == 64 || C == 128, "pm_sketch: C=%d (64 or 128)", C);
    NCT_REQUIRE(na > 0 && nb > 0, "pm_sketch: empty map");
    return C == 64 ? sketch_run<64>(ctx, s, a_hwc, na, b_hwc, nb, skA, skB) : sketch_run<128>(ctx, s, a_hwc, na, b_hwc, nb, skA, skB);
}
