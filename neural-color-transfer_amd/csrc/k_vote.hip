// k_vote.hip — bidirectional-similarity votes (B1 image domain, B2 feature domain).
// Reference: reconstruct_bds GeneralizedPatchMatch.cu:122-235 (serial host loops in the reference);
//            avg_vote_bds_a :1074-1126, avg_vote_bds_b :1128-1178 (float atomicAdd scatter), avg_vote_bds :1180-1202.
//
// MI355X design: the completeness vote is a *gather*, not an atomic scatter. The R->S NNF is inverted once per
// level (stable radix sort of (matched S pixel, R pixel) pairs + per-S-pixel segment starts), then every S pixel
// merges the <=9 per-tap source lists in ascending source order. That (a) removes 9*C float atomics per R pixel,
// (b) makes the fp32 sum order deterministic (ascending source pixel, taps dx-outer/dy-inner) == oracle/orc_vote.c,
// (c) lets the coherence vote, the completeness vote and the final division fuse into ONE kernel per domain.
// Roofline: HBM/L2 gather, ~18 feature vectors read + 1 written per S pixel.
#include "nct_internal.h"
#include "nct_device.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>   // rocPRIM directly (no CUB-compatibility layer)

// ---------------------------------------------------------------- inverse map of the R->S NNF
__global__ void k_inv_keys(const uint32_t* __restrict__ bnn, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int nb, int aw) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nb) return;
    uint32_t v = bnn[q];
    keys[q] = (uint32_t)(nnf_y(v) * aw + nnf_x(v));
    vals[q] = (uint32_t)q;
}
// start[s] = first index i with keys_sorted[i] >= s, s in [0, na]
__global__ void k_inv_starts(const uint32_t* __restrict__ keys, int nb, int* __restrict__ start, int na) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > na) return;
    int lo = 0, hi = nb;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (keys[mid] < (uint32_t)s) lo = mid + 1; else hi = mid; }
    start[s] = lo;
}

struct InvMap {
    DevBuf<uint32_t> keys, vals, keys_s, vals_s;
    DevBuf<int> start;
    InvMap(nct_ctx* c, int nb, int na) : keys(c, nb), vals(c, nb), keys_s(c, nb), vals_s(c, nb), start(c, na + 1) {}
    bool ok() const { return keys.ok() && vals.ok() && keys_s.ok() && vals_s.ok() && start.ok(); }
};

static int build_inverse(nct_ctx* ctx, hipStream_t s, const uint32_t* bnn, int bh, int bw, int ah, int aw, InvMap& inv) {
    const int nb = bh * bw, na = ah * aw;
    hipLaunchKernelGGL(k_inv_keys, dim3(cdiv(nb, 256)), dim3(256), 0, s, bnn, (uint32_t*)inv.keys, (uint32_t*)inv.vals, nb, aw);
    NCT_LAUNCH_CHECK();
    int end_bit = 1; while ((1u << end_bit) < (unsigned)na && end_bit < 32) ++end_bit;
    size_t tmp_bytes = 0;
    NCT_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint32_t*)inv.keys, (uint32_t*)inv.keys_s,
                                               (const uint32_t*)inv.vals, (uint32_t*)inv.vals_s, nb, 0, end_bit, s));
    DevBuf<char> tmp(ctx, tmp_bytes ? tmp_bytes : 16);
    if (!tmp.ok()) return NCT_ERR_HIP;
    NCT_HIP(rocprim::radix_sort_pairs((void*)(char*)tmp, tmp_bytes, (const uint32_t*)inv.keys, (uint32_t*)inv.keys_s,
                                               (const uint32_t*)inv.vals, (uint32_t*)inv.vals_s, nb, 0, end_bit, s));
    hipLaunchKernelGGL(k_inv_starts, dim3(cdiv(na + 1, 256)), dim3(256), 0, s, (const uint32_t*)inv.keys_s, nb, (int*)inv.start, na);
    NCT_LAUNCH_CHECK();
    return 0;                              // tmp returns to the arena (recycled in stream order)
}

// iterate the sources of target (xa,ya) in canonical order; F(q, dx, dy) is called for every (source, tap) pair
// whose tapped source pixel (qx+dx, qy+dy) is inside B.
template <typename F>
__device__ __forceinline__ void for_each_source(const uint32_t* __restrict__ vals, const int* __restrict__ start,
                                                int xa, int ya, int ah, int aw, int bh, int bw, F&& f) {
    int pos[9], end[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dx = t / 3 - 1, dy = t % 3 - 1;        // dx outer, dy inner (the kernels' loop order)
        const int sx = xa - dx, sy = ya - dy;
        const bool in = sx >= 0 && sx < aw && sy >= 0 && sy < ah;
        const int s = in ? sy * aw + sx : 0;
        pos[t] = in ? start[s] : 0;
        end[t] = in ? start[s + 1] : 0;
    }
    // 9-way merge with the list heads in registers: only the list that advanced is re-read
    uint32_t head[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) head[t] = pos[t] < end[t] ? vals[pos[t]] : 0xFFFFFFFFu;
    while (true) {
        uint32_t qbest = 0xFFFFFFFFu; int tsel = -1;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (head[t] < qbest) { qbest = head[t]; tsel = t; }
        if (tsel < 0) break;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (t == tsel) { pos[t]++; head[t] = pos[t] < end[t] ? vals[pos[t]] : 0xFFFFFFFFu; }
        const int dx = tsel / 3 - 1, dy = tsel % 3 - 1;
        const int qy = (int)qbest / bw, qx = (int)qbest - qy * bw;
        const int xb = qx + dx, yb = qy + dy;
        if (xb < bw && xb >= 0 && yb < bh && yb >= 0) f(yb * bw + xb);
    }
}

// ---------------------------------------------------------------- B2 feature vote (one 16-lane row per S pixel)
// b_img / out_img (nullable): the IMAGE-domain vote of the same pixel (B1, reconstruct_bds) rides along — it walks exactly the same coherence taps and the same merged
// source list, so lanes 0..2 of the pixel's row accumulate the three colour channels (integer sums: order-free) and write the guidance pixel. One traversal of the inverse
// map per level instead of two (k_vote_image alone: 0.75 ms at 700x700).
constexpr int VK = 8;                   // entries one tap list can contribute to a round (2 / 4 / 6 / 8: votes 2.32 / 2.09 / 2.03 / 2.01 ms per pair)
constexpr int VOTE_MAXS = 9 * VK;
template <int NCH>   // float4 chunks per lane (C = 64*NCH), 0 = generic (loops, re-reads pout from memory)
__global__ __launch_bounds__(256) void k_vote_features(const uint32_t* __restrict__ ann, const uint32_t* __restrict__ inv_vals, const int* __restrict__ inv_start,
                                                       const float* __restrict__ pin, float* __restrict__ pout, float* __restrict__ pw_out,
                                                       int C, int ah, int aw, int bh, int bw, double wa, double wb,
                                                       const uint8_t* __restrict__ b_img, uint8_t* __restrict__ out_img) {
    __shared__ uint32_t s_keys[16][VOTE_MAXS], s_sorted[16][VOTE_MAXS];
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int v = threadIdx.x & 15;
    const bool live = pix < ah * aw;
    const int ay = live ? pix / aw : 0, ax = live ? pix - ay * aw : 0;
    constexpr int NR = NCH > 0 ? NCH : 8;                 // generic path supports C <= 512
    const int nch = NCH > 0 ? NCH : ((C >> 2) + 15 - v) / 16;   // chunks owned by this lane
    float4 acc[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float pw = 0.f;
    const bool img = b_img != nullptr && v < 3;
    int ia = 0, ib = 0, acnt = 0, bcnt = 0;
    // coherence (avg_vote_bds_a): float += double  ==> evaluate in double, round to float each time
    if (live)
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
            const int nx = ax + dx, ny = ay + dy;
            if (nx < aw && nx >= 0 && ny < ah && ny >= 0) {
                const uint32_t vp = ann[ny * aw + nx];
                const int xp = nnf_x(vp) - dx, yp = nnf_y(vp) - dy;
                if (xp < bw && xp >= 0 && yp < bh && yp >= 0) {
                    pw = (float)((double)pw + wa);
                    if (img) ia += b_img[((size_t)yp * bw + xp) * 3 + v];
                    ++acnt;
                    const float4* src = reinterpret_cast<const float4*>(pin + ((size_t)yp * bw + xp) * C);
#pragma unroll
                    for (int k = 0; k < NR; ++k)
                        if (k < nch) {
                            const float4 x = src[v + 16 * k];
                            acc[k].x = (float)((double)acc[k].x + (double)x.x * wa);
                            acc[k].y = (float)((double)acc[k].y + (double)x.y * wa);
                            acc[k].z = (float)((double)acc[k].z + (double)x.z * wa);
                            acc[k].w = (float)((double)acc[k].w + (double)x.w * wa);
                        }
                }
            }
        }
    // completeness (avg_vote_bds_b): atomicAdd(float*, (float)(wb*pin)) in ascending source order
    const float wbf = (float)wb;
    // The source order is fixed (ascending source pixel), the way it is produced is not. A 9-way merge waits at every step for the advanced list's next head and then
    // for that source's feature row: one dependent chain per SOURCE — and the in-degree of the inverse map is skewed (700x700 bench pair: median 2, mean 9, p99 94,
    // max 816 sources per pixel; scripts/vote_sources_hist.py): the launch lasted as long as its worst pixel's chain (0.96 of 1.36 ms at 700x700). Now in rounds:
    // lane t < 9 of the pixel fetches the next VK + 1 entries of ITS tap's list; pivot = the smallest (VK + 1)-th entry over the nine lists; every entry below the pivot (at
    // most VK per list, all VK of the list that set it) is smaller than everything that stays behind, so the round's <= 9 VK (source, tap) words are rank-sorted
    // through LDS (sources are unique: a source pixel has one correspondence) and their rows are requested four at a time and accumulated in that order.
    int lpos = 0, lend = 0;
    if (live && v < 9) {
        const int dx = v / 3 - 1, dy = v % 3 - 1;
        const int sx = ax - dx, sy = ay - dy;
        if (sx >= 0 && sx < aw && sy >= 0 && sy < ah) { const int sidx = sy * aw + sx; lpos = inv_start[sidx]; lend = inv_start[sidx + 1]; }
    }
    uint32_t* keys = s_keys[threadIdx.x >> 4];
    uint32_t* sorted = s_sorted[threadIdx.x >> 4];
    int more = 1;
    while (more) {
        uint32_t e[VK + 1];
        const int rem = lend - lpos;
#pragma unroll
        for (int k = 0; k <= VK; ++k) e[k] = k < rem ? inv_vals[lpos + k] : 0xFFFFFFFFu;
        uint32_t pivot = e[VK];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const uint32_t o = __shfl_xor(pivot, off, 16); pivot = o < pivot ? o : pivot; }
        int take = 0;
#pragma unroll
        for (int k = 0; k < VK; ++k) take += e[k] < pivot ? 1 : 0;          // e is ascending; the padding 0xFFFFFFFF is never below a pivot
        int incl = take;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int up = __shfl_up(incl, off, 16); if (v >= off) incl += up; }
        const int total = __shfl(incl, 15, 16), excl = incl - take;
#pragma unroll
        for (int k = 0; k < VK; ++k) if (k < take) keys[excl + k] = (e[k] << 4) | (uint32_t)v;
        lpos += take;
        __syncthreads();
        for (int el = v; el < total; el += 16) {
            const uint32_t key = keys[el];
            int rank = 0;
            for (int o = 0; o < total; ++o) rank += keys[o] < key ? 1 : 0;
            const int t = (int)(key & 15u), q = (int)(key >> 4);
            const int dx = t / 3 - 1, dy = t % 3 - 1;
            const int qy = q / bw, qx = q - qy * bw;
            const int xb = qx + dx, yb = qy + dy;
            sorted[rank] = (xb < bw && xb >= 0 && yb < bh && yb >= 0) ? (uint32_t)(yb * bw + xb) : 0xFFFFFFFFu;
        }
        __syncthreads();
        for (int e0 = 0; e0 < total; e0 += 4) {
            uint32_t bid[4]; float4 row[4][NR];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bid[u] = e0 + u < total ? sorted[e0 + u] : 0xFFFFFFFFu;
                if (bid[u] != 0xFFFFFFFFu) {
                    const float4* src = reinterpret_cast<const float4*>(pin + (size_t)bid[u] * C);
#pragma unroll
                    for (int k = 0; k < NR; ++k) if (k < nch) row[u][k] = src[v + 16 * k];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bid[u] == 0xFFFFFFFFu) continue;
                pw = pw + wbf;
                if (img) ib += b_img[(size_t)bid[u] * 3 + v];
                ++bcnt;
#pragma unroll
                for (int k = 0; k < NR; ++k)
                    if (k < nch) {
                        const float4 x = row[u][k];
                        acc[k].x = acc[k].x + (float)(wb * (double)x.x);
                        acc[k].y = acc[k].y + (float)(wb * (double)x.y);
                        acc[k].z = acc[k].z + (float)(wb * (double)x.z);
                        acc[k].w = acc[k].w + (float)(wb * (double)x.w);
                    }
            }
        }
        more = __syncthreads_or(lpos < lend ? 1 : 0);                   // also: nobody still reads keys / sorted when the next round overwrites them
    }
    if (!live) return;
    // avg_vote_bds
    float4* dst = reinterpret_cast<float4*>(pout + (size_t)pix * C);
#pragma unroll
    for (int k = 0; k < NR; ++k)
        if (k < nch) {
            float4 r = acc[k];
            if (pw > 0) { r.x /= pw; r.y /= pw; r.z /= pw; r.w /= pw; }
            dst[v + 16 * k] = r;
        }
    if (pw_out && v == 0) pw_out[pix] = pw;
    if (img) {                                            // k_vote_image's arithmetic, channel v
        const double awt = acnt * wa, bwt = bcnt * wb;
        const double den = awt + bwt;
        out_img[(size_t)pix * 3 + v] = (uint8_t)((ia * wa + ib * wb) / den);
    }
}

static int launch_vote_features(nct_ctx* ctx, hipStream_t s, const InvMap& inv, const uint32_t* ann, const float* pin_hwc, float* pout_hwc, float* pw,
                                int C, int ah, int aw, int bh, int bw, float w_coh, float w_comp, const uint8_t* b_img = nullptr, uint8_t* out_img = nullptr) {
    const double wa = w_coh / (double)(aw * ah);
    const double wb = w_comp / (double)(bw * bh);
    dim3 grid(cdiv(ah * aw, 16)), block(256);
#define NCT_VOTE_LAUNCH(N) hipLaunchKernelGGL(k_vote_features<N>, grid, block, 0, s, ann, (const uint32_t*)inv.vals_s, (const int*)inv.start, \
                                              pin_hwc, pout_hwc, pw, C, ah, aw, bh, bw, wa, wb, b_img, out_img)
    switch (C) {
        case 64: NCT_VOTE_LAUNCH(1); break;
        case 128: NCT_VOTE_LAUNCH(2); break;
        case 256: NCT_VOTE_LAUNCH(4); break;
        case 512: NCT_VOTE_LAUNCH(8); break;
        default: NCT_VOTE_LAUNCH(0); break;
    }
#undef NCT_VOTE_LAUNCH
    NCT_LAUNCH_CHECK();
    return 0;
}
int nctk_bds_vote_features(nct_ctx* ctx, hipStream_t s, const uint32_t* ann, const uint32_t* bnn, const float* pin_hwc, float* pout_hwc, float* pw,
                           int C, int ah, int aw, int bh, int bw, float w_coh, float w_comp) {
    NCT_REQUIRE(C > 0 && (C & 3) == 0 && C <= 512, "bds_vote_features: C=%d must be a multiple of 4 and <= 512", C);
    InvMap inv(ctx, bh * bw, ah * aw);
    if (!inv.ok()) return NCT_ERR_HIP;
    int rc = build_inverse(ctx, s, bnn, bh, bw, ah, aw, inv);
    if (rc) return rc;
    return launch_vote_features(ctx, s, inv, ann, pin_hwc, pout_hwc, pw, C, ah, aw, bh, bw, w_coh, w_comp);
}

// ---------------------------------------------------------------- B1 image vote (one thread per S pixel; integer sums)
__global__ void k_vote_image(const uint8_t* __restrict__ b, const uint32_t* __restrict__ ann, const uint32_t* __restrict__ inv_vals, const int* __restrict__ inv_start,
                             int ah, int aw, int bh, int bw, double wa, double wb, uint8_t* __restrict__ out) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= ah * aw) return;
    const int ay = pix / aw, ax = pix - ay * aw;
    int a0 = 0, a1 = 0, a2 = 0, acnt = 0;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
            const int nx = ax + dx, ny = ay + dy;
            if (nx < aw && nx >= 0 && ny < ah && ny >= 0) {
                const uint32_t vp = ann[ny * aw + nx];
                const int xp = nnf_x(vp) - dx, yp = nnf_y(vp) - dy;
                if (xp < bw && xp >= 0 && yp < bh && yp >= 0) {
                    const uint8_t* bv = b + ((size_t)yp * bw + xp) * 3;
                    a0 += bv[0]; a1 += bv[1]; a2 += bv[2]; ++acnt;
                }
            }
        }
    int b0 = 0, b1 = 0, b2 = 0, bcnt = 0;
    for_each_source(inv_vals, inv_start, ax, ay, ah, aw, bh, bw, [&](int bid) {
        const uint8_t* bv = b + (size_t)bid * 3;
        b0 += bv[0]; b1 += bv[1]; b2 += bv[2]; ++bcnt;
    });
    const double awt = acnt * wa, bwt = bcnt * wb;
    const double den = awt + bwt;
    out[(size_t)pix * 3 + 0] = (uint8_t)((a0 * wa + b0 * wb) / den);
    out[(size_t)pix * 3 + 1] = (uint8_t)((a1 * wa + b1 * wb) / den);
    out[(size_t)pix * 3 + 2] = (uint8_t)((a2 * wa + b2 * wb) / den);
}

static int launch_vote_image(nct_ctx* ctx, hipStream_t s, const InvMap& inv, const uint8_t* b_bgr, const uint32_t* ann,
                             int ah, int aw, int bh, int bw, double w_coh, double w_comp, uint8_t* out_bgr) {
    const double wa = w_coh / (double)(aw * ah);
    const double wb = w_comp / (double)(bw * bh);
    hipLaunchKernelGGL(k_vote_image, dim3(cdiv(ah * aw, 256)), dim3(256), 0, s, b_bgr, ann, (const uint32_t*)inv.vals_s, (const int*)inv.start,
                       ah, aw, bh, bw, wa, wb, out_bgr);
    NCT_LAUNCH_CHECK();
    return 0;
}
int nctk_bds_vote_image(nct_ctx* ctx, hipStream_t s, const uint8_t* b_bgr, const uint32_t* ann, const uint32_t* bnn,
                        int ah, int aw, int bh, int bw, double w_coh, double w_comp, uint8_t* out_bgr) {
    InvMap inv(ctx, bh * bw, ah * aw);
    if (!inv.ok()) return NCT_ERR_HIP;
    int rc = build_inverse(ctx, s, bnn, bh, bw, ah, aw, inv);
    if (rc) return rc;
    return launch_vote_image(ctx, s, inv, b_bgr, ann, ah, aw, bh, bw, w_coh, w_comp, out_bgr);
}
// both votes of a level (main.cu:291 and :303-318) from ONE inversion of the R->S field
int nctk_bds_vote_both(nct_ctx* ctx, hipStream_t s, const uint8_t* b_bgr, const float* pin_hwc, const uint32_t* ann, const uint32_t* bnn, int C,
                       int ah, int aw, int bh, int bw, double w_coh, double w_comp, uint8_t* out_bgr, float* pout_hwc) {
    NCT_REQUIRE(C > 0 && (C & 3) == 0 && C <= 512, "bds_vote: C=%d must be a multiple of 4 and <= 512", C);
    InvMap inv(ctx, bh * bw, ah * aw);
    if (!inv.ok()) return NCT_ERR_HIP;
    int rc = build_inverse(ctx, s, bnn, bh, bw, ah, aw, inv);
    if (rc) return rc;
    // the image vote rides in the feature kernel (same taps, same source lists). Its weights are the doubles w / (w h) of launch_vote_image; the feature kernel derives
    // them from the float casts of the same values — identical as long as the cast is exact, which it is checked to be (bds weights are small decimals like 2.0: if a caller
    // ever passes a weight that is not a float, the two votes run as two kernels)
    if ((double)(float)w_coh == w_coh && (double)(float)w_comp == w_comp)
        return launch_vote_features(ctx, s, inv, ann, pin_hwc, pout_hwc, nullptr, C, ah, aw, bh, bw, (float)w_coh, (float)w_comp, b_bgr, out_bgr);
    rc = launch_vote_image(ctx, s, inv, b_bgr, ann, ah, aw, bh, bw, w_coh, w_comp, out_bgr);
    if (rc) return rc;
    return launch_vote_features(ctx, s, inv, ann, pin_hwc, pout_hwc, nullptr, C, ah, aw, bh, bw, (float)w_coh, (float)w_comp);
}
