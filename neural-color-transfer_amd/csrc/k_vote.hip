// k_vote.hip — bidirectional-similarity votes (B1 image domain, B2 feature domain).
// Reference: reconstruct_bds GeneralizedPatchMatch.cu:122-235 (serial host loops in the reference);
//            avg_vote_bds_a :1074-1126, avg_vote_bds_b :1128-1178 (float atomicAdd scatter), avg_vote_bds :1180-1202.
//
// MI355X design: the completeness vote is a *gather*, not an atomic scatter. The R->S NNF is inverted once per
// level (stable radix sort of (matched S pixel, R pixel) pairs + per-S-pixel segment starts), then every S pixel
// walks the <=9 per-tap source lists, tap after tap, each in ascending source order. That (a) removes 9*C float atomics per R pixel,
// (b) makes the fp32 sum order deterministic (taps dx-outer/dy-inner, ascending source pixel inside a tap, long lists in blocks of 64) == oracle/orc_vote.c,
// (c) lets the coherence vote, the completeness vote and the final division fuse into ONE kernel per domain.
// Roofline: HBM/L2 gather, ~18 feature vectors read + 1 written per S pixel.
#include "nct_internal.h"
#include "nct_device.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>   // rocPRIM directly (no CUB-compatibility layer)
#include <rocprim/device/device_scan.hpp>

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
// Canonical order v2 (round 5; = oracle/orc_vote.c): after the coherence part, TAP-MAJOR — for the nine taps (dx outer, dy inner) the sources whose match is the tapped
// neighbour s = target - tap, ascending source pixel: that is simply the contiguous range start[s] .. start[s + 1] of the sorted inverse map, no merge across lists.
// Rounds 1-4 added in ascending source pixel across all nine lists (a rank sort through LDS per round); the order is this project's own choice (the reference's float
// atomics have none) and tap-major is what makes LONG lists tractable: natural photographs collapse 10^4 R pixels of a flat region onto one match (in4/tar4: 47 535
// sources on one target — 24 ms for the finest vote where the synthetic pair takes 1.2). A list is added in blocks of VT = 64: the first block source by source into the
// target's sums; every further block summed from zero in the same way by k_vote_hub — one 16-lane row per (block, tap), all blocks of a level in parallel — and then
// added as ONE addend, in block order. Lists of <= 64 sources (all but a few hundred on the synthetic pairs) never leave this kernel.
// b_img / out_img (nullable): the IMAGE-domain vote of the same pixel (B1, reconstruct_bds) rides along — it walks exactly the same taps and sources, so lanes 0..2 of the
// pixel's row accumulate the three colour channels (integer sums: order-free) and write the guidance pixel.
constexpr int VT = 64;
struct VoteHub {
    const int* hstart;       // [na + 1] index of the first further block of S pixel s (exclusive scan of (cnt - 1) / VT for cnt > VT); hstart[na] = number of blocks
    const int* blk_s;        // [blocks] the S pixel whose list the block belongs to
    const int* blk_pos;      // [blocks] position of the block's first entry in the sorted inverse map
    float* P;                // [blocks * 9][C] block sums per tap
    float* PW;               // [blocks * 9]    their weight sums
    int* CNT;                // [blocks * 9]    in-bounds sources (image vote)
    int* IB;                 // [blocks * 9][3] their colour sums (image vote)
};
// entry code: bit 31 = block partial (index = (block * 9 + tap)), else source row (index = tapped R pixel); 0xFFFFFFFF = nothing (out of bounds / beyond the end)
constexpr uint32_t V_NONE = 0xFFFFFFFFu, V_PART = 0x80000000u;

template <int NCH, int NR>
__device__ __forceinline__ void vote_accumulate(uint32_t mycode, int count, int v, int nch, int C, const float* __restrict__ pin, const VoteHub& H, double wb, float wbf,
                                                bool img, const uint8_t* __restrict__ b_img, float4 (&acc)[NR], float& pw, int& ib, int& bcnt) {
    // `count` entries of the row's 16 lanes (lane u holds entry u), consumed in lane order, four rows in flight
    for (int e0 = 0; e0 < count; e0 += 4) {
        uint32_t code[4]; float4 row[4][NR];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            code[u] = __shfl(mycode, e0 + u, 16);
            if (e0 + u >= count) code[u] = V_NONE;
            if (code[u] != V_NONE) {
                const bool part = (code[u] & V_PART) != 0;
                const float4* src = reinterpret_cast<const float4*>((part ? H.P : pin) + (size_t)(code[u] & ~V_PART) * C);
#pragma unroll
                for (int k = 0; k < NR; ++k) if (k < nch) row[u][k] = src[v + 16 * k];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (code[u] == V_NONE) continue;
            if (code[u] & V_PART) {
                const uint32_t idx = code[u] & ~V_PART;
                pw = pw + H.PW[idx];
                bcnt += H.CNT[idx];
                if (img) ib += H.IB[(size_t)idx * 3 + v];
#pragma unroll
                for (int k = 0; k < NR; ++k)
                    if (k < nch) { acc[k].x = acc[k].x + row[u][k].x; acc[k].y = acc[k].y + row[u][k].y; acc[k].z = acc[k].z + row[u][k].z; acc[k].w = acc[k].w + row[u][k].w; }
            } else {
                pw = pw + wbf;
                ++bcnt;
                if (img) ib += b_img[(size_t)code[u] * 3 + v];
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
    }
}

// one 16-lane row per (further block, tap): the block's <= VT sources in ascending order, summed from zero exactly like a first block
template <int NCH>
__global__ __launch_bounds__(256) void k_vote_hub(const uint32_t* __restrict__ inv_vals, const int* __restrict__ inv_start, const float* __restrict__ pin, VoteHub H,
                                                  int C, int na, int bh, int bw, double wb, const uint8_t* __restrict__ b_img) {
    const int nunits = H.hstart[na] * 9;
    const int v = threadIdx.x & 15;
    constexpr int NR = NCH > 0 ? NCH : 8;
    const int nch = NCH > 0 ? NCH : ((C >> 2) + 15 - v) / 16;
    const float wbf = (float)wb;
    const bool img = b_img != nullptr && v < 3;
    for (int unit = blockIdx.x * 16 + (threadIdx.x >> 4); unit < nunits; unit += gridDim.x * 16) {
        const int blk = unit / 9, t = unit - blk * 9;
        const int dx = t / 3 - 1, dy = t % 3 - 1;
        const int pos = H.blk_pos[blk], end = min(pos + VT, inv_start[H.blk_s[blk] + 1]);
        float4 acc[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        float pw = 0.f; int ib = 0, bcnt = 0;
        for (int j0 = pos; j0 < end; j0 += 16) {
            uint32_t code = V_NONE;
            if (j0 + v < end) {
                const int q = (int)inv_vals[j0 + v];
                const int qy = q / bw, qx = q - qy * bw;
                const int xb = qx + dx, yb = qy + dy;
                if (xb < bw && xb >= 0 && yb < bh && yb >= 0) code = (uint32_t)(yb * bw + xb);
            }
            vote_accumulate<NCH, NR>(code, min(16, end - j0), v, nch, C, pin, H, wb, wbf, img, b_img, acc, pw, ib, bcnt);
        }
        float4* dst = reinterpret_cast<float4*>(H.P + (size_t)unit * C);
#pragma unroll
        for (int k = 0; k < NR; ++k) if (k < nch) dst[v + 16 * k] = acc[k];
        if (v == 0) { H.PW[unit] = pw; H.CNT[unit] = bcnt; }
        if (b_img != nullptr && v < 3) H.IB[(size_t)unit * 3 + v] = ib;
    }
}

template <int NCH>   // float4 chunks per lane (C = 64*NCH), 0 = generic (C <= 512, any multiple of 4)
__global__ __launch_bounds__(256) void k_vote_features(const uint32_t* __restrict__ ann, const uint32_t* __restrict__ inv_vals, const int* __restrict__ inv_start,
                                                       const float* __restrict__ pin, float* __restrict__ pout, float* __restrict__ pw_out, VoteHub H,
                                                       int C, int ah, int aw, int bh, int bw, double wa, double wb,
                                                       const uint8_t* __restrict__ b_img, uint8_t* __restrict__ out_img) {
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
    // coherence (avg_vote_bds_a): float += double  ==> evaluate in double, round to float each time.
    // The nine NNF words are requested together, then the feature rows of up to G taps together (an absent tap reads row 0 and is not added): as a dx / dy loop with the
    // loads inside its two conditions, every tap was two dependent round trips — eighteen before the completeness part started (round 6, from the ISA). Same order of
    // additions (dx outer, dy inner): same bits.
    if (live) {
        uint32_t vps[9]; bool ok[9]; int src_px[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dx = t / 3 - 1, dy = t % 3 - 1;
            const int nx = ax + dx, ny = ay + dy;
            ok[t] = nx < aw && nx >= 0 && ny < ah && ny >= 0;
            vps[t] = ann[ok[t] ? ny * aw + nx : ay * aw + ax];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dx = t / 3 - 1, dy = t % 3 - 1;
            const int xp = nnf_x(vps[t]) - dx, yp = nnf_y(vps[t]) - dy;
            ok[t] = ok[t] && xp < bw && xp >= 0 && yp < bh && yp >= 0;
            src_px[t] = ok[t] ? yp * bw + xp : 0;
        }
        constexpr int G = NR == 1 ? 9 : (NR == 2 ? 3 : 1);
#pragma unroll
        for (int t0 = 0; t0 < 9; t0 += G) {
            float4 xs[G][NR]; int ib3[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const float4* src = reinterpret_cast<const float4*>(pin + (size_t)src_px[t0 + u] * C);
#pragma unroll
                for (int k = 0; k < NR; ++k) xs[u][k] = k < nch ? src[v + 16 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
                ib3[u] = img ? (int)b_img[(size_t)src_px[t0 + u] * 3 + v] : 0;
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
                if (ok[t0 + u]) {
                    pw = (float)((double)pw + wa);
                    if (img) ia += ib3[u];
                    ++acnt;
#pragma unroll
                    for (int k = 0; k < NR; ++k)
                        if (k < nch) {
                            const float4 x = xs[u][k];
                            acc[k].x = (float)((double)acc[k].x + (double)x.x * wa);
                            acc[k].y = (float)((double)acc[k].y + (double)x.y * wa);
                            acc[k].z = (float)((double)acc[k].z + (double)x.z * wa);
                            acc[k].w = (float)((double)acc[k].w + (double)x.w * wa);
                        }
                }
        }
    }
    // completeness (avg_vote_bds_b): atomicAdd(float*, (float)(wb*pin)) in the canonical order of the header comment. The nine lists' first blocks and the block sums
    // of their further blocks form ONE sequence per target (list 0's sources, list 0's block sums, list 1's sources, …); lane u of the row decodes entry j0 + u of it —
    // sixteen inverse-map words requested at once — and the row then consumes the sixteen entries in order, four feature rows in flight.
    const float wbf = (float)wb;
    int lpos = 0, nfirst = 0, nblk = 0, hb = 0;
    if (live && v < 9) {
        const int dx = v / 3 - 1, dy = v % 3 - 1;
        const int sx = ax - dx, sy = ay - dy;
        if (sx >= 0 && sx < aw && sy >= 0 && sy < ah) {
            const int sidx = sy * aw + sx;
            lpos = inv_start[sidx];
            const int cnt = inv_start[sidx + 1] - lpos;
            nfirst = min(cnt, VT);
            hb = H.hstart[sidx]; nblk = H.hstart[sidx + 1] - hb;
        }
    }
    int incl = nfirst + nblk;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { const int up = __shfl_up(incl, off, 16); if (v >= off) incl += up; }
    const int total = __shfl(incl, 15, 16);
    int offs[9];                                          // exclusive start of tap t's entries in the target's sequence
#pragma unroll
    for (int t = 0; t < 9; ++t) offs[t] = t == 0 ? 0 : __shfl(incl, t - 1, 16);
    for (int j0 = 0; j0 < total; j0 += 16) {
        const int j = j0 + v;
        int t = 0;
#pragma unroll
        for (int u = 1; u < 9; ++u) t += offs[u] <= j ? 1 : 0;
        int ot = 0;
#pragma unroll
        for (int u = 0; u < 9; ++u) ot = u == t ? offs[u] : ot;
        const int r = j - ot;
        const int lp = __shfl(lpos, t, 16), nf = __shfl(nfirst, t, 16), hbt = __shfl(hb, t, 16);
        uint32_t code = V_NONE;
        if (j < total) {
            if (r < nf) {
                const int dx = t / 3 - 1, dy = t % 3 - 1;
                const int q = (int)inv_vals[lp + r];
                const int qy = q / bw, qx = q - qy * bw;
                const int xb = qx + dx, yb = qy + dy;
                if (xb < bw && xb >= 0 && yb < bh && yb >= 0) code = (uint32_t)(yb * bw + xb);
            } else code = V_PART | (uint32_t)((hbt + (r - nf)) * 9 + t);
        }
        vote_accumulate<NCH, NR>(code, min(16, total - j0), v, nch, C, pin, H, wb, wbf, img, b_img, acc, pw, ib, bcnt);
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

// block table of the long lists: per S pixel the number of further blocks, scanned; one thread per S pixel fills its blocks' entries
__global__ void k_vote_blk_counts(const int* __restrict__ inv_start, int na, int* __restrict__ cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > na) return;
    int c = 0;
    if (s < na) { const int n = inv_start[s + 1] - inv_start[s]; c = n > VT ? (n - 1) / VT : 0; }
    cnt[s] = c;
}
__global__ void k_vote_blk_fill(const int* __restrict__ inv_start, const int* __restrict__ hstart, int na, int* __restrict__ blk_s, int* __restrict__ blk_pos) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= na) return;
    const int b0 = hstart[s], nb = hstart[s + 1] - b0, p0 = inv_start[s];
    for (int k = 0; k < nb; ++k) { blk_s[b0 + k] = s; blk_pos[b0 + k] = p0 + (k + 1) * VT; }
}
struct VoteHubBufs {
    DevBuf<int> cnt, hstart, blk_s, blk_pos, CNT, IB; DevBuf<float> P, PW;
    VoteHubBufs(nct_ctx* c, int nb, int na, int C) : cnt(c, (size_t)na + 1), hstart(c, (size_t)na + 1), blk_s(c, (size_t)nb / VT + 1), blk_pos(c, (size_t)nb / VT + 1),
        CNT(c, ((size_t)nb / VT + 1) * 9), IB(c, ((size_t)nb / VT + 1) * 27), P(c, ((size_t)nb / VT + 1) * 9 * C), PW(c, ((size_t)nb / VT + 1) * 9) {}
    bool ok() const { return cnt.ok() && hstart.ok() && blk_s.ok() && blk_pos.ok() && CNT.ok() && IB.ok() && P.ok() && PW.ok(); }
    VoteHub view() const { return VoteHub{hstart, blk_s, blk_pos, P, PW, CNT, IB}; }
};

static int launch_vote_features(nct_ctx* ctx, hipStream_t s, const InvMap& inv, const uint32_t* ann, const float* pin_hwc, float* pout_hwc, float* pw,
                                int C, int ah, int aw, int bh, int bw, float w_coh, float w_comp, const uint8_t* b_img = nullptr, uint8_t* out_img = nullptr) {
    const double wa = w_coh / (double)(aw * ah);
    const double wb = w_comp / (double)(bw * bh);
    const int na = ah * aw, nb = bh * bw;
    VoteHubBufs hb(ctx, nb, na, C);
    if (!hb.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_vote_blk_counts, dim3(cdiv(na + 1, 256)), dim3(256), 0, s, (const int*)inv.start, na, (int*)hb.cnt); NCT_LAUNCH_CHECK();
    size_t scan_bytes = 0;
    NCT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (const int*)hb.cnt, (int*)hb.hstart, 0, (size_t)na + 1, rocprim::plus<int>(), s));
    DevBuf<char> tmp(ctx, scan_bytes + 16);
    if (!tmp.ok()) return NCT_ERR_HIP;
    NCT_HIP(rocprim::exclusive_scan((void*)(char*)tmp, scan_bytes, (const int*)hb.cnt, (int*)hb.hstart, 0, (size_t)na + 1, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_vote_blk_fill, dim3(cdiv(na, 256)), dim3(256), 0, s, (const int*)inv.start, (const int*)hb.hstart, na, (int*)hb.blk_s, (int*)hb.blk_pos); NCT_LAUNCH_CHECK();
    const VoteHub H = hb.view();
    // the hub pass: the number of further blocks is known on the device only; a fixed grid walks them with a stride and exits at once where there are none
    int hub_grid = cdiv((nb / VT + 1) * 9, 16); if (hub_grid > 2048) hub_grid = 2048;
    dim3 grid(cdiv(na, 16)), block(256);
#define NCT_VOTE_LAUNCH(N) do { \
        hipLaunchKernelGGL(k_vote_hub<N>, dim3(hub_grid), block, 0, s, (const uint32_t*)inv.vals_s, (const int*)inv.start, pin_hwc, H, C, na, bh, bw, wb, b_img); \
        hipLaunchKernelGGL(k_vote_features<N>, grid, block, 0, s, ann, (const uint32_t*)inv.vals_s, (const int*)inv.start, pin_hwc, pout_hwc, pw, H, C, ah, aw, bh, bw, wa, wb, b_img, out_img); } while (0)
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
