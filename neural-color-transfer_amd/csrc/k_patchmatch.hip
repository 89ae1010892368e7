// k_patchmatch.hip — P1: dense-correspondence PatchMatch over normalised deep features.
// Reference: patchmatch_single GeneralizedPatchMatch.cu:677-831 (dist_compute_single :355-405,
// improve_guess_single :505-515), launched twice per level from main.cu:283-284.
//
// MI355X design (not a translation of the 24x24-thread, one-thread-per-query CUDA kernel):
//  * features are channel-last (HWC): a candidate's 3x3xC tile is three contiguous runs of 3*C floats, so every
//    load instruction of a query group is a fully coalesced C*4-byte row segment;
//  * one 16-lane DPP row per query (4 queries per wave64, 16 per 256-thread workgroup, arranged as a 4x4 pixel
//    tile so that neighbouring queries — whose candidate tiles overlap when the NNF is coherent — share L1/L2);
//    lane v owns float4 channel chunks v, v+16, …; the 9*C-term dot product is one fmaf chain per lane followed
//    by a 4-step DPP rotate-add (no LDS traffic, no bpermute). At C = 64 HALF a row serves a query (8 lanes, two of the
//    sixteen chains each, advanced together by packed FMAs over a lane-interleaved copy of the maps, the same summation
//    tree — pm_dist8): a wave then keeps eight instead of four candidate tiles in flight;
//  * the region of A shared by the queries of a workgroup (8x8 queries at C = 64: 10x10xC) is staged once per launch in LDS;
//  * the racy single launch of the reference becomes 1 + iters*4 Jacobi steps on a double-buffered NNF
//    (one launch per (iteration, jump)); random search is fused into the jump==1 step; RNG is counter based.
//    => results are deterministic and bit-identical to oracle/orc_nnf.c;
//  * a propagation candidate that cannot win is not evaluated: the neighbour proposes the query's current match, or the neighbour's match has not changed since
//    the same jump of the previous iteration (that candidate lost then, and the query's best only decreases). The step a match last changed in travels in the
//    spare top byte of the double-buffered NNF word. -47 % evaluations at the finest level of a 700x700 pair, -40 ... -60 % below, same NNF and distances.
// Roofline: memory (gather of candidate tiles): algorithmic bytes = evals*9*C*4 (+ query tile + NNF r/w). Measured at C = 64:
// DRAM-side traffic 0.55 of the HBM peak, the L1 data return path (64 B/clk/CU) ~90 % busy, VALU issue 43 %, and a wave's candidates are a
// dependent chain (neighbour NNF -> tile loads -> dot -> compare -> next), so the remaining lever is latency hiding: 8 candidate tiles per
// wave in flight and five waves per SIMD. Perfectly local candidates, half-size fp16 tiles at this level, packed FMAs (-38 % VALU) each left
// the launch time unchanged.
#include "nct_internal.h"
#include "nct_device.h"
#include <cfloat>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <hip/hip_fp16.h>

// workgroups per CU (= waves per SIMD) the C = 64 kernels are compiled for. Round 2: five (92 registers) beat four, 16.0 vs 17.4 ms per pair, six (80, spills) 19.1 ms.
// Round 3: with two patch rows per round trip (pm_dist8 STAGE 1: 109 registers) four waves beat five-with-one-row: 13.0 vs 14.6 ms (five with spills: 15.3); C = 128 at five: 10.4 vs 8.5 ms
#ifndef NCT_PM_OCC8
#define NCT_PM_OCC8 4
#endif
#ifndef NCT_PM_FAR_MAG
#define NCT_PM_FAR_MAG 16       // random samples drawn with this radius or more are "far" (mostly rejected after their first row)
#endif
#ifndef NCT_PM_FAR_STAGE
#define NCT_PM_FAR_STAGE 1
#endif
#ifndef NCT_PM_NEAR_STAGE
#define NCT_PM_NEAR_STAGE 1
#endif
#ifndef NCT_PM_FAST_MAX
#define NCT_PM_FAST_MAX 2
#endif
struct PMGeom { int C, ah, aw, bh, bw, tiles_x, tiles_y; };

__device__ __forceinline__ float dot4h_acc(const float4 a, const uint2 bh, float acc) {
    const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&bh.x));
    const float2 hi = __half22float2(*reinterpret_cast<const __half2*>(&bh.y));
    acc = __builtin_fmaf(a.x, lo.x, acc);
    acc = __builtin_fmaf(a.y, lo.y, acc);
    acc = __builtin_fmaf(a.z, hi.x, acc);
    acc = __builtin_fmaf(a.w, hi.y, acc);
    return acc;
}

// ---- C = 64, EIGHT lanes per query, packed fp32 FMAs (k_pm_step<1, MODE, .., 8>)
// A 16-lane row computes two queries: each lane carries TWO of the sixteen fmaf chains of the layout above — chains j and j + 8, the pair the
// first step of row16_sum adds — so the per-candidate work of a wave instruction (candidate generation, RNG, validity tests, addresses,
// reductions) serves eight queries, and the two independent chains advance in ONE v_pk_fma_f32 (each half an IEEE fma of its own chain:
// the same bits as two fmaf). For the packed operands to sit in aligned register pairs the feature maps are read from a lane-interleaved copy
// (k_pm_interleave64): float4 l of a pixel = [A.x B.x A.y B.y], float4 l + 8 = [A.z B.z A.w B.w] (each load instruction of a query still
// reads ONE contiguous 128-byte line; 32-byte lane records touched both lines of the pixel per instruction and cost 4 %) with A = channels
// 4j..4j+3, B = channels 4(j+8)..4(j+8)+3, j = pm_chunk8(l) — so that the remaining reduction steps (j^4, j^2, j^1) are quad permutes and a half-row
// mirror (half8_sum). Same chains, same tree: NNF and distances are bit-identical to the 16-lane form (tests/test_gpu_correspondence.py).
typedef float pm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pm_f2 pk_dot4_acc(const float4 a0, const float4 a1, const float4 b0, const float4 b1, pm_f2 acc) {
    acc = __builtin_elementwise_fma(pm_f2{a0.x, a0.y}, pm_f2{b0.x, b0.y}, acc);     // x of both chains
    acc = __builtin_elementwise_fma(pm_f2{a0.z, a0.w}, pm_f2{b0.z, b0.w}, acc);     // y
    acc = __builtin_elementwise_fma(pm_f2{a1.x, a1.y}, pm_f2{b1.x, b1.y}, acc);     // z
    acc = __builtin_elementwise_fma(pm_f2{a1.z, a1.w}, pm_f2{b1.z, b1.w}, acc);     // w
    return acc;
}
__global__ void k_pm_interleave64(const float4* __restrict__ src, float4* __restrict__ dst, int npix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 8) return;
    const int px = i >> 3, l = i & 7, j = pm_chunk8(l);
    const float4 a = src[(size_t)px * 16 + j], b = src[(size_t)px * 16 + j + 8];
    dst[(size_t)px * 16 + l] = make_float4(a.x, b.x, a.y, b.y);
    dst[(size_t)px * 16 + l + 8] = make_float4(a.z, b.z, a.w, b.w);
}
// B, a_lds: interleaved layout; l = lane of the query's 8-lane group. Same contract as pm_dist below.
// STAGE: how the three patch rows of an interior candidate are fetched. The level is bound by the latency of DEPENDENT fetches (row -> partial sum -> test -> next row), so:
//   0  one row at a time, rejection test after each of the first two (three round trips for a survivor);
//   1  the first row alone (most far random samples stop here), then the other two TOGETHER (two round trips);
//   2  the whole tile at once (one round trip): for candidates that are rarely rejected early — propagated matches and near random samples (radius < NCT_PM_FAR_MAG).
// The fmaf chains run in the same order in every form, and an early return only ever replaces a value that could not win: same bits.
template <int MODE, int RW, int STAGE = 0>
__device__ __forceinline__ float pm_dist8(const float* __restrict__ B, const PMGeom& g, int ax, int ay, unsigned amask, int bx, int by, int l,
                                          const float4* __restrict__ a_lds, int lx, int ly, float need) {
    constexpr bool EX = MODE == NCT_PM_ROWREJECT;
    constexpr int C4 = 16;
    const bool inside = amask == 0x1FFu && bx >= 1 && bx < g.bw - 1 && by >= 1 && by < g.bh - 1;
    if (__builtin_amdgcn_ballot_w64(inside) == __builtin_amdgcn_ballot_w64(true)) {
        const float4* pbc = reinterpret_cast<const float4*>(B) + (size_t)(unsigned)(by * g.bw + bx) * C4 + l;
        const float4* pac = a_lds + ((ly + 1) * RW + (lx + 1)) * C4 + l;
        pm_f2 acc = {0.f, 0.f};
        if constexpr (STAGE == 0 || !EX) {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {                 // one patch row at a time; with EX a hopeless candidate stops after a row
                const float4* pbr = pbc + dy * g.bw * C4;
                const float4* par = pac + dy * RW * C4;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) acc = pk_dot4_acc(par[dx * C4], par[dx * C4 + 8], pbr[dx * C4], pbr[dx * C4 + 8], acc);
                if (EX && dy < 1 && need > -FLT_MAX) {
                    const float rem = dy < 0 ? 6.0007f : 3.0004f;
                    if (half8_sum(acc.x, acc.y) + rem < need) return FLT_MAX;
                }
            }
        } else {
            float4 t[3][3][2];                                 // [row][dx][half of the pixel record]
            if constexpr (STAGE == 2) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) { t[dy + 1][dx + 1][0] = pbc[(dy * g.bw + dx) * C4]; t[dy + 1][dx + 1][1] = pbc[(dy * g.bw + dx) * C4 + 8]; }
            } else {
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) { t[0][dx + 1][0] = pbc[(-g.bw + dx) * C4]; t[0][dx + 1][1] = pbc[(-g.bw + dx) * C4 + 8]; }
            }
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) acc = pk_dot4_acc(pac[(-RW + dx) * C4], pac[(-RW + dx) * C4 + 8], t[0][dx + 1][0], t[0][dx + 1][1], acc);
            if constexpr (STAGE == 1) {
                if (need > -FLT_MAX && half8_sum(acc.x, acc.y) + 6.0007f < need) return FLT_MAX;
#pragma unroll
                for (int dy = 0; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) { t[dy + 1][dx + 1][0] = pbc[(dy * g.bw + dx) * C4]; t[dy + 1][dx + 1][1] = pbc[(dy * g.bw + dx) * C4 + 8]; }
            }
#pragma unroll
            for (int dy = 0; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) acc = pk_dot4_acc(pac[(dy * RW + dx) * C4], pac[(dy * RW + dx) * C4 + 8], t[dy + 1][dx + 1][0], t[dy + 1][dx + 1][1], acc);
        }
        const float sfull = half8_sum(acc.x, acc.y);
        if (EX && need > -FLT_MAX && sfull + 1e-4f < need) return FLT_MAX;
        return (-sfull) / 9.0f;
    }
    // border queries / candidates (a few percent of the waves): a patch row at a time, so that this path does not set the kernel's register count
    pm_f2 acc = {0.f, 0.f};
    int n = 0;
#pragma unroll 1
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = by + dy;
        const int yc = clampi(yy, 0, g.bh - 1);
        float4 b0[3], b1[3]; bool valid[3];
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = bx + dx, t = (dy + 1) * 3 + dx + 1;
            valid[dx + 1] = ((amask >> t) & 1u) && yy >= 0 && yy < g.bh && xx >= 0 && xx < g.bw;
            const float4* pb = reinterpret_cast<const float4*>(B + ((size_t)yc * g.bw + clampi(xx, 0, g.bw - 1)) * 64) + l;
            b0[dx + 1] = pb[0]; b1[dx + 1] = pb[8];
        }
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const float4* pa = a_lds + ((ly + 1 + dy) * RW + (lx + 1 + dx)) * C4 + l;
            n += valid[dx + 1] ? 1 : 0;
            float4 c0 = b0[dx + 1], c1 = b1[dx + 1], a0 = pa[0], a1 = pa[8];
            // adding +0 products == skipping the tap — with BOTH factors zeroed: the other side of a skipped tap may hold a NaN vector (a dead feature pixel: `norm` has
            // no epsilon), and NaN x 0 would poison a distance the reference computes without that tap (dist_compute_single tests the bounds of both pixels first)
            if (!valid[dx + 1]) { c0 = make_float4(0.f, 0.f, 0.f, 0.f); c1 = c0; a0 = c0; a1 = c0; }
            acc = pk_dot4_acc(a0, a1, c0, c1, acc);
        }
    }
    const float sum = half8_sum(acc.x, acc.y);
    return (n == 0) ? 1.0f : (-sum) / (float)n;
}

// ---- distance of query (ax,ay) to candidate (bx,by): -(sum over valid taps of <a,b>) / n_valid
// `need`: early-rejection threshold on the tap sum for UNIT-NORM features (every per-pixel vector has norm <= 1, so a tap adds at most 1
// by Cauchy-Schwarz): a candidate whose partial sum after a patch row cannot reach `need` any more cannot beat the current best and
// its remaining rows are not fetched (the caller gets FLT_MAX = "not better"). -FLT_MAX disables the test. That is MODE NCT_PM_ROWREJECT;
// NCT_PM_FP16 (opt-in reduced precision) reads the candidate tile from the fp16 shadow map Bh instead of B (fp32 accumulate).
template <int NCH, int MODE, int RW>
__device__ __forceinline__ float pm_dist(const float* __restrict__ A, const float* __restrict__ B, const uint2* __restrict__ Bh, const PMGeom& g, int ax, int ay, unsigned amask,
                                         int bx, int by, int v, const float4* __restrict__ a_lds, int lx, int ly, float need) {
    constexpr bool EX = MODE == NCT_PM_ROWREJECT, HALF = MODE == NCT_PM_FP16;
    // Fast path: when every tap of the query AND of the candidate lies inside its image — for every query of the wave, so the
    // branch is uniform — the nine B rows are the centre pointer plus wave-uniform offsets, the nine LDS rows are immediates,
    // nothing is masked and n = 9: ~70 VALU instructions per evaluation instead of ~270 (clamps, validity tests, selects and 64-bit
    // address arithmetic per tap). Same fmaf chain, same result. It buys 6.5 % at C = 64 and ~20 % at C = 128 (one patch row at a time,
    // to bound the loads in flight): the kernel is bound by the L2-miss traffic (5.4 TB/s on the fabric side, PMC FETCH_SIZE), not by
    // issue slots; for C >= 256 the fast path measured slower (the LDS-staged 36-73 KB query regions already limit occupancy), so
    // those instantiations keep the general loop.
    static_assert(NCH != 1, "C = 64 has its own 8-lane forms (pm_dist8, pm_dist8h)");
    if constexpr (NCH >= 1 && (HALF || NCH <= NCT_PM_FAST_MAX)) {
        const bool inside = amask == 0x1FFu && bx >= 1 && bx < g.bw - 1 && by >= 1 && by < g.bh - 1;
        if (__builtin_amdgcn_ballot_w64(inside) == __builtin_amdgcn_ballot_w64(true)) {
            constexpr int C4 = 16 * NCH;
            if constexpr (HALF) {
                // half-size tiles through FULL-WIDTH (16-byte) loads: what limits these levels is the number of vector-memory instructions and the
                // bytes the L1 returns, so the fp16 tile is read as 16-byte units (8 channels) — a pixel is 8 * NCH units: C = 128: 9 loads per
                // tile instead of 18, … (C = 64 has its own 8-lane form, pm_dist8h)
                const uint4* ph = reinterpret_cast<const uint4*>(Bh);
                float h = 0.f;
                {
                    constexpr int U = 8 * NCH;                     // 16-byte units per pixel (>= 16: every lane of the row reads U/16 units of one pixel)
                    auto tap = [&](int dy, int dx) {
                        const uint4* pp = ph + (size_t)(unsigned)((by + dy) * g.bw + bx + dx) * U + v;
                        const float4* pa = a_lds + ((ly + 1 + dy) * RW + (lx + 1 + dx)) * C4 + 2 * v;
#pragma unroll
                        for (int m = 0; m < U / 16; ++m) {
                            const uint4 raw = pp[16 * m];
                            h = dot4h_acc(pa[32 * m], make_uint2(raw.x, raw.y), h);
                            h = dot4h_acc(pa[32 * m + 1], make_uint2(raw.z, raw.w), h);
                        }
                    };
                    if constexpr (NCH <= 4) {                      // whole tile in flight (9 / 18 loads)
#pragma unroll
                        for (int t = 0; t < 9; ++t) tap(t / 3 - 1, t % 3 - 1);
                    } else {
                        // one patch row at a time, its 3 * U/16 requests issued together before the first use (written as "tap(dy, dx)" calls the
                        // compiler serialised them load -> wait -> fma at some register budgets: 2.1 vs 0.9 ms at the 44x44 level)
#pragma unroll 1
                        for (int dy = -1; dy <= 1; ++dy) {
                            uint4 raw[3][U / 16];
#pragma unroll
                            for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                                for (int m = 0; m < U / 16; ++m) raw[dx + 1][m] = ph[(size_t)(unsigned)((by + dy) * g.bw + bx + dx) * U + v + 16 * m];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int dx = -1; dx <= 1; ++dx) {
                                const float4* pa = a_lds + ((ly + 1 + dy) * RW + (lx + 1 + dx)) * C4 + 2 * v;
#pragma unroll
                                for (int m = 0; m < U / 16; ++m) {
                                    h = dot4h_acc(pa[32 * m], make_uint2(raw[dx + 1][m].x, raw[dx + 1][m].y), h);
                                    h = dot4h_acc(pa[32 * m + 1], make_uint2(raw[dx + 1][m].z, raw[dx + 1][m].w), h);
                                }
                            }
                        }
                    }
                }
                return (-row16_sum(h)) / 9.0f;
            }
            const float4* pbc = reinterpret_cast<const float4*>(B) + (size_t)(unsigned)(by * g.bw + bx) * C4 + v;
            const float4* pac = a_lds + ((ly + 1) * RW + (lx + 1)) * C4 + v;
            float facc = 0.f;
            if constexpr (EX) {
                // one patch row at a time; a hopeless candidate stops after a row (margins: a tap of unit vectors adds <= 1 + 2e-6, fp32 accumulation
                // error < 1e-4). Two rows per round trip as in pm_dist8 measured slower here: 6.56 vs 6.25 ms for the C = 128 level
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const float4* pbr = pbc + dy * g.bw * C4;
                    const float4* par = pac + dy * RW * C4;
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                        for (int k = 0; k < NCH; ++k) facc = dot4_acc(par[dx * C4 + 16 * k], pbr[dx * C4 + 16 * k], facc);
                    if (dy < 1 && need > -FLT_MAX) {
                        const float rem = dy < 0 ? 6.0007f : 3.0004f;
                        if (row16_sum(facc) + rem < need) return FLT_MAX;
                    }
                }
                // the complete sum: a candidate 1e-4 short of `need` = -9 dbest loses by > 1e-5 in distance, far outside the rounding of the
                // product and of the division — rejected without paying for the correctly rounded division (~10 VALU instructions)
                const float sfull = row16_sum(facc);
                if (need > -FLT_MAX && sfull + 1e-4f < need) return FLT_MAX;
                return (-sfull) / 9.0f;
            } else {
                // one patch row (3 taps x NCH chunks) at a time: bounds the loads in flight, and with them the register count
#pragma unroll 1
                for (int dy = -1; dy <= 1; ++dy) {
                    const float4* pbr = pbc + dy * g.bw * C4;
                    const float4* par = pac + dy * RW * C4;
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                        for (int k = 0; k < NCH; ++k) facc = dot4_acc(par[dx * C4 + 16 * k], pbr[dx * C4 + 16 * k], facc);
                }
            }
            return (-row16_sum(facc)) / 9.0f;
        }
    }
    float acc = 0.f;
    int n = 0;
    const int nchunk = g.C >> 2;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const int yy = by + dy, xx = bx + dx;
        const bool valid = ((amask >> t) & 1u) && yy >= 0 && yy < g.bh && xx >= 0 && xx < g.bw;
        const int yc = clampi(yy, 0, g.bh - 1), xc = clampi(xx, 0, g.bw - 1);
        const float4* pb = reinterpret_cast<const float4*>(B + ((size_t)yc * g.bw + xc) * g.C);
        n += valid ? 1 : 0;
        const int yac = clampi(ay + dy, 0, g.ah - 1), xac = clampi(ax + dx, 0, g.aw - 1);
        // C = 64*NCH: the workgroup's 6x6xC query region sits in LDS (a_lds); generic C reads the query tile through L1
        const float4* pa = (NCH >= 1) ? a_lds + ((ly + 1 + dy) * RW + (lx + 1 + dx)) * (g.C >> 2)
                                      : reinterpret_cast<const float4*>(A + ((size_t)yac * g.aw + xac) * g.C);
        if constexpr (NCH > 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                float4 a = pa[v + 16 * k];
                if (!valid) a = make_float4(0.f, 0.f, 0.f, 0.f);         // both factors: see pm_dist8
                if constexpr (HALF) {
                    uint2 b = Bh[((size_t)yc * g.bw + xc) * (size_t)nchunk + v + 16 * k];
                    if (!valid) b = make_uint2(0u, 0u);
                    acc = dot4h_acc(a, b, acc);
                } else {
                    float4 b = pb[v + 16 * k];
                    if (!valid) b = make_float4(0.f, 0.f, 0.f, 0.f);     // adding +0 products == skipping the tap
                    acc = dot4_acc(a, b, acc);
                }
            }
        } else {
            for (int j = v; j < nchunk; j += 16) {
                float4 a = pa[j];
                float4 b = pb[j];
                if (!valid) { b = make_float4(0.f, 0.f, 0.f, 0.f); a = b; }
                acc = dot4_acc(a, b, acc);
            }
        }
    }
    const float sum = row16_sum(acc);
    return (n == 0) ? 1.0f : (-sum) / (float)n;
}

// ---- NNF word / distance accesses. PLAIN (COH = false): ordinary loads and stores — the one-launch-per-step kernels, whose steps are ordered by kernel boundaries.
// COHERENT (COH = true): system-scope relaxed accesses (global_load / global_store ... sc0 sc1: served by memory, never by a CU's L1 or an XCD's L2) — the persistent
// level kernel, whose steps are ordered by per-tile flags inside ONE launch, where a neighbour tile's words were written by a workgroup of another XCD microseconds ago
// (MI355X_MICROARCH.md "inter-workgroup visibility": {sc0 sc1 stores and loads on both sides} is a valid hand-off without any fence). The feature maps are read-only
// and stay ordinary (cached) loads in both.
#ifndef NCT_PM_STEP_COH
#define NCT_PM_STEP_COH false      // experiment hooks: the per-step kernels with coherent accesses / the level kernel with plain ones (wrong results, timing only)
#endif
#ifndef NCT_PM_LEVEL_COH
#define NCT_PM_LEVEL_COH true
#endif
#ifndef NCT_PM_COH_SYSTEM
#define PM_COH_SCOPE __HIP_MEMORY_SCOPE_AGENT        // sc1: loads bypass the L1 and are served by the XCD's L2, stores write through
#else
#define PM_COH_SCOPE __HIP_MEMORY_SCOPE_SYSTEM       // sc0 sc1: served by memory (measured: 1.5 - 2.4 x slower levels — every NNF word a fabric request to a handful of channels)
#endif
template <bool COH> __device__ __forceinline__ uint32_t pm_ld(const uint32_t* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, PM_COH_SCOPE); else return *p;
}
template <bool COH> __device__ __forceinline__ float pm_ld(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, PM_COH_SCOPE); else return *p;
}
template <bool COH> __device__ __forceinline__ void pm_st(uint32_t* p, uint32_t v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, PM_COH_SCOPE); else *p = v;
}
template <bool COH> __device__ __forceinline__ void pm_st(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, PM_COH_SCOPE); else *p = v;
}

// mode 0: init (dist of the current NNF, no cutoff); mode 1: propagation step with `jump`; random search if jump==1
// A launch carries up to two independent jobs (the S->R and the R->S field of one level): workgroups [0, nblk0) belong to
// job 0, the rest to job 1. Fusing the two directions doubles the number of resident workgroups at the coarse levels
// (44x44 queries are only 121 workgroups for 256 CUs) and halves the launch count; each job's result is unaffected.
struct PMJob { const float* A; const float* B; const uint2* Bh; const uint32_t* nnf_in; const float* d_in; uint32_t* nnf_out; float* d_out; PMGeom g; int rs_max; uint32_t seed; };

// TQX x TQY = 4x4 query sub-tiles per workgroup: the workgroup stages the (4 TQX + 2) x (4 TQY + 2) x C region of A once and then walks its
// sub-tiles one after the other (16 queries at a time, one 16-lane row per query as before). A launch of the round-1 kernel (one sub-tile per
// workgroup) spent ~165 of its 557 us at 700x700 before the first evaluation: 61 000 workgroups per launch, each paying the serial prologue
// "stage the region -> barrier -> NNF -> neighbour NNF -> first tile" with only four workgroups resident per CU. Larger tiles pay it once per
// 64 queries and stage 1.56 instead of 2.25 region pixels per query. Which query a thread serves does not enter any value: results are unchanged.
// The staged region limits C = 512 to two workgroups per CU (2 waves per SIMD): telling the compiler so (second launch-bounds argument) lets its scheduler
// keep a patch row's loads in flight together instead of serialising them to save registers for an occupancy the LDS rules out.
// LPQ = 8 (C = 64, 128 with fp32 tiles): 32 queries per pass, an 8x4 sub-tile, two queries per DPP row (see pm_dist).
// One tile of one step (the body of k_pm_step and of the persistent k_pm_level): the workgroup serves the 4 TQX x 4 TQY queries of tile (tx, ty) of job J.
// s_a: the dynamic LDS region for the staged part of A. nevals / naccept: per-thread counters, accumulated.
// The staged region of A: RW x RH pixels x C floats, 256 threads. All of a thread's 16-byte loads are issued before the first LDS store (batches of at most nine): written as
// "for (e = tid; e < N; e += 256) s_a[e] = A[..]" hipcc emitted load -> s_waitcnt vmcnt(0) -> ds_write per trip — six to eighteen DEPENDENT global round trips at the head
// of every workgroup, the larger part of the per-launch floor of the propagation steps (round 6, from the ISA).
template <int NCH, int RW, int RH>
__device__ __forceinline__ void pm_stage_region(const float* __restrict__ A, const PMGeom& g, int ox, int oy, float4* __restrict__ s_a) {
    constexpr int c4 = 16 * NCH, N = RW * RH * c4, NIT = (N + 255) / 256, BATCH = NIT <= 9 ? NIT : 9;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
        float4 tmp[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e0 = (int)threadIdx.x + 256 * (it0 + u), e = e0 < N ? e0 : N - 1;
            const int r = e / c4, j = e - r * c4;
            const int ry = clampi(oy - 1 + r / RW, 0, g.ah - 1), rx = clampi(ox - 1 + r % RW, 0, g.aw - 1);
            tmp[u] = reinterpret_cast<const float4*>(A + ((size_t)ry * g.aw + rx) * (64 * NCH))[j];
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e0 = (int)threadIdx.x + 256 * (it0 + u);
            if (it0 + u < NIT && e0 < N) s_a[e0] = tmp[u];
        }
    }
}

template <int NCH, int MODE, int TQX, int TQY, int LPQ, bool COH>
__device__ __forceinline__ void pm_step_tile(const PMJob& J, int tx, int ty, int mode, int jump, int iter, int tstep, int strip, float4* __restrict__ s_a, unsigned& nevals, unsigned& naccept) {
    constexpr int RW = 4 * TQX + 2, RH = 4 * TQY + 2;
    const float* __restrict__ A = J.A; const float* __restrict__ B = J.B; const uint2* __restrict__ Bh = J.Bh;
    constexpr bool EX = MODE == NCT_PM_ROWREJECT;
    const uint32_t* nnf_in = J.nnf_in; const float* d_in = J.d_in;
    uint32_t* nnf_out = J.nnf_out; float* d_out = J.d_out;
    const PMGeom g = J.g; const int rs_max = J.rs_max; const uint32_t seed = J.seed;
    constexpr int QW = LPQ == 16 ? 4 : 8, QH = 256 / LPQ / QW;      // the sub-tile of one pass: QW x QH queries (4x4 or 8x4)
    static_assert((LPQ == 16 || (LPQ == 8 && NCH == 1 && MODE != NCT_PM_FP16)) && (4 * TQX) % QW == 0 && (4 * TQY) % QH == 0, "8 lanes per query: C = 64, fp32 tiles");
    const int grp = threadIdx.x / LPQ, lane = threadIdx.x % LPQ;
    const int v = lane;                                     // LPQ 16: the lane's channel chunk; LPQ 8: its 32-byte record of the interleaved pixel
    const int ox = tx * 4 * TQX, oy = ty * 4 * TQY;        // origin of the workgroup's query tile

    constexpr int NSX = 4 * TQX / QW, NSUB = NSX * (4 * TQY / QH);
    // The NNF word, distance and four neighbour words of a pass's query do not depend on the previous pass: they are requested one pass ahead, so the dependent chain of a
    // pass is candidate rows only (the chain — NNF words -> first row -> other rows, per pass, per round of resident workgroups — is what a late propagation launch costs)
    struct QIn { uint32_t vbest; float d; uint32_t vnb[4]; };
    auto qfetch = [&](int sub, QIn& q) {
        const int sx = sub % NSX, sy = sub / NSX;
        const int qx = ox + sx * QW + (grp % QW), qy = oy + sy * QH + (grp / QW);
        const int ax = qx < g.aw ? qx : g.aw - 1, ay = qy < g.ah ? qy : g.ah - 1;
        const int qi = ay * g.aw + ax;
        q.vbest = pm_ld<COH>(nnf_in + qi);
        if (mode != 0) {
            q.d = pm_ld<COH>(d_in + qi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nx = ax + ((k == 0) ? -jump : (k == 1 ? jump : 0)), ny = ay + ((k == 2) ? -jump : (k == 3 ? jump : 0));
                q.vnb[k] = pm_ld<COH>(nnf_in + clampi(ny, 0, g.ah - 1) * g.aw + clampi(nx, 0, g.aw - 1));
            }
        }
    };
    // (C = 64 only: at C = 128 the six extra registers cost the fourth wave per SIMD — 6.98 vs 6.22 ms for that level, 11.4 with the first pass's words requested early as
    //  well. The first pass's words are requested in front of the staging loop, whose barrier they then fly under.)
    constexpr bool AHEAD = LPQ == 8 && NSUB > 1;
    QIn qnext;
    if constexpr (AHEAD) qfetch(0, qnext);

    // stage the region of A that the queries of this workgroup read (their 3x3 tiles overlap) into LDS once per launch. Without it every
    // evaluation re-reads its 9*C*4-byte query tile through L1.
    if constexpr (NCH >= 1) {
        pm_stage_region<NCH, RW, RH>(A, g, ox, oy, s_a);
        __syncthreads();
    }
    int rs_start = rs_max;
    { const int mx = g.bw > g.bh ? g.bw : g.bh; if (rs_start > mx) rs_start = mx; }
    int nrand = 0;
    if (jump == 1) for (int mag = rs_start; mag >= 1; mag >>= 1) ++nrand;

    QIn qcur;
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
        if constexpr (AHEAD) { qcur = qnext; if (sub + 1 < NSUB) qfetch(sub + 1, qnext); }
        else qfetch(sub, qcur);
        const int sx = sub % NSX, sy = sub / NSX;
        const int qx = ox + sx * QW + (grp % QW), qy = oy + sy * QH + (grp / QW);
        if (NSUB > 1 && ox + sx * QW >= g.aw) continue;                 // sub-tile entirely outside the image (uniform over the workgroup)
        if (NSUB > 1 && oy + sy * QH >= g.ah) continue;
        const bool live = qx < g.aw && qy < g.ah;
        // a dead query (beyond the image in a partially filled tile) is clamped PER AXIS, so it stays inside its tile's staged region and has the tap mask of
        // a live border query
        const int ax = qx < g.aw ? qx : g.aw - 1, ay = qy < g.ah ? qy : g.ah - 1;
        // (it walks the candidates of that border query and writes nothing; masking dead queries out of every evaluation instead measured slower: the random
        // search's validity became a per-lane value)
        const int qi = ay * g.aw + ax;
        const int lx = ax - ox, ly = ay - oy;

        // validity of the query's own taps
        unsigned amask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const int yy = ay + dy, xx = ax + dx;
            amask |= ((yy >= 0 && yy < g.ah && xx >= 0 && xx < g.aw) ? 1u : 0u) << t;
        }

        const uint32_t vbest = qcur.vbest;
        int xbest = nnf_x(vbest), ybest = nnf_y(vbest);
        const int x0 = xbest, y0 = ybest;
        float dbest;

        if (mode == 0) {
            if constexpr (LPQ == 8) dbest = pm_dist8<MODE, RW>(B, g, ax, ay, amask, xbest, ybest, v, s_a, lx, ly, -FLT_MAX);
            else dbest = pm_dist<NCH, MODE, RW>(A, B, Bh, g, ax, ay, amask, xbest, ybest, v, s_a, lx, ly, -FLT_MAX);
            float cut = (float)INT_MAX;                 // dist_single default cutoff
            if (dbest >= cut) dbest = cut;
            if (live && v == 0) nevals += 1;
        } else {
            dbest = qcur.d;
            // the four neighbours' matches are independent of the evaluations (inside the candidate loop each would be one more dependent round trip per candidate)
            const uint32_t (&vnb)[4] = qcur.vnb;
            // ---- propagation candidates: 0 left, 1 right, 2 up, 3 down — the neighbour's match shifted back by the jump. Each query first packs the
            // candidates that can still win into a short list (in that order), then the wave walks the lists round by round: with most candidates
            // dropping out (below) a wave needs max-over-its-queries rounds instead of four, and no round is spent on a slot whose queries all sit out.
            uint32_t cl0 = 0, cl1 = 0, cl2 = 0, cl3 = 0; int ncl = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sxx = (k == 0) ? -jump : (k == 1 ? jump : 0);
                const int syy = (k == 2) ? -jump : (k == 3 ? jump : 0);
                const int nx = ax + sxx, ny = ay + syy;
                const uint32_t vp = vnb[k];
                const int xp = nnf_x(vp) - sxx, yp = nnf_y(vp) - syy;
                bool valid = nx >= 0 && nx < g.aw && ny >= 0 && ny < g.ah && yp >= 0 && yp < g.bh && xp >= 0 && xp < g.bw;
#ifndef NCT_PM_EVAL_STALE
                // A neighbour whose match has not changed since the step of the PREVIOUS iteration with this jump (four steps ago) proposes the candidate that
                // was evaluated then and lost (or won and has been the best, or been beaten, since): the query's best only ever decreases, so d >= dbest
                // again — it cannot win and is not fetched. Exact: same NNF, same distances (the oracle evaluates everything). The step a match last
                // changed in travels in the top byte of its (double-buffered) NNF word, so the test costs no load. As the field converges — 70-95 % of the
                // matches are unchanged from iteration 3 on — most propagation candidates drop out; the random search is always fresh.
                valid = valid && !(tstep > 4 && (int)(vp >> 24) < tstep - 4);
#endif
                if (valid) {
                    const uint32_t c = xy_pack(xp, yp);
                    cl0 = ncl == 0 ? c : cl0; cl1 = ncl == 1 ? c : cl1; cl2 = ncl == 2 ? c : cl2; cl3 = ncl == 3 ? c : cl3;
                    ++ncl;
                }
            }
            int nprop = 0;                                    // wave-uniform: the longest list among the wave's queries
#pragma unroll
            for (int i = 1; i <= 4; ++i) nprop += __builtin_amdgcn_ballot_w64(ncl >= i) != 0 ? 1 : 0;
            int mag = rs_start;
            const int ncand = nprop + nrand;
            for (int k = 0; k < ncand; ++k) {
                int xp, yp; bool valid; float rr; bool far = false;
                if (k < nprop) {
                    const uint32_t c = k == 0 ? cl0 : (k == 1 ? cl1 : (k == 2 ? cl2 : cl3));
                    xp = nnf_x(c); yp = nnf_y(c);
                    valid = k < ncl;
#ifndef NCT_PM_EVAL_SAME
                    // a neighbour that proposes the current match cannot improve it (d == dbest is not < dbest): its lanes sit the evaluation out
                    // (with the wave near the L1 bandwidth limit the unissued tile requests are what is saved, not instructions)
                    valid = valid && !(xp == xbest && yp == ybest);
#endif
                    rr = 0.f;
                } else {
                    const int step = k - nprop;
                    const int xmin = max(xbest - mag, 0), xmax = min(xbest + mag + 1, g.bw);
                    const int ymin = max(ybest - mag, 0), ymax = min(ybest + mag + 1, g.bh);
                    // (int)(u * w) % w with u in (0, 1]: the product never exceeds w, so the modulo only folds the value w back to 0 — a
                    // select instead of two integer divisions
                    const int wx = xmax - xmin, wy = ymax - ymin;
                    const int rx = (int)(rand_u01(seed, ax, ay, iter, step, 0) * (float)wx), ry = (int)(rand_u01(seed, ax, ay, iter, step, 1) * (float)wy);
                    xp = xmin + (rx == wx ? 0 : rx);
                    yp = ymin + (ry == wy ? 0 : ry);
                    far = mag >= NCT_PM_FAR_MAG;
                    mag >>= 1;
                    valid = true; rr = FLT_MIN;
                }
                if (valid) {
                    // to win, -sum/9 (+rr) < dbest, i.e. sum > -9 dbest: unreachable sums are cut off (unit-norm features only)
                    float d;
                    if constexpr (LPQ == 8) {
                        // (wave-uniform branch: the search radius is the same for every query of a step)
                        if constexpr (NCT_PM_FAR_STAGE == NCT_PM_NEAR_STAGE) d = pm_dist8<MODE, RW, NCT_PM_FAR_STAGE>(B, g, ax, ay, amask, xp, yp, v, s_a, lx, ly, EX ? -9.0f * dbest : -FLT_MAX);   // one inlined copy, not two
                        else if (far) d = pm_dist8<MODE, RW, NCT_PM_FAR_STAGE>(B, g, ax, ay, amask, xp, yp, v, s_a, lx, ly, EX ? -9.0f * dbest : -FLT_MAX);
                        else d = pm_dist8<MODE, RW, NCT_PM_NEAR_STAGE>(B, g, ax, ay, amask, xp, yp, v, s_a, lx, ly, EX ? -9.0f * dbest : -FLT_MAX);
                    }
                    else d = pm_dist<NCH, MODE, RW>(A, B, Bh, g, ax, ay, amask, xp, yp, v, s_a, lx, ly, EX ? -9.0f * dbest : -FLT_MAX);
                    if (d >= dbest) d = dbest;                       // cutoff clamp of dist_compute_single
                    if (d + rr < dbest) { xbest = xp; ybest = yp; dbest = d; if (live && v == 0) ++naccept; }
                    if (live && v == 0) ++nevals;
                }
            }
        }
        if (live && v == 0) {
            if (mode != 0) {
                // top byte: the step this match last changed in (0 = still the initial one); the last step of a run writes the plain (y << 12) | x word
                const uint32_t stamp = (xbest != x0 || ybest != y0) ? (uint32_t)tstep : (vbest >> 24);
                pm_st<COH>(nnf_out + qi, xy_pack(xbest, ybest) | (strip ? 0u : stamp << 24));
            }
            pm_st<COH>(d_out + qi, dbest);
        }
    }
}

// XCD-aware tile order of the one-launch-per-step kernels: consecutive workgroup ids round-robin over the 8 XCDs; give each XCD a contiguous
// band of tiles so that overlapping candidate tiles of neighbouring queries meet in the same L2.
__device__ __forceinline__ int pm_xcd_tile(int bid, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void pm_count(unsigned long long* __restrict__ counter, int v, unsigned nevals, unsigned naccept) {
    if (counter) {
        __shared__ unsigned s_cnt[2];
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        if (v == 0) { atomicAdd(&s_cnt[0], nevals); atomicAdd(&s_cnt[1], naccept); }
        __syncthreads();
        if (threadIdx.x < 2) atomicAdd(counter + threadIdx.x, (unsigned long long)s_cnt[threadIdx.x]);
    }
}

template <int NCH, int MODE, int TQX, int TQY, int LPQ>
__global__ __launch_bounds__(256, NCH == 8 ? 2 : (LPQ == 8 ? NCT_PM_OCC8 : 1)) void k_pm_step(PMJob j0, PMJob j1, int nblk0, int mode, int jump, int iter,
                                                 int tstep, int strip, unsigned long long* __restrict__ counter) {
    const bool second = (int)blockIdx.x >= nblk0;
    const PMJob& J = second ? j1 : j0;
    const int bid = pm_xcd_tile((int)blockIdx.x - (second ? nblk0 : 0), J.g.tiles_x * J.g.tiles_y);
    const int ty = bid / J.g.tiles_x, tx = bid - ty * J.g.tiles_x;
    extern __shared__ float4 s_a[];
    unsigned nevals = 0, naccept = 0;
    pm_step_tile<NCH, MODE, TQX, TQY, LPQ, NCT_PM_STEP_COH>(J, tx, ty, mode, jump, iter, tstep, strip, s_a, nevals, naccept);
    pm_count(counter, (int)(threadIdx.x % LPQ), nevals, naccept);
}

// ---- C = 64 / 128, propagation-only steps (jump 8 / 4 / 2: three of the four launches of an iteration) with the candidates of a wave's SIXTEEN (C = 128: eight) queries packed (round 4).
// k_pm_step lets each query walk its own short candidate list, so a wave spends max-over-its-eight-queries rounds per pass — with 1.5 live candidates per query on
// average (0.4 in late iterations) half to two thirds of the lane groups idle in every round, and the rounds are a dependent chain (tile rows -> sum -> next). Here every wave
// first lists the live (query, candidate) pairs of BOTH passes (its 16 queries) in LDS, then its eight lane groups take eight list entries per round whatever query they belong
// to, and finally each query scans ITS results in candidate order with the reference's accept rule (d < dbest, first wins). A propagation candidate's distance does not depend
// on the other candidates of its query — only the early-rejection threshold did, and the initial dbest is a valid (weaker) threshold — so NNF and distances are the same bits.
// Rounds per wave: ceil(live candidates / 8) instead of sum over the passes of the longest list.
#ifndef NCT_PM_PACK
#define NCT_PM_PACK 3        // 0: off, 1: C = 64 only, 2: C = 64 and 128, 3: every level (C >= 256: one pass, the four queries of a wave share its lane groups)
#endif
#ifndef NCT_PM_PROP_OCC
#define NCT_PM_PROP_OCC NCT_PM_OCC8        // workgroups per CU the packed kernel is compiled for, and how it fetches a candidate's rows (pm_dist8 STAGE): one row at a time
#define NCT_PM_PROP_STAGE 0                // measured 11.41 ms for the finest level of a 700x700 pair vs 12.17 (first row, then two together) / 12.18 (whole tile); five workgroups per CU 11.53, six 13.5
#endif
template <int NCH, int MODE, int TQX, int TQY, int LPQ, bool COH>
__device__ __forceinline__ void pm_prop_tile(const PMJob& J, int tx, int ty, int jump, int tstep, int strip, float4* __restrict__ s_a, unsigned& nevals, unsigned& naccept) {
    constexpr int RW = 4 * TQX + 2, RH = 4 * TQY + 2, QW = LPQ == 16 ? 4 : 8, QH = 256 / LPQ / QW, NSX = 4 * TQX / QW, NSUB = NSX * (4 * TQY / QH);
    constexpr int GPW = 64 / LPQ;                          // lane groups (= queries per pass) of a wave
    static_assert(NSUB == 1 || NSUB == 2, "one or two passes per workgroup: 16 (C = 64), 8 (C = 128) or 4 (C >= 256) queries per wave");
    constexpr bool EX = MODE == NCT_PM_ROWREJECT;
    const float* __restrict__ A = J.A; const float* __restrict__ B = J.B; const uint2* __restrict__ Bh = J.Bh;
    const uint32_t* nnf_in = J.nnf_in; const float* d_in = J.d_in;
    uint32_t* nnf_out = J.nnf_out; float* d_out = J.d_out;
    const PMGeom g = J.g;
    const int grp = threadIdx.x / LPQ, v = threadIdx.x % LPQ, wv = threadIdx.x >> 6, gw = grp % GPW;
    const int ox = tx * 4 * TQX, oy = ty * 4 * TQY;
    __shared__ uint32_t s_list[4][4 * NSUB * GPW];     // per wave: live (slot, k, candidate) entries
    __shared__ uint4 s_q[4][NSUB * GPW];               // per wave and query slot: ax | ay << 16, lx | ly << 8 | amask << 16, dbest, -
    __shared__ float s_res[4][4 * NSUB * GPW];         // per wave: distance of candidate k of slot s at [s * 4 + k]

    // ---- the NNF words of both passes first (they fly under the staging loop's barrier)
    uint32_t vbest[NSUB], vnb[NSUB][4]; float dq[NSUB]; int qi[NSUB]; bool live[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int qx = ox + (sub % NSX) * QW + (grp % QW), qy = oy + (sub / NSX) * QH + (grp / QW);
        live[sub] = qx < g.aw && qy < g.ah;
        const int ax = live[sub] ? qx : 0, ay = live[sub] ? qy : 0;
        qi[sub] = ay * g.aw + ax;
        vbest[sub] = pm_ld<COH>(nnf_in + qi[sub]); dq[sub] = pm_ld<COH>(d_in + qi[sub]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int nx = ax + ((k == 0) ? -jump : (k == 1 ? jump : 0)), ny = ay + ((k == 2) ? -jump : (k == 3 ? jump : 0));
            vnb[sub][k] = pm_ld<COH>(nnf_in + clampi(ny, 0, g.ah - 1) * g.aw + clampi(nx, 0, g.aw - 1));
        }
    }
    pm_stage_region<NCH, RW, RH>(A, g, ox, oy, s_a);
    // ---- phase A: every query lists its live candidates (the rules of k_pm_step: inside both images, not stale, not the current match)
    uint32_t cl[NSUB][4]; int ncl[NSUB];
    int base = 0;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int ax = qi[sub] % g.aw, ay = qi[sub] / g.aw;
        const int xb0 = nnf_x(vbest[sub]), yb0 = nnf_y(vbest[sub]);
        ncl[sub] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cl[sub][k] = 0;
            const int sxx = (k == 0) ? -jump : (k == 1 ? jump : 0), syy = (k == 2) ? -jump : (k == 3 ? jump : 0);
            const int nx = ax + sxx, ny = ay + syy;
            const uint32_t vp = vnb[sub][k];
            const int xp = nnf_x(vp) - sxx, yp = nnf_y(vp) - syy;
            bool valid = live[sub] && nx >= 0 && nx < g.aw && ny >= 0 && ny < g.ah && yp >= 0 && yp < g.bh && xp >= 0 && xp < g.bw;
            valid = valid && !(tstep > 4 && (int)(vp >> 24) < tstep - 4);
            valid = valid && !(xp == xb0 && yp == yb0);
            if (valid) {
                const uint32_t c = xy_pack(xp, yp);
#pragma unroll
                for (int i = 0; i < 4; ++i) if (ncl[sub] == i) cl[sub][i] = c;
                ++ncl[sub];
            }
        }
        const int slot = sub * GPW + gw;
        if (v == 0) {
            unsigned amask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = ay + t / 3 - 1, xx = ax + t % 3 - 1;
                amask |= ((yy >= 0 && yy < g.ah && xx >= 0 && xx < g.aw) ? 1u : 0u) << t;
            }
            s_q[wv][slot] = make_uint4((unsigned)ax | ((unsigned)ay << 16), (unsigned)(ax - ox) | ((unsigned)(ay - oy) << 8) | (amask << 16), __float_as_uint(dq[sub]), 0u);
        }
        const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(v == 0 && ncl[sub] > i);
            if (v == 0 && ncl[sub] > i) s_list[wv][base + __builtin_popcountll(m & below)] = ((unsigned)slot << 26) | ((unsigned)i << 24) | cl[sub][i];
            base += __builtin_popcountll(m);
        }
    }
    __syncthreads();                                     // the staged region, the lists and the query records are complete
    // ---- phase B: one list entry per lane group and round
    for (int r = 0; r * GPW < base; ++r) {
        const int e = r * GPW + gw;
        if (e < base) {
            const uint32_t ent = s_list[wv][e];
            const int slot = ent >> 26, k = (ent >> 24) & 3;
            const uint4 q = s_q[wv][slot];
            const int ax = q.x & 0xFFFF, ay = q.x >> 16, lx = q.y & 0xFF, ly = (q.y >> 8) & 0xFF;
            const unsigned amask = q.y >> 16;
            const float dbest = __uint_as_float(q.z);
            float d;
            if constexpr (LPQ == 8) d = pm_dist8<MODE, RW, NCT_PM_PROP_STAGE>(B, g, ax, ay, amask, nnf_x(ent & 0xFFFFFFu), nnf_y(ent & 0xFFFFFFu), v, s_a, lx, ly, EX ? -9.0f * dbest : -FLT_MAX);
            else d = pm_dist<NCH, MODE, RW>(A, B, Bh, g, ax, ay, amask, nnf_x(ent & 0xFFFFFFu), nnf_y(ent & 0xFFFFFFu), v, s_a, lx, ly, EX ? -9.0f * dbest : -FLT_MAX);
            if (v == 0) s_res[wv][slot * 4 + k] = d;
        }
    }
    __syncthreads();
    // ---- phase C: every query scans its candidates in order (accept d < dbest: the first of equal distances wins, as in the sequential walk)
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        if (!(live[sub] && v == 0)) continue;
        const int slot = sub * GPW + gw;
        uint32_t best = vbest[sub] & 0xFFFFFFu; float dbest = dq[sub];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ncl[sub]) {
                float d = s_res[wv][slot * 4 + k];
                if (d >= dbest) d = dbest;
                if (d < dbest) { best = cl[sub][k]; dbest = d; ++naccept; }
                ++nevals;
            }
        const uint32_t stamp = best != (vbest[sub] & 0xFFFFFFu) ? (uint32_t)tstep : (vbest[sub] >> 24);
        pm_st<COH>(nnf_out + qi[sub], best | (strip ? 0u : stamp << 24));
        pm_st<COH>(d_out + qi[sub], dbest);
    }
}

template <int NCH, int MODE, int TQX, int TQY, int LPQ>
__global__ __launch_bounds__(256, LPQ == 8 ? NCT_PM_PROP_OCC : (NCH == 8 ? 2 : 1)) void k_pm_prop(PMJob j0, PMJob j1, int nblk0, int jump, int tstep, int strip, unsigned long long* __restrict__ counter) {
    const bool second = (int)blockIdx.x >= nblk0;
    const PMJob& J = second ? j1 : j0;
    const int bid = pm_xcd_tile((int)blockIdx.x - (second ? nblk0 : 0), J.g.tiles_x * J.g.tiles_y);
    const int ty = bid / J.g.tiles_x, tx = bid - ty * J.g.tiles_x;
    extern __shared__ float4 s_a[];
    unsigned nevals = 0, naccept = 0;
    pm_prop_tile<NCH, MODE, TQX, TQY, LPQ, NCT_PM_STEP_COH>(J, tx, ty, jump, tstep, strip, s_a, nevals, naccept);
    pm_count(counter, (int)(threadIdx.x % LPQ), nevals, naccept);
}

// ---- Round 6: one PERSISTENT launch per pyramid level — k_pm_level. The 1 + 4 iters Jacobi steps of a level used to be as many launches (41), each with its own
// fill and drain of ~15 rounds of resident workgroups (and at the coarse levels — 242 tiles per step for 44x44 queries — hardly anything but launch latency). Here the
// level's work items (step, tile) are handed out in step-major order by per-(step, XCD) ticket counters to however many workgroups are resident, and what orders two steps
// is DATA FLOW, not a grid barrier: tile T may run step s once T and the tiles a jump of 8 reaches in its row and its column have published step s - 1 (their words are
// read by T at step s — and T's own words of two steps ago, which step s overwrites, were read by exactly those tiles at step s - 1: the same set covers the write-after-read
// hazard of the double buffer). NNF words and distances cross workgroups through system-scope accesses (pm_ld / pm_st<true>), a tile's flag is stored after its words have
// been acknowledged (s_waitcnt vmcnt(0) + workgroup barrier): no fence anywhere, the feature maps stay L2-resident across steps.
// Placement independence (MI355X_MICROARCH.md: HIP promises no co-residency, and up to four pairs are in flight on one device): a workgroup draws tickets of step s + 1 only
// after it has seen all eight queues of step s exhausted, so whoever holds an item of step s + 1 waits only for items that resident workgroups already hold — a single
// resident workgroup would finish the level alone. The XCC id only picks the queue a workgroup tries first (its XCD's band of tiles, the same bands as the XCD-aware
// order of the per-step launches), never correctness. Spins are bounded by a wall-clock watchdog: on expiry (or when another workgroup has flagged one) the workgroup
// sets ctl[0] and leaves; the host reads the word at the end of the pair and fails it.
// Same arithmetic per (step, tile) as k_pm_step / k_pm_prop (the same device functions): NNFs and distances are the same bits.
struct PMLevel {
    PMJob j0, j1;                                       // nnf_in / d_in / nnf_out / d_out are set per step from the buffers below
    uint32_t* nn0[2]; float* dd0[2]; uint32_t* nn1[2]; float* dd1[2];       // double-buffered NNF words and distances of the two directions
    int nblk0, nblk1, nsteps, packed;
    uint32_t* err;                                      // context-lifetime word the host reads (nctk_pm_check): set when the watchdog fires
    uint32_t* ctl;                                      // [0] watchdog / error word, [16 + 8 s + x] ticket counter of (step s, XCD queue x), [PM_FLAG0 + tile] last step published + 1
    unsigned long long* counter;
    long long timeout_ticks;                            // wall_clock64 ticks (100 MHz)
};
constexpr int PM_MAX_STEPS = 256, PM_FLAG0 = 16 + 8 * PM_MAX_STEPS;
__device__ __forceinline__ int pm_band_start(int n, int x) { const int q = n >> 3, r = n & 7; return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q; }
__device__ __forceinline__ int pm_band_cnt(int n, int x) { return (n >> 3) + (x < (n & 7) ? 1 : 0); }

// uniform 32- / 64-bit values read back from LDS, forced into scalar registers (the tile bodies address the maps from SGPR bases, as in the per-step kernels)
__device__ __forceinline__ int pm_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T* pm_sgpr(T* p) {
    const unsigned long long u = (unsigned long long)p;
    // through a GLOBAL-address-space pointer: a pointer that comes out of LDS has lost what the compiler knew about kernel arguments, and every access through it would be a
    // FLAT instruction (aperture check, counted in lgkmcnt as well as vmcnt: the candidate loads would serialise with the LDS reads)
    typedef T __attribute__((address_space(1))) GT;
    return (T*)(GT*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)u));
}
// what a tile body is called with: written to LDS by lane 0 of wave 0, read back (into scalar registers) by every wave inside the body's own function
struct PMCall { PMJob J; int tx, ty, mode, jump, iter, tstep, strip, counting; };
typedef __attribute__((address_space(3))) float4 pm_lds_f4;
typedef __attribute__((address_space(3))) const PMCall pm_lds_call;
typedef __attribute__((address_space(3))) unsigned pm_lds_u32;
__device__ __forceinline__ PMJob pm_job_sgpr(pm_lds_call* c) {
    PMJob r;
    r.A = pm_sgpr(c->J.A); r.B = pm_sgpr(c->J.B); r.Bh = pm_sgpr(c->J.Bh);
    r.nnf_in = pm_sgpr(c->J.nnf_in); r.d_in = pm_sgpr(c->J.d_in); r.nnf_out = pm_sgpr(c->J.nnf_out); r.d_out = pm_sgpr(c->J.d_out);
    r.g.C = pm_sgpr(c->J.g.C); r.g.ah = pm_sgpr(c->J.g.ah); r.g.aw = pm_sgpr(c->J.g.aw); r.g.bh = pm_sgpr(c->J.g.bh); r.g.bw = pm_sgpr(c->J.g.bw);
    r.g.tiles_x = pm_sgpr(c->J.g.tiles_x); r.g.tiles_y = pm_sgpr(c->J.g.tiles_y); r.rs_max = pm_sgpr(c->J.rs_max); r.seed = (uint32_t)pm_sgpr((int)c->J.seed);
    return r;
}
// The two tile bodies as functions of their OWN (noinline): inlined into the persistent kernel next to each other and to the ticket logic, the scheduler gave up on
// clustering the candidate loads (one s_waitcnt vmcnt(0) behind every load of a patch row where the per-step kernels keep five in flight — the register pressure of the
// merged regions exceeded its occupancy target and it fell back to source order: levels 1.5 - 2.3 x slower). As functions they are scheduled and allocated like the
// per-step kernels; their arguments are LDS addresses, everything uniform is read back into scalar registers inside.
template <int NCH, int MODE, int TQX, int TQY, int LPQ>
__device__ __attribute__((noinline)) void pm_level_step_fn(pm_lds_f4* s_a, pm_lds_call* c, pm_lds_u32* cnt) {
    const PMJob J = pm_job_sgpr(c);
    unsigned nevals = 0, naccept = 0;
    pm_step_tile<NCH, MODE, TQX, TQY, LPQ, NCT_PM_LEVEL_COH>(J, pm_sgpr(c->tx), pm_sgpr(c->ty), pm_sgpr(c->mode), pm_sgpr(c->jump), pm_sgpr(c->iter), pm_sgpr(c->tstep), pm_sgpr(c->strip),
                                                             (float4*)s_a, nevals, naccept);
    if (pm_sgpr(c->counting) && (threadIdx.x % LPQ) == 0 && (nevals | naccept)) { atomicAdd((unsigned*)cnt, nevals); atomicAdd((unsigned*)cnt + 1, naccept); }
}
template <int NCH, int MODE, int TQX, int TQY, int LPQ>
__device__ __attribute__((noinline)) void pm_level_prop_fn(pm_lds_f4* s_a, pm_lds_call* c, pm_lds_u32* cnt) {
    const PMJob J = pm_job_sgpr(c);
    unsigned nevals = 0, naccept = 0;
    pm_prop_tile<NCH, MODE, TQX, TQY, LPQ, NCT_PM_LEVEL_COH>(J, pm_sgpr(c->tx), pm_sgpr(c->ty), pm_sgpr(c->jump), pm_sgpr(c->tstep), pm_sgpr(c->strip), (float4*)s_a, nevals, naccept);
    if (pm_sgpr(c->counting) && (threadIdx.x % LPQ) == 0 && (nevals | naccept)) { atomicAdd((unsigned*)cnt, nevals); atomicAdd((unsigned*)cnt + 1, naccept); }
}

template <int NCH, int MODE, int TQX, int TQY, int LPQ>
__global__ __launch_bounds__(256, NCH == 8 ? 2 : (LPQ == 8 ? NCT_PM_OCC8 : 1)) void k_pm_level(PMLevel L) {
    extern __shared__ float4 s_a[];
    __shared__ PMCall s_call;
    __shared__ int s_ok;
    __shared__ unsigned s_cnt2[2];
    if (threadIdx.x < 2) s_cnt2[threadIdx.x] = 0;
    constexpr int TW = 4 * TQX, TH = 4 * TQY, DX = (8 + TW - 1) / TW, DY = (8 + TH - 1) / TH;      // tiles a jump of 8 reaches in x / in y
    static_assert(2 * DX + 2 * DY + 1 <= 64, "one lane of wave 0 per dependency");
    const int lane = threadIdx.x & 63;
    unsigned home;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(home));
    home &= 7u;
    const int nt0 = L.nblk0, nt1 = L.nblk1, nsteps = L.nsteps;
    uint32_t* const ctl = L.ctl;
    int cur = 0, tryq = (int)home;                      // wave 0: the step this workgroup draws tickets from; the queue that gave it its last ticket (-1: look at all eight)
    for (;;) {
        int flag_tile = 0, flag_step = 0;
        if (threadIdx.x < 64) {                          // ---- wave 0: ticket, dependencies, the call record
            int step = -1, job = 0, tile = 0, ok = 0;
            while (cur < nsteps) {
                int xs = tryq;
                if (xs < 0) {                            // lanes 0..7 look at the eight counters of the step; the first open queue from home on is tried
                    const int x = lane & 7;
                    const uint32_t cv = pm_ld<true>(ctl + 16 + cur * 8 + x);
                    const unsigned open = (unsigned)__builtin_amdgcn_ballot_w64(lane < 8 && (int)cv < pm_band_cnt(nt0, x) + pm_band_cnt(nt1, x)) & 0xFFu;
                    if (!open) { ++cur; tryq = (int)home; continue; }        // every item of this step is in some resident workgroup's hands (or done): the next step may be drawn
                    const unsigned rot = ((open >> home) | (open << (8 - home))) & 0xFFu;
                    xs = (int)((home + (unsigned)__builtin_ctz(rot)) & 7u);
                }
                uint32_t c = 0;
                if (lane == 0) c = __hip_atomic_fetch_add(ctl + 16 + cur * 8 + xs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                const int c0 = pm_band_cnt(nt0, xs), c1 = pm_band_cnt(nt1, xs);
                if ((int)c < c0 + c1) {
                    step = cur; job = (int)c >= c0 ? 1 : 0; tile = job ? pm_band_start(nt1, xs) + ((int)c - c0) : pm_band_start(nt0, xs) + (int)c; ok = 1; tryq = xs;
                    break;
                }
                tryq = -1;                               // that queue is exhausted: look at all of them
            }
            const PMGeom& g = job ? L.j1.g : L.j0.g;
            const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
            if (ok && step > 0) {                        // dependencies: own tile, DX tiles either side in the row, DY either side in the column, all at step - 1
                int dx = 0, dy = 0; bool has = false;
                if (lane <= 2 * DX) { dx = lane - DX; has = true; }
                else if (lane <= 2 * DX + 2 * DY) { const int r = lane - 2 * DX - 1; dy = r < DY ? r - DY : r - DY + 1; has = true; }
                const int nx = tx + dx, ny = ty + dy;
                has = has && nx >= 0 && nx < g.tiles_x && ny >= 0 && ny < g.tiles_y;
                const uint32_t* f = ctl + PM_FLAG0 + (job ? nt0 : 0) + (has ? ny * g.tiles_x + nx : 0);
                const long long t0 = wall_clock64();
                for (;;) {
                    const uint32_t v = has ? pm_ld<true>(f) : (uint32_t)step;
                    if (__builtin_amdgcn_ballot_w64(v >= (uint32_t)step) == ~0ull) break;
                    if (pm_ld<true>(ctl) != 0u || wall_clock64() - t0 > L.timeout_ticks) { if (lane == 0) { pm_st<true>(ctl, 1u); pm_st<true>(L.err, 1u); } ok = -1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (lane == 0) {
                s_ok = ok;
                if (ok > 0) {
                    PMCall c;
                    c.J = job ? L.j1 : L.j0;
                    uint32_t* const* nn = job ? L.nn1 : L.nn0; float* const* dd = job ? L.dd1 : L.dd0;
                    c.tx = tx; c.ty = ty; c.counting = L.counter != nullptr; c.tstep = step;
                    if (step == 0) {                     // init: distances of the level's start field
                        c.J.nnf_in = nn[0]; c.J.d_in = dd[0]; c.J.nnf_out = nullptr; c.J.d_out = dd[0];
                        c.mode = 0; c.jump = 0; c.iter = 0; c.strip = 0;
                    } else {
                        const int in = (step - 1) & 1, out = step & 1;
                        c.J.nnf_in = nn[in]; c.J.d_in = dd[in]; c.J.nnf_out = nn[out]; c.J.d_out = dd[out];
                        c.iter = (step - 1) >> 2; c.jump = 8 >> ((step - 1) & 3); c.strip = step == nsteps - 1 ? 1 : 0;
                        c.mode = (c.jump != 1 && L.packed && MODE != NCT_PM_FP16) ? 2 : 1;
                    }
                    s_call = c;
                }
            }
            flag_tile = (job ? nt0 : 0) + tile; flag_step = step;
        }
        __syncthreads();
        if (s_ok <= 0) break;                            // no items left (0) or the watchdog fired (-1)
        if constexpr (MODE != NCT_PM_FP16) {
            if (s_call.mode == 2) pm_level_prop_fn<NCH, MODE, TQX, TQY, LPQ>((pm_lds_f4*)s_a, (pm_lds_call*)&s_call, (pm_lds_u32*)s_cnt2);
            else pm_level_step_fn<NCH, MODE, TQX, TQY, LPQ>((pm_lds_f4*)s_a, (pm_lds_call*)&s_call, (pm_lds_u32*)s_cnt2);
        } else pm_level_step_fn<NCH, MODE, TQX, TQY, LPQ>((pm_lds_f4*)s_a, (pm_lds_call*)&s_call, (pm_lds_u32*)s_cnt2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's words have been acknowledged by memory …
        __syncthreads();                                       // … and so have everybody's: the tile's step may be published (also: s_a and s_call are free again)
        if (threadIdx.x == 0) pm_st<true>(ctl + PM_FLAG0 + flag_tile, (uint32_t)(flag_step + 1));
    }
    __syncthreads();
    if (L.counter && threadIdx.x < 2) atomicAdd(L.counter + threadIdx.x, (unsigned long long)s_cnt2[threadIdx.x]);
}

// query tile of a workgroup per channel count: 8x8 at C = 64 (25 KB of LDS), 8x4 at C = 128 (31 KB), 4x4 above (37 / 74 KB)
template <int NCH> struct PMTile { static constexpr int TQX = NCH == 1 ? 2 : (NCH == 2 ? 2 : 1), TQY = NCH == 1 ? 2 : 1; };
// lanes per query: 8 at C = 64 (pm_dist8), else 16. Measured alternatives (two unpacked chains per lane): C = 128 with 8 lanes
// 10.7 vs 8.6 ms, C = 64 with 4 lanes 22.0 vs 17.6 ms per pair and level.
template <int NCH> struct PMLanes { static constexpr int LPQ = NCH == 1 ? 8 : 16; };
template <int NCH, int MODE>
static int launch_mode(nct_ctx* ctx, hipStream_t s, const PMJob& j0, const PMJob& j1, int nblk0, int nblk1, int mode, int jump, int iter, int tstep, int strip, unsigned long long* counter) {
    constexpr int TQX = PMTile<NCH>::TQX, TQY = PMTile<NCH>::TQY, LPQ = PMLanes<NCH>::LPQ;
    const size_t lds = NCH >= 1 ? (size_t)(4 * TQX + 2) * (4 * TQY + 2) * NCH * 16 * sizeof(float4) : 0;      // region pixels x C/4 float4
    // > 64 KB of dynamic LDS (C=512: 72 KB for the staged query region) needs the opt-in attribute on this device: set once per context (= per device) and instantiation
    constexpr unsigned abit = 1u << ((NCH > 8 ? 9 : NCH) * 3 + MODE);
    if (lds > 32768 && !(ctx->pm_attr_mask & abit)) {
        NCT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pm_step<NCH, MODE, TQX, TQY, LPQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->pm_attr_mask |= abit;
    }
    // k_pm_prop hard-codes the two exact skip rules (stale candidates, candidates equal to the current match); the checking builds that evaluate them anyway
    // (-DNCT_PM_EVAL_STALE / -DNCT_PM_EVAL_SAME) therefore take k_pm_step for every step, which honours the macros. Its eval counters also differ in meaning:
    // k_pm_prop filters against the query's INITIAL match of the step, so a candidate proposed by two neighbours is evaluated and counted twice (+0.08 % evaluations).
#if defined(NCT_PM_EVAL_STALE) || defined(NCT_PM_EVAL_SAME)
    constexpr bool packed_ok = false;
#else
    constexpr bool packed_ok = true;
#endif
    if constexpr (packed_ok && (NCH == 1 || (NCH == 2 && NCT_PM_PACK >= 2) || (NCH >= 4 && NCT_PM_PACK >= 3)) && MODE != NCT_PM_FP16 && NCT_PM_PACK != 0) {
        if (mode == 1 && jump != 1) {              // propagation-only step: the packed form (its dynamic LDS is the same staged region)
            if constexpr (NCH >= 4) {                  // > 32 KB of dynamic LDS: the same opt-in as k_pm_step, once per context and instantiation (bits 27..30 of the mask)
                constexpr unsigned pbit = 1u << (27 + (NCH == 8 ? 2 : 0) + (MODE == NCT_PM_ROWREJECT ? 1 : 0));
                if (!(ctx->pm_attr_mask & pbit)) {
                    NCT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pm_prop<NCH, MODE, TQX, TQY, LPQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    ctx->pm_attr_mask |= pbit;
                }
            }
            hipLaunchKernelGGL((k_pm_prop<NCH, MODE, TQX, TQY, LPQ>), dim3(nblk0 + nblk1), dim3(256), lds, s, j0, j1, nblk0, jump, tstep, strip, counter);
            NCT_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL((k_pm_step<NCH, MODE, TQX, TQY, LPQ>), dim3(nblk0 + nblk1), dim3(256), lds, s, j0, j1, nblk0, mode, jump, iter, tstep, strip, counter);
    NCT_LAUNCH_CHECK();
    return 0;
}

template <int NCH>
static int launch_step(nct_ctx* ctx, hipStream_t s, const PMJob& j0, const PMJob& j1, int nblk0, int nblk1, int mode, int jump, int iter, int tstep, int strip, unsigned long long* counter, int pm_mode) {
    // fp16 tiles from C = 128 on. The C = 64 level is latency bound — a wave walks its candidates one after the other — not byte bound: with
    // fp16 tiles on the 8-lane kernel (one 16-byte load per lane and tap) it took 15.8 vs 16.0 ms per pair, so it keeps the exact fp32 tiles.
    if constexpr (NCH >= 2) { if (pm_mode == NCT_PM_FP16) return launch_mode<NCH, NCT_PM_FP16>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, counter); }
    if (NCH == 1 && pm_mode == NCT_PM_FP16) pm_mode = NCT_PM_PLAIN;
    // unit-norm features (the pipeline): the instantiation with the exact early rejection; it exists for the C with an fp32 interior fast path
    if (NCH >= 1 && NCH <= NCT_PM_FAST_MAX && pm_mode == NCT_PM_ROWREJECT) return launch_mode<NCH, NCT_PM_ROWREJECT>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, counter);
    return launch_mode<NCH, NCT_PM_PLAIN>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, counter);
}

// one persistent launch for the whole level (k_pm_level); false = this channel count / mode has no persistent form
template <int NCH, int MODE>
static int launch_level_mode(nct_ctx* ctx, hipStream_t s, PMLevel& L) {
    constexpr int TQX = PMTile<NCH>::TQX, TQY = PMTile<NCH>::TQY, LPQ = PMLanes<NCH>::LPQ;
    const size_t lds = (size_t)(4 * TQX + 2) * (4 * TQY + 2) * NCH * 16 * sizeof(float4);
    const void* fn = reinterpret_cast<const void*>(&k_pm_level<NCH, MODE, TQX, TQY, LPQ>);
    if (lds > 32768) NCT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      // > 32 KB of dynamic LDS: per-device opt-in (a host-side call, five per pair)
    static int occ = 0;                                  // per instantiation; every device of this process is the same part
    if (!occ) { int o = 0; NCT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, 256, lds)); occ = o > 0 ? o : 1; }
#if defined(NCT_PM_EVAL_STALE) || defined(NCT_PM_EVAL_SAME)
    L.packed = 0;
#else
    L.packed = ((NCH == 1 || (NCH == 2 && NCT_PM_PACK >= 2) || (NCH >= 4 && NCT_PM_PACK >= 3)) && MODE != NCT_PM_FP16 && NCT_PM_PACK != 0) ? 1 : 0;
#endif
    int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int wgs = ctx->pm_persist_wgs > 0 ? ctx->pm_persist_wgs : cus * occ;
    if (wgs > L.nblk0 + L.nblk1) wgs = L.nblk0 + L.nblk1;
    hipLaunchKernelGGL((k_pm_level<NCH, MODE, TQX, TQY, LPQ>), dim3(wgs), dim3(256), lds, s, L);
    NCT_LAUNCH_CHECK();
    return 0;
}
template <int NCH>
static int launch_level(nct_ctx* ctx, hipStream_t s, PMLevel& L, int pm_mode) {
    if constexpr (NCH >= 2) { if (pm_mode == NCT_PM_FP16) return launch_level_mode<NCH, NCT_PM_FP16>(ctx, s, L); }
    if (NCH == 1 && pm_mode == NCT_PM_FP16) pm_mode = NCT_PM_PLAIN;
    if (NCH <= NCT_PM_FAST_MAX && pm_mode == NCT_PM_ROWREJECT) return launch_level_mode<NCH, NCT_PM_ROWREJECT>(ctx, s, L);
    return launch_level_mode<NCH, NCT_PM_PLAIN>(ctx, s, L);
}
int nctk_pm_check(nct_ctx* ctx) {
    if (!ctx->d_pm_err) return 0;
    uint32_t e = 0;
    NCT_HIP(hipMemcpyAsync(&e, ctx->d_pm_err, sizeof e, hipMemcpyDeviceToHost, ctx->stream));
    NCT_HIP(hipStreamSynchronize(ctx->stream));
    if (e) {
        (void)hipMemsetAsync(ctx->d_pm_err, 0, sizeof e, ctx->stream);
        return ctx->fail(NCT_ERR_HIP, "patchmatch: the persistent level kernel's watchdog fired (a tile waited > 0.25 s for a neighbour's step); results are invalid");
    }
    return 0;
}

// Runs one PatchMatch (bnn == nullptr) or both directions of a level fused in the same launches (A->B in ann, B->A in bnn).
static int pm_run(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, const void* a_h16, const void* b_h16, int C, int ah, int aw, int bh, int bw, int iters, int rs_max,
                  uint32_t seed_ab, uint32_t seed_ba, uint32_t* ann, float* annd, uint32_t* bnn, float* bnnd, unsigned long long* eval_counter, int pm_mode) {
    NCT_REQUIRE(C > 0 && (C & 3) == 0, "patchmatch: C=%d must be a positive multiple of 4", C);
    NCT_REQUIRE(ah >= 1 && aw >= 1 && bh >= 1 && bw >= 1 && ah < 4096 && aw < 4096 && bh < 4096 && bw < 4096,
                "patchmatch: dims out of range (%dx%d vs %dx%d); NNF coordinates are 12-bit", ah, aw, bh, bw);
    NCT_REQUIRE(iters >= 0 && rs_max >= 0, "patchmatch: iters/rs_max must be >= 0");
    NCT_REQUIRE(pm_mode >= NCT_PM_PLAIN && pm_mode <= NCT_PM_FP16, "patchmatch: unknown evaluation mode %d", pm_mode);
    const bool two = bnn != nullptr;
    if (pm_mode == NCT_PM_FP16) {
        NCT_REQUIRE(C == 64 || (b_h16 && (!two || a_h16)), "patchmatch: the fp16 mode needs the fp16 shadow maps");      // C = 64 stays fp32 (launch_step)
        NCT_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, "patchmatch: the fp16 mode exists for C = 64, 128, 256, 512 (got %d)", C);
    }
    const int na = ah * aw, nb = bh * bw;
    // C = 64 runs on lane-interleaved copies of both maps (pm_dist8): written once per PatchMatch from the natural maps
    const bool lanes8 = C == 64;
    DevBuf<float> a_il(ctx, lanes8 ? (size_t)na * 64 : 1), b_il(ctx, lanes8 ? (size_t)nb * 64 : 1);
    if (!a_il.ok() || !b_il.ok()) return NCT_ERR_HIP;
    if (lanes8) {
        hipLaunchKernelGGL(k_pm_interleave64, dim3(cdiv(na * 8, 256)), dim3(256), 0, s, (const float4*)a_hwc, (float4*)(float*)a_il, na); NCT_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_pm_interleave64, dim3(cdiv(nb * 8, 256)), dim3(256), 0, s, (const float4*)b_hwc, (float4*)(float*)b_il, nb); NCT_LAUNCH_CHECK();
        a_hwc = a_il; b_hwc = b_il;
    }
    DevBuf<uint32_t> a_tmp(ctx, na), b_tmp(ctx, two ? nb : 1);
    DevBuf<float> ad_tmp(ctx, na), bd_tmp(ctx, two ? nb : 1);
    if (!a_tmp.ok() || !b_tmp.ok() || !ad_tmp.ok() || !bd_tmp.ok()) return NCT_ERR_HIP;
    const int tqx = C == 64 ? PMTile<1>::TQX : (C == 128 ? PMTile<2>::TQX : 1), tqy = C == 64 ? PMTile<1>::TQY : (C == 128 ? PMTile<2>::TQY : 1);
    const PMGeom ga{C, ah, aw, bh, bw, cdiv(aw, 4 * tqx), cdiv(ah, 4 * tqy)}, gb{C, bh, bw, ah, aw, cdiv(bw, 4 * tqx), cdiv(bh, 4 * tqy)};
    const int nblk0 = ga.tiles_x * ga.tiles_y, nblk1 = two ? gb.tiles_x * gb.tiles_y : 0;
    uint32_t* na_buf[2] = {ann, a_tmp}; float* da_buf[2] = {annd, ad_tmp};
    uint32_t* nb_buf[2] = {bnn, b_tmp}; float* db_buf[2] = {bnnd, bd_tmp};
    // tstep: 1-based index of the propagation step (0 = init; 0 throughout when the step count does not fit the NNF word's spare byte: no stale-candidate
    // test then); strip: the run's last step writes NNF words without the step stamp
    const bool stamps = 4 * iters <= 250;
    auto step = [&](int in, int out, int mode, int jump, int iter, int tstep, int strip) -> int {
        PMJob j0{a_hwc, b_hwc, (const uint2*)b_h16, na_buf[in], da_buf[in], mode ? na_buf[out] : nullptr, da_buf[out], ga, rs_max, seed_ab};
        PMJob j1{b_hwc, a_hwc, (const uint2*)a_h16, nb_buf[in], db_buf[in], mode ? nb_buf[out] : nullptr, db_buf[out], gb, rs_max, seed_ba};
        switch (C) {
            case 64:  return launch_step<1>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, eval_counter, pm_mode);
            case 128: return launch_step<2>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, eval_counter, pm_mode);
            case 256: return launch_step<4>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, eval_counter, pm_mode);
            case 512: return launch_step<8>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, eval_counter, pm_mode);
            default:  return launch_step<0>(ctx, s, j0, j1, nblk0, nblk1, mode, jump, iter, tstep, strip, eval_counter, pm_mode == NCT_PM_FP16 ? NCT_PM_PLAIN : pm_mode);
        }
    };
    // ---- one persistent launch for the level (NCT_PM_PERSIST=1): the same steps, handed out as (step, tile) items inside k_pm_level
    if (ctx->pm_persist && stamps && 1 + 4 * iters <= PM_MAX_STEPS && (C == 64 || C == 128 || C == 256 || C == 512)) {
        if (!ctx->d_pm_err) { NCT_HIP(hipMalloc(&ctx->d_pm_err, sizeof(uint32_t))); NCT_HIP(hipMemsetAsync(ctx->d_pm_err, 0, sizeof(uint32_t), s)); }
        const size_t nctl = (size_t)PM_FLAG0 + nblk0 + nblk1;
        DevBuf<uint32_t> ctl(ctx, nctl);
        if (!ctl.ok()) return NCT_ERR_HIP;
        NCT_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(uint32_t), s));
        PMLevel L;
        L.j0 = PMJob{a_hwc, b_hwc, (const uint2*)b_h16, nullptr, nullptr, nullptr, nullptr, ga, rs_max, seed_ab};
        L.j1 = PMJob{b_hwc, a_hwc, (const uint2*)a_h16, nullptr, nullptr, nullptr, nullptr, gb, rs_max, seed_ba};
        for (int i = 0; i < 2; ++i) { L.nn0[i] = na_buf[i]; L.dd0[i] = da_buf[i]; L.nn1[i] = nb_buf[i]; L.dd1[i] = db_buf[i]; }
        L.nblk0 = nblk0; L.nblk1 = nblk1; L.nsteps = 1 + 4 * iters; L.packed = 0; L.ctl = ctl; L.err = ctx->d_pm_err; L.counter = eval_counter;
        L.timeout_ticks = 25000000;                    // 0.25 s of the 100 MHz wall clock
        switch (C) {
            case 64:  return launch_level<1>(ctx, s, L, pm_mode);
            case 128: return launch_level<2>(ctx, s, L, pm_mode);
            case 256: return launch_level<4>(ctx, s, L, pm_mode);
            default:  return launch_level<8>(ctx, s, L, pm_mode);
        }
    }
    // the total number of Jacobi steps is even (iters*4), so ping-ponging (nnf,dist) <-> (tmp) ends in (nnf,dist)
    int rc = step(0, 0, 0, 0, 0, 0, 0);            // init: dist(current NNF), NNF untouched
    if (rc) return rc;
    int cur = 0, t = 0;
    for (int iter = 0; iter < iters; ++iter)
        for (int jump = 8; jump > 0; jump >>= 1) {
            ++t;
            rc = step(cur, cur ^ 1, 1, jump, iter, stamps ? t : 0, (!stamps || t == 4 * iters) ? 1 : 0);
            if (rc) return rc;
            cur ^= 1;
        }
    // cur == 0 here. The tmp buffers return to the arena now; that is safe because arena blocks are recycled in
    // stream order (every user enqueues on the same stream or joins into it before releasing).
    return 0;
}

int nctk_patchmatch(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, int C, int ah, int aw, int bh, int bw,
                    int iters, int rs_max, uint32_t seed, uint32_t* nnf, float* dist, unsigned long long* eval_counter) {
    return pm_run(ctx, s, a_hwc, b_hwc, nullptr, nullptr, C, ah, aw, bh, bw, iters, rs_max, seed, 0u, nnf, dist, nullptr, nullptr, eval_counter, NCT_PM_PLAIN);
}

int nctk_patchmatch_bidir(nct_ctx* ctx, hipStream_t s, const float* a_hwc, const float* b_hwc, const void* a_h16, const void* b_h16, int C, int ah, int aw, int bh, int bw,
                          int iters, int rs_max, uint32_t seed_ab, uint32_t seed_ba, uint32_t* ann, float* annd, uint32_t* bnn, float* bnnd, int pm_mode,
                          unsigned long long* counters) {
    return pm_run(ctx, s, a_hwc, b_hwc, a_h16, b_h16, C, ah, aw, bh, bw, iters, rs_max, seed_ab, seed_ba, ann, annd, bnn, bnnd, counters, pm_mode);
}
