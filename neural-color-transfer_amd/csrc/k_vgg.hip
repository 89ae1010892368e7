// k_vgg.hip — VGG19 feature extractor kernels (V2/R1): preprocess, 3x3 conv + bias + ReLU on f32 MFMA, 2x2 max-pool.
// Reference semantics: Caffe conv (im2col GEMM, K index = ci*9 + ky*3 + kx: code/src/caffe/util/im2col.cpp:19-56,
// layers/base_conv_layer.cpp:257-283), in-place ReLU (layers/relu_layer.cpp:14-17), MAX pooling with ceil-mode
// output size and clipped windows (layers/pooling_layer.cpp:90-93,147-165); preprocessing Classifier.cpp:211-275.
//
// MI355X design — im2col-free implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32, = a k-ordered fmaf chain, so the
// result is bit-identical to oracle/orc_vgg.c):
//   D[cout][pixel] += W[cout][k] * In[k][pixel],  k = ci*9 + tap ascending, two k per MFMA.
//   * A operand = weights pre-packed K-major ([k][cout]): lane (i = l&31, half = l>>5) reads Wp[k0+half][m0+i] — a
//     coalesced 128-B row segment per half-wave;
//   * B operand = activations in planar CHW: lane (j = l&31, half) reads In[ci][y+dy][x0+j+dx] — 32 consecutive
//     pixels of one plane, again a coalesced 128-B segment. No im2col buffer, no LDS round trip: an f32 MFMA needs only
//     512 B of operands per 4096 FLOP, each wave register-tiles 2x2 MFMA tiles (64 cout x 64 pixels, 64 accumulators)
//     and the 9 taps of a channel re-hit the same L1 lines, so the L1/L2 path feeds the matrix pipe without barriers;
//   * the D fragment has pixels on lanes and couts on registers, so every store instruction writes 32 consecutive
//     pixels of one output plane (coalesced), with bias + ReLU fused.
// Roofline: MFMA (f32 peak 157.3 TF). FLOPs = 2*9*Cin*Cout*H*W per layer.
#include "nct_internal.h"
#include "nct_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- preprocess: u8 BGR HWC -> float planar [4][H][W] (plane 3 = 0)
__global__ void k_vgg_preprocess(const uint8_t* __restrict__ bgr, int stride, float* __restrict__ out, int H, int W) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    int y = i / W, x = i - y * W;
    const uint8_t* p = bgr + (size_t)y * stride + x * 3;
    out[i] = (float)p[0] - 103.939f;
    out[(size_t)H * W + i] = (float)p[1] - 116.779f;
    out[(size_t)2 * H * W + i] = (float)p[2] - 123.68f;
    out[(size_t)3 * H * W + i] = 0.f;
}

// ---------------------------------------------------------------- conv3x3 pad1 stride1 + bias + ReLU
// Pixels are tiled in 1-D (row-major pixel index p = y*W + x): a 32-pixel MFMA column tile is 32 consecutive pixels (it may wrap
// over a row end — validity is per lane anyway), a wave owns two consecutive ones plus two 32-cout row tiles, a workgroup =
// 4 waves = WCO along cout x (4/WCO) along pixels. 1-D tiling wastes no lanes on ragged 2-D tile edges: conv4_x at 88x88 needs
// 61 x 4 = 244 workgroups (one per CU, one round) where 16x8 tiles needed 264 — 8 CUs with two workgroups doubled the layer time.
struct ConvGeom { int Cin, Cout, H, W, npx_blocks, nblk_n; };
#ifndef NCT_CONV_PT1_BELOW
#define NCT_CONV_PT1_BELOW 128
#endif

// WCO = waves along cout (1 => block covers 64 cout x 256 px; 2 => 128 cout x 128 px); PT = 32-pixel tiles per wave (2, or 1 for the
// layers whose grid would otherwise leave SIMDs with a single wave: half the pixels per workgroup, twice the workgroups)
template <int WCO, int PT>
__global__ __launch_bounds__(256) void k_conv3x3_mfma(const float* __restrict__ in, const float* __restrict__ wp /*[Cin*9][Cout]*/,
                                                      const float* __restrict__ bias, float* __restrict__ out, ConvGeom g, int relu) {
    constexpr int WPX = 4 / WCO;             // waves along pixels
    constexpr int BLK_PX = WPX * 32 * PT;    // pixels covered by a workgroup
    const int HW = g.H * g.W;
    // block -> (pixel block, cout block); blocks of one pixel block differ by multiples of 8 => same XCD/L2
    int bid = blockIdx.x;
    const int np = g.npx_blocks;
    int pt, nb;
    {
        const int grp = bid / (8 * g.nblk_n), rem = bid - grp * 8 * g.nblk_n;
        nb = rem >> 3; pt = grp * 8 + (rem & 7);
    }
    if (pt >= np) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int m0 = nb * (64 * WCO) + wco * 64;                 // first cout of this wave
    const int p0 = pt * BLK_PX + wpx * 32 * PT + l31, p1 = p0 + 32; // this lane's pixel in tile 0 / tile 1
    const bool live0 = p0 < HW, live1 = PT == 2 && p1 < HW;
    const int pc0 = live0 ? p0 : HW - 1, pc1 = live1 ? p1 : HW - 1;
    const int y0 = pc0 / g.W, x0 = pc0 - y0 * g.W, y1 = pc1 / g.W, x1 = pc1 - y1 * g.W;

    // per-lane tap tables for the 9 k-steps of a channel pair (k = 2s + half within 18)
    // out-of-image taps read the (in-bounds) own pixel instead and are zeroed by a select: no divergent branches in the K loop
    // (one offset table for both pixel tiles: an out-of-image tap is redirected to the lane's own pixel of that channel by a select on the
    //  ADDRESS, nine registers fewer than a table per tile — the difference between two and three waves per SIMD)
    int boff[9]; unsigned vm0 = 0, vm1 = 0;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int k = 2 * s + half;
        const int ci = k / 9, tap = k - 9 * ci, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        boff[s] = ci * HW + dy * g.W + dx;
        const bool v0 = live0 && (x0 + dx) >= 0 && (x0 + dx) < g.W && (y0 + dy) >= 0 && (y0 + dy) < g.H;
        const bool v1 = live1 && (x1 + dx) >= 0 && (x1 + dx) < g.W && (y1 + dy) >= 0 && (y1 + dy) < g.H;
        vm0 |= (unsigned)v0 << s; vm1 |= (unsigned)v1 << s;
    }
    auto own = [&](int s) { return s < 4 ? 0 : (s > 4 ? HW : half * HW); };      // ci * HW of k-step s (k = 2 s + half, ci = k / 9)
    const float* b0p = in + pc0;
    const float* b1p = in + pc1;
    const float4* ap4 = reinterpret_cast<const float4*>(wp) + (size_t)half * g.Cout + m0 + l31;       // s 0..3 of the pair; + 2 Cout float4: s 4..7
    const float* ap1 = wp + (size_t)16 * g.Cout + (size_t)half * g.Cout + m0 + l31;                  // s 8

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};   // [cout tile][pixel tile]
    // Software pipeline in registers: the 36 operands of channel pair c2+2 are requested before the 36 MFMAs of pair c2 issue
    // (36 x 64 = 2304 cycles of matrix work cover the L1/L2 latency of the next pair even at one wave per SIMD).
    // Two register sets in ping-pong, no copies: while the 36 MFMAs of one channel pair issue (36 x 64 = 2304 cycles of matrix work),
    // the 36 operand loads of the next pair are in flight. load_pair only ISSUES loads (raw values); the out-of-image select is
    // applied right before each MFMA group, so the only s_waitcnt in front of an MFMA block is for loads issued a whole block
    // earlier. (Selects or register copies directly behind the loads put the full L2 latency in front of every block: the conv
    // layers ran at 55-60 % MFMA utilisation that way.)
    struct ASet { float4 q0, q1; float s8; };               // the nine A values of a pair: s 0..3, s 4..7, s 8
    ASet a0x, a1x, a0y, a1y; float b0x[9], b1x[9], b0y[9], b1y[9];
    auto load_pair = [&](ASet& a0, ASet& a1, float (&b0)[9], float (&b1)[9]) {
        a0.q0 = ap4[0]; a1.q0 = ap4[32];
        a0.q1 = ap4[(size_t)2 * g.Cout]; a1.q1 = ap4[(size_t)2 * g.Cout + 32];
        a0.s8 = ap1[0]; a1.s8 = ap1[32];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            b0[s] = b0p[((vm0 >> s) & 1u) ? boff[s] : own(s)];
            if constexpr (PT == 2) b1[s] = b1p[((vm1 >> s) & 1u) ? boff[s] : own(s)];
        }
        ap4 += (size_t)18 * g.Cout / 4;
        ap1 += (size_t)18 * g.Cout;
        b0p += (size_t)2 * HW;
        b1p += (size_t)2 * HW;
    };
    auto aval = [](const ASet& a, int s) -> float {
        switch (s) { case 0: return a.q0.x; case 1: return a.q0.y; case 2: return a.q0.z; case 3: return a.q0.w;
                     case 4: return a.q1.x; case 5: return a.q1.y; case 6: return a.q1.z; case 7: return a.q1.w; default: return a.s8; }
    };
    auto mma_pair = [&](const ASet& a0, const ASet& a1, const float (&b0)[9], const float (&b1)[9]) {
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const float x0 = ((vm0 >> s) & 1u) ? b0[s] : 0.f;
            const float av0 = aval(a0, s), av1 = aval(a1, s);
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, x0, acc00, 0, 0, 0);
            if constexpr (PT == 2) {
                const float x1 = ((vm1 >> s) & 1u) ? b1[s] : 0.f;
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, x1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, x0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, x1, acc11, 0, 0, 0);
            } else {
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, x0, acc10, 0, 0, 0);
            }
        }
    };
    // schedule of one half iteration: MFMA, load, MFMA, load, ... — the 36 loads ride in the shadow of the 36 MFMAs instead of
    // draining the matrix pipe while they issue in one burst
    auto interleave = [] {
        if constexpr (PT == 2) {                                       // 36 MFMAs, 24 loads (6 A + 18 B)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one VMEM read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        } else {                                                       // 18 MFMAs, 15 loads (6 A + 9 B)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    };
    load_pair(a0x, a1x, b0x, b1x);
    int c2 = 0;
    for (; c2 + 4 <= g.Cin; c2 += 4) {
        load_pair(a0y, a1y, b0y, b1y);                            // channels c2+2, c2+3
        mma_pair(a0x, a1x, b0x, b1x);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // channels c2+4, c2+5 — unconditionally: a branch around these loads makes the waitcnt pass assume they were NOT issued and
        // drain everything inside the next MFMA block. Past the last pair the pointers are rewound and the (unused) loads re-read it.
        if (c2 + 4 >= g.Cin) { ap4 -= (size_t)18 * g.Cout / 4; ap1 -= (size_t)18 * g.Cout; b0p -= (size_t)2 * HW; b1p -= (size_t)2 * HW; }
        load_pair(a0x, a1x, b0x, b1x);
        mma_pair(a0y, a1y, b0y, b1y);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c2 < g.Cin) mma_pair(a0x, a1x, b0x, b1x);                 // odd number of channel pairs

    // epilogue: D row (cout) = (r&3) + 8*(r>>2) + 4*half, D col (pixel) = l31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int co0 = m0 + row, co1 = m0 + 32 + row;
        const float bb0 = bias[co0], bb1 = bias[co1];
        float v00 = acc00[r] + bb0, v01 = acc01[r] + bb0, v10 = acc10[r] + bb1, v11 = acc11[r] + bb1;
        if (relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
        if (live0) { out[(size_t)co0 * HW + p0] = v00; out[(size_t)co1 * HW + p0] = v10; }
        if constexpr (PT == 2) { if (live1) { out[(size_t)co0 * HW + p1] = v01; out[(size_t)co1 * HW + p1] = v11; } }
    }
}

int nctk_conv3x3(nct_ctx* ctx, hipStream_t s, const float* in, const float* wp, const float* bias, float* out,
                 int Cin, int Cout, int H, int W, int relu) {
    NCT_REQUIRE((Cin & 1) == 0 && (Cout & 63) == 0, "conv3x3: Cin=%d must be even (pad) and Cout=%d a multiple of 64", Cin, Cout);
    const int WCO = (Cout % 128 == 0) ? 2 : 1;
    // two pixel tiles per wave (best operand reuse) unless that grid has fewer than 512 workgroups (two per CU): then one tile per wave
    const int full_blocks = cdiv(H * W, (4 / WCO) * 64) * (Cout / (64 * WCO));
    const int PT = full_blocks >= NCT_CONV_PT1_BELOW ? 2 : 1;
    const int blk_px = (4 / WCO) * 32 * PT;
    ConvGeom g{Cin, Cout, H, W, cdiv(H * W, blk_px), Cout / (64 * WCO)};
    const int nblocks = cdiv(g.npx_blocks, 8) * 8 * g.nblk_n;
    if (WCO == 2 && PT == 2)      hipLaunchKernelGGL((k_conv3x3_mfma<2, 2>), dim3(nblocks), dim3(256), 0, s, in, wp, bias, out, g, relu);
    else if (WCO == 2)            hipLaunchKernelGGL((k_conv3x3_mfma<2, 1>), dim3(nblocks), dim3(256), 0, s, in, wp, bias, out, g, relu);
    else if (PT == 2)             hipLaunchKernelGGL((k_conv3x3_mfma<1, 2>), dim3(nblocks), dim3(256), 0, s, in, wp, bias, out, g, relu);
    else                          hipLaunchKernelGGL((k_conv3x3_mfma<1, 1>), dim3(nblocks), dim3(256), 0, s, in, wp, bias, out, g, relu);
    NCT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- 2x2/2 MAX pool, ceil mode, clipped window (CHW)
__global__ void k_maxpool2x2(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int Ho, int Wo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * Ho * Wo) return;
    const int px = (int)(i % Wo); const size_t t = i / Wo; const int py = (int)(t % Ho); const int c = (int)(t / Ho);
    const int hs = py * 2, ws = px * 2, he = min(hs + 2, H), we = min(ws + 2, W);
    const float* p = in + (size_t)c * H * W;
    float m = -3.402823466e+38f;
    for (int y = hs; y < he; ++y)
        for (int x = ws; x < we; ++x) { const float v = p[(size_t)y * W + x]; m = v > m ? v : m; }
    out[i] = m;
}

int nctk_maxpool2x2(nct_ctx* ctx, hipStream_t s, const float* in, float* out, int C, int H, int W) {
    const int Ho = (H - 2 + 1) / 2 + 1, Wo = (W - 2 + 1) / 2 + 1;      // ceil((n-2)/2)+1
    const size_t n = (size_t)C * Ho * Wo;
    hipLaunchKernelGGL(k_maxpool2x2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, C, H, W, Ho, Wo);
    NCT_LAUNCH_CHECK();
    return 0;
}

int nctk_vgg_preprocess(nct_ctx* ctx, hipStream_t s, const uint8_t* bgr, int stride, float* out, int H, int W) {
    hipLaunchKernelGGL(k_vgg_preprocess, dim3(cdiv(H * W, 256)), dim3(256), 0, s, bgr, stride, out, H, W);
    NCT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- weight packing: Caffe [Cout][Cin][3][3] -> [Cin_pad*9][Cout]
// Layout per channel pair P (18 k values, k18 = 2 s + h: s = k-step of the pair, h = lane half of the MFMA A operand; ci = 2 P + k18 / 9,
// tap = k18 % 9), 18 * Cout floats: [s 0..3: [h][cout][4]] [s 4..7: [h][cout][4]] [s 8: [h][cout]] — a lane fetches its nine A values of a
// pair with two 16-byte loads and one 4-byte load (coalesced over the 32 couts of its half) instead of nine 4-byte loads: the operand
// stream of the implicit GEMM is bound by vector-memory INSTRUCTIONS on the L1 address path (~12 cycles per 4-byte-per-lane load, 16 per
// 16-byte one), not by bytes.
__global__ void k_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int Cin_pad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cin_pad * 9 * Cout) return;
    const int co = (int)(i % Cout); const int k = (int)(i / Cout);
    const int ci = k / 9, tap = k - ci * 9;
    const float v = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
    const int P = ci >> 1, k18 = (ci & 1) * 9 + tap, s_ = k18 >> 1, h = k18 & 1;
    const size_t base = (size_t)P * 18 * Cout;
    const size_t dst = s_ < 8 ? base + (size_t)(s_ >> 2) * 8 * Cout + ((size_t)h * Cout + co) * 4 + (s_ & 3)
                              : base + (size_t)16 * Cout + (size_t)h * Cout + co;
    wp[dst] = v;
}

int nctk_pack_weights(nct_ctx* ctx, hipStream_t s, const float* w, float* wp, int Cout, int Cin, int Cin_pad) {
    const size_t n = (size_t)Cin_pad * 9 * Cout;
    hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, Cout, Cin, Cin_pad);
    NCT_LAUNCH_CHECK();
    return 0;
}
