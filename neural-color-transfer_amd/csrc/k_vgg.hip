// k_vgg.hip — VGG19 feature extractor kernels (V2/R1): preprocess, 3x3 conv + bias + ReLU on f32 MFMA, 2x2 max-pool.
// Reference semantics: Caffe conv (im2col GEMM, K index = ci*9 + ky*3 + kx: code/src/caffe/util/im2col.cpp:19-56,
// layers/base_conv_layer.cpp:257-283), in-place ReLU (layers/relu_layer.cpp:14-17), MAX pooling with ceil-mode
// output size and clipped windows (layers/pooling_layer.cpp:90-93,147-165); preprocessing Classifier.cpp:211-275.
//
// MI355X design — im2col-free implicit GEMM on the two-block f32 MFMA v_mfma_f32_32x32x1_2b_f32 (exact f32: one fmaf per output element and k, k ascending,
// so the result is bit-identical to oracle/orc_vgg.c):
//   D[cout][pixel] += W[cout][k] * In[k][pixel],  k = ci*9 + tap ascending, one k per MFMA, two 32-pixel blocks per MFMA.
//   * B operand = activations in planar CHW: lane = pixel (64 consecutive pixels per wave), the three dx taps of an input row come from ONE 12-byte
//     buffer load (x-1, x, x+1): six vector-memory instructions per channel pair and wave for the activations;
//   * A operand = weights pre-packed per channel pair so that lane (i = l&31, h = l>>5) fetches its nine values k = 2 s + h with two 16-byte loads and one
//     4-byte load, coalesced over the 32 couts of a half wave; cbsz:1 abid:h hands half h's value to both pixel blocks;
//   * no im2col buffer and no LDS round trip: an f32 MFMA needs 512 B of operands per 4096 FLOP, every wave register-tiles 64 couts x 64 pixels
//     (64 accumulators) and runs a two-stage register pipeline (operands of the next channel pair in flight under the 36 MFMAs of this one);
//   * the D fragment has pixels on lanes and couts on registers, so every store instruction writes 32 consecutive pixels of one output plane,
//     with bias + ReLU (+ the following 2x2 max-pool where the tile shape fits the map) fused.
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
// Pixels are tiled in 1-D (row-major pixel index p = y*W + x): a wave owns 64 consecutive pixels (the run may wrap over a row end — validity is per lane anyway)
// and 64 couts, a workgroup = 4 waves = WCO along cout x (4/WCO) along pixels. 1-D tiling wastes no lanes on ragged 2-D tile edges: conv4_x at 88x88 needs
// 61 x 4 = 244 workgroups (one per CU, one round) where 16x8 tiles needed 264 — 8 CUs with two workgroups doubled the layer time.
struct ConvGeom { int Cin, Cout, H, W, npx_blocks, nblk_n, tiles_x, Ho, Wo; };   // tiles_x, Ho, Wo: fused 2x2 pooling only
#ifndef NCT_CONV_PT1_BELOW
#define NCT_CONV_PT1_BELOW 128
#endif

// The kernel. Rounds 1-2 used v_mfma_f32_32x32x2_f32 (lane halves = two k values): every tap of every pixel tile was its own 4-byte load, 24 vector-memory
// instructions per 36 MFMAs, MFMA busy 0.61-0.68 (git history: k_conv3x3_mfma). Round 3: v_mfma_f32_32x32x1_2b_f32 multiplies two independent 32x32 blocks
// by ONE k each (still an exact fmaf per element, k ascending: bit-identical). Block = lane half, so a wave's 64 lanes are 64 consecutive PIXELS (block 0 = pixels 0..31, block 1 = 32..63) and every lane needs ALL 18
// k values of a channel pair for its own pixel: the three dx taps of an input row are the three dwords of ONE unaligned 12-byte load (x-1, x, x+1), six
// loads per channel pair instead of eighteen. The A operand (32 couts x one k) is the same for both blocks: the weights stay packed as for the 32x32x2
// kernel — lane half h holds k = 2 s + h — and cbsz:1 abid:h broadcasts half h's value to both blocks. 36 MFMAs per channel pair as before, 12 vector-memory
// instructions instead of 24, and no per-load address selects: the loads go through a buffer descriptor over the whole input, so the one dword before the
// first plane / behind the last one that edge lanes touch is range-checked away by the hardware (per dword: scripts/probes/bufload_probe.hip) and rows
// outside the image read the neighbouring plane or nothing — all of those taps are zeroed by the tap masks (selects, never multiplies).
// voffset is unsigned, so offsets carry a bias of one row + one pixel (soffset = plane offset - bias); plane 0 of the first pair cannot be biased and is
// fetched in the prologue with the window of the x == 0 lanes moved one pixel right and rotated back.
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// WCO = waves along cout; CT = 32-cout tiles per wave (2: 64 couts x 64 px per wave; 1: 32 x 64, for grids that would otherwise be too small).
// POOL = the following 2x2/2 max-pool (ceil mode, clipped windows: k_maxpool2x2) in the epilogue: the wave's 64 pixels are then a 32 x 2 tile (block 0 = row 2 ty,
// block 1 = row 2 ty + 1), so a pooling window is two registers of one lane (vertical) and two neighbouring lanes (horizontal, one DPP quad_perm); only the pooled
// map is written. The K loop does not change: every lane fetches the rows of its own pixel.
template <int WCO, int CT, bool POOL>
__device__ __forceinline__ void conv3x3_mfma2b_body(const float* __restrict__ in, const float* __restrict__ wp /*packed, see k_pack_weights*/,
                                                    const float* __restrict__ bias, float* __restrict__ out, const ConvGeom& g, int relu, float* __restrict__ out_hwc, int bid) {
    constexpr int WPX = 4 / WCO;             // waves along pixels
    constexpr int BLK_PX = WPX * 64;
    constexpr int BLK_CO = WCO * 32 * CT;
    const int HW = g.H * g.W;
    int pt, nb;
    {
        const int grp = bid / (8 * g.nblk_n), rem = bid - grp * 8 * g.nblk_n;
        nb = rem >> 3; pt = grp * 8 + (rem & 7);
    }
    if (pt >= g.npx_blocks) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int m0 = nb * BLK_CO + wco * 32 * CT;                 // first cout of this wave
    int y, x, tile_y = 0, tile_x = 0; bool live;                // this lane's pixel
    if constexpr (POOL) {
        const int wt = pt * WPX + wpx;                          // wave tile, row-major over (row pairs, 32-pixel column tiles)
        tile_y = wt / g.tiles_x; tile_x = wt - tile_y * g.tiles_x;
        y = 2 * tile_y + half; x = 32 * tile_x + l31;
        live = y < g.H && x < g.W;
    } else {
        const int p = pt * BLK_PX + wpx * 64 + lane;
        live = p < HW;
        const int pc = live ? p : 0;
        y = pc / g.W; x = pc - y * g.W;
    }
    const int bias_px = g.H >= 2 ? g.W + 1 : 1;              // one row + one pixel; a one-row map (the tail of a tiny pyramid) has no room for the row: its dy taps are masked anyway
    const unsigned plane_bytes = (unsigned)HW * 4u, bias_bytes = (unsigned)bias_px * 4u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (unsigned)g.Cin * plane_bytes, 0x00020000);

    unsigned voff[3];                        // biased byte offset of (row y + r - 1, pixel x - 1) inside a plane; dead lanes: out of range
    bool rowv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        voff[r] = live ? (unsigned)(((y + r - 1) * g.W + x - 1 + bias_px) * 4) : 0xFFF00000u;       // >= 0 for every row inside the image
        rowv[r] = live && (y + r - 1) >= 0 && (y + r - 1) < g.H;
    }
    const bool xl = x > 0, xr = x < g.W - 1;
    auto tapv = [&](int r, int d) { return d == 0 ? (rowv[r] && xl) : (d == 2 ? (rowv[r] && xr) : rowv[r]); };
    bool tm[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tm[t] = tapv(t / 3, t % 3);

    const float4* ap4 = reinterpret_cast<const float4*>(wp) + (size_t)half * g.Cout + m0 + l31;
    const float* ap1 = wp + (size_t)16 * g.Cout + (size_t)half * g.Cout + m0 + l31;
    unsigned soff = plane_bytes - bias_bytes;              // plane 1 (the prologue fetches plane 0 unbiased)

    f32x32 acc0 = {0}, acc1 = {0};           // [cout tile]: registers 0..15 = pixel block 0, 16..31 = block 1
    struct ASet { float4 q0, q1; float s8; };
    struct BSet { u32x3 r[2][3]; };
    ASet a0x, a1x, a0y, a1y; BSet bx, by;
    // weights first, then the activation rows (requesting them in the order the MFMAs consume them, or activations first, measured 3-5 % slower: DESIGN.md 9)
    auto bload = [&](unsigned so, int r) { return __builtin_amdgcn_raw_buffer_load_b96(rs, voff[r], so, 0); };
#ifndef NCT_CONV_TIMING_SKIP
#define NCT_CONV_TIMING_SKIP 0                                   // timing experiments only (results wrong): 1 = K loop without the A stream, 2 = without the B stream, 3 = neither
#endif
    auto load_pair = [&](ASet& a0, ASet& a1, BSet& b) {          // channels (soff - plane + bias) / plane and the next one
        constexpr bool LA = !(NCT_CONV_TIMING_SKIP & 1), LB = !(NCT_CONV_TIMING_SKIP & 2);
        if constexpr (LA) {
            a0.q0 = ap4[0]; a0.q1 = ap4[(size_t)2 * g.Cout]; a0.s8 = ap1[0];
            if constexpr (CT == 2) { a1.q0 = ap4[32]; a1.q1 = ap4[(size_t)2 * g.Cout + 32]; a1.s8 = ap1[32]; }
        }
        if constexpr (LB) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { b.r[0][r] = bload(soff - plane_bytes, r); b.r[1][r] = bload(soff, r); }
        }
        ap4 += (size_t)18 * g.Cout / 4;
        ap1 += (size_t)18 * g.Cout;
        soff += 2u * plane_bytes;
    };
    auto aval = [](const ASet& a, int s) -> float {
        switch (s) { case 0: return a.q0.x; case 1: return a.q0.y; case 2: return a.q0.z; case 3: return a.q0.w;
                     case 4: return a.q1.x; case 5: return a.q1.y; case 6: return a.q1.z; case 7: return a.q1.w; default: return a.s8; }
    };
    auto mma_pair = [&](const ASet& a0, const ASet& a1, const BSet& b) {
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const int ci = k / 9, t = k - 9 * ci, r = t / 3, d = t - 3 * r, s = k >> 1;
            const float bv = tm[t] ? __uint_as_float(b.r[ci][r][d]) : 0.f;
            if (k & 1) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x1f32(aval(a0, s), bv, acc0, 1, 1, 0);
                if constexpr (CT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x1f32(aval(a1, s), bv, acc1, 1, 1, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x1f32(aval(a0, s), bv, acc0, 1, 0, 0);
                if constexpr (CT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x1f32(aval(a1, s), bv, acc1, 1, 0, 0);
            }
        }
    };
#ifndef NCT_CONV_SCHED
#define NCT_CONV_SCHED 0
#endif
    auto interleave = [] {
        if constexpr (NCT_CONV_SCHED == 2) return;
        if constexpr (CT == 2) {                                       // 36 MFMAs, 12 loads (6 A + 6 B)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, NCT_CONV_SCHED == 1 ? 1 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NCT_CONV_SCHED == 1 ? 2 : 1, 0);
            }
        } else {                                                       // 18 MFMAs, 9 loads (3 A + 6 B)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
    };
    // prologue: channel pair 0. Plane 0 has no room for the bias in front of it: unbiased offsets, x == 0 lanes fetch (x, x+1, x+2) and rotate.
    {
        a0x.q0 = ap4[0]; a0x.q1 = ap4[(size_t)2 * g.Cout]; a0x.s8 = ap1[0];
        if constexpr (CT == 2) { a1x.q0 = ap4[32]; a1x.q1 = ap4[(size_t)2 * g.Cout + 32]; a1x.s8 = ap1[32]; }
        ap4 += (size_t)18 * g.Cout / 4;
        ap1 += (size_t)18 * g.Cout;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned v0 = voff[r] - bias_bytes + (xl ? 0u : 4u);          // rows above the image wrap to out-of-range values: they read zeros and are masked
            u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, v0, 0, 0);
            if (!xl) { t.z = t.y; t.y = t.x; }
            bx.r[0][r] = t;
            bx.r[1][r] = __builtin_amdgcn_raw_buffer_load_b96(rs, voff[r], soff, 0);
        }
        soff += 2u * plane_bytes;
    }
    int c2 = 0;
    for (; c2 + 4 <= g.Cin; c2 += 4) {
        load_pair(a0y, a1y, by);                                  // channels c2+2, c2+3
        mma_pair(a0x, a1x, bx);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // channels c2+4, c2+5 — unconditionally (see k_conv3x3_mfma); past the last pair the pointers are rewound and the unused loads re-read it
        if (c2 + 4 >= g.Cin) { ap4 -= (size_t)18 * g.Cout / 4; ap1 -= (size_t)18 * g.Cout; soff -= 2u * plane_bytes; }
        load_pair(a0x, a1x, bx);
        mma_pair(a0y, a1y, by);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c2 < g.Cin) mma_pair(a0x, a1x, bx);                       // odd number of channel pairs

    // epilogue: register 16 b + r of a tile = (cout row (r&3) + 8 (r>>2) + 4 half, pixel 32 b + l31)
    if constexpr (POOL) {
        const int xx = 32 * tile_x + l31, y0 = 2 * tile_y;                   // column of this lane in both blocks (whatever its half), top row of the tile
        const bool ok0 = y0 < g.H && xx < g.W, ok1 = y0 + 1 < g.H && xx < g.W;
        const size_t po = (size_t)tile_y * g.Wo + (xx >> 1), pn = (size_t)g.Ho * g.Wo;
        const bool writer = ok0 && !(l31 & 1);
        auto pool4 = [&](float top, float bot) {                            // window order of the Caffe loop: (y0, x), (y0, x+1), (y0+1, x), (y0+1, x+1); clipped = skipped
            top = ok0 ? top : -3.402823466e+38f; bot = ok1 ? bot : -3.402823466e+38f;
            const float top_r = dpp_rot<0xB1>(top), bot_r = dpp_rot<0xB1>(bot);          // quad_perm:[1,0,3,2]: the odd neighbour's values in the even lane
            float m = -3.402823466e+38f;
            m = top > m ? top : m; m = top_r > m ? top_r : m; m = bot > m ? bot : m; m = bot_r > m ? bot_r : m;
            return m;
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co0 = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float bb0 = bias[co0];
            float t0 = acc0[r] + bb0, u0 = acc0[16 + r] + bb0;
            if (relu) { t0 = fmaxf(t0, 0.f); u0 = fmaxf(u0, 0.f); }
            const float m0v = pool4(t0, u0);
            if (writer) out[(size_t)co0 * pn + po] = m0v;
            if constexpr (CT == 2) {
                const float bb1 = bias[co0 + 32];
                float t1 = acc1[r] + bb1, u1 = acc1[16 + r] + bb1;
                if (relu) { t1 = fmaxf(t1, 0.f); u1 = fmaxf(u1, 0.f); }
                const float m1v = pool4(t1, u1);
                if (writer) out[(size_t)(co0 + 32) * pn + po] = m1v;
            }
        }
    } else {
        const int pw = pt * BLK_PX + wpx * 64;
        // tap layers (conv*_1) can write the map channel-last as well / instead: PatchMatch, votes and normalisation read HWC, and a lane holds four consecutive couts of its
        // pixel in registers 4 g .. 4 g + 3 (rows 8 g + 4 half + 0..3), i.e. one 16-byte store per group — the separate CHW -> HWC transpose pass of a tap (read + write) becomes
        // one more write here. Same values: bias add and ReLU are the same operations as in the planar stores below.
        if (out_hwc) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pp = pw + 32 * b + l31;
                if (pp < HW) {
                    float* dst = out_hwc + (size_t)pp * g.Cout + m0 + 4 * half;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float4 v0, v1;
                        float* e0 = &v0.x; float* e1 = &v1.x;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int row = j + 8 * gq + 4 * half;
                            float t0 = acc0[16 * b + 4 * gq + j] + bias[m0 + row];
                            if (relu) t0 = fmaxf(t0, 0.f);
                            e0[j] = t0;
                            if constexpr (CT == 2) { float t1 = acc1[16 * b + 4 * gq + j] + bias[m0 + 32 + row]; if (relu) t1 = fmaxf(t1, 0.f); e1[j] = t1; }
                        }
                        *reinterpret_cast<float4*>(dst + 8 * gq) = v0;
                        if constexpr (CT == 2) *reinterpret_cast<float4*>(dst + 32 + 8 * gq) = v1;
                    }
                }
            }
        }
        if (out)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int pp = pw + 32 * b + l31;
            const bool lv = pp < HW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int co0 = m0 + row;
                float v0 = acc0[16 * b + r] + bias[co0];
                if (relu) v0 = fmaxf(v0, 0.f);
                if (lv) out[(size_t)co0 * HW + pp] = v0;
                if constexpr (CT == 2) {
                    float v1 = acc1[16 * b + r] + bias[co0 + 32];
                    if (relu) v1 = fmaxf(v1, 0.f);
                    if (lv) out[(size_t)(co0 + 32) * HW + pp] = v1;
                }
            }
        }
    }
}

// Does the 32 x 2 tile of the fused-pool form fit a H x W map? (auto rule: at most 4 % more wave tiles than the 1-D tiling — 700^2, 350^2: +0.6 %; 175^2, 88^2: +9 %, not fused)
bool nctk_conv3x3_pool_fits(int H, int W) {
    const long t2 = (long)cdiv(W, 32) * cdiv(H, 2), t1 = cdiv(H * W, 64);
    return t2 * 100 <= t1 * 104;
}

// pool = 1: `out` receives only the 2x2/2 max-pooled map [Cout][(H-1)/2+1][(W-1)/2+1] (what nctk_maxpool2x2 would make of the conv output)
template <int WCO, int CT, bool POOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_conv3x3_mfma2b(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                        float* __restrict__ out, ConvGeom g, int relu, float* __restrict__ out_hwc) {
    conv3x3_mfma2b_body<WCO, CT, POOL>(in, wp, bias, out, g, relu, out_hwc, (int)blockIdx.x);
}
// Two images through the SAME layer in one launch (own geometry each, shared weights): the first nblocks1 workgroups take image 1, the rest image 2. For conv5_1 of
// the source and the reference (44 x 44 pixels at 700 x 700: 124 workgroups per image, half the chip idle for 190 us, twice per pair).
template <int WCO, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_conv3x3_mfma2b_pair(const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, float* __restrict__ out1, float* __restrict__ out2, ConvGeom g1, ConvGeom g2, int nblocks1, int relu,
                                                        float* __restrict__ hwc1, float* __restrict__ hwc2) {
    const bool second = (int)blockIdx.x >= nblocks1;
    conv3x3_mfma2b_body<WCO, CT, false>(second ? in2 : in1, wp, bias, second ? out2 : out1, second ? g2 : g1, relu, second ? hwc2 : hwc1, second ? (int)blockIdx.x - nblocks1 : (int)blockIdx.x);
}
// out_hwc (nullable, pool == 0 only): the same map channel-last [H*W][Cout]; `out` may then be null
int nctk_conv3x3(nct_ctx* ctx, hipStream_t s, const float* in, const float* wp, const float* bias, float* out,
                 int Cin, int Cout, int H, int W, int relu, int pool, float* out_hwc) {
    NCT_REQUIRE(out || out_hwc, "conv3x3: no output");
    NCT_REQUIRE(!(pool && out_hwc), "conv3x3: the channel-last output exists for un-pooled layers only");
    NCT_REQUIRE((Cin & 1) == 0 && (Cout & 63) == 0, "conv3x3: Cin=%d must be even (pad) and Cout=%d a multiple of 64", Cin, Cout);
    NCT_REQUIRE((size_t)Cin * H * W * 4 < ((size_t)1 << 32), "conv3x3: input of %d x %d x %d floats exceeds the 4 GB a buffer descriptor addresses", Cin, H, W);
    const int WCO = (Cout % 128 == 0) ? 2 : 1;
    const int tiles_x = cdiv(W, 32), ntiles = pool ? tiles_x * cdiv(H, 2) : cdiv(H * W, 64);         // 64-pixel wave tiles
    // 64 couts x 64 px per wave unless that grid has fewer than NCT_CONV_PT1_BELOW workgroups: then 32 couts per wave, four waves along cout
    const int full_blocks = cdiv(ntiles, 4 / WCO) * (Cout / (64 * WCO));
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
#define NCT_CONV_LAUNCH(wco, ct, wpx, nblk_n_)                                                                                                    \
    do {                                                                                                                                          \
        ConvGeom g{Cin, Cout, H, W, cdiv(ntiles, wpx), nblk_n_, tiles_x, Ho, Wo};                                                                  \
        const int nblocks = cdiv(g.npx_blocks, 8) * 8 * g.nblk_n;                                                                                 \
        if (pool) hipLaunchKernelGGL((k_conv3x3_mfma2b<wco, ct, true>), dim3(nblocks), dim3(256), 0, s, in, wp, bias, out, g, relu, (float*)nullptr);             \
        else      hipLaunchKernelGGL((k_conv3x3_mfma2b<wco, ct, false>), dim3(nblocks), dim3(256), 0, s, in, wp, bias, out, g, relu, out_hwc);            \
    } while (0)
    if (full_blocks >= NCT_CONV_PT1_BELOW) {
        if (WCO == 2) NCT_CONV_LAUNCH(2, 2, 2, Cout / 128);
        else          NCT_CONV_LAUNCH(1, 2, 4, Cout / 64);
    } else if (Cout % 128 == 0) NCT_CONV_LAUNCH(4, 1, 1, Cout / 128);
    else                        NCT_CONV_LAUNCH(2, 1, 2, Cout / 64);
#undef NCT_CONV_LAUNCH
    NCT_LAUNCH_CHECK();
    return 0;
}

// The same layer for two images in ONE launch where both would run on grids too small to fill the chip (k_conv3x3_mfma2b_pair); otherwise two launches. No pooling.
int nctk_conv3x3_pair(nct_ctx* ctx, hipStream_t s, const float* in1, int H1, int W1, const float* in2, int H2, int W2, const float* wp, const float* bias,
                      float* out1, float* out2, int Cin, int Cout, int relu, float* hwc1, float* hwc2) {
    auto small = [&](int H, int W) {
        const int WCO = (Cout % 128 == 0) ? 2 : 1;
        return cdiv(cdiv(H * W, 64), 4 / WCO) * (Cout / (64 * WCO)) < NCT_CONV_PT1_BELOW;
    };
    const bool ok = (Cin & 1) == 0 && (Cout & 63) == 0 && small(H1, W1) && small(H2, W2) && (size_t)Cin * H1 * W1 * 4 < ((size_t)1 << 32) && (size_t)Cin * H2 * W2 * 4 < ((size_t)1 << 32) &&
                    (out1 || hwc1) && (out2 || hwc2) && ctx->conv_pair != 0;
    if (!ok) {
        int rc = nctk_conv3x3(ctx, s, in1, wp, bias, out1, Cin, Cout, H1, W1, relu, 0, hwc1); if (rc) return rc;
        return nctk_conv3x3(ctx, s, in2, wp, bias, out2, Cin, Cout, H2, W2, relu, 0, hwc2);
    }
    const int nt1 = cdiv(H1 * W1, 64), nt2 = cdiv(H2 * W2, 64);
#define NCT_CONV_PAIR(wco, ct, wpx, nblk_n_)                                                                                                      \
    do {                                                                                                                                          \
        ConvGeom g1{Cin, Cout, H1, W1, cdiv(nt1, wpx), nblk_n_, 0, 0, 0}, g2{Cin, Cout, H2, W2, cdiv(nt2, wpx), nblk_n_, 0, 0, 0};                 \
        const int nb1 = cdiv(g1.npx_blocks, 8) * 8 * g1.nblk_n, nb2 = cdiv(g2.npx_blocks, 8) * 8 * g2.nblk_n;                                     \
        hipLaunchKernelGGL((k_conv3x3_mfma2b_pair<wco, ct>), dim3(nb1 + nb2), dim3(256), 0, s, in1, in2, wp, bias, out1, out2, g1, g2, nb1, relu, hwc1, hwc2); \
    } while (0)
    if (Cout % 128 == 0) NCT_CONV_PAIR(4, 1, 1, Cout / 128);
    else                 NCT_CONV_PAIR(2, 1, 2, Cout / 64);
#undef NCT_CONV_PAIR
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
