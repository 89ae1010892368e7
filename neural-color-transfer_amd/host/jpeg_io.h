// jpeg_io.h — baseline / extended-sequential / progressive (Huffman) JPEG decoder for the CLI (the reference reads its inputs with cv::imread, main.cu:483,491,
// which accepts JPEG; OpenCV / libjpeg are not available in this environment and the image is decoded on the host either side of the
// GPU path). Restates what cv::imread's libjpeg path computes so that the pixels handed to the GPU are the ones the
// reference would have seen: Huffman sequential DCT (SOF0 / SOF1, 8 bit), the accurate integer inverse DCT (libjpeg "islow",
// 13-bit constants, two passes), "fancy" triangle-filter chroma upsampling for 2x1 and 2x2 subsampling (pixel replication for other
// factors), JFIF YCbCr -> RGB with the 16-bit fixed-point tables, output as 8-bit 3-channel BGR (grayscale replicated). EXIF orientation
// is ignored, as OpenCV 2.4 does. Progressive files (SOF2: spectral selection + successive approximation, jdphuff.c) are decoded into the same
// coefficient arrays. Arithmetic-coded / lossless / 12-bit / CMYK files are rejected with a message.
// tests/test_cli.py checks the decoder bit-for-bit against Pillow (libjpeg-turbo, same algorithms) for 4:4:4, 4:2:2, 4:2:0, grayscale,
// restart intervals, odd sizes and progressive files.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "png_io.h"

namespace jpegio {

struct Huff { uint8_t bits[17]; uint8_t vals[256]; int mincode[18], maxcode[18], valptr[18]; int look_nbits[256]; uint8_t look_sym[256]; bool set = false; };

inline void huff_build(Huff& h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k; h.mincode[l] = code;
        code += h.bits[l]; k += h.bits[l];
        h.maxcode[l] = h.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    memset(h.look_nbits, 0, sizeof h.look_nbits);
    // 8-bit lookahead table
    int p = 0; code = 0;
    for (int l = 1; l <= 8; ++l) {
        for (int i = 0; i < h.bits[l]; ++i, ++p) {
            const int look = code << (8 - l);
            for (int c = 0; c < (1 << (8 - l)); ++c) { h.look_nbits[look + c] = l; h.look_sym[look + c] = h.vals[p]; }
            ++code;
        }
        code <<= 1;
    }
    h.set = true;
}

struct BitReader {
    const uint8_t* d; size_t n, pos; uint32_t acc = 0; int cnt = 0; bool hit_marker = false;
    BitReader(const uint8_t* data, size_t len, size_t start) : d(data), n(len), pos(start) {}
    void fill() {
        while (cnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    const int b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) pos += 2; else { hit_marker = true; b = 0; }     // a marker: feed zeros (libjpeg does the same)
                } else ++pos;
            }
            acc |= (uint32_t)b << (24 - cnt); cnt += 8;
        }
    }
    int peek(int nb) { if (cnt < nb) fill(); return (int)(acc >> (32 - nb)); }
    void skip(int nb) { acc <<= nb; cnt -= nb; }
    int get(int nb) { if (!nb) return 0; const int v = peek(nb); skip(nb); return v; }
    void reset() { acc = 0; cnt = 0; hit_marker = false; }
};

inline int huff_decode(BitReader& br, const Huff& h) {
    const int look = br.peek(8);
    if (h.look_nbits[look]) { br.skip(h.look_nbits[look]); return h.look_sym[look]; }
    int code = br.peek(16), l = 9;
    for (; l <= 16; ++l) { const int c = code >> (16 - l); if (c <= h.maxcode[l] && h.maxcode[l] >= 0 && c >= h.mincode[l]) { br.skip(l); return h.vals[h.valptr[l] + c - h.mincode[l]]; } }
    br.skip(16);
    return 0;                                              // corrupt data: libjpeg substitutes zero
}
inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// libjpeg jidctint.c (islow): CONST_BITS 13, PASS1_BITS 2
inline void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137,
                   F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    auto descale = [](long x, int n) -> long { return (x + (1L << (n - 1))) >> n; };
    int ws[64];
    for (int c = 0; c < 8; ++c) {
        const int16_t* in = coef + c; const uint16_t* qq = q + c;
        if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
            const int dc = (int)(((long)in[0] * qq[0]) * (1L << P1));
            for (int r = 0; r < 8; ++r) ws[r * 8 + c] = dc;
            continue;
        }
        long z2 = (long)in[16] * qq[16], z3 = (long)in[48] * qq[48];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        z2 = (long)in[0] * qq[0]; z3 = (long)in[32] * qq[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (long)in[56] * qq[56]; tmp1 = (long)in[40] * qq[40]; tmp2 = (long)in[24] * qq[24]; tmp3 = (long)in[8] * qq[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = (int)descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = (int)descale(tmp10 - tmp3, CB - P1);
        ws[1 * 8 + c] = (int)descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = (int)descale(tmp11 - tmp2, CB - P1);
        ws[2 * 8 + c] = (int)descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = (int)descale(tmp12 - tmp1, CB - P1);
        ws[3 * 8 + c] = (int)descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = (int)descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const int* w = ws + r * 8; uint8_t* o = out + (size_t)r * stride;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        long tmp0 = ((long)w[0] + w[4]) * (1L << CB), tmp1 = ((long)w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int SH = CB + P1 + 3;
        o[0] = clamp8((int)descale(tmp10 + tmp3, SH) + 128); o[7] = clamp8((int)descale(tmp10 - tmp3, SH) + 128);
        o[1] = clamp8((int)descale(tmp11 + tmp2, SH) + 128); o[6] = clamp8((int)descale(tmp11 - tmp2, SH) + 128);
        o[2] = clamp8((int)descale(tmp12 + tmp1, SH) + 128); o[5] = clamp8((int)descale(tmp12 - tmp1, SH) + 128);
        o[3] = clamp8((int)descale(tmp13 + tmp0, SH) + 128); o[4] = clamp8((int)descale(tmp13 - tmp0, SH) + 128);
    }
}

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int bw = 0, bh = 0;   // blocks per row / column (padded to whole MCUs)
              int dw = 0, dh = 0;                                                     // downsampled_width / height (real samples)
              std::vector<int16_t> coef; std::vector<uint8_t> plane; int pred = 0; };

inline bool read(const std::string& path, ImageBGR& img, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open"; return false; }
    std::vector<uint8_t> d;
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) { err = "not a JPEG file"; return false; }
    uint16_t qt[4][64]; bool qset[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    std::vector<Comp> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, mcux = 0, mcuy = 0;
    bool adobe = false; int adobe_transform = -1;
    bool progressive = false;
    size_t pos = 2;
    bool have_frame = false, any_scan = false;
    int nscans = 0;
    while (pos + 4 <= d.size()) {
        if (d[pos] != 0xFF) { ++pos; continue; }
        const int m = d[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { pos += 2; continue; }
        if (m == 0xD9) break;
        const size_t len = ((size_t)d[pos + 2] << 8) | d[pos + 3];
        if (len < 2 || pos + 2 + len > d.size()) { err = "truncated JPEG segment"; return false; }
        const uint8_t* p = &d[pos + 4]; const size_t L = len - 2;
        if (m == 0xDB) {                                    // DQT
            size_t i = 0;
            while (i < L) {
                const int pq = p[i] >> 4, tq = p[i] & 15; ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > L) { err = "bad DQT"; return false; }
                for (int k = 0; k < 64; ++k) { qt[tq][kZigzag[k]] = pq ? (uint16_t)((p[i] << 8) | p[i + 1]) : p[i]; i += pq ? 2 : 1; }
                qset[tq] = true;
            }
        } else if (m == 0xC4) {                             // DHT
            size_t i = 0;
            while (i + 17 <= L) {
                const int tc = p[i] >> 4, th = p[i] & 15; ++i;
                if (th > 3 || tc > 1) { err = "bad DHT"; return false; }
                Huff& h = tc ? hac[th] : hdc[th];
                int cnt = 0; h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = p[i + l - 1]; cnt += h.bits[l]; }
                i += 16;
                if (cnt > 256 || i + cnt > L) { err = "bad DHT"; return false; }
                memcpy(h.vals, p + i, cnt); i += cnt;
                if (!tc) for (int k = 0; k < cnt; ++k) if (h.vals[k] > 15) { err = "bad DHT (DC category > 15)"; return false; }   // jdhuff.c: JERR_BAD_HUFF_TABLE
                { int code = 0; for (int l = 1; l <= 16; ++l) { code += h.bits[l]; if (code > (1 << l)) { err = "bad DHT (oversubscribed code lengths)"; return false; } code <<= 1; } }
                huff_build(h);
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {   // SOF0 / SOF1 (sequential), SOF2 (progressive)
            progressive = m == 0xC2;
            if (have_frame) { err = "multiple frames"; return false; }
            if (L < 6 || p[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
            H = (p[1] << 8) | p[2]; W = (p[3] << 8) | p[4];
            const int nc = p[5];
            if (W <= 0 || H <= 0 || (size_t)W * H > (64u << 20)) { err = "bad JPEG dimensions"; return false; }
            if ((nc != 1 && nc != 3) || L < 6 + 3 * (size_t)nc) { err = "unsupported number of components (grayscale and YCbCr only)"; return false; }
            comps.resize(nc);
            for (int c = 0; c < nc; ++c) {
                comps[c].id = p[6 + 3 * c]; comps[c].h = p[7 + 3 * c] >> 4; comps[c].v = p[7 + 3 * c] & 15; comps[c].tq = p[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) { err = "bad SOF"; return false; }
                if (comps[c].h > hmax) hmax = comps[c].h;
                if (comps[c].v > vmax) vmax = comps[c].v;
            }
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }     // a single component is never subsampled
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                c.bw = mcux * c.h; c.bh = mcuy * c.v;
                c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
            have_frame = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            err = "unsupported JPEG coding process (lossless / hierarchical / arithmetic)";
            return false;
        } else if (m == 0xDD) {
            if (L >= 2) restart = (p[0] << 8) | p[1];
        } else if (m == 0xEE) {
            if (L >= 12 && !memcmp(p, "Adobe", 5)) { adobe = true; adobe_transform = p[11]; }
        } else if (m == 0xDA) {                             // SOS + entropy-coded data
            if (!have_frame) { err = "SOS before SOF"; return false; }
            // every scan walks all MCUs of the frame (up to 64 MP): a crafted file with thousands of tiny SOS segments would cost O(scans x blocks).
            // libjpeg-turbo's max_scans guard; real progressive files have ~10 scans
            if (++nscans > 256) { err = "too many scans (limit: 256)"; return false; }
            const int ns = p[0];
            if (ns < 1 || ns > (int)comps.size() || L < 1 + 2 * (size_t)ns + 3) { err = "bad SOS"; return false; }
            std::vector<Comp*> sc;
            for (int i = 0; i < ns; ++i) {
                Comp* c = nullptr;
                for (auto& k : comps) if (k.id == p[1 + 2 * i]) c = &k;
                if (!c) { err = "bad SOS component"; return false; }
                for (Comp* prev : sc) if (prev == c) { err = "duplicate component in SOS"; return false; }
                c->td = p[2 + 2 * i] >> 4; c->ta = p[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || (!progressive && (!hdc[c->td].set || !hac[c->ta].set))) { err = "missing Huffman table"; return false; }
                c->pred = 0;
                sc.push_back(c);
            }
            const int Ss = p[1 + 2 * ns], Se = p[2 + 2 * ns], Ah = p[3 + 2 * ns] >> 4, Al = p[3 + 2 * ns] & 15;
            BitReader br(d.data(), d.size(), pos + 2 + len);
            if (progressive) {
                // jdphuff.c: DC scans (Ss = Se = 0, possibly interleaved) and single-component AC scans over the band Ss..Se, each either a first pass
                // (Ah = 0: values shifted left by Al) or a refinement pass (Ah > 0: one more bit of every coefficient of the band)
                if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) { err = "bad progressive scan parameters"; return false; }
                for (auto* c : sc) if ((Ss == 0 && Ah == 0 && !hdc[c->td].set) || (Ss > 0 && !hac[c->ta].set)) { err = "missing Huffman table"; return false; }
                const bool inter = ns > 1;
                const int nmx = inter ? mcux : (sc[0]->dw + 7) / 8, nmy = inter ? mcuy : (sc[0]->dh + 7) / 8;
                int rst_left = restart; unsigned eobrun = 0;
                const int p1 = 1 << Al, m1 = -(1 << Al);
                for (int my = 0; my < nmy; ++my)
                    for (int mx = 0; mx < nmx; ++mx) {
                        if (restart && rst_left == 0) {
                            br.reset();
                            size_t q = br.pos;
                            while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7)) ++q;
                            if (q + 1 < d.size()) q += 2;
                            br.pos = q; br.reset();
                            rst_left = restart; eobrun = 0;
                            for (auto* c : sc) c->pred = 0;
                        }
                        for (auto* c : sc) {
                            const int bxn = inter ? c->h : 1, byn = inter ? c->v : 1;
                            for (int by = 0; by < byn; ++by)
                                for (int bx = 0; bx < bxn; ++bx) {
                                    int16_t* blk = &c->coef[((size_t)(my * byn + by) * c->bw + mx * bxn + bx) * 64];
                                    if (Ss == 0) {
                                        if (Ah == 0) { const int s = huff_decode(br, hdc[c->td]); c->pred += extend(br.get(s), s); blk[0] = (int16_t)(c->pred * (1 << Al)); }
                                        else if (br.get(1)) blk[0] |= (int16_t)p1;
                                    } else if (Ah == 0) {                  // decode_mcu_AC_first
                                        if (eobrun > 0) { --eobrun; continue; }
                                        for (int k = Ss; k <= Se; ++k) {
                                            const int rs = huff_decode(br, hac[c->ta]), r = rs >> 4, s = rs & 15;
                                            if (s) { k += r; if (k > 63) break; blk[kZigzag[k]] = (int16_t)(extend(br.get(s), s) * (1 << Al)); }
                                            else if (r == 15) k += 15;
                                            else { eobrun = 1u << r; if (r) eobrun += (unsigned)br.get(r); --eobrun; break; }
                                        }
                                    } else {                               // decode_mcu_AC_refine
                                        int k = Ss;
                                        if (eobrun == 0) {
                                            for (; k <= Se; ++k) {
                                                const int rs = huff_decode(br, hac[c->ta]); int r = rs >> 4, s = rs & 15;
                                                if (s) s = br.get(1) ? p1 : m1;                  // a newly nonzero coefficient: its sign
                                                else if (r != 15) { eobrun = 1u << r; if (r) eobrun += (unsigned)br.get(r); break; }
                                                // skip r zero-history coefficients, appending a correction bit to every nonzero one on the way
                                                do {
                                                    int16_t& cf = blk[kZigzag[k]];
                                                    if (cf != 0) { if (br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1)); }
                                                    else if (--r < 0) break;
                                                    ++k;
                                                } while (k <= Se);
                                                if (s && k <= 63) blk[kZigzag[k]] = (int16_t)s;
                                            }
                                        }
                                        if (eobrun > 0) {                  // the rest of the band: correction bits of the already nonzero coefficients only
                                            for (; k <= Se; ++k) {
                                                int16_t& cf = blk[kZigzag[k]];
                                                if (cf != 0 && br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                                            }
                                            --eobrun;
                                        }
                                    }
                                }
                        }
                        if (restart) --rst_left;
                    }
                any_scan = true;
                size_t q = pos + 2 + len;
                while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] != 0 && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7) && d[q + 1] != 0xFF)) ++q;
                pos = q;
                continue;
            }
            // interleaved scan: MCUs of hmax x vmax blocks; single-component scan: one block per "MCU", only the blocks covering real samples
            const bool inter = ns > 1;
            const int nmx = inter ? mcux : (sc[0]->dw + 7) / 8, nmy = inter ? mcuy : (sc[0]->dh + 7) / 8;
            int rst_left = restart, next_rst = 0;
            for (int my = 0; my < nmy; ++my)
                for (int mx = 0; mx < nmx; ++mx) {
                    if (restart && rst_left == 0) {
                        // align to the next marker, expect RSTn
                        br.reset();
                        size_t q = br.pos;
                        while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7)) ++q;
                        if (q + 1 < d.size()) q += 2;
                        br.pos = q; br.reset();
                        next_rst = (next_rst + 1) & 7; rst_left = restart;
                        for (auto* c : sc) c->pred = 0;
                    }
                    for (auto* c : sc) {
                        const int bxn = inter ? c->h : 1, byn = inter ? c->v : 1;
                        for (int by = 0; by < byn; ++by)
                            for (int bx = 0; bx < bxn; ++bx) {
                                const int gx = mx * bxn + bx, gy = my * byn + by;
                                int16_t* blk = &c->coef[((size_t)gy * c->bw + gx) * 64];
                                const int s = huff_decode(br, hdc[c->td]);
                                c->pred += extend(br.get(s), s);
                                blk[0] = (int16_t)c->pred;
                                for (int k = 1; k < 64;) {
                                    const int rs = huff_decode(br, hac[c->ta]), r = rs >> 4, ss = rs & 15;
                                    if (ss == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) break;
                                    blk[kZigzag[k]] = (int16_t)extend(br.get(ss), ss);
                                    ++k;
                                }
                            }
                    }
                    if (restart) --rst_left;
                }
            any_scan = true;
            // continue behind the entropy-coded segment: the next marker that is not RSTn / stuffed zero
            size_t q = pos + 2 + len;
            while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] != 0 && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7) && d[q + 1] != 0xFF)) ++q;
            pos = q;
            continue;
        }
        pos += 2 + len;
    }
    if (!have_frame || !any_scan) { err = "no image data"; return false; }
    for (auto& c : comps) if (!qset[c.tq]) { err = "missing quantisation table"; return false; }
    // inverse DCT into component planes (padded to whole blocks)
    for (auto& c : comps) {
        const int pw = c.bw * 8, ph = c.bh * 8;
        c.plane.assign((size_t)pw * ph, 0);
        for (int by = 0; by < c.bh; ++by)
            for (int bx = 0; bx < c.bw; ++bx)
                idct_islow(&c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq], &c.plane[(size_t)by * 8 * pw + bx * 8], pw);
        c.coef.clear(); c.coef.shrink_to_fit();
    }
    // upsample every component to full resolution
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ++ci) {
        Comp& c = comps[ci];
        const int pw = c.bw * 8;
        const int fx = hmax / c.h, fy = vmax / c.v;
        if (hmax % c.h || vmax % c.v) { err = "fractional sampling factors are not supported"; return false; }
        std::vector<uint8_t>& o = full[ci];
        o.assign((size_t)W * H, 0);
        auto in = [&](int y, int x) -> int { return c.plane[(size_t)y * pw + x]; };
        if (fx == 1 && fy == 1) {
            for (int y = 0; y < H; ++y) memcpy(&o[(size_t)y * W], &c.plane[(size_t)y * pw], W);
        } else if (fx == 2 && fy == 1 && c.dw > 2) {        // h2v1_fancy_upsample (jdsample.c)
            std::vector<uint8_t> row(2 * (size_t)c.dw + 2);
            for (int y = 0; y < H; ++y) {
                const int n = c.dw;
                row[0] = (uint8_t)in(y, 0); row[1] = (uint8_t)((in(y, 0) * 3 + in(y, 1) + 2) >> 2);
                for (int i = 1; i < n - 1; ++i) { const int v = in(y, i) * 3; row[2 * i] = (uint8_t)((v + in(y, i - 1) + 1) >> 2); row[2 * i + 1] = (uint8_t)((v + in(y, i + 1) + 2) >> 2); }
                row[2 * n - 2] = (uint8_t)((in(y, n - 1) * 3 + in(y, n - 2) + 1) >> 2); row[2 * n - 1] = (uint8_t)in(y, n - 1);
                memcpy(&o[(size_t)y * W], row.data(), W);
            }
        } else if (fx == 2 && fy == 2 && c.dw > 2) {        // h2v2_fancy_upsample: 3/4 nearer row + 1/4 farther row, then the same horizontally
            std::vector<uint8_t> row(2 * (size_t)c.dw + 2);
            std::vector<int> cs(c.dw);
            for (int y = 0; y < H; ++y) {
                const int iy = y >> 1;
                int oy = (y & 1) ? iy + 1 : iy - 1;                      // the farther input row; the image's edge rows stand in for missing context
                if (oy < 0) oy = 0;
                if (oy > c.dh - 1) oy = c.dh - 1;
                const int n = c.dw;
                for (int i = 0; i < n; ++i) cs[i] = in(iy, i) * 3 + in(oy, i);
                row[0] = (uint8_t)((cs[0] * 4 + 8) >> 4); row[1] = (uint8_t)((cs[0] * 3 + cs[1] + 7) >> 4);
                for (int i = 1; i < n - 1; ++i) { row[2 * i] = (uint8_t)((cs[i] * 3 + cs[i - 1] + 8) >> 4); row[2 * i + 1] = (uint8_t)((cs[i] * 3 + cs[i + 1] + 7) >> 4); }
                row[2 * n - 2] = (uint8_t)((cs[n - 1] * 3 + cs[n - 2] + 8) >> 4); row[2 * n - 1] = (uint8_t)((cs[n - 1] * 4 + 7) >> 4);
                memcpy(&o[(size_t)y * W], row.data(), W);
            }
        } else {                                            // int_upsample: replication
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) o[(size_t)y * W + x] = (uint8_t)in(y / fy, x / fx);
        }
        c.plane.clear(); c.plane.shrink_to_fit();
    }
    img.h = H; img.w = W; img.px.assign((size_t)H * W * 3, 0);
    if (comps.size() == 1) {
        for (size_t i = 0; i < (size_t)W * H; ++i) { img.px[3 * i] = img.px[3 * i + 1] = img.px[3 * i + 2] = full[0][i]; }
        return true;
    }
    const bool rgb_direct = (adobe && adobe_transform == 0) || (!adobe && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
    if (rgb_direct) {
        for (size_t i = 0; i < (size_t)W * H; ++i) { img.px[3 * i] = full[2][i]; img.px[3 * i + 1] = full[1][i]; img.px[3 * i + 2] = full[0][i]; }
        return true;
    }
    // jdcolor.c ycc_rgb_convert tables
    int crr[256], cbb[256]; long crg[256], cbg[256];
    for (int i = 0; i < 256; ++i) {
        const long x = i - 128;
        crr[i] = (int)((91881L * x + 32768L) >> 16);       // FIX(1.40200)
        cbb[i] = (int)((116130L * x + 32768L) >> 16);      // FIX(1.77200)
        crg[i] = -46802L * x;                              // FIX(0.71414)
        cbg[i] = -22554L * x + 32768L;                     // FIX(0.34414) + ONE_HALF
    }
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const int y = full[0][i], cb = full[1][i], cr = full[2][i];
        img.px[3 * i + 2] = clamp8(y + crr[cr]);
        img.px[3 * i + 1] = clamp8(y + (int)((cbg[cb] + crg[cr]) >> 16));
        img.px[3 * i] = clamp8(y + cbb[cb]);
    }
    return true;
}

}  // namespace jpegio

// cv::imread by content: PNG or JPEG by signature (the file extension is not consulted, like OpenCV's decoder lookup)
namespace imgio {
inline bool read(const std::string& path, ImageBGR& img, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open"; return false; }
    uint8_t sig[4] = {0, 0, 0, 0};
    const size_t n = fread(sig, 1, 4, f);
    fclose(f);
    if (n >= 2 && sig[0] == 0xFF && sig[1] == 0xD8) return jpegio::read(path, img, err);
    return pngio::read(path, img, err);
}
}  // namespace imgio
