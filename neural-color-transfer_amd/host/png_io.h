// png_io.h — minimal PNG codec over zlib for the CLI (the reference uses cv::imread / cv::imwrite, main.cu:483,491,538;
// OpenCV/libpng are not available in this environment). imread semantics restated: always 8-bit 3-channel BGR, alpha dropped
// (not blended), 16-bit -> 8-bit (high byte), grayscale replicated, palette expanded. Non-interlaced files only.
#pragma once
#include <zlib.h>
#include <atomic>
#include <random>
#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct ImageBGR { int h = 0, w = 0; std::vector<uint8_t> px; bool ok() const { return w > 0 && h > 0; } };

namespace pngio {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
inline int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

inline bool read(const std::string& path, ImageBGR& img, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open"; return false; }
    std::vector<uint8_t> d;
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool first = true, have_ihdr = false;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]);
        if (pos + 12 + (size_t)len > d.size()) { err = "truncated chunk"; return false; }
        const char* type = (const char*)&d[pos + 4];
        const uint8_t* body = &d[pos + 8];
        // pairs.txt inputs are untrusted files: every chunk's CRC is verified, IHDR must be the first chunk and exactly 13 bytes
        if ((uint32_t)crc32(0L, &d[pos + 4], (uInt)(4 + len)) != be32(&d[pos + 8 + len])) { err = "chunk CRC mismatch"; return false; }
        if (first != !memcmp(type, "IHDR", 4)) { err = "IHDR must be the first chunk (and appear once)"; return false; }
        first = false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR length"; return false; }
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; have_ihdr = true;
        }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w <= 0 || h <= 0 || w > 16384 || h > 16384 || (size_t)w * h > (64u << 20)) { err = "bad dimensions (limit: 64 megapixels)"; return false; }
    if (interlace) { err = "interlaced PNG not supported"; return false; }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || !(depth == 8 || depth == 16 || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) || (ctype == 0 && depth < 8))) { err = "unsupported colour type/depth"; return false; }
    const int bpp_bits = ch * depth;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    const int bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf outlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outlen, idat.data(), (uLong)idat.size()) != Z_OK || outlen != raw.size()) { err = "zlib inflate failed"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    img.h = h; img.w = w; img.px.assign((size_t)h * w * 3, 0);
    for (int y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(stride + 1) * (size_t)y];
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int x = line[1 + i], a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v;
            switch (ft) { case 0: v = x; break; case 1: v = x + a; break; case 2: v = x + b; break; case 3: v = x + ((a + b) >> 1); break; case 4: v = x + paeth(a, b, c); break; default: err = "bad filter"; return false; }
            cur[i] = (uint8_t)v;
        }
        for (int x = 0; x < w; ++x) {
            uint8_t r, g, bl;
            auto sample = [&](int idx) -> int {      // idx-th sample of this pixel, 8-bit result
                if (depth == 8) return cur[(size_t)x * ch + idx];
                if (depth == 16) return cur[((size_t)x * ch + idx) * 2];
                const int per = 8 / depth, sh = (per - 1 - (x % per)) * depth;
                return (cur[x / per] >> sh) & ((1 << depth) - 1);
            };
            if (ctype == 3) { const size_t i = (size_t)sample(0); if (i * 3 + 2 < plte.size()) { r = plte[i * 3]; g = plte[i * 3 + 1]; bl = plte[i * 3 + 2]; } else { r = g = bl = 0; } }
            else if (ctype == 0 || ctype == 4) { int v = sample(0); if (depth < 8) v = v * 255 / ((1 << depth) - 1); r = g = bl = (uint8_t)v; }
            else { r = (uint8_t)sample(0); g = (uint8_t)sample(1); bl = (uint8_t)sample(2); }
            uint8_t* o = &img.px[((size_t)y * w + x) * 3];
            o[0] = bl; o[1] = g; o[2] = r;
        }
        prev.swap(cur);
    }
    return true;
}

inline bool write(const std::string& path, const uint8_t* bgr, int h, int w, std::string& err) {
    std::vector<uint8_t> raw(((size_t)w * 3 + 1) * h);
    for (int y = 0; y < h; ++y) {
        uint8_t* line = &raw[((size_t)w * 3 + 1) * y];
        line[0] = y == 0 ? 1 : 2;            // Sub on the first row, Up elsewhere
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                const int v = bgr[((size_t)y * w + x) * 3 + (2 - c)];
                const int pred = y == 0 ? (x > 0 ? bgr[((size_t)y * w + x - 1) * 3 + (2 - c)] : 0) : bgr[((size_t)(y - 1) * w + x) * 3 + (2 - c)];
                line[1 + (size_t)x * 3 + c] = (uint8_t)(v - pred);
            }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 3) != Z_OK) { err = "zlib deflate failed"; return false; }   // level 3 = cv::imwrite default
    std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
    auto chunk = [&](const char* type, const uint8_t* body, uint32_t len) {
        put32(out, len);
        const size_t s = out.size();
        out.insert(out.end(), type, type + 4);
        if (len) out.insert(out.end(), body, body + len);
        put32(out, (uint32_t)crc32(0L, &out[s], (uInt)(out.size() - s)));
    };
    uint8_t ihdr[13]; uint32_t ww = (uint32_t)w, hh = (uint32_t)h;
    ihdr[0] = ww >> 24; ihdr[1] = ww >> 16; ihdr[2] = ww >> 8; ihdr[3] = ww; ihdr[4] = hh >> 24; ihdr[5] = hh >> 16; ihdr[6] = hh >> 8; ihdr[7] = hh;
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13); chunk("IDAT", comp.data(), (uint32_t)clen); chunk("IEND", nullptr, 0);
    // written under a temporary name and renamed: a run killed mid-write (or a full disk) never leaves a truncated file under the final name,
    // which -resume would take for a finished pair
    // The temporary name is unique per writer (pid + a process-wide counter) and created exclusively: two pairs.txt lines that map to the same output name may be
    // encoded concurrently by the I/O pool, and a stale file of a killed run must not be appended to. The last finished writer wins the rename.
    // A run killed mid-write leaves its temporary behind, and in a container the next run often has the SAME pid (and the counter restarts at 0) — while two LIVE
    // processes in different pid namespaces that share the output directory may have the same pid too: a name that exists (EEXIST) is therefore never deleted, only
    // skipped, and the name carries a per-process random tag besides pid and counter, so a collision needs the same pid AND the same 32 random bits.
    static std::atomic<unsigned long> seq{0};
    static const unsigned long tag = [] { std::random_device rd; return (unsigned long)rd(); }();
    std::string tmp; int fd = -1;
    for (int attempt = 0; attempt < 64 && fd < 0; ++attempt) {
        tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(tag) + "." + std::to_string(seq.fetch_add(1));
        fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0644);
        if (fd < 0 && errno != EEXIST) break;
    }
    FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) close(fd); err = "cannot create"; return false; }
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) { err = "short write"; remove(tmp.c_str()); return false; }
    if (rename(tmp.c_str(), path.c_str()) != 0) { err = "cannot rename the finished file into place"; remove(tmp.c_str()); return false; }
    return true;
}

// -resume: is `path` a complete PNG? (signature, IHDR first, an IEND chunk closing the file) — without decoding it
inline bool looks_complete(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t head[24], tail[12];
    bool ok = fread(head, 1, 24, f) == 24 && fseek(f, -12, SEEK_END) == 0 && fread(tail, 1, 12, f) == 12;
    fclose(f);
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    return ok && !memcmp(head, sig, 8) && !memcmp(head + 12, "IHDR", 4) && !memcmp(tail, iend, 12);
}

}  // namespace pngio
