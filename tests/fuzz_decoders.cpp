// Mutation fuzzer for the CLI's image decoders (host/png_io.h, host/jpeg_io.h): built by tests/test_cli.py with
// -fsanitize=address,undefined; every mutated file must either decode or be rejected with a message — never crash, hang or
// allocate past the decoders' own limits. usage: fuzz_decoders <iterations> <seed> <scratch file> <seed image>...
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>
#include "png_io.h"
#include "jpeg_io.h"

static uint64_t rng_state;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int iters = atoi(argv[1]);
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[2]);
    const std::string scratch = argv[3];
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 4; i < argc; ++i) {
        FILE* f = fopen(argv[i], "rb"); if (!f) return 2;
        std::vector<uint8_t> b; uint8_t buf[4096]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(f); seeds.push_back(b);
    }
    int ok = 0, rejected = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> b = seeds[rnd() % seeds.size()];
        const int kind = rnd() % 6;
        if (kind == 0 && b.size() > 8) b.resize(8 + rnd() % (b.size() - 8));                     // truncate
        else if (kind == 1) { const std::vector<uint8_t>& o = seeds[rnd() % seeds.size()];        // splice another file's tail
                              const size_t cut = rnd() % b.size(), from = rnd() % o.size(); b.resize(cut); b.insert(b.end(), o.begin() + from, o.end()); }
        else if (kind == 2 && b.size() > 64) { const size_t p = 2 + rnd() % 60; b[p] = (uint8_t)rnd(); }     // header byte
        else { const int nflip = 1 + rnd() % 8; for (int k = 0; k < nflip; ++k) { const size_t p = rnd() % b.size(); b[p] = (kind == 3) ? (uint8_t)0xFF : (uint8_t)rnd(); } }
        FILE* f = fopen(scratch.c_str(), "wb"); if (!f) return 2;
        fwrite(b.data(), 1, b.size(), f); fclose(f);
        ImageBGR im; std::string err;
        if (imgio::read(scratch, im, err)) { if (im.px.size() != (size_t)im.h * im.w * 3) { printf("size mismatch\n"); return 1; } ++ok; }
        else { if (err.empty()) { printf("rejected without a message\n"); return 1; } ++rejected; }
    }
    printf("decoded %d rejected %d\n", ok, rejected);
    return 0;
}
