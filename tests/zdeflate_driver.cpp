// The deflate restatement of vit_amd/csrc/zdeflate.h against the zlib library itself (compress2 at the default
// level, which is what the reference's compress() call at src/commit.cpp:100 runs): random, text-like, constant,
// long-distance and short-distance repetitive inputs of every size up to the 64 KiB the restatement covers, byte
// for byte.  Inputs of up to 16,382 bytes (one block) also go through zwave.h's emulation of the wave-per-object formulation (chains
// in a bucket table, two precomputed search answers per position, the decision loop over them, symbols coded by lanes).
// Built and run by tests/test_zdeflate.py:  zdeflate_driver <inputs> <seed> [wave-only]
#include "zdeflate.h"
#include "zwave.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <string>
struct PtrIn {
    const uint8_t* p; uint32_t n;
    uint8_t operator()(uint32_t i) const { return p[i]; }
    // bytes i..i+3, the first in the top byte; what lies beyond the input is the caller's to mask
    uint32_t be32a(uint32_t i) const { return be32(i); }    // the same from a multiple of four
    uint32_t be32(uint32_t i) const { uint32_t w = 0; for (uint32_t k = 0; k < 4; ++k) w = (w << 8) | (i + k < n ? p[i + k] : 0xEEu); return w; }
};
int main(int argc, char** argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    const bool wave_only = argc > 3;                            // smaller inputs, the wave formulation alone: many more of them per second
    int waved = 0;
    zd::Tables t; zd::make_tables(t);
    for (uint32_t lc = 0; lc < 256; ++lc) if (zd::length_code_of(lc) != t.length_code[lc]) { printf("length_code_of(%u)\n", lc); return 3; }
    for (uint32_t d = 0; d < 32768; ++d) if (zd::dist_code_of(d) != (d < 256 ? t.dist_code[d] : t.dist_code[256 + (d >> 7)])) { printf("dist_code_of(%u)\n", d); return 3; }
    std::vector<uint16_t> head(zd::kHashSize * 3), prev(32768 * 3), dbuf(zd::kLitBufSize * 3);
    std::vector<uint8_t> lbuf(zd::kLitBufSize * 3);
    std::vector<uint16_t> small(zd::kSmallEntries * 5);
    std::vector<uint32_t> heap(zd::kHeapEntries * 5), heap_low(2 * 288 + 2);
    std::mt19937_64 rng(atoi(argc > 2 ? argv[2] : "777"));
    static zd::Deflater<PtrIn> d;
    int bad = 0; size_t total = 0;
    const char* words[] = {"the ", "quick ", "brown ", "fox ", "int ", "return ", "void ", "#include ", "\n", "    ", "for (", "i++) {", "}", "const ", "0x", "==", "abcabcabc"};
    for (int r = 0; r < rounds; ++r) {
        uint32_t n;
        switch (rng() % 8) { case 0: n = rng() % 40; break; case 1: n = rng() % 700; break; case 2: n = 4000 + rng() % 300; break;
            case 3: n = 16000 + rng() % 1000; break; case 4: n = 32700 + rng() % 200; break; case 5: n = 65536 - rng() % 3; break; default: n = rng() % 65537; }
        if (wave_only) {
            static const uint32_t edges[] = {512, 1024, 1536, 2048, 4096, 4608, 9216};     // tiles and size classes of the kernels
            switch (rng() % 8) { case 0: n = rng() % 64; break; case 1: n = 4000 + rng() % 700; break; case 2: n = 9000 + rng() % 300; break;
                case 3: n = zw::kMaxInput - rng() % 40; break; case 4: n = edges[rng() % 7] - 17 + rng() % 35; break; default: n = rng() % 9300; }
        }
        std::vector<uint8_t> in(n + 1);
        int kind = rng() % 7;
        if (kind == 0) for (auto& b : in) b = uint8_t(rng());
        else if (kind == 1) for (auto& b : in) b = uint8_t(rng() % 4 + 'a');
        else if (kind == 2) { size_t k = 0; while (k < n) { const char* w = words[rng() % 17]; size_t l = strlen(w); for (size_t j = 0; j < l && k < n; ++j) in[k++] = uint8_t(w[j]); } }
        else if (kind == 3) { uint8_t v = uint8_t(rng()); for (auto& b : in) b = v; }
        else if (kind == 4) { for (size_t k = 0; k < n; ++k) in[k] = (k >= 5000 && rng() % 10) ? in[k - 5000 + (rng() % 3 == 0)] : uint8_t(rng()); }
        else if (kind == 5) { for (size_t k = 0; k < n; ++k) in[k] = (k >= 40 && rng() % 50) ? in[k - 1 - rng() % 40] : uint8_t(rng() % 7); }
        else { for (size_t k = 0; k < n; ++k) in[k] = uint8_t((k * 7 + (k >> 8)) ^ (rng() % 64 == 0 ? rng() : 0)); }
        if (rng() % 3 == 0 && n > 40) {                          // a tail of zeros or of a repeated pattern: matches that run into the end of the input
            const size_t tail = 1 + rng() % 40;
            for (size_t k = n - tail; k < n; ++k) in[k] = rng() % 2 ? 0 : in[k - 1 - rng() % 3];
        }
        uLongf zl = compressBound(n);
        std::vector<uint8_t> z(zl);
        if (compress2(z.data(), &zl, in.data(), n, 6) != Z_OK) { puts("zlib failed"); return 2; }
        std::vector<uint8_t> out(zd::bound(n));
        if (n <= zw::kMaxInput) {
            std::vector<uint8_t> wo(zd::bound(n));
            const uint32_t wg = zw::deflate_emulated(&t, PtrIn{in.data(), n}, n, wo.data(), uint32_t(wo.size()));
            ++waved;
            if (wg != zl || memcmp(wo.data(), z.data(), zl)) {
                size_t k = 0; while (k < wg && k < zl && wo[k] == z[k]) ++k;
                printf("WAVE MISMATCH round %d kind %d n %u: got %u bytes, zlib %lu, first difference at %zu\n", r, kind, n, wg, (unsigned long)zl, k);
                if (++bad > 5) return 1;
            }
        }
        if (wave_only) { total += n; continue; }
        std::fill(head.begin(), head.end(), 0);
        const uint32_t st = r % 2 ? 3 : 1;                     // the arrays strided as the GPU lanes interleave theirs, and plain
        zd::Work w{{head.data(), st}, {prev.data(), st}, {dbuf.data(), st}, {lbuf.data(), st}, small.data() + (r % 3), heap.data() + (r % 4), uint32_t(r % 2 ? 5 : 1), out.data(), uint32_t(out.size())};
        if (r % 3 == 0) {                                       // the Huffman heap in two places, as the GPU's block writer keeps it
            w.heap = heap_low.data() + (r % 2); w.heap_stride = 2; w.heap_high = heap.data() + (r % 4); w.heap_split = r % 2 ? 288 : 40;
        }
        d.in = PtrIn{in.data(), n};
        uint32_t got = d.run(&t, w, n);
        total += n;
        if (got != zl || memcmp(out.data(), z.data(), zl)) {
            size_t k = 0; while (k < got && k < zl && out[k] == z[k]) ++k;
            printf("MISMATCH round %d kind %d n %u: got %u bytes, zlib %lu, first difference at %zu\n", r, kind, n, got, (unsigned long)zl, k);
            if (++bad > 5) return 1;
        }
    }
    printf("%d inputs (%d also through the wave formulation), %zu bytes, %d mismatches\n", rounds, waved, total, bad);
    return bad != 0;
}
