// Host-side packer and SHA-1 lane under AddressSanitizer / UBSan (built and run by
// tests/test_sanitize_host.py): random batches, content buffers of exact size, every packed slot
// compared byte for byte with the framed, padded, word-swapped message.
#include "vit_hash.h"
#include "packer.h"
#include "sha1_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    for (int round = 0; round < 300; ++round) {
        const int n = 1 + rng() % 40;
        std::vector<std::vector<uint8_t>> blobs(n);
        std::vector<const uint8_t*> ptr(n); std::vector<uint64_t> len(n);
        uint64_t need = 0;
        for (int i = 0; i < n; ++i) {
            uint64_t L = (rng() % 4 == 0) ? rng() % 70000 : rng() % 600;
            blobs[i].resize(L);                       // exact size: an over-read trips the sanitizer
            for (auto& b : blobs[i]) b = uint8_t(rng());
            ptr[i] = L ? blobs[i].data() : nullptr; len[i] = L;
            need += (vit_object_blocks("blob", L) * 64 + 127) / 128 * 128;
        }
        const uint32_t align = (round & 1) ? 128 : 64;
        void* arena = nullptr;
        if (posix_memalign(&arena, 128, need + 128)) return 2;
        std::vector<uint64_t> off(n); std::vector<uint32_t> nb(n); uint64_t used = 0;
        int rc = vit_arena_pack_aligned("blob", ptr.data(), len.data(), n, arena, need, off.data(), nb.data(), &used, 1 + round % 4, align);
        if (rc != 0) { std::printf("rc %d\n", rc); return 1; }
        for (int i = 0; i < n; ++i) {                 // every slot: words byte-reversed message
            const uint8_t* s = static_cast<uint8_t*>(arena) + off[i];
            uint8_t hdr[32]; uint32_t hl = vit_write_header("blob", len[i], hdr);
            const uint64_t msg = hl + len[i], total = uint64_t(nb[i]) * 64;
            for (uint64_t p = 0; p < total; ++p) {
                uint8_t want = p < hl ? hdr[p] : p < msg ? blobs[i][p - hl] : p == msg ? 0x80 : p >= total - 8 ? uint8_t((msg * 8) >> (8 * (total - 1 - p))) : 0;
                if (s[p ^ 3] != want) { std::printf("mismatch round %d obj %d pos %llu\n", round, i, (unsigned long long)p); return 1; }
            }
            uint8_t d1[20];
            vit::host_object_id("blob", ptr[i], len[i], d1);
            vit::host_object_id(nullptr, ptr[i], len[i], d1);
            // the same buffer without Git framing (type length 0), as vit_sha1_buffers packs it
            const uint64_t rb = (len[i] + 9 + 63) / 64;
            void* raw = nullptr;
            if (posix_memalign(&raw, 64, rb * 64)) return 2;
            vit::pack_one(nullptr, 0, ptr[i], len[i], static_cast<uint8_t*>(raw), rb);
            vit::pack_fence();
            const uint8_t* r = static_cast<uint8_t*>(raw);
            for (uint64_t p = 0; p < rb * 64; ++p) {
                uint8_t want = p < len[i] ? blobs[i][p] : p == len[i] ? 0x80 : p >= rb * 64 - 8 ? uint8_t((len[i] * 8) >> (8 * (rb * 64 - 1 - p))) : 0;
                if (r[p ^ 3] != want) { std::printf("raw mismatch round %d obj %d pos %llu\n", round, i, (unsigned long long)p); return 1; }
            }
            std::free(raw);
        }
        std::free(arena);
        // the interleaved host lane (several objects' streams side by side on one core) against one stream at a time,
        // framed and raw, on buffers of exact size and every count up to a few groups
        for (const char* type : {static_cast<const char*>("blob"), static_cast<const char*>(nullptr)}) {
            std::vector<uint8_t> got(size_t(n) * 20), want(size_t(n) * 20);
            vit::host_object_ids(type, ptr.data(), len.data(), size_t(n), got.data());
            for (int i = 0; i < n; ++i) vit::host_object_id(type, ptr[i], len[i], &want[size_t(i) * 20]);
            if (got != want) { std::printf("interleaved host lane differs, round %d\n", round); return 1; }
        }
    }
    std::puts("pack + host lane clean");
    return 0;
}
