// packer.cpp — host side of the arena format: Git object framing, SHA-1
// padding, length binning and the synthetic byte stream.  No device code.
//
// Arena slot of one object (all offsets multiples of 64):
//   "<type> <decimal len>\0"   header, as reference src/commit.cpp:90 builds it
//   content bytes              (src/commit.cpp:91 concatenates the two)
//   0x80, zeros                FIPS 180-4 5.1.1 padding ...
//   8-byte big-endian bit length of header+content, ending the last block
// so that the device kernel only ever sees whole 64-byte blocks; and every
// 4-byte group of that stream is stored byte-reversed, i.e. SHA-1's big-endian
// message word as a native little-endian dword, so that the kernel swaps
// nothing (16 vector ops per block that the copy here absorbs for free).
#include "../../include/vit_hash.h"
#include "../../include/vit_hash_bench.h"
#include "host_util.h"
#include "packer.h"

#include <algorithm>
#include <cstring>
#include <immintrin.h>
#include <numeric>

namespace vit {

uint32_t type_len_checked(const char* type)
{
    if (!type) return 0;
    size_t l = std::strlen(type);
    if (l == 0 || l > 8) return 0;           // "blob", "tree", "commit", "tag"
    for (size_t i = 0; i < l; ++i)
        if (type[i] == ' ' || type[i] == '\0') return 0;
    return uint32_t(l);
}

uint32_t write_header(const char* type, uint32_t tl, uint64_t len, uint8_t* out)
{
    if (tl == 0) return 0;                    // raw buffer: no Git framing
    std::memcpy(out, type, tl);
    out[tl] = ' ';
    const uint32_t nd = decimal_digits(len);
    uint64_t v = len;
    for (uint32_t i = 0; i < nd; ++i) { out[tl + nd - i] = uint8_t('0' + v % 10); v /= 10; }
    out[tl + 1 + nd] = 0;
    return tl + 1 + nd + 1;
}

// Cached stores for one short slot (a few blocks): simple and rarely hot.
static void pack_small(const char* type, uint32_t tl, const uint8_t* data, uint64_t len, uint8_t* slot, uint64_t nblocks)
{
    const uint32_t hl = write_header(type, tl, len, slot);
    if (len) std::memcpy(slot + hl, data, size_t(len));
    const uint64_t msg = hl + len, total = nblocks * 64;
    slot[msg] = 0x80;
    std::memset(slot + msg + 1, 0, size_t(total - 8 - (msg + 1)));
    const uint64_t bits = msg * 8;
    for (int i = 0; i < 8; ++i) slot[total - 8 + i] = uint8_t(bits >> (56 - 8 * i));
    for (uint64_t o = 0; o < total; o += 4) {              // message words -> native dwords
        uint32_t w;
        std::memcpy(&w, slot + o, 4);
        w = __builtin_bswap32(w);
        std::memcpy(slot + o, &w, 4);
    }
}

// 64 message bytes from src (any alignment) to dst (64-byte aligned): words
// byte-reversed, non-temporal stores.
__attribute__((target("ssse3"))) static inline void stream64(uint8_t* dst, const uint8_t* src)
{
    const __m128i rev = _mm_set_epi8(12, 13, 14, 15, 8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3);
    for (int k = 0; k < 4; ++k)
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + k,
                         _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + k), rev));
}

// The staging arena is written once and next read by the copy engine, never by
// this core again: non-temporal stores skip the read-for-ownership of every
// destination line and keep the source data from being evicted by it, which
// matters once packing, the copy engine and the host hash lanes share the
// memory controllers.  The slot is 64-byte aligned; the header shifts the
// content by hl bytes, so loads are unaligned and stores aligned.
__attribute__((target("ssse3"))) static void pack_streaming(const char* type, uint32_t tl, const uint8_t* data,
                                                            uint64_t len, uint8_t* slot, uint64_t nblocks)
{
    alignas(64) uint8_t edge[128];
    const uint32_t hl = write_header(type, tl, len, edge);       // hl <= 31, len >= 64*3 - 40 here
    const uint64_t msg = hl + len, total = nblocks * 64;
    std::memcpy(edge + hl, data, 64 - hl);
    stream64(slot, edge);
    uint64_t off = 64;
    for (; off + 64 <= msg; off += 64) stream64(slot + off, data + (off - hl));   // message offset off = content offset off - hl
    // What is left: under 64 content bytes, the 0x80, zeros and the bit length, in one or two blocks.
    const uint64_t left = total - off;                           // 64 or 128
    std::memset(edge, 0, sizeof edge);
    if (msg > off) std::memcpy(edge, data + (off - hl), size_t(msg - off));
    edge[msg - off] = 0x80;
    const uint64_t bits = msg * 8;
    for (int i = 0; i < 8; ++i) edge[left - 8 + i] = uint8_t(bits >> (56 - 8 * i));
    stream64(slot + off, edge);
    if (left == 128) stream64(slot + off + 64, edge + 64);
}

void pack_one(const char* type, uint32_t tl, const uint8_t* data, uint64_t len, uint8_t* slot, uint64_t nblocks)
{
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (nblocks < 4 || !ssse3 || (reinterpret_cast<uintptr_t>(slot) & 63)) pack_small(type, tl, data, len, slot, nblocks);
    else pack_streaming(type, tl, data, len, slot, nblocks);
}

void pack_fence() { _mm_sfence(); }

// LSD radix sort of indices by block count, descending, stable.
bool bin_order(const uint32_t* nblocks, uint64_t n, uint32_t* order)
{
    bool uniform = true;
    for (uint64_t i = 1; i < n && uniform; ++i) uniform = nblocks[i] == nblocks[0];
    std::iota(order, order + n, 0u);
    if (uniform) return false;
    std::vector<uint32_t> tmp(n);
    uint32_t* src = order;
    uint32_t* dst = tmp.data();
    uint32_t maxv = 0;
    for (uint64_t i = 0; i < n; ++i) maxv = std::max(maxv, nblocks[i]);
    for (int shift = 0; shift < 32 && (maxv >> shift); shift += 11) {
        uint64_t count[2049] = {0};
        // descending: bucket by the complemented digit
        for (uint64_t i = 0; i < n; ++i) ++count[(2047u - ((nblocks[src[i]] >> shift) & 2047u)) + 1];
        for (int d = 0; d < 2048; ++d) count[d + 1] += count[d];
        for (uint64_t i = 0; i < n; ++i) dst[count[2047u - ((nblocks[src[i]] >> shift) & 2047u)]++] = src[i];
        std::swap(src, dst);
    }
    if (src != order) std::memcpy(order, src, n * sizeof(uint32_t));
    return true;
}

void plan_host_split(const uint64_t* len, uint64_t n, const SplitRates& r, int queues, int threads,
                     std::vector<uint64_t>& out, const ParallelFor* parallel_for)
{
    out.clear();
    if (!(r.lane > 0) || !(r.core > 0) || !(r.copy > 0) || !(r.pack > 0) || queues < 1 || threads < 1) return;
    // Candidates as (length, index) pairs, so that the sort below streams through
    // memory instead of looking every length up by index.
    struct Cand { uint64_t len, idx; };
    uint64_t total = 0, maxlen = 0;
    std::vector<Cand> cand;
    if (parallel_for && n >= (1u << 18)) {
        // parts in index order, each with its own list; joined in that order, the
        // candidates are in index order as the serial scan leaves them
        constexpr size_t kParts = 64;
        struct Part { std::vector<Cand> c; uint64_t total = 0, maxlen = 0; };
        std::vector<Part> part(kParts);
        (*parallel_for)(kParts, [&](size_t p) {
            Part& me = part[p];
            for (uint64_t i = n * p / kParts; i < n * (p + 1) / kParts; ++i) {
                me.total += len[i];
                if (len[i] >= r.min_bytes) { me.c.push_back(Cand{len[i], i}); me.maxlen = std::max(me.maxlen, len[i]); }
            }
        });
        size_t count = 0;
        for (const Part& p : part) { count += p.c.size(); total += p.total; maxlen = std::max(maxlen, p.maxlen); }
        cand.reserve(count);
        for (const Part& p : part) cand.insert(cand.end(), p.c.begin(), p.c.end());
    } else {
        cand.reserve(size_t(n / 8 + 16));
        for (uint64_t i = 0; i < n; ++i) {
            total += len[i];
            if (len[i] >= r.min_bytes) { cand.push_back(Cand{len[i], i}); maxlen = std::max(maxlen, len[i]); }
        }
    }
    if (cand.empty()) return;
    // Longest first, equal lengths in index order: a stable LSD radix sort on the
    // lengths (a slice of a large repository has hundreds of thousands of
    // candidates per call; a comparison sort of them costs more than the hashing
    // of a chunk).  All digit histograms come from one pass.
    {
        constexpr int kBits = 11, kDigits = 1 << kBits;
        int passes = 0;
        while (passes * kBits < 64 && (maxlen >> (passes * kBits))) ++passes;
        std::vector<size_t> count(size_t(passes) * (kDigits + 1), 0);
        for (const Cand& c : cand)
            for (int p = 0; p < passes; ++p) ++count[size_t(p) * (kDigits + 1) + (kDigits - 1 - ((c.len >> (p * kBits)) & (kDigits - 1))) + 1];
        std::vector<Cand> tmp(cand.size());
        Cand* src = cand.data();
        Cand* dst = tmp.data();
        for (int p = 0; p < passes; ++p) {
            size_t* cnt = &count[size_t(p) * (kDigits + 1)];
            for (int d = 0; d < kDigits; ++d) cnt[d + 1] += cnt[d];
            for (size_t k = 0; k < cand.size(); ++k) dst[cnt[kDigits - 1 - ((src[k].len >> (p * kBits)) & (kDigits - 1))]++] = src[k];
            std::swap(src, dst);
        }
        if (src != cand.data()) std::memcpy(cand.data(), src, cand.size() * sizeof(Cand));
    }
    const double nt = double(threads);
    double copy_all = r.copy * double(queues);
    if (r.copy_total > 0) copy_all = std::min(copy_all, r.copy_total);     // queues that share a link or the host's memory
    const double queue_rate = std::min(0.8 * copy_all, r.pack * nt);
    const double longest = double(cand[0].len);
    double gpu_bytes = double(total), host_bytes = 0;
    size_t moved = 0;
    for (const Cand& c : cand) {
        const double L = double(c.len);
        const double gpu_time = std::max(gpu_bytes / queue_rate, L / r.lane);
        // streams a core would run side by side with this many objects moved: 1 .. 4
        const double per_thread = double(moved + 1) / nt;
        const double streams = per_thread < 1 ? 1 : (per_thread > 4 ? 4 : per_thread);
        const double core_eff = r.core_multi > r.core ? r.core + (r.core_multi - r.core) * (streams - 1) / 3 : r.core;
        const double host_time = std::max(r.host_bias * ((host_bytes + L) / core_eff + (gpu_bytes - L) / r.pack) / nt, longest / r.core);
        if (host_time >= gpu_time) break;
        gpu_bytes -= L; host_bytes += L;
        ++moved;
    }
    out.resize(moved);
    for (size_t k = 0; k < moved; ++k) out[k] = cand[k].idx;
}

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace vit

extern "C" {

uint32_t vit_header_len(const char* type, uint64_t len)
{
    const uint32_t tl = vit::type_len_checked(type);
    if (!tl) return 0;
    return tl + 1 + vit::decimal_digits(len) + 1;
}

uint32_t vit_write_header(const char* type, uint64_t len, uint8_t* out)
{
    const uint32_t tl = vit::type_len_checked(type);
    if (!tl || !out) return 0;
    return vit::write_header(type, tl, len, out);
}

uint64_t vit_object_blocks(const char* type, uint64_t len)
{
    const uint32_t hl = vit_header_len(type, len);
    if (!hl) return 0;
    return (hl + len + 9 + 63) / 64;
}

int vit_arena_pack(const char* type, const uint8_t* const* data, const uint64_t* len, uint64_t n,
                   void* arena, uint64_t arena_bytes, uint64_t* offsets, uint32_t* nblocks,
                   uint64_t* used_bytes, int threads)
{
    return vit_arena_pack_aligned(type, data, len, n, arena, arena_bytes, offsets, nblocks, used_bytes, threads, 64);
}

int vit_arena_pack_aligned(const char* type, const uint8_t* const* data, const uint64_t* len, uint64_t n,
                           void* arena, uint64_t arena_bytes, uint64_t* offsets, uint32_t* nblocks,
                           uint64_t* used_bytes, int threads, uint32_t slot_align)
{
    const uint32_t tl = vit::type_len_checked(type);
    if (!tl || (n && (!data || !len || !arena || !offsets || !nblocks))) return VIT_ERR_ARG;
    if (slot_align < 64 || (slot_align & (slot_align - 1))) return VIT_ERR_ARG;
    try {
    const uint64_t per = slot_align / 64;                   // slot starts, in blocks
    std::vector<uint64_t> prefix(n + 1, 0);                 // prefix[i] = start of slot i, in blocks
    std::vector<uint32_t> nb_of(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (len[i] && !data[i]) return VIT_ERR_ARG;
        const uint64_t nb = (tl + 1 + vit::decimal_digits(len[i]) + 1 + len[i] + 9 + 63) / 64;
        if (nb > 0xFFFFFFFFull) return VIT_ERR_RANGE;
        nb_of[i] = uint32_t(nb);
        prefix[i + 1] = (prefix[i] + nb + per - 1) / per * per;
    }
    const uint64_t total = n ? (prefix[n - 1] + nb_of[n - 1]) * 64 : 0;
    if (used_bytes) *used_bytes = total;
    if (total > arena_bytes) return VIT_ERR_RANGE;
    for (uint64_t i = 0; i < n; ++i) {
        offsets[i] = prefix[i] * 64;
        nblocks[i] = nb_of[i];
    }
    uint8_t* base = static_cast<uint8_t*>(arena);
    if (total < (8u << 20)) threads = 1;
    vit::parallel_ranges(n, threads, prefix.data(), [&](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) vit::pack_one(type, tl, data[i], len[i], base + offsets[i], nblocks[i]);
        vit::pack_fence();
    });
    return VIT_OK;
    } catch (const std::exception&) { return VIT_ERR_NOMEM; }     // nothing propagates out of the C ABI
}

int vit_bin_order(const uint32_t* nblocks, uint64_t n, uint32_t* order)
{
    if (n && (!nblocks || !order)) return VIT_ERR_ARG;
    if (n > 0xFFFFFFFFull) return VIT_ERR_RANGE;
    try { return vit::bin_order(nblocks, n, order) ? 1 : 0; }
    catch (const std::exception&) { return VIT_ERR_NOMEM; }
}

int vit_host_split_plan(const uint64_t* len, uint64_t n, uint64_t lane_rate, uint64_t core_rate, uint64_t copy_rate,
                        uint64_t copy_rate_total, uint64_t pack_rate, int n_queues, int threads, uint8_t* on_host)
{
    if (n && (!len || !on_host)) return VIT_ERR_ARG;
    if (!lane_rate || !core_rate || !copy_rate || !pack_rate || n_queues < 1 || threads < 1) return VIT_ERR_ARG;
    try {
        std::vector<uint64_t> moved;
        vit::SplitRates rates{double(lane_rate), double(core_rate), double(copy_rate), double(pack_rate)};
        rates.copy_total = double(copy_rate_total);
        vit::plan_host_split(len, n, rates, n_queues, threads, moved);
        std::memset(on_host, 0, size_t(n));
        for (uint64_t i : moved) on_host[i] = 1;
        return VIT_OK;
    } catch (const std::exception&) { return VIT_ERR_NOMEM; }
}

void vit_synth_bytes(uint64_t seed, uint64_t index, uint8_t* dst, uint64_t len)
{
    const uint64_t key = seed ^ (index * 0xD1342543DE82EF95ull);
    uint64_t i = 0, k = 0;
    for (; i + 8 <= len; i += 8, ++k) {
        const uint64_t w = vit::mix64(key + (k + 1) * 0x9E3779B97F4A7C15ull);
        std::memcpy(dst + i, &w, 8);       // little-endian host
    }
    if (i < len) {
        const uint64_t w = vit::mix64(key + (k + 1) * 0x9E3779B97F4A7C15ull);
        std::memcpy(dst + i, &w, size_t(len - i));
    }
}


int vit_synth_fill_host(uint64_t seed, uint64_t first_index, uint64_t n, const uint64_t* lens, const uint64_t* offsets,
                        uint8_t* dst, int threads)
{
    if (n && (!lens || !offsets || !dst)) return VIT_ERR_ARG;
    try {
        std::vector<uint64_t> prefix(n + 1, 0);
        for (uint64_t i = 0; i < n; ++i) prefix[i + 1] = prefix[i] + lens[i] + 64;      // weight: bytes, and a little per object
        vit::parallel_ranges(n, threads < 1 ? 1 : threads, prefix.data(), [&](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) vit_synth_bytes(seed, first_index + i, dst + offsets[i], lens[i]);
        });
        return VIT_OK;
    } catch (const std::exception&) { return VIT_ERR_NOMEM; }
}

}  // extern "C"
