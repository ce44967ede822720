// sha1_host.cpp — streaming SHA-1 on a host core (see sha1_host.h for its role).
#include "sha1_host.h"

#include "../../include/vit_hash.h"
#include "packer.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace vit {
namespace {

inline uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
inline uint32_t load_be(const uint8_t* p)
{
    return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
}

// Portable block function: rolling 16-word schedule, rounds in groups by phase.
void blocks_portable(uint32_t h[5], const uint8_t* p, size_t nblocks)
{
    while (nblocks--) {
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = load_be(p + 4 * i);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        auto sched = [&](int t) {
            uint32_t x = w[(t - 3) & 15] ^ w[(t - 8) & 15] ^ w[(t - 14) & 15] ^ w[t & 15];
            return w[t & 15] = rol(x, 1);
        };
        auto step = [&](uint32_t f, uint32_t k, uint32_t wt) {
            uint32_t tmp = rol(a, 5) + f + e + k + wt;
            e = d; d = c; c = rol(b, 30); b = a; a = tmp;
        };
        int t = 0;
        for (; t < 16; ++t) step(d ^ (b & (c ^ d)), 0x5A827999u, w[t]);
        for (; t < 20; ++t) step(d ^ (b & (c ^ d)), 0x5A827999u, sched(t));
        for (; t < 40; ++t) step(b ^ c ^ d, 0x6ED9EBA1u, sched(t));
        for (; t < 60; ++t) step((b & c) | (d & (b | c)), 0x8F1BBCDCu, sched(t));
        for (; t < 80; ++t) step(b ^ c ^ d, 0xCA62C1D6u, sched(t));
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
        p += 64;
    }
}

#if defined(__x86_64__)
// x86 SHA extensions: four rounds per sha1rnds4, schedule by sha1msg1/sha1msg2.
__attribute__((target("sha,sse4.1,ssse3")))
void blocks_shani(uint32_t h[5], const uint8_t* p, size_t nblocks)
{
    const __m128i bswap = _mm_set_epi64x(0x0001020304050607ll, 0x08090a0b0c0d0e0fll);
    __m128i abcd = _mm_shuffle_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i*>(h)), 0x1B);
    __m128i e = _mm_set_epi32(int(h[4]), 0, 0, 0);

    while (nblocks--) {
        const __m128i abcd0 = abcd, e0 = e;
        __m128i m[4];
        for (int i = 0; i < 4; ++i)
            m[i] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16 * i)), bswap);

        __m128i prev = abcd, cur;
        // Group g covers rounds 4g..4g+3 with round function g/5; after using
        // m[g&3] it is replaced by the words of group g+4.
#define VIT_G(g)                                                                              \
        cur = (g) == 0 ? _mm_add_epi32(e, m[0]) : _mm_sha1nexte_epu32(prev, m[(g) & 3]);          \
        prev = abcd;                                                                          \
        abcd = _mm_sha1rnds4_epu32(abcd, cur, (g) / 5);                                        \
        if ((g) < 16)                                                                         \
            m[(g) & 3] = _mm_sha1msg2_epu32(                                                   \
                _mm_xor_si128(_mm_sha1msg1_epu32(m[(g) & 3], m[((g) + 1) & 3]), m[((g) + 2) & 3]), \
                m[((g) + 3) & 3]);
        VIT_G(0)  VIT_G(1)  VIT_G(2)  VIT_G(3)  VIT_G(4)  VIT_G(5)  VIT_G(6)  VIT_G(7)  VIT_G(8)  VIT_G(9)
        VIT_G(10) VIT_G(11) VIT_G(12) VIT_G(13) VIT_G(14) VIT_G(15) VIT_G(16) VIT_G(17) VIT_G(18) VIT_G(19)
#undef VIT_G
        e = _mm_sha1nexte_epu32(prev, e0);
        abcd = _mm_add_epi32(abcd, abcd0);
        p += 64;
    }
    _mm_storeu_si128(reinterpret_cast<__m128i*>(h), _mm_shuffle_epi32(abcd, 0x1B));
    h[4] = uint32_t(_mm_extract_epi32(e, 3));
}

// S independent streams, n blocks each, instruction streams interleaved group
// by group so that one stream's sha1rnds4 issues while another's is in flight.
template <int S>
__attribute__((target("sha,sse4.1,ssse3")))
void blocks_shani_multi(uint32_t* const* hs, const uint8_t** ps, size_t nblocks)
{
    const __m128i bswap = _mm_set_epi64x(0x0001020304050607ll, 0x08090a0b0c0d0e0fll);
    __m128i abcd[S], e[S];
    const uint8_t* p[S];
    for (int s = 0; s < S; ++s) {
        abcd[s] = _mm_shuffle_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i*>(hs[s])), 0x1B);
        e[s] = _mm_set_epi32(int(hs[s][4]), 0, 0, 0);
        p[s] = ps[s];
    }
    while (nblocks--) {
        __m128i abcd0[S], e0[S], m[S][4], prev[S], cur[S];
        for (int s = 0; s < S; ++s) {
            abcd0[s] = abcd[s]; e0[s] = e[s]; prev[s] = abcd[s];
            for (int i = 0; i < 4; ++i)
                m[s][i] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p[s] + 16 * i)), bswap);
        }
#define VIT_GM(g)                                                                                      \
        _Pragma("GCC unroll 4") for (int s = 0; s < S; ++s) {                                          \
            cur[s] = (g) == 0 ? _mm_add_epi32(e[s], m[s][0]) : _mm_sha1nexte_epu32(prev[s], m[s][(g) & 3]); \
            prev[s] = abcd[s];                                                                         \
            abcd[s] = _mm_sha1rnds4_epu32(abcd[s], cur[s], (g) / 5);                                    \
            if ((g) < 16)                                                                              \
                m[s][(g) & 3] = _mm_sha1msg2_epu32(                                                    \
                    _mm_xor_si128(_mm_sha1msg1_epu32(m[s][(g) & 3], m[s][((g) + 1) & 3]), m[s][((g) + 2) & 3]), \
                    m[s][((g) + 3) & 3]);                                                              \
        }
        VIT_GM(0)  VIT_GM(1)  VIT_GM(2)  VIT_GM(3)  VIT_GM(4)  VIT_GM(5)  VIT_GM(6)  VIT_GM(7)  VIT_GM(8)  VIT_GM(9)
        VIT_GM(10) VIT_GM(11) VIT_GM(12) VIT_GM(13) VIT_GM(14) VIT_GM(15) VIT_GM(16) VIT_GM(17) VIT_GM(18) VIT_GM(19)
#undef VIT_GM
        for (int s = 0; s < S; ++s) {
            e[s] = _mm_sha1nexte_epu32(prev[s], e0[s]);
            abcd[s] = _mm_add_epi32(abcd[s], abcd0[s]);
            p[s] += 64;
        }
    }
    for (int s = 0; s < S; ++s) {
        _mm_storeu_si128(reinterpret_cast<__m128i*>(hs[s]), _mm_shuffle_epi32(abcd[s], 0x1B));
        hs[s][4] = uint32_t(_mm_extract_epi32(e[s], 3));
        ps[s] = p[s];
    }
}

bool cpu_has_sha()
{
    static const bool yes = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") &&
                            __builtin_cpu_supports("ssse3");
    return yes;
}
#else
bool cpu_has_sha() { return false; }
#endif

void blocks(uint32_t h[5], const uint8_t* p, size_t n)
{
#if defined(__x86_64__)
    if (cpu_has_sha()) { blocks_shani(h, p, n); return; }
#endif
    blocks_portable(h, p, n);
}

}  // namespace

bool HostSha1::accelerated() { return cpu_has_sha(); }

HostSha1::HostSha1() : nbytes_(0)
{
    h_[0] = 0x67452301u; h_[1] = 0xEFCDAB89u; h_[2] = 0x98BADCFEu; h_[3] = 0x10325476u; h_[4] = 0xC3D2E1F0u;
}

void HostSha1::update(const void* data, size_t len)
{
    if (len == 0) return;                               // (an empty object passes a null pointer)
    const uint8_t* p = static_cast<const uint8_t*>(data);
    size_t have = size_t(nbytes_ & 63);
    nbytes_ += len;
    if (have) {
        size_t take = 64 - have < len ? 64 - have : len;
        std::memcpy(buf_ + have, p, take);
        p += take; len -= take; have += take;
        if (have < 64) return;
        blocks(h_, buf_, 1);
    }
    if (len >= 64) { blocks(h_, p, len / 64); p += len & ~size_t(63); len &= 63; }
    if (len) std::memcpy(buf_, p, len);
}

void HostSha1::finish(uint8_t out[20])
{
    const uint64_t bits = nbytes_ * 8;
    size_t have = size_t(nbytes_ & 63);
    buf_[have++] = 0x80;
    if (have > 56) { std::memset(buf_ + have, 0, 64 - have); blocks(h_, buf_, 1); have = 0; }
    std::memset(buf_ + have, 0, 56 - have);
    for (int i = 0; i < 8; ++i) buf_[56 + i] = uint8_t(bits >> (56 - 8 * i));
    blocks(h_, buf_, 1);
    for (int i = 0; i < 5; ++i) {
        out[4 * i] = uint8_t(h_[i] >> 24); out[4 * i + 1] = uint8_t(h_[i] >> 16);
        out[4 * i + 2] = uint8_t(h_[i] >> 8); out[4 * i + 3] = uint8_t(h_[i]);
    }
}

namespace {
void feed_header(HostSha1& s, const char* type, uint64_t len)
{
    if (!type) return;                                    // nullptr: a raw buffer, no Git framing
    char hdr[40];
    int hl = std::snprintf(hdr, sizeof hdr, "%s %llu", type, static_cast<unsigned long long>(len)) + 1;
    s.update(hdr, size_t(hl));
}
}  // namespace

void host_object_id(const char* type, const uint8_t* content, uint64_t len, uint8_t out[20])
{
    HostSha1 s;
    feed_header(s, type, len);
    s.update(content, size_t(len));
    s.finish(out);
}

void host_object_ids(const char* type, const uint8_t* const* content, const uint64_t* len, size_t count, uint8_t* out)
{
#if defined(__x86_64__)
    const bool interleave = cpu_has_sha();
#else
    const bool interleave = false;
#endif
    for (size_t first = 0; first < count; first += kHostStreams) {
        const int k = int(count - first < size_t(kHostStreams) ? count - first : size_t(kHostStreams));
        if (k == 1 || !interleave) {
            for (int j = 0; j < k; ++j) host_object_id(type, content[first + j], len[first + j], out + (first + j) * 20);
            continue;
        }
#if defined(__x86_64__)
        // Head of every stream through the buffered path: the header and as much
        // content as completes its block.  From there whole blocks come straight
        // from the caller's memory, in lock step over the streams that still have
        // any; what is left at the end (under 64 bytes, and the padding) goes
        // through the buffered path again.
        HostSha1 st[kHostStreams];
        const uint8_t* p[kHostStreams];
        uint64_t blocks_left[kHostStreams], tail[kHostStreams];
        for (int j = 0; j < k; ++j) {
            const uint8_t* c = content[first + j];
            uint64_t n = len[first + j];
            feed_header(st[j], type, n);
            const uint64_t fill = std::min<uint64_t>(n, st[j].to_block_boundary());
            st[j].update(c, size_t(fill));
            c += fill; n -= fill;
            if (st[j].to_block_boundary()) { p[j] = c; blocks_left[j] = 0; tail[j] = 0; continue; }   // shorter than its first block
            p[j] = c;
            blocks_left[j] = n / 64;
            tail[j] = n % 64;
        }
        for (;;) {
            int live[kHostStreams], nl = 0;
            uint64_t step = ~0ull;
            for (int j = 0; j < k; ++j)
                if (blocks_left[j]) { live[nl++] = j; if (blocks_left[j] < step) step = blocks_left[j]; }
            if (nl == 0) break;
            uint32_t* hs[kHostStreams];
            const uint8_t* ps[kHostStreams];
            for (int a = 0; a < nl; ++a) { hs[a] = st[live[a]].state(); ps[a] = p[live[a]]; }
            // size_t steps: a stream of more than 2^32 blocks on a 32-bit size_t is not a case x86-64 has
            switch (nl) {
            case 1: blocks_shani(hs[0], ps[0], size_t(step)); ps[0] += step * 64; break;
            case 2: blocks_shani_multi<2>(hs, ps, size_t(step)); break;
            case 3: blocks_shani_multi<3>(hs, ps, size_t(step)); break;
            default: blocks_shani_multi<4>(hs, ps, size_t(step)); break;
            }
            for (int a = 0; a < nl; ++a) {
                const int j = live[a];
                p[j] = ps[a];
                blocks_left[j] -= step;
                st[j].account_blocks(step);
            }
        }
        for (int j = 0; j < k; ++j) {
            st[j].update(p[j], size_t(tail[j]));
            st[j].finish(out + (first + j) * 20);
        }
#endif
    }
}

}  // namespace vit

/* ---------- C ABI (include/vit_hash.h): the host lane on its own, no device needed ---------- */
extern "C" {

int vit_host_object_id(const char* type, const uint8_t* data, uint64_t len, uint8_t* digest)
{
    if (!vit::type_len_checked(type) || !digest || (len && !data)) return VIT_ERR_ARG;
    vit::host_object_id(type, data, len, digest);
    return VIT_OK;
}

int vit_host_object_ids(const char* type, const uint8_t* const* data, const uint64_t* len, uint64_t n, uint8_t* digests)
{
    if (!vit::type_len_checked(type) || (n && (!data || !len || !digests))) return VIT_ERR_ARG;
    for (uint64_t i = 0; i < n; ++i) if (len[i] && !data[i]) return VIT_ERR_ARG;
    vit::host_object_ids(type, data, len, size_t(n), digests);
    return VIT_OK;
}

}  // extern "C"
