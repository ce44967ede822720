// sha1_device.h — SHA-1 compression function for one gfx950 lane.
//
// One lane owns one message.  The 16-word block lives in VGPRs and doubles as
// the rolling message schedule (w[t & 15] is overwritten in place), so a block
// costs no memory traffic beyond its 64 input bytes.  Instruction selection is
// steered to the 3-input ops gfx950 has:
//   v_bitop3_b32   any 3-input boolean: Ch (0xCA), Parity (0x96), Maj (0xE8),
//                  and the 3-way xor inside the schedule
//   v_alignbit_b32 rotates
//   v_add3_u32     the two 3-way adds of a round
// which gives 597 vector ops per 64-byte block:
//   schedule 64 x (bitop3 + xor + alignbit)                       = 192
//   rounds   80 x (alignbit + bitop3 + 2 x add3 + alignbit)       = 400
//   5 chaining adds                                               =   5
// The 16 byte swaps a little-endian machine owes SHA-1's big-endian words are
// paid once by whoever packs the arena (the host packer as it copies, the
// device generator as it writes): the arena holds every message word as a
// native dword, so a loaded register IS the word.
//
// Replaces the OpenSSL SHA1() call of the reference (src/commit.cpp:93-94);
// the algorithm is FIPS 180-4 section 6.1.2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vit {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_rotateleft32(x, n); }
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

#define VIT_CH(b, c, d)  __builtin_amdgcn_bitop3_b32((b), (c), (d), 0xCA)   /* b ? c : d          */
#define VIT_PAR(b, c, d) __builtin_amdgcn_bitop3_b32((b), (c), (d), 0x96)   /* b ^ c ^ d          */
#define VIT_MAJ(b, c, d) __builtin_amdgcn_bitop3_b32((b), (c), (d), 0xE8)   /* majority           */

#define VIT_W(t) w[(t) & 15]
#define VIT_SCHED(t)                                                                       \
    (VIT_W(t) = rotl32(VIT_PAR(VIT_W((t) - 3), VIT_W((t) - 8), VIT_W((t) - 14)) ^ VIT_W(t), 1))

#define VIT_ROUND(a, b, c, d, e, F, K, WT)                 \
    {                                                      \
        e += rotl32(a, 5) + F(b, c, d) + ((K) + (WT));     \
        b = rotl32(b, 30);                                 \
    }
#define VIT_ROUND5(F, K, t, X)                                                        \
    VIT_ROUND(a, b, c, d, e, F, K, X(t))     VIT_ROUND(e, a, b, c, d, F, K, X((t) + 1)) \
    VIT_ROUND(d, e, a, b, c, F, K, X((t) + 2)) VIT_ROUND(c, d, e, a, b, F, K, X((t) + 3)) \
    VIT_ROUND(b, c, d, e, a, F, K, X((t) + 4))

// h: chaining value; w: the block's 16 message words (destroyed).
__device__ __forceinline__ void sha1_compress(uint32_t (&h)[5], uint32_t (&w)[16])
{
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    VIT_ROUND5(VIT_CH, 0x5A827999u, 0, VIT_W)
    VIT_ROUND5(VIT_CH, 0x5A827999u, 5, VIT_W)
    VIT_ROUND5(VIT_CH, 0x5A827999u, 10, VIT_W)
    VIT_ROUND(a, b, c, d, e, VIT_CH, 0x5A827999u, VIT_W(15))
    VIT_ROUND(e, a, b, c, d, VIT_CH, 0x5A827999u, VIT_SCHED(16))
    VIT_ROUND(d, e, a, b, c, VIT_CH, 0x5A827999u, VIT_SCHED(17))
    VIT_ROUND(c, d, e, a, b, VIT_CH, 0x5A827999u, VIT_SCHED(18))
    VIT_ROUND(b, c, d, e, a, VIT_CH, 0x5A827999u, VIT_SCHED(19))
    VIT_ROUND5(VIT_PAR, 0x6ED9EBA1u, 20, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0x6ED9EBA1u, 25, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0x6ED9EBA1u, 30, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0x6ED9EBA1u, 35, VIT_SCHED)
    VIT_ROUND5(VIT_MAJ, 0x8F1BBCDCu, 40, VIT_SCHED)
    VIT_ROUND5(VIT_MAJ, 0x8F1BBCDCu, 45, VIT_SCHED)
    VIT_ROUND5(VIT_MAJ, 0x8F1BBCDCu, 50, VIT_SCHED)
    VIT_ROUND5(VIT_MAJ, 0x8F1BBCDCu, 55, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0xCA62C1D6u, 60, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0xCA62C1D6u, 65, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0xCA62C1D6u, 70, VIT_SCHED)
    VIT_ROUND5(VIT_PAR, 0xCA62C1D6u, 75, VIT_SCHED)
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

__device__ __forceinline__ void sha1_init(uint32_t (&h)[5])
{
    h[0] = 0x67452301u; h[1] = 0xEFCDAB89u; h[2] = 0x98BADCFEu;
    h[3] = 0x10325476u; h[4] = 0xC3D2E1F0u;
}

// 16 arena bytes = 4 message words, already in register order.
__device__ __forceinline__ void put_words(uint32_t (&w)[16], int at, uint4 v)
{
    w[at] = v.x; w[at + 1] = v.y; w[at + 2] = v.z; w[at + 3] = v.w;
}

// Digest bytes are the chaining words in big-endian order; 20 B per object,
// 4-byte aligned, so five dword stores.
__device__ __forceinline__ void store_digest(uint8_t* digests, uint64_t index, const uint32_t (&h)[5])
{
    uint32_t* out = reinterpret_cast<uint32_t*>(digests + index * 20u);
#pragma unroll
    for (int i = 0; i < 5; ++i) out[i] = bswap32(h[i]);
}

}  // namespace vit
