// zdeflate.h — the zlib stream that compress() (zlib 1.2.11, default level) writes for one
// buffer, restated so that a GPU lane can produce it: one object per lane, every table the
// algorithm needs in a caller-supplied workspace, no allocation, no recursion, no library.
//
// The reference stores every object as compress(header + content) (src/commit.cpp:98-105);
// the object store writes the SAME BYTES, so whoever deflates has to make zlib's choices, all
// of them: its hash function and chains (hash of 3 bytes, 15 bits, most recent position first,
// at most 128 links, a quarter of that once a match of 8 is in hand, stop at 128 bytes), lazy
// matching (a match is emitted only if the next position does not find a longer one; none is
// looked for after a match of 16), its rule against 3-byte matches further than 4096 back,
// blocks of 16383 symbols, its Huffman construction (heap order, ties by depth, the length
// limit repair), its run-length code for the code lengths, and its choice between a stored, a
// fixed and a dynamic block.  The zlib library is a third-party dependency of the reference
// (vcpkg.json, unpinned; 1.2.11 on this system): this file follows its published algorithm
// (deflate.c: deflate_slow, longest_match; trees.c) and is pinned by comparing its output with
// the library's, byte for byte, on the host (tests/test_zdeflate.py) and on the GPU.
//
// Scope: inputs of at most 65536 bytes (framed), for which zlib's 64 KiB window never slides;
// longer objects stay with the host library.  Compiles as plain C++ for the host tests and as
// device code under hipcc (ZD_FN).
//
// Which zlib: 1.2.11's deflate.  A host whose libz deflates differently (another version's
// tuning, zlib-ng behind the zlib interface) would give other bytes for the same object; the
// object store compares the first device-made batch with the linked library and leaves the
// work to that library when they differ (object_store.cpp).
//
// This file is written after the algorithm of the zlib library, whose notice follows:
//
//   zlib.h -- interface of the 'zlib' general purpose compression library
//   version 1.2.11, January 15th, 2017
//
//   Copyright (C) 1995-2017 Jean-loup Gailly and Mark Adler
//
//   This software is provided 'as-is', without any express or implied
//   warranty.  In no event will the authors be held liable for any damages
//   arising from the use of this software.
//
//   Permission is granted to anyone to use this software for any purpose,
//   including commercial applications, and to alter it and redistribute it
//   freely, subject to the following restrictions:
//
//   1. The origin of this software must not be misrepresented; you must not
//      claim that you wrote the original software. If you use this software
//      in a product, an acknowledgment in the product documentation would be
//      appreciated but is not required.
//   2. Altered source versions must be plainly marked as such, and must not be
//      misrepresented as being the original software.
//   3. This notice may not be removed or altered from any source distribution.
//
//   Jean-loup Gailly        Mark Adler
//   jloup@gzip.org          madler@alumni.caltech.edu
//
// (This is an altered restatement, not the original software: restructured for one object per
// GPU lane as described above.)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZD_FN __host__ __device__ inline
#else
#define ZD_FN inline
#endif

namespace zd {

constexpr int kMaxInput   = 65536;          // window_size: 2 * 32 KiB
constexpr int kHashSize   = 32768, kHashMask = 32767, kHashShift = 5;
constexpr int kWMask      = 32767;
constexpr int kMinMatch   = 3, kMaxMatch = 258, kMinLookahead = kMaxMatch + kMinMatch + 1;
constexpr int kMaxDist    = 32768 - kMinLookahead;
constexpr int kTooFar     = 4096;
constexpr int kLitBufSize = 16384;
constexpr int kLCodes = 286, kDCodes = 30, kBLCodes = 19, kHeapSize = 2 * kLCodes + 1, kLiterals = 256;
constexpr int kEndBlock = 256, kRep3_6 = 16, kRepz3_10 = 17, kRepz11_138 = 18, kMaxBits = 15, kMaxBLBits = 7;
// level 6: good_length 8, max_lazy 16, nice_length 128, max_chain 128
constexpr int kGood = 8, kLazy = 16, kNice = 128, kChain = 128;
// Candidates of a hash chain a lane looks at in one turn of the scan's loop (see run).  One keeps the lanes of a GPU wave
// busiest; a few cost little waiting and spare the other parts of the loop body some of their turns (measured on
// text-like objects: 4 +9 % over 1; 16 -13 %; a whole chain of 128, as zlib nests it, -51 %).
constexpr int kPerTurn = 4;

// Constant tables (trees.c: tr_static_init), built once on the host by make_tables().
struct Tables {
    uint8_t  length_code[256];              // match length - 3 -> length code
    uint8_t  dist_code[512];                // distance - 1 -> distance code (two halves, see d_code)
    uint16_t base_length[29], base_dist[30];
    uint8_t  extra_lbits[29], extra_dbits[30], extra_blbits[19], bl_order[19];
    uint16_t static_lcode[288]; uint8_t static_llen[288];
    uint16_t static_dcode[30];                                  // every static distance code has 5 bits
};

inline unsigned bi_reverse_host(unsigned code, int len)
{
    unsigned res = 0;
    do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0);
    return res >> 1;
}

inline void make_tables(Tables& t)
{
    static const uint8_t el[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint8_t ed[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    static const uint8_t eb[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
    static const uint8_t bo[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    for (int i = 0; i < 29; ++i) t.extra_lbits[i] = el[i];
    for (int i = 0; i < 30; ++i) t.extra_dbits[i] = ed[i];
    for (int i = 0; i < 19; ++i) { t.extra_blbits[i] = eb[i]; t.bl_order[i] = bo[i]; }
    int length = 0, code;
    for (code = 0; code < 28; code++) {
        t.base_length[code] = uint16_t(length);
        for (int n = 0; n < (1 << el[code]); n++) t.length_code[length++] = uint8_t(code);
    }
    t.base_length[28] = 0;                   // as trees.c leaves it: code 28 (length 258) has no extra bits
    t.length_code[length - 1] = uint8_t(code);
    int dist = 0;
    for (code = 0; code < 16; code++) {
        t.base_dist[code] = uint16_t(dist);
        for (int n = 0; n < (1 << ed[code]); n++) t.dist_code[dist++] = uint8_t(code);
    }
    dist >>= 7;
    for (; code < 30; code++) {
        t.base_dist[code] = uint16_t(dist << 7);
        for (int n = 0; n < (1 << (ed[code] - 7)); n++) t.dist_code[256 + dist++] = uint8_t(code);
    }
    uint16_t bl_count[kMaxBits + 1] = {0};
    int n = 0;
    while (n <= 143) { t.static_llen[n++] = 8; bl_count[8]++; }
    while (n <= 255) { t.static_llen[n++] = 9; bl_count[9]++; }
    while (n <= 279) { t.static_llen[n++] = 7; bl_count[7]++; }
    while (n <= 287) { t.static_llen[n++] = 8; bl_count[8]++; }
    uint16_t next_code[kMaxBits + 1];
    unsigned c = 0;
    for (int bits = 1; bits <= kMaxBits; bits++) { c = (c + bl_count[bits - 1]) << 1; next_code[bits] = uint16_t(c); }
    for (n = 0; n < 288; n++) t.static_lcode[n] = uint16_t(bi_reverse_host(next_code[t.static_llen[n]]++, t.static_llen[n]));
    for (n = 0; n < 30; n++) t.static_dcode[n] = uint16_t(bi_reverse_host(unsigned(n), 5));
}

// Tables::length_code and Tables::dist_code worked out instead of looked up (the scan counts a symbol's codes at every
// match; a table costs it a memory round trip each).  lc = match length - 3; d = distance - 1.
ZD_FN uint32_t length_code_of(uint32_t lc)
{
    if (lc < 8) return lc;
    if (lc == 255) return 28;
    const uint32_t b = 31u - uint32_t(__builtin_clz(lc));
    return 4 * (b - 1) + ((lc >> (b - 2)) & 3u);
}
ZD_FN uint32_t dist_code_of(uint32_t d)
{
    if (d < 4) return d;
    const uint32_t b = 31u - uint32_t(__builtin_clz(d));
    return 2 * b + ((d >> (b - 1)) & 1u);
}

// A small array of 16-bit entries whose elements lie `stride` entries apart: on the GPU the 64
// lanes of a wave interleave theirs (stride 64, lane l starting at entry l), so that lanes
// reading the same index of their own arrays — the common case in the tree code — touch one
// cache line between them; on the host the stride is 1.
struct Arr16 {
    uint16_t* p; uint32_t stride;
    ZD_FN uint16_t& operator[](int i) const { return p[uint32_t(i) * stride]; }
};
// ++a[i] for a counter nothing reads before counts_settled(): on the GPU an atomic add that returns nothing (on the dword
// that holds the entry and a neighbouring lane's), so the lane does not wait for the memory round trip a load, add and
// store would cost it at every symbol.  Counts stay below 2^14 (a block's symbols), so nothing carries between the halves.
ZD_FN void bump(const Arr16& a, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint16_t* const e = &a[i];
    const uintptr_t at = reinterpret_cast<uintptr_t>(e);
    __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(at & ~uintptr_t(3)), (at & 2u) ? 0x10000u : 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
#else
    a[i]++;
#endif
}
// Before the counters bumped so far are read: the adds have been done and this lane's loads see them.
ZD_FN void counts_settled()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
#endif
}
// (An array in two places: entries below `split` at p, the others at p_high.  The heap of the Huffman construction is
// one: the part that is sifted, a long chain of dependent accesses, is its low entries, and a caller may have fast
// memory for those only.)
struct Arr32 {
    uint32_t* p; uint32_t stride;
    uint32_t* p_high = nullptr; uint32_t stride_high = 0; uint32_t split = ~0u;
    ZD_FN uint32_t& operator[](int i) const { return uint32_t(i) < split ? p[uint32_t(i) * stride] : p_high[(uint32_t(i) - split) * stride_high]; }
};
struct Arr8 {
    uint8_t* p; uint32_t stride;
    ZD_FN uint8_t& operator[](int i) const { return p[uint32_t(i) * stride]; }
};

// Entries of the small arrays, in this order, in Work::small.
constexpr int kLtN = kHeapSize, kDtN = 2 * kDCodes + 1, kBtN = 2 * kBLCodes + 1;
constexpr int kSmallEntries = 2 * kLtN + 2 * kDtN + 2 * kBtN + 16 /*bl_count*/ + 16 /*next_code*/;   // 16-bit entries
constexpr int kHeapEntries = kHeapSize;                                                             // 32-bit entries

// Per-object workspace.  head must be all zero on entry (the caller clears it); the other
// arrays need no initialisation.
struct Work {
    Arr16 head;                              // kHashSize entries
    Arr16 prev;                              // 32768 entries
    Arr16 d_buf;                             // kLitBufSize entries: match distances (0 = literal)
    Arr8  l_buf;                             // kLitBufSize entries: literals / match lengths - 3
    uint16_t* small;                         // kSmallEntries entries, `small_stride` apart: the trees and counts
    uint32_t* heap;                          // kHeapEntries entries, `small_stride` apart
    uint32_t  small_stride;
    uint8_t*  out;                           // the stream; capacity out_cap
    uint32_t  out_cap;
    uint32_t  heap_stride = 0;               // of `heap`, if other than small_stride (a caller may keep the heap elsewhere)
    uint32_t* heap_high = nullptr;           // if set: entries from heap_split on are here, small_stride apart (entry heap_split first)
    uint32_t  heap_split = ~0u;
};

constexpr uint32_t work_bytes() { return kHashSize * 2 + 32768 * 2 + kLitBufSize * 2 + kLitBufSize; }
constexpr uint32_t small_bytes() { return kSmallEntries * 2 + kHeapEntries * 4; }
// What compress() itself asks for (compressBound): enough for any input, stored blocks included.
ZD_FN uint32_t bound(uint32_t n) { return n + (n >> 12) + (n >> 14) + (n >> 25) + 13; }

// fc = frequency, later the code; dl = parent, later the bit length (trees.c keeps each pair in a
// union, and the order of reads and writes below relies on it).
struct State {
    const Tables* t;
    Work w;
    uint32_t n;                              // input bytes
    Arr16 lt_fc, lt_dl, dt_fc, dt_dl, bt_fc, bt_dl;   // literal/length tree (573 nodes: 286 leaves + internal), distance, bit-length
    // The heap of build_tree holds, per entry, a node with its sort key: frequency << 16 | depth << 10 | node (a block
    // has at most 16384 symbols and a tree over them at most 23 levels, so 15 + 6 + 10 bits do), which makes a
    // comparison one load per side instead of three (trees.c keeps node numbers and looks Freq and depth up).
    Arr32 heap;
    Arr16 bl_count, next_code;
    int l_max_code, d_max_code;
    int heap_len, heap_max;
    uint32_t opt_len, static_len;
    uint32_t last_lit;
    // bit output
    uint32_t pending;
    uint16_t bi_buf; int bi_valid;
    bool overflow;
};

template <class In> struct Deflater {
    State s;
    In in;                                   // in(i): byte i of the input, i < n

    ZD_FN uint8_t at(uint32_t i) const { return i < s.n ? in(i) : uint8_t(0); }

    // ---- bit output (trees.c: send_bits, bi_windup; deflate.c: put_byte)
    ZD_FN void put_byte(uint32_t b) { if (s.pending < s.w.out_cap) s.w.out[s.pending++] = uint8_t(b); else s.overflow = true; }
    ZD_FN void put_short(uint32_t v) { put_byte(v & 0xff); put_byte((v >> 8) & 0xff); }
    ZD_FN void send_bits(uint32_t value, int length)
    {
        if (s.bi_valid > 16 - length) {
            s.bi_buf = uint16_t(s.bi_buf | (value << s.bi_valid));
            put_short(s.bi_buf);
            s.bi_buf = uint16_t((value & 0xffffu) >> (16 - s.bi_valid));
            s.bi_valid += length - 16;
        } else {
            s.bi_buf = uint16_t(s.bi_buf | (value << s.bi_valid));
            s.bi_valid += length;
        }
    }
    ZD_FN void bi_windup()
    {
        if (s.bi_valid > 8) put_short(s.bi_buf);
        else if (s.bi_valid > 0) put_byte(s.bi_buf & 0xff);
        s.bi_buf = 0; s.bi_valid = 0;
    }
    ZD_FN static unsigned bi_reverse(unsigned code, int len)
    {
        unsigned res = 0;
        do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0);
        return res >> 1;
    }

    // ---- trees.c
    ZD_FN void init_block()
    {
        for (int n = 0; n < kLCodes; n++) s.lt_fc[n] = 0;
        for (int n = 0; n < kDCodes; n++) s.dt_fc[n] = 0;
        for (int n = 0; n < kBLCodes; n++) s.bt_fc[n] = 0;
        s.lt_fc[kEndBlock] = 1;
        s.opt_len = s.static_len = 0;
        s.last_lit = 0;
    }

    // smaller(tree, n, m, depth) of trees.c on packed entries: lower frequency first, ties by lower depth, equal keys either way
    ZD_FN static bool smaller(uint32_t a, uint32_t b) { return (a >> 10) <= (b >> 10); }
    ZD_FN void pqdownheap(int k)
    {
        const uint32_t v = s.heap[k];
        int j = k << 1;
        while (j <= s.heap_len) {
            uint32_t hj = s.heap[j];
            if (j < s.heap_len) { const uint32_t hj1 = s.heap[j + 1]; if (smaller(hj1, hj)) { hj = hj1; j++; } }
            if (smaller(v, hj)) break;
            s.heap[k] = hj; k = j;
            j <<= 1;
        }
        s.heap[k] = v;
    }

    // which: 0 literal/length, 1 distance, 2 bit-length tree
    ZD_FN void gen_bitlen(const Arr16& fc, const Arr16& dl, int max_code, int which)
    {
        const uint8_t* extra = which == 0 ? s.t->extra_lbits : which == 1 ? s.t->extra_dbits : s.t->extra_blbits;
        const int base = which == 0 ? kLiterals + 1 : 0;
        const int max_length = which == 2 ? kMaxBLBits : kMaxBits;
        int h, n, m, bits, xbits, overflow = 0;
        for (bits = 0; bits <= kMaxBits; bits++) s.bl_count[bits] = 0;
        dl[int(s.heap[s.heap_max] & 1023u)] = 0;
        for (h = s.heap_max + 1; h < kHeapSize; h++) {
            n = int(s.heap[h] & 1023u);
            bits = dl[dl[n]] + 1;
            if (bits > max_length) { bits = max_length; overflow++; }
            dl[n] = uint16_t(bits);
            if (n > max_code) continue;
            s.bl_count[bits]++;
            xbits = 0;
            if (n >= base) xbits = extra[n - base];
            const uint32_t f = fc[n];
            s.opt_len += f * uint32_t(bits + xbits);
            if (which == 0) s.static_len += f * uint32_t(s.t->static_llen[n] + xbits);
            else if (which == 1) s.static_len += f * uint32_t(5 + xbits);
        }
        if (overflow == 0) return;
        do {
            bits = max_length - 1;
            while (s.bl_count[bits] == 0) bits--;
            s.bl_count[bits]--;
            s.bl_count[bits + 1] = uint16_t(s.bl_count[bits + 1] + 2);
            s.bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = s.bl_count[bits];
            while (n != 0) {
                m = int(s.heap[--h] & 1023u);
                if (m > max_code) continue;
                if (unsigned(dl[m]) != unsigned(bits)) {
                    s.opt_len += (uint32_t(bits) - uint32_t(dl[m])) * uint32_t(fc[m]);
                    dl[m] = uint16_t(bits);
                }
                n--;
            }
        }
    }

    ZD_FN void gen_codes(const Arr16& fc, const Arr16& dl, int max_code)
    {
        const Arr16& next_code = s.next_code;
        unsigned code = 0;
        for (int bits = 1; bits <= kMaxBits; bits++) {
            code = (code + s.bl_count[bits - 1]) << 1;
            next_code[bits] = uint16_t(code);
        }
        for (int n = 0; n <= max_code; n++) {
            const int len = dl[n];
            if (len == 0) continue;
            { const unsigned c = next_code[len]; next_code[len] = uint16_t(c + 1); fc[n] = uint16_t(bi_reverse(c, len)); }
        }
    }

    ZD_FN int build_tree(const Arr16& fc, const Arr16& dl, int elems, int which)
    {
        int n, m, max_code = -1, node;
        s.heap_len = 0; s.heap_max = kHeapSize;
        for (n = 0; n < elems; n++) {
            const uint32_t f = fc[n];
            if (f != 0) { max_code = n; s.heap[++s.heap_len] = (f << 16) | uint32_t(n); }
            else dl[n] = 0;
        }
        while (s.heap_len < 2) {
            node = (max_code < 2 ? ++max_code : 0);
            s.heap[++s.heap_len] = (1u << 16) | uint32_t(node);
            fc[node] = 1;
            s.opt_len--;
            if (which == 0) s.static_len -= s.t->static_llen[node];
            else if (which == 1) s.static_len -= 5;
        }
        for (n = s.heap_len / 2; n >= 1; n--) pqdownheap(n);
        node = elems;
        do {
            const uint32_t en = s.heap[1];
            s.heap[1] = s.heap[s.heap_len]; s.heap_len--;
            pqdownheap(1);
            const uint32_t em = s.heap[1];
            s.heap[--s.heap_max] = en;
            s.heap[--s.heap_max] = em;
            n = int(en & 1023u); m = int(em & 1023u);
            const uint32_t dn = (en >> 10) & 63u, dm = (em >> 10) & 63u;
            const uint32_t f = (en >> 16) + (em >> 16);
            fc[node] = uint16_t(f);
            dl[n] = dl[m] = uint16_t(node);
            s.heap[1] = (f << 16) | (((dn >= dm ? dn : dm) + 1u) << 10) | uint32_t(node);
            node++;
            pqdownheap(1);
        } while (s.heap_len >= 2);
        { const uint32_t top = s.heap[1]; s.heap[--s.heap_max] = top; }
        gen_bitlen(fc, dl, max_code, which);
        gen_codes(fc, dl, max_code);
        return max_code;
    }

    ZD_FN void scan_tree(const Arr16& dl, int max_code)
    {
        int prevlen = -1, curlen, nextlen = dl[0], count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        dl[max_code + 1] = 0xffff;
        for (int n = 0; n <= max_code; n++) {
            curlen = nextlen; nextlen = dl[n + 1];
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) s.bt_fc[curlen] = uint16_t(s.bt_fc[curlen] + count);
            else if (curlen != 0) { if (curlen != prevlen) s.bt_fc[curlen]++; s.bt_fc[kRep3_6]++; }
            else if (count <= 10) s.bt_fc[kRepz3_10]++;
            else s.bt_fc[kRepz11_138]++;
            count = 0; prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }

    ZD_FN void send_bl(int c) { send_bits(s.bt_fc[c], s.bt_dl[c]); }
    ZD_FN void send_tree(const Arr16& dl, int max_code)
    {
        int prevlen = -1, curlen, nextlen = dl[0], count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        for (int n = 0; n <= max_code; n++) {
            curlen = nextlen; nextlen = dl[n + 1];
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) { do { send_bl(curlen); } while (--count != 0); }
            else if (curlen != 0) {
                if (curlen != prevlen) { send_bl(curlen); count--; }
                send_bl(kRep3_6); send_bits(uint32_t(count - 3), 2);
            } else if (count <= 10) { send_bl(kRepz3_10); send_bits(uint32_t(count - 3), 3); }
            else { send_bl(kRepz11_138); send_bits(uint32_t(count - 11), 7); }
            count = 0; prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }

    ZD_FN int build_bl_tree()
    {
        scan_tree(s.lt_dl, s.l_max_code);
        scan_tree(s.dt_dl, s.d_max_code);
        build_tree(s.bt_fc, s.bt_dl, kBLCodes, 2);
        int max_blindex;
        for (max_blindex = kBLCodes - 1; max_blindex >= 3; max_blindex--)
            if (s.bt_dl[s.t->bl_order[max_blindex]] != 0) break;
        s.opt_len += 3 * (uint32_t(max_blindex) + 1) + 5 + 5 + 4;
        return max_blindex;
    }


    // compress_block: the block's symbols in their codes.  Four symbols at a time: their two buffer entries are loaded
    // together, then the codes of the literals among them, and only then are bits sent, so a lane waits for two memory
    // round trips in four symbols instead of two in each.  A match's extra bits and their count are worked out from its
    // codes (trees.c looks them up in extra_lbits, base_length, extra_dbits and base_dist): from code 8 on a length code
    // has (code >> 2) - 1 extra bits and they are the low bits of length - 3, from code 4 on a distance code has
    // (code >> 1) - 1 and they are the low bits of distance - 1; length code 28 (258) has none.
    ZD_FN void compress_block(bool fixed)
    {
        uint32_t lx = 0;
        while (lx < s.last_lit) {
            const uint32_t g = s.last_lit - lx < 4u ? s.last_lit - lx : 4u;
            unsigned dist[4], lc[4], code[4], bits[4];
            for (uint32_t k = 0; k < 4; ++k) {
                dist[k] = 0; lc[k] = 0; code[k] = 0; bits[k] = 0;
                if (k < g) { dist[k] = s.w.d_buf[int(lx + k)]; lc[k] = s.w.l_buf[int(lx + k)]; }
            }
            for (uint32_t k = 0; k < 4; ++k)
                if (k < g && dist[k] == 0) {
                    code[k] = fixed ? s.t->static_lcode[lc[k]] : s.lt_fc[int(lc[k])];
                    bits[k] = fixed ? s.t->static_llen[lc[k]] : s.lt_dl[int(lc[k])];
                }
            for (uint32_t k = 0; k < g; ++k) {
                if (dist[k] == 0) { send_bits(code[k], int(bits[k])); continue; }
                const unsigned lcode = length_code_of(lc[k]);
                const unsigned sym = lcode + kLiterals + 1;
                if (fixed) send_bits(s.t->static_lcode[sym], s.t->static_llen[sym]); else send_bits(s.lt_fc[int(sym)], s.lt_dl[int(sym)]);
                if (lcode >= 8 && lcode < 28) { const int extra = int(lcode >> 2) - 1; send_bits(lc[k] & ((1u << extra) - 1u), extra); }
                const unsigned d = dist[k] - 1;
                const unsigned dcode = dist_code_of(d);
                if (fixed) send_bits(s.t->static_dcode[dcode], 5); else send_bits(s.dt_fc[int(dcode)], s.dt_dl[int(dcode)]);
                if (dcode >= 4) { const int extra = int(dcode >> 1) - 1; send_bits(d & ((1u << extra) - 1u), extra); }
            }
            lx += g;
        }
        if (fixed) send_bits(s.t->static_lcode[kEndBlock], s.t->static_llen[kEndBlock]);
        else send_bits(s.lt_fc[kEndBlock], s.lt_dl[kEndBlock]);
    }

    // _tr_flush_block in two halves (a caller may want to write a stored block's bytes, or a block's symbols, itself):
    // plan_block builds the three trees from the counts and chooses the block's form, emit_block writes it.
    enum { kStored = 0, kFixed = 1, kDynamic = 2 };
    int max_blindex_;
    ZD_FN int plan_block(uint32_t stored_len)
    {
        s.l_max_code = build_tree(s.lt_fc, s.lt_dl, kLCodes, 0);
        s.d_max_code = build_tree(s.dt_fc, s.dt_dl, kDCodes, 1);
        max_blindex_ = build_bl_tree();
        uint32_t opt_lenb = (s.opt_len + 3 + 7) >> 3;
        const uint32_t static_lenb = (s.static_len + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        if (stored_len + 4 <= opt_lenb) return kStored;
        return static_lenb == opt_lenb ? kFixed : kDynamic;
    }
    // What precedes a block's content: the three header bits; for a stored block the padding to a byte and the two
    // length words; for a dynamic block the description of its trees.
    ZD_FN void emit_block_head(int kind, uint32_t stored_len, int last)
    {
        if (kind == kStored) {
            send_bits(uint32_t((0 << 1) + last), 3);
            bi_windup();
            put_short(stored_len & 0xffff);
            put_short(~stored_len & 0xffff);
        } else if (kind == kFixed) {
            send_bits(uint32_t((1 << 1) + last), 3);
        } else {
            send_bits(uint32_t((2 << 1) + last), 3);
            const int lcodes = s.l_max_code + 1, dcodes = s.d_max_code + 1, blcodes = max_blindex_ + 1;
            send_bits(uint32_t(lcodes - 257), 5);
            send_bits(uint32_t(dcodes - 1), 5);
            send_bits(uint32_t(blcodes - 4), 4);
            for (int rank = 0; rank < blcodes; rank++) send_bits(s.bt_dl[s.t->bl_order[rank]], 3);
            send_tree(s.lt_dl, lcodes - 1);
            send_tree(s.dt_dl, dcodes - 1);
        }
    }
    // _tr_flush_block: the block is input bytes [block_start, block_start + stored_len)
    ZD_FN void flush_block(uint32_t block_start, uint32_t stored_len, int last)
    {
        const int kind = plan_block(stored_len);
        emit_block_head(kind, stored_len, last);
        if (kind == kStored) { for (uint32_t i = 0; i < stored_len; ++i) put_byte(at(block_start + i)); }
        else compress_block(kind == kFixed);
        init_block();
        if (last) bi_windup();
    }

    // ---- deflate.c: deflate_slow with longest_match, INSERT_STRING and _tr_tally in line.
    // Everything the scan touches per input byte is a local here (registers on the GPU): only
    // the tables it must index live in memory.  The State is brought up to date around the
    // block flushes, which happen once in 16383 symbols.
    // The whole stream for input [0, n).  Returns its length, 0 if out_cap was too small.
    // Where everything is: the small arrays carved out of the workspace.  Touches no memory.
    ZD_FN void setup(const Tables* tables, const Work& work, uint32_t n)
    {
        s.t = tables; s.w = work; s.n = n;
        s.pending = 0; s.bi_buf = 0; s.bi_valid = 0; s.overflow = false;
        uint16_t* p = work.small;
        const uint32_t st = work.small_stride;
        auto take = [&](int entries) { Arr16 a{p, st}; p += uint32_t(entries) * st; return a; };
        s.lt_fc = take(kLtN); s.lt_dl = take(kLtN);
        s.dt_fc = take(kDtN); s.dt_dl = take(kDtN);
        s.bt_fc = take(kBtN); s.bt_dl = take(kBtN);
        s.bl_count = take(16); s.next_code = take(16);
        s.heap = Arr32{work.heap, work.heap_stride ? work.heap_stride : st, work.heap_high, st, work.heap_high ? work.heap_split : ~0u};
        s.opt_len = s.static_len = 0;
        s.last_lit = 0;
    }
    // The state before the first input byte: an empty block, the two bytes every stream starts with.
    ZD_FN void begin(const Tables* tables, const Work& work, uint32_t n)
    {
        setup(tables, work, n);
        init_block();
        put_byte(0x78); put_byte(0x9c);                                   // deflate, 32 KiB window, default level
    }

    ZD_FN uint32_t run(const Tables* tables, const Work& work, uint32_t n)
    {
        begin(tables, work, n);

        const In input = in;
        const Arr16 head = work.head, prev = work.prev, d_buf = work.d_buf;
        const Arr8 l_buf = work.l_buf;
        const Arr16 lt_fc = s.lt_fc, dt_fc = s.dt_fc;
        auto byte_at = [&](uint32_t i) -> uint32_t { return i < n ? uint32_t(input(i)) : 0u; };

        uint32_t strstart = 0, lookahead = n, block_start = 0, last_lit = 0;
        uint32_t match_length = kMinMatch - 1, prev_length = kMinMatch - 1, match_start = 0, prev_match = 0;
        bool match_available = false;
        uint32_t ins_h = 0;
        if (n >= 2) ins_h = ((byte_at(0) << kHashShift) ^ byte_at(1)) & kHashMask;

        // The bytes around the scan's position, in registers: input bytes [wpos - 16, wpos), the last in the lowest byte,
        // topped up from whole words of the input.  The hash wants the byte two ahead of every position, a literal the
        // byte one behind, a search the four from the position on.
        unsigned __int128 win = 0;
        uint32_t wpos = 0;
        const uint32_t win_cap = (n + 6u) & ~3u;                        // nothing from byte n + 3 on is ever wanted
        auto top_up = [&](uint32_t upto) {                              // bytes below min(upto, win_cap) are in the window after this
            if (upto > win_cap) upto = win_cap;
            while (wpos < upto) { win = (win << 32) | input.be32a(wpos); wpos += 4; }
        };
        auto win_byte = [&](uint32_t j) -> uint32_t { return uint32_t(win >> (8 * (wpos - 1 - j))) & 0xffu; };   // wpos - 16 <= j < wpos

        // INSERT_STRING for kAhead positions at a time, ahead of the scan.  Which positions enter the hash chains, and in
        // which order, does not depend on the matches chosen (deflate_slow inserts every position that has three bytes
        // left), and a position's chain only ever leads to earlier ones, so the positions [ins_upto, ins_upto + kAhead)
        // are inserted together when the scan reaches the first of them: their hashes from the window, their kAhead
        // heads loaded at once (each a cache miss of its own in a table of 64 KiB per object: together they cost one
        // miss's wait instead of kAhead), positions of the same batch with the same hash chained among themselves, and
        // every position's previous head kept for when the scan gets there.
        constexpr int kAhead = 8;
        uint32_t ins_upto = 0;
        uint64_t heads_lo = 0, heads_hi = 0;                            // previous heads of the batch's positions, 16 bits each
        auto insert_batch = [&]() {
            const uint32_t b = ins_upto;
            top_up(b + uint32_t(kAhead) + 4);
            const uint32_t cnt = n >= b + 3u ? (n - 2u - b < uint32_t(kAhead) ? n - 2u - b : uint32_t(kAhead)) : 0u;
            uint32_t h[kAhead], v[kAhead];
            for (int j = 0; j < kAhead; ++j) {
                h[j] = 0; v[j] = 0;
                if (uint32_t(j) < cnt) { ins_h = ((ins_h << kHashShift) ^ win_byte(b + uint32_t(j) + 2)) & kHashMask; h[j] = ins_h; }
            }
            for (int j = 0; j < kAhead; ++j) if (uint32_t(j) < cnt) v[j] = head[int(h[j])];
            heads_lo = 0; heads_hi = 0;
            for (int j = 0; j < kAhead; ++j) {
                if (uint32_t(j) < cnt) {
                    uint32_t x = v[j];
                    for (int i = 0; i < j; ++i) if (h[i] == h[j]) x = b + uint32_t(i);
                    prev[int((b + uint32_t(j)) & kWMask)] = uint16_t(x);
                    head[int(h[j])] = uint16_t(b + uint32_t(j));
                    if (j < 4) heads_lo |= uint64_t(x) << (16 * j); else heads_hi |= uint64_t(x) << (16 * (j - 4));
                }
            }
            ins_upto = b + uint32_t(kAhead);
        };
        auto head_before = [&](uint32_t str) -> uint32_t {              // the head that INSERT_STRING(str) found; str in the last batch
            const uint32_t j = str - (ins_upto - uint32_t(kAhead));
            return uint32_t((j < 4 ? heads_lo >> (16 * j) : heads_hi >> (16 * (j - 4))) & 0xffffu);
        };
        auto tally = [&](uint32_t dist, uint32_t lc) -> bool {          // _tr_tally
            d_buf[int(last_lit)] = uint16_t(dist);
            l_buf[int(last_lit)] = uint8_t(lc);
            last_lit++;
            if (dist == 0) bump(lt_fc, int(lc));
            else {
                bump(lt_fc, int(length_code_of(lc)) + kLiterals + 1);
                bump(dt_fc, int(dist_code_of(dist - 1)));
            }
            return last_lit == uint32_t(kLitBufSize - 1);
        };
        // 4 input bytes from i on, the first in the top byte; zeros from the end of the input on (zlib's window
        // is zeroed behind the input, and its comparisons may run into that)
        auto word_at = [&](uint32_t i) -> uint32_t {
            if (i >= n) return 0;
            const uint32_t w = input.be32(i);
            const uint32_t over = i + 4 > n ? i + 4 - n : 0;
            return over ? w & (0xFFFFFFFFu << (8 * over)) : w;
        };

        // The scan is deflate_slow with longest_match in line, laid out as ONE loop whose every turn does one unit
        // of work: either the start of a position (INSERT_STRING; positions inside a match that has been emitted do
        // nothing else) or ONE candidate of a hash chain, and then, when the position's search is over or there was
        // none, zlib's choice between match, literal and waiting.  zlib nests three loops here (positions, chain,
        // comparison; and a fourth that inserts the positions of an emitted match).  On a GPU the 64 lanes of a wave
        // run 64 objects through this code together, and nested loops make every lane wait for the longest chain and
        // the longest match among them at every step; in the flat form a lane with a long chain takes more turns and
        // holds nobody up.  The candidate's link, its two bytes at the end of the best match so far and its first four
        // bytes are loaded together, before any of them is looked at.
        // Bytes off and off + 1 behind position base, of which `first` holds the first four: no load while they lie
        // within those four (a search starts from a best length of 2, where most candidates are turned down).
        auto pair_at = [&](uint32_t first, uint32_t base, uint32_t off) -> uint32_t {
            return off <= 2 ? (first << (8 * off)) >> 16 : word_at(base + off) >> 16;
        };

        // zlib flushes a block from inside the scan when the symbol buffer is full.  Here the scan stops instead, the one
        // flush below is done and the scan is entered again where it stopped: the block code is large, and in line at
        // every place that may flush it is three times the rest of the kernel.
        bool searching = false;
        uint32_t cur = 0, chain = 0, limit = 0, best_len = 0, nice = 0;  // longest_match's state
        uint32_t scan_head = 0, scan_end2 = 0;                           // bytes 0..3 at strstart; bytes best_len-1, best_len
        for (;;) {
        bool full = false;
        uint32_t flush_at = 0;
        while ((strstart < n || searching) && !full) {
            bool decide = false;
            lookahead = n - strstart;
            if (!searching) {
                if (strstart >= ins_upto) {                  // after a long match: one batch a turn until the scan's position is in
                    insert_batch();
                    if (strstart >= ins_upto) continue;
                }
                const uint32_t hash_head = lookahead >= uint32_t(kMinMatch) ? head_before(strstart) : 0u;
                prev_length = match_length; prev_match = match_start;
                match_length = kMinMatch - 1;
                if (hash_head != 0 && prev_length < uint32_t(kLazy) && strstart - hash_head <= uint32_t(kMaxDist)) {
                    searching = true;
                    cur = hash_head;
                    chain = prev_length >= uint32_t(kGood) ? kChain >> 2 : kChain;
                    best_len = prev_length;
                    nice = uint32_t(kNice) > lookahead ? lookahead : uint32_t(kNice);
                    limit = strstart > uint32_t(kMaxDist) ? strstart - uint32_t(kMaxDist) : 0;
                    scan_head = strstart + 4 <= n ? uint32_t(win >> (8 * (wpos - 4 - strstart))) : word_at(strstart);
                    scan_end2 = pair_at(scan_head, strstart, best_len - 1);
                } else decide = true;
            }
            if (searching) {
              bool over = false;
              for (int rep = 0; rep < kPerTurn && !over; ++rep) {
                const uint32_t next = prev[int(cur & kWMask)];
                const uint32_t m_head = word_at(cur);
                const uint32_t m_end2 = pair_at(m_head, cur, best_len - 1);
                uint32_t x = m_head ^ scan_head;
                if (m_end2 == scan_end2 && (x >> 16) == 0) {
                    // bytes 0 and 1 match, and byte 2 with them (same hash of three bytes, wide enough to tell);
                    // zlib resumes at byte 3, which comes to the same length
                    uint32_t len = 4;
                    if (x) len = uint32_t(__builtin_clz(x)) >> 3;
                    else {
                        while (len < uint32_t(kMaxMatch)) {
                            x = word_at(strstart + len) ^ word_at(cur + len);
                            if (x) { len += uint32_t(__builtin_clz(x)) >> 3; break; }
                            len += 4;
                        }
                        if (len > uint32_t(kMaxMatch)) len = kMaxMatch;
                    }
                    if (len > best_len) {
                        match_start = cur;
                        best_len = len;
                        if (len >= nice) over = true;
                        else scan_end2 = pair_at(scan_head, strstart, best_len - 1);
                    }
                }
                if (!over) {
                    cur = next;
                    if (cur <= limit || --chain == 0) over = true;
                }
              }
                if (over) {
                    searching = false;
                    match_length = best_len <= lookahead ? best_len : lookahead;
                    if (match_length == uint32_t(kMinMatch) && strstart - match_start > uint32_t(kTooFar))
                        match_length = kMinMatch - 1;
                    decide = true;
                }
            }
            if (decide) {
                if (prev_length >= uint32_t(kMinMatch) && match_length <= prev_length) {
                    // the match found at the position before stands; its remaining positions are passed over (the batches
                    // insert them; zlib inserts them here and flushes after them: the flush reads neither table)
                    const bool bflush = tally(strstart - 1 - prev_match, prev_length - kMinMatch);
                    const uint32_t next_start = strstart + prev_length - 1;
                    match_available = false;
                    match_length = kMinMatch - 1;
                    if (bflush) { full = true; flush_at = next_start; }
                    strstart = next_start;
                } else if (match_available) {
                    const bool bflush = tally(0, win_byte(strstart - 1));
                    if (bflush) { full = true; flush_at = strstart; }
                    ++strstart;
                } else {
                    match_available = true;
                    ++strstart;
                }
            }
        }
        lookahead = n - strstart;
        const int last = full ? 0 : 1;
        if (last) {
            if (match_available) tally(0, byte_at(strstart - 1));
            flush_at = strstart;
        }
        s.last_lit = last_lit;
        counts_settled();
        flush_block(block_start, flush_at - block_start, last);
        last_lit = 0;
        block_start = flush_at;
        if (last) break;
        }

        return finish_stream();
    }

    // Behind the last block: adler32 of the input, big-endian (the input four bytes to a load).  Returns the stream's
    // length, 0 if out_cap was too small.
    ZD_FN uint32_t finish_stream()
    {
        const In input = in;
        const uint32_t n = s.n;
        uint32_t a = 1, b = 0;
        for (uint32_t i = 0; i < n;) {
            const uint32_t end = i + 5552 < n ? i + 5552 : n;           // 5552 is a multiple of four: i stays one
            for (; i + 4 <= end; i += 4) {
                const uint32_t w = input.be32a(i);
                a += w >> 24; b += a; a += (w >> 16) & 0xffu; b += a; a += (w >> 8) & 0xffu; b += a; a += w & 0xffu; b += a;
            }
            for (; i < end; ++i) { a += input(i); b += a; }
            a %= 65521u; b %= 65521u;
        }
        put_byte(b >> 8); put_byte(b & 0xff); put_byte(a >> 8); put_byte(a & 0xff);
        return s.overflow ? 0 : s.pending;
    }
};

}  // namespace zd
