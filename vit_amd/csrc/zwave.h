// zwave.h — the same zlib stream as zdeflate.h writes, worked out by the waves of a GPU workgroup together on ONE
// object whose tables fit the workgroup's share of LDS, instead of by one lane per object with its tables in device memory.
//
// What zlib's default-level deflate (deflate.c: deflate_slow, longest_match) does per input position is a serial
// chain only in its DECISIONS (take the match found at the position before, or wait for this one's).  Everything
// those decisions consult is a pure function of the input:
//   * the hash chains: every position that has three bytes is inserted, in order, whatever matches are chosen, so
//     "the previous position with the same hash" is fixed by the bytes alone;
//   * the longest match at a position, given how many candidates may be looked at (128, or 32 once the position before
//     has a match of 8 or more): the first candidate that reaches the greatest length among them.  The length the
//     position before found (zlib's prev_length) only decides whether that match COUNTS (it must be longer).
// So a workgroup builds the chains 64 positions at a time, then searches a tile of positions at a time, a lane walking
// one position's chain and keeping two answers (best of the first 32 candidates, best of the first 128), and only
// then is zlib's decision loop run over those answers, position by position, without touching a chain.  What the
// scan leaves (the block's symbols and their counts) goes to device memory; the block itself (trees, codes, bits:
// zdeflate.h's flush_block, trees.c restated) is written one object per LANE for a group of objects at a time, because
// building a Huffman tree is a long serial chain of dependent accesses that 64 lanes cannot share but can each do.
//
// Chains in a small table.  zlib keeps head[32768] by the 15-bit hash.  Here a bucket table of about twice the
// object's positions is indexed by the LOW bits of the hash; the 16-bit entry of position p holds the previous position
// in p's BUCKET and, above it, the HIGH bits of p's own hash, so a walk tells a candidate of the same full hash from a
// mere bucket-mate by the entry it reads anyway.  The candidates a walk sees, and their order, are zlib's.  Position 0
// is never a candidate (its link value is zlib's NIL), as in zlib.
//
// Exactness at the end of the input.  zlib stops a search at the first improving candidate of nice_match = min(128,
// bytes left) or more.  With 16 or more bytes left such a candidate always improves (a search only happens after a
// match shorter than 16), so the two precomputed answers are what zlib finds.  In the last 15 positions a candidate can
// reach nice_match without improving; there the decision loop searches in line, with zlib's own loop.
//
// This header holds what host and device share: the per-position functions (a chain walk in steps), the decision step,
// how a symbol is settled, and an emulation of the whole algorithm in plain C++ (deflate_emulated) that tests/zdeflate_driver.cpp runs against
// the zlib library on thousands of inputs.  deflate_kernels.hip is the same flow with wave intrinsics for what the
// emulation does with loops over lanes.
#pragma once
#include "zdeflate.h"

#include <stdint.h>

namespace zw {

constexpr uint32_t kLanes = 64;
constexpr uint32_t kTile = 256;                // positions searched at a time (four waves)
// Largest input: one short of what fills zlib's symbol buffer (16383 symbols end a block in mid-scan; an input of
// fewer bytes cannot have that many), so that the stream is always ONE block, written after the scan.
constexpr uint32_t kMaxInput = zd::kLitBufSize - 2;
constexpr uint32_t kPad = 272;                 // zero bytes kept behind the input (a comparison may read 258 + 3 + 3 past a position)
constexpr uint32_t kNone = zd::kMinMatch - 1;  // "no match", zlib's value for it
constexpr uint32_t kAny = 1u << 31;            // in Found::r128: the position has a chain (zlib's hash_head != NIL)

// A size class: objects of up to nmax bytes, 2^bkbits buckets, positions in posbits bits of an entry and the
// 15 - bkbits high bits of the position's hash above them (16 bits in all, or fewer).
struct Geometry { uint32_t nmax, bkbits, posbits; };
constexpr Geometry kClasses[4] = {{1536, 12, 11}, {4608, 13, 13}, {9216, 14, 14}, {kMaxInput, 15, 14}};
ZD_FN Geometry class_for(uint32_t n) { return n <= 1536 ? kClasses[0] : n <= 4608 ? kClasses[1] : n <= 9216 ? kClasses[2] : kClasses[3]; }

ZD_FN uint32_t ctz32(uint32_t x) { return uint32_t(__builtin_ctz(x)); }

// 4 (2) bytes from byte i of the message, byte i lowest: the two aligned dwords around them and a funnel shift.  (LDS
// serves a load from any byte address, but a wave's unaligned load stalls it for tens of cycles: measured, four fifths
// of the LDS's busy time went there.)
struct Pair { uint32_t lo, hi; };
// "These values are needed here": the compiler would otherwise move a load into the branch that uses its value, and a
// wave would wait for its loads one after the other instead of once for all of them.  (Not `volatile`: on this target a
// volatile load of LDS becomes a system-coherent FLAT load with a wait of its own, several times as slow.)
#if defined(__HIP_DEVICE_COMPILE__)
#define ZW_LOADED_2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define ZW_LOADED_3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define ZW_LOADED_6(a, b, c, d, e, f) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
#else
#define ZW_LOADED_2(a, b) ((void)0)
#define ZW_LOADED_3(a, b, c) ((void)0)
#define ZW_LOADED_6(a, b, c, d, e, f) ((void)0)
#endif
ZD_FN uint32_t from_pair(Pair v, uint32_t i) { return uint32_t((uint64_t(v.hi) << 32 | v.lo) >> ((i & 3u) * 8u)); }
ZD_FN uint32_t ld32(const uint32_t* d, uint32_t i) { return from_pair(Pair{d[i >> 2], d[(i >> 2) + 1]}, i); }
ZD_FN uint32_t ld16(const uint32_t* d, uint32_t i) { return ld32(d, i) & 0xffffu; }
ZD_FN uint32_t byte_at(const uint32_t* d, uint32_t i) { return (d[i >> 2] >> ((i & 3u) * 8u)) & 0xffu; }

// zlib's hash of the three bytes at a position (UPDATE_HASH three times: 15 bits, shift 5), from ld32 there.
ZD_FN uint32_t hash_of(uint32_t w)
{
    return (((w & 0xffu) << 10) ^ (((w >> 8) & 0xffu) << 5) ^ ((w >> 16) & 0xffu)) & uint32_t(zd::kHashMask);
}

// Bytes in common from positions a and b on, at most 258 (zlib compares into the zeros behind the input too).
ZD_FN uint32_t match_len(const uint32_t* d, uint32_t a, uint32_t b)
{
    uint32_t len = 0;
    while (len < uint32_t(zd::kMaxMatch)) {
        const uint32_t x = ld32(d, a + len) ^ ld32(d, b + len);
        if (x) { len += ctz32(x) >> 3; break; }
        len += 4;
    }
    return len > uint32_t(zd::kMaxMatch) ? uint32_t(zd::kMaxMatch) : len;
}

// The chains of one object: entry[p] = previous position in p's bucket | high hash bits of p << posbits.
struct Chain {
    const uint16_t* entry;
    uint32_t bkbits, posbits;
    ZD_FN uint32_t prev(uint32_t e) const { return e & ((1u << posbits) - 1u); }
    ZD_FN uint32_t high(uint32_t e) const { return e >> posbits; }
    ZD_FN uint32_t high_of_hash(uint32_t h) const { return h >> bkbits; }
    ZD_FN uint32_t make(uint32_t prev_pos, uint32_t h) const { return prev_pos | ((h >> bkbits) << posbits); }
};

ZD_FN uint32_t pack(uint32_t len, uint32_t start) { return len | (start << 9); }
ZD_FN uint32_t len_of(uint32_t r) { return r & 511u; }
ZD_FN uint32_t start_of(uint32_t r) { return (r & ~kAny) >> 9; }

struct Found { uint32_t r32, r128; };          // pack(length, start) after at most 32 / 128 candidates; kAny in r128

// longest_match at position p for both chain limits at once, starting from "no match" (see the top of the file), as a
// walk in small steps.  A wave's lanes each walk a chain of their own, of very different lengths, and a step of the
// wave costs what its costliest lane does: so the steps are kept apart (hop to the next candidate worth comparing;
// compare eight bytes more; done), and the caller runs each kind for all the lanes that are due one, and hands a lane
// that is done its next position while the others walk on.
struct Walk {
    enum : uint32_t { kHop, kCompare, kDone };
    uint32_t state = kDone;
    uint32_t p, w, hh, nice;                   // the position, its first bytes, its hash's high bits, zlib's nice_match there
    uint32_t best, start, end2;                // the longest so far, where, and this position's two bytes at best-1, best
    uint32_t cur, count, any, r32;             // next entry to look at, candidates seen, hash_head != NIL, the answer after 32
    uint32_t cand, same;                       // being compared: the candidate, and the bytes found equal so far
};

ZD_FN void walk_begin(Walk& k, const uint32_t* d, uint32_t n, const Chain& c, uint32_t p)
{
    k.p = p; k.best = kNone; k.start = 0; k.count = 0; k.any = 0; k.r32 = 0; k.cur = 0; k.state = Walk::kDone;
    if (p + uint32_t(zd::kMinMatch) > n) return;            // fewer than three bytes left: zlib inserts nothing, hash_head = NIL
    k.w = ld32(d, p); k.hh = c.high_of_hash(hash_of(k.w));
    k.nice = n - p < uint32_t(zd::kNice) ? n - p : uint32_t(zd::kNice);
    k.end2 = ld16(d, p + k.best - 1);
    k.cur = c.prev(c.entry[p]);
    if (k.cur) k.state = Walk::kHop;
}
// One entry of the bucket's chain (state kHop).  Two loads that depend on nothing but the entry's position: the entry,
// and the two bytes zlib tests first (a candidate can only be longer than `best` if its bytes at best-1, best agree).
ZD_FN void walk_hop(Walk& k, const uint32_t* d, const Chain& c)
{
    // (both loads before anything is decided: one wait for the two, not one after the other)
    const uint32_t here = k.cur, at = here + k.best - 1;
    uint32_t e = c.entry[here], lo = d[at >> 2], hi = d[(at >> 2) + 1];
    ZW_LOADED_3(e, lo, hi);
    const uint32_t t = from_pair(Pair{lo, hi}, at) & 0xffffu;
    k.cur = c.prev(e);
    if (c.high(e) == k.hh) {                                // same bucket and same high bits: one of zlib's candidates
        k.any = kAny;
        if (k.count == uint32_t(zd::kChain >> 2)) k.r32 = pack(k.best, k.start) | kAny;   // what 32 candidates gave (kAny: taken)
        ++k.count;
        if (t == k.end2) { k.cand = here; k.same = 0; k.state = Walk::kCompare; return; }
        if (k.count == uint32_t(zd::kChain)) k.cur = 0;
    }
    if (k.cur == 0) k.state = Walk::kDone;
}
// Eight bytes more of the comparison with the candidate (state kCompare); at its end zlib's update of the best match.
ZD_FN void walk_compare(Walk& k, const uint32_t* d)
{
    // eight bytes of each side: three aligned dwords around them
    const uint32_t pa = k.p + k.same, pc = k.cand + k.same;
    const uint32_t* const va = d + (pa >> 2);
    const uint32_t* const vc = d + (pc >> 2);
    uint32_t a0 = va[0], a1 = va[1], a2 = va[2], c0 = vc[0], c1 = vc[1], c2 = vc[2];
    ZW_LOADED_6(a0, a1, a2, c0, c1, c2);
    const uint32_t x0 = from_pair(Pair{a0, a1}, pa) ^ from_pair(Pair{c0, c1}, pc), x1 = from_pair(Pair{a1, a2}, pa) ^ from_pair(Pair{c1, c2}, pc);
    if ((x0 | x1) == 0) { k.same += 8; if (k.same < uint32_t(zd::kMaxMatch)) return; }
    uint32_t len = x0 ? k.same + (ctz32(x0) >> 3) : x1 ? k.same + 4 + (ctz32(x1) >> 3) : k.same;
    if (len > uint32_t(zd::kMaxMatch)) len = uint32_t(zd::kMaxMatch);
    k.state = Walk::kHop;
    if (len > k.best) {                                     // (more than kNone = 2: the first three bytes agree, zlib's own test)
        k.best = len; k.start = k.cand;
        if (len >= k.nice) { k.state = Walk::kDone; return; }
        k.end2 = ld16(d, k.p + k.best - 1);
    }
    if (k.count == uint32_t(zd::kChain) || k.cur == 0) k.state = Walk::kDone;
}
ZD_FN Found walk_result(const Walk& k)
{
    Found f;
    f.r32 = (k.r32 & kAny) ? k.r32 & ~kAny : pack(k.best, k.start);
    f.r128 = pack(k.best, k.start) | k.any;
    return f;
}
ZD_FN Found search_dual(const uint32_t* d, uint32_t n, const Chain& c, uint32_t p)
{
    Walk k;
    walk_begin(k, d, n, c, p);
    while (k.state != Walk::kDone) { if (k.state == Walk::kHop) walk_hop(k, d, c); else walk_compare(k, d); }
    return walk_result(k);
}

// zlib's longest_match itself, for the positions near the end of the input: best_len starts at prev_length, the chain
// limit follows it.  Returns the best length (uncapped); match_start is replaced only by an improvement; any: hash_head != NIL.
ZD_FN uint32_t search_exact(const uint32_t* d, uint32_t n, const Chain& c, uint32_t p, uint32_t prev_length,
                            uint32_t& match_start, bool& any)
{
    any = false;
    uint32_t best = prev_length;
    if (p + uint32_t(zd::kMinMatch) > n) return best;
    const uint32_t hh = c.high_of_hash(hash_of(ld32(d, p)));
    const uint32_t nice = n - p < uint32_t(zd::kNice) ? n - p : uint32_t(zd::kNice);
    uint32_t chain = prev_length >= uint32_t(zd::kGood) ? uint32_t(zd::kChain >> 2) : uint32_t(zd::kChain);
    for (uint32_t cur = c.prev(c.entry[p]); cur != 0;) {
        const uint32_t e = c.entry[cur];
        if (c.high(e) == hh) {
            any = true;
            const uint32_t len = match_len(d, p, cur);
            if (len > best) { match_start = cur; best = len; if (len >= nice) break; }
            if (--chain == 0) break;
        }
        cur = c.prev(e);
    }
    return best;
}

// deflate_slow's state between two positions, and its step at one position.
struct Scan {
    uint32_t strstart = 0, match_length = kNone, prev_length = kNone, match_start = 0, prev_match = 0;
    bool match_available = false;
};
// What a step emits.  code < kLiteral: a match of length lc + 3 at distance `code`; else the literal at position
// code - kLiteral, which the step does not look at: the decisions are the serial part, so they only note where the
// literal is, and settle() reads the bytes and counts the symbols afterwards, many at a time.
struct Symbol { bool emit; uint32_t code, lc; };
constexpr uint32_t kLiteral = 0x8000u;
// One symbol as the scan noted it (d = code, l = lc) into what zlib's buffers hold (distance or 0; length - 3 or the
// literal), and the two codes whose counts it adds to: returns the literal/length code, dcode = the distance code or ~0.
ZD_FN uint32_t settle(const uint32_t* d, uint32_t code, uint32_t lc, uint32_t& dist_out, uint32_t& lc_out, uint32_t& dcode)
{
    if (code >= kLiteral) { dist_out = 0; lc_out = byte_at(d, code - kLiteral); dcode = ~0u; return lc_out; }
    dist_out = code; lc_out = lc; dcode = zd::dist_code_of(code - 1u);
    return zd::length_code_of(lc) + uint32_t(zd::kLiterals) + 1u;
}

// A position at which nothing can happen when no match is pending: no chain, or no candidate of three bytes or more.
// With 16 or more bytes left (the caller's to check) a run of such positions is a run of literals.
ZD_FN bool plain(const Found& f) { return len_of(f.r128) < uint32_t(zd::kMinMatch); }

// The step at s.strstart with 16 or more bytes left: found = search_dual's answer for that position.
ZD_FN Symbol scan_step(Scan& s, const uint32_t* d, uint32_t n, Found found)
{
    const uint32_t lookahead = n - s.strstart;
    s.prev_length = s.match_length; s.prev_match = s.match_start;
    s.match_length = kNone;
    if ((found.r128 & kAny) && s.prev_length < uint32_t(zd::kLazy)) {
        const uint32_t r = s.prev_length >= uint32_t(zd::kGood) ? found.r32 : found.r128;
        uint32_t best = s.prev_length;
        if (len_of(r) > best) { best = len_of(r); s.match_start = start_of(r); }
        s.match_length = best <= lookahead ? best : lookahead;
        if (s.match_length == uint32_t(zd::kMinMatch) && s.strstart - s.match_start > uint32_t(zd::kTooFar)) s.match_length = kNone;
    }
    Symbol out{false, 0, 0};
    if (s.prev_length >= uint32_t(zd::kMinMatch) && s.match_length <= s.prev_length) {
        out = Symbol{true, s.strstart - 1 - s.prev_match, s.prev_length - uint32_t(zd::kMinMatch)};
        s.strstart += s.prev_length - 1;
        s.match_available = false;
        s.match_length = kNone;
    } else if (s.match_available) {
        out = Symbol{true, kLiteral + s.strstart - 1, 0};
        ++s.strstart;
    } else {
        s.match_available = true;
        ++s.strstart;
    }
    return out;
}

// The step at s.strstart with fewer than 16 bytes left: the search is zlib's own, done here (see the top of the file).
ZD_FN Symbol scan_step_tail(Scan& s, const uint32_t* d, uint32_t n, const Chain& c)
{
    const uint32_t lookahead = n - s.strstart;
    s.prev_length = s.match_length; s.prev_match = s.match_start;
    s.match_length = kNone;
    if (lookahead >= uint32_t(zd::kMinMatch) && s.prev_length < uint32_t(zd::kLazy)) {
        bool any = false;
        uint32_t start = s.match_start;
        const uint32_t best = search_exact(d, n, c, s.strstart, s.prev_length, start, any);
        if (any) {
            s.match_start = start;
            s.match_length = best <= lookahead ? best : lookahead;
            if (s.match_length == uint32_t(zd::kMinMatch) && s.strstart - s.match_start > uint32_t(zd::kTooFar)) s.match_length = kNone;
        }
    }
    Symbol out{false, 0, 0};
    if (s.prev_length >= uint32_t(zd::kMinMatch) && s.match_length <= s.prev_length) {
        out = Symbol{true, s.strstart - 1 - s.prev_match, s.prev_length - uint32_t(zd::kMinMatch)};
        s.strstart += s.prev_length - 1;
        s.match_available = false;
        s.match_length = kNone;
    } else if (s.match_available) {
        out = Symbol{true, kLiteral + s.strstart - 1, 0};
        ++s.strstart;
    } else {
        s.match_available = true;
        ++s.strstart;
    }
    return out;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- The whole algorithm in plain C++, lanes as loops: what the kernels do, for the host tests.
// in: the input as zdeflate.h's Deflater reads it (in(i), be32a, be32).  Returns the stream's length, 0 if out_cap was too small.
template <class In>
inline uint32_t deflate_emulated(const zd::Tables* t, In in, uint32_t n, uint8_t* out, uint32_t out_cap)
{
    const Geometry g = class_for(n);
    const uint32_t bk = 1u << g.bkbits;
    uint32_t* d = new uint32_t[(n + kPad + 3) / 4 + 1]();    // the LDS image: the message, zeros behind it
    for (uint32_t i = 0; i < n; ++i) d[i >> 2] |= uint32_t(in(i)) << ((i & 3u) * 8u);
    uint16_t* bucket = new uint16_t[bk]();
    uint16_t* entry = new uint16_t[n + kLanes]();
    uint16_t* d_buf = new uint16_t[n + 1];
    uint8_t* l_buf = new uint8_t[n + 1];
    uint16_t* small = new uint16_t[zd::kSmallEntries]();
    uint32_t* heap = new uint32_t[zd::kHeapEntries]();
    const Chain c{entry, g.bkbits, g.posbits};

    // chains, 64 positions at a time: a lane's link is the nearest lane below it with the same bucket, else the bucket's
    // head from before the chunk; the last lane of a bucket becomes its head
    for (uint32_t base = 0; base + 2 < n; base += kLanes) {
        uint32_t key[kLanes], h[kLanes], old[kLanes];
        bool valid[kLanes];
        for (uint32_t l = 0; l < kLanes; ++l) {
            const uint32_t p = base + l;
            valid[l] = p + 2 < n;
            h[l] = valid[l] ? hash_of(ld32(d, p)) : 0;
            key[l] = h[l] & (bk - 1);
            old[l] = valid[l] ? bucket[key[l]] : 0;
        }
        for (uint32_t l = 0; l < kLanes; ++l) {
            if (!valid[l]) continue;
            uint32_t lk = old[l];
            bool last = true;
            for (uint32_t j = 0; j < kLanes; ++j) {
                if (!valid[j] || key[j] != key[l]) continue;
                if (j < l) lk = base + j;
                if (j > l) last = false;
            }
            entry[base + l] = uint16_t(c.make(lk, h[l]));
            if (last) bucket[key[l]] = uint16_t(base + l);
        }
    }

    zd::Deflater<In>* df = new zd::Deflater<In>();
    df->in = in;
    zd::Work w{{nullptr, 1}, {nullptr, 1}, {d_buf, 1}, {l_buf, 1}, small, heap, 1, out, out_cap};
    df->begin(t, w, n);
    zd::State& st = df->s;
    uint32_t last_lit = 0;
    auto note = [&](uint32_t code, uint32_t lc) { d_buf[last_lit] = uint16_t(code); l_buf[last_lit] = uint8_t(lc); ++last_lit; };

    // searches 256 positions at a time from the scan's position on, then the decisions over them
    Scan sc;
    Found* f = new Found[kTile];
    while (sc.strstart < n) {
        const uint32_t tile = sc.strstart;
        for (uint32_t l = 0; l < kTile; ++l) f[l] = tile + l < n ? search_dual(d, n, c, tile + l) : Found{pack(kNone, 0), pack(kNone, 0)};
        while (sc.strstart < n && sc.strstart < tile + kTile) {
            if (n - sc.strstart < 16u) {                      // the last 15 positions: zlib's own search, one by one
                const Symbol sy = scan_step_tail(sc, d, n, c);
                if (sy.emit) note(sy.code, sy.lc);
                continue;
            }
            // a run of plain positions entered with no match pending: literals, all at once
            uint32_t run = 0;
            if (sc.match_length < uint32_t(zd::kMinMatch))
                while (sc.strstart + run < tile + kTile && sc.strstart + run + 16u <= n && plain(f[sc.strstart + run - tile])) ++run;
            if (run > 1) {
                const uint32_t first = sc.match_available ? 0u : 1u;      // the scan's very first position only opens the window
                for (uint32_t k = first; k < run; ++k) note(kLiteral + sc.strstart + k - 1, 0);
                sc.strstart += run;
                sc.match_available = true;
                sc.prev_length = sc.match_length = kNone;
                continue;
            }
            const Symbol sy = scan_step(sc, d, n, f[sc.strstart - tile]);
            if (sy.emit) note(sy.code, sy.lc);
        }
    }
    if (sc.match_available) note(kLiteral + sc.strstart - 1, 0);
    // the symbols into zlib's form, and their counts (the kernel: all threads, a symbol each)
    for (uint32_t k = 0; k < last_lit; ++k) {
        uint32_t dist, lc, dcode;
        const uint32_t lcode = settle(d, d_buf[k], l_buf[k], dist, lc, dcode);
        d_buf[k] = uint16_t(dist); l_buf[k] = uint8_t(lc);
        st.lt_fc[int(lcode)]++;
        if (dcode != ~0u) st.dt_fc[int(dcode)]++;
    }

    // the block, one lane's work: zdeflate.h's own
    st.last_lit = last_lit;
    df->flush_block(0, n, 1);
    const uint32_t total = df->finish_stream();
    delete df; delete[] f; delete[] d; delete[] bucket; delete[] entry; delete[] d_buf; delete[] l_buf; delete[] small; delete[] heap;
    return total;
}
#endif

}  // namespace zw
