// deflate_kernels.hip — the loose-object file contents, made where the objects already are.
//
// The reference stores every object as compress(header + content) (src/commit.cpp:98-105), and
// that call is three quarters of its write-tree.  The arena the hasher stages in device memory
// holds exactly header + content; the kernels here write the zlib stream for every object of it,
// with zlib's own choices (zdeflate.h, zwave.h), so the host is left with creating files.
//
// Two kernels.  Objects of up to 16382 framed bytes (one deflate block), which is nearly all of a
// source tree, are deflated by one workgroup each with every table in LDS (second half of the
// file).  The longer ones, up to 65536 bytes, by one LANE each (first half): deflate is a serial
// scan with a hash table, as SHA-1 is a serial chain.  The tables a lane needs (zlib's head and
// prev arrays, the symbol buffers of a block, the Huffman construction's arrays: 181 KiB) live in
// device memory, one workspace per wave with the 64 lanes' entries interleaved.  A wave takes 64
// objects at a time and moves on to its next 64 when the longest of them is done.  That kernel is
// bound by scattered two-byte accesses to those tables (the scan, and the Huffman heap of every
// block), i.e. by the memory system's transaction rate, not by arithmetic: every lane's object,
// tables and position are its own, so a wave's memory instruction touches up to 64 cache lines.
// Incompressible content runs at 8-11 us per input byte and lane, text-like content, whose
// positions have hash chains to walk, at about four times that.
#include "kernels.h"
#include "zdeflate.h"
#include "zwave.h"

namespace vit {

namespace {

// Byte i of the framed message of one object: the arena stores every 4-byte group reversed.
struct ArenaBytes {
    const uint8_t* p;
    __host__ __device__ uint8_t operator()(uint32_t i) const { return p[i ^ 3u]; }
    // Bytes i..i+3 of the message, the first in the top byte: the arena's dwords are the message's big-endian words, so
    // one load of two dwords and a shift.  Reads up to byte (i | 3) + 4 of the slot: within the padding for every i < n.
    // The same from a multiple of four: the arena's dword itself.
    __host__ __device__ uint32_t be32a(uint32_t i) const { return reinterpret_cast<const uint32_t*>(p)[i >> 2]; }
    __host__ __device__ uint32_t be32(uint32_t i) const
    {
        struct __attribute__((packed, aligned(4))) Pair { uint64_t v; };          // one 8-byte load on a 4-byte boundary
        const uint64_t v = reinterpret_cast<const Pair*>(p + (i & ~3u))->v;
        const uint64_t both = v << 32 | v >> 32;                                  // the earlier dword on top
        return (uint32_t)((both << ((i & 3u) * 8u)) >> 32);
    }
};

}  // namespace

// (The register allocator takes every register it may for this much inlined code: one wave per SIMD, 65536 lanes
// on the chip, whatever occupancy is asked for; more lanes would only wait on the same memory system.)
__global__ __launch_bounds__(64) void deflate_kernel(DeflateBatch b, uint32_t only_above)
{
    const uint32_t lane = threadIdx.x;
    // One workspace per wave; in every table of it the 64 lanes' entries are interleaved (entry e of
    // lane l at e * 64 + l).  Objects of similar length advance through their tables in step, so a
    // wave's 64 accesses to the sequential ones (prev, the symbol buffers, the tree arrays) fall into
    // one or two cache lines; laid out lane by lane, the 65536 lanes' partial lines thrash the L2.
    constexpr uint64_t kWaveBytes = 64ull * (zd::work_bytes() + zd::small_bytes());
    uint8_t* const ws = b.workspace + (uint64_t)blockIdx.x * kWaveBytes;
    uint16_t* const head16 = reinterpret_cast<uint16_t*>(ws);
    uint16_t* const prev16 = head16 + 64u * zd::kHashSize;
    uint16_t* const dbuf16 = prev16 + 64u * 32768u;
    uint8_t*  const lbuf8  = reinterpret_cast<uint8_t*>(dbuf16 + 64u * zd::kLitBufSize);
    uint16_t* const small16 = reinterpret_cast<uint16_t*>(lbuf8 + 64u * zd::kLitBufSize);
    zd::Work w;
    w.head  = zd::Arr16{head16 + lane, 64};
    w.prev  = zd::Arr16{prev16 + lane, 64};
    w.d_buf = zd::Arr16{dbuf16 + lane, 64};
    w.l_buf = zd::Arr8{lbuf8 + lane, 64};
    w.small = small16 + lane;
    w.heap = reinterpret_cast<uint32_t*>(small16 + 64u * zd::kSmallEntries) + lane;
    w.small_stride = 64;
    zd::Deflater<ArenaBytes> d;

    for (uint64_t first = (uint64_t)blockIdx.x * 64u; first < b.n; first += (uint64_t)gridDim.x * 64u) {
        uint64_t i = first + lane;
        bool mine = i < b.n;
        if (mine && b.order) { i = b.order[i]; if (i >= b.n) { atomicOr(b.status, STATUS_BAD_ORDER); mine = false; } }
        uint64_t off = 0, n = 0;
        uint32_t nb = 0;
        if (mine) {
            off = b.offsets[i];
            nb = b.nblocks[i];
            uint32_t bad = 0;
            if (off & 15u) bad |= STATUS_MISALIGNED;
            if (off > b.arena_bytes || (uint64_t)nb * 64u > b.arena_bytes - off) bad |= STATUS_OUT_OF_BOUNDS;
            if (bad) { atomicOr(b.status, bad); b.out_len[i] = 0; mine = false; }
            else if (nb == 0) { b.out_len[i] = 0; mine = false; }
            else if ((uint64_t)nb * 64u <= only_above) mine = false;      // a wave kernel's, whatever its padding says (framed + 9 <= 64 nb)
        }
        if (mine) {
            // the framed length is in the padding: the last two dwords of the last block hold it in bits
            const uint32_t* last = reinterpret_cast<const uint32_t*>(b.arena + off + (uint64_t)(nb - 1u) * 64u);
            n = (((uint64_t)last[14] << 32) | last[15]) >> 3;
            if (n > (uint64_t)zd::kMaxInput || n + 9u > (uint64_t)nb * 64u) { b.out_len[i] = 0; mine = false; }   // the host's share
            else if (n <= only_above || n > b.max_input) mine = false;      // a wave kernel's object, or the host's
        }
        if (__ballot(mine) == 0) continue;                // none of the 64 is this kernel's: the tables stay as they are (cleared or unused)
        // zlib starts from an empty hash table: the wave clears its 64 interleaved tables together
        uint4* const hd = reinterpret_cast<uint4*>(head16);
        for (uint32_t k = lane; k < 64u * zd::kHashSize * 2u / 16u; k += 64u) hd[k] = make_uint4(0, 0, 0, 0);
        __syncthreads();                                  // one wave: orders the lanes' stores before each other's loads

        if (mine) {
            w.out = b.out + b.out_off[i];
            w.out_cap = zd::bound((uint32_t)n);
            d.in = ArenaBytes{b.arena + off};
            b.out_len[i] = d.run(static_cast<const zd::Tables*>(b.tables_raw), w, (uint32_t)n);
        }
        __syncthreads();                                  // the tables are cleared again only when every lane is done with them
    }
}


// ---- One object per WORKGROUP of four waves, its tables in LDS (zwave.h) ---------------------------------------------
//
// The lane kernel above keeps 181 KiB of tables per object in device memory and is bound by their scattered traffic
// (277 bytes over the fabric per input byte, waves issuing in 7 % of their cycles).  For objects of up to 16382 bytes
// (one deflate block) the tables shrink to a few times the object (bucket table, chain entries, the block's symbols)
// and fit a workgroup's share of LDS.  Per object: the message is loaded into LDS (and its checksum summed, 64 bytes
// a thread); one wave, the "serial" one, builds the hash chains 64 positions at a time; then, 512 positions at a
// time, the lanes of the other three waves walk the chains (a lane takes the next position nobody has, stores the
// answer, takes another) while the serial wave runs zlib's decisions over the 512 positions before, in scalar
// registers, and joins the walkers when it is through; the block's symbols are settled (a literal's byte read, the
// codes counted) by all threads and go, with the counts, to a slot of the workgroup's scratch in device memory.
// After kGroup objects the first kGroup lanes of the serial wave write one object's block each (trees, codes, bits:
// zdeflate.h's own code, the stream straight to its place): a Huffman tree is a serial chain of dependent accesses
// that lanes cannot share but can each have one of.  A block that zlib stores is copied by all threads.  Device
// memory sees the object once, its symbols out and in, the tree arrays, and its stream once: 10 (text-like) to 20
// (incompressible) bytes over the fabric per input byte.  Four size classes (LDS per workgroup, workgroups per CU):
// up to 1536 bytes (37 KiB, 4), 4608 (39 KiB, 4), 9216 (76 KiB, 2), 16382 (145 KiB, 1); every class is one launch
// over the batch, each workgroup taking the objects of its class.
// What bounds it (text-like 4 KiB objects, 1.5 M/s): a tile is done when its longest chain is, entry by entry at
// some 900 clocks an entry in a wave that shares its SIMD with three others; the VALUs are busy 40 % of the time.
constexpr uint32_t kGroup = 32;                              // objects a workgroup scans before it writes their blocks
constexpr uint32_t kHeapInLds = zd::kLCodes + 2;            // entries of a Huffman heap kept in LDS: all that are ever sifted (at most one per code)
constexpr uint32_t kRefill = 16;                            // searching lanes that must be free before the wave stops to hand out positions
constexpr uint32_t kHopsInARow = 2;                         // chain entries a lane looks at between two comparisons
constexpr uint32_t kHopsAtTheEnd = 8;                       // ... once the tile's positions are all handed out

// A class's workgroup: threads, and positions per tile.  The two small classes fit a CU four times with four waves each;
// of the large ones a CU holds two or one, which then have eight and sixteen waves and longer tiles to keep them busy.
template <int NMAX> struct WaveClass {
    static constexpr int kThreads = NMAX <= 4608 ? 256 : NMAX <= 9216 ? 512 : 1024;
    static constexpr uint32_t kTile = NMAX <= 4608 ? 512 : NMAX <= 9216 ? 1024 : 2048;
    static constexpr int kWavesPerSimd = 4;                 // 128 registers a thread in every class
};

template <int NMAX, int BKBITS> struct WaveShared {
    static constexpr uint32_t kScanTile = WaveClass<NMAX>::kTile;
    uint32_t data[(NMAX + zw::kPad + 8) / 4];               // the framed message, byte i at byte i, zeros behind it
    union {
        uint16_t bucket[1 << BKBITS];                       // while the chains are built
        struct { uint16_t d_buf[NMAX + 2]; uint8_t l_buf[NMAX + 2]; } sym;   // then the block's symbols
    } a;
    uint16_t entry[NMAX + 64];                              // the chains (zw::Chain)
    uint32_t spare[(NMAX <= 1536 ? 15616 : 4) / 4];         // (the smallest class has LDS to spare: room for the Huffman heaps of all 32 lanes that build trees)
    zw::Found found[2][kScanTile];                          // the searches' answers for the positions of two tiles
    uint32_t grab[2];                                       // per tile: how many of its positions have been handed out
    uint16_t lt_fc[zd::kLCodes + 2], dt_fc[zd::kDCodes + 2];   // the symbols' counts
    unsigned long long trace[14];                           // diagnostics (DeflateBatch::trace), here and not in registers: the loops need those
    uint32_t n_sym;                                         // the scan's symbols, for the waves that do not run it
    uint32_t elect;                                         // which wave runs the serial parts
    uint32_t sum_a;                                         // the checksum's two sums before their reduction
    unsigned long long sum_b;
};

// A workgroup's scratch in device memory: kGroup slots of symbols, the kGroup objects' tree arrays interleaved
// (entry e of slot s at e * kGroup + s: the lanes that build the trees read the same entry of their own arrays
// together; the construction's heap is in LDS), and a few words per slot.
template <int NMAX> struct WaveScratch {
    static constexpr uint64_t kSymBytes = (3ull * (NMAX + 2) + 15) / 16 * 16;
    static constexpr uint64_t kSmallBytes = 2ull * zd::kSmallEntries * kGroup;
    static constexpr uint64_t kHeapBytes = 4ull * (zd::kHeapEntries - kHeapInLds) * kGroup;   // the heap's entries that are not in LDS
    static constexpr uint32_t kMetaWords = 8;                // per slot: object (2), length, symbols, checksum, and what begin_block_by_lane leaves (3)
    static constexpr uint64_t kMetaBytes = 4ull * kMetaWords * kGroup;
    static constexpr uint64_t kBytes = (kGroup * kSymBytes + kSmallBytes + kHeapBytes + kMetaBytes + 255) / 256 * 256;
};

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)uniform(lane)); }
__device__ __forceinline__ void lds_settled() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }   // within a wave: earlier LDS writes before later reads

// ++ of a 16-bit counter in LDS by any number of lanes at once (the counters lie in pairs in a dword).
__device__ __forceinline__ void bump_lds(uint16_t* e)
{
    const uintptr_t at = reinterpret_cast<uintptr_t>(e);
    __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(at & ~uintptr_t(3)), (at & 2u) ? 0x10000u : 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One object's block as far as one lane must make it: the trees from the symbols' counts and the block's head
// (zdeflate.h's code as it is).  The heap of the Huffman construction, its one array that is read and written at
// every step of a long dependent chain, is the caller's to place (its low entries, the part that is sifted, in LDS, the
// rest with the tree arrays in the scratch).  What
// follows the head is left to the workgroup: meta[5] = where a stored block's bytes go (else 0), meta[6] = bytes of the
// stream so far, meta[7] = the bits not yet in them (value, count << 16) and the block's kind << 24.  Not inlined: its
// registers are its own, the scan's loops keep theirs.
template <int NMAX>
__device__ __attribute__((noinline)) void begin_block_by_lane(const DeflateBatch& b, const zd::Tables* tables, uint8_t* sym, uint16_t* small,
                                                              uint32_t* heap, uint32_t heap_stride, uint32_t* heap_high, uint32_t* meta)
{
    const uint64_t i = ((uint64_t)meta[1] << 32) | meta[0];
    const uint32_t n = meta[2];
    zd::Deflater<ArenaBytes> df;
    df.in = ArenaBytes{b.arena + b.offsets[i]};
    zd::Work w;
    w.head = zd::Arr16{nullptr, 1}; w.prev = zd::Arr16{nullptr, 1};
    w.d_buf = zd::Arr16{reinterpret_cast<uint16_t*>(sym), 1};
    w.l_buf = zd::Arr8{sym + 2u * (NMAX + 2), 1};
    w.small = small; w.heap = heap; w.small_stride = kGroup; w.heap_stride = heap_stride; w.heap_high = heap_high; w.heap_split = kHeapInLds;
    w.out = b.out + b.out_off[i];
    w.out_cap = zd::bound(n);
    df.setup(tables, w, n);
    df.put_byte(0x78); df.put_byte(0x9c);
    df.s.last_lit = meta[3];
    const int kind = df.plan_block(n);
    df.emit_block_head(kind, n, 1);
    meta[5] = kind == zd::Deflater<ArenaBytes>::kStored ? df.s.pending : 0u;
    meta[6] = df.s.overflow ? ~0u : df.s.pending;
    meta[7] = uint32_t(df.s.bi_buf) | uint32_t(df.s.bi_valid) << 16 | uint32_t(kind) << 24;
}

// What the workgroup keeps in LDS while it writes a block's content (in place of everything the scan kept there).
template <int NMAX, int THREADS> struct EmitShared {
    uint32_t lit[zd::kLCodes + 2];                          // literal/length codes: code | bits << 16
    uint32_t dist[zd::kDCodes + 2];                         // distance codes likewise
    uint32_t waves[THREADS / 64 + 1];                       // bits of each wave's symbols, for the running sum
    uint32_t out[(NMAX + 64) / 4 + 4];                      // the block's content from the stream's current byte on
};

template <int NMAX, int BKBITS, int POSBITS>
__global__ __launch_bounds__(WaveClass<NMAX>::kThreads, WaveClass<NMAX>::kWavesPerSimd) void deflate_wave_kernel(DeflateBatch b, uint32_t n_above)
{
    constexpr uint32_t kThreads = WaveClass<NMAX>::kThreads, kScanTile = WaveClass<NMAX>::kTile;
    using Shared = WaveShared<NMAX, BKBITS>;
    __shared__ Shared sh;
    using Scratch = WaveScratch<NMAX>;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    bool laps = false;                                      // this wave's phases are the ones traced
    const zd::Tables* const tables = static_cast<const zd::Tables*>(b.tables_raw);
    uint8_t* const scratch = b.workspace + (uint64_t)blockIdx.x * Scratch::kBytes;
    uint16_t* const g_small = reinterpret_cast<uint16_t*>(scratch + kGroup * Scratch::kSymBytes);
    uint32_t* const g_heap = reinterpret_cast<uint32_t*>(scratch + kGroup * Scratch::kSymBytes + Scratch::kSmallBytes);
    uint32_t* const g_meta = reinterpret_cast<uint32_t*>(scratch + kGroup * Scratch::kSymBytes + Scratch::kSmallBytes + Scratch::kHeapBytes);
    const zw::Chain chain{sh.entry, BKBITS, POSBITS};
    constexpr uint32_t kBuckets = 1u << BKBITS;
    // Diagnostics (DeflateBatch::trace, normally off): cycles by phase as the serial wave sees them, and what one searching
    // wave does.  Kept in LDS and touched at the phases' ends only: as arrays in registers they were spilled in every loop.
    const bool tracing = b.trace != nullptr;
    uint64_t mark = tracing ? __builtin_readcyclecounter() : 0;
    auto count = [&](int slot, uint64_t v) { if (tracing && lane == 0) sh.trace[slot] += v; };
    auto lap = [&](int phase) {                             // (the serial wave's: set once it is known which wave that is)
        if (!tracing) return;
        const uint64_t now = __builtin_readcyclecounter();
        if (laps && lane == 0) sh.trace[phase] += now - mark;
        mark = now;
    };

    // One wave of the four runs what is serial (the chains, the decisions, the group's blocks).  A CU holds several
    // workgroups and gives each of them one wave on every SIMD, in the same wave slot: the wave whose SIMD number is
    // its slot's (mod 4) is the serial one, so that the workgroups of a CU have theirs on different SIMDs.  (Where
    // waves land is the hardware's to say: the election below yields exactly one wave whatever it says.)
    if (tid == 0) sh.elect = ~0u;
    if (tid < 14) sh.trace[tid] = 0;
    __syncthreads();
    {
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: wave slot [3:0], SIMD [5:4]
        if (lane == 0) atomicMin(&sh.elect, ((((hw >> 4) - hw) & 3u) << 4) | wave);
    }
    __syncthreads();
    const uint32_t serial_wave = sh.elect & 15u;
    const bool serial = wave == serial_wave;
    const bool reports = wave == (serial_wave + 1u) % (kThreads / 64u);   // the searching wave whose doings are traced
    laps = serial;

    // The blocks of a group of objects.  First one lane per object: the trees and the block's head (a Huffman
    // construction is a chain of dependent accesses that lanes cannot share but can each have one of), as many lanes at
    // a time as LDS has room for their heaps.  Then object by object all threads: a stored block's bytes copied; else
    // every symbol's code and extra bits (compress_block of trees.c), each thread a run of symbols, placed by a running
    // sum of their lengths and OR-ed into LDS, then the content, the checksum and the length to their places.
    constexpr uint32_t kMeta = Scratch::kMetaWords;
    constexpr uint32_t kHeapRoom = uint32_t(offsetof(Shared, grab)) / (4u * kHeapInLds);
    constexpr uint32_t kHeapLanes = kHeapRoom >= 32 ? 32 : kHeapRoom >= 16 ? 16 : 8;
    static_assert(kHeapRoom >= 8, "LDS holds the heaps of eight lanes at least");
    static_assert(sizeof(EmitShared<NMAX, WaveClass<NMAX>::kThreads>) <= offsetof(Shared, grab), "the block's content fits where the scan's arrays were");
    auto write_blocks = [&](uint32_t count) {
        __syncthreads();
        if (serial) {
            uint32_t* const heaps = reinterpret_cast<uint32_t*>(&sh);
            for (uint32_t first = 0; first < count; first += kHeapLanes) {
                const uint32_t slot = first + lane;
                if (lane < kHeapLanes && slot < count)
                    begin_block_by_lane<NMAX>(b, tables, scratch + (uint64_t)slot * Scratch::kSymBytes, g_small + slot, heaps + lane, kHeapLanes, g_heap + slot,
                                              g_meta + slot * kMeta);
            }
        }
        __syncthreads();
        lap(5);
        auto& em = *reinterpret_cast<EmitShared<NMAX, WaveClass<NMAX>::kThreads>*>(&sh);
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t* const meta = g_meta + k * kMeta;
            const uint64_t i = ((uint64_t)meta[1] << 32) | meta[0];
            const uint32_t n = meta[2], n_sym = meta[3], adler = meta[4], stored_at = meta[5], pending = meta[6];
            const uint32_t kind = meta[7] >> 24, held = meta[7] & 0xffffu, held_bits = (meta[7] >> 16) & 0xffu;
            uint8_t* const stream = b.out + b.out_off[i];
            if (pending == ~0u) { if (tid == 0) b.out_len[i] = 0; continue; }       // (no room: cannot happen within the bound)
            if (stored_at) {
                const uint8_t* const from = b.arena + b.offsets[i];
                for (uint32_t j = tid; j < n; j += kThreads) stream[stored_at + j] = from[j ^ 3u];
                if (tid == 0) {
                    uint8_t* const end = stream + stored_at + n;
                    end[0] = uint8_t(adler >> 24); end[1] = uint8_t(adler >> 16); end[2] = uint8_t(adler >> 8); end[3] = uint8_t(adler);
                    b.out_len[i] = stored_at + n + 4u;
                }
                continue;
            }
            // the codes into LDS (a fixed block's from the tables, a dynamic block's from the trees the lane built)
            const bool fixed = kind == uint32_t(zd::Deflater<ArenaBytes>::kFixed);
            for (uint32_t e = tid; e < uint32_t(zd::kLCodes); e += kThreads)
                em.lit[e] = fixed ? uint32_t(tables->static_lcode[e]) | uint32_t(tables->static_llen[e]) << 16
                                  : uint32_t(g_small[e * kGroup + k]) | uint32_t(g_small[(zd::kLtN + e) * kGroup + k]) << 16;
            if (tid < uint32_t(zd::kDCodes))
                em.dist[tid] = fixed ? uint32_t(tables->static_dcode[tid]) | 5u << 16
                                     : uint32_t(g_small[(2u * zd::kLtN + tid) * kGroup + k]) | uint32_t(g_small[(2u * zd::kLtN + zd::kDtN + tid) * kGroup + k]) << 16;
            for (uint32_t w = tid; w < uint32_t(sizeof(em.out) / 4); w += kThreads) em.out[w] = 0;
            __syncthreads();
            // a symbol's bits, low bit first: code, extra length bits, distance code, extra distance bits (at most 48)
            const uint16_t* const d_buf = reinterpret_cast<const uint16_t*>(scratch + (uint64_t)k * Scratch::kSymBytes);
            const uint8_t* const l_buf = scratch + (uint64_t)k * Scratch::kSymBytes + 2u * (NMAX + 2);
            auto bits_of = [&](uint32_t at, uint32_t& count_out) -> uint64_t {
                const uint32_t dist = d_buf[at], lc = l_buf[at];
                if (dist == 0) { const uint32_t c = em.lit[lc]; count_out = c >> 16; return c & 0xffffu; }
                const uint32_t lcode = zd::length_code_of(lc), c = em.lit[lcode + zd::kLiterals + 1];
                uint64_t v = c & 0xffffu;
                uint32_t have = c >> 16;
                if (lcode >= 8 && lcode < 28) { const uint32_t extra = (lcode >> 2) - 1u; v |= uint64_t(lc & ((1u << extra) - 1u)) << have; have += extra; }
                const uint32_t dm = dist - 1u, dcode = zd::dist_code_of(dm), dc = em.dist[dcode];
                v |= uint64_t(dc & 0xffffu) << have; have += dc >> 16;
                if (dcode >= 4) { const uint32_t extra = (dcode >> 1) - 1u; v |= uint64_t(dm & ((1u << extra) - 1u)) << have; have += extra; }
                count_out = have;
                return v;
            };
            const uint32_t each = (n_sym + kThreads - 1u) / kThreads, from = tid * each, upto = from + each < n_sym ? from + each : n_sym;
            uint32_t mine = 0;
            for (uint32_t at = from; at < upto; ++at) { uint32_t c; (void)bits_of(at, c); mine += c; }
            // where this thread's run begins: the running sum over the threads before it
            uint32_t before = mine;
#pragma unroll
            for (uint32_t step = 1; step < 64; step <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)before, step); if (lane >= step) before += up; }
            if (lane == 63) em.waves[wave] = before;
            __syncthreads();
            uint32_t position = held_bits + before - mine, all_bits = held_bits;
            for (uint32_t w = 0; w < kThreads / 64u; ++w) { const uint32_t t = em.waves[w]; if (w < wave) position += t; all_bits += t; }
            auto put = [&](uint64_t v, uint32_t at_bit) {
                const uint32_t word = at_bit >> 5, shift = at_bit & 31u;
                const uint64_t low = v << shift;
                const uint32_t w0 = uint32_t(low), w1 = uint32_t(low >> 32), w2 = shift ? uint32_t(v >> (64u - shift)) : 0u;
                if (w0) atomicOr(&em.out[word], w0);
                if (w1) atomicOr(&em.out[word + 1u], w1);
                if (w2) atomicOr(&em.out[word + 2u], w2);
            };
            if (tid == 0) put(held, 0);                      // the head's last bits, which the lane had not yet written
            for (uint32_t at = from; at < upto; ++at) { uint32_t c; const uint64_t v = bits_of(at, c); put(v, position); position += c; }
            if (tid == 0) { const uint32_t c = em.lit[zd::kEndBlock]; put(c & 0xffffu, all_bits); }
            __syncthreads();
            const uint32_t content = (all_bits + (em.lit[zd::kEndBlock] >> 16) + 7u) >> 3;      // bi_windup: to a whole byte
            const bool fits = pending + content + 4u <= zd::bound(n);
            if (fits) {
                const uint8_t* const made = reinterpret_cast<const uint8_t*>(em.out);
                for (uint32_t j = tid; j < content; j += kThreads) stream[pending + j] = made[j];
                if (tid == 0) {
                    uint8_t* const end = stream + pending + content;
                    end[0] = uint8_t(adler >> 24); end[1] = uint8_t(adler >> 16); end[2] = uint8_t(adler >> 8); end[3] = uint8_t(adler);
                }
            }
            if (tid == 0) b.out_len[i] = fits ? pending + content + 4u : 0u;
            __syncthreads();                                // the next object's codes replace these
        }
        __syncthreads();
        lap(6);
    };

    uint32_t in_group = 0;
    for (uint64_t slot = blockIdx.x; slot < b.n; slot += gridDim.x) {
        // which object, and is it this launch's (everything here is the same in all threads)
        uint64_t i = slot;
        if (b.order) { i = b.order[slot]; if (i >= b.n) { if (tid == 0 && NMAX == 1536) atomicOr(b.status, STATUS_BAD_ORDER); continue; } }
        const uint64_t off = b.offsets[i];
        const uint32_t nb = b.nblocks[i];
        uint32_t bad = 0;
        if (off & 15u) bad |= STATUS_MISALIGNED;
        if (off > b.arena_bytes || (uint64_t)nb * 64u > b.arena_bytes - off) bad |= STATUS_OUT_OF_BOUNDS;
        if (bad || nb == 0) {                               // flagged once, by the first class's launch
            if (tid == 0 && NMAX == 1536) { if (bad) atomicOr(b.status, bad); b.out_len[i] = 0; }
            continue;
        }
        if ((uint64_t)nb * 64u > (uint64_t)NMAX + 9u + 63u || (uint64_t)nb * 64u <= n_above) continue;   // too long or too short for this class, whatever its padding says
        const uint32_t* const src = reinterpret_cast<const uint32_t*>(b.arena + off);
        // the framed length is in the padding: the last two dwords of the last block hold it in bits
        const uint64_t bits = ((uint64_t)src[(nb - 1u) * 16u + 14u] << 32) | src[(nb - 1u) * 16u + 15u];
        const uint32_t n = uniform((uint32_t)(bits >> 3));
        if ((bits >> 3) > (uint64_t)NMAX || n <= n_above || n > b.max_input || (uint64_t)n + 9u > (uint64_t)nb * 64u || n == 0) continue;

        // ---- the message into LDS, in byte order (the arena holds its big-endian words as native dwords), zeros behind it
        const uint32_t words = (n + 3u) >> 2, all_words = (n + zw::kPad + 8u) >> 2;
        for (uint32_t w = tid; w < all_words; w += kThreads) sh.data[w] = w < words ? __builtin_bswap32(src[w]) : 0u;
        for (uint32_t k = tid; k < kBuckets / 8u; k += kThreads) reinterpret_cast<uint4*>(sh.a.bucket)[k] = make_uint4(0, 0, 0, 0);
        for (uint32_t k = tid; k < uint32_t(zd::kLCodes + 2); k += kThreads) sh.lt_fc[k] = k == uint32_t(zd::kEndBlock) ? 1 : 0;   // init_block
        if (tid < uint32_t(zd::kDCodes + 2)) sh.dt_fc[tid] = 0;
        if (tid == 0) { sh.sum_a = 0; sh.sum_b = 0; }
        __syncthreads();
        if ((n & 3u) && tid == 0) sh.data[n >> 2] &= (1u << ((n & 3u) * 8u)) - 1u;   // the padding's 0x80 and what follows it
        __syncthreads();
        const uint32_t* const d = sh.data;
        {   // adler32: a = 1 + the bytes' sum, b = n + the sum of (n - i) * byte i, both mod 65521; a thread takes 64 bytes
            uint32_t a = 0, bsum = 0;
            const uint32_t from = tid * 64u;
            if (from < n) {
                const uint32_t upto = from + 64u < n ? from + 64u : n;
                for (uint32_t w = from >> 2; w * 4u < upto; ++w) {          // (bytes behind n are zero)
                    const uint32_t v = d[w], at = n - w * 4u;
                    const uint32_t b0 = v & 0xffu, b1 = (v >> 8) & 0xffu, b2 = (v >> 16) & 0xffu, b3 = v >> 24;
                    a += b0 + b1 + b2 + b3;
                    bsum += at * b0 + (at - 1u) * b1 + (at - 2u) * b2 + (at - 3u) * b3;   // (a factor wraps only where its byte is zero)
                }
                atomicAdd(&sh.sum_a, a);
                atomicAdd(&sh.sum_b, (unsigned long long)bsum);
            }
        }
        lap(0);

        // ---- the hash chains, by the serial wave, 64 positions at a time.  A lane's link is the nearest lane below it
        // with the same bucket, else the bucket's head from before the chunk; the last lane of a bucket becomes its head.
        // Which lanes share a lane's bucket: one ballot per bit of the bucket number, the same work whatever the content.
        if (serial) {
            for (uint32_t base = 0; base + 2u < n; base += 64u) {
                const uint32_t p = base + lane;
                const bool valid = p + 2u < n;
                const uint32_t h = valid ? zw::hash_of(zw::ld32(d, p)) : 0u;
                const uint32_t key = h & (kBuckets - 1u);
                const uint32_t old = valid ? sh.a.bucket[key] : 0u;
                uint64_t same = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < BKBITS; ++bit) {
                    const bool set = (key >> bit) & 1u;
                    const uint64_t with = __ballot(set);
                    same &= set ? with : ~with;
                }
                if (valid) {
                    const uint64_t below = same & ((1ull << lane) - 1ull);
                    const uint32_t lk = below ? base + 63u - (uint32_t)__builtin_clzll(below) : old;
                    sh.entry[p] = uint16_t(chain.make(lk, h));
                    if ((same >> lane) >> 1 == 0) sh.a.bucket[key] = uint16_t(p);
                }
                // (the wave's LDS accesses are served in the order it issues them: the next chunk's loads see these stores)
            }
        }
        if (tid == 0) { sh.grab[0] = 0; sh.grab[1] = 0; }
        __syncthreads();
        lap(1);

        // ---- the scan, tile by tile: three waves search a tile's positions while the serial wave decides over the tile
        // before.  A searching lane takes the next position of the tile that nobody has, walks its chain, stores the
        // answer and takes another: chains are of very different lengths, and a wave steps as slowly as its longest.
        zw::Scan sc;
        uint32_t last_lit = 0;
        const uint32_t tiles = (n + kScanTile - 1u) / kScanTile;
        for (uint32_t tile = 0; tile <= tiles; ++tile) {
            // A searching wave: its lanes take positions of tile `t` until none is left.
            auto search_tile = [&](uint32_t t) {
                const uint32_t first = t * kScanTile;
                zw::Found* const found = sh.found[t & 1u];
                uint32_t* const grab = &sh.grab[t & 1u];
                zw::Walk k;
                bool holds = false, dry = false, hurried = false;   // holds: k's answer is not stored yet; dry: the tile has no position left
                const uint64_t t_begin = tracing ? __builtin_readcyclecounter() : 0;
                uint64_t t_dry = 0;
                uint32_t rounds = 0, rounds_after = 0, entries_after = 0;
                for (;;) {
                    const uint64_t done = __ballot(k.state == zw::Walk::kDone);
                    if (done == ~0ull || (!dry && __builtin_popcountll(done) >= int(kRefill))) {
                        if (k.state == zw::Walk::kDone && holds) { found[k.p - first] = zw::walk_result(k); holds = false; }
                        if (!dry) {                           // as many positions as there are free lanes, with one atomic
                            const uint32_t want = (uint32_t)__builtin_popcountll(done), leader = (uint32_t)__builtin_ctzll(done);
                            uint32_t got = 0;
                            if (lane == leader) got = atomicAdd(grab, want);
                            got = lane_value(got, leader);
                            dry = got + want >= kScanTile;
                            if (k.state == zw::Walk::kDone) {
                                const uint32_t at = got + (uint32_t)__builtin_popcountll(done & ((1ull << lane) - 1ull));
                                const uint32_t p = first + at;
                                if (at < kScanTile && p + 16u <= n) { zw::walk_begin(k, d, n, chain, p); holds = true; }   // (the last 15 are not asked for)
                            }
                        }
                        if (__ballot(k.state != zw::Walk::kDone || holds) == 0) break;
                    }
                    if (!dry) {
                        ++rounds;
#pragma unroll
                        for (uint32_t hop = 0; hop < kHopsInARow; ++hop)
                            if (k.state == zw::Walk::kHop) zw::walk_hop(k, d, chain);
                    } else {
                        // The tile is handed out and ends with its longest chain: what is still walking goes entry by entry
                        // in a loop of nothing else, ahead of the waves that have positions left to take.
                        if (!hurried) { hurried = true; __builtin_amdgcn_s_setprio(2); t_dry = tracing ? __builtin_readcyclecounter() : 0; }
                        ++rounds_after;
                        for (uint32_t hop = 0; hop < kHopsAtTheEnd && __ballot(k.state == zw::Walk::kHop) != 0; ++hop) {
                            ++entries_after;
                            if (k.state == zw::Walk::kHop) zw::walk_hop(k, d, chain);
                        }
                    }
                    if (k.state == zw::Walk::kCompare) zw::walk_compare(k, d);
                }
                if (hurried) __builtin_amdgcn_s_setprio(0);
                if (reports && tracing) {
                    const uint64_t t_end = __builtin_readcyclecounter();
                    count(8, (t_dry ? t_dry : t_end) - t_begin); count(9, t_dry ? t_end - t_dry : 0);
                    count(10, rounds); count(11, rounds_after); count(13, entries_after);
                }
            };
            if (!serial) {
                if (tile < tiles) search_tile(tile);
            } else {
                __builtin_amdgcn_s_setprio(3);               // the one wave everything waits for
                if (tile + 1u < tiles && lane == 0) sh.grab[(tile + 1u) & 1u] = 0;
                if (tile > 0) {
                    const uint32_t first = (tile - 1u) * kScanTile, end = first + kScanTile;
                    const zw::Found* const found = sh.found[(tile - 1u) & 1u];
                    uint32_t part = ~0u;
                    zw::Found fq{0, 0};
                    uint64_t plains = 0;
                    // a symbol as the scan notes it: lane 0 stores two values that are the same in all lanes, and waits for nothing
                    auto note = [&](const zw::Symbol& sy) {
                        if (lane == 0) { sh.a.sym.d_buf[last_lit] = uint16_t(sy.code); sh.a.sym.l_buf[last_lit] = uint8_t(sy.lc); }
                        ++last_lit;
                    };
                    // The decisions proper.  Everything they work with is the same in all lanes, and said to be (uniform()),
                    // so that the compiler keeps it in scalar registers and branches instead of masking lanes.
                    sc.strstart = uniform(sc.strstart); sc.match_length = uniform(sc.match_length); sc.match_start = uniform(sc.match_start);
                    sc.prev_length = uniform(sc.prev_length); sc.prev_match = uniform(sc.prev_match);
                    sc.match_available = uniform(sc.match_available ? 1u : 0u) != 0u;
                    last_lit = uniform(last_lit);
                    while (sc.strstart + 16u <= n && sc.strstart < end) {
                        const uint32_t q = (sc.strstart - first) >> 6, idx = (sc.strstart - first) & 63u;
                        if (q != part) {                      // this lane's answer among the 64 positions, and which of them are plain
                            part = q;
                            fq = found[q * 64u + lane];
                            plains = __ballot(zw::plain(fq) && first + q * 64u + lane + 16u <= n);
                        }
                        // a run of plain positions entered with no match pending: literals, by as many lanes
                        uint32_t run = 0;
                        if (sc.match_length < uint32_t(zd::kMinMatch)) {
                            const uint64_t rest = plains >> idx;                    // bit k: position idx + k is plain
                            run = rest == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~rest);   // (idx > 0 shifts zeros in: never all ones then)
                        }
                        if (run > 1u) {
                            const uint32_t skip = sc.match_available ? 0u : 1u;     // the scan's very first position only opens the window
                            if (lane >= skip && lane < run) {
                                sh.a.sym.d_buf[last_lit + lane - skip] = uint16_t(zw::kLiteral + sc.strstart + lane - 1u);
                                sh.a.sym.l_buf[last_lit + lane - skip] = 0;
                            }
                            last_lit += run - skip;
                            sc.strstart += run;
                            sc.match_available = true;
                            sc.prev_length = sc.match_length = zw::kNone;
                            continue;
                        }
                        const zw::Found fi{lane_value(fq.r32, idx), lane_value(fq.r128, idx)};
                        const zw::Symbol sy = zw::scan_step(sc, d, n, fi);
                        if (sy.emit) note(sy);
                    }
                    while (sc.strstart < n && sc.strstart < end) {       // the last 15 positions: zlib's own search, one by one
                        const zw::Symbol sy = zw::scan_step_tail(sc, d, n, chain);
                        if (sy.emit) note(sy);
                    }
                    if (sc.strstart >= n && tile == tiles) {
                        if (sc.match_available) note(zw::Symbol{true, zw::kLiteral + sc.strstart - 1u, 0});
                        if (lane == 0) sh.n_sym = last_lit;
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                lap(3);
                if (tile < tiles) search_tile(tile);         // and then it searches with the others
            }
            const uint64_t t_arrive = tracing ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            if (reports && tracing) count(12, __builtin_readcyclecounter() - t_arrive);
            lap(2);                                         // (what the serial wave waited for the searches)
        }
        // ---- the symbols into zlib's form (a literal's byte is read only now) and their counts, a symbol per thread
        const uint32_t n_sym = sh.n_sym;
        for (uint32_t k = tid; k < n_sym; k += kThreads) {
            uint32_t dist, lc, dcode;
            const uint32_t lcode = zw::settle(d, sh.a.sym.d_buf[k], sh.a.sym.l_buf[k], dist, lc, dcode);
            sh.a.sym.d_buf[k] = uint16_t(dist); sh.a.sym.l_buf[k] = uint8_t(lc);
            bump_lds(&sh.lt_fc[lcode]);
            if (dcode != ~0u) bump_lds(&sh.dt_fc[dcode]);
        }
        if (tid == 0) {
            uint32_t* const meta = g_meta + in_group * kMeta;
            meta[0] = (uint32_t)i; meta[1] = (uint32_t)(i >> 32); meta[2] = n; meta[3] = n_sym;
            meta[4] = (uint32_t)((sh.sum_b + n) % 65521u) << 16 | (sh.sum_a + 1u) % 65521u;
        }
        __syncthreads();

        // ---- symbols and counts to the object's slot of the workgroup's scratch
        {
            uint32_t* const sym = reinterpret_cast<uint32_t*>(scratch + (uint64_t)in_group * Scratch::kSymBytes);
            const uint32_t* const from = reinterpret_cast<const uint32_t*>(&sh.a.sym);
            constexpr uint32_t kSymWords = (3u * (NMAX + 2) + 3u) / 4u;          // d_buf and l_buf lie back to back in LDS as in the slot
            const uint32_t d_words = (2u * n_sym + 3u) / 4u, l_first = (2u * (NMAX + 2)) / 4u, l_words = (n_sym + 3u) / 4u;
            for (uint32_t k = tid; k < d_words && k < kSymWords; k += kThreads) sym[k] = from[k];
            for (uint32_t k = tid; k < l_words && l_first + k < kSymWords; k += kThreads) sym[l_first + k] = from[l_first + k];
            // the counts into the tree arrays as Deflater::setup lays them out: lt_fc, lt_dl, dt_fc, dt_dl, bt_fc, ...
            for (uint32_t k = tid; k < uint32_t(zd::kLCodes); k += kThreads) g_small[k * kGroup + in_group] = sh.lt_fc[k];
            if (tid < uint32_t(zd::kDCodes)) g_small[(2u * zd::kLtN + tid) * kGroup + in_group] = sh.dt_fc[tid];
            if (tid < uint32_t(zd::kBLCodes)) g_small[(2u * zd::kLtN + 2u * zd::kDtN + tid) * kGroup + in_group] = 0;
        }
        lap(4);
        if (++in_group == kGroup) { write_blocks(in_group); in_group = 0; }
        else __syncthreads();                               // the next object reuses every array
        if (serial) count(7, 1);
    }
    if (in_group) write_blocks(in_group);
    __syncthreads();
    if (tracing && tid < 14 && sh.trace[tid]) atomicAdd(reinterpret_cast<unsigned long long*>(&b.trace[tid]), sh.trace[tid]);
}

uint64_t deflate_workspace_bytes(uint32_t lanes) { return (uint64_t)lanes * (zd::work_bytes() + zd::small_bytes()); }
uint64_t deflate_tables_bytes() { return sizeof(zd::Tables); }
void deflate_make_tables(void* host_dst) { zd::make_tables(*static_cast<zd::Tables*>(host_dst)); }
uint32_t deflate_bound(uint32_t framed_len) { return zd::bound(framed_len); }
uint32_t deflate_max_input() { return zd::kMaxInput; }

uint32_t deflate_wave_max_input() { return zw::kMaxInput; }

// Workgroups a launch of one class keeps resident: what the device holds of them at once (LDS decides).
template <int NMAX, int BKBITS, int POSBITS>
static hipError_t wave_class_shape(uint32_t* groups, uint64_t* scratch_bytes)
{
    static int per_cu = 0, cus = 0;                        // the same on every gfx950; asked once
    if (!per_cu) {
        int dev = 0, n = 0, c = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, deflate_wave_kernel<NMAX, BKBITS, POSBITS>, WaveClass<NMAX>::kThreads, 0);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        per_cu = n > 0 ? n : 1; cus = c > 0 ? c : 256;
    }
    *groups = uint32_t(per_cu) * uint32_t(cus);
    *scratch_bytes = uint64_t(*groups) * WaveScratch<NMAX>::kBytes;
    return hipSuccess;
}

template <int NMAX, int BKBITS, int POSBITS>
static hipError_t launch_wave_class(const DeflateBatch& b, uint32_t n_above, hipStream_t stream)
{
    uint32_t groups = 0;
    uint64_t bytes = 0;
    hipError_t e = wave_class_shape<NMAX, BKBITS, POSBITS>(&groups, &bytes);
    if (e != hipSuccess) return e;
    if (groups > b.n) groups = b.n;
    hipLaunchKernelGGL((deflate_wave_kernel<NMAX, BKBITS, POSBITS>), dim3(groups), dim3(WaveClass<NMAX>::kThreads), 0, stream, b, n_above);
    return hipGetLastError();
}

// Device memory the wave kernels need behind DeflateBatch::workspace: the largest class's scratch (the classes run one
// after the other).  With the device current.
hipError_t deflate_wave_workspace_bytes(uint64_t* out)
{
    uint32_t g;
    uint64_t bytes = 0, most = 0;
    hipError_t e = wave_class_shape<1536, 12, 11>(&g, &bytes); most = bytes > most ? bytes : most;
    if (e == hipSuccess) { e = wave_class_shape<4608, 13, 13>(&g, &bytes); most = bytes > most ? bytes : most; }
    if (e == hipSuccess) { e = wave_class_shape<9216, 14, 14>(&g, &bytes); most = bytes > most ? bytes : most; }
    if (e == hipSuccess) { e = wave_class_shape<16382, 15, 14>(&g, &bytes); most = bytes > most ? bytes : most; }
    *out = most;
    return e;
}

// All objects of up to deflate_wave_max_input() bytes: one launch per size class, each over the whole batch.
hipError_t launch_deflate_wave(const DeflateBatch& b, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    if (!b.tables_raw || !b.workspace) return hipErrorInvalidValue;
    static_assert(zw::kClasses[0].nmax == 1536 && zw::kClasses[1].bkbits == 13 && zw::kClasses[2].posbits == 14 && zw::kClasses[3].nmax == 16382,
                  "the launches below are zwave.h's classes");
    hipError_t e = launch_wave_class<1536, 12, 11>(b, 0, stream);
    if (e == hipSuccess) e = launch_wave_class<4608, 13, 13>(b, 1536, stream);
    if (e == hipSuccess) e = launch_wave_class<9216, 14, 14>(b, 4608, stream);
    if (e == hipSuccess) e = launch_wave_class<16382, 15, 14>(b, 9216, stream);
    return e;
}

// The lane kernel, for the objects longer than only_above (at most deflate_max_input()); its tables: lane_workspace.
hipError_t launch_deflate(DeflateBatch b, uint8_t* lane_workspace, uint32_t lanes, uint32_t only_above, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    b.workspace = lane_workspace;
    if (!b.workspace || !b.tables_raw || lanes < 64) return hipErrorInvalidValue;
    // (Spreading a batch of fewer objects than lanes over more waves with fewer lanes at work each was measured and is
    // slower: 32 768 text-like objects 82 ms on 1 024 half-empty waves against 78 ms on 512 full ones.)
    uint32_t waves = lanes / 64u;
    const uint64_t need = ((uint64_t)b.n + 63u) / 64u;
    if (waves > need) waves = (uint32_t)need;
    hipLaunchKernelGGL(deflate_kernel, dim3(waves), dim3(64), 0, stream, b, only_above);
    return hipGetLastError();
}

}  // namespace vit
