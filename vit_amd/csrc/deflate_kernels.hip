// deflate_kernels.hip — the loose-object file contents, made where the objects already are.
//
// The reference stores every object as compress(header + content) (src/commit.cpp:98-105), and
// that call is three quarters of its write-tree.  The arena the hasher stages in device memory
// holds exactly header + content; this kernel writes the zlib stream for every object of it, one
// object per lane, with zlib's own choices (zdeflate.h), so the host is left with creating files.
//
// One lane runs one object from start to finish: deflate is a serial scan with a hash table, as
// SHA-1 is a serial chain.  The tables a lane needs (zlib's head and prev arrays, the symbol
// buffers of a block, the Huffman construction's arrays: 181 KiB) live in device memory, one
// workspace per wave with the 64 lanes' entries interleaved.  A wave takes 64 objects at a time
// and moves on to its next 64 when the longest of them is done.  The work is bound by scattered
// two-byte accesses to those tables (the scan, and the Huffman heap of every block), i.e. by the
// memory system's transaction rate, not by arithmetic: every lane's object, tables and position are
// its own, so a wave's memory instruction touches up to 64 cache lines.  Incompressible content runs
// at 8-11 us per input byte and lane (1.4-2 M objects of 4 KiB per second over the chip), text-like
// content, whose positions have hash chains to walk, at about four times that.
#include "kernels.h"
#include "zdeflate.h"

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
__global__ __launch_bounds__(64) void deflate_kernel(DeflateBatch b)
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
        // zlib starts from an empty hash table: the wave clears its 64 interleaved tables together
        uint4* const hd = reinterpret_cast<uint4*>(head16);
        for (uint32_t k = lane; k < 64u * zd::kHashSize * 2u / 16u; k += 64u) hd[k] = make_uint4(0, 0, 0, 0);
        __syncthreads();                                  // one wave: orders the lanes' stores before each other's loads

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
        }
        if (mine) {
            // the framed length is in the padding: the last two dwords of the last block hold it in bits
            const uint32_t* last = reinterpret_cast<const uint32_t*>(b.arena + off + (uint64_t)(nb - 1u) * 64u);
            n = (((uint64_t)last[14] << 32) | last[15]) >> 3;
            if (n > (uint64_t)zd::kMaxInput || n + 9u > (uint64_t)nb * 64u) { b.out_len[i] = 0; mine = false; }   // the host's share
        }
        if (mine) {
            w.out = b.out + b.out_off[i];
            w.out_cap = zd::bound((uint32_t)n);
            d.in = ArenaBytes{b.arena + off};
            b.out_len[i] = d.run(static_cast<const zd::Tables*>(b.tables_raw), w, (uint32_t)n);
        }
        __syncthreads();                                  // the tables are cleared again only when every lane is done with them
    }
}

uint64_t deflate_workspace_bytes(uint32_t lanes) { return (uint64_t)lanes * (zd::work_bytes() + zd::small_bytes()); }
uint64_t deflate_tables_bytes() { return sizeof(zd::Tables); }
void deflate_make_tables(void* host_dst) { zd::make_tables(*static_cast<zd::Tables*>(host_dst)); }
uint32_t deflate_bound(uint32_t framed_len) { return zd::bound(framed_len); }
uint32_t deflate_max_input() { return zd::kMaxInput; }

hipError_t launch_deflate(const DeflateBatch& b, uint32_t lanes, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    if (!b.workspace || !b.tables_raw || lanes < 64) return hipErrorInvalidValue;
    // (Spreading a batch of fewer objects than lanes over more waves with fewer lanes at work each was measured and is
    // slower: 32 768 text-like objects 82 ms on 1 024 half-empty waves against 78 ms on 512 full ones.)
    uint32_t waves = lanes / 64u;
    const uint64_t need = ((uint64_t)b.n + 63u) / 64u;
    if (waves > need) waves = (uint32_t)need;
    hipLaunchKernelGGL(deflate_kernel, dim3(waves), dim3(64), 0, stream, b);
    return hipGetLastError();
}

}  // namespace vit
