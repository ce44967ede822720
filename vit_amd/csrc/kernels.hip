// kernels.hip — gfx950 kernels for batched Git-object SHA-1.
//
// Work decomposition: ONE LANE PER OBJECT.  SHA-1 is a serial chain over a
// message's blocks, so the parallel unit is the object; a wave hashes 64 objects
// in lock step, block by block.  The data-movement problem is that the 64
// objects of a wave live at 64 unrelated arena addresses:
//
//  * sha1_lds_kernel (default) lets the wave fetch cooperatively: 4*BPS
//    consecutive lanes read one object's next BPS blocks as one contiguous
//    64*BPS-byte run (whole cache lines, 16 B per lane), written straight into
//    LDS by global_load_lds_dwordx4.  Each lane then pulls its own object's
//    blocks out of LDS with ds_read_b128.  The LDS image is XOR-swizzled on the
//    global SOURCE address (the DMA destination is fixed at base + lane*16) so
//    those reads are bank-conflict free.
//  * sha1_lane_kernel is the plain form (each lane loads its own 64 bytes with
//    four global_load_dwordx4, one block prefetched ahead); kept as the
//    first-principles comparison and for A/B measurement.
//
// Both hash whole pre-padded blocks (see DeviceBatch) and write 20-byte digests.
#include "kernels.h"
#include "sha1_device.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <type_traits>

namespace vit {

typedef __attribute__((address_space(1))) const void* global_cptr;
typedef __attribute__((address_space(3))) void*       lds_ptr;

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t other = (uint32_t)__shfl_xor((int)v, o, 64);
        v = other > v ? other : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t other = (uint32_t)__shfl_xor((int)v, o, 64);
        v = other < v ? other : v;
    }
    return v;
}

// 64-bit wave reductions; the result is wave-uniform (held in SGPRs).
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t other = (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
        v = other < v ? other : v;
    }
    // readfirstlane returns int: go through uint32_t, or the low half sign-extends into the high one
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t other = (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
        v = other > v ? other : v;
    }
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return ((uint64_t)hi << 32) | lo;
}

// Resolve slot -> (object index, offset, block count) with the input checks
// that keep a bad batch from faulting: anything out of range hashes 0 blocks
// and raises a status bit instead.  `owns` says the slot has a digest to store.
struct Slot { uint32_t idx; uint32_t nb; uint64_t off; bool owns; };

__device__ __forceinline__ Slot resolve_slot(const DeviceBatch& b, uint32_t slot)
{
    const bool in_range = slot < b.n;
    uint32_t idx = slot;
    if (in_range && b.order) idx = b.order[slot];
    const bool idx_ok = in_range && idx < b.n;
    uint64_t off = 0;
    uint32_t nb = 0;
    if (idx_ok) { off = b.offsets[idx]; nb = b.nblocks[idx]; }
    uint32_t bad = 0;
    if (in_range && !idx_ok) bad |= STATUS_BAD_ORDER;
    if (off & 15u) bad |= STATUS_MISALIGNED;
    if (off > b.arena_bytes || (uint64_t)nb * 64u > b.arena_bytes - off) bad |= STATUS_OUT_OF_BOUNDS;
    if (bad) { atomicOr(b.status, bad); nb = 0; }
    Slot s;
    s.idx = idx_ok ? idx : 0u; s.nb = nb; s.off = off; s.owns = idx_ok;
    return s;
}

// Plain per-lane loop: the lane loads its own 64-byte blocks (four 16-byte
// loads, the next block prefetched while the current one is hashed).
__device__ __forceinline__ void hash_lane_direct(const uint8_t* first_block, uint32_t nb, uint32_t (&h)[5])
{
    const uint4* p = reinterpret_cast<const uint4*>(first_block);
    uint4 q0, q1, q2, q3;
    if (nb) { q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3]; }
    for (uint32_t i = 0; i < nb; ++i) {
        uint32_t w[16];
        put_words(w, 0, q0); put_words(w, 4, q1); put_words(w, 8, q2); put_words(w, 12, q3);
        if (i + 1 < nb) {
            const uint4* nx = p + 4ull * (i + 1);
            q0 = nx[0]; q1 = nx[1]; q2 = nx[2]; q3 = nx[3];
        }
        sha1_compress(h, w);
    }
}

// Diagnostics: when and where a wave ran (DeviceBatch::wave_trace).  HW_ID (hardware register 4)
// holds wave, SIMD, CU, SH and SE ids, XCC_ID (register 20) the XCD.
__device__ __forceinline__ void trace_wave(uint64_t* trace, uint32_t wave_index, uint64_t t0, uint64_t tag)
{
    uint64_t* rec = trace + (uint64_t)wave_index * 4u;
    rec[0] = t0;
    rec[1] = wall_clock64();
    rec[2] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    rec[3] = tag;
}

// ---------------------------------------------------------------- LDS-staged --
// No workgroup barrier anywhere: a wave only ever touches its own slice of LDS.
// Between its own LDS writes and reads it needs the LDS queue drained and the
// compiler kept from moving accesses across.
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// The stage pipeline for a wave whose objects span more than 4 GiB: the same
// LDS image and reads as in sha1_lds_kernel below, every load predicated, the
// source a 64-bit address per lane and instruction (two more vector ops per
// load than the scalar-base form).
template <int BPS>
__device__ __forceinline__ void far_stages(const uint8_t* arena, uint8_t* stage, const uint64_t (&src_off)[4 * BPS],
                                        const int32_t (&lim)[4 * BPS], uint32_t my_row, uint32_t my_sw,
                                        uint32_t my_nb, uint32_t nst, uint32_t (&h)[5])
{
    constexpr int LPO = 4 * BPS, ROWB = 64 * BPS;
    auto issue = [&](uint32_t t) {
#pragma unroll
        for (int j = 0; j < LPO; ++j)
            if ((int32_t)(BPS * t) < lim[j])
                __builtin_amdgcn_global_load_lds((global_cptr)(arena + src_off[j] + (uint64_t)t * ROWB),
                                                 (lds_ptr)(stage + j * 1024), 16, 0, 0);
    };
    if (nst) issue(0);
    for (uint32_t t = 0; t < nst; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 q[LPO];
#pragma unroll
        for (int k = 0; k < LPO; ++k)
            q[k] = *reinterpret_cast<const uint4*>(stage + my_row + ((k ^ my_sw) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 1 < nst) issue(t + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < BPS; ++s) {
            if (BPS * t + s < my_nb) {
                uint32_t w[16];
                put_words(w, 0, q[4 * s]);     put_words(w, 4, q[4 * s + 1]);
                put_words(w, 8, q[4 * s + 2]); put_words(w, 12, q[4 * s + 3]);
                sha1_compress(h, w);
            }
        }
    }
}

// Stage t moves blocks [BPS*t, BPS*t + BPS) of all 64 objects.  Instruction j of
// a stage (j < LPO) serves the OPI = 64/LPO objects j*OPI .. j*OPI+OPI-1, LPO =
// 4*BPS lanes each: lane l serves object j*OPI + l/LPO and lands at LDS byte
// j*1024 + l*16, i.e. row (l/LPO) of ROWB = 64*BPS bytes, 16-byte position
// p = l % LPO.  Position p holds chunk c = p ^ sw(object) of the row, so the
// lane fetches global chunk c.  Reader lane i wants chunk k of object i and
// finds it at position k ^ sw(i).  sw(i) = (i / (16/LPO)) % LPO makes the 16
// lanes of every ds_read_b128 lane group hit 16 different 16-byte bank slots.
// LDS (64*BPS*64 bytes per wave) admits 160 KiB / 8 KiB = 20 waves per CU = 5 per
// SIMD at BPS = 2; the launch bound asks the register allocator for the same.
//
// WPB waves share a workgroup only to get past the limit on resident
// workgroups per CU (one-wave workgroups filled 3.6 of the 5 wave slots per
// SIMD); each wave still works on its own 64 objects and its own slice of LDS.
// One wave hashes the 64 objects of slots [64*group, 64*group + 64) through its
// own 64*64*BPS-byte stage buffer in LDS.
template <int BPS>
__device__ __forceinline__ void hash_group(const DeviceBatch& b, uint8_t* const stage, uint32_t group, uint32_t lane)
{
    constexpr int LPO  = 4 * BPS;        // lanes per object per instruction == instructions per stage
    constexpr int OPI  = 64 / LPO;       // objects per instruction
    constexpr int ROWB = 64 * BPS;       // bytes per object per stage
    constexpr int RP   = 16 / LPO > 0 ? 16 / LPO : 1;   // rows per 256-byte bank line

    const uint32_t slot = group * 64u + lane;

    const uint64_t trace_t0 = b.wave_trace ? wall_clock64() : 0;
    const Slot me = resolve_slot(b, slot);
    const uint32_t idx = me.idx, my_nb = me.nb;
    const uint64_t my_off = me.off;
    const bool owns = me.owns;

    // Publish (offset, nblocks) of the wave's 64 objects; the tables borrow the
    // stage buffer, which is not in use yet.
    uint64_t* t_off = reinterpret_cast<uint64_t*>(stage);
    uint32_t* t_nb  = reinterpret_cast<uint32_t*>(stage + 64 * 8);
    t_off[lane] = my_off;
    t_nb[lane]  = my_nb;
    wave_lds_sync();

    const uint32_t p = lane % LPO;
    const uint32_t my_sw  = (lane / RP) % LPO;
    const uint32_t my_row = (lane / OPI) * 1024u + (lane % OPI) * ROWB;
    const uint32_t max_nb = __builtin_amdgcn_readfirstlane(wave_max_u32(my_nb));
    const uint32_t nst    = (max_nb + BPS - 1) / BPS;

    // Wave-uniform facts: the lowest object offset of the wave is the base of
    // every fetch, a lane adds a fixed 32-bit offset to it.  That needs the 64
    // objects within 4 GiB of each other.  A wave whose objects are not (an
    // arena beyond 4 GiB hashed in binned order: equal lengths lie all over it)
    // runs the same pipeline with a 64-bit address per lane and instruction
    // instead (far_stages below).  full_stages: stages complete for all 64 lanes
    // (a lane without an object has 0 blocks, so a ragged last wave has none).
    const uint32_t min_nb  = __builtin_amdgcn_readfirstlane(wave_min_u32(my_nb));
    const uint64_t min_off = wave_min_u64(my_off), max_off = wave_max_u64(my_off);
    const bool near = (max_off - min_off) < 0xFFF00000ull;
    const uint32_t full_stages = min_nb / BPS;
    const uint8_t* wave_base = b.arena + min_off;

    uint32_t voff[LPO];          // near: this lane's source for instruction j is wave_base + voff[j] (+ stage offset)
    uint64_t foff[LPO];          // far:  ... is arena + foff[j] (+ stage offset)
    int32_t  lim[LPO];           // instruction j is live for this lane while BPS*t < lim[j]
#pragma unroll
    for (int j = 0; j < LPO; ++j) {
        const uint32_t o  = j * OPI + lane / LPO;          // object (lane id) served
        const uint32_t sw = (o / RP) % LPO;
        const uint32_t c  = p ^ sw;                        // chunk of the row to fetch
        const uint64_t src = t_off[o] + c * 16u;
        if (near) voff[j] = (uint32_t)(src - min_off);
        else      foff[j] = src;
        lim[j]  = (int32_t)t_nb[o] - (int32_t)(c >> 2);    // chunk c lies in block c/4 of the stage
    }
    wave_lds_sync();             // table reads are done before the first DMA lands on them

    uint32_t h[5];
    sha1_init(h);

    if (!near) {
        far_stages<BPS>(b.arena, stage, foff, lim, my_row, my_sw, my_nb, nst, h);
        if (owns) store_digest(b.digests, idx, h);
        if (b.wave_trace && lane == 0) trace_wave(b.wave_trace, group, trace_t0, max_nb | (1ull << 40));
        return;
    }

    // One LDS-DMA load: 16 bytes per live lane from (scalar base + lane offset)
    // to LDS byte lds_dst + lane*16.  hipcc does not select this scalar-base form
    // from the builtin, so it is written out; with it a stage costs no vector
    // instructions for addressing (the scalar unit advances the base).  M0
    // carries the LDS destination and is saved and restored because the compiler
    // keeps its own value in it.  The load has no register result; its
    // completion is awaited by the explicit vmcnt(0) at the top of every stage.
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)stage;
    auto dma16 = [](uint32_t lane_off, const uint8_t* base, uint32_t lds_dst) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(base), "s"(lds_dst)
                     : "memory");
    };
    // FULL: every lane is live (no predicates at all).
    auto issue = [&](uint32_t t, auto full_tag) {
        const uint8_t* sb = wave_base + (uint64_t)t * ROWB;
#pragma unroll
        for (int j = 0; j < LPO; ++j)
            if (decltype(full_tag)::value || (int32_t)(BPS * t) < lim[j])
                dma16(voff[j], sb, lds_base + (uint32_t)j * 1024u);
    };

    // One stage: wait for its DMA, pull this lane's blocks into registers, start
    // the next stage's DMA, hash.  FULL: this stage and the next are complete for
    // all 64 lanes.
    auto stage_body = [&](uint32_t t, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        // The DMA writes are tracked by vmcnt only; nothing else orders the
        // ds_reads below behind them.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 q[LPO];
#pragma unroll
        for (int k = 0; k < LPO; ++k)
            q[k] = *reinterpret_cast<const uint4*>(stage + my_row + ((k ^ my_sw) << 4));
        // All of this stage is in registers before the next stage's DMA may
        // overwrite the buffer.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (FULL || t + 1 < nst) issue(t + 1, full_tag);
        // The loads must be in flight while this stage is hashed: without the
        // barrier the scheduler sinks them below the rounds (nothing in the
        // arithmetic depends on them) and the wave then waits out the whole
        // memory latency at the top of the next stage.
        __builtin_amdgcn_sched_barrier(0);

#pragma unroll
        for (int s = 0; s < BPS; ++s) {
            if (FULL || BPS * t + s < my_nb) {
                uint32_t w[16];
                put_words(w, 0, q[4 * s]);     put_words(w, 4, q[4 * s + 1]);
                put_words(w, 8, q[4 * s + 2]); put_words(w, 12, q[4 * s + 3]);
                sha1_compress(h, w);
            }
        }
    };

    uint32_t t = 0;
    if (nst) issue(0, std::false_type());
    for (; t + 1 < full_stages; ++t) stage_body(t, std::true_type());
    for (; t < nst; ++t) stage_body(t, std::false_type());
    if (owns) store_digest(b.digests, idx, h);
    if (b.wave_trace && lane == 0) trace_wave(b.wave_trace, group, trace_t0, max_nb);
}

// The launch gate (DeviceBatch::gate_percent): wave-uniform, two dependent scalar loads.
__device__ __forceinline__ bool gate_picks_scheduled(const DeviceBatch& b)
{
    uint32_t first = 0;
    if (b.order) first = b.order[0];                    // binned order: the first object is the longest
    const uint32_t longest = first < b.n ? b.nblocks[first] : 0u;
    return tail_dominates(longest, b.arena_bytes, b.gate_simds, b.gate_percent);
}

template <int BPS, int WPB>
__global__ __launch_bounds__(64 * WPB, BPS == 4 ? 2 : (BPS == 1 ? 4 : 5)) void sha1_lds_kernel(DeviceBatch b)
{
    if (b.gate_percent && gate_picks_scheduled(b)) return;
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[WPB][64 * 64 * BPS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    hash_group<BPS>(b, stage_all[wave], blockIdx.x * WPB + wave, lane);
}

// ------------------------------------------------ software-scheduled launch --
// For batches in binned order (group 0 holds the longest objects).  The hardware
// dispatcher hands out workgroups as wave slots free up, and a SIMD shares its
// issue slots among its resident waves: on a batch with a long tail the SIMD
// that holds the longest wave also gets four of the next-longest, the long wave
// then runs at a fraction of the SIMD's rate, and the launch ends when that
// SIMD does (measured on a 17 GB skewed batch: 6.9 to 7.1 ms where the chip's
// issue capacity allows 4.7).
//
// Here the grid is a fixed set of resident waves, one per SIMD by default, that
// pull 64-object groups from ONE cursor running down the binned order:
//   * the first wave to arrive on a physical SIMD is that SIMD's LEADER.  It
//     pulls the next group whenever it is free: the longest groups go to
//     different SIMDs, and a SIMD takes its next long group only when it has
//     finished the last (list scheduling of a sorted list, i.e. longest
//     processing time first).  Every SIMD then ends within 1 % of the others.
//   * any other wave that lands on the SIMD is a SPARE.  A spare pulls a group
//     only while its leader is on a group of at most `share_below` blocks and
//     otherwise sleeps (long sleeps: spares that polled every 2 us cost the
//     leaders 12 %).  Spares exist so that an uneven placement of the grid by
//     the hardware cannot leave two long groups on one SIMD; measured, they do
//     not pay for themselves as helpers (two waves per SIMD: 5.76 ms against
//     5.60 ms for leaders alone), so the default launches none.
// What bounds the leaders-only launch is the rate of ONE wave on a SIMD: 1.32 us
// per block against 1.10 for five waves sharing it (the arithmetic alone runs
// at about 1.2 us per block on a lone wave; its LDS reads and DMA issue, which
// other waves would cover, add the rest).  A second stage of DMA in flight for
// such waves was tried and changed nothing: the next stage's data already lands
// in time.  A 4097-block group cannot run on two waves (one SHA-1 stream per
// lane), so 4097 x 1.32 us = 5.4 ms is this batch's floor.
// Every wave leaves once the cursor has passed the last group, which every wave
// observes.  The state returns to all-zero at the end of the launch (each wave
// unregisters, each leader withdraws its length, the last wave out clears the
// cursor), ready for the next launch that uses it.
__device__ __forceinline__ uint32_t physical_simd()
{
    const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
    const uint32_t simd = (hw >> 4) & 3u, cu = (hw >> 8) & 0xFu, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    // 8 XCDs x 8 SE x 2 SH x 16 CU x 4 SIMD = SCHED_SIMD_KEYS keys.  The XCC_ID field is four bits wide: a
    // part that reported an id above 7 would index past the tables, so the key is folded into their size (two
    // SIMDs would then share an entry, which costs one of them its leader, never a digest).
    static_assert(SCHED_SIMD_KEYS == 8u * 8u * 2u * 16u * 4u && (SCHED_SIMD_KEYS & (SCHED_SIMD_KEYS - 1u)) == 0u,
                  "the key's fields and the size of SchedState's tables belong together");
    return (((xcc * 8u + se) * 2u + sh) * 64u + cu * 4u + simd) & (SCHED_SIMD_KEYS - 1u);
}

__device__ __forceinline__ uint32_t peek(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int BPS, int WPB>
__global__ __launch_bounds__(64 * WPB, 2) void sha1_sched_kernel(DeviceBatch b, SchedState* st, uint32_t groups,
                                                                 uint32_t nwaves, uint32_t share_below)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[WPB][64 * 64 * BPS];
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t* const stage = stage_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    if (b.gate_percent && !gate_picks_scheduled(b)) return;      // before anything touches the state: it stays all zero
    const uint32_t key = physical_simd();
    uint32_t rank = 0;
    if (lane == 0) rank = atomicAdd(&st->simd_waves[key], 1u);
    const bool leader = __builtin_amdgcn_readfirstlane(rank) == 0;
    if (leader) __builtin_amdgcn_s_setprio(3);
    for (;;) {
        if (!leader) {
            // simd_len: 0 = the leader has not published yet (or has left); else its group's length + 1
            bool go = false;
            for (;;) {
                if (peek(&st->next) >= groups) break;
                const uint32_t len1 = peek(&st->simd_len[key]);
                if (len1 != 0 && len1 - 1u <= share_below) { go = true; break; }
                for (int z = 0; z < 16; ++z) __builtin_amdgcn_s_sleep(127);
            }
            if (!go) break;
        }
        uint32_t g = 0xFFFFFFFFu;
        if (lane == 0) {
            const uint32_t c = atomicAdd(&st->next, 1u);
            if (c < groups) g = c;
        }
        g = __builtin_amdgcn_readfirstlane(g);
        if (g == 0xFFFFFFFFu) break;
        if (leader && lane == 0) {
            uint32_t first = g * 64u;                      // binned order: the group's first object is its longest
            if (b.order) first = b.order[first];
            const uint32_t len = first < b.n ? b.nblocks[first] : 0u;
            atomicExch(&st->simd_len[key], len == 0xFFFFFFFFu ? len : len + 1u);
        }
        hash_group<BPS>(b, stage, g, lane);
        wave_lds_sync();                 // the next group's tables go where this group's last stage was read from
    }
    if (lane == 0) {
        if (leader) atomicExch(&st->simd_len[key], 0u);
        atomicSub(&st->simd_waves[key], 1u);
        __threadfence();
        if (atomicAdd(&st->exited, 1u) == nwaves - 1u) { atomicExch(&st->next, 0u); atomicExch(&st->exited, 0u); }
    }
}

hipError_t device_simds(uint32_t* out)
{
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) *out = uint32_t(cus) * 4u;
    return e;
}

hipError_t launch_sha1_sched(const DeviceBatch& b, SchedState* state, uint32_t waves_per_simd, uint32_t share_below,
                             hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    if (!state) return hipErrorInvalidValue;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const uint32_t groups = uint32_t((uint64_t(b.n) + 63u) / 64u);
    if (waves_per_simd == 0 || waves_per_simd > 2) waves_per_simd = 2;
    // one four-wave workgroup per CU and wave-per-SIMD; never more waves than groups
    uint32_t wgs = uint32_t(cus) * waves_per_simd;
    if (wgs > (groups + 3u) / 4u) wgs = (groups + 3u) / 4u;
    const uint32_t nwaves = wgs * 4u;
    hipLaunchKernelGGL((sha1_sched_kernel<2, 4>), dim3(wgs), dim3(256), 0, stream, b, state, groups, nwaves, share_below);
    return hipGetLastError();
}

// ------------------------------------------------------------- lane-direct --
__global__ __launch_bounds__(256) void sha1_lane_kernel(DeviceBatch b)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const Slot me = resolve_slot(b, slot);
    if (!me.owns) return;
    const uint32_t idx = me.idx, nb = me.nb;
    const uint64_t off = me.off;

    uint32_t h[5];
    sha1_init(h);
    hash_lane_direct(b.arena + off, nb, h);
    store_digest(b.digests, idx, h);
}

// Workgroups and threads per workgroup of a launch over n objects.  64-bit
// arithmetic: n may be anything up to 2^32 - 1, and n + 255 must not wrap.
void launch_shape(KernelVariant v, uint64_t n, uint32_t* blocks, uint32_t* threads)
{
    uint64_t wpb;                       // waves per workgroup
    switch (int(v)) {
    case KERNEL_LANE: wpb = 4; break;
    case 2: case 3: case 5: wpb = 1; break;
    case 4: wpb = 2; break;
    default: wpb = 4; break;
    }
    const uint64_t waves = (n + 63u) / 64u;
    *blocks = uint32_t((waves + wpb - 1) / wpb);
    *threads = uint32_t(64u * wpb);
}

hipError_t launch_sha1(KernelVariant v, const DeviceBatch& b, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    uint32_t blocks = 0, threads = 0;
    launch_shape(v, b.n, &blocks, &threads);
    const dim3 grid(blocks), block(threads);
    switch (int(v)) {          // 2..5: other stage depths / workgroup shapes, kept for A/B measurement
    case KERNEL_LANE: hipLaunchKernelGGL(sha1_lane_kernel, grid, block, 0, stream, b); break;
    case 2:  hipLaunchKernelGGL((sha1_lds_kernel<1, 1>), grid, block, 0, stream, b); break;
    case 3:  hipLaunchKernelGGL((sha1_lds_kernel<4, 1>), grid, block, 0, stream, b); break;
    case 4:  hipLaunchKernelGGL((sha1_lds_kernel<2, 2>), grid, block, 0, stream, b); break;
    case 5:  hipLaunchKernelGGL((sha1_lds_kernel<2, 1>), grid, block, 0, stream, b); break;
    default: hipLaunchKernelGGL((sha1_lds_kernel<2, 4>), grid, block, 0, stream, b); break;
    }
    return hipGetLastError();
}

// ----------------------------------------------------------- synthetic data --
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t ndigits_u64(uint64_t v)
{
    uint32_t d = 1;
    while (v >= 10) { v /= 10; ++d; }
    return d;
}

// One wave per object; lanes stride over the object's 16-byte chunks and
// produce header, content, 0x80, zero fill and the bit length byte by byte,
// each message word stored as a native dword (the arena format).
__global__ __launch_bounds__(256) void synth_arena_kernel(SynthSpec s)
{
    const uint32_t obj = blockIdx.x * 4u + threadIdx.x / 64u;
    if (obj >= s.n) return;
    const uint32_t lane = threadIdx.x & 63u;

    const uint64_t len = s.lens ? s.lens[obj] : s.len_each;
    uint32_t tl = 0;
    while (tl < 8 && s.type[tl]) ++tl;
    const uint32_t nd = ndigits_u64(len);
    const uint64_t hl = tl + 1 + nd + 1;                 // "<type> <len>\0"
    const uint64_t msg = hl + len;                       // framed message bytes
    const uint64_t nb = (msg + 9 + 63) / 64;
    uint64_t off;
    if (s.lens) off = s.offsets[obj];
    else { off = (uint64_t)obj * nb * 64u; if (lane == 0) s.offsets[obj] = off; }
    if (lane == 0) s.nblocks[obj] = (uint32_t)nb;
    if (off + nb * 64u > s.arena_bytes) return;          // host sized the arena; never write past it

    const uint64_t key = s.seed ^ ((s.first_index + obj) * 0xD1342543DE82EF95ull);
    const uint64_t bits = msg * 8u;
    const uint64_t total = nb * 64u;
    for (uint64_t c0 = (uint64_t)lane * 16u; c0 < total; c0 += 64u * 16u) {
        uint32_t out[4] = {0, 0, 0, 0};
        uint64_t cached_k = ~0ull, cached_w = 0;
        for (uint32_t i = 0; i < 16; ++i) {
            const uint64_t pos = c0 + i;
            uint32_t byte;
            if (pos < hl) {
                if (pos < tl) byte = (uint8_t)s.type[pos];
                else if (pos == tl) byte = ' ';
                else if (pos == hl - 1) byte = 0;
                else {                                    // decimal digit, most significant first
                    uint32_t from_right = (uint32_t)(hl - 2 - pos);
                    uint64_t v = len;
                    for (uint32_t r = 0; r < from_right; ++r) v /= 10;
                    byte = '0' + (uint32_t)(v % 10);
                }
            } else if (pos < msg) {
                const uint64_t j = pos - hl, k = j >> 3;
                if (k != cached_k) { cached_k = k; cached_w = mix64(key + (k + 1) * 0x9E3779B97F4A7C15ull); }
                byte = (uint32_t)(cached_w >> (8 * (j & 7))) & 0xFFu;
            } else if (pos == msg) byte = 0x80;
            else if (pos >= total - 8) byte = (uint32_t)(bits >> (8 * (total - 1 - pos))) & 0xFFu;
            else byte = 0;
            out[i >> 2] |= byte << (8 * (3 - (i & 3)));      // message byte 4k+j is byte 3-j of dword k
        }
        *reinterpret_cast<uint4*>(s.arena + off + c0) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

hipError_t launch_synth(const SynthSpec& s, hipStream_t stream)
{
    if (s.n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_arena_kernel, dim3((s.n + 3u) / 4u), dim3(256), 0, stream, s);
    return hipGetLastError();
}

// ------------------------------------------------------------ device binning --
__global__ __launch_bounds__(256) void iota_kernel(uint32_t* out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = i;
}

// scratch = [sorted keys, n dwords][0..n-1, n dwords][rocPRIM's own temporary storage]
static size_t sort_temp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                         static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0, 32);
    return bytes;
}

size_t bin_order_scratch_bytes(uint32_t n)
{
    const size_t lists = (size_t(n) * 8 + 255) / 256 * 256;
    return lists + sort_temp_bytes(n) + 256;
}

hipError_t bin_order_device(const uint32_t* d_nblocks, uint32_t n, uint32_t* d_order, void* scratch,
                            size_t scratch_bytes, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (scratch_bytes < bin_order_scratch_bytes(n)) return hipErrorInvalidValue;
    uint32_t* keys_out = static_cast<uint32_t*>(scratch);
    uint32_t* iota = keys_out + n;
    const size_t lists = (size_t(n) * 8 + 255) / 256 * 256;
    void* temp = static_cast<uint8_t*>(scratch) + lists;
    size_t temp_bytes = scratch_bytes - lists;
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, iota, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // a stable sort by descending key keeps equal block counts in index order
    return rocprim::radix_sort_pairs_desc(temp, temp_bytes, d_nblocks, keys_out, iota, d_order, n, 0, 32, stream);
}

// ------------------------------------------- binning into static scratch --
// Length binning without any allocation, for the entry point that may not
// allocate: a counting sort of the objects by a monotone bucket of their block
// count, longest bucket first, into a file-scope device array.
//
//   bucket(nb) = nb                                    for nb < 256
//              = 256 + 16*(log2(nb) - 8) + next 4 bits  above (16 buckets per octave)
//
// so short objects are binned exactly and long ones to within 1/16 of their
// length, which is all the wave needs (its lanes finish within ~6 % of each
// other).  Three launches: histogram (LDS-privatised per workgroup), one
// workgroup turning the histogram into start positions in descending bucket
// order, and a scatter in which every workgroup reserves one range per bucket
// (an LDS counter ranks its own objects; one global atomic per workgroup and
// bucket), so a workgroup's objects of one bucket stay together and in a batch
// of equal lengths a wave still reads neighbouring slots.
constexpr uint32_t kBins = 640;                     // 256 + 16 * (32 - 8)
constexpr uint32_t kBinItems = 4;                   // objects per thread in the histogram / scatter kernels
constexpr uint64_t kScratchStatus  = 0;             // u32 status word (STATUS_* bits)
constexpr uint64_t kScratchHist    = 256;           // u32[kBins]  histogram
constexpr uint64_t kScratchCursor  = 256 + 4096;    // u32[kBins]  next free position per bucket
constexpr uint64_t kScratchSched   = 256 + 8192;    // SchedState of the software-scheduled launch
constexpr uint64_t kScratchOrder   = kScratchSched + ((sizeof(SchedState) + 255) / 256) * 256;   // u32[STATIC_MAX_OBJECTS]
constexpr uint64_t kScratchBytes   = kScratchOrder + 4ull * STATIC_MAX_OBJECTS;
static_assert(kBins * 4 <= 4096, "histogram and cursor regions are 4 KiB each");

// The scratch of vit_hash_arena_device_static.  Each GPU holds its own copy of
// this array at its own address; host code never uses the NAME below as an
// address (it is only a host-side handle): static_scratch_address() asks the
// runtime for the current device's address, once per device, and everything
// else — kernel arguments, memsets, copies — uses that pointer.
static __device__ __align__(256) uint8_t scratch[kScratchBytes];

hipError_t static_scratch_address(void** out)
{
    return hipGetSymbolAddress(out, HIP_SYMBOL(scratch));
}

uint64_t static_scratch_bytes() { return kScratchBytes; }

__device__ __forceinline__ uint32_t bin_of(uint32_t nb)
{
    if (nb < 256u) return nb;
    const uint32_t lg = 31u - (uint32_t)__clz((int)nb);            // 8..31
    return 256u + 16u * (lg - 8u) + ((nb >> (lg - 4u)) & 15u);
}

__global__ __launch_bounds__(256) void bin_hist_kernel(const uint32_t* nblocks, uint32_t n, uint32_t* hist)
{
    __shared__ uint32_t local[kBins];
    for (uint32_t b = threadIdx.x; b < kBins; b += 256u) local[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (256u * kBinItems);
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; ++k) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        if (i < n) atomicAdd(&local[bin_of(nblocks[i])], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kBins; b += 256u)
        if (local[b]) atomicAdd(&hist[b], local[b]);
}

// One workgroup of kBins threads: cursor[b] = number of objects in buckets above b.
__global__ __launch_bounds__(kBins) void bin_start_kernel(const uint32_t* hist, uint32_t* cursor)
{
    __shared__ uint32_t part[kBins];
    const uint32_t t = threadIdx.x;                 // thread t looks at bucket kBins-1-t: descending order
    part[t] = hist[kBins - 1u - t];
    __syncthreads();
    for (uint32_t step = 1; step < kBins; step <<= 1) {            // inclusive scan, Hillis-Steele
        const uint32_t add = t >= step ? part[t - step] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    cursor[kBins - 1u - t] = part[t] - hist[kBins - 1u - t];       // exclusive
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const uint32_t* nblocks, uint32_t n, uint32_t* cursor,
                                                          uint32_t* order)
{
    __shared__ uint32_t count[kBins];
    __shared__ uint32_t start[kBins];
    for (uint32_t b = threadIdx.x; b < kBins; b += 256u) count[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (256u * kBinItems);
    uint32_t bin[kBinItems], rank[kBinItems];
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; ++k) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        bin[k] = rank[k] = 0;
        if (i < n) { bin[k] = bin_of(nblocks[i]); rank[k] = atomicAdd(&count[bin[k]], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kBins; b += 256u)
        if (count[b]) start[b] = atomicAdd(&cursor[b], count[b]);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kBinItems; ++k) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        if (i < n) {
            const uint32_t pos = start[bin[k]] + rank[k];
            if (pos < n) order[pos] = i;            // always true; keeps a corrupted count from writing past the array
        }
    }
}

hipError_t launch_sha1_static(KernelVariant v, DeviceBatch b, void* scratch_ptr, uint32_t sched_waves,
                              uint32_t sched_share_below, uint32_t tail_percent, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    if (b.n > STATIC_MAX_OBJECTS || !scratch_ptr) return hipErrorInvalidValue;
    uint8_t* s = static_cast<uint8_t*>(scratch_ptr);
    uint32_t* hist = reinterpret_cast<uint32_t*>(s + kScratchHist);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(s + kScratchCursor);
    uint32_t* order = reinterpret_cast<uint32_t*>(s + kScratchOrder);
    // The array keeps the last call's contents: the histogram must start at zero
    // (the cursors are overwritten, the order entries [0, n) all rewritten).
    hipError_t e = hipMemsetAsync(hist, 0, kBins * 4, stream);
    if (e != hipSuccess) return e;
    const uint32_t groups = (b.n + 256u * kBinItems - 1u) / (256u * kBinItems);
    hipLaunchKernelGGL(bin_hist_kernel, dim3(groups), dim3(256), 0, stream, b.nblocks, b.n, hist);
    hipLaunchKernelGGL(bin_start_kernel, dim3(1), dim3(kBins), 0, stream, hist, cursor);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(groups), dim3(256), 0, stream, b.nblocks, b.n, cursor, order);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    b.order = order;
    b.status = reinterpret_cast<uint32_t*>(s + kScratchStatus);
    if (!sched_waves) return launch_sha1(v, b, stream);
    // The scheduler's state must be all zero at launch: a launch leaves it so, but a
    // launch that was cut short (device reset, aborted stream) would not have.
    if ((e = hipMemsetAsync(s + kScratchSched, 0, sizeof(SchedState), stream)) != hipSuccess) return e;
    // The order exists on the device only: with a tail criterion both launches go out and their waves pick one.
    b.gate_percent = tail_percent;
    if (tail_percent && (e = device_simds(&b.gate_simds)) != hipSuccess) return e;
    e = launch_sha1_sched(b, reinterpret_cast<SchedState*>(s + kScratchSched), sched_waves, sched_share_below, stream);
    if (e == hipSuccess && tail_percent) e = launch_sha1(v, b, stream);
    return e;
}

// --------------------------------------------------------- compute ceiling --
__global__ __launch_bounds__(256) void valu_peak_kernel(uint32_t* out, uint32_t n_lanes, uint32_t blocks)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t h[5];
    sha1_init(h);
    uint32_t seedw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) seedw[i] = gid * 0x9E3779B9u + i * 0x85EBCA6Bu;
    for (uint32_t bl = 0; bl < blocks; ++bl) {
        uint32_t w[16];
        // Feed the previous digest back in so nothing is loop invariant.
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = seedw[i] ^ h[i % 5];
        sha1_compress(h, w);
    }
    if (gid < n_lanes) out[gid] = h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4];
}

hipError_t launch_valu_peak(uint32_t* out, uint32_t n_lanes, uint32_t blocks_per_lane, hipStream_t stream)
{
    if (n_lanes == 0) return hipSuccess;
    hipLaunchKernelGGL(valu_peak_kernel, dim3((n_lanes + 255u) / 256u), dim3(256), 0, stream, out, n_lanes,
                       blocks_per_lane);
    return hipGetLastError();
}

}  // namespace vit
