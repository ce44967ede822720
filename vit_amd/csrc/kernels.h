// kernels.h — launch interface between the C-ABI host code and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vit {

// Which SHA-1 kernel a launch uses.
enum KernelVariant : int {
    KERNEL_LDS  = 0,   // wave-cooperative coalesced loads staged through LDS (default)
    KERNEL_LANE = 1,   // each lane loads its own blocks straight from global memory
};

// A packed arena in device memory.  Object i occupies nblocks[i] 64-byte blocks
// starting at arena + offsets[i]; they hold the framed message
// ("<type> <len>\0" + content) followed by SHA-1 padding (0x80, zeros, 64-bit
// big-endian bit length), every 4-byte message word stored as a native
// (little-endian) dword, so the kernel hashes whole blocks, swaps no bytes and
// does nothing else.
struct DeviceBatch {
    const uint8_t*  arena;        // device, 16-byte aligned
    uint64_t        arena_bytes;  // bytes readable from arena
    const uint64_t* offsets;      // device, n entries, each a multiple of 16 (the packer uses 64)
    const uint32_t* nblocks;      // device, n entries
    const uint32_t* order;        // device, n entries or nullptr: slot s hashes object order[s]
    uint32_t        n;
    uint8_t*        digests;      // device, n * 20 bytes, indexed by object
    uint32_t*       status;       // device word, OR-ed with STATUS_* bits on bad input
    uint64_t*       wave_trace;   // diagnostics, normally nullptr: 4 words per wave (start and end of the wave on the
                                  // 100 MHz wall clock, hardware id, longest object in blocks)
    // Launch gate, for batches whose lengths the host never saw.  gate_percent == 0: the launch runs.  Otherwise
    // the SAME batch is launched twice, through the software-scheduled kernel and through the plain one, and every
    // wave of both evaluates tail_dominates() on the batch's first (longest) object: the scheduled launch runs only
    // if it holds, the plain one only if it does not; the waves of the other leave at once.
    uint32_t        gate_percent;
    uint32_t        gate_simds;
};

// Does the batch have a tail worth scheduling for?  The scheduled launch keeps ONE wave per SIMD busy, which costs
// about 1.32 us per 64-byte block and wave where five waves sharing the SIMD cost 1.05 us between them, but it never
// lets a long group share its SIMD; under the hardware dispatcher a long group does, and the launch then lasts about
// 1.7 us per block of the longest object.  With W = the mean work of a SIMD in wave-blocks (arena bytes / 64 bytes /
// 64 lanes / SIMDs) the scheduled launch therefore wins once the longest object exceeds about 0.78 W; `percent` is
// that ratio (default 75).
__host__ __device__ inline bool tail_dominates(uint64_t longest_blocks, uint64_t arena_bytes, uint32_t simds, uint32_t percent)
{
    return longest_blocks * 4096ull * simds * 100ull >= arena_bytes * percent;
}

enum : uint32_t {
    STATUS_OUT_OF_BOUNDS = 1u,    // offsets[i] + 64*nblocks[i] > arena_bytes
    STATUS_MISALIGNED    = 2u,    // offsets[i] not a multiple of 16
    STATUS_BAD_ORDER     = 4u,    // order[s] >= n
};

hipError_t launch_sha1(KernelVariant v, const DeviceBatch& b, hipStream_t stream);
// Launch shape for n objects (n up to 2^32 - 1): workgroups, threads per workgroup.
void launch_shape(KernelVariant v, uint64_t n, uint32_t* blocks, uint32_t* threads);

// The software-scheduled launch for batches in binned order (b.order lists the objects longest first; without
// an order the batch itself is taken to be in that order): a fixed grid of resident waves pulls 64-object
// groups down that order, one leader wave per SIMD always, the SIMD's other waves only while the leader is on
// a group of at most share_below blocks (see kernels.hip).  `state` is device memory, all zero before the
// first launch that uses it; a launch leaves it all zero again, so launches that follow each other on a
// stream may share one, launches that overlap may not.
constexpr uint32_t SCHED_SIMD_KEYS = 8192;
struct SchedState {
    uint32_t next;                        // next group to hand out
    uint32_t exited;                      // waves that have left
    uint32_t pad[14];
    uint32_t simd_waves[SCHED_SIMD_KEYS]; // waves registered on each physical SIMD
    uint32_t simd_len[SCHED_SIMD_KEYS];   // 1 + length in blocks of the group that SIMD's leader is on; 0 = none
};
hipError_t launch_sha1_sched(const DeviceBatch& b, SchedState* state, uint32_t waves_per_simd, uint32_t share_below,
                             hipStream_t stream);

// Synthetic framed arena, generated on the device (benchmark and full-size
// parity inputs; the reference has no benchmark data of its own).
// Content byte j of object i is byte j of the counter-based splitmix64 stream
// keyed by (seed, first_index + i).  lens == nullptr: every object has len_each
// bytes and offsets/nblocks are written by the kernel (slot = nblocks*64,
// objects back to back); otherwise lens and offsets are inputs.
struct SynthSpec {
    uint64_t        seed;
    uint64_t        first_index;
    uint32_t        n;
    uint64_t        len_each;
    const uint64_t* lens;         // device or nullptr
    uint64_t*       offsets;      // device: output if lens == nullptr, else input
    uint32_t*       nblocks;      // device output
    uint8_t*        arena;        // device output
    uint64_t        arena_bytes;
    char            type[8];      // "blob", "tree", "commit" (NUL padded)
};
hipError_t launch_synth(const SynthSpec& s, hipStream_t stream);

// Length binning on the device: order[] = object indices by block count, longest first, ties in
// index order (the same permutation the host routine gives), for arenas whose block counts never
// were in host memory.  scratch: device memory of at least bin_order_scratch_bytes(n).
size_t bin_order_scratch_bytes(uint32_t n);
hipError_t bin_order_device(const uint32_t* d_nblocks, uint32_t n, uint32_t* d_order, void* scratch,
                            size_t scratch_bytes, hipStream_t stream);

// The launch of the entry point that allocates nothing: length-bins the batch on the device into a
// file-scope device array (about 8 MiB per GPU, held for the life of the process), then hashes in that
// order; status bits go to a word in the same array.  b.order and b.status are ignored.  scratch_ptr is
// what static_scratch_address() returned ON THE DEVICE THE LAUNCH GOES TO.
constexpr uint32_t STATIC_MAX_OBJECTS = 1u << 21;
hipError_t static_scratch_address(void** out);        // hipGetSymbolAddress for the current device
uint64_t static_scratch_bytes();
// sched_waves != 0: the scheduled launch may be used; tail_percent != 0: the waves decide (DeviceBatch::gate_percent).
hipError_t launch_sha1_static(KernelVariant v, DeviceBatch b, void* scratch_ptr, uint32_t sched_waves,
                              uint32_t sched_share_below, uint32_t tail_percent, hipStream_t stream);
// SIMDs of the current device (compute units x 4).
hipError_t device_simds(uint32_t* out);

// The zlib streams of the objects of an arena (what the reference's compress() call writes for each,
// src/commit.cpp:98-105), one object per lane; see deflate_kernels.hip.  Object i's stream goes to
// out + out_off[i] (room for deflate_bound(framed length) bytes) and its length to out_len[i]; 0 there
// means the object is not the kernel's to do (framed length above deflate_max_input(), or bad input:
// status word) and is left to the host.  workspace: deflate_workspace_bytes(lanes) bytes of device
// memory, lanes a multiple of 64: the objects in flight at once.
struct DeflateBatch {
    const uint8_t*  arena; uint64_t arena_bytes;
    const uint64_t* offsets; const uint32_t* nblocks; uint32_t n;
    const uint32_t* order;        // optional: slot s deflates object order[s] (objects sorted by length, so that a wave's 64 end together)
    uint8_t*        out; const uint64_t* out_off; uint32_t* out_len;
    uint8_t*        workspace;    // the wave kernels' scratch (see launch_deflate_wave)
    const void*     tables_raw;   // device copy of what deflate_make_tables() wrote
    uint32_t*       status;
    uint32_t        max_input;    // objects of more framed bytes are nobody's here (their out_len stays as it is)
    uint64_t*       trace;        // diagnostics, normally nullptr: 14 counters.  0-7: the workgroup-per-object kernels add their serial
                                  // wave's shader-clock cycles by phase (load + checksum, chains, waiting for the searches, decisions,
                                  // symbols to scratch, the group's blocks by lanes, stored bytes; last: objects).  8-13: one searching
                                  // wave's cycles while a tile's positions are handed out and after, rounds of either, cycles at the
                                  // tiles' barriers, chain entries looked at after the tile was handed out
};
uint64_t deflate_workspace_bytes(uint32_t lanes);
uint64_t deflate_tables_bytes();
void deflate_make_tables(void* host_dst);
uint32_t deflate_bound(uint32_t framed_len);
uint32_t deflate_max_input();
// Two kernels share the work.  launch_deflate_wave: one object per workgroup with its tables in LDS, for every object
// of up to deflate_wave_max_input() framed bytes (one deflate block); DeflateBatch::workspace is then
// deflate_wave_workspace_bytes() of scratch (symbols and tree arrays of the objects in flight).  launch_deflate: one
// object per LANE with its tables in lane_workspace (deflate_workspace_bytes(lanes)), for the objects longer than
// only_above.  Each writes out_len only for the objects it does (and 0 for a slot with bad input); out_len must be
// zero beforehand for the rest to read as "the host's".
hipError_t launch_deflate_wave(const DeflateBatch& b, hipStream_t stream);
uint32_t deflate_wave_max_input();
hipError_t deflate_wave_workspace_bytes(uint64_t* out);
hipError_t launch_deflate(DeflateBatch b, uint8_t* lane_workspace, uint32_t lanes, uint32_t only_above, hipStream_t stream);

// Compute-only ceiling: every lane runs `blocks_per_lane` SHA-1 compressions on
// register-resident data (no loads in the loop).  out: one dword per lane.
hipError_t launch_valu_peak(uint32_t* out, uint32_t n_lanes, uint32_t blocks_per_lane,
                            hipStream_t stream);

}  // namespace vit
