// packer.h — host-side arena packing primitives (see packer.cpp).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace vit {

// strlen(type) if type is a usable object type name (1..8 bytes, no space), else 0.
uint32_t type_len_checked(const char* type);
// "<type> <len>\0" into out; returns the header length.  type_len 0 = a raw
// buffer without Git framing: nothing is written, 0 is returned.
uint32_t write_header(const char* type, uint32_t type_len, uint64_t len, uint8_t* out);
// Header + content + SHA-1 padding into a slot of nblocks*64 bytes.
void pack_one(const char* type, uint32_t type_len, const uint8_t* data, uint64_t len, uint8_t* slot,
              uint64_t nblocks);
// pack_one writes with non-temporal stores: each packing thread calls this once
// after its last slot, before another agent (a copy engine) reads the arena.
void pack_fence();
// order[] = indices by block count, longest first, stable; false if all equal.
bool bin_order(const uint32_t* nblocks, uint64_t n, uint32_t* order);

// Which objects of a batch leave the GPU for host threads.  One SHA-1 stream is
// serial: a GPU lane sustains `lane` bytes/s and a host core `core`, while a
// whole batch moves through pack + PCIe at 0.8 x `copy` per queue (the pipeline's
// measured efficiency) and no faster than `threads` cores pack at `pack` each.
// Starting from the longest object, an object of kHostSplitMin bytes or more
// moves to the host threads as long as the host side would still finish before
// the GPU side would with the object on it:
//   GPU side:  its bytes through the queues, never less than its longest object
//              on one lane;
//   host side: the thread-seconds of hashing the moved bytes AND of packing the
//              GPU side's bytes, over the thread budget (one pool carries both),
//              never less than the longest moved object on one core.
// Objects move one by one in (length, index) order, so a run of equal lengths
// may be split between the two sides.  out: the moved indices, longest first.
// core_multi (0 = none): what one core gets through when it runs kHostStreams
// objects' SHA-1 streams interleaved (sha1_host.h); the host side counts on it
// in proportion to the moved objects per thread (one object per thread: `core`,
// four or more: `core_multi`).
// min_bytes: objects shorter than this never move (default kHostSplitMin).
// host_bias: factor on the host side's estimated time (1 = the model as it is; the hasher learns
// it from when the two sides of its earlier calls finished).
// copy_total (0 = none measured): what all the queues get when they copy at once; the GPU side
// drains at 0.8 x min(queues x copy, copy_total).
struct SplitRates { double lane, core, copy, pack; double core_multi = 0; uint64_t min_bytes = 16u << 10; double host_bias = 1; double copy_total = 0; };
constexpr uint64_t kHostSplitMin = 16u << 10;
// parallel_for (optional): runs fn(0) .. fn(parts - 1) on the caller's threads and returns when all have; the
// scan of a large batch for candidates is cut into parts with it.
using ParallelFor = std::function<void(size_t parts, const std::function<void(size_t)>& fn)>;
void plan_host_split(const uint64_t* len, uint64_t n, const SplitRates& r, int queues, int threads,
                     std::vector<uint64_t>& out, const ParallelFor* parallel_for = nullptr);

}  // namespace vit
