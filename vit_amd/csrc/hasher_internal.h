// hasher_internal.h — what the translation units behind include/vit_hash.h share:
// the hasher's state and the helpers that more than one of them calls.
//
//   hasher.cpp         lifetime, options, counters, status word, pinned arenas
//   pipeline.cpp       staging sets, chunks, device queues, the sharding of a batch
//                      (vit_hash_objects, vit_hash_arena, vit_hash_arena_device ...)
//   balance.cpp        the rates a batch is balanced with: calibration on this
//                      machine, what N queues share, NUMA placement, what a call teaches
//   deflate_host.cpp   host side of the deflate kernel (vit_deflate_*)
//   bench_support.cpp  what include/vit_hash_bench.h declares: generators, timers,
//                      device memory for callers without a HIP binding
#pragma once
#include "../../include/vit_hash.h"
#include "../../include/vit_hash_bench.h"
#include "host_util.h"
#include "kernels.h"
#include "packer.h"
#include "sha1_host.h"
#include "work_pool.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace vit {

using clk = std::chrono::steady_clock;
inline uint64_t ns_since(clk::time_point t0)
{
    return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count());
}

struct Staging {
    uint8_t*  h_arena = nullptr;  uint64_t h_cap = 0;     // pinned host arena
    uint8_t*  d_arena = nullptr;  uint64_t d_cap = 0;     // device arena
    uint64_t* h_off = nullptr;    uint32_t* h_nb = nullptr;  uint32_t* h_order = nullptr;  uint8_t* h_dig = nullptr;
    uint64_t* d_off = nullptr;    uint32_t* d_nb = nullptr;  uint32_t* d_order = nullptr;  uint8_t* d_dig = nullptr;
    uint8_t*  h_meta = nullptr;   uint8_t* d_meta = nullptr;   // the two allocations the eight pointers above point into
    uint64_t  m_cap = 0;                                  // objects the meta buffers hold
    uint32_t* h_status = nullptr;                         // pinned word behind the meta arrays: the queue's status word
                                                          // as it stood after this chunk's kernel
    hipEvent_t done = nullptr;
    hipEvent_t mark[4] = {nullptr, nullptr, nullptr, nullptr};   // trace only: before copy, after copy, after kernel, after read-back
    double host_ms[4] = {0, 0, 0, 0};                            // trace only: submit begin, previous chunk retired, packed, queued
    bool in_flight = false, traced = false;
    std::vector<uint64_t> obj;                            // caller indices of the chunk in flight; when empty, the chunk is
    uint64_t first = 0, count = 0;                        // caller objects [first, first + count)
};

// What the static entry point's launches on one GPU share, whatever hasher or
// queue they come through: the file-scope scratch array of that GPU and the
// event that orders them (see static_device in pipeline.cpp).
struct StaticDevice {
    void*      scratch = nullptr;                         // this DEVICE's address of the kernels' file-scope scratch array
    hipEvent_t done = nullptr;                            // end of the last launch that used it
    std::mutex order;                                     // held from the wait on `done` to its next record
};

struct Queue {
    int device = 0;
    int numa_node = -1;                                    // NUMA node of the GPU's PCIe root (-1: unknown or one node)
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};   // [0],[1]: alternating chunks; [2]: long objects
    uint32_t* d_status = nullptr;                          // device: word 0 = STATUS_* bits; bytes 1024.. = calibration output
    uint32_t* h_status = nullptr;                          // pinned: where vit_device_status reads that word to
    uint32_t  status_seen = 0;                             // bits the chunks of host-memory calls have brought back
    vit::SchedState* d_sched = nullptr;                    // device: kSchedStates states of the software-scheduled launch,
    uint32_t  sched_next = 0;                              // used in turn (a launch leaves its state all zero again)
    uint32_t  simds = 1024;                                // SIMDs of the device
    StaticDevice* shared = nullptr;                        // the device's entry in the process-wide table (never freed)
    uint64_t  chunks_this_call = 0;
    bool      grew_this_call = false;                      // a pinned staging arena was (re)allocated during the call
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    void* sort_scratch = nullptr; size_t sort_scratch_bytes = 0;   // vit_bin_order_device
    uint8_t* z_workspace = nullptr; uint32_t z_lanes = 0;          // the lane deflate kernel's tables of the objects in flight
    uint8_t* z_scratch = nullptr;                                  // the wave deflate kernels' scratch
    void* z_tables = nullptr;                                      // ... and zlib's constant tables
    hipEvent_t z_done = nullptr;                                   // ... and the end of the last launch that used them: the next one, on
                                                                   // whatever stream, waits for it (one workspace per queue)
    struct StreamSet {                                             // vit_deflate_objects, per chunk in flight (two alternate):
        uint8_t* d_out = nullptr; uint64_t d_out_cap = 0;          // the chunk's streams on the device
        uint8_t* h_out = nullptr; uint64_t h_out_cap = 0;          // ... and in pinned memory, when the caller's buffer is not pinned
        uint8_t* meta_d = nullptr; uint8_t* meta_h = nullptr; uint64_t meta_cap = 0;   // ... their offsets and lengths (12 bytes per object)
    } z[2];
    std::vector<hipEvent_t> timing;                       // pool for vit_time_hash_device
    uint64_t* wave_trace = nullptr;                       // diagnostics: device buffer, 32 bytes per wave, that the next
                                                          // launches fill with their waves' timeline (tools/wave_timeline.py)
    std::string trace_log;                                // per-chunk timeline of the last call, when tracing
    clk::time_point trace_t0;
    Staging st[3];
    std::string err;                                      // set by the queue's worker thread
};

constexpr uint64_t kStatusBytes = 4096;                   // allocation behind Queue::d_status
constexpr uint64_t kCalibOut = 1024;                      // byte offset of the calibration kernel's output in it
// Scheduler states per queue.  A launch owns its state until it ends.  The three
// staging sets have one each (a set has at most one launch in flight); launches
// on device-resident arenas take the others in turn, so two of those would meet
// in one state only with more than kSchedStates - 3 of them in flight at once.
constexpr uint32_t kSchedStates = 16;
// Objects of at least this many blocks (256 KiB) are "long": a lane hashes about
// 50 MB/s, so each of them holds its wave for milliseconds.  They are launched
// first, on a stream and staging set of their own, and the short objects'
// chunks stream past them on the other two.
constexpr uint32_t kLongBlocks = 4096;

// What the caller set (vit_hasher_set_option); nothing here changes during a call.
struct Options {
    int kernel = VIT_KERNEL_LDS;
    // Which objects go to host threads instead of a GPU lane: 0 never, 1 auto
    // (balanced per batch, see hash_batch), 2 fixed threshold.
    int host_policy = 1;
    uint64_t host_threshold = 0;
    uint64_t host_min_bytes = vit::kHostSplitMin;        // auto: objects shorter than this stay on the GPU
    uint64_t chunk_bytes = 64ull << 20;
    uint64_t chunk_object_multiple = 8192;
    bool trace = false;
    // Batches in binned order of at least sched_min_groups 64-object groups go through the
    // software-scheduled launch (kernels.hip) when the LDS kernel is selected; 0 waves = off.
    uint32_t sched_waves = 1;                             // resident waves per SIMD of that launch (1 or 2)
    uint32_t sched_min_groups = 4096;
    uint32_t sched_share_below = 128;                     // blocks: a SIMD's spare waves work only beside a leader group this short
    uint32_t sched_tail_percent = 75;                     // ... and only if the longest object is at least this share of a SIMD's
                                                          // mean work (vit::tail_dominates); 0 = whatever the lengths
    int pack_threads = 8;                                 // the host thread budget of a call, helpers included
    bool numa_pin = true;                                 // packing tasks prefer threads on the NUMA node of their queue's GPU
    uint64_t fail_at_chunk = 0;                           // fault injection for tests: 0 = off
    bool fail_deflate = false;                            // option debug_fail_deflate: vit_deflate_objects fails (tests)
    bool corrupt_deflate = false;                         // option debug_corrupt_deflate: one byte of the first stream of every
                                                          // vit_deflate_objects call is flipped (tests of the caller's checks)
    uint64_t deflate_max_bytes = 16382;                   // longer objects (framed) are left to the host; up to this default a wave
                                                          // deflates an object in LDS, beyond it a lane with tables in device memory
    uint64_t deflate_chunk_bytes = 64ull << 20;           // vit_deflate_objects: packed bytes per chunk (two chunks are in flight)
    uint32_t deflate_lanes = 4096;                        // objects the LANE kernel keeps in flight (181 KiB of device memory each)
};

// Rates the auto policy balances with, bytes per second; 0 = not measured yet.
// *_fixed: set through an option, never measured.
struct Rates {
    double lane = 0, core = 0, copy = 0, pack = 0;
    double core_multi = 0;                                // one core, kHostStreams objects interleaved (sha1_host.h); <= core: not used
    double copy_total = 0;                                // every queue copying at once: what they get TOGETHER (one queue: = copy)
    bool lane_fixed = false, core_fixed = false, copy_fixed = false, pack_fixed = false, core_multi_fixed = false,
         copy_total_fixed = false;
    bool calibrated = false, host_calibrated = false;
    double host_bias = 1;                                 // see learn_from_call
    uint64_t calibrate_ns = 0;
};

struct Counters {
    std::atomic<uint64_t> objects_gpu{0}, objects_host{0}, bytes_host{0}, blocks_gpu{0}, bytes_h2d{0}, launches{0};
};

// Of the last call.
struct LastCall {
    uint64_t host_threshold = 0, host_objects = 0;
    uint64_t kernel_ns = 0, pack_ns = 0, total_ns = 0;
    uint64_t plan_ns = 0;                                 // its serial preamble: split plan, block counts, sharding
    uint64_t gpu_done_ns = 0;                             // when its device queues had finished, from its start
    std::atomic<uint64_t> host_done_ns{0};                // ... and when its last host-lane task had
};

}  // namespace vit

struct vit_hasher {
    std::vector<vit::Queue> queues;
    mutable std::string err;
    vit::Options opt;
    vit::Rates rates;
    vit::Counters count;
    vit::LastCall last;
    std::unique_ptr<vit::WorkPool> pool;
    std::vector<int> worker_node;                         // NUMA node each pool worker is pinned to (-1: not pinned)
};

namespace vit {

// Record the text (per hasher; h == nullptr: per thread, for vit_hasher_create) and return the code.
int fail(const vit_hasher* h, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return vit::fail((h), e_ == hipErrorOutOfMemory ? VIT_ERR_NOMEM : VIT_ERR_HIP,    \
                             "%s failed: %s", #expr, hipGetErrorString(e_));                  \
    } while (0)

// Same, inside a queue worker: record the text on the queue, return the code.
#define HIP_TRYQ(q, expr)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (q).err = std::string(#expr) + " failed: " + hipGetErrorString(e_);               \
            return e_ == hipErrorOutOfMemory ? VIT_ERR_NOMEM : VIT_ERR_HIP;                   \
        }                                                                                     \
    } while (0)

// Nothing may propagate out of an extern "C" entry point.
#define VIT_NOEXCEPT_BEGIN try {
#define VIT_NOEXCEPT_END(h)                                                                         \
    } catch (const std::bad_alloc&) { return vit::fail((h), VIT_ERR_NOMEM, "out of host memory"); } \
      catch (const std::exception& e_) { return vit::fail((h), VIT_ERR_HIP, "internal error: %s", e_.what()); }

// Joins on scope exit, so an early return or an exception never leaves a
// running thread behind.
struct ThreadGroup {
    std::vector<std::thread> t;
    ~ThreadGroup() { for (auto& x : t) if (x.joinable()) x.join(); }
    void join() { for (auto& x : t) if (x.joinable()) x.join(); }
};

inline Queue* queue_of(vit_hasher* h, int dev)
{
    if (!h || dev < 0 || size_t(dev) >= h->queues.size()) return nullptr;
    return &h->queues[size_t(dev)];
}

inline uint64_t round_up(uint64_t v, uint64_t q) { return (v + q - 1) / q * q; }

// hasher.cpp
int pool_threads_for(const vit_hasher* h);
void restart_pool(vit_hasher* h);                         // (re)start the pool's workers for the current budget and placement
int read_status_word(vit_hasher* h, Queue& q, uint32_t* out);

// pipeline.cpp
uint64_t grown(uint64_t have, uint64_t bytes);
int ensure_device_arena(Queue& q, Staging& s, uint64_t bytes);
int ensure_arena(Queue& q, Staging& s, uint64_t bytes);
int ensure_meta(Queue& q, Staging& s, uint64_t n);
uint64_t carve_meta(Staging& s, uint64_t n);
void free_staging(Staging& s);
DeviceBatch make_batch(const Queue& q, const void* d_arena, uint64_t arena_bytes, const uint64_t* d_off,
                       const uint32_t* d_nb, const uint32_t* d_order, uint64_t n, uint8_t* d_dig);
void drain(Queue& q);
void shard(const std::vector<uint64_t>& idx, const uint32_t* nb, size_t nq, std::vector<std::vector<uint64_t>>& out);
StaticDevice* static_device(int device, hipError_t* err);   // with `device` current

// balance.cpp
void calibrate_host(vit_hasher* h);
int calibrate(vit_hasher* h, const uint8_t* sample, uint64_t sample_len, const uint8_t* const* data = nullptr,
              const uint64_t* len_all = nullptr, uint64_t n_all = 0);
// After a call: what it measured corrects the rates the next one balances with.
void learn_from_call(vit_hasher* h, const uint64_t* len, const std::vector<uint64_t>& gpu_idx, size_t host_objects);
// NUMA node of a HIP device's PCIe function, from sysfs (-1: unknown); node_cpus: the CPUs of a node.
int device_numa_node(int device);
int node_of_address(const void* p);                      // NUMA node a page lies on (-1: unknown)
// The pool task a packing piece of queue q should be: pieces are handed out so that
// a worker pinned to the queue's node takes them first (see WorkPool::run_on).
int preferred_node(const vit_hasher* h, const Queue& q);

}  // namespace vit
