// hasher.cpp — the C ABI of include/vit_hash.h over the gfx950 kernels.
//
// Host pipeline of vit_hash_objects(), per device queue:
//
//   pack chunk k+1 into pinned arena B   |  H2D copy, kernel, digest D2H of
//   (header, content, padding; pool)     |  chunk k from pinned arena A (stream)
//
// Two staging sets per queue alternate, so host packing overlaps the copy
// engine and the kernel; a third set and stream carry the chunks of long
// objects, and the longest objects of a skewed batch are hashed by host
// threads meanwhile (see hash_batch).  Objects are assigned to queues by
// padded size (bytes), not by count; nothing is exchanged between devices
// because every digest depends on its own object only and the consumer of the
// digests (tree assembly, reference src/commit.cpp:164-172) runs on the host.
//
// Host threads: ONE pool per hasher (work_pool.h) serves the packing of every
// queue's chunks (high priority) and the host-lane objects (low priority); a
// queue's own thread only submits work and helps with its own packing.
#include "../../include/vit_hash.h"
#include "host_util.h"
#include "kernels.h"
#include "packer.h"
#include "sha1_host.h"
#include "work_pool.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

using clk = std::chrono::steady_clock;
inline uint64_t ns_since(clk::time_point t0)
{
    return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count());
}

thread_local std::string g_create_error;

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

struct Queue {
    int device = 0;
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};   // [0],[1]: alternating chunks; [2]: long objects
    uint32_t* d_status = nullptr;                          // device: word 0 = STATUS_* bits; bytes 1024.. = calibration output
    uint32_t* h_status = nullptr;                          // pinned: where vit_device_status reads that word to
    uint32_t  status_seen = 0;                             // bits the chunks of host-memory calls have brought back
    vit::SchedState* d_sched = nullptr;                    // device: kSchedStates states of the software-scheduled launch,
    uint32_t  sched_next = 0;                              // used in turn (a launch leaves its state all zero again)
    uint32_t  simds = 1024;                                // SIMDs of the device
    void*     scratch = nullptr;                           // this DEVICE's address of the kernels' file-scope scratch array
                                                           // (vit_hash_arena_device_static); shared by every queue on the device
    uint64_t  chunks_this_call = 0;
    bool      grew_this_call = false;                      // a pinned staging arena was (re)allocated during the call
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    void* sort_scratch = nullptr; size_t sort_scratch_bytes = 0;   // vit_bin_order_device
    uint8_t* z_workspace = nullptr; uint32_t z_lanes = 0;          // vit_deflate_arena_device: tables of the objects in flight
    void* z_tables = nullptr;                                      // ... and zlib's constant tables
    hipEvent_t z_done = nullptr;                                   // ... and the end of the last launch that used them: the next one, on
                                                                   // whatever stream, waits for it (one workspace per queue)
    uint8_t* z_out = nullptr; uint64_t z_out_cap = 0;              // vit_deflate_objects: a chunk's streams on the device
    uint8_t* z_meta_d = nullptr; uint8_t* z_meta_h = nullptr; uint64_t z_meta_cap = 0;   // ... their offsets and lengths (12 bytes per object)
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

}  // namespace

struct vit_hasher {
    std::vector<Queue> queues;
    mutable std::string err;
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
    std::unique_ptr<vit::WorkPool> pool;
    // Rates the auto policy balances with, bytes per second; 0 = not measured
    // yet.  *_fixed: set through an option, never measured.
    double lane_rate = 0, core_rate = 0, copy_rate = 0, pack_rate = 0;
    double core_rate_multi = 0;                           // one core, kHostStreams objects interleaved (sha1_host.h); <= core_rate: not used
    bool lane_fixed = false, core_fixed = false, copy_fixed = false, pack_fixed = false, core_multi_fixed = false;
    bool calibrated = false, host_calibrated = false;
    double host_bias = 1;                                 // see the end of hash_batch
    uint64_t calibrate_ns = 0;
    uint64_t fail_at_chunk = 0;                           // fault injection for tests: 0 = off
    bool fail_deflate = false;                            // option debug_fail_deflate: vit_deflate_objects fails (tests)
    uint64_t deflate_max_bytes = 16384;                   // vit_deflate_objects: longer objects (framed) are left to the host
    uint32_t deflate_lanes = 65536;                       // objects the deflate kernel keeps in flight (181 KiB of device memory each)
    std::atomic<uint64_t> objects_gpu{0}, objects_host{0}, bytes_host{0}, blocks_gpu{0}, bytes_h2d{0}, launches{0};
    uint64_t last_host_threshold = 0, last_host_objects = 0;
    uint64_t last_kernel_ns = 0, last_pack_ns = 0, last_total_ns = 0;
    uint64_t last_plan_ns = 0;                            // the last call's serial preamble: split plan, block counts, sharding
    uint64_t last_gpu_done_ns = 0;                        // when the device queues of the last call had finished, from its start
    std::atomic<uint64_t> last_host_done_ns{0};           // ... and when its last host-lane task had
};

namespace {

int fail(const vit_hasher* h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((h), e_ == hipErrorOutOfMemory ? VIT_ERR_NOMEM : VIT_ERR_HIP,         \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                       \
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

// Joins on scope exit, so an early return or an exception never leaves a
// running thread behind.
struct ThreadGroup {
    std::vector<std::thread> t;
    ~ThreadGroup() { for (auto& x : t) if (x.joinable()) x.join(); }
    void join() { for (auto& x : t) if (x.joinable()) x.join(); }
};

// Nothing may propagate out of an extern "C" entry point.
#define VIT_NOEXCEPT_BEGIN try {
#define VIT_NOEXCEPT_END(h)                                                                    \
    } catch (const std::bad_alloc&) { return fail((h), VIT_ERR_NOMEM, "out of host memory"); }  \
      catch (const std::exception& e_) { return fail((h), VIT_ERR_HIP, "internal error: %s", e_.what()); }

Queue* queue_of(vit_hasher* h, int dev)
{
    if (!h || dev < 0 || size_t(dev) >= h->queues.size()) return nullptr;
    return &h->queues[size_t(dev)];
}

uint64_t round_up(uint64_t v, uint64_t q) { return (v + q - 1) / q * q; }

// Worker threads of the shared pool: the thread budget less the threads that
// help from outside it (the caller with one queue, the queue threads with more).
int pool_threads_for(const vit_hasher* h)
{
    const int helpers = h->queues.size() > 1 ? int(h->queues.size()) : 1;
    return std::max(1, h->pack_threads - helpers);
}

// Staging capacity: freeing and re-allocating costs milliseconds, so start at
// 8 MiB and at least double.
uint64_t grown(uint64_t have, uint64_t bytes)
{
    const uint64_t want = std::max<uint64_t>({bytes, 8ull << 20, std::min<uint64_t>(2 * have, 1ull << 30)});
    return round_up(want, 1ull << 20);
}

int ensure_device_arena(Queue& q, Staging& s, uint64_t bytes)
{
    if (bytes <= s.d_cap) return VIT_OK;
    const uint64_t cap = grown(s.d_cap, bytes);
    if (s.d_arena) { HIP_TRYQ(q, hipFree(s.d_arena)); s.d_arena = nullptr; s.d_cap = 0; }
    HIP_TRYQ(q, hipMalloc(reinterpret_cast<void**>(&s.d_arena), cap));
    s.d_cap = cap;
    return VIT_OK;
}

int ensure_arena(Queue& q, Staging& s, uint64_t bytes)
{
    if (bytes > s.h_cap) {
        const uint64_t cap = grown(s.h_cap, bytes);
        if (s.h_arena) { HIP_TRYQ(q, hipHostFree(s.h_arena)); s.h_arena = nullptr; s.h_cap = 0; }
        HIP_TRYQ(q, hipHostMalloc(reinterpret_cast<void**>(&s.h_arena), cap, hipHostMallocDefault));
        s.h_cap = cap;
        q.grew_this_call = true;
    }
    return ensure_device_arena(q, s, bytes);
}

// Offsets, block counts, order and digests of a chunk live in ONE pinned and ONE
// device allocation (36 bytes per object): allocations cost about a millisecond
// each, which is most of a small first call.  The pinned one ends with the word
// the chunk's copy of the device status lands in.
int ensure_meta(Queue& q, Staging& s, uint64_t n)
{
    if (n <= s.m_cap) return VIT_OK;
    const uint64_t cap = round_up(n < 4096 ? 4096 : n, 4096);
    if (s.h_meta) { HIP_TRYQ(q, hipHostFree(s.h_meta)); s.h_meta = nullptr; }
    if (s.d_meta) { HIP_TRYQ(q, hipFree(s.d_meta)); s.d_meta = nullptr; }
    s.m_cap = 0;
    HIP_TRYQ(q, hipHostMalloc(reinterpret_cast<void**>(&s.h_meta), cap * 36 + 64, hipHostMallocDefault));
    HIP_TRYQ(q, hipMalloc(reinterpret_cast<void**>(&s.d_meta), cap * 36));
    s.h_status = reinterpret_cast<uint32_t*>(s.h_meta + cap * 36);
    *s.h_status = 0;
    s.m_cap = cap;
    return VIT_OK;
}

// Lay the four arrays out for a chunk of n objects, back to back, so that
// offsets + block counts (+ order) travel to the device in one copy.
// Returns the bytes of that copy without the order array; with it, 4*m more.
uint64_t carve_meta(Staging& s, uint64_t n)
{
    const uint64_t m = round_up(n ? n : 1, 64);
    auto carve = [m](uint8_t* base, uint64_t*& off, uint32_t*& nb, uint32_t*& order, uint8_t*& dig) {
        off   = reinterpret_cast<uint64_t*>(base);
        nb    = reinterpret_cast<uint32_t*>(base + m * 8);
        order = reinterpret_cast<uint32_t*>(base + m * 12);
        dig   = base + m * 16;
    };
    carve(s.h_meta, s.h_off, s.h_nb, s.h_order, s.h_dig);
    carve(s.d_meta, s.d_off, s.d_nb, s.d_order, s.d_dig);
    return m * 12;
}

void free_staging(Staging& s)
{
    if (s.h_arena) (void)hipHostFree(s.h_arena);
    if (s.d_arena) (void)hipFree(s.d_arena);
    if (s.h_meta) (void)hipHostFree(s.h_meta);
    if (s.d_meta) (void)hipFree(s.d_meta);
    if (s.done) (void)hipEventDestroy(s.done);
    for (hipEvent_t m : s.mark) if (m) (void)hipEventDestroy(m);
    s = Staging();
}

vit::DeviceBatch make_batch(const Queue& q, const void* d_arena, uint64_t arena_bytes, const uint64_t* d_off,
                            const uint32_t* d_nb, const uint32_t* d_order, uint64_t n, uint8_t* d_dig)
{
    vit::DeviceBatch b;
    b.arena = static_cast<const uint8_t*>(d_arena);
    b.arena_bytes = arena_bytes;
    b.offsets = d_off; b.nblocks = d_nb; b.order = d_order;
    b.n = uint32_t(n);
    b.digests = d_dig;
    b.status = q.d_status;
    b.wave_trace = q.wave_trace;
    b.gate_percent = 0; b.gate_simds = 0;
    return b;
}

// The SHA-1 launch of a batch: software-scheduled when the batch is in binned
// order and large enough to fill the resident grid several times, else plain.
// The scheduled launch pays only where the batch's longest object is long against
// a SIMD's mean share of the work (vit::tail_dominates; batches of short objects
// lose a factor of three under it).  longest_blocks: that object's block count
// when the host has seen the lengths, 0 when they exist in device memory only —
// then both launches go out and their waves decide (DeviceBatch::gate_percent).
// staging_set: 0..2 for a chunk of that staging set, -1 for a caller's device-resident arena.
hipError_t launch_hash(vit_hasher* h, Queue& q, const vit::DeviceBatch& b, hipStream_t stream, int staging_set,
                       uint64_t longest_blocks)
{
    const uint64_t groups = (uint64_t(b.n) + 63) / 64;
    const vit::KernelVariant v = vit::KernelVariant(h->kernel);
    if (h->kernel != VIT_KERNEL_LDS || !b.order || !h->sched_waves || groups < h->sched_min_groups)
        return vit::launch_sha1(v, b, stream);
    const uint32_t pct = h->sched_tail_percent;
    if (pct && longest_blocks && !vit::tail_dominates(longest_blocks, b.arena_bytes, q.simds, pct))
        return vit::launch_sha1(v, b, stream);
    const uint32_t state = staging_set >= 0 ? uint32_t(staging_set) : 3u + q.sched_next++ % (kSchedStates - 3u);
    if (!pct || longest_blocks) return vit::launch_sha1_sched(b, &q.d_sched[state], h->sched_waves, h->sched_share_below, stream);
    vit::DeviceBatch gated = b;
    gated.gate_percent = pct;
    gated.gate_simds = q.simds;
    const hipError_t e = vit::launch_sha1_sched(gated, &q.d_sched[state], h->sched_waves, h->sched_share_below, stream);
    return e != hipSuccess ? e : vit::launch_sha1(v, gated, stream);
}

// Collect a finished chunk: wait for it, scatter its digests to the caller and
// note the status word it brought back.
int retire(Queue& q, Staging& s, uint8_t* digests)
{
    if (!s.in_flight) return VIT_OK;
    HIP_TRYQ(q, hipEventSynchronize(s.done));
    if (s.obj.empty()) std::memcpy(digests + s.first * 20, s.h_dig, s.count * 20);
    else for (size_t i = 0; i < s.obj.size(); ++i) std::memcpy(digests + s.obj[i] * 20, s.h_dig + i * 20, 20);
    q.status_seen |= *s.h_status;
    s.in_flight = false;
    if (s.traced) {
        float g[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) HIP_TRYQ(q, hipEventElapsedTime(&g[k], q.ev_a, s.mark[k]));
        char line[256];
        std::snprintf(line, sizeof line,
                      "set %d objs %8zu | host ms: begin %8.2f retired-prev %8.2f packed %8.2f queued %8.2f collected %8.2f"
                      " | gpu ms: copy %8.2f..%8.2f kernel ..%8.2f read-back ..%8.2f\n",
                      int(&s - q.st), s.obj.empty() ? size_t(s.count) : s.obj.size(), s.host_ms[0], s.host_ms[1],
                      s.host_ms[2], s.host_ms[3], double(ns_since(q.trace_t0)) * 1e-6, g[0], g[1], g[2], g[3]);
        q.trace_log += line;
    }
    return VIT_OK;
}

// Forget whatever an earlier call left in flight on this queue (a call that
// failed midway returns with chunks still queued): wait for the streams and
// drop the chunks WITHOUT delivering their digests, whose destination belonged
// to that earlier call.
void drain(Queue& q)
{
    bool any = false;
    for (const Staging& s : q.st) any = any || s.in_flight;
    if (!any) return;
    (void)hipSetDevice(q.device);
    for (hipStream_t st : q.stream) if (st) (void)hipStreamSynchronize(st);
    for (Staging& s : q.st) { s.in_flight = false; s.obj.clear(); s.count = 0; }
}

struct Job {
    vit_hasher* h;
    const char* type; uint32_t tl;
    const uint8_t* const* data; const uint64_t* len;
    const uint32_t* nb;                    // block count per object (all objects)
    uint8_t* digests;
};

// Queue copy, kernel, digest and status read-back of a chunk whose arena bytes
// [0, bytes) are at `src` in pinned (or at least host) memory.
int launch_chunk(Queue& q, vit_hasher* h, Staging& s, hipStream_t stream, const uint8_t* src, uint64_t bytes,
                 uint64_t meta_bytes, bool binned, uint64_t cnt, bool trace)
{
    if (trace) { s.host_ms[2] = double(ns_since(q.trace_t0)) * 1e-6; HIP_TRYQ(q, hipEventRecord(s.mark[0], stream)); }
    HIP_TRYQ(q, hipMemcpyAsync(s.d_arena, src, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRYQ(q, hipMemcpyAsync(s.d_meta, s.h_meta, meta_bytes + (binned ? meta_bytes / 3 : 0), hipMemcpyHostToDevice, stream));
    if (trace) HIP_TRYQ(q, hipEventRecord(s.mark[1], stream));
    // the host packed or checked these objects: it knows the longest (first in binned order)
    const uint64_t longest = cnt ? (binned ? s.h_nb[s.h_order[0]] : s.h_nb[0]) : 0;
    HIP_TRYQ(q, launch_hash(h, q, make_batch(q, s.d_arena, bytes, s.d_off, s.d_nb, binned ? s.d_order : nullptr, cnt,
                                             s.d_dig), stream, int(&s - q.st), std::max<uint64_t>(longest, 1)));
    if (trace) HIP_TRYQ(q, hipEventRecord(s.mark[2], stream));
    HIP_TRYQ(q, hipMemcpyAsync(s.h_dig, s.d_dig, cnt * 20, hipMemcpyDeviceToHost, stream));
    HIP_TRYQ(q, hipMemcpyAsync(s.h_status, q.d_status, 4, hipMemcpyDeviceToHost, stream));
    if (trace) { HIP_TRYQ(q, hipEventRecord(s.mark[3], stream)); s.host_ms[3] = double(ns_since(q.trace_t0)) * 1e-6; }
    HIP_TRYQ(q, hipEventRecord(s.done, stream));
    s.in_flight = true;
    return VIT_OK;
}

// Pack one chunk (objects idx[0..cnt)) into staging set s and queue its copy,
// kernel and digest read-back on `stream`.
int submit_chunk(Queue& q, const Job& job, Staging& s, hipStream_t stream, const uint64_t* idx, uint64_t cnt,
                 uint64_t* pack_ns)
{
    vit_hasher* h = job.h;
    const double begin_ms = double(ns_since(q.trace_t0)) * 1e-6;
    int rc = retire(q, s, job.digests);
    if (rc) return rc;
    if (h->fail_at_chunk && ++q.chunks_this_call == h->fail_at_chunk) {
        q.err = "failure injected at chunk " + std::to_string(q.chunks_this_call) + " (option debug_fail_at_chunk)";
        return VIT_ERR_HIP;
    }
    if (h->trace) {
        for (hipEvent_t& m : s.mark) if (!m) HIP_TRYQ(q, hipEventCreate(&m));
        s.host_ms[0] = begin_ms;
        s.host_ms[1] = double(ns_since(q.trace_t0)) * 1e-6;
    }
    const bool trace = s.traced = h->trace && s.mark[3];
    uint64_t blocks = 0;
    for (uint64_t i = 0; i < cnt; ++i) blocks += job.nb[idx[i]];
    const uint64_t bytes = blocks * 64;
    if ((rc = ensure_arena(q, s, bytes))) return rc;
    if ((rc = ensure_meta(q, s, cnt))) return rc;
    const uint64_t meta_bytes = carve_meta(s, cnt);

    const auto t0 = clk::now();
    s.obj.assign(idx, idx + cnt);
    s.first = s.count = 0;
    std::vector<uint64_t> prefix(cnt + 1, 0);
    for (uint64_t i = 0; i < cnt; ++i) {
        s.h_off[i] = prefix[i] * 64;
        s.h_nb[i] = job.nb[idx[i]];
        prefix[i + 1] = prefix[i] + s.h_nb[i];
    }
    auto pack_range = [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            const uint64_t o = idx[i];
            vit::pack_one(job.type, job.tl, job.data[o], job.len[o], s.h_arena + s.h_off[i], s.h_nb[i]);
        }
        vit::pack_fence();
    };
    // Pieces of about 2 MiB, at most four per thread of the budget: short enough
    // that the pool's threads even out, long enough that claiming one is free.
    const uint64_t pieces = std::min<uint64_t>({cnt, bytes / (2u << 20), uint64_t(h->pack_threads) * 4});
    if (pieces <= 1) pack_range(0, cnt);
    else {
        const std::vector<uint64_t> cut = vit::balanced_cuts(cnt, int(pieces), prefix.data());
        h->pool->run(vit::WorkPool::HIGH, size_t(pieces), [&](size_t p) { if (cut[p] < cut[p + 1]) pack_range(cut[p], cut[p + 1]); });
    }
    const bool binned = vit::bin_order(s.h_nb, cnt, s.h_order);
    *pack_ns += ns_since(t0);

    if ((rc = launch_chunk(q, h, s, stream, s.h_arena, bytes, meta_bytes, binned, cnt, trace))) return rc;
    h->objects_gpu += cnt; h->blocks_gpu += blocks; h->bytes_h2d += bytes + cnt * 16; ++h->launches;
    return VIT_OK;
}

// Objects of at least this many blocks (256 KiB) are "long": a lane hashes about
// 50 MB/s, so each of them holds its wave for milliseconds.  They are launched
// first, on a stream and staging set of their own, and the short objects'
// chunks stream past them on the other two.
constexpr uint32_t kLongBlocks = 4096;

int run_queue_body(Queue& q, const Job& job, const std::vector<uint64_t>& mine, uint64_t* pack_ns)
{
    vit_hasher* h = job.h;
    HIP_TRYQ(q, hipSetDevice(q.device));
    q.trace_t0 = clk::now();
    q.trace_log.clear();
    q.chunks_this_call = 0;
    q.grew_this_call = false;
    if (h->trace) HIP_TRYQ(q, hipEventRecord(q.ev_a, q.stream[0]));
    // One pass over the queue's objects: which are long, and what the rest add up to.
    // Without long objects (the common case) the rest IS the list as given: no copy.
    std::vector<uint64_t> longs, rest_copy;
    uint64_t rest_bytes = 0, rest_longest = 0;
    for (uint64_t i : mine) {
        const uint32_t b = job.nb[i];
        if (b >= kLongBlocks) { longs.push_back(i); continue; }
        rest_bytes += uint64_t(b) * 64;
        rest_longest = std::max<uint64_t>(rest_longest, b);
    }
    if (!longs.empty()) {
        rest_copy.reserve(mine.size() - longs.size());
        for (uint64_t i : mine) if (job.nb[i] < kLongBlocks) rest_copy.push_back(i);
        std::stable_sort(longs.begin(), longs.end(), [&](uint64_t a, uint64_t b) { return job.nb[a] > job.nb[b]; });
    }
    const std::vector<uint64_t>& rest = longs.empty() ? mine : rest_copy;

    // Chunk boundaries: as many objects from v[pos] on as fit the staging budget (at
    // least one); returns the end.  Cut as the chunks are submitted, not up front.
    auto cut_from = [&](const std::vector<uint64_t>& v, size_t pos, uint64_t budget) {
        uint64_t blocks = 0;
        size_t end = pos;
        while (end < v.size() && (end == pos || (blocks + job.nb[v[end]]) * 64 <= budget)) blocks += job.nb[v[end++]];
        return end;
    };
    // A chunk's kernel lasts at least as long as its longest object on one lane
    // (45 MB/s; 1.5 ms for 64 KiB), while its transfer runs about 1000 times
    // faster per byte, and measured transfers slow down while another chunk's
    // kernel is running.  Chunks therefore grow with the longest object they may
    // hold, to chunk_object_multiple (8192) times its size: 512 MiB for 64 KiB
    // objects gave twice the throughput of 64 MiB on a skewed 15 GB batch.
    // Bounds: [configured chunk size, 1 GiB], and at most 1/8 of the batch so
    // that packing, copying and hashing still overlap.
    uint64_t chunk = std::min<uint64_t>(rest_longest * 64 * h->chunk_object_multiple, rest_bytes / 8);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, h->chunk_bytes), 1ull << 30);
    const uint64_t long_chunk = std::max<uint64_t>(chunk, 256ull << 20);

    // Long chunks go first and then whenever their staging set has drained, so
    // that their lane-bound kernels run beside the short objects' chunks and
    // not after them.
    int rc;
    size_t next_long = 0;
    auto submit_long = [&]() -> int {
        const size_t end = cut_from(longs, next_long, long_chunk);
        const int r = submit_chunk(q, job, q.st[2], q.stream[2], longs.data() + next_long, end - next_long, pack_ns);
        next_long = end;
        return r;
    };
    if (!longs.empty() && (rc = submit_long())) return rc;
    int turn = 0;
    for (size_t pos = 0; pos < rest.size();) {
        const size_t end = cut_from(rest, pos, chunk);
        if ((rc = submit_chunk(q, job, q.st[turn], q.stream[turn], rest.data() + pos, end - pos, pack_ns))) return rc;
        pos = end;
        turn ^= 1;
        if (next_long < longs.size() && hipEventQuery(q.st[2].done) == hipSuccess && (rc = submit_long())) return rc;
    }
    while (next_long < longs.size())
        if ((rc = submit_long())) return rc;
    for (Staging& s : q.st)
        if ((rc = retire(q, s, job.digests))) return rc;
    return VIT_OK;
}

// Hash the objects `mine` (caller indices, in order) on queue q.  However it
// ends, nothing stays in flight on the queue.
int run_queue(Queue& q, const Job& job, const std::vector<uint64_t>& mine, uint64_t* pack_ns)
{
    int rc;
    try { rc = run_queue_body(q, job, mine, pack_ns); }
    catch (const std::bad_alloc&) { q.err = "out of host memory"; rc = VIT_ERR_NOMEM; }
    catch (const std::exception& e) { q.err = e.what(); rc = VIT_ERR_HIP; }
    if (rc != VIT_OK) drain(q);
    return rc;
}

// Split object indices over the queues by padded size: the few objects that
// are large against a fair share go greedily to the least loaded queue, the
// rest fill the queues in index order up to the fair share.
void shard(const std::vector<uint64_t>& idx, const uint32_t* nb, size_t nq, std::vector<std::vector<uint64_t>>& out)
{
    out.assign(nq, {});
    if (nq == 1) { out[0] = idx; return; }
    uint64_t total = 0;
    for (uint64_t i : idx) total += nb[i];
    const uint64_t fair = total / nq + 1, big = fair / 4 + 1;
    std::vector<uint64_t> load(nq, 0), large;
    for (uint64_t i : idx) if (nb[i] >= big) large.push_back(i);
    for (auto& v : out) v.reserve(idx.size() / nq + idx.size() / (8 * nq) + 16);
    std::stable_sort(large.begin(), large.end(), [&](uint64_t a, uint64_t b) { return nb[a] > nb[b]; });
    for (uint64_t i : large) {
        size_t best = 0;
        for (size_t k = 1; k < nq; ++k) if (load[k] < load[best]) best = k;
        out[best].push_back(i); load[best] += nb[i];
    }
    std::vector<size_t> n_large(nq);
    for (size_t q = 0; q < nq; ++q) n_large[q] = out[q].size();
    size_t k = 0;
    for (uint64_t i : idx) {
        if (nb[i] >= big) continue;
        while (k + 1 < nq && load[k] >= fair) ++k;
        out[k].push_back(i); load[k] += nb[i];
    }
    // Every queue's list in index order.  The small objects arrived in the order
    // of idx (ascending, as the callers pass it): sort the few large ones and
    // merge, which is linear; a million-object list costs a full sort 20 ms.
    for (size_t q = 0; q < nq; ++q) {
        auto& v = out[q];
        const auto mid = v.begin() + long(n_large[q]);
        if (std::is_sorted(mid, v.end())) {
            std::sort(v.begin(), mid);
            std::inplace_merge(v.begin(), mid, v.end());
        } else {
            std::sort(v.begin(), v.end());
        }
    }
}

// Read the queue's device status word, and the one in the device's static
// scratch, through the queue's first stream; clear what was set.
int read_status_word(vit_hasher* h, Queue& q, uint32_t* out)
{
    HIP_TRY(h, hipMemcpyAsync(&q.h_status[0], q.d_status, 4, hipMemcpyDeviceToHost, q.stream[0]));
    HIP_TRY(h, hipMemcpyAsync(&q.h_status[1], q.scratch, 4, hipMemcpyDeviceToHost, q.stream[0]));
    HIP_TRY(h, hipStreamSynchronize(q.stream[0]));
    *out = q.h_status[0] | q.h_status[1];
    if (*out) {
        if (q.h_status[0]) HIP_TRY(h, hipMemsetAsync(q.d_status, 0, 4, q.stream[0]));
        if (q.h_status[1]) HIP_TRY(h, hipMemsetAsync(q.scratch, 0, 4, q.stream[0]));
        HIP_TRY(h, hipStreamSynchronize(q.stream[0]));
    }
    return VIT_OK;
}

// The two host hashing rates: one core on one SHA-1 stream, and on kHostStreams
// streams interleaved.  About 5 ms, once per hasher; needed by the balance
// (calibrate) and by the grouping of host-lane objects under any policy.
void calibrate_host(vit_hasher* h)
{
    if (h->host_calibrated) return;
    constexpr uint64_t kHost = 4ull << 20;
    std::vector<uint8_t> src(kHost, 0x5A);
    if (!h->core_fixed) {
        double best = 1e30;
        uint8_t d[20];
        for (int r = 0; r < 2; ++r) {
            const auto t = clk::now();
            vit::host_object_id("blob", src.data(), kHost, d);
            best = std::min(best, double(ns_since(t)) * 1e-9);
        }
        h->core_rate = double(kHost) / best;
    }
    if (!h->core_multi_fixed) {
        // the same bytes as four objects side by side on this core
        const uint8_t* four[vit::kHostStreams];
        uint64_t lens[vit::kHostStreams];
        uint8_t d[20 * vit::kHostStreams];
        for (int k = 0; k < vit::kHostStreams; ++k) { four[k] = src.data() + size_t(k) * (kHost / vit::kHostStreams); lens[k] = kHost / vit::kHostStreams; }
        double best = 1e30;
        for (int r = 0; r < 2; ++r) {
            const auto t = clk::now();
            vit::host_object_ids("blob", four, lens, vit::kHostStreams, d);
            best = std::min(best, double(ns_since(t)) * 1e-9);
        }
        h->core_rate_multi = double(kHost) / best;
    }
    h->host_calibrated = true;
}

// Measure, once per hasher and on its first queue, the four rates the auto
// policy balances with (anything set through an option stays as set):
//   core_rate  one host thread's SHA-1 (vit::host_object_id over 4 MiB)
//   pack_rate  one host thread packing into a pinned staging arena: up to 8 MiB of
//              the batch's own longest object when that is 1 MiB or more (data
//              as cold in the caches as the batch will be), else a 4 MiB buffer
//              of the library's (cache-warm, optimistic); real batches of 256 MiB
//              and more then correct the rate (hash_batch)
//   copy_rate  one pinned host-to-device copy (32 MiB)
//   lane_rate  one GPU lane: a lone wave running 1024 compressions back to back,
//              times 0.8: with four other waves resident, the SIMD's oldest wave
//              (a long object's wave always is) gets about that share of issue
//              (1.15 us per block alone, 1.4 to 1.5 us in a full launch).
int calibrate(vit_hasher* h, const uint8_t* sample, uint64_t sample_len, const uint8_t* const* data = nullptr,
              const uint64_t* len_all = nullptr, uint64_t n_all = 0)
{
    if (h->calibrated) return VIT_OK;
    const auto t0 = clk::now();
    Queue& q = h->queues[0];
    HIP_TRY(h, hipSetDevice(q.device));
    constexpr uint64_t kHost = 4ull << 20, kCopy = 32ull << 20;
    std::vector<uint8_t> src(kHost, 0x5A);
    calibrate_host(h);
    if (!h->pack_fixed || !h->copy_fixed) {
        Staging& s = q.st[0];
        if (ensure_arena(q, s, kCopy) != VIT_OK) return fail(h, VIT_ERR_NOMEM, "calibration: %s", q.err.c_str());
        if (!h->pack_fixed) {
            const bool own = sample && sample_len >= (1u << 20);
            const uint8_t* from = own ? sample : src.data();
            const uint64_t len = own ? std::min<uint64_t>(sample_len, 8ull << 20) : kHost - 64;
            const uint64_t nb = (4 + 1 + vit::decimal_digits(len) + 1 + len + 9 + 63) / 64;
            double best = 1e30;
            for (int r = 0; r < (own ? 1 : 2); ++r) {     // the caller's data is cold exactly once
                const auto t = clk::now();
                vit::pack_one("blob", 4, from, len, s.h_arena, nb);
                vit::pack_fence();
                best = std::min(best, double(ns_since(t)) * 1e-9);
            }
            h->pack_rate = double(len) / best;
            // A large batch says more: pack 64 MiB of it (objects of at most 8 MiB) the way a chunk
            // is packed, on every thread of the budget at once, and take the rate per thread.  One
            // thread alone on one cold object measures the latency of a single stream of misses, not
            // what sixteen threads get out of the memory system together.  The 64 MiB before it are
            // packed first, untimed, into the same stretch of the arena: the first touch of freshly
            // pinned memory runs at a fraction of the rate (17 against 170 GB/s measured) and is a
            // cost of the allocation, not of packing.
            constexpr uint64_t kSample = 64ull << 20, kEach = 8ull << 20;
            uint64_t next = 0;
            auto pooled_pack = [&](double* seconds) -> uint64_t {          // returns the payload bytes packed
                std::vector<uint64_t> pick, prefix(1, 0);
                uint64_t payload = 0;
                for (; data && len_all && next < n_all && payload < kSample; ++next) {
                    if (len_all[next] > kEach || (len_all[next] && !data[next])) continue;
                    pick.push_back(next);
                    prefix.push_back(prefix.back() + (4 + 1 + vit::decimal_digits(len_all[next]) + 1 + len_all[next] + 9 + 63) / 64);
                    payload += len_all[next];
                }
                if (payload < kSample / 2 || ensure_arena(q, s, std::max<uint64_t>(prefix.back() * 64, kCopy)) != VIT_OK) return 0;
                const size_t pieces = std::min<size_t>(pick.size(), size_t(h->pack_threads) * 4);
                const std::vector<uint64_t> cut = vit::balanced_cuts(pick.size(), int(pieces), prefix.data());
                const auto t = clk::now();
                h->pool->run(vit::WorkPool::HIGH, pieces, [&](size_t p) {
                    for (uint64_t k = cut[p]; k < cut[p + 1]; ++k)
                        vit::pack_one("blob", 4, data[pick[k]], len_all[pick[k]], s.h_arena + prefix[k] * 64, prefix[k + 1] - prefix[k]);
                    vit::pack_fence();
                });
                *seconds = double(ns_since(t)) * 1e-9;
                return payload;
            };
            double sec = 0;
            if (h->pack_threads > 1 && pooled_pack(&sec)) {
                uint64_t payload = pooled_pack(&sec);
                if (!payload) { next = 0; payload = pooled_pack(&sec); }      // a batch of under 128 MiB: the same stretch again
                if (payload) h->pack_rate = double(payload) / sec / double(h->pack_threads);
            }
        }
        if (!h->copy_fixed) {
            float best = 1e30f;
            for (int r = 0; r < 2; ++r) {
                HIP_TRY(h, hipEventRecord(q.ev_a, q.stream[0]));
                HIP_TRY(h, hipMemcpyAsync(s.d_arena, s.h_arena, kCopy, hipMemcpyHostToDevice, q.stream[0]));
                HIP_TRY(h, hipEventRecord(q.ev_b, q.stream[0]));
                HIP_TRY(h, hipEventSynchronize(q.ev_b));
                float ms = 0;
                HIP_TRY(h, hipEventElapsedTime(&ms, q.ev_a, q.ev_b));
                best = std::min(best, ms);
            }
            h->copy_rate = double(kCopy) / (double(best) * 1e-3);
        }
    }
    if (!h->lane_fixed) {
        constexpr uint32_t kBlocks = 1024;
        uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(q.d_status) + kCalibOut);
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {                     // the first launch loads the code object
            HIP_TRY(h, hipEventRecord(q.ev_a, q.stream[0]));
            HIP_TRY(h, vit::launch_valu_peak(out, 64, kBlocks, q.stream[0]));
            HIP_TRY(h, hipEventRecord(q.ev_b, q.stream[0]));
            HIP_TRY(h, hipEventSynchronize(q.ev_b));
            float ms = 0;
            HIP_TRY(h, hipEventElapsedTime(&ms, q.ev_a, q.ev_b));
            if (r) best = std::min(best, ms);
        }
        h->lane_rate = 0.8 * double(kBlocks) * 64.0 / (double(best) * 1e-3);
    }
    h->calibrated = true;
    h->calibrate_ns = ns_since(t0);
    return VIT_OK;
}

}  // namespace

extern "C" {

int vit_hasher_create(const int* device_ids, int n_devices, vit_hasher** out)
{
    if (!out || n_devices < 0 || (n_devices > 0 && !device_ids)) return fail(nullptr, VIT_ERR_ARG, "bad arguments");
    *out = nullptr;
    int visible = 0;
    hipError_t e = hipGetDeviceCount(&visible);
    if (e != hipSuccess || visible <= 0)
        return fail(nullptr, VIT_ERR_NODEVICE, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    std::vector<int> ids;
    if (n_devices == 0) for (int d = 0; d < visible; ++d) ids.push_back(d);
    else ids.assign(device_ids, device_ids + n_devices);

    vit_hasher* h = new (std::nothrow) vit_hasher();
    if (!h) return fail(nullptr, VIT_ERR_NOMEM, "out of host memory");
    try {
        h->queues.resize(ids.size());
        // Thread budget: the machine's cores, at most 16 per device queue (what one
        // PCIe link's worth of packing can use); option "pack_threads" overrides.
        unsigned hw = std::thread::hardware_concurrency();
        unsigned budget = std::min<unsigned>(hw ? hw : 4, 16u * unsigned(ids.size()));
        h->pack_threads = int(std::max(1u, budget));
        h->pool.reset(new vit::WorkPool(pool_threads_for(h)));
    } catch (const std::exception&) {
        delete h;
        return fail(nullptr, VIT_ERR_NOMEM, "out of host memory");
    }
    for (size_t k = 0; k < ids.size(); ++k) {
        Queue& q = h->queues[k];
        q.device = ids[k];
        hipDeviceProp_t prop;
        hipError_t r = (ids[k] < 0 || ids[k] >= visible) ? hipErrorInvalidDevice : hipGetDeviceProperties(&prop, ids[k]);
        if (r == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fail(nullptr, VIT_ERR_NODEVICE, "device %d is %s; the kernels are built for gfx950 only", ids[k], prop.gcnArchName);
            vit_hasher_destroy(h);
            return VIT_ERR_NODEVICE;
        }
        if (r == hipSuccess) r = hipSetDevice(ids[k]);
        for (hipStream_t& st : q.stream)
            if (r == hipSuccess) r = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (r == hipSuccess) r = vit::device_simds(&q.simds);
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&q.d_status), kStatusBytes);
        if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void**>(&q.h_status), 64, hipHostMallocDefault);
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&q.d_sched), sizeof(vit::SchedState) * kSchedStates);
        if (r == hipSuccess) r = hipMemsetAsync(q.d_sched, 0, sizeof(vit::SchedState) * kSchedStates, q.stream[0]);
        // The kernels run on non-blocking streams, which the null stream does not
        // order with: clear the word on the stream that will use it, and wait.
        if (r == hipSuccess) r = hipMemsetAsync(q.d_status, 0, kStatusBytes, q.stream[0]);
        if (r == hipSuccess) r = hipStreamSynchronize(q.stream[0]);
        // The file-scope scratch array has one copy per GPU: ask for this device's
        // address now, with the device current, and clear its status word.
        if (r == hipSuccess) r = vit::static_scratch_address(&q.scratch);
        if (r == hipSuccess) r = hipMemsetAsync(q.scratch, 0, 256, q.stream[0]);
        if (r == hipSuccess) r = hipStreamSynchronize(q.stream[0]);
        if (r == hipSuccess) r = hipEventCreate(&q.ev_a);
        if (r == hipSuccess) r = hipEventCreate(&q.ev_b);
        for (Staging& s : q.st)
            if (r == hipSuccess) r = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (r != hipSuccess) {
            fail(nullptr, VIT_ERR_HIP, "setting up device %d failed: %s", ids[k], hipGetErrorString(r));
            vit_hasher_destroy(h);
            return VIT_ERR_HIP;
        }
    }
    *out = h;
    return VIT_OK;
}

void vit_hasher_destroy(vit_hasher* h)
{
    if (!h) return;
    for (Queue& q : h->queues) {
        (void)hipSetDevice(q.device);
        for (hipStream_t st : q.stream) if (st) (void)hipStreamSynchronize(st);
        for (Staging& s : q.st) free_staging(s);
        if (q.d_status) (void)hipFree(q.d_status);
        if (q.h_status) (void)hipHostFree(q.h_status);
        if (q.d_sched) (void)hipFree(q.d_sched);
        if (q.ev_a) (void)hipEventDestroy(q.ev_a);
        if (q.ev_b) (void)hipEventDestroy(q.ev_b);
        if (q.sort_scratch) (void)hipFree(q.sort_scratch);
        if (q.z_workspace) (void)hipFree(q.z_workspace);
        if (q.z_tables) (void)hipFree(q.z_tables);
        if (q.z_done) (void)hipEventDestroy(q.z_done);
        if (q.z_out) (void)hipFree(q.z_out);
        if (q.z_meta_d) (void)hipFree(q.z_meta_d);
        if (q.z_meta_h) (void)hipHostFree(q.z_meta_h);
        for (hipEvent_t e : q.timing) (void)hipEventDestroy(e);
        for (hipStream_t st : q.stream) if (st) (void)hipStreamDestroy(st);
    }
    delete h;
}

const char* vit_last_error(const vit_hasher* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int vit_hasher_device_count(const vit_hasher* h) { return h ? int(h->queues.size()) : 0; }

int vit_hasher_set_option(vit_hasher* h, const char* name, uint64_t value)
{
    if (!h || !name) return VIT_ERR_ARG;
    VIT_NOEXCEPT_BEGIN
    const std::string k(name);
    auto rate = [&](double& r, bool& fixed) { r = double(value); fixed = value != 0; if (!value) h->calibrated = h->host_calibrated = false; };
    if (k == "kernel") {
        if (value > 5) return fail(h, VIT_ERR_ARG, "unknown kernel %llu", (unsigned long long)value);
        h->kernel = int(value);
    } else if (k == "host_threshold_bytes") { h->host_threshold = value; h->host_policy = value ? 2 : 0; }
    else if (k == "host_min_bytes") h->host_min_bytes = value;
    else if (k == "host_auto") { h->host_policy = value ? 1 : (h->host_threshold ? 2 : 0); }
    else if (k == "chunk_bytes") h->chunk_bytes = value < (1u << 20) ? (1u << 20) : value;
    else if (k == "chunk_object_multiple") h->chunk_object_multiple = value;
    else if (k == "trace") h->trace = value != 0;
    else if (k == "wave_trace_ptr") { for (Queue& q : h->queues) q.wave_trace = reinterpret_cast<uint64_t*>(uintptr_t(value)); }
    else if (k == "pack_threads") {
        h->pack_threads = value < 1 ? 1 : int(value > 256 ? 256 : value);
        h->pool->resize(pool_threads_for(h));
    }
    else if (k == "lane_rate") rate(h->lane_rate, h->lane_fixed);
    else if (k == "core_rate") rate(h->core_rate, h->core_fixed);
    else if (k == "core_rate_multi") rate(h->core_rate_multi, h->core_multi_fixed);
    else if (k == "copy_rate") rate(h->copy_rate, h->copy_fixed);
    else if (k == "pack_rate") rate(h->pack_rate, h->pack_fixed);
    else if (k == "host_bias_percent") h->host_bias = value ? double(value) / 100.0 : 1.0;
    else if (k == "debug_fail_at_chunk") h->fail_at_chunk = value;
    else if (k == "debug_fail_deflate") h->fail_deflate = value != 0;
    else if (k == "deflate_max_bytes") h->deflate_max_bytes = std::min<uint64_t>(value, vit::deflate_max_input());
    else if (k == "deflate_lanes") h->deflate_lanes = uint32_t(std::min<uint64_t>(std::max<uint64_t>(value, 64), 1u << 20) / 64 * 64);
    else if (k == "sched_waves") h->sched_waves = uint32_t(value > 8 ? 8 : value);
    else if (k == "sched_share_below") h->sched_share_below = uint32_t(value > 0xFFFFFFFEull ? 0xFFFFFFFEu : value);
    else if (k == "sched_min_groups") h->sched_min_groups = uint32_t(value > 0xFFFFFFFFull ? 0xFFFFFFFFu : value);
    else if (k == "sched_tail_percent") h->sched_tail_percent = uint32_t(value > 1000 ? 1000 : value);
    else return fail(h, VIT_ERR_ARG, "unknown option '%s'", name);
    return VIT_OK;
    VIT_NOEXCEPT_END(h)
}

int vit_hasher_get_stat(const vit_hasher* h, const char* name, uint64_t* value)
{
    if (!h || !name || !value) return VIT_ERR_ARG;
    const std::string k(name);
    if (k == "objects_gpu") *value = h->objects_gpu;
    else if (k == "objects_host") *value = h->objects_host;
    else if (k == "bytes_host") *value = h->bytes_host;
    else if (k == "last_host_threshold") *value = h->last_host_threshold;
    else if (k == "last_host_objects") *value = h->last_host_objects;
    else if (k == "blocks_gpu") *value = h->blocks_gpu;
    else if (k == "bytes_h2d") *value = h->bytes_h2d;
    else if (k == "launches") *value = h->launches;
    else if (k == "last_kernel_ns") *value = h->last_kernel_ns;
    else if (k == "last_pack_ns") *value = h->last_pack_ns;
    else if (k == "last_total_ns") *value = h->last_total_ns;
    else if (k == "last_gpu_done_ns") *value = h->last_gpu_done_ns;
    else if (k == "last_plan_ns") *value = h->last_plan_ns;
    else if (k == "host_bias_percent") *value = uint64_t(h->host_bias * 100.0 + 0.5);
    else if (k == "last_host_done_ns") *value = h->last_host_done_ns;
    else if (k == "host_sha_accelerated") *value = vit::HostSha1::accelerated();
    else if (k == "lane_rate") *value = uint64_t(h->lane_rate);
    else if (k == "core_rate") *value = uint64_t(h->core_rate);
    else if (k == "core_rate_multi") *value = uint64_t(h->core_rate_multi);
    else if (k == "copy_rate") *value = uint64_t(h->copy_rate);
    else if (k == "pack_rate") *value = uint64_t(h->pack_rate);
    else if (k == "calibrate_ns") *value = h->calibrate_ns;
    else if (k == "pool_threads") *value = uint64_t(h->pool->threads());
    else if (k == "pool_peak_busy") *value = uint64_t(h->pool->peak_busy());
    else return fail(h, VIT_ERR_ARG, "unknown stat '%s'", name);
    return VIT_OK;
}

// ids of n objects of one type, or (type == nullptr, tl == 0) plain SHA-1 of n buffers.
static int hash_batch(vit_hasher* h, const char* type, uint32_t tl, const uint8_t* const* data, const uint64_t* len,
                      uint64_t n, uint8_t* digests);

int vit_hash_objects(vit_hasher* h, const char* type, const uint8_t* const* data, const uint64_t* len,
                     uint64_t n, uint8_t* digests)
{
    if (!h) return VIT_ERR_ARG;
    const uint32_t tl = vit::type_len_checked(type);
    if (!tl) return fail(h, VIT_ERR_ARG, "bad object type");
    return hash_batch(h, type, tl, data, len, n, digests);
}

int vit_sha1_buffers(vit_hasher* h, const uint8_t* const* data, const uint64_t* len, uint64_t n, uint8_t* digests)
{
    if (!h) return VIT_ERR_ARG;
    return hash_batch(h, nullptr, 0, data, len, n, digests);
}

static int hash_batch(vit_hasher* h, const char* type, uint32_t tl, const uint8_t* const* data, const uint64_t* len,
                      uint64_t n, uint8_t* digests)
{
    if (n == 0) return VIT_OK;
    if (!data || !len || !digests) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    const auto t0 = clk::now();
    VIT_NOEXCEPT_BEGIN
    for (Queue& q : h->queues) { drain(q); q.status_seen = 0; }
    h->last_host_done_ns = 0;
    h->last_gpu_done_ns = 0;

    // Which objects leave the GPU for host threads: a fixed threshold, or (auto)
    // the balance vit::plan_host_split strikes with the rates measured on this
    // machine (calibrate), see packer.h.
    std::vector<uint32_t> nb(n);
    std::vector<uint64_t> gpu_idx, host_idx;
    uint64_t threshold = 0;
    if (h->host_policy == 2) threshold = h->host_threshold;
    else if (h->host_policy == 1) {
        bool any = false;
        for (uint64_t i = 0; i < n && !any; ++i) any = len[i] >= h->host_min_bytes;
        if (any) {
            uint64_t longest = 0;
            if (!h->calibrated) for (uint64_t i = 1; i < n; ++i) if (len[i] > len[longest]) longest = i;   // calibration's sample
            if (int rc = calibrate(h, data[longest], len[longest], data, len, n)) return rc;
            const vit::ParallelFor on_pool = [h](size_t parts, const std::function<void(size_t)>& fn) {
                h->pool->run(vit::WorkPool::HIGH, parts, fn);
            };
            vit::plan_host_split(len, n, vit::SplitRates{h->lane_rate, h->core_rate, h->copy_rate, h->pack_rate, h->core_rate_multi, h->host_min_bytes, h->host_bias},
                                 int(h->queues.size()), h->pack_threads, host_idx, &on_pool);
            if (!host_idx.empty()) threshold = len[host_idx.back()] - 1;
        }
    }
    h->last_host_threshold = threshold;

    uint64_t host_bytes = 0;
    if (h->host_policy == 2 && threshold)
        for (uint64_t i = 0; i < n; ++i) if (len[i] > threshold) host_idx.push_back(i);
    for (uint64_t i : host_idx) { nb[i] = 0xFFFFFFFFu; host_bytes += len[i]; }      // marker: not for a lane
    // Block counts and argument checks, on the pool for large batches (a million
    // objects cost 3 ms on one thread, as much as two staging chunks take to copy).
    {
        std::atomic<uint64_t> bad_null{~0ull}, bad_long{~0ull};
        const bool marked = !host_idx.empty();
        auto count = [&](uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                if (len[i] && !data[i]) { uint64_t cur = bad_null.load(); while (i < cur && !bad_null.compare_exchange_weak(cur, i)) {} continue; }
                if (marked && nb[i] == 0xFFFFFFFFu) continue;                 // a host-lane object: keeps its marker
                const uint64_t hl = tl ? tl + 1 + vit::decimal_digits(len[i]) + 1 : 0;
                const uint64_t blocks = (hl + len[i] + 9 + 63) / 64;
                if (blocks >= 0xFFFFFFFFull) { uint64_t cur = bad_long.load(); while (i < cur && !bad_long.compare_exchange_weak(cur, i)) {} continue; }
                nb[i] = uint32_t(blocks);
            }
        };
        const uint64_t slices = n >= 65536 ? std::min<uint64_t>(uint64_t(h->pack_threads) * 2, n / 16384) : 1;
        if (slices <= 1) count(0, n);
        else h->pool->run(vit::WorkPool::HIGH, size_t(slices), [&](size_t s) { count(n * s / slices, n * (s + 1) / slices); });
        if (bad_null != ~0ull)
            return fail(h, VIT_ERR_ARG, "object %llu: null data with non-zero length", (unsigned long long)bad_null.load());
        if (bad_long != ~0ull)
            return fail(h, VIT_ERR_RANGE, "object %llu too large for one lane", (unsigned long long)bad_long.load());
    }
    gpu_idx.resize(n - host_idx.size());
    if (host_idx.empty()) for (uint64_t i = 0; i < n; ++i) gpu_idx[i] = i;
    else {
        uint64_t k = 0;
        for (uint64_t i = 0; i < n; ++i) { if (nb[i] == 0xFFFFFFFFu) nb[i] = 0; else gpu_idx[k++] = i; }
    }
    h->last_host_objects = host_idx.size();

    // Oversized objects: one SHA-1 stream each on the pool's threads, at low
    // priority, beside the GPU batch; longest first, so the tail is short.
    std::vector<uint32_t> host_group;                     // group g = host_idx[host_group[g] .. host_group[g+1]): streams run side by side
    std::vector<uint32_t> host_task;                      // pool task t = groups [host_task[t], host_task[t+1])
    vit::WorkPool::Ticket host_work;
    // From the submit below on, the pool's workers hold references into this
    // frame: every path out waits for them first.
    struct WaitHost {
        vit::WorkPool& pool; vit::WorkPool::Ticket& t;
        ~WaitHost() { if (t) pool.wait(t); }
    } wait_host{*h->pool, host_work};
    // A core gets through more bytes with several objects' streams interleaved
    // (sha1_host.h), where the machine's SHA instructions are pipelined: when the
    // objects outnumber the threads, the objects after the longest few go in
    // groups of up to kHostStreams neighbours in length order.  The longest stay
    // alone: a group lasts as long as its objects' lengths added up, and an
    // object of more than 1/8 of a thread's share would make its group the tail.
    if (!host_idx.empty()) {
        if (h->host_policy != 1)                          // the balance hands its list over longest first
            std::stable_sort(host_idx.begin(), host_idx.end(), [&](uint64_t a, uint64_t b) { return len[a] > len[b]; });
        const size_t nh = host_idx.size(), nt = size_t(h->pack_threads);
        if (nh >= 2 * nt) calibrate_host(h);
        size_t streams = 1;
        if (h->core_rate_multi > 1.15 * h->core_rate && h->core_rate > 0) streams = std::min<size_t>(std::max<size_t>(nh / nt, 1), vit::kHostStreams);
        const uint64_t alone_above = host_bytes / (8 * nt);
        host_group.push_back(0);
        for (size_t k = 0; k < nh;) {
            const size_t take = (streams == 1 || len[host_idx[k]] > alone_above) ? 1 : std::min(streams, nh - k);
            k += take;
            host_group.push_back(uint32_t(k));
        }
        // One pool task = consecutive groups of at least kHostTaskBytes in all: claiming a
        // task takes the pool's lock, which sixteen threads claiming 16 KiB objects one
        // group at a time would spend their time waiting for.
        constexpr uint64_t kHostTaskBytes = 1u << 20;
        host_task.push_back(0);
        uint64_t acc = 0;
        for (size_t g = 0; g + 1 < host_group.size(); ++g) {
            for (uint32_t k = host_group[g]; k < host_group[g + 1]; ++k) acc += len[host_idx[k]];
            if (acc >= kHostTaskBytes || g + 2 == host_group.size()) { host_task.push_back(uint32_t(g + 1)); acc = 0; }
        }
        host_work = h->pool->submit(vit::WorkPool::LOW, host_task.size() - 1, [&](size_t t) {
            struct Stamp {                                // the latest finish over all tasks
                vit_hasher* h; clk::time_point t0;
                ~Stamp() { const uint64_t now = ns_since(t0); uint64_t cur = h->last_host_done_ns.load();
                           while (now > cur && !h->last_host_done_ns.compare_exchange_weak(cur, now)) {} }
            } stamp{h, t0};
            for (uint32_t g = host_task[t]; g < host_task[t + 1]; ++g) {
                const size_t b = host_group[g], cnt = host_group[g + 1] - b;
                if (cnt == 1) { const uint64_t i = host_idx[b]; vit::host_object_id(type, data[i], len[i], digests + i * 20); continue; }
                const uint8_t* ptr[vit::kHostStreams];
                uint64_t ln[vit::kHostStreams];
                uint8_t out[20 * vit::kHostStreams];
                for (size_t j = 0; j < cnt; ++j) { ptr[j] = data[host_idx[b + j]]; ln[j] = len[host_idx[b + j]]; }
                vit::host_object_ids(type, ptr, ln, cnt, out);
                for (size_t j = 0; j < cnt; ++j) std::memcpy(digests + host_idx[b + j] * 20, out + 20 * j, 20);
            }
        });
        h->objects_host += host_idx.size();
        h->bytes_host += host_bytes;
    }

    Job job{h, type, tl, data, len, nb.data(), digests};
    std::vector<std::vector<uint64_t>> parts;
    shard(gpu_idx, nb.data(), h->queues.size(), parts);
    std::vector<int> rcs(h->queues.size(), VIT_OK);
    std::vector<uint64_t> pack_ns(h->queues.size(), 0);
    h->last_plan_ns = ns_since(t0);
    if (h->queues.size() == 1) {
        rcs[0] = run_queue(h->queues[0], job, parts[0], &pack_ns[0]);
    } else {
        ThreadGroup workers;
        for (size_t k = 0; k < h->queues.size(); ++k)
            workers.t.emplace_back([&, k] { rcs[k] = run_queue(h->queues[k], job, parts[k], &pack_ns[k]); });
        workers.join();
    }
    h->last_gpu_done_ns = ns_since(t0);
    h->pool->wait(host_work);
    host_work.reset();
    h->last_pack_ns = 0;
    for (uint64_t p : pack_ns) h->last_pack_ns = std::max(h->last_pack_ns, p);
    // The calibration packs one cache-warm buffer; a real batch streams from
    // DRAM on every thread at once.  Let what this call measured correct the
    // rate the next call balances with: bytes packed over the packing time of
    // the queue that packed longest, per thread of the budget (batches of 256 MiB and more only).
    // (Not from a call that packed into freshly pinned memory: its first touch runs at a
    // tenth of the rate and says nothing about the calls to come.)
    bool fresh_arena = false;
    for (const Queue& q : h->queues) fresh_arena = fresh_arena || q.grew_this_call;
    if (!h->pack_fixed && h->calibrated && h->last_pack_ns && !fresh_arena) {
        uint64_t packed = 0;
        for (uint64_t i : gpu_idx) packed += len[i];
        if (packed >= (256ull << 20)) {
            const double per_thread = double(packed) / (double(h->last_pack_ns) * 1e-9) / double(h->pack_threads);
            h->pack_rate = 0.5 * h->pack_rate + 0.5 * per_thread;
        }
    }
    // The rates are measured one at a time on an idle machine; a batch runs them
    // all at once against one memory system, and the more GPUs a hasher drives the
    // less of the sum the host's DRAM delivers.  What the model cannot know it
    // learns: the host side's estimate is scaled by how much later (or earlier)
    // than the device queues the host threads finished, damped, and only from
    // calls in which both sides had real work spread over many objects.
    if (h->host_policy == 1 && !fresh_arena && host_idx.size() >= 4 * size_t(h->pack_threads) &&
        gpu_idx.size() >= 1024 && h->last_gpu_done_ns > 1000000 && h->last_host_done_ns > 1000000) {
        const double ratio = std::min(2.0, std::max(0.5, double(h->last_host_done_ns.load()) / double(h->last_gpu_done_ns)));
        h->host_bias = std::min(5.0, std::max(0.2, h->host_bias * std::pow(ratio, 0.7)));
    }
    h->last_total_ns = ns_since(t0);
    if (h->trace)
        for (size_t k = 0; k < h->queues.size(); ++k)
            std::fprintf(stderr, "[vit trace] queue %zu, %.2f ms in all, %zu objects on host threads\n%s", k,
                         double(h->last_total_ns) * 1e-6, host_idx.size(), h->queues[k].trace_log.c_str());
    for (size_t k = 0; k < rcs.size(); ++k)
        if (rcs[k] != VIT_OK) return fail(h, rcs[k], "device queue %zu: %s", k, h->queues[k].err.c_str());
    // Every chunk brought the status word back behind its digests: no extra
    // synchronisation here unless a kernel did flag something.
    for (size_t k = 0; k < h->queues.size(); ++k) {
        Queue& q = h->queues[k];
        if (!q.status_seen) continue;
        const uint32_t seen = q.status_seen;
        q.status_seen = 0;
        uint32_t word = 0;
        HIP_TRY(h, hipSetDevice(q.device));
        if (int rc = read_status_word(h, q, &word)) return rc;      // clears the device word
        return fail(h, VIT_ERR_DEVICE_STATUS, "device queue %zu flagged status %u", k, seen | word);
    }
    return VIT_OK;
    VIT_NOEXCEPT_END(h)
}

int vit_hash_blobs(vit_hasher* h, const uint8_t* const* data, const uint64_t* len, uint64_t n, uint8_t* digests)
{
    return vit_hash_objects(h, "blob", data, len, n, digests);
}

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

int vit_shard_plan(const uint32_t* nblocks, uint64_t n, int n_queues, uint32_t* queue_of)
{
    if (n_queues < 1 || (n && (!nblocks || !queue_of))) return VIT_ERR_ARG;
    VIT_NOEXCEPT_BEGIN
    std::vector<uint64_t> idx(n);
    for (uint64_t i = 0; i < n; ++i) idx[i] = i;
    std::vector<std::vector<uint64_t>> parts;
    shard(idx, nblocks, size_t(n_queues), parts);
    for (size_t q = 0; q < parts.size(); ++q)
        for (uint64_t i : parts[q]) queue_of[i] = uint32_t(q);
    return VIT_OK;
    VIT_NOEXCEPT_END(nullptr)
}

int vit_launch_shape(int kernel, uint64_t n, uint32_t* blocks, uint32_t* threads)
{
    if (!blocks || !threads || kernel < 0 || kernel > 5 || n > 0xFFFFFFFFull) return VIT_ERR_ARG;
    vit::launch_shape(vit::KernelVariant(kernel), n, blocks, threads);
    return VIT_OK;
}

int vit_arena_alloc(vit_hasher* h, uint64_t bytes, void** out)
{
    if (!h || !out) return VIT_ERR_ARG;
    *out = nullptr;
    HIP_TRY(h, hipSetDevice(h->queues[0].device));
    HIP_TRY(h, hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocPortable));
    return VIT_OK;
}

int vit_arena_free(vit_hasher* h, void* arena)
{
    if (!h) return VIT_ERR_ARG;
    if (arena) HIP_TRY(h, hipHostFree(arena));
    return VIT_OK;
}

// The caller's arena is the host side of the staging: pieces of it go to the
// queue's two device arenas in turn, so the copy of piece k+1 runs beside the
// kernel of piece k.  Pieces are runs of consecutive objects whose slots ascend
// through the arena (what vit_arena_pack writes); an arena whose offsets are
// not in ascending order goes over in one piece.
int vit_hash_arena(vit_hasher* h, int dev, const void* arena, uint64_t arena_bytes, const uint64_t* offsets,
                   const uint32_t* nblocks, uint64_t n, uint8_t* digests)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!arena || !offsets || !nblocks || !digests) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    VIT_NOEXCEPT_BEGIN
    bool ascending = true;
    uint64_t blocks = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t sz = uint64_t(nblocks[i]) * 64;
        if (offsets[i] & 15) return fail(h, VIT_ERR_RANGE, "object %llu: offset not 16-byte aligned", (unsigned long long)i);
        if (offsets[i] > arena_bytes || sz > arena_bytes - offsets[i])
            return fail(h, VIT_ERR_RANGE, "object %llu: %llu blocks at offset %llu exceed the arena (%llu bytes)",
                        (unsigned long long)i, (unsigned long long)nblocks[i], (unsigned long long)offsets[i],
                        (unsigned long long)arena_bytes);
        if (i && offsets[i] < offsets[i - 1] + uint64_t(nblocks[i - 1]) * 64) ascending = false;
        blocks += nblocks[i];
    }
    drain(*q);
    q->status_seen = 0;
    HIP_TRY(h, hipSetDevice(q->device));
    const uint8_t* base = static_cast<const uint8_t*>(arena);

    // [first, end) object ranges, each spanning at most `budget` bytes of the arena (one object at least)
    std::vector<std::pair<uint64_t, uint64_t>> pieces;
    if (!ascending) pieces.emplace_back(0, n);
    else {
        const uint64_t budget = std::max<uint64_t>(h->chunk_bytes, 1ull << 20);
        for (uint64_t pos = 0; pos < n;) {
            const uint64_t lo = offsets[pos] & ~uint64_t(63);
            uint64_t end = pos + 1;
            while (end < n && offsets[end] + uint64_t(nblocks[end]) * 64 - lo <= budget) ++end;
            pieces.emplace_back(pos, end);
            pos = end;
        }
    }
    int rc = VIT_OK;
    int turn = 0;
    for (const auto& pc : pieces) {
        Staging& s = q->st[turn];
        hipStream_t stream = q->stream[turn];
        turn ^= 1;
        if ((rc = retire(*q, s, digests))) break;
        const uint64_t cnt = pc.second - pc.first;
        uint64_t lo = ~0ull, hi = 0;
        for (uint64_t i = pc.first; i < pc.second; ++i) {
            lo = std::min(lo, offsets[i]);
            hi = std::max(hi, offsets[i] + uint64_t(nblocks[i]) * 64);
        }
        lo &= ~uint64_t(63);
        const uint64_t span = hi > lo ? hi - lo : 0;
        if ((rc = ensure_device_arena(*q, s, std::max<uint64_t>(span, 64)))) break;
        if ((rc = ensure_meta(*q, s, cnt))) break;
        const uint64_t meta_bytes = carve_meta(s, cnt);
        for (uint64_t i = 0; i < cnt; ++i) { s.h_off[i] = offsets[pc.first + i] - lo; s.h_nb[i] = nblocks[pc.first + i]; }
        const bool binned = vit::bin_order(s.h_nb, cnt, s.h_order);
        s.obj.clear(); s.first = pc.first; s.count = cnt; s.traced = false;
        if ((rc = launch_chunk(*q, h, s, stream, base + lo, span, meta_bytes, binned, cnt, false))) break;
        h->bytes_h2d += span + cnt * 16; ++h->launches;
    }
    for (Staging& s : q->st)
        if (rc == VIT_OK) rc = retire(*q, s, digests);
    if (rc != VIT_OK) { drain(*q); return fail(h, rc, "device queue %d: %s", dev, q->err.c_str()); }
    h->objects_gpu += n; h->blocks_gpu += blocks;
    if (q->status_seen) {
        const uint32_t seen = q->status_seen;
        q->status_seen = 0;
        uint32_t word = 0;
        if (int rc2 = read_status_word(h, *q, &word)) return rc2;
        return fail(h, VIT_ERR_DEVICE_STATUS, "device queue %d flagged status %u", dev, seen | word);
    }
    return VIT_OK;
    VIT_NOEXCEPT_END(h)
}

int vit_hash_arena_device(vit_hasher* h, int dev, const void* d_arena, uint64_t arena_bytes, const uint64_t* d_offsets,
                          const uint32_t* d_nblocks, const uint32_t* d_order, uint64_t n, uint8_t* d_digests, void* stream)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!d_arena || !d_offsets || !d_nblocks || !d_digests) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    if (reinterpret_cast<uintptr_t>(d_arena) & 15) return fail(h, VIT_ERR_RANGE, "arena not 16-byte aligned");
    HIP_TRY(h, hipSetDevice(q->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : q->stream[0];
    HIP_TRY(h, launch_hash(h, *q, make_batch(*q, d_arena, arena_bytes, d_offsets, d_nblocks, d_order, n, d_digests), s, -1, 0));
    ++h->launches; h->objects_gpu += n;
    return VIT_OK;
}

// Allocates nothing: the order array, the binning counters and the status word
// live in the kernels' file-scope device array, whose address on this queue's
// device was taken at vit_hasher_create.
int vit_hash_arena_device_static(vit_hasher* h, int dev, const void* d_arena, uint64_t arena_bytes,
                                 const uint64_t* d_offsets, const uint32_t* d_nblocks, uint64_t n, uint8_t* d_digests,
                                 void* stream)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!d_arena || !d_offsets || !d_nblocks || !d_digests) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > vit::STATIC_MAX_OBJECTS)
        return fail(h, VIT_ERR_RANGE, "%llu objects: the entry point that allocates nothing takes at most %u per call",
                    (unsigned long long)n, vit::STATIC_MAX_OBJECTS);
    if (reinterpret_cast<uintptr_t>(d_arena) & 15) return fail(h, VIT_ERR_RANGE, "arena not 16-byte aligned");
    HIP_TRY(h, hipSetDevice(q->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : q->stream[0];
    const bool sched = h->kernel == VIT_KERNEL_LDS && h->sched_waves && (n + 63) / 64 >= h->sched_min_groups;
    HIP_TRY(h, vit::launch_sha1_static(vit::KernelVariant(h->kernel),
                                       make_batch(*q, d_arena, arena_bytes, d_offsets, d_nblocks, nullptr, n, d_digests),
                                       q->scratch, sched ? h->sched_waves : 0, h->sched_share_below,
                                       h->sched_tail_percent, s));
    ++h->launches; h->objects_gpu += n;
    return VIT_OK;
}

// Device memory of the deflate kernel, made on first use: the constant tables and one
// workspace slot per lane in flight.
static int ensure_deflate(vit_hasher* h, Queue& q)
{
    if (!q.z_tables) {
        std::vector<uint8_t> host(vit::deflate_tables_bytes());
        vit::deflate_make_tables(host.data());
        HIP_TRY(h, hipMalloc(&q.z_tables, host.size()));
        HIP_TRY(h, hipMemcpy(q.z_tables, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    if (!q.z_done) HIP_TRY(h, hipEventCreateWithFlags(&q.z_done, hipEventDisableTiming));
    if (q.z_lanes != h->deflate_lanes) {
        HIP_TRY(h, hipDeviceSynchronize());               // a launch on a caller's stream may still be using the old workspace
        if (q.z_workspace) { HIP_TRY(h, hipFree(q.z_workspace)); q.z_workspace = nullptr; q.z_lanes = 0; }
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&q.z_workspace), vit::deflate_workspace_bytes(h->deflate_lanes)));
        q.z_lanes = h->deflate_lanes;
    }
    return VIT_OK;
}

int vit_deflate_arena_device(vit_hasher* h, int dev, const void* d_arena, uint64_t arena_bytes, const uint64_t* d_offsets,
                             const uint32_t* d_nblocks, const uint32_t* d_order, uint64_t n, uint8_t* d_out, const uint64_t* d_out_offsets,
                             uint32_t* d_out_lens, void* stream)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!d_arena || !d_offsets || !d_nblocks || !d_out || !d_out_offsets || !d_out_lens) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    if (reinterpret_cast<uintptr_t>(d_arena) & 15) return fail(h, VIT_ERR_RANGE, "arena not 16-byte aligned");
    HIP_TRY(h, hipSetDevice(q->device));
    if (int rc = ensure_deflate(h, *q)) return rc;
    vit::DeflateBatch b;
    b.arena = static_cast<const uint8_t*>(d_arena); b.arena_bytes = arena_bytes;
    b.offsets = d_offsets; b.nblocks = d_nblocks; b.n = uint32_t(n); b.order = d_order;
    b.out = d_out; b.out_off = d_out_offsets; b.out_len = d_out_lens;
    b.workspace = q->z_workspace; b.tables_raw = q->z_tables; b.status = q->d_status;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : q->stream[0];
    HIP_TRY(h, hipStreamWaitEvent(s, q->z_done, 0));      // launches share the queue's workspace: one after the other
    HIP_TRY(h, vit::launch_deflate(b, q->z_lanes, s));
    HIP_TRY(h, hipEventRecord(q->z_done, s));
    return VIT_OK;
}

// Objects from host memory: those of at most "deflate_max_bytes" (framed) are packed into the
// queue's staging arena chunk by chunk, copied over, deflated by the kernel in order of length
// and their streams copied back into `out`; the others get length 0 and are the caller's.  One
// lane works through about 0.1 MB/s of input (a dependent chain of scattered table accesses per
// byte), so the device is for the many small objects of a batch and host cores for the long ones.
int vit_deflate_objects(vit_hasher* h, const char* type, const uint8_t* const* data, const uint64_t* len, uint64_t n,
                        uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint32_t* out_len)
{
    if (!h) return VIT_ERR_ARG;
    const uint32_t tl = vit::type_len_checked(type);
    if (!tl) return fail(h, VIT_ERR_ARG, "bad object type");
    if (n == 0) return VIT_OK;
    if (!data || !len || !out || !out_off || !out_len) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    VIT_NOEXCEPT_BEGIN
    Queue& q = h->queues[0];
    drain(q);
    if (h->fail_deflate) return fail(h, VIT_ERR_HIP, "failure injected (option debug_fail_deflate)");
    HIP_TRY(h, hipSetDevice(q.device));
    if (int rc = ensure_deflate(h, q)) return rc;

    // which objects, their block counts, and where their streams go
    std::vector<uint64_t> mine;
    std::vector<uint32_t> nb(n, 0);
    uint64_t total_out = 0;
    for (uint64_t i = 0; i < n; ++i) {
        out_len[i] = 0; out_off[i] = 0;
        if (len[i] && !data[i]) return fail(h, VIT_ERR_ARG, "object %llu: null data with non-zero length", (unsigned long long)i);
        const uint64_t framed = tl + 1 + vit::decimal_digits(len[i]) + 1 + len[i];
        if (framed > h->deflate_max_bytes || framed > vit::deflate_max_input()) continue;
        nb[i] = uint32_t((framed + 9 + 63) / 64);
        out_off[i] = total_out;
        total_out += (uint64_t(vit::deflate_bound(uint32_t(framed))) + 15) / 16 * 16;
        mine.push_back(i);
    }
    if (total_out > out_cap) return fail(h, VIT_ERR_RANGE, "the streams may need %llu bytes, %llu given", (unsigned long long)total_out, (unsigned long long)out_cap);

    constexpr uint64_t kChunkBytes = 256ull << 20;
    Staging& s = q.st[0];
    hipStream_t stream = q.stream[0];
    for (size_t pos = 0; pos < mine.size();) {
        uint64_t blocks = 0;
        size_t end = pos;
        while (end < mine.size() && (end == pos || (blocks + nb[mine[end]]) * 64 <= kChunkBytes)) blocks += nb[mine[end++]];
        const uint64_t cnt = end - pos, bytes = blocks * 64;
        const uint64_t out_lo = out_off[mine[pos]];
        const uint64_t out_hi = end < mine.size() ? out_off[mine[end]] : total_out;
        int rc;
        if ((rc = ensure_arena(q, s, bytes)) || (rc = ensure_meta(q, s, cnt))) return fail(h, rc, "device queue 0: %s", q.err.c_str());
        const uint64_t meta_bytes = carve_meta(s, cnt);
        if (out_hi - out_lo > q.z_out_cap) {
            if (q.z_out) { HIP_TRY(h, hipFree(q.z_out)); q.z_out = nullptr; q.z_out_cap = 0; }
            const uint64_t cap = grown(q.z_out_cap, out_hi - out_lo);
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&q.z_out), cap));
            q.z_out_cap = cap;
        }
        if (cnt > q.z_meta_cap) {
            if (q.z_meta_d) { HIP_TRY(h, hipFree(q.z_meta_d)); q.z_meta_d = nullptr; }
            if (q.z_meta_h) { HIP_TRY(h, hipHostFree(q.z_meta_h)); q.z_meta_h = nullptr; }
            const uint64_t cap = round_up(cnt, 65536);
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&q.z_meta_d), cap * 12));
            HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&q.z_meta_h), cap * 12, hipHostMallocDefault));
            q.z_meta_cap = cap;
        }
        uint64_t* h_zoff = reinterpret_cast<uint64_t*>(q.z_meta_h);
        uint32_t* h_zlen = reinterpret_cast<uint32_t*>(q.z_meta_h + q.z_meta_cap * 8);
        uint64_t* d_zoff = reinterpret_cast<uint64_t*>(q.z_meta_d);
        uint32_t* d_zlen = reinterpret_cast<uint32_t*>(q.z_meta_d + q.z_meta_cap * 8);

        std::vector<uint64_t> prefix(cnt + 1, 0);
        for (uint64_t k = 0; k < cnt; ++k) {
            const uint64_t i = mine[pos + k];
            s.h_off[k] = prefix[k] * 64;
            s.h_nb[k] = nb[i];
            prefix[k + 1] = prefix[k] + nb[i];
            h_zoff[k] = out_off[i] - out_lo;
        }
        const uint64_t pieces = std::min<uint64_t>({cnt, std::max<uint64_t>(bytes / (2u << 20), 1), uint64_t(h->pack_threads) * 4});
        const std::vector<uint64_t> cut = vit::balanced_cuts(cnt, int(pieces), prefix.data());
        h->pool->run(vit::WorkPool::HIGH, size_t(pieces), [&](size_t p) {
            for (uint64_t k = cut[p]; k < cut[p + 1]; ++k) {
                const uint64_t i = mine[pos + k];
                vit::pack_one(type, tl, data[i], len[i], s.h_arena + s.h_off[k], s.h_nb[k]);
            }
            vit::pack_fence();
        });
        const bool binned = vit::bin_order(s.h_nb, cnt, s.h_order);
        HIP_TRY(h, hipMemcpyAsync(s.d_arena, s.h_arena, bytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(h, hipMemcpyAsync(s.d_meta, s.h_meta, meta_bytes + (binned ? meta_bytes / 3 : 0), hipMemcpyHostToDevice, stream));
        HIP_TRY(h, hipMemcpyAsync(d_zoff, h_zoff, cnt * 8, hipMemcpyHostToDevice, stream));
        vit::DeflateBatch b;
        b.arena = s.d_arena; b.arena_bytes = bytes; b.offsets = s.d_off; b.nblocks = s.d_nb; b.n = uint32_t(cnt);
        b.order = binned ? s.d_order : nullptr;
        b.out = q.z_out; b.out_off = d_zoff; b.out_len = d_zlen;
        b.workspace = q.z_workspace; b.tables_raw = q.z_tables; b.status = q.d_status;
        HIP_TRY(h, hipStreamWaitEvent(stream, q.z_done, 0));
        HIP_TRY(h, vit::launch_deflate(b, q.z_lanes, stream));
        HIP_TRY(h, hipEventRecord(q.z_done, stream));
        HIP_TRY(h, hipMemcpyAsync(out + out_lo, q.z_out, out_hi - out_lo, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipMemcpyAsync(h_zlen, d_zlen, cnt * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
        for (uint64_t k = 0; k < cnt; ++k) out_len[mine[pos + k]] = h_zlen[k];
        pos = end;
    }
    uint32_t word = 0;
    if (int rc = read_status_word(h, q, &word)) return rc;
    if (word) return fail(h, VIT_ERR_DEVICE_STATUS, "device queue 0 flagged status %u", word);
    return VIT_OK;
    VIT_NOEXCEPT_END(h)
}

uint64_t vit_deflate_out_bytes(const vit_hasher* h, const char* type, const uint64_t* len, uint64_t n)
{
    const uint32_t tl = vit::type_len_checked(type);
    if (!h || !tl || (n && !len)) return 0;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t framed = tl + 1 + vit::decimal_digits(len[i]) + 1 + len[i];
        if (framed > h->deflate_max_bytes || framed > vit::deflate_max_input()) continue;
        total += (uint64_t(vit::deflate_bound(uint32_t(framed))) + 15) / 16 * 16;
    }
    return total;
}

uint32_t vit_deflate_bound(uint64_t framed_len)
{
    return framed_len > vit::deflate_max_input() ? 0u : vit::deflate_bound(uint32_t(framed_len));
}

int vit_bin_order_device(vit_hasher* h, int dev, const uint32_t* d_nblocks, uint64_t n, uint32_t* d_order, void* stream)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!d_nblocks || !d_order) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    HIP_TRY(h, hipSetDevice(q->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : q->stream[0];
    const size_t need = vit::bin_order_scratch_bytes(uint32_t(n));
    if (need > q->sort_scratch_bytes) {
        HIP_TRY(h, hipStreamSynchronize(s));              // an earlier sort may still be using the old block
        if (q->sort_scratch) { HIP_TRY(h, hipFree(q->sort_scratch)); q->sort_scratch = nullptr; q->sort_scratch_bytes = 0; }
        HIP_TRY(h, hipMalloc(&q->sort_scratch, need));
        q->sort_scratch_bytes = need;
    }
    HIP_TRY(h, vit::bin_order_device(d_nblocks, uint32_t(n), d_order, q->sort_scratch, q->sort_scratch_bytes, s));
    return VIT_OK;
}

// Waits for the queue's own three streams only (a launch the caller put on a
// stream of its own is the caller's to wait for), then reads the word with an
// asynchronous copy on one of them: no other stream of the device is stalled.
int vit_device_status(vit_hasher* h, int dev, uint32_t* status)
{
    Queue* q = queue_of(h, dev);
    if (!q || !status) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    for (hipStream_t st : q->stream) HIP_TRY(h, hipStreamSynchronize(st));
    uint32_t word = 0;
    if (int rc = read_status_word(h, *q, &word)) return rc;
    *status = word | q->status_seen;
    q->status_seen = 0;
    return VIT_OK;
}

int vit_device_alloc(vit_hasher* h, int dev, uint64_t bytes, void** out)
{
    Queue* q = queue_of(h, dev);
    if (!q || !out) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    *out = nullptr;
    HIP_TRY(h, hipSetDevice(q->device));
    HIP_TRY(h, hipMalloc(out, bytes ? bytes : 256));
    return VIT_OK;
}

int vit_device_free(vit_hasher* h, int dev, void* p)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    if (p) HIP_TRY(h, hipFree(p));
    return VIT_OK;
}

int vit_device_upload(vit_hasher* h, int dev, void* d_dst, const void* src, uint64_t bytes)
{
    Queue* q = queue_of(h, dev);
    if (!q || (bytes && (!d_dst || !src))) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    HIP_TRY(h, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return VIT_OK;
}

int vit_device_download(vit_hasher* h, int dev, void* dst, const void* d_src, uint64_t bytes)
{
    Queue* q = queue_of(h, dev);
    if (!q || (bytes && (!dst || !d_src))) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    HIP_TRY(h, hipStreamSynchronize(q->stream[0]));
    HIP_TRY(h, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return VIT_OK;
}

int vit_device_sync(vit_hasher* h, int dev)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    HIP_TRY(h, hipDeviceSynchronize());
    return VIT_OK;
}

int vit_synth_arena_device(vit_hasher* h, int dev, const char* type, uint64_t seed, uint64_t first_index, uint64_t n,
                           uint64_t len_each, const uint64_t* d_lens, uint64_t* d_offsets, uint32_t* d_nblocks,
                           void* d_arena, uint64_t arena_bytes, void* stream)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    const uint32_t tl = vit::type_len_checked(type);
    if (!tl) return fail(h, VIT_ERR_ARG, "bad object type");
    if (n == 0) return VIT_OK;
    if (!d_offsets || !d_nblocks || !d_arena) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 - 1 objects in one call");
    if (!d_lens) {
        const uint64_t need = vit_object_blocks(type, len_each) * 64 * n;
        if (need > arena_bytes) return fail(h, VIT_ERR_RANGE, "arena holds %llu bytes, %llu needed",
                                            (unsigned long long)arena_bytes, (unsigned long long)need);
    }
    vit::SynthSpec s;
    s.seed = seed; s.first_index = first_index; s.n = uint32_t(n); s.len_each = len_each;
    s.lens = d_lens; s.offsets = d_offsets; s.nblocks = d_nblocks;
    s.arena = static_cast<uint8_t*>(d_arena); s.arena_bytes = arena_bytes;
    std::memset(s.type, 0, sizeof s.type);
    std::memcpy(s.type, type, tl);
    HIP_TRY(h, hipSetDevice(q->device));
    HIP_TRY(h, vit::launch_synth(s, stream ? static_cast<hipStream_t>(stream) : q->stream[0]));
    return VIT_OK;
}

int vit_time_hash_device(vit_hasher* h, int dev, const void* d_arena, uint64_t arena_bytes, const uint64_t* d_offsets,
                         const uint32_t* d_nblocks, const uint32_t* d_order, uint64_t n, uint8_t* d_digests,
                         int warmup, int iters, float* ms_each)
{
    Queue* q = queue_of(h, dev);
    if (!q || iters < 0 || warmup < 0 || (iters && !ms_each)) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    VIT_NOEXCEPT_BEGIN
    // Timing events are made by the first call, warm-up calls (iters == 0) included, and kept:
    // a timed run that was preceded by a warm-up call creates none.  (Not at vit_hasher_create:
    // a hundred events cost milliseconds that a process hashing one small tree would pay for nothing.)
    std::vector<hipEvent_t>& ev = q->timing;
    while (ev.size() < std::max<size_t>(size_t(iters) + 1, 257)) {
        hipEvent_t e = nullptr;
        HIP_TRY(h, hipEventCreate(&e));
        ev.push_back(e);
    }
    for (int i = 0; i < warmup; ++i)
        if (int rc = vit_hash_arena_device(h, dev, d_arena, arena_bytes, d_offsets, d_nblocks, d_order, n, d_digests, nullptr)) return rc;
    HIP_TRY(h, hipStreamSynchronize(q->stream[0]));
    if (iters == 0) return VIT_OK;
    // The launches are queued back to back with an event between each two and the
    // host waits once, at the end: no host round trip separates the kernels.
    HIP_TRY(h, hipEventRecord(ev[0], q->stream[0]));
    for (int i = 0; i < iters; ++i) {
        if (int rc = vit_hash_arena_device(h, dev, d_arena, arena_bytes, d_offsets, d_nblocks, d_order, n, d_digests, nullptr)) return rc;
        HIP_TRY(h, hipEventRecord(ev[size_t(i) + 1], q->stream[0]));
    }
    HIP_TRY(h, hipEventSynchronize(ev[size_t(iters)]));
    for (int i = 0; i < iters; ++i) {
        HIP_TRY(h, hipEventElapsedTime(&ms_each[i], ev[size_t(i)], ev[size_t(i) + 1]));
        h->last_kernel_ns = uint64_t(double(ms_each[i]) * 1e6);
    }
    return VIT_OK;
    VIT_NOEXCEPT_END(h)
}

int vit_valu_peak(vit_hasher* h, int dev, uint32_t n_lanes, uint32_t blocks_per_lane, int iters, float* ms)
{
    Queue* q = queue_of(h, dev);
    if (!q || !ms || iters < 1 || n_lanes == 0) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    uint32_t* d_out = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&d_out), uint64_t(n_lanes) * 4));
    float best = 1e30f;
    for (int i = 0; i < iters + 1; ++i) {            // first launch is a warm-up
        hipError_t e = hipEventRecord(q->ev_a, q->stream[0]);
        if (e == hipSuccess) e = vit::launch_valu_peak(d_out, n_lanes, blocks_per_lane, q->stream[0]);
        if (e == hipSuccess) e = hipEventRecord(q->ev_b, q->stream[0]);
        if (e == hipSuccess) e = hipEventSynchronize(q->ev_b);
        float t = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, q->ev_a, q->ev_b);
        if (e != hipSuccess) { (void)hipFree(d_out); return fail(h, VIT_ERR_HIP, "valu_peak: %s", hipGetErrorString(e)); }
        if (i > 0 && t < best) best = t;
    }
    HIP_TRY(h, hipFree(d_out));
    *ms = best;
    return VIT_OK;
}

}  // extern "C"
