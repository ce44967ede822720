// hasher.cpp — the C ABI of include/vit_hash.h over the gfx950 kernels.
//
// Host pipeline of vit_hash_objects(), per device queue:
//
//   pack chunk k+1 into pinned arena B   |  H2D copy, kernel, digest D2H of
//   (header, content, padding; threads)  |  chunk k from pinned arena A (stream)
//
// Two staging sets per queue alternate, so host packing overlaps the copy
// engine and the kernel; a third set and stream carry the chunks of long
// objects, and the longest objects of a skewed batch are hashed by host
// threads meanwhile (see vit_hash_objects).  Objects are assigned to queues by
// padded size (bytes), not by count; nothing is exchanged between devices
// because every digest depends on its own object only and the consumer of the
// digests (tree assembly, reference src/commit.cpp:164-172) runs on the host.
#include "../../include/vit_hash.h"
#include "host_util.h"
#include "kernels.h"
#include "packer.h"
#include "sha1_host.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
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
    hipEvent_t done = nullptr;
    hipEvent_t mark[4] = {nullptr, nullptr, nullptr, nullptr};   // trace only: before copy, after copy, after kernel, after read-back
    double host_ms[4] = {0, 0, 0, 0};                            // trace only: submit begin, previous chunk retired, packed, queued
    bool in_flight = false, traced = false;
    std::vector<uint64_t> obj;                            // caller indices of the chunk in flight
};

struct Queue {
    int device = 0;
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};   // [0],[1]: alternating chunks; [2]: long objects
    uint32_t* d_status = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    void* sort_scratch = nullptr; size_t sort_scratch_bytes = 0;   // vit_bin_order_device
    std::vector<hipEvent_t> timing;                       // pool for vit_time_hash_device
    uint64_t* wave_trace = nullptr;                       // diagnostics: device buffer, 32 bytes per wave, that the next
                                                          // launches fill with their waves' timeline (tools/wave_timeline.py)
    std::string trace_log;                                // per-chunk timeline of the last call, when tracing
    clk::time_point trace_t0;
    Staging st[3];
    std::string err;                                      // set by the queue's worker thread
};

}  // namespace

struct vit_hasher {
    std::vector<Queue> queues;
    mutable std::string err;
    int kernel = VIT_KERNEL_LDS;
    // Which objects go to host threads instead of a GPU lane: 0 never, 1 auto
    // (threshold derived from the batch, see vit_hash_objects), 2 fixed threshold.
    int host_policy = 1;
    uint64_t host_threshold = 0;
    uint64_t chunk_bytes = 64ull << 20;
    uint64_t chunk_object_multiple = 8192;
    bool trace = false;
    int pack_threads = 8;
    std::atomic<uint64_t> objects_gpu{0}, objects_host{0}, bytes_host{0}, blocks_gpu{0}, bytes_h2d{0}, launches{0};
    uint64_t last_host_threshold = 0;
    uint64_t last_kernel_ns = 0, last_pack_ns = 0, last_total_ns = 0;
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

int ensure_arena(Queue& q, Staging& s, uint64_t bytes)
{
    if (bytes <= s.h_cap && bytes <= s.d_cap) return VIT_OK;
    // Freeing and re-allocating staging costs milliseconds: start at 8 MiB and at least double.
    const uint64_t want = std::max<uint64_t>({bytes, 8ull << 20, std::min<uint64_t>(2 * s.h_cap, 1ull << 30)});
    const uint64_t cap = round_up(want, 1ull << 20);
    if (s.h_arena) { HIP_TRYQ(q, hipHostFree(s.h_arena)); s.h_arena = nullptr; s.h_cap = 0; }
    if (s.d_arena) { HIP_TRYQ(q, hipFree(s.d_arena)); s.d_arena = nullptr; s.d_cap = 0; }
    HIP_TRYQ(q, hipHostMalloc(reinterpret_cast<void**>(&s.h_arena), cap, hipHostMallocDefault));
    s.h_cap = cap;
    HIP_TRYQ(q, hipMalloc(reinterpret_cast<void**>(&s.d_arena), cap));
    s.d_cap = cap;
    return VIT_OK;
}

// Offsets, block counts, order and digests of a chunk live in ONE pinned and ONE
// device allocation (36 bytes per object): allocations cost about a millisecond
// each, which is most of a small first call.
int ensure_meta(Queue& q, Staging& s, uint64_t n)
{
    if (n <= s.m_cap) return VIT_OK;
    const uint64_t cap = round_up(n < 4096 ? 4096 : n, 4096);
    if (s.h_meta) { HIP_TRYQ(q, hipHostFree(s.h_meta)); s.h_meta = nullptr; }
    if (s.d_meta) { HIP_TRYQ(q, hipFree(s.d_meta)); s.d_meta = nullptr; }
    s.m_cap = 0;
    HIP_TRYQ(q, hipHostMalloc(reinterpret_cast<void**>(&s.h_meta), cap * 36, hipHostMallocDefault));
    HIP_TRYQ(q, hipMalloc(reinterpret_cast<void**>(&s.d_meta), cap * 36));
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
    return b;
}

// Collect a finished chunk: wait for it and scatter its digests to the caller.
int retire(Queue& q, Staging& s, uint8_t* digests)
{
    if (!s.in_flight) return VIT_OK;
    HIP_TRYQ(q, hipEventSynchronize(s.done));
    for (size_t i = 0; i < s.obj.size(); ++i) std::memcpy(digests + s.obj[i] * 20, s.h_dig + i * 20, 20);
    s.in_flight = false;
    if (s.traced) {
        float g[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) HIP_TRYQ(q, hipEventElapsedTime(&g[k], q.ev_a, s.mark[k]));
        char line[256];
        std::snprintf(line, sizeof line,
                      "set %d objs %8zu | host ms: begin %8.2f retired-prev %8.2f packed %8.2f queued %8.2f collected %8.2f"
                      " | gpu ms: copy %8.2f..%8.2f kernel ..%8.2f read-back ..%8.2f\n",
                      int(&s - q.st), s.obj.size(), s.host_ms[0], s.host_ms[1], s.host_ms[2], s.host_ms[3],
                      double(ns_since(q.trace_t0)) * 1e-6, g[0], g[1], g[2], g[3]);
        q.trace_log += line;
    }
    return VIT_OK;
}

struct Job {
    vit_hasher* h;
    const char* type; uint32_t tl;
    const uint8_t* const* data; const uint64_t* len;
    const uint32_t* nb;                    // block count per object (all objects)
    uint8_t* digests;
};

// Pack one chunk (objects idx[0..cnt)) into staging set s and queue its copy,
// kernel and digest read-back on `stream`.
int submit_chunk(Queue& q, const Job& job, Staging& s, hipStream_t stream, const uint64_t* idx, uint64_t cnt,
                 uint64_t* pack_ns)
{
    vit_hasher* h = job.h;
    const double begin_ms = double(ns_since(q.trace_t0)) * 1e-6;
    int rc = retire(q, s, job.digests);
    if (rc) return rc;
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
    std::vector<uint64_t> prefix(cnt + 1, 0);
    for (uint64_t i = 0; i < cnt; ++i) {
        s.h_off[i] = prefix[i] * 64;
        s.h_nb[i] = job.nb[idx[i]];
        prefix[i + 1] = prefix[i] + s.h_nb[i];
    }
    const int threads = bytes < (4u << 20) ? 1 : h->pack_threads;
    vit::parallel_ranges(cnt, threads, prefix.data(), [&](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            const uint64_t o = idx[i];
            vit::pack_one(job.type, job.tl, job.data[o], job.len[o], s.h_arena + s.h_off[i], s.h_nb[i]);
        }
        vit::pack_fence();
    });
    const bool binned = vit::bin_order(s.h_nb, cnt, s.h_order);
    *pack_ns += ns_since(t0);

    if (trace) { s.host_ms[2] = double(ns_since(q.trace_t0)) * 1e-6; HIP_TRYQ(q, hipEventRecord(s.mark[0], stream)); }
    HIP_TRYQ(q, hipMemcpyAsync(s.d_arena, s.h_arena, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRYQ(q, hipMemcpyAsync(s.d_meta, s.h_meta, meta_bytes + (binned ? meta_bytes / 3 : 0), hipMemcpyHostToDevice, stream));
    if (trace) HIP_TRYQ(q, hipEventRecord(s.mark[1], stream));
    HIP_TRYQ(q, vit::launch_sha1(vit::KernelVariant(h->kernel),
                                 make_batch(q, s.d_arena, bytes, s.d_off, s.d_nb, binned ? s.d_order : nullptr, cnt,
                                            s.d_dig),
                                 stream));
    if (trace) HIP_TRYQ(q, hipEventRecord(s.mark[2], stream));
    HIP_TRYQ(q, hipMemcpyAsync(s.h_dig, s.d_dig, cnt * 20, hipMemcpyDeviceToHost, stream));
    if (trace) { HIP_TRYQ(q, hipEventRecord(s.mark[3], stream)); s.host_ms[3] = double(ns_since(q.trace_t0)) * 1e-6; }
    HIP_TRYQ(q, hipEventRecord(s.done, stream));
    s.in_flight = true;
    h->objects_gpu += cnt; h->blocks_gpu += blocks; h->bytes_h2d += bytes + cnt * 16; ++h->launches;
    return VIT_OK;
}

// Objects of at least this many blocks (256 KiB) are "long": a lane hashes about
// 50 MB/s, so each of them holds its wave for milliseconds.  They are launched
// first, on a stream and staging set of their own, and the short objects'
// chunks stream past them on the other two.
constexpr uint32_t kLongBlocks = 4096;

// Hash the objects `mine` (caller indices, in order) on queue q.
int run_queue(Queue& q, const Job& job, const std::vector<uint64_t>& mine, uint64_t* pack_ns)
{
    vit_hasher* h = job.h;
    HIP_TRYQ(q, hipSetDevice(q.device));
    q.trace_t0 = clk::now();
    q.trace_log.clear();
    if (h->trace) HIP_TRYQ(q, hipEventRecord(q.ev_a, q.stream[0]));
    std::vector<uint64_t> longs, rest;
    for (uint64_t i : mine) (job.nb[i] >= kLongBlocks ? longs : rest).push_back(i);
    std::stable_sort(longs.begin(), longs.end(), [&](uint64_t a, uint64_t b) { return job.nb[a] > job.nb[b]; });

    // Chunk boundaries: as many objects as fit the staging budget (at least one).
    auto cut = [&](const std::vector<uint64_t>& v, uint64_t budget) {
        std::vector<std::pair<size_t, size_t>> out;
        for (size_t pos = 0; pos < v.size();) {
            uint64_t blocks = 0;
            size_t end = pos;
            while (end < v.size() && (end == pos || (blocks + job.nb[v[end]]) * 64 <= budget)) blocks += job.nb[v[end++]];
            out.emplace_back(pos, end);
            pos = end;
        }
        return out;
    };
    // A chunk's kernel lasts at least as long as its longest object on one lane
    // (45 MB/s; 1.5 ms for 64 KiB), while its transfer runs about 1000 times
    // faster per byte, and measured transfers slow down while another chunk's
    // kernel is running.  Chunks therefore grow with the longest object they may
    // hold, to chunk_object_multiple (8192) times its size: 512 MiB for 64 KiB
    // objects gave twice the throughput of 64 MiB on a skewed 15 GB batch.
    // Bounds: [configured chunk size, 1 GiB], and at most 1/8 of the batch so
    // that packing, copying and hashing still overlap.
    uint64_t rest_bytes = 0, rest_longest = 0;
    for (uint64_t i : rest) {
        rest_bytes += uint64_t(job.nb[i]) * 64;
        rest_longest = std::max<uint64_t>(rest_longest, job.nb[i]);
    }
    uint64_t chunk = std::min<uint64_t>(rest_longest * 64 * h->chunk_object_multiple, rest_bytes / 8);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, h->chunk_bytes), 1ull << 30);
    const auto long_chunks = cut(longs, std::max<uint64_t>(chunk, 256ull << 20));
    const auto rest_chunks = cut(rest, chunk);

    // Long chunks go first and then whenever their staging set has drained, so
    // that their lane-bound kernels run beside the short objects' chunks and
    // not after them.
    int rc;
    size_t next_long = 0;
    auto submit_long = [&]() -> int {
        const auto& c = long_chunks[next_long++];
        return submit_chunk(q, job, q.st[2], q.stream[2], longs.data() + c.first, c.second - c.first, pack_ns);
    };
    if (!long_chunks.empty() && (rc = submit_long())) return rc;
    int turn = 0;
    for (const auto& c : rest_chunks) {
        if ((rc = submit_chunk(q, job, q.st[turn], q.stream[turn], rest.data() + c.first, c.second - c.first, pack_ns)))
            return rc;
        turn ^= 1;
        if (next_long < long_chunks.size() && hipEventQuery(q.st[2].done) == hipSuccess && (rc = submit_long())) return rc;
    }
    while (next_long < long_chunks.size())
        if ((rc = submit_long())) return rc;
    for (Staging& s : q.st)
        if ((rc = retire(q, s, job.digests))) return rc;
    return VIT_OK;
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
    std::vector<uint64_t> load(nq, 0), large, small;
    for (uint64_t i : idx) (nb[i] >= big ? large : small).push_back(i);
    std::stable_sort(large.begin(), large.end(), [&](uint64_t a, uint64_t b) { return nb[a] > nb[b]; });
    for (uint64_t i : large) {
        size_t best = 0;
        for (size_t k = 1; k < nq; ++k) if (load[k] < load[best]) best = k;
        out[best].push_back(i); load[best] += nb[i];
    }
    size_t k = 0;
    for (uint64_t i : small) {
        while (k + 1 < nq && load[k] >= fair) ++k;
        out[k].push_back(i); load[k] += nb[i];
    }
    for (auto& v : out) std::sort(v.begin(), v.end());
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
    h->queues.resize(ids.size());
    unsigned hw = std::thread::hardware_concurrency();
    h->pack_threads = int(hw ? (hw > 16 ? 16 : hw) : 4);
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
        if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void**>(&q.d_status), 256);
        if (r == hipSuccess) r = hipMemset(q.d_status, 0, 256);
        if (r == hipSuccess) r = hipEventCreate(&q.ev_a);
        if (r == hipSuccess) r = hipEventCreate(&q.ev_b);
        // Timing events are made here so that a timed run creates none.
        for (int i = 0; i < 129 && r == hipSuccess; ++i) {
            hipEvent_t e = nullptr;
            r = hipEventCreate(&e);
            if (r == hipSuccess) q.timing.push_back(e);
        }
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
        if (q.ev_a) (void)hipEventDestroy(q.ev_a);
        if (q.ev_b) (void)hipEventDestroy(q.ev_b);
        if (q.sort_scratch) (void)hipFree(q.sort_scratch);
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
    const std::string k(name);
    if (k == "kernel") {
        if (value > 5) return fail(h, VIT_ERR_ARG, "unknown kernel %llu", (unsigned long long)value);
        h->kernel = int(value);
    } else if (k == "host_threshold_bytes") { h->host_threshold = value; h->host_policy = value ? 2 : 0; }
    else if (k == "host_auto") { h->host_policy = value ? 1 : (h->host_threshold ? 2 : 0); }
    else if (k == "chunk_bytes") h->chunk_bytes = value < (1u << 20) ? (1u << 20) : value;
    else if (k == "chunk_object_multiple") h->chunk_object_multiple = value;
    else if (k == "trace") h->trace = value != 0;
    else if (k == "wave_trace_ptr") { for (Queue& q : h->queues) q.wave_trace = reinterpret_cast<uint64_t*>(uintptr_t(value)); }
    else if (k == "pack_threads") h->pack_threads = value < 1 ? 1 : int(value > 256 ? 256 : value);
    else return fail(h, VIT_ERR_ARG, "unknown option '%s'", name);
    return VIT_OK;
}

int vit_hasher_get_stat(const vit_hasher* h, const char* name, uint64_t* value)
{
    if (!h || !name || !value) return VIT_ERR_ARG;
    const std::string k(name);
    if (k == "objects_gpu") *value = h->objects_gpu;
    else if (k == "objects_host") *value = h->objects_host;
    else if (k == "bytes_host") *value = h->bytes_host;
    else if (k == "last_host_threshold") *value = h->last_host_threshold;
    else if (k == "blocks_gpu") *value = h->blocks_gpu;
    else if (k == "bytes_h2d") *value = h->bytes_h2d;
    else if (k == "launches") *value = h->launches;
    else if (k == "last_kernel_ns") *value = h->last_kernel_ns;
    else if (k == "last_pack_ns") *value = h->last_pack_ns;
    else if (k == "last_total_ns") *value = h->last_total_ns;
    else if (k == "host_sha_accelerated") *value = vit::HostSha1::accelerated();
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
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 objects in one call");
    const auto t0 = clk::now();
    VIT_NOEXCEPT_BEGIN

    // One SHA-1 stream is serial: a lane sustains about 45 MB/s and a host core
    // with SHA extensions about 1.8 GB/s, while a whole batch moves through
    // pack + PCIe at about 40 GB/s per GPU.  The auto policy balances the two
    // sides: starting from the longest object, an object of 64 KiB or more
    // moves to the host threads as long as the host side (its bytes over the
    // threads, never less than its longest object on one core) still finishes
    // before the GPU side would with the object on it (transfer time of its
    // bytes, never less than its longest object on one lane).  Packing is
    // memory-bound and light, so the same threads carry both.
    uint64_t threshold = 0;
    if (h->host_policy == 2) threshold = h->host_threshold;
    else if (h->host_policy == 1) {
        constexpr double kLaneRate = 45e6, kQueueRate = 40e9;
        const double core_rate = vit::HostSha1::accelerated() ? 1.8e9 : 0.45e9;
        const double host_rate = core_rate * double(h->pack_threads);
        uint64_t total = 0;
        std::vector<uint64_t> cand;
        for (uint64_t i = 0; i < n; ++i) {
            total += len[i];
            if (len[i] >= (64u << 10)) cand.push_back(len[i]);
        }
        std::sort(cand.begin(), cand.end(), std::greater<uint64_t>());
        double gpu_bytes = double(total), host_bytes = 0;
        for (uint64_t L : cand) {
            const double gpu_time = std::max(gpu_bytes / (kQueueRate * double(h->queues.size())), double(L) / kLaneRate);
            const double host_time = std::max((host_bytes + double(L)) / host_rate, double(cand[0]) / core_rate);
            if (host_time >= gpu_time) break;
            gpu_bytes -= double(L); host_bytes += double(L);
            threshold = L - 1;
        }
    }
    h->last_host_threshold = threshold;

    std::vector<uint32_t> nb(n);
    std::vector<uint64_t> gpu_idx, host_idx;
    gpu_idx.reserve(n);
    uint64_t host_bytes = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (len[i] && !data[i]) return fail(h, VIT_ERR_ARG, "object %llu: null data with non-zero length", (unsigned long long)i);
        const uint64_t hl = tl ? tl + 1 + vit::decimal_digits(len[i]) + 1 : 0;
        const uint64_t b = (hl + len[i] + 9 + 63) / 64;
        if (threshold && len[i] > threshold) { host_idx.push_back(i); host_bytes += len[i]; nb[i] = 0; continue; }
        if (b > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "object %llu too large for one lane", (unsigned long long)i);
        nb[i] = uint32_t(b);
        gpu_idx.push_back(i);
    }

    // Oversized objects: one SHA-1 stream each on host threads, beside the GPU batch.
    ThreadGroup host_workers;
    std::atomic<size_t> next_host{0};
    if (!host_idx.empty()) {
        // longest first, so the tail of the host work is short
        std::stable_sort(host_idx.begin(), host_idx.end(), [&](uint64_t a, uint64_t b) { return len[a] > len[b]; });
        const size_t nt = std::min<size_t>(host_idx.size(), size_t(h->pack_threads));
        for (size_t t = 0; t < nt; ++t)
            host_workers.t.emplace_back([&] {
                for (size_t k; (k = next_host++) < host_idx.size();) {
                    const uint64_t i = host_idx[k];
                    vit::host_object_id(type, data[i], len[i], digests + i * 20);
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
    if (h->queues.size() == 1) {
        rcs[0] = run_queue(h->queues[0], job, parts[0], &pack_ns[0]);
    } else {
        ThreadGroup workers;
        for (size_t k = 0; k < h->queues.size(); ++k)
            workers.t.emplace_back([&, k] {
                try { rcs[k] = run_queue(h->queues[k], job, parts[k], &pack_ns[k]); }
                catch (const std::exception& e) { h->queues[k].err = e.what(); rcs[k] = VIT_ERR_NOMEM; }
            });
        workers.join();
    }
    host_workers.join();
    h->last_pack_ns = 0;
    for (uint64_t p : pack_ns) h->last_pack_ns = std::max(h->last_pack_ns, p);
    h->last_total_ns = ns_since(t0);
    if (h->trace)
        for (size_t k = 0; k < h->queues.size(); ++k)
            std::fprintf(stderr, "[vit trace] queue %zu, %.2f ms in all, %zu objects on host threads\n%s", k,
                         double(h->last_total_ns) * 1e-6, host_idx.size(), h->queues[k].trace_log.c_str());
    for (size_t k = 0; k < rcs.size(); ++k)
        if (rcs[k] != VIT_OK) return fail(h, rcs[k], "device queue %zu: %s", k, h->queues[k].err.c_str());
    for (size_t k = 0; k < h->queues.size(); ++k) {
        uint32_t st = 0;
        int rc = vit_device_status(h, int(k), &st);
        if (rc) return rc;
        if (st) return fail(h, VIT_ERR_DEVICE_STATUS, "device queue %zu flagged status %u", k, st);
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

int vit_hash_arena(vit_hasher* h, int dev, const void* arena, uint64_t arena_bytes, const uint64_t* offsets,
                   const uint32_t* nblocks, uint64_t n, uint8_t* digests)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!arena || !offsets || !nblocks || !digests) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 objects in one call");
    VIT_NOEXCEPT_BEGIN
    uint64_t lo = ~0ull, hi = 0, blocks = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t sz = uint64_t(nblocks[i]) * 64;
        if (offsets[i] & 15) return fail(h, VIT_ERR_RANGE, "object %llu: offset not 16-byte aligned", (unsigned long long)i);
        if (offsets[i] > arena_bytes || sz > arena_bytes - offsets[i])
            return fail(h, VIT_ERR_RANGE, "object %llu: %llu blocks at offset %llu exceed the arena (%llu bytes)",
                        (unsigned long long)i, (unsigned long long)nblocks[i], (unsigned long long)offsets[i],
                        (unsigned long long)arena_bytes);
        lo = std::min(lo, offsets[i]); hi = std::max(hi, offsets[i] + sz); blocks += nblocks[i];
    }
    lo &= ~uint64_t(63);
    const uint64_t span = hi > lo ? hi - lo : 0;
    HIP_TRY(h, hipSetDevice(q->device));
    Staging& s = q->st[0];
    // Device-side staging only: the caller's arena is the host side.
    if (span > s.d_cap) {
        if (s.d_arena) { HIP_TRY(h, hipFree(s.d_arena)); s.d_arena = nullptr; s.d_cap = 0; }
        if (s.h_arena) { HIP_TRY(h, hipHostFree(s.h_arena)); s.h_arena = nullptr; s.h_cap = 0; }
        const uint64_t cap = round_up(span < (1ull << 20) ? (1ull << 20) : span, 1ull << 20);
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_arena), cap));
        s.d_cap = cap;
    }
    if (int rc = ensure_meta(*q, s, n)) return fail(h, rc, "%s", q->err.c_str());
    const uint64_t meta_bytes = carve_meta(s, n);
    for (uint64_t i = 0; i < n; ++i) { s.h_off[i] = offsets[i] - lo; s.h_nb[i] = nblocks[i]; }
    const bool binned = vit::bin_order(s.h_nb, n, s.h_order);
    HIP_TRY(h, hipMemcpyAsync(s.d_arena, static_cast<const uint8_t*>(arena) + lo, span, hipMemcpyHostToDevice, q->stream[0]));
    HIP_TRY(h, hipMemcpyAsync(s.d_meta, s.h_meta, meta_bytes + (binned ? meta_bytes / 3 : 0), hipMemcpyHostToDevice, q->stream[0]));
    HIP_TRY(h, vit::launch_sha1(vit::KernelVariant(h->kernel),
                                make_batch(*q, s.d_arena, span, s.d_off, s.d_nb, binned ? s.d_order : nullptr, n, s.d_dig),
                                q->stream[0]));
    HIP_TRY(h, hipMemcpyAsync(s.h_dig, s.d_dig, n * 20, hipMemcpyDeviceToHost, q->stream[0]));
    HIP_TRY(h, hipStreamSynchronize(q->stream[0]));
    std::memcpy(digests, s.h_dig, n * 20);
    h->objects_gpu += n; h->blocks_gpu += blocks; h->bytes_h2d += span + n * 16; ++h->launches;
    uint32_t st = 0;
    if (int rc = vit_device_status(h, dev, &st)) return rc;
    if (st) return fail(h, VIT_ERR_DEVICE_STATUS, "device queue %d flagged status %u", dev, st);
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
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 objects in one call");
    if (reinterpret_cast<uintptr_t>(d_arena) & 15) return fail(h, VIT_ERR_RANGE, "arena not 16-byte aligned");
    HIP_TRY(h, hipSetDevice(q->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : q->stream[0];
    HIP_TRY(h, vit::launch_sha1(vit::KernelVariant(h->kernel),
                                make_batch(*q, d_arena, arena_bytes, d_offsets, d_nblocks, d_order, n, d_digests), s));
    ++h->launches; h->objects_gpu += n;
    return VIT_OK;
}

int vit_bin_order_device(vit_hasher* h, int dev, const uint32_t* d_nblocks, uint64_t n, uint32_t* d_order, void* stream)
{
    Queue* q = queue_of(h, dev);
    if (!q) return h ? fail(h, VIT_ERR_ARG, "no device queue %d", dev) : VIT_ERR_ARG;
    if (n == 0) return VIT_OK;
    if (!d_nblocks || !d_order) return fail(h, VIT_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 objects in one call");
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

int vit_device_status(vit_hasher* h, int dev, uint32_t* status)
{
    Queue* q = queue_of(h, dev);
    if (!q || !status) return h ? fail(h, VIT_ERR_ARG, "bad arguments") : VIT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(q->device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(status, q->d_status, 4, hipMemcpyDeviceToHost));
    if (*status) HIP_TRY(h, hipMemset(q->d_status, 0, 4));
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
    if (n > 0xFFFFFFFFull) return fail(h, VIT_ERR_RANGE, "more than 2^32 objects in one call");
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
    for (int i = 0; i < warmup; ++i)
        if (int rc = vit_hash_arena_device(h, dev, d_arena, arena_bytes, d_offsets, d_nblocks, d_order, n, d_digests, nullptr)) return rc;
    HIP_TRY(h, hipStreamSynchronize(q->stream[0]));
    if (iters == 0) return VIT_OK;
    // The launches are queued back to back with an event between each two and the
    // host waits once, at the end: no host round trip separates the kernels.
    VIT_NOEXCEPT_BEGIN
    std::vector<hipEvent_t>& ev = q->timing;
    while (ev.size() < size_t(iters) + 1) {
        hipEvent_t e = nullptr;
        HIP_TRY(h, hipEventCreate(&e));
        ev.push_back(e);
    }
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
