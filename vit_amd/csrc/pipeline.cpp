// pipeline.cpp — how a batch gets through the device queues.
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
#include "hasher_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>

namespace vit {

// Staging capacity: freeing and re-allocating costs milliseconds, so start at
// 8 MiB and at least double.
uint64_t grown(uint64_t have, uint64_t bytes)
{
    const uint64_t want = std::max<uint64_t>({bytes, 8ull << 20, std::min<uint64_t>(2 * have, 1ull << 30)});
    return round_up(want, 1ull << 20);
}

// The process-wide entry of a GPU: made by the first queue on it (device current),
// never freed (the scratch array is the library's for the life of the process anyway).
StaticDevice* static_device(int device, hipError_t* err)
{
    static std::mutex m;
    static std::map<int, std::unique_ptr<StaticDevice>> table;
    std::lock_guard<std::mutex> lk(m);
    auto it = table.find(device);
    if (it != table.end()) { *err = hipSuccess; return it->second.get(); }
    std::unique_ptr<StaticDevice> d(new StaticDevice());
    hipError_t r = static_scratch_address(&d->scratch);
    if (r == hipSuccess) r = hipMemset(d->scratch, 0, 256);                 // its status word
    if (r == hipSuccess) r = hipDeviceSynchronize();
    if (r == hipSuccess) r = hipEventCreateWithFlags(&d->done, hipEventDisableTiming);
    *err = r;
    if (r != hipSuccess) return nullptr;
    return (table[device] = std::move(d)).get();
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
    const vit::KernelVariant v = vit::KernelVariant(h->opt.kernel);
    if (h->opt.kernel != VIT_KERNEL_LDS || !b.order || !h->opt.sched_waves || groups < h->opt.sched_min_groups)
        return vit::launch_sha1(v, b, stream);
    const uint32_t pct = h->opt.sched_tail_percent;
    if (pct && longest_blocks && !vit::tail_dominates(longest_blocks, b.arena_bytes, q.simds, pct))
        return vit::launch_sha1(v, b, stream);
    const uint32_t state = staging_set >= 0 ? uint32_t(staging_set) : 3u + q.sched_next++ % (kSchedStates - 3u);
    if (!pct || longest_blocks) return vit::launch_sha1_sched(b, &q.d_sched[state], h->opt.sched_waves, h->opt.sched_share_below, stream);
    vit::DeviceBatch gated = b;
    gated.gate_percent = pct;
    gated.gate_simds = q.simds;
    const hipError_t e = vit::launch_sha1_sched(gated, &q.d_sched[state], h->opt.sched_waves, h->opt.sched_share_below, stream);
    return e != hipSuccess ? e : vit::launch_sha1(v, gated, stream);
}

// Collect a finished chunk: wait for it, scatter its digests to the caller and
// note the status word it brought back.
// While the chunk is not done, the waiting thread works through tasks of `help` (short host-lane tasks of
// the same call, when there are any) instead of sleeping: the threads that drive the queues count against
// the call's thread budget, and with eight queues they would otherwise be half of it, asleep.
int retire(Queue& q, Staging& s, uint8_t* digests, WorkPool* pool = nullptr, const WorkPool::Ticket* help = nullptr)
{
    if (!s.in_flight) return VIT_OK;
    if (pool && help && *help)
        while (hipEventQuery(s.done) == hipErrorNotReady && pool->help(*help)) {}
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

namespace {

struct Job {
    vit_hasher* h;
    const char* type; uint32_t tl;
    const uint8_t* const* data; const uint64_t* len;
    const uint32_t* nb;                    // block count per object (all objects)
    uint8_t* digests;
    const WorkPool::Ticket* help;          // host-lane tasks short enough for a thread that waits on a chunk (or null)
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
    int rc = retire(q, s, job.digests, h->pool.get(), job.help);
    if (rc) return rc;
    if (h->opt.fail_at_chunk && ++q.chunks_this_call == h->opt.fail_at_chunk) {
        q.err = "failure injected at chunk " + std::to_string(q.chunks_this_call) + " (option debug_fail_at_chunk)";
        return VIT_ERR_HIP;
    }
    if (h->opt.trace) {
        for (hipEvent_t& m : s.mark) if (!m) HIP_TRYQ(q, hipEventCreate(&m));
        s.host_ms[0] = begin_ms;
        s.host_ms[1] = double(ns_since(q.trace_t0)) * 1e-6;
    }
    const bool trace = s.traced = h->opt.trace && s.mark[3];
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
    const uint64_t pieces = std::min<uint64_t>({cnt, bytes / (2u << 20), uint64_t(h->opt.pack_threads) * 4});
    if (pieces <= 1) pack_range(0, cnt);
    else {
        const std::vector<uint64_t> cut = vit::balanced_cuts(cnt, int(pieces), prefix.data());
        h->pool->run(vit::WorkPool::HIGH, size_t(pieces), [&](size_t p) { if (cut[p] < cut[p + 1]) pack_range(cut[p], cut[p + 1]); },
                     vit::preferred_node(h, q));
    }
    const bool binned = vit::bin_order(s.h_nb, cnt, s.h_order);
    *pack_ns += ns_since(t0);

    if ((rc = launch_chunk(q, h, s, stream, s.h_arena, bytes, meta_bytes, binned, cnt, trace))) return rc;
    h->count.objects_gpu += cnt; h->count.blocks_gpu += blocks; h->count.bytes_h2d += bytes + cnt * 16; ++h->count.launches;
    return VIT_OK;
}

int run_queue_body(Queue& q, const Job& job, const std::vector<uint64_t>& mine, uint64_t* pack_ns)
{
    vit_hasher* h = job.h;
    HIP_TRYQ(q, hipSetDevice(q.device));
    q.trace_t0 = clk::now();
    q.trace_log.clear();
    q.chunks_this_call = 0;
    q.grew_this_call = false;
    if (h->opt.trace) HIP_TRYQ(q, hipEventRecord(q.ev_a, q.stream[0]));
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
    uint64_t chunk = std::min<uint64_t>(rest_longest * 64 * h->opt.chunk_object_multiple, rest_bytes / 8);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, h->opt.chunk_bytes), 1ull << 30);
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
        if ((rc = retire(q, s, job.digests, h->pool.get(), job.help))) return rc;
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

}  // namespace

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

}  // namespace vit

using namespace vit;

extern "C" {

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
    h->last.host_done_ns = 0;
    h->last.gpu_done_ns = 0;

    // Which objects leave the GPU for host threads: a fixed threshold, or (auto)
    // the balance vit::plan_host_split strikes with the rates measured on this
    // machine (calibrate), see packer.h.
    std::vector<uint32_t> nb(n);
    std::vector<uint64_t> gpu_idx, host_idx;
    uint64_t threshold = 0;
    if (h->opt.host_policy == 2) threshold = h->opt.host_threshold;
    else if (h->opt.host_policy == 1) {
        bool any = false;
        for (uint64_t i = 0; i < n && !any; ++i) any = len[i] >= h->opt.host_min_bytes;
        if (any) {
            uint64_t longest = 0;
            if (!h->rates.calibrated) for (uint64_t i = 1; i < n; ++i) if (len[i] > len[longest]) longest = i;   // calibration's sample
            if (int rc = calibrate(h, data[longest], len[longest], data, len, n)) return rc;
            const vit::ParallelFor on_pool = [h](size_t parts, const std::function<void(size_t)>& fn) {
                h->pool->run(vit::WorkPool::HIGH, parts, fn);
            };
            vit::plan_host_split(len, n, vit::SplitRates{h->rates.lane, h->rates.core, h->rates.copy, h->rates.pack, h->rates.core_multi, h->opt.host_min_bytes, h->rates.host_bias, h->rates.copy_total},
                                 int(h->queues.size()), h->opt.pack_threads, host_idx, &on_pool);
            if (!host_idx.empty()) threshold = len[host_idx.back()] - 1;
        }
    }
    h->last.host_threshold = threshold;

    uint64_t host_bytes = 0;
    if (h->opt.host_policy == 2 && threshold)
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
        const uint64_t slices = n >= 65536 ? std::min<uint64_t>(uint64_t(h->opt.pack_threads) * 2, n / 16384) : 1;
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
    h->last.host_objects = host_idx.size();

    // Oversized objects: one SHA-1 stream each on the pool's threads, at low
    // priority, beside the GPU batch; longest first, so the tail is short.
    std::vector<uint32_t> host_group;                     // group g = host_idx[host_group[g] .. host_group[g+1]): streams run side by side
    std::vector<uint32_t> host_task;                      // pool task t = groups [host_task[t], host_task[t+1])
    std::function<void(size_t)> run_host_task;
    vit::WorkPool::Ticket host_work, host_work_short;
    // From the submit below on, the pool's workers hold references into this
    // frame: every path out waits for them first.
    struct WaitHost {
        vit::WorkPool& pool; vit::WorkPool::Ticket& a; vit::WorkPool::Ticket& b;
        ~WaitHost() { if (a) pool.wait(a); if (b) pool.wait(b); }
    } wait_host{*h->pool, host_work, host_work_short};
    // A core gets through more bytes with several objects' streams interleaved
    // (sha1_host.h), where the machine's SHA instructions are pipelined: when the
    // objects outnumber the threads, the objects after the longest few go in
    // groups of up to kHostStreams neighbours in length order.  The longest stay
    // alone: a group lasts as long as its objects' lengths added up, and an
    // object of more than 1/8 of a thread's share would make its group the tail.
    if (!host_idx.empty()) {
        if (h->opt.host_policy != 1)                          // the balance hands its list over longest first
            std::stable_sort(host_idx.begin(), host_idx.end(), [&](uint64_t a, uint64_t b) { return len[a] > len[b]; });
        const size_t nh = host_idx.size(), nt = size_t(h->opt.pack_threads);
        if (nh >= 2 * nt) calibrate_host(h);
        size_t streams = 1;
        if (h->rates.core_multi > 1.15 * h->rates.core && h->rates.core > 0) streams = std::min<size_t>(std::max<size_t>(nh / nt, 1), vit::kHostStreams);
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
        // Two groups of tasks, both for the pool's workers, the second also for threads that wait on a device
        // chunk (retire): the tasks of objects long enough to keep such a thread from its queue for milliseconds
        // (host_idx is longest first, so they are the leading ones) stay with the workers alone.
        constexpr uint64_t kHelpMaxBytes = 4u << 20;
        size_t long_tasks = 0;
        while (long_tasks + 1 < host_task.size() && len[host_idx[host_group[host_task[long_tasks]]]] > kHelpMaxBytes) ++long_tasks;
        run_host_task = [&, h, t0](size_t t) {
            struct Stamp {                                // the latest finish over all tasks
                vit_hasher* h; clk::time_point t0;
                ~Stamp() { const uint64_t now = ns_since(t0); uint64_t cur = h->last.host_done_ns.load();
                           while (now > cur && !h->last.host_done_ns.compare_exchange_weak(cur, now)) {} }
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
        };
        const size_t all_tasks = host_task.size() - 1;
        host_work = h->pool->submit(vit::WorkPool::LOW, long_tasks, [&](size_t t) { run_host_task(t); });
        host_work_short = h->pool->submit(vit::WorkPool::LOW, all_tasks - long_tasks, [&, long_tasks](size_t t) { run_host_task(long_tasks + t); });
        h->count.objects_host += host_idx.size();
        h->count.bytes_host += host_bytes;
    }

    Job job{h, type, tl, data, len, nb.data(), digests, &host_work_short};
    std::vector<std::vector<uint64_t>> parts;
    shard(gpu_idx, nb.data(), h->queues.size(), parts);
    std::vector<int> rcs(h->queues.size(), VIT_OK);
    std::vector<uint64_t> pack_ns(h->queues.size(), 0);
    h->last.plan_ns = ns_since(t0);
    if (h->queues.size() == 1) {
        rcs[0] = run_queue(h->queues[0], job, parts[0], &pack_ns[0]);
    } else {
        ThreadGroup workers;
        for (size_t k = 0; k < h->queues.size(); ++k)
            workers.t.emplace_back([&, k] { rcs[k] = run_queue(h->queues[k], job, parts[k], &pack_ns[k]); });
        workers.join();
    }
    h->last.gpu_done_ns = ns_since(t0);
    h->pool->wait(host_work);
    h->pool->wait(host_work_short);
    host_work.reset();
    host_work_short.reset();
    h->last.pack_ns = 0;
    for (uint64_t p : pack_ns) h->last.pack_ns = std::max(h->last.pack_ns, p);
    vit::learn_from_call(h, len, gpu_idx, host_idx.size());
    h->last.total_ns = ns_since(t0);
    if (h->opt.trace)
        for (size_t k = 0; k < h->queues.size(); ++k)
            std::fprintf(stderr, "[vit trace] queue %zu, %.2f ms in all, %zu objects on host threads\n%s", k,
                         double(h->last.total_ns) * 1e-6, host_idx.size(), h->queues[k].trace_log.c_str());
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
        const uint64_t budget = std::max<uint64_t>(h->opt.chunk_bytes, 1ull << 20);
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
        h->count.bytes_h2d += span + cnt * 16; ++h->count.launches;
    }
    for (Staging& s : q->st)
        if (rc == VIT_OK) rc = retire(*q, s, digests);
    if (rc != VIT_OK) { drain(*q); return fail(h, rc, "device queue %d: %s", dev, q->err.c_str()); }
    h->count.objects_gpu += n; h->count.blocks_gpu += blocks;
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
    ++h->count.launches; h->count.objects_gpu += n;
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
    const bool sched = h->opt.kernel == VIT_KERNEL_LDS && h->opt.sched_waves && (n + 63) / 64 >= h->opt.sched_min_groups;
    // Every launch of this entry point on the GPU, from whatever hasher, queue or stream, works in the
    // one scratch array: each waits for the one before it (an event per GPU, made with the GPU's first
    // queue), and the lock keeps wait, launch and record of two callers from interleaving.
    std::lock_guard<std::mutex> in_turn(q->shared->order);
    HIP_TRY(h, hipStreamWaitEvent(s, q->shared->done, 0));
    HIP_TRY(h, vit::launch_sha1_static(vit::KernelVariant(h->opt.kernel),
                                       make_batch(*q, d_arena, arena_bytes, d_offsets, d_nblocks, nullptr, n, d_digests),
                                       q->shared->scratch, sched ? h->opt.sched_waves : 0, h->opt.sched_share_below,
                                       h->opt.sched_tail_percent, s));
    HIP_TRY(h, hipEventRecord(q->shared->done, s));
    ++h->count.launches; h->count.objects_gpu += n;
    return VIT_OK;
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

}  // extern "C"
