// hasher.cpp — lifetime, options and counters of a vit_hasher, the status word,
// pinned arenas: the part of the C ABI of include/vit_hash.h that every other part
// builds on (see hasher_internal.h for which file holds what).
//
// Host threads: ONE pool per hasher (work_pool.h) serves the packing of every
// queue's chunks (high priority) and the host-lane objects (low priority); a
// queue's own thread only submits work and helps with its own packing.  Where
// the GPUs hang off more than one NUMA node the pool's workers are pinned to the
// nodes in proportion to the queues there (numa.h) and a queue's packing goes
// to its node's workers first.
#include "hasher_internal.h"
#include "numa.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {
thread_local std::string g_create_error;
}

namespace vit {

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

// Worker threads of the shared pool: the thread budget less the threads that
// help from outside it (the caller with one queue, the queue threads with more).
int pool_threads_for(const vit_hasher* h)
{
    const int helpers = h->queues.size() > 1 ? int(h->queues.size()) : 1;
    return std::max(1, h->opt.pack_threads - helpers);
}

// NUMA node a page of memory lies on (-1: unknown): move_pages with no target nodes only reports.
int node_of_address(const void* p)
{
#ifdef SYS_move_pages
    if (!p) return -1;
    void* page = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(4095));
    int status = -1;
    if (::syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0) return -1;
    return status < 0 ? -1 : status;
#else
    (void)p;
    return -1;
#endif
}

static std::string sysfs_root()
{
    const char* e = std::getenv("VIT_SYSFS_ROOT");        // tests point this at a made-up tree
    return e && *e ? e : "/sys";
}

int device_numa_node(int device)
{
    char id[32] = {0};
    if (hipDeviceGetPCIBusId(id, int(sizeof id), device) != hipSuccess) return -1;
    return numa::pci_device_node(sysfs_root(), id);
}

// Start the pool's workers for the current budget; pin them when the option asks for it and the
// plan allows it (numa.h).  A worker pinned to node n may run on the CPUs of n that this process
// may run on at all.
void restart_pool(vit_hasher* h)
{
    const int workers = pool_threads_for(h);
    std::vector<std::vector<int>> usable;
    std::vector<int> plan(size_t(workers), -1);
    if (h->opt.numa_pin) {
        std::vector<int> qn;
        int top = -1;
        for (const Queue& q : h->queues) { qn.push_back(q.numa_node); top = std::max(top, q.numa_node); }
        cpu_set_t mine;
        CPU_ZERO(&mine);
        const bool have_mask = sched_getaffinity(0, sizeof mine, &mine) == 0;
        usable.resize(size_t(top + 1));
        const std::string root = sysfs_root();
        for (int n = 0; n <= top; ++n)
            for (int c : numa::node_cpus(root, n))
                if (c < CPU_SETSIZE && (!have_mask || CPU_ISSET(c, &mine))) usable[size_t(n)].push_back(c);
        plan = numa::plan_worker_nodes(qn, workers, usable);
    }
    h->worker_node = plan;
    auto place = [usable](int, int node) {
        if (node < 0 || size_t(node) >= usable.size()) return;
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : usable[size_t(node)]) CPU_SET(c, &set);
        (void)sched_setaffinity(0, sizeof set, &set);      // a refusal leaves the thread where the scheduler puts it
    };
    h->pool.reset();
    h->pool.reset(new WorkPool(workers, place, plan));
}

int preferred_node(const vit_hasher* h, const Queue& q)
{
    for (int n : h->worker_node) if (n == q.numa_node) return n;
    return -1;                                             // nobody is pinned there: any worker
}

// Read the queue's device status word, and the one in the device's static
// scratch, through the queue's first stream; clear what was set.
int read_status_word(vit_hasher* h, Queue& q, uint32_t* out)
{
    HIP_TRY(h, hipMemcpyAsync(&q.h_status[0], q.d_status, 4, hipMemcpyDeviceToHost, q.stream[0]));
    HIP_TRY(h, hipMemcpyAsync(&q.h_status[1], q.shared->scratch, 4, hipMemcpyDeviceToHost, q.stream[0]));
    HIP_TRY(h, hipStreamSynchronize(q.stream[0]));
    *out = q.h_status[0] | q.h_status[1];
    if (*out) {
        if (q.h_status[0]) HIP_TRY(h, hipMemsetAsync(q.d_status, 0, 4, q.stream[0]));
        if (q.h_status[1]) HIP_TRY(h, hipMemsetAsync(q.shared->scratch, 0, 4, q.stream[0]));
        HIP_TRY(h, hipStreamSynchronize(q.stream[0]));
    }
    return VIT_OK;
}

}  // namespace vit

using namespace vit;

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
        h->opt.pack_threads = int(std::max(1u, budget));
        if (const char* e = std::getenv("VIT_NUMA_PIN")) h->opt.numa_pin = std::atoi(e) != 0;
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
        if (r == hipSuccess) q.numa_node = vit::device_numa_node(ids[k]);
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
        // The file-scope scratch array has one copy per GPU: this device's address is taken
        // (and its status word cleared) the first time the process makes a queue on the device,
        // with the device current; every later queue on it, of whatever hasher, shares the entry.
        if (r == hipSuccess) q.shared = vit::static_device(ids[k], &r);
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
    try { vit::restart_pool(h); }
    catch (const std::exception&) {
        vit_hasher_destroy(h);
        return fail(nullptr, VIT_ERR_NOMEM, "out of host memory");
    }
    *out = h;
    return VIT_OK;
}

void vit_hasher_destroy(vit_hasher* h)
{
    if (!h) return;
    h->pool.reset();                                       // no worker outlives the queues
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
        if (q.z_scratch) (void)hipFree(q.z_scratch);
        if (q.z_tables) (void)hipFree(q.z_tables);
        if (q.z_done) (void)hipEventDestroy(q.z_done);
        for (Queue::StreamSet& z : q.z) {
            if (z.d_out) (void)hipFree(z.d_out);
            if (z.h_out) (void)hipHostFree(z.h_out);
            if (z.meta_d) (void)hipFree(z.meta_d);
            if (z.meta_h) (void)hipHostFree(z.meta_h);
        }
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
    auto rate = [&](double& r, bool& fixed) { r = double(value); fixed = value != 0; if (!value) h->rates.calibrated = h->rates.host_calibrated = false; };
    if (k == "kernel") {
        if (value > 5) return fail(h, VIT_ERR_ARG, "unknown kernel %llu", (unsigned long long)value);
        h->opt.kernel = int(value);
    } else if (k == "host_threshold_bytes") { h->opt.host_threshold = value; h->opt.host_policy = value ? 2 : 0; }
    else if (k == "host_min_bytes") h->opt.host_min_bytes = value;
    else if (k == "host_auto") { h->opt.host_policy = value ? 1 : (h->opt.host_threshold ? 2 : 0); }
    else if (k == "chunk_bytes") h->opt.chunk_bytes = value < (1u << 20) ? (1u << 20) : value;
    else if (k == "chunk_object_multiple") h->opt.chunk_object_multiple = value;
    else if (k == "trace") h->opt.trace = value != 0;
    else if (k == "wave_trace_ptr") { for (Queue& q : h->queues) q.wave_trace = reinterpret_cast<uint64_t*>(uintptr_t(value)); }
    else if (k == "pack_threads") {
        h->opt.pack_threads = value < 1 ? 1 : int(value > 256 ? 256 : value);
        vit::restart_pool(h);
    }
    else if (k == "numa_pin") { h->opt.numa_pin = value != 0; vit::restart_pool(h); }
    else if (k == "lane_rate") rate(h->rates.lane, h->rates.lane_fixed);
    else if (k == "core_rate") rate(h->rates.core, h->rates.core_fixed);
    else if (k == "core_rate_multi") rate(h->rates.core_multi, h->rates.core_multi_fixed);
    else if (k == "copy_rate") rate(h->rates.copy, h->rates.copy_fixed);
    else if (k == "pack_rate") rate(h->rates.pack, h->rates.pack_fixed);
    else if (k == "copy_rate_total") rate(h->rates.copy_total, h->rates.copy_total_fixed);
    else if (k == "host_bias_percent") h->rates.host_bias = value ? double(value) / 100.0 : 1.0;
    else if (k == "debug_fail_at_chunk") h->opt.fail_at_chunk = value;
    else if (k == "debug_fail_deflate") h->opt.fail_deflate = value != 0;
    else if (k == "debug_corrupt_deflate") h->opt.corrupt_deflate = value != 0;
    else if (k == "deflate_max_bytes") h->opt.deflate_max_bytes = std::min<uint64_t>(value, vit::deflate_max_input());
    else if (k == "deflate_chunk_bytes") h->opt.deflate_chunk_bytes = std::min<uint64_t>(std::max<uint64_t>(value, 1u << 20), 1ull << 30);
    else if (k == "deflate_lanes") h->opt.deflate_lanes = uint32_t(std::min<uint64_t>(std::max<uint64_t>(value, 64), 1u << 20) / 64 * 64);
    else if (k == "sched_waves") h->opt.sched_waves = uint32_t(value > 8 ? 8 : value);
    else if (k == "sched_share_below") h->opt.sched_share_below = uint32_t(value > 0xFFFFFFFEull ? 0xFFFFFFFEu : value);
    else if (k == "sched_min_groups") h->opt.sched_min_groups = uint32_t(value > 0xFFFFFFFFull ? 0xFFFFFFFFu : value);
    else if (k == "sched_tail_percent") h->opt.sched_tail_percent = uint32_t(value > 1000 ? 1000 : value);
    else return fail(h, VIT_ERR_ARG, "unknown option '%s'", name);
    return VIT_OK;
    VIT_NOEXCEPT_END(h)
}

int vit_hasher_get_stat(const vit_hasher* h, const char* name, uint64_t* value)
{
    if (!h || !name || !value) return VIT_ERR_ARG;
    const std::string k(name);
    if (k == "objects_gpu") *value = h->count.objects_gpu;
    else if (k == "objects_host") *value = h->count.objects_host;
    else if (k == "bytes_host") *value = h->count.bytes_host;
    else if (k == "last_host_threshold") *value = h->last.host_threshold;
    else if (k == "last_host_objects") *value = h->last.host_objects;
    else if (k == "blocks_gpu") *value = h->count.blocks_gpu;
    else if (k == "bytes_h2d") *value = h->count.bytes_h2d;
    else if (k == "launches") *value = h->count.launches;
    else if (k == "last_kernel_ns") *value = h->last.kernel_ns;
    else if (k == "last_pack_ns") *value = h->last.pack_ns;
    else if (k == "last_total_ns") *value = h->last.total_ns;
    else if (k == "last_gpu_done_ns") *value = h->last.gpu_done_ns;
    else if (k == "last_plan_ns") *value = h->last.plan_ns;
    else if (k == "host_bias_percent") *value = uint64_t(h->rates.host_bias * 100.0 + 0.5);
    else if (k == "last_host_done_ns") *value = h->last.host_done_ns;
    else if (k == "host_sha_accelerated") *value = vit::HostSha1::accelerated();
    else if (k == "lane_rate") *value = uint64_t(h->rates.lane);
    else if (k == "core_rate") *value = uint64_t(h->rates.core);
    else if (k == "core_rate_multi") *value = uint64_t(h->rates.core_multi);
    else if (k == "copy_rate") *value = uint64_t(h->rates.copy);
    else if (k == "pack_rate") *value = uint64_t(h->rates.pack);
    else if (k == "copy_rate_total") *value = uint64_t(h->rates.copy_total);
    else if (k == "workers_pinned") { *value = 0; for (int n : h->worker_node) *value += n >= 0; }
    else if (k.compare(0, 16, "queue_numa_node:") == 0 || k.compare(0, 16, "arena_numa_node:") == 0) {
        char* end = nullptr;
        const unsigned long q = std::strtoul(name + 16, &end, 10);
        if (end == name + 16 || *end || q >= h->queues.size()) return fail(h, VIT_ERR_ARG, "no device queue in '%s'", name);
        const int node = k[0] == 'q' ? h->queues[q].numa_node : vit::node_of_address(h->queues[q].st[0].h_arena);
        *value = node < 0 ? ~0ull : uint64_t(node);
    }
    else if (k == "calibrate_ns") *value = h->rates.calibrate_ns;
    else if (k == "pool_threads") *value = uint64_t(h->pool->threads());
    else if (k == "pool_peak_busy") *value = uint64_t(h->pool->peak_busy());
    else return fail(h, VIT_ERR_ARG, "unknown stat '%s'", name);
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

}  // extern "C"
