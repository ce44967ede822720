// bench_support.cpp — what include/vit_hash_bench.h declares and needs a device for: launch
// arithmetic, device memory for callers without a HIP binding, the synthetic-arena generator, the
// timed launch loop and the compute-only ceiling.  (Its host-only entries live with the code they
// expose: vit_host_split_plan and vit_synth_bytes in packer.cpp, vit_numa_* in numa.cpp.)
#include "hasher_internal.h"

#include <algorithm>
#include <cstring>

using namespace vit;

extern "C" {

int vit_launch_shape(int kernel, uint64_t n, uint32_t* blocks, uint32_t* threads)
{
    if (!blocks || !threads || kernel < 0 || kernel > 5 || n > 0xFFFFFFFFull) return VIT_ERR_ARG;
    vit::launch_shape(vit::KernelVariant(kernel), n, blocks, threads);
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
        h->last.kernel_ns = uint64_t(double(ms_each[i]) * 1e6);
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
