// deflate_host.cpp — host side of the deflate kernel: its device memory, the launch on a resident
// arena (vit_deflate_arena_device) and the chunked path for objects in host memory
// (vit_deflate_objects).  The kernel itself is deflate_kernels.hip / zdeflate.h.
#include "hasher_internal.h"

#include <algorithm>
#include <cstring>

using namespace vit;

// Device memory of the deflate kernel, made on first use: the constant tables and one
// workspace slot per lane in flight.
namespace {

int ensure_deflate(vit_hasher* h, Queue& q)
{
    if (!q.z_tables) {
        std::vector<uint8_t> host(vit::deflate_tables_bytes());
        vit::deflate_make_tables(host.data());
        HIP_TRY(h, hipMalloc(&q.z_tables, host.size()));
        HIP_TRY(h, hipMemcpy(q.z_tables, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    if (!q.z_done) HIP_TRY(h, hipEventCreateWithFlags(&q.z_done, hipEventDisableTiming));
    if (!q.z_scratch) {                                    // symbols and tree arrays of the objects the wave kernels have in flight
        uint64_t bytes = 0;
        HIP_TRY(h, vit::deflate_wave_workspace_bytes(&bytes));
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&q.z_scratch), bytes));
    }
    return VIT_OK;
}

// The lane kernel's tables in device memory (181 KiB per object in flight): only objects too long for the wave
// kernels need them, so they are made when such an object may come and not before.
int ensure_lane_workspace(vit_hasher* h, Queue& q)
{
    if (q.z_lanes != h->opt.deflate_lanes) {
        HIP_TRY(h, hipDeviceSynchronize());               // a launch on a caller's stream may still be using the old workspace
        if (q.z_workspace) { HIP_TRY(h, hipFree(q.z_workspace)); q.z_workspace = nullptr; q.z_lanes = 0; }
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&q.z_workspace), vit::deflate_workspace_bytes(h->opt.deflate_lanes)));
        q.z_lanes = h->opt.deflate_lanes;
    }
    return VIT_OK;
}

// Both kernels over a batch, on `stream`, in turn with every other launch that uses the queue's tables.  out_len is
// cleared first: what neither kernel does reads as "the host's".  with_lanes: some object may be too long for a wave.
int launch_both(vit_hasher* h, Queue& q, vit::DeflateBatch b, bool with_lanes, hipStream_t stream)
{
    b.tables_raw = q.z_tables; b.status = q.d_status; b.trace = q.wave_trace;
    b.max_input = uint32_t(std::min<uint64_t>(h->opt.deflate_max_bytes, vit::deflate_max_input()));
    with_lanes = with_lanes && b.max_input > vit::deflate_wave_max_input();
    if (with_lanes) { if (int rc = ensure_lane_workspace(h, q)) return rc; }
    b.workspace = q.z_scratch;
    HIP_TRY(h, hipStreamWaitEvent(stream, q.z_done, 0));  // launches share the queue's scratch and tables: one after the other
    HIP_TRY(h, hipMemsetAsync(b.out_len, 0, uint64_t(b.n) * 4u, stream));
    HIP_TRY(h, vit::launch_deflate_wave(b, stream));
    if (with_lanes) HIP_TRY(h, vit::launch_deflate(b, q.z_workspace, q.z_lanes, vit::deflate_wave_max_input(), stream));
    HIP_TRY(h, hipEventRecord(q.z_done, stream));
    return VIT_OK;
}

}  // namespace

extern "C" {

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
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : q->stream[0];
    return launch_both(h, *q, b, true, s);                // the lengths are the device's to know: any object may be a long one
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
    if (h->opt.fail_deflate) return fail(h, VIT_ERR_HIP, "failure injected (option debug_fail_deflate)");
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
        if (framed > h->opt.deflate_max_bytes || framed > vit::deflate_max_input()) continue;
        nb[i] = uint32_t((framed + 9 + 63) / 64);
        out_off[i] = total_out;
        total_out += (uint64_t(vit::deflate_bound(uint32_t(framed))) + 15) / 16 * 16;
        mine.push_back(i);
    }
    if (total_out > out_cap) return fail(h, VIT_ERR_RANGE, "the streams may need %llu bytes, %llu given", (unsigned long long)total_out, (unsigned long long)out_cap);

    // Chunks of the batch alternate between two sets of buffers and two streams: while the device deflates one chunk,
    // the host packs the next and brings the streams of the one before to the caller.  (The kernels of the two streams
    // still run one after the other: they share the queue's scratch.)  Into a caller's buffer that is not pinned the
    // streams go through a pinned one: the copy from the device runs at the link's speed, and the pool's threads then
    // copy every stream's own bytes, not the whole slots the bound reserves.
    const uint64_t chunk_bytes = std::max<uint64_t>(h->opt.deflate_chunk_bytes, 1u << 20);
    bool direct = false;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, out) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                      // ordinary memory: not an error
    }
    struct Flight { bool live = false; size_t pos = 0; uint64_t cnt = 0, out_lo = 0, out_hi = 0; } flight[2];
    struct Settle {                                        // on every way out: nothing of this call is still running
        Queue& q; ~Settle() { (void)hipStreamSynchronize(q.stream[0]); (void)hipStreamSynchronize(q.stream[1]); }
    } settle{q};
    auto land = [&](int set) -> int {                      // the set's chunk: wait for it, lengths and streams to the caller
        Flight& f = flight[set];
        if (!f.live) return VIT_OK;
        f.live = false;
        Queue::StreamSet& z = q.z[set];
        HIP_TRY(h, hipStreamSynchronize(q.stream[set]));
        const uint32_t* h_zlen = reinterpret_cast<const uint32_t*>(z.meta_h + z.meta_cap * 8);
        for (uint64_t k = 0; k < f.cnt; ++k) out_len[mine[f.pos + k]] = h_zlen[k];
        if (!direct) {
            const uint64_t pieces = std::min<uint64_t>(f.cnt, uint64_t(h->opt.pack_threads) * 4);
            h->pool->run(vit::WorkPool::HIGH, size_t(pieces), [&](size_t p) {
                for (uint64_t k = f.cnt * p / pieces; k < f.cnt * (p + 1) / pieces; ++k) {
                    const uint64_t i = mine[f.pos + k];
                    if (out_len[i]) std::memcpy(out + out_off[i], z.h_out + (out_off[i] - f.out_lo), out_len[i]);
                }
            });
        }
        return VIT_OK;
    };
    int turn = 0;
    for (size_t pos = 0; pos < mine.size(); turn ^= 1) {
        uint64_t blocks = 0;
        size_t end = pos;
        while (end < mine.size() && (end == pos || (blocks + nb[mine[end]]) * 64 <= chunk_bytes)) blocks += nb[mine[end++]];
        const uint64_t cnt = end - pos, bytes = blocks * 64;
        const uint64_t out_lo = out_off[mine[pos]];
        const uint64_t out_hi = end < mine.size() ? out_off[mine[end]] : total_out;
        if (int rc = land(turn)) return rc;                // the set's chunk before this one
        Staging& s = q.st[turn];
        Queue::StreamSet& z = q.z[turn];
        hipStream_t stream = q.stream[turn];
        int rc;
        if ((rc = ensure_arena(q, s, bytes)) || (rc = ensure_meta(q, s, cnt))) return fail(h, rc, "device queue 0: %s", q.err.c_str());
        const uint64_t meta_bytes = carve_meta(s, cnt);
        if (out_hi - out_lo > z.d_out_cap) {
            if (z.d_out) { HIP_TRY(h, hipFree(z.d_out)); z.d_out = nullptr; z.d_out_cap = 0; }
            const uint64_t cap = grown(z.d_out_cap, out_hi - out_lo);
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&z.d_out), cap));
            z.d_out_cap = cap;
        }
        if (!direct && out_hi - out_lo > z.h_out_cap) {
            if (z.h_out) { HIP_TRY(h, hipHostFree(z.h_out)); z.h_out = nullptr; z.h_out_cap = 0; }
            const uint64_t cap = grown(z.h_out_cap, out_hi - out_lo);
            HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&z.h_out), cap, hipHostMallocDefault));
            z.h_out_cap = cap;
        }
        if (cnt > z.meta_cap) {
            if (z.meta_d) { HIP_TRY(h, hipFree(z.meta_d)); z.meta_d = nullptr; }
            if (z.meta_h) { HIP_TRY(h, hipHostFree(z.meta_h)); z.meta_h = nullptr; }
            z.meta_cap = 0;
            const uint64_t cap = round_up(cnt, 65536);
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&z.meta_d), cap * 12));
            HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&z.meta_h), cap * 12, hipHostMallocDefault));
            z.meta_cap = cap;
        }
        uint64_t* h_zoff = reinterpret_cast<uint64_t*>(z.meta_h);
        uint32_t* h_zlen = reinterpret_cast<uint32_t*>(z.meta_h + z.meta_cap * 8);
        uint64_t* d_zoff = reinterpret_cast<uint64_t*>(z.meta_d);
        uint32_t* d_zlen = reinterpret_cast<uint32_t*>(z.meta_d + z.meta_cap * 8);

        std::vector<uint64_t> prefix(cnt + 1, 0);
        for (uint64_t k = 0; k < cnt; ++k) {
            const uint64_t i = mine[pos + k];
            s.h_off[k] = prefix[k] * 64;
            s.h_nb[k] = nb[i];
            prefix[k + 1] = prefix[k] + nb[i];
            h_zoff[k] = out_off[i] - out_lo;
        }
        const uint64_t pieces = std::min<uint64_t>({cnt, std::max<uint64_t>(bytes / (2u << 20), 1), uint64_t(h->opt.pack_threads) * 4});
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
        b.out = z.d_out; b.out_off = d_zoff; b.out_len = d_zlen;
        bool long_ones = false;                            // any object of this chunk beyond what a workgroup takes?  (By its framed
        for (uint64_t k = 0; k < cnt && !long_ones; ++k) { // length: the block count is the same for the 57 lengths around the limit.)
            const uint64_t i = mine[pos + k];
            long_ones = tl + 1 + vit::decimal_digits(len[i]) + 1 + len[i] > vit::deflate_wave_max_input();
        }
        if (int rc = launch_both(h, q, b, long_ones, stream)) return rc;
        HIP_TRY(h, hipMemcpyAsync(direct ? out + out_lo : z.h_out, z.d_out, out_hi - out_lo, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipMemcpyAsync(h_zlen, d_zlen, cnt * 4, hipMemcpyDeviceToHost, stream));
        flight[turn].live = true; flight[turn].pos = pos; flight[turn].cnt = cnt; flight[turn].out_lo = out_lo; flight[turn].out_hi = out_hi;
        pos = end;
    }
    if (int rc = land(turn)) return rc;                    // the older of the two first
    if (int rc = land(turn ^ 1)) return rc;
    uint32_t word = 0;
    if (int rc = read_status_word(h, q, &word)) return rc;
    if (word) return fail(h, VIT_ERR_DEVICE_STATUS, "device queue 0 flagged status %u", word);
    if (h->opt.corrupt_deflate)                           // tests: what a silent fault on the device would hand the caller
        for (uint64_t i : mine)
            if (out_len[i] > 12) { out[out_off[i] + out_len[i] / 2] ^= 0x10; break; }
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
        if (framed > h->opt.deflate_max_bytes || framed > vit::deflate_max_input()) continue;
        total += (uint64_t(vit::deflate_bound(uint32_t(framed))) + 15) / 16 * 16;
    }
    return total;
}

uint32_t vit_deflate_bound(uint64_t framed_len)
{
    return framed_len > vit::deflate_max_input() ? 0u : vit::deflate_bound(uint32_t(framed_len));
}

}  // extern "C"
