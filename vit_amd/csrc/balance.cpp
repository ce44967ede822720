// balance.cpp — the rates a batch is balanced with.
//
// One SHA-1 stream is serial: a GPU lane runs it at some 50 MB/s, a host core with
// SHA extensions at gigabytes per second, and a whole batch moves through pack +
// PCIe at tens of gigabytes per second per GPU.  vit::plan_host_split (packer.cpp)
// decides from such rates which objects of a batch leave the GPU for host threads;
// this file measures them on the machine, once per hasher, and corrects them by
// what the calls that follow achieve.
#include "hasher_internal.h"

#include <algorithm>
#include <cmath>

namespace vit {

// The two host hashing rates: one core on one SHA-1 stream, and on kHostStreams
// streams interleaved.  About 5 ms, once per hasher; needed by the balance
// (calibrate) and by the grouping of host-lane objects under any policy.
void calibrate_host(vit_hasher* h)
{
    if (h->rates.host_calibrated) return;
    constexpr uint64_t kHost = 4ull << 20;
    std::vector<uint8_t> src(kHost, 0x5A);
    if (!h->rates.core_fixed) {
        double best = 1e30;
        uint8_t d[20];
        for (int r = 0; r < 2; ++r) {
            const auto t = clk::now();
            vit::host_object_id("blob", src.data(), kHost, d);
            best = std::min(best, double(ns_since(t)) * 1e-9);
        }
        h->rates.core = double(kHost) / best;
    }
    if (!h->rates.core_multi_fixed) {
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
        h->rates.core_multi = double(kHost) / best;
    }
    h->rates.host_calibrated = true;
}

// Measure, once per hasher and on its first queue (copy_total: on all of them), the
// rates the auto policy balances with (anything set through an option stays as set):
//   core_rate  one host thread's SHA-1 (vit::host_object_id over 4 MiB)
//   pack_rate  one host thread packing into a pinned staging arena: up to 8 MiB of
//              the batch's own longest object when that is 1 MiB or more (data
//              as cold in the caches as the batch will be), else a 4 MiB buffer
//              of the library's (cache-warm, optimistic); real batches of 256 MiB
//              and more then correct the rate (hash_batch)
//   copy_rate  one pinned host-to-device copy (32 MiB)
//   copy_total every queue copying 64 MiB at once: the sum they get (one queue: copy_rate)
//   lane_rate  one GPU lane: a lone wave running 1024 compressions back to back,
//              times 0.8: with four other waves resident, the SIMD's oldest wave
//              (a long object's wave always is) gets about that share of issue
//              (1.15 us per block alone, 1.4 to 1.5 us in a full launch).
int calibrate(vit_hasher* h, const uint8_t* sample, uint64_t sample_len, const uint8_t* const* data,
              const uint64_t* len_all, uint64_t n_all)
{
    if (h->rates.calibrated) return VIT_OK;
    const auto t0 = clk::now();
    Queue& q = h->queues[0];
    HIP_TRY(h, hipSetDevice(q.device));
    constexpr uint64_t kHost = 4ull << 20, kCopy = 32ull << 20;
    std::vector<uint8_t> src(kHost, 0x5A);
    calibrate_host(h);
    if (!h->rates.pack_fixed || !h->rates.copy_fixed) {
        Staging& s = q.st[0];
        if (ensure_arena(q, s, kCopy) != VIT_OK) return fail(h, VIT_ERR_NOMEM, "calibration: %s", q.err.c_str());
        if (!h->rates.pack_fixed) {
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
            h->rates.pack = double(len) / best;
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
                const size_t pieces = std::min<size_t>(pick.size(), size_t(h->opt.pack_threads) * 4);
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
            if (h->opt.pack_threads > 1 && pooled_pack(&sec)) {
                uint64_t payload = pooled_pack(&sec);
                if (!payload) { next = 0; payload = pooled_pack(&sec); }      // a batch of under 128 MiB: the same stretch again
                if (payload) h->rates.pack = double(payload) / sec / double(h->opt.pack_threads);
            }
        }
        if (!h->rates.copy_fixed) {
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
            h->rates.copy = double(kCopy) / (double(best) * 1e-3);
        }
    }
    // What the queues get TOGETHER.  One copy alone says what a PCIe link carries; N queues may share
    // a link (an ordinal that repeats), a socket's root complex, or simply the host's DRAM, and then the
    // GPU side of a batch drains at the shared rate, not at N times the single one.  One pinned copy on
    // every queue at the same moment, timed on the host from the first submission to the last completion.
    if (!h->rates.copy_total_fixed) {
        if (h->queues.size() == 1) h->rates.copy_total = h->rates.copy;
        else {
            constexpr uint64_t kEach = 64ull << 20;
            for (Queue& k : h->queues) {
                HIP_TRY(h, hipSetDevice(k.device));
                if (ensure_arena(k, k.st[0], kEach) != VIT_OK) return fail(h, VIT_ERR_NOMEM, "calibration: %s", k.err.c_str());
            }
            double best = 1e30;
            for (int r = 0; r < 3; ++r) {                 // the first round also wakes every copy engine
                const auto t = clk::now();
                for (Queue& k : h->queues) {
                    HIP_TRY(h, hipSetDevice(k.device));
                    HIP_TRY(h, hipMemcpyAsync(k.st[0].d_arena, k.st[0].h_arena, kEach, hipMemcpyHostToDevice, k.stream[0]));
                }
                for (Queue& k : h->queues) {
                    HIP_TRY(h, hipSetDevice(k.device));
                    HIP_TRY(h, hipStreamSynchronize(k.stream[0]));
                }
                if (r) best = std::min(best, double(ns_since(t)) * 1e-9);
            }
            h->rates.copy_total = double(kEach) * double(h->queues.size()) / best;
            HIP_TRY(h, hipSetDevice(q.device));
        }
    }
    if (!h->rates.lane_fixed) {
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
        h->rates.lane = 0.8 * double(kBlocks) * 64.0 / (double(best) * 1e-3);
    }
    h->rates.calibrated = true;
    h->rates.calibrate_ns = ns_since(t0);
    return VIT_OK;
}

// After a call: what it measured corrects the rates the next one balances with.
void learn_from_call(vit_hasher* h, const uint64_t* len, const std::vector<uint64_t>& gpu_idx, size_t host_objects)
{
    // The calibration packs one cache-warm buffer; a real batch streams from
    // DRAM on every thread at once.  Let what this call measured correct the
    // rate the next call balances with: bytes packed over the packing time of
    // the queue that packed longest, per thread of the budget (batches of 256 MiB and more only).
    // (Not from a call that packed into freshly pinned memory: its first touch runs at a
    // tenth of the rate and says nothing about the calls to come.)
    bool fresh_arena = false;
    for (const Queue& q : h->queues) fresh_arena = fresh_arena || q.grew_this_call;
    if (!h->rates.pack_fixed && h->rates.calibrated && h->last.pack_ns && !fresh_arena) {
        uint64_t packed = 0;
        for (uint64_t i : gpu_idx) packed += len[i];
        if (packed >= (256ull << 20)) {
            const double per_thread = double(packed) / (double(h->last.pack_ns) * 1e-9) / double(h->opt.pack_threads);
            h->rates.pack = 0.5 * h->rates.pack + 0.5 * per_thread;
        }
    }
    // The rates are measured one at a time on an idle machine; a batch runs them
    // all at once against one memory system, and the more GPUs a hasher drives the
    // less of the sum the host's DRAM delivers.  What the model cannot know it
    // learns: the host side's estimate is scaled by how much later (or earlier)
    // than the device queues the host threads finished, damped, and only from
    // calls in which both sides had real work spread over many objects.
    if (h->opt.host_policy == 1 && !fresh_arena && host_objects >= 4 * size_t(h->opt.pack_threads) &&
        gpu_idx.size() >= 1024 && h->last.gpu_done_ns > 1000000 && h->last.host_done_ns > 1000000) {
        const double ratio = std::min(2.0, std::max(0.5, double(h->last.host_done_ns.load()) / double(h->last.gpu_done_ns)));
        h->rates.host_bias = std::min(5.0, std::max(0.2, h->rates.host_bias * std::pow(ratio, 0.7)));
    }
}

}  // namespace vit
