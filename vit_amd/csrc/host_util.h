// host_util.h — small host-side helpers shared by the packer and the hasher.
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace vit {

// Run fn(t, begin, end) over [0, n) split into `threads` contiguous ranges whose
// weights (weight_prefix, n + 1 entries, non-decreasing) are balanced.  With one
// thread, or little work, runs inline.
template <typename Fn>
void parallel_ranges(uint64_t n, int threads, const uint64_t* weight_prefix, Fn fn)
{
    if (n == 0) return;
    if (threads <= 1 || n < 2) { fn(0, uint64_t(0), n); return; }
    if (uint64_t(threads) > n) threads = int(n);
    std::vector<uint64_t> cut(size_t(threads) + 1, 0);
    cut[size_t(threads)] = n;
    const uint64_t total = weight_prefix ? weight_prefix[n] - weight_prefix[0] : n;
    for (int t = 1; t < threads; ++t) {
        if (weight_prefix) {
            const uint64_t target = weight_prefix[0] + total * uint64_t(t) / uint64_t(threads);
            cut[size_t(t)] = uint64_t(std::lower_bound(weight_prefix, weight_prefix + n + 1, target) - weight_prefix);
            if (cut[size_t(t)] > n) cut[size_t(t)] = n;
        } else {
            cut[size_t(t)] = n * uint64_t(t) / uint64_t(threads);
        }
        if (cut[size_t(t)] < cut[size_t(t) - 1]) cut[size_t(t)] = cut[size_t(t) - 1];
    }
    std::vector<std::thread> pool;
    pool.reserve(size_t(threads));
    for (int t = 0; t < threads; ++t) {
        if (cut[size_t(t)] == cut[size_t(t) + 1]) continue;
        pool.emplace_back([=, &fn] { fn(t, cut[size_t(t)], cut[size_t(t) + 1]); });
    }
    for (auto& th : pool) th.join();
}

inline uint32_t decimal_digits(uint64_t v)
{
    uint32_t d = 1;
    while (v >= 10) { v /= 10; ++d; }
    return d;
}

}  // namespace vit
