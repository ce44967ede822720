// host_util.h — small host-side helpers shared by the packer and the hasher.
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace vit {

// Cut [0, n) into `parts` contiguous ranges of balanced weight (weight_prefix:
// n + 1 non-decreasing entries, or nullptr for equal weights).  Returns
// parts + 1 boundaries; a range may be empty.
inline std::vector<uint64_t> balanced_cuts(uint64_t n, int parts, const uint64_t* weight_prefix)
{
    if (parts < 1) parts = 1;
    std::vector<uint64_t> cut(size_t(parts) + 1, 0);
    cut[size_t(parts)] = n;
    const uint64_t total = weight_prefix ? weight_prefix[n] - weight_prefix[0] : n;
    for (int t = 1; t < parts; ++t) {
        if (weight_prefix) {
            const uint64_t target = weight_prefix[0] + total * uint64_t(t) / uint64_t(parts);
            cut[size_t(t)] = uint64_t(std::lower_bound(weight_prefix, weight_prefix + n + 1, target) - weight_prefix);
            if (cut[size_t(t)] > n) cut[size_t(t)] = n;
        } else {
            cut[size_t(t)] = n * uint64_t(t) / uint64_t(parts);
        }
        if (cut[size_t(t)] < cut[size_t(t) - 1]) cut[size_t(t)] = cut[size_t(t) - 1];
    }
    return cut;
}

// Run fn(t, begin, end) over [0, n) split into `threads` contiguous ranges whose
// weights are balanced, on threads started for the call.  With one thread, or
// little work, runs inline.
template <typename Fn>
void parallel_ranges(uint64_t n, int threads, const uint64_t* weight_prefix, Fn fn)
{
    if (n == 0) return;
    if (threads <= 1 || n < 2) { fn(0, uint64_t(0), n); return; }
    if (uint64_t(threads) > n) threads = int(n);
    const std::vector<uint64_t> cut = balanced_cuts(n, threads, weight_prefix);
    std::vector<std::thread> pool;
    pool.reserve(size_t(threads));
    for (int t = 0; t < threads; ++t) {
        if (cut[size_t(t)] == cut[size_t(t) + 1]) continue;
        pool.emplace_back([=, &fn, &cut] { fn(t, cut[size_t(t)], cut[size_t(t) + 1]); });
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
