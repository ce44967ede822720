// The hasher's shared worker pool (vit_amd/csrc/work_pool.h) under ThreadSanitizer (built and run by
// tests/test_sanitize_host.py): several "device queue" threads submit packing groups at high priority
// while one long low-priority group (the host lane) runs; every task runs exactly once, high-priority
// groups overtake the low-priority one, and the threads busy at once never exceed the pool's size.
#include "work_pool.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

int main()
{
    const int kThreads = 6, kQueues = 4;
    vit::WorkPool pool(kThreads);
    std::atomic<int> live{0}, peak{0};
    auto enter = [&] { int v = ++live, p = peak.load(); while (v > p && !peak.compare_exchange_weak(p, v)) {} };
    auto leave = [&] { --live; };

    // low priority: 400 slow tasks
    std::vector<std::atomic<int>> low_hits(400);
    std::atomic<int> low_done{0};
    auto low = pool.submit(vit::WorkPool::LOW, low_hits.size(), [&](size_t i) {
        enter(); std::this_thread::sleep_for(std::chrono::microseconds(300)); ++low_hits[i]; ++low_done; leave();
    });
    // high priority from four submitting threads, each 30 groups of 16 tasks
    std::atomic<long> high_sum{0};
    std::atomic<int> low_done_when_high_finished{0};
    std::vector<std::thread> queues;
    for (int q = 0; q < kQueues; ++q)
        queues.emplace_back([&, q] {
            for (int g = 0; g < 30; ++g) {
                std::vector<int> hit(16, 0);
                pool.run(vit::WorkPool::HIGH, hit.size(), [&](size_t i) { enter(); ++hit[i]; high_sum += long(i) + q; leave(); });
                for (int v : hit) if (v != 1) { std::printf("high task ran %d times\n", v); std::exit(1); }
            }
        });
    for (auto& t : queues) t.join();
    low_done_when_high_finished = low_done.load();
    pool.wait(low);
    for (auto& v : low_hits) if (v != 1) { std::printf("low task ran %d times\n", v.load()); return 1; }
    long want = 0;
    for (int q = 0; q < kQueues; ++q) want += 30L * (120 + 16 * q);
    if (high_sum != want) { std::printf("high sum %ld != %ld\n", high_sum.load(), want); return 1; }
    // the helpers (4 submitting threads + main) run tasks too: at most pool threads + helpers at once
    if (peak > kThreads + kQueues + 1) { std::printf("peak %d threads in tasks\n", peak.load()); return 1; }
    if (pool.peak_busy() > kThreads) { std::printf("pool peak %d > %d\n", pool.peak_busy(), kThreads); return 1; }
    // 1920 quick high-priority tasks finished while the 400 x 300 us of low-priority work (20 ms on 6 threads) was
    // still going on: they were not queued behind it
    if (low_done_when_high_finished >= 400) { std::printf("high-priority groups waited for the low-priority one\n"); return 1; }
    // resize, empty groups, a pool that must be helped
    pool.resize(1);
    pool.run(vit::WorkPool::LOW, 0, [](size_t) {});
    std::atomic<int> n{0};
    pool.run(vit::WorkPool::HIGH, 1000, [&](size_t) { ++n; });
    if (n != 1000) return 1;
    std::printf("pool clean: peak %d in tasks, pool peak %d of %d, low done at high finish %d\n", peak.load(), pool.peak_busy(), kThreads,
                low_done_when_high_finished.load());
    return 0;
}
