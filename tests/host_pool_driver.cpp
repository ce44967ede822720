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
        enter(); std::this_thread::sleep_for(std::chrono::microseconds(1500)); ++low_hits[i]; ++low_done; leave();
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
    // 1920 quick high-priority tasks finished while the 400 x 1500 us of low-priority work (100 ms on 6 threads) was
    // still going on: they were not queued behind it
    if (low_done_when_high_finished >= 400) { std::printf("high-priority groups waited for the low-priority one\n"); return 1; }
    // Placement: workers 0-2 carry node 0, workers 3-5 node 1.  A group that names a node goes to that node's
    // workers first, but is never left waiting for them: every task runs once even when only the other node's
    // workers are free (node 7 has no worker at all), and help() runs a group's tasks on a thread that is not in
    // the pool and says when none are left.
    {
        thread_local int my_node = -2;
        vit::WorkPool placed(6, [](int, int node) { my_node = node; }, std::vector<int>{0, 0, 0, 1, 1, 1});
        std::atomic<int> ran{0}, local{0}, helped{0};
        std::vector<vit::WorkPool::Ticket> tickets;
        for (int g = 0; g < 40; ++g) {
            const int node = g % 4 == 3 ? 7 : g % 2;
            tickets.push_back(placed.submit(vit::WorkPool::HIGH, 25, [&, node](size_t) {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                ++ran;
                if (my_node == node) ++local;
            }, node));
        }
        auto low2 = placed.submit(vit::WorkPool::LOW, 200, [&](size_t) { if (my_node == -2) ++helped; ++ran; });
        while (placed.help(low2)) {}                       // this thread is no worker: my_node stays -2
        if (placed.help(low2)) { std::printf("help() found a task in a group with none left\n"); return 1; }
        for (auto& t : tickets) placed.wait(t);
        placed.wait(low2);
        if (ran != 40 * 25 + 200) { std::printf("placed groups: %d tasks ran\n", ran.load()); return 1; }
        if (helped == 0) { std::printf("help() ran nothing\n"); return 1; }
        // 30 of the 40 groups name node 0 or 1: most of their 750 tasks should have run on their own node's workers
        if (local < 375) { std::printf("only %d of 750 tasks ran on their own node\n", local.load()); return 1; }
        std::printf("placement: %d of 750 node-tagged tasks ran on their node, %d low tasks by the helping thread\n", local.load(), helped.load());
    }
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
