// work_pool.h — a fixed set of worker threads serving groups of indexed tasks
// at two priorities.  Two users, each with ONE pool of its own sized to the
// cores, whatever the number of GPUs:
//   * the hasher (hasher.cpp): packing staging chunks for every device queue
//     (high priority: short tasks that keep a copy engine fed) and the SHA-1
//     streams of the objects kept off the GPU (low priority: long tasks nobody
//     waits on until the end of the batch);
//   * the object store (object_store.cpp): file reads of the next batch (high:
//     they feed the GPU) and deflate + loose-object writes (low: most of the work).
//
// Workers always serve the oldest high-priority group first, then the oldest
// low-priority group.  The thread that waits for a group helps with that group's
// tasks, so a group completes even on a pool of zero threads and a waiting
// thread is not idle.
//
// Placement: a worker may carry a node (the NUMA node its thread is pinned to) and
// a group may name the node it would like to run on (the node of the memory it
// writes).  A worker with a node takes the oldest high-priority group of its own
// node or of no node first and only then any other high-priority group: work is
// never left waiting for the right thread, it only goes to the right thread first.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace vit {

class WorkPool {
public:
    enum Priority { HIGH = 0, LOW = 1 };

    struct Group {
        std::function<void(size_t)> fn;
        size_t n = 0;
        int node = -1;                   // preferred node of the workers that run it; -1 = any
        size_t next = 0;                 // next unclaimed task (guarded by the pool's mutex)
        size_t done = 0;                 // finished tasks (guarded by the pool's mutex)
        std::condition_variable finished;
    };
    using Ticket = std::shared_ptr<Group>;

    // on_thread_start, when given, runs once at the start of every worker thread with the worker's index
    // and node (it is what pins the thread, if anything does).
    explicit WorkPool(int threads, std::function<void(int worker, int node)> on_thread_start = nullptr,
                      std::vector<int> worker_node = {})
        : on_start_(std::move(on_thread_start))
    {
        resize(threads, std::move(worker_node));
    }
    ~WorkPool() { stop(); }
    WorkPool(const WorkPool&) = delete;
    WorkPool& operator=(const WorkPool&) = delete;

    int threads() const { return int(workers_.size()); }

    // Most worker threads that were inside a task at the same moment since the
    // last reset (the calling threads that help are not counted).
    int peak_busy() const { return peak_busy_.load(); }
    void reset_peak() { peak_busy_ = 0; }

    // Stop the workers and start `threads` new ones; worker w gets node worker_node[w] (-1, or
    // no entry: none).  Only between batches.
    void resize(int threads, std::vector<int> worker_node = {})
    {
        stop();
        stop_ = false;
        worker_node.resize(size_t(threads < 0 ? 0 : threads), -1);
        for (int i = 0; i < threads; ++i) {
            const int node = worker_node[size_t(i)];
            workers_.emplace_back([this, i, node] { work(i, node); });
        }
    }

    // Queue fn(0) .. fn(n-1); returns at once.  fn must not throw.  node: see "Placement" above.
    Ticket submit(Priority prio, size_t n, std::function<void(size_t)> fn, int node = -1)
    {
        Ticket g = std::make_shared<Group>();
        g->fn = std::move(fn);
        g->n = n;
        g->node = node;
        if (n == 0) return g;
        {
            std::lock_guard<std::mutex> lk(m_);
            open_[prio].push_back(g);
        }
        if (n == 1) wake_.notify_one(); else wake_.notify_all();
        return g;
    }

    // Help with the group's unclaimed tasks, then wait for the claimed ones.
    void wait(const Ticket& g)
    {
        if (!g || g->n == 0) return;
        std::unique_lock<std::mutex> lk(m_);
        while (g->next < g->n) {
            const size_t i = g->next++;
            if (g->next == g->n) drop(g);
            lk.unlock();
            g->fn(i);
            lk.lock();
            ++g->done;
        }
        g->finished.wait(lk, [&] { return g->done == g->n; });
    }

    // Run ONE unclaimed task of the group on the calling thread, if it has any left.  For a thread that
    // has something else to wait for (a device event) and would rather work than sleep meanwhile.
    bool help(const Ticket& g)
    {
        if (!g || g->n == 0) return false;
        std::unique_lock<std::mutex> lk(m_);
        if (g->next >= g->n) return false;
        const size_t i = g->next++;
        if (g->next == g->n) drop(g);
        lk.unlock();
        g->fn(i);
        lk.lock();
        if (++g->done == g->n) g->finished.notify_all();
        return true;
    }

    // submit + wait.
    void run(Priority prio, size_t n, std::function<void(size_t)> fn, int node = -1) { wait(submit(prio, n, std::move(fn), node)); }

private:
    void drop(const Ticket& g)           // with m_ held: the group has no unclaimed task left
    {
        for (auto& q : open_)
            for (auto it = q.begin(); it != q.end(); ++it)
                if (*it == g) { q.erase(it); return; }
    }

    // with m_ held: the oldest group of that priority this worker should take first, or null
    Ticket local(Priority prio, int my_node) const
    {
        for (const Ticket& g : open_[prio])
            if (g->node < 0 || my_node < 0 || g->node == my_node) return g;
        return nullptr;
    }

    void work(int index, int my_node)
    {
        if (on_start_) on_start_(index, my_node);
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            wake_.wait(lk, [this] { return stop_ || !open_[HIGH].empty() || !open_[LOW].empty(); });
            if (stop_) return;
            Ticket g = local(HIGH, my_node);
            if (!g && !open_[HIGH].empty()) g = open_[HIGH].front();
            if (!g) g = local(LOW, my_node);
            if (!g) g = open_[LOW].front();
            const size_t i = g->next++;
            if (g->next == g->n) drop(g);
            const int busy = ++busy_;
            int peak = peak_busy_.load();
            while (busy > peak && !peak_busy_.compare_exchange_weak(peak, busy)) {}
            lk.unlock();
            g->fn(i);
            lk.lock();
            --busy_;
            if (++g->done == g->n) g->finished.notify_all();
        }
    }

    void stop()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        wake_.notify_all();
        for (auto& w : workers_) if (w.joinable()) w.join();
        workers_.clear();
    }

    std::function<void(int, int)> on_start_;
    std::vector<std::thread> workers_;
    std::deque<Ticket> open_[2];
    std::mutex m_;
    std::condition_variable wake_;
    bool stop_ = false;
    int busy_ = 0;
    std::atomic<int> peak_busy_{0};
};

}  // namespace vit
