// Small persistent thread pool for the host tails of a batched submission (prover_pipeline.hip) — in a header of its own so that
// tools/tail_pool_test.cpp can hammer it without a GPU (tests/test_host_cpu.py).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <stdint.h>

namespace zk {
// The host tails of the proofs of ONE batched submission are independent (0.3 ms each: the scalar multiplications of the
// final assembly) — and a collector that ran them one after the other was what bounded a 2^14 / 2^15 circuit: eight proofs per
// submission = 2.4 ms of tail for 1.6 ms of GPU work.  A few persistent threads shared by every prover of the process; the
// calling thread takes its share, so a machine with no spare core still makes progress.
struct TailPool {
    // fn points at the CALLER's std::function: valid because for_each() returns only when done == count, i.e. after the last
    // call of fn has returned — a job whose indices are all handed out is never entered again (work() leaves at next >= count).
    struct Job { const std::function<void(uint32_t)> *fn; std::atomic<uint32_t> next{0}, done{0}; uint32_t count = 0; };
    std::mutex m;
    std::condition_variable cv, fin;
    std::deque<std::shared_ptr<Job>> q;
    std::vector<std::thread> th;
    TailPool() {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned n = hw > 8 ? 7 : (hw > 1 ? hw - 1 : 0);
        for (unsigned i = 0; i < n; i++) th.emplace_back([this] { run(); });
    }
    static void work(Job &j) {
        for (;;) {
            const uint32_t k = j.next.fetch_add(1);
            if (k >= j.count) return;
            (*j.fn)(k);
            j.done.fetch_add(1);
        }
    }
    void run() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m);
                // any job that still has indices to hand out — not just the front one: with several GPUs, each running its own
                // collector, the batched tails of one submission used to queue behind another's last straggler
                cv.wait(lk, [&] {
                    for (auto &x : q)
                        if (x->next.load() < x->count) { j = x; return true; }
                    return false;
                });
            }
            work(*j);
            std::lock_guard<std::mutex> lk(m);
            fin.notify_all();
        }
    }
    // fn(0) .. fn(count - 1), each exactly once, on the pool's threads and the calling one; returns when all have returned.
    // fn must not throw.
    void for_each(uint32_t count, const std::function<void(uint32_t)> &fn) {
        if (count <= 1 || th.empty()) {
            for (uint32_t k = 0; k < count; k++) fn(k);
            return;
        }
        auto j = std::make_shared<Job>();
        j->fn = &fn;
        j->count = count;
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(j);
        }
        cv.notify_all();
        work(*j);
        std::unique_lock<std::mutex> lk(m);
        fin.wait(lk, [&] { return j->done.load() == count; });
        for (auto it = q.begin(); it != q.end(); ++it)
            if (*it == j) { q.erase(it); break; }
    }
};
inline TailPool &tail_pool() {
    static TailPool *pool = new TailPool();        // (never destroyed, like the staging pool)
    return *pool;
}
}   // namespace zk
