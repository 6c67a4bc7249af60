// Stress of csrc/tail_pool.hpp without a GPU: several caller threads (one per "prover") run for_each concurrently with counts
// 1 .. 8; every index of every call must run exactly once, and for_each must not return before its last index has.
//   g++ -O2 -std=c++17 -pthread [-fsanitize=thread] -I rapidsnark-old_amd/csrc tools/tail_pool_test.cpp -o /tmp/tail_pool_test && /tmp/tail_pool_test
#include "tail_pool.hpp"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv) {
    const int callers = argc > 1 ? atoi(argv[1]) : 6, rounds = argc > 2 ? atoi(argv[2]) : 3000;
    std::atomic<long> bad{0}, total{0};
    std::vector<std::thread> th;
    for (int c = 0; c < callers; c++)
        th.emplace_back([&, c] {
            unsigned seed = 12345u + 77u * (unsigned)c;
            for (int r = 0; r < rounds; r++) {
                seed = seed * 1664525u + 1013904223u;
                const uint32_t count = 1u + (seed >> 24) % 8u;
                std::atomic<int> hit[8];
                for (auto &h : hit) h.store(0);
                zk::tail_pool().for_each(count, [&](uint32_t k) {
                    volatile unsigned spin_sink = 0;
                    for (unsigned i = 0; i < 200u + (seed & 1023u); i++) spin_sink = spin_sink + i;      // a little work
                    hit[k].fetch_add(1);
                });
                for (uint32_t k = 0; k < 8; k++)
                    if (hit[k].load() != (k < count ? 1 : 0)) bad.fetch_add(1);
                total.fetch_add(count);
            }
        });
    for (auto &t : th) t.join();
    printf("%d callers x %d calls, %ld items, %ld wrong counts, pool threads %zu\n", callers, rounds, total.load(), bad.load(), zk::tail_pool().th.size());
    return bad.load() ? 1 : 0;
}
