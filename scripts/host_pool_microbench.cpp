// Cost of one parallel_for of csrc/parallel.hpp with nothing to do, and of a streaming loop with a nested loop inside.
//   g++ -O2 -std=c++17 -pthread scripts/host_pool_microbench.cpp -o /tmp/host_pool && /tmp/host_pool
// Measured on the 256-core host of an MI355X box, 64 threads: 1.57 ms per loop when every loop started and joined its
// own threads, 0.155 ms on the persistent pool.
#include "../ferreus_rbf_rs_amd/csrc/parallel.hpp"
#include <chrono>
#include <cstdio>
#include <stdexcept>
int main() {
    using namespace bbfmm;
    // (round 5; what scripts/sanitize_host.sh runs under -fsanitize=thread) loops started from two host threads at
    // once -- one gets the pool, the other falls back to threads of its own -- and an exception thrown inside a loop
    // while the other thread keeps looping: carried to the caller after every helper is back (parallel.hpp)
    {
        std::atomic<int64_t> total{0};
        std::atomic<int> caught{0};
        auto driver = [&](int id) {
            for (int i = 0; i < 300; ++i) {
                std::vector<int64_t> part(64, 0);
                parallel_for_chunks(64 * 1000, 1000, [&](int64_t b, int64_t e) { part[static_cast<size_t>(b / 1000)] = e - b; });
                int64_t s = 0;
                for (int64_t v : part) s += v;
                total += s;
                if (id == 1 && i % 50 == 7) {
                    try {
                        parallel_for(4096, 16, [&](int64_t k) { if (k == 1234) throw std::runtime_error("from a helper"); });
                    } catch (const std::runtime_error &) { ++caught; }
                }
            }
        };
        std::thread a(driver, 0), b(driver, 1);
        a.join();
        b.join();
        std::printf("two concurrent drivers: covered %lld of %lld, %d exceptions carried to their callers\n", (long long)total.load(),
                    (long long)(2 * 300 * 64000), caught.load());
        if (total.load() != 2 * 300 * 64000 || caught.load() != 6) return 1;
    }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::atomic<int64_t> s{0};
        for (int i = 0; i < 2000; ++i) parallel_for(host_threads(), 1, [&](int64_t k) { s += k; });
        auto t1 = std::chrono::steady_clock::now();
        std::printf("threads %d: %.1f us per empty parallel_for (sum %lld)\n", host_threads(), std::chrono::duration<double>(t1 - t0).count() / 2000 * 1e6, (long long)s.load());
    }
    // a real loop: sum of squares over 64M, nested loops inside
    std::vector<double> x(1 << 24, 1.5);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::atomic<int64_t> cnt{0};
        parallel_for_chunks((int64_t)x.size(), 1 << 16, [&](int64_t b, int64_t e) {
            double s = 0; for (int64_t i = b; i < e; ++i) s += x[i] * x[i];
            if (s > 0) cnt += e - b;
            if (b == 0) parallel_for(8, 1, [&](int64_t) { cnt += 0; }); // nested: falls back
        });
        auto t1 = std::chrono::steady_clock::now();
        std::printf("16M-element loop: %.2f ms, covered %lld\n", std::chrono::duration<double>(t1 - t0).count() * 1e3, (long long)cnt.load());
    }
}
