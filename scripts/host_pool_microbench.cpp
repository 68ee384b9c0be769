// Cost of one parallel_for of csrc/parallel.hpp with nothing to do, and of a streaming loop with a nested loop inside.
//   g++ -O2 -std=c++17 -pthread scripts/host_pool_microbench.cpp -o /tmp/host_pool && /tmp/host_pool
// Measured on the 256-core host of an MI355X box, 64 threads: 1.57 ms per loop when every loop started and joined its
// own threads, 0.155 ms on the persistent pool.
#include "../ferreus_rbf_rs_amd/csrc/parallel.hpp"
#include <chrono>
#include <cstdio>
int main() {
    using namespace bbfmm;
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
