// Minimal host thread pool helper: dynamic chunked parallel_for over [0, n).
// (The reference parallelises its host loops with rayon; this plays that role for
// the host-side tree / list / operator setup.  No OpenMP runtime dependency.)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

namespace bbfmm {

// std::vector without the serial zero fill of resize(): the elements are written by parallel loops right after.
template <class T> struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
using PodDoubles = std::vector<double, DefaultInitAllocator<double>>;

inline int host_threads() {
    static int n = [] {
        if (const char *e = std::getenv("BBFMM_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) return v;
        }
        const unsigned hc = std::thread::hardware_concurrency();
        return static_cast<int>(hc == 0 ? 1 : std::min(hc, 64u));
    }();
    return n;
}

// fn(begin, end) is called on disjoint chunks.
template <class F> void parallel_for_chunks(int64_t n, int64_t chunk, F &&fn) {
    if (n <= 0) return;
    const int nt = static_cast<int>(std::min<int64_t>(host_threads(), (n + chunk - 1) / chunk));
    if (nt <= 1) {
        fn(int64_t(0), n);
        return;
    }
    std::atomic<int64_t> next{0};
    auto worker = [&]() {
        while (true) {
            const int64_t b = next.fetch_add(chunk);
            if (b >= n) break;
            fn(b, std::min(n, b + chunk));
        }
    };
    std::vector<std::thread> threads;
    threads.reserve(nt - 1);
    for (int i = 0; i < nt - 1; ++i) threads.emplace_back(worker);
    worker();
    for (auto &t : threads) t.join();
}

template <class F> void parallel_for(int64_t n, int64_t chunk, F &&fn) {
    parallel_for_chunks(n, chunk, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) fn(i);
    });
}

// Stable LSD radix sort of (key, value) pairs by the low `nbits` bits of the keys, 8-bit digits,
// threaded: per-thread histograms over fixed chunks, one prefix over (digit, thread), scatter.
inline void parallel_radix_sort_pairs(std::vector<uint64_t> *keys, std::vector<int64_t> *vals, int nbits) {
    const int64_t n = static_cast<int64_t>(keys->size());
    if (n <= 1) return;
    const int T = std::max(1, host_threads());
    std::vector<uint64_t> ktmp(static_cast<size_t>(n));
    std::vector<int64_t> vtmp(static_cast<size_t>(n));
    const int64_t chunk = (n + T - 1) / T;
    std::vector<int64_t> hist(static_cast<size_t>(T) * 256);
    for (int shift = 0; shift < nbits; shift += 8) {
        std::fill(hist.begin(), hist.end(), 0);
        const std::vector<uint64_t> &k = *keys;
        parallel_for(T, 1, [&](int64_t th) {
            int64_t *h = &hist[static_cast<size_t>(th) * 256];
            for (int64_t i = th * chunk; i < std::min(n, (th + 1) * chunk); ++i) ++h[(k[i] >> shift) & 255];
        });
        int64_t run = 0;
        for (int bin = 0; bin < 256; ++bin)
            for (int th = 0; th < T; ++th) {
                const int64_t c = hist[static_cast<size_t>(th) * 256 + bin];
                hist[static_cast<size_t>(th) * 256 + bin] = run;
                run += c;
            }
        parallel_for(T, 1, [&](int64_t th) {
            int64_t *h = &hist[static_cast<size_t>(th) * 256];
            for (int64_t i = th * chunk; i < std::min(n, (th + 1) * chunk); ++i) {
                const int64_t dst = h[(k[i] >> shift) & 255]++;
                ktmp[dst] = k[i];
                vtmp[dst] = (*vals)[i];
            }
        });
        keys->swap(ktmp);
        vals->swap(vtmp);
    }
}

} // namespace bbfmm
