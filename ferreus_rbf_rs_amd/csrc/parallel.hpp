// Minimal host thread pool: dynamic chunked parallel_for over [0, n).
// (The reference parallelises its host loops with rayon; this plays that role for
// the host-side tree / list / operator setup and the solvers' vector loops.  No OpenMP runtime dependency.)
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include <pthread.h>

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

// CPUs of bandwidth the container grants (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`); 0: no limit.
inline double cgroup_cpu_quota() {
    double q = 0.0, p = 0.0;
    if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char first[32] = {0};
        const int got = std::fscanf(f, "%31s %lf", first, &p);
        std::fclose(f);
        if (got == 2 && first[0] != 'm' && p > 0.0) return std::atof(first) / p;
        if (got >= 1) return 0.0; // "max": unlimited
    }
    if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        const int got = std::fscanf(f, "%lf", &q);
        std::fclose(f);
        if (got == 1 && q > 0.0)
            if (std::FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                const int got2 = std::fscanf(g, "%lf", &p);
                std::fclose(g);
                if (got2 == 1 && p > 0.0) return q / p;
            }
    }
    return 0.0;
}

// Default: the hardware threads, at most 64 -- and at most two per CPU of a cgroup quota: the MI355X boxes show 256
// hardware threads and grant 16 CPUs of bandwidth; threads beyond that only get the process throttled (round 5: the
// Schwarz setup at 10M points 2.6-2.8 s on 64 threads, 2.2-2.3 s on 32).
inline int host_threads() {
    static int n = [] {
        if (const char *e = std::getenv("BBFMM_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) return v;
        }
        const unsigned hc = std::thread::hardware_concurrency();
        int t = static_cast<int>(hc == 0 ? 1 : std::min(hc, 64u));
        const double quota = cgroup_cpu_quota();
        if (quota > 0.0) t = std::min(t, std::max(4, static_cast<int>(2.0 * quota + 0.5)));
        return t;
    }();
    return n;
}

// Persistent helper threads.  Starting and joining 63 threads costs 1.6 ms on the 256-core host of an MI355X box
// (scripts/: measured), which a setup made of hundreds of short loops -- and every vector operation of FGMRES -- paid
// per loop.  One job at a time: a loop started while another one runs (a nested loop, a second host thread) falls back
// to threads of its own.  The pool is created on first use and never destroyed (its threads sleep until the process
// ends); a forked child starts a new one.
class HostPool {
public:
    static HostPool *get() {
        HostPool *p = slot().load(std::memory_order_acquire);
        if (p) return p;
        static std::mutex create;
        std::lock_guard<std::mutex> g(create);
        p = slot().load(std::memory_order_acquire);
        if (!p) {
            static const int registered = pthread_atfork(nullptr, nullptr, [] { slot().store(nullptr); busy_flag() = false; });
            (void)registered;
            p = new HostPool(std::max(0, host_threads() - 1));
            slot().store(p, std::memory_order_release);
        }
        return p;
    }
    // Runs worker() on the caller and on up to `helpers` pool threads; false (nothing done) when the pool is busy.
    template <class W> bool run(int helpers, W &worker) {
        if (busy_flag() || threads_.empty() || !job_.try_lock()) return false;
        busy_flag() = true;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = [](void *a) { (*static_cast<W *>(a))(); };
            arg_ = &worker;
            wanted_ = std::min<int>(helpers, static_cast<int>(threads_.size()));
        }
        cv_work_.notify_all();
        worker();
        {
            std::unique_lock<std::mutex> lk(m_);
            wanted_ = 0; // the caller ran out of chunks: helpers that have not started are not needed any more
            cv_done_.wait(lk, [&] { return running_ == 0; });
            fn_ = nullptr;
        }
        busy_flag() = false;
        job_.unlock();
        return true;
    }

private:
    explicit HostPool(int n) {
        threads_.reserve(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { loop(); });
    }
    void loop() {
        busy_flag() = true; // loops started from inside a job use threads of their own
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return wanted_ > 0; });
            --wanted_;
            ++running_;
            void (*fn)(void *) = fn_;
            void *arg = arg_;
            lk.unlock();
            fn(arg);
            lk.lock();
            if (--running_ == 0 && wanted_ == 0) cv_done_.notify_all();
        }
    }
    static std::atomic<HostPool *> &slot() {
        static std::atomic<HostPool *> p{nullptr};
        return p;
    }
    static bool &busy_flag() { // this thread is inside a job (as its caller or as a helper)
        static thread_local bool b = false;
        return b;
    }
    std::mutex m_, job_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> threads_;
    void (*fn_)(void *) = nullptr;
    void *arg_ = nullptr;
    int wanted_ = 0, running_ = 0;
};

// fn(begin, end) is called on disjoint chunks.  An exception thrown by fn on any thread (bad_alloc from a push_back,
// say) never leaves that thread's worker: the first one is kept, the remaining chunks are dropped, every thread
// finishes its current chunk, and the exception is rethrown on the caller once all of them are back -- the pool's
// job lock is released and no helper is left running on a dead stack frame.
template <class F> void parallel_for_chunks(int64_t n, int64_t chunk, F &&fn) {
    if (n <= 0) return;
    const int nt = static_cast<int>(std::min<int64_t>(host_threads(), (n + chunk - 1) / chunk));
    if (nt <= 1) {
        fn(int64_t(0), n);
        return;
    }
    std::atomic<int64_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr error;
    auto worker = [&]() noexcept {
        try {
            while (true) {
                const int64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                fn(b, std::min(n, b + chunk));
            }
        } catch (...) {
            if (!failed.exchange(true)) error = std::current_exception();
            next.store(n); // nobody starts another chunk
        }
    };
    if (!HostPool::get()->run(nt - 1, worker)) {
        std::vector<std::thread> threads;
        threads.reserve(nt - 1);
        for (int i = 0; i < nt - 1; ++i) threads.emplace_back(worker);
        worker();
        for (auto &t : threads) t.join();
    }
    if (failed.load()) std::rethrow_exception(error);
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
