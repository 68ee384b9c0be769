// Internals shared by the translation units of FmmTree (fmm_tree.cpp: creation, upload, passes, entry points;
// fmm_m2l_tables.cpp: the stacked M2L tables and the shared-basis extension; fmm_plans.cpp: target sets, restricted
// downward plans, target subsets, partitions): error macros, the device-buffer member templates, small helpers.
#pragma once
#include "fmm_tree.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <numeric>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "morton.hpp"
#include "parallel.hpp"
#include "ddm_solver.hpp"
#include "tree_device.hpp"


namespace bbfmm {

#define HIPCHK(expr)                                              \
    do {                                                          \
        hipError_t e__ = (expr);                                  \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);       \
    } while (0)
#define CHK(expr)                                                 \
    do {                                                          \
        int rc__ = (expr);                                        \
        if (rc__ != BBFMM_OK) return rc__;                        \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

namespace {
struct StageTimer { // BBFMM_VERBOSE=1 prints host setup stage times to stderr
    bool on = std::getenv("BBFMM_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm] %-28s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};
} // namespace

template <class T> int FmmTree::dalloc(DevBuf<T> *b, size_t n, bool zero) {
    b->n = n;
    b->p = nullptr;
    if (n == 0) n = 1;
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, n * sizeof(T)));
    owned_.push_back(p);
    b->p = static_cast<T *>(p);
    if (zero) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), stream_));
    return BBFMM_OK;
}
template <class T> int FmmTree::dupload(DevBuf<T> *b, const std::vector<T> &v) {
    CHK(dalloc(b, v.size()));
    if (!v.empty()) HIPCHK(hipMemcpy(b->p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BBFMM_OK;
}
template <class T> int FmmTree::talloc(DevBuf<T> *b, size_t n, bool zero) {
    if (!arena_active_) return dalloc(b, n, zero);
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~size_t(255);
    arena_need_ += bytes;
    if (arena_used_ + bytes > arena_.n) return dalloc(b, n, zero); // this call overflows: the arena grows afterwards
    b->p = reinterpret_cast<T *>(arena_.p + arena_used_);
    b->n = n;
    b->borrowed = true;
    arena_used_ += bytes;
    if (zero) HIPCHK(hipMemsetAsync(b->p, 0, std::max<size_t>(n, 1) * sizeof(T), stream_));
    return BBFMM_OK;
}
template <class T> int FmmTree::tupload(DevBuf<T> *b, const std::vector<T> &v) {
    CHK(talloc(b, v.size()));
    if (!v.empty()) HIPCHK(hipMemcpy(b->p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BBFMM_OK;
}
template <class T> void FmmTree::dfree(DevBuf<T> *b) {
    if (b->borrowed) {
        b->p = nullptr;
        b->n = 0;
        b->borrowed = false;
        return;
    }
    if (b->p) {
        auto it = std::find(owned_.begin(), owned_.end(), static_cast<void *>(b->p));
        if (it != owned_.end()) owned_.erase(it);
        (void)hipFree(b->p);
    }
    b->p = nullptr;
    b->n = 0;
}

// ------------------------------------------------------------------ M2L tables
// Folds the reference's symmetry permutations (bbfmm.rs:910-931,964-982) into stacked
// per-octant-class operators; see device.hip "M2L".
// One workgroup per CU runs at a time, so a launch of T equal tiles takes ceil(T / CUs) rounds.
// When the last round is at most half full its tiles are halved (a workgroup whose upper four
// waves hold no cells runs one wave per SIMD and takes about half the time): the tail costs half a
// round instead of a whole one.
inline void split_tile_tail(std::vector<M2lTileDesc> *tiles, int n_cu) {
    const size_t T = tiles->size();
    const size_t r = T % static_cast<size_t>(n_cu);
    if (r == 0 || r > static_cast<size_t>(n_cu) / 2) return;
    std::vector<M2lTileDesc> out(tiles->begin(), tiles->end() - static_cast<std::ptrdiff_t>(r));
    for (size_t i = T - r; i < T; ++i) {
        const M2lTileDesc td = (*tiles)[i];
        if (td.count <= kM2lTile / 2) {
            out.push_back(td);
            continue;
        }
        M2lTileDesc a = td, b = td;
        a.count = kM2lTile / 2;
        b.first = td.first + kM2lTile / 2;
        b.count = td.count - kM2lTile / 2;
        out.push_back(a);
        out.push_back(b);
    }
    tiles->swap(out);
}

} // namespace bbfmm
