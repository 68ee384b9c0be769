// See tree_device.hpp.  Integer work on HBM-resident arrays: coalesced streaming kernels + rocPRIM sorts / scans.
#include "tree_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "morton.hpp"

namespace bbfmm {
namespace {

__device__ inline uint64_t f64_to_u64_sat_d(double v) { // Rust `f64 as u64` (morton.rs:46)
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return ~uint64_t(0);
    return static_cast<uint64_t>(v);
}
__device__ inline uint64_t spread_d(uint64_t v, int d) { // bit i -> bit i*d (morton.rs:58-119)
    v &= 0xFFFF;
    if (d == 1) return v;
    if (d == 2) {
        v = (v | (v << 8)) & 0x00FF00FFull;
        v = (v | (v << 4)) & 0x0F0F0F0Full;
        v = (v | (v << 2)) & 0x33333333ull;
        v = (v | (v << 1)) & 0x55555555ull;
        return v;
    }
    v = (v | (v << 16)) & 0x0000FF0000FFull;
    v = (v | (v << 8)) & 0x00F00F00F00Full;
    v = (v | (v << 4)) & 0x0C30C30C30C3ull;
    v = (v | (v << 2)) & 0x249249249249ull;
    return v;
}

struct Disp {
    double v[3];
};

// level-16 anchors of every point, interleaved (the level-l code is this one shifted right by d (16 - l):
// the side lengths differ by exact powers of two); bad = a point outside the root box
__global__ __launch_bounds__(256) void codes_kernel(const double *__restrict__ x0, const double *__restrict__ x1,
                                                    const double *__restrict__ x2, int64_t n, int d, Disp disp,
                                                    double side16, uint64_t *__restrict__ code, uint32_t *__restrict__ idx,
                                                    int *__restrict__ bad) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const double *xs[3] = {x0, x1, x2};
    uint64_t c = 0;
    bool out = false;
    for (int a = 0; a < d; ++a) {
        const double q = floor((xs[a][i] - disp.v[a]) / side16); // point_to_anchor, morton.rs:35-51
        if (!(q >= 0.0 && q < 65536.0)) out = true;
        c |= spread_d(f64_to_u64_sat_d(q), d) << a;
    }
    code[i] = c;
    idx[i] = static_cast<uint32_t>(i);
    if (out) atomicOr(bad, 1);
}

// boundaries of the 2^d children of every active cell inside its range of the sorted codes
__global__ __launch_bounds__(256) void bounds_kernel(const uint64_t *__restrict__ code, const int32_t *__restrict__ act, int A,
                                                     const int32_t *__restrict__ cb, const int32_t *__restrict__ ce, int shift,
                                                     int nchild, int32_t *__restrict__ bounds) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<int64_t>(A) * (nchild + 1)) return;
    const int a = static_cast<int>(t / (nchild + 1)), s = static_cast<int>(t % (nchild + 1));
    const int ci = act[a];
    int lo = cb[ci], hi = ce[ci];
    if (s == 0) {
        bounds[t] = lo;
        return;
    }
    if (s == nchild) {
        bounds[t] = hi;
        return;
    }
    while (lo < hi) { // first position whose digit is >= s (digits are non-decreasing over the range)
        const int mid = lo + ((hi - lo) >> 1);
        const int dig = static_cast<int>((code[mid] >> shift) & static_cast<uint64_t>(nchild - 1));
        if (dig < s) lo = mid + 1;
        else hi = mid;
    }
    bounds[t] = lo;
}

__global__ __launch_bounds__(256) void exists_kernel(const int32_t *__restrict__ bounds, int A, int nchild, int64_t max_pts,
                                                     int store_empty, int adaptive, int32_t *__restrict__ flags,
                                                     int *__restrict__ any_exceeds) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<int64_t>(A) * nchild) return;
    const int a = static_cast<int>(t / nchild), s = static_cast<int>(t % nchild);
    const int32_t cnt = bounds[a * (nchild + 1) + s + 1] - bounds[a * (nchild + 1) + s];
    flags[t] = (cnt > 0 || store_empty) ? 1 : 0; // linear_tree.rs:69-75
    if (!adaptive && cnt > max_pts) atomicOr(any_exceeds, 1); // linear_tree.rs:96-101
}

struct LevelArrays {
    uint64_t *key;
    int32_t *b, *e, *parent;
    uint8_t *leaf;
    int32_t *next; // 1: subdivided at the next level
};

__global__ __launch_bounds__(256) void emit_kernel(const int32_t *__restrict__ act, int A, int nchild, int d,
                                                   const uint64_t *__restrict__ pkey, const int32_t *__restrict__ pb,
                                                   int32_t parent_off, const int32_t *__restrict__ bounds,
                                                   const int32_t *__restrict__ flags, const int32_t *__restrict__ pos,
                                                   int64_t max_pts, int store_empty, int adaptive, int child_level,
                                                   LevelArrays ch) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= static_cast<int64_t>(A) * nchild) return;
    if (!flags[t]) return;
    const int a = static_cast<int>(t / nchild), s = static_cast<int>(t % nchild);
    const int ci = act[a];
    const int o = pos[t];
    const int32_t b = bounds[a * (nchild + 1) + s], e = bounds[a * (nchild + 1) + s + 1];
    const int32_t cnt = e - b;
    const uint64_t k = pkey[ci];
    const uint64_t level = k & 0x7FFF;
    ch.key[o] = ((((k >> 15) << d) | static_cast<uint64_t>(s)) << 15) | (level + 1); // get_child, morton.rs:266-297
    ch.b[o] = cnt > 0 ? b : pb[ci]; // an empty child of a non-sparse tree keeps (b, b) of its parent
    ch.e[o] = cnt > 0 ? e : pb[ci];
    ch.parent[o] = parent_off + ci;
    uint8_t leaf = 0;
    int32_t next = 0;
    if (cnt > 0) { // linear_tree.rs:87-102
        if (adaptive) {
            if (cnt > max_pts && child_level < 16) next = 1;
            else leaf = 1;
        }
    } else if (adaptive && store_empty) { // 103-105
        leaf = 1;
    }
    if (!adaptive) next = 1; // 110-112
    ch.leaf[o] = leaf;
    ch.next[o] = next;
}

__global__ __launch_bounds__(256) void mark_leaves_kernel(const int32_t *__restrict__ next, int C, uint8_t *__restrict__ leaf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C && next[i]) leaf[i] = 1; // linear_tree.rs:123-130
}

inline unsigned blocks(int64_t n) { return static_cast<unsigned>((n + 255) / 256); }

struct DevPool { // everything allocated during a build, freed at the end unless released
    std::vector<void *> ptrs;
    ~DevPool() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T> T *get(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return static_cast<T *>(p);
    }
    void release(void *p) { ptrs.erase(std::remove(ptrs.begin(), ptrs.end(), p), ptrs.end()); }
};

#define TD_HIP(expr)                                       \
    do {                                                   \
        const hipError_t e__ = (expr);                     \
        if (e__ != hipSuccess) return -static_cast<int>(e__); \
    } while (0)
#define TD_PTR(p) \
    if (!(p)) return -static_cast<int>(hipErrorOutOfMemory)

} // namespace

void free_dev_tree_points(DevTreePoints *p) {
    for (double *&q : p->xyz) {
        if (q) (void)hipFree(q);
        q = nullptr;
    }
    if (p->order) (void)hipFree(p->order);
    p->order = nullptr;
    p->n = 0;
}

int build_tree_cells_device(const double *pts, int64_t n, int64_t ld, int d, const double *center, double radius,
                            int64_t max_points_per_cell, bool store_empty_leaves, bool adaptive_tree, HostTree *out,
                            std::vector<BuildCell> *cells, DevTreePoints *dev_points, hipStream_t s) {
    if (n < 1 || n >= (int64_t(1) << 31)) return 1;
    const int nchild = 1 << d;
    DevPool pool;
    Disp disp{{0, 0, 0}};
    for (int a = 0; a < d; ++a) disp.v[a] = center[a] - radius; // linear_tree.rs:30
    const uint64_t optimal_depth = // linear_tree.rs:31-32
        f64_to_u64_saturating(std::ceil(std::log2(static_cast<double>(n)) / static_cast<double>(d)));

    // points up, codes, one stable sort for all levels
    double *dx[3] = {nullptr, nullptr, nullptr};
    for (int a = 0; a < d; ++a) {
        dx[a] = pool.get<double>(static_cast<size_t>(n));
        TD_PTR(dx[a]);
        TD_HIP(hipMemcpyAsync(dx[a], pts + a * ld, static_cast<size_t>(n) * sizeof(double), hipMemcpyHostToDevice, s));
    }
    uint64_t *code_in = pool.get<uint64_t>(static_cast<size_t>(n)), *code = pool.get<uint64_t>(static_cast<size_t>(n));
    uint32_t *idx_in = pool.get<uint32_t>(static_cast<size_t>(n)), *idx = pool.get<uint32_t>(static_cast<size_t>(n));
    int *d_flags = pool.get<int>(4); // [0] bad point, [1] any child exceeds, [2], [3] counts
    TD_PTR(code_in);
    TD_PTR(code);
    TD_PTR(idx_in);
    TD_PTR(idx);
    TD_PTR(d_flags);
    TD_HIP(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(codes_kernel, dim3(blocks(n)), dim3(256), 0, s, dx[0], dx[1], dx[2], n, d, disp,
                       get_side_length(radius, kMaximumLevel), code_in, idx_in, d_flags);
    int h_flags[4] = {0, 0, 0, 0};
    TD_HIP(hipMemcpyAsync(h_flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
    TD_HIP(hipStreamSynchronize(s));
    if (h_flags[0]) return 1; // a point outside the root box: arbitrary keys per level, host path (tree.cpp)
    {
        size_t bytes = 0;
        TD_HIP(rocprim::radix_sort_pairs(nullptr, bytes, code_in, code, idx_in, idx, static_cast<size_t>(n), 0u,
                                         static_cast<unsigned>(16 * d), s));
        void *tmp = pool.get<uint8_t>(bytes);
        TD_PTR(tmp);
        TD_HIP(rocprim::radix_sort_pairs(tmp, bytes, code_in, code, idx_in, idx, static_cast<size_t>(n), 0u,
                                         static_cast<unsigned>(16 * d), s));
    }

    // level-by-level subdivision (linear_tree.rs:45-135)
    struct HostLevel {
        LevelArrays dev;
        int32_t count;
    };
    std::vector<HostLevel> levels;
    auto alloc_level = [&](int32_t cap, LevelArrays *la) {
        la->key = pool.get<uint64_t>(cap);
        la->b = pool.get<int32_t>(cap);
        la->e = pool.get<int32_t>(cap);
        la->parent = pool.get<int32_t>(cap);
        la->leaf = pool.get<uint8_t>(cap);
        la->next = pool.get<int32_t>(cap);
        return la->key && la->b && la->e && la->parent && la->leaf && la->next;
    };
    {
        LevelArrays root;
        if (!alloc_level(1, &root)) return -static_cast<int>(hipErrorOutOfMemory);
        const uint64_t k0 = 0;
        const int32_t b0 = 0, e0 = static_cast<int32_t>(n), p0 = -1, nx = 1;
        const uint8_t l0 = 0;
        TD_HIP(hipMemcpyAsync(root.key, &k0, 8, hipMemcpyHostToDevice, s));
        TD_HIP(hipMemcpyAsync(root.b, &b0, 4, hipMemcpyHostToDevice, s));
        TD_HIP(hipMemcpyAsync(root.e, &e0, 4, hipMemcpyHostToDevice, s));
        TD_HIP(hipMemcpyAsync(root.parent, &p0, 4, hipMemcpyHostToDevice, s));
        TD_HIP(hipMemcpyAsync(root.leaf, &l0, 1, hipMemcpyHostToDevice, s));
        TD_HIP(hipMemcpyAsync(root.next, &nx, 4, hipMemcpyHostToDevice, s));
        TD_HIP(hipStreamSynchronize(s));
        levels.push_back(HostLevel{root, 1});
    }
    int32_t n_active = 1;
    int32_t *act = pool.get<int32_t>(1);
    TD_PTR(act);
    TD_HIP(hipMemsetAsync(act, 0, sizeof(int32_t), s));
    int64_t cell_off = 0; // global index of the first cell of the current level
    uint64_t current_level = 0;
    while (true) {
        const uint64_t child_level = current_level + 1;
        const HostLevel par = levels.back();
        const int A = n_active;
        const int64_t nslots = static_cast<int64_t>(A) * nchild;
        if (nslots >= (int64_t(1) << 31)) return 1;
        int32_t *bounds = pool.get<int32_t>(static_cast<size_t>(A) * (nchild + 1));
        int32_t *flags = pool.get<int32_t>(static_cast<size_t>(nslots) + 1), *pos = pool.get<int32_t>(static_cast<size_t>(nslots) + 1);
        TD_PTR(bounds);
        TD_PTR(flags);
        TD_PTR(pos);
        const int shift = d * static_cast<int>(kMaximumLevel - child_level);
        hipLaunchKernelGGL(bounds_kernel, dim3(blocks(static_cast<int64_t>(A) * (nchild + 1))), dim3(256), 0, s, code, act, A,
                           par.dev.b, par.dev.e, shift, nchild, bounds);
        TD_HIP(hipMemsetAsync(flags + nslots, 0, sizeof(int32_t), s)); // one past the end: the scan leaves the total there
        hipLaunchKernelGGL(exists_kernel, dim3(blocks(nslots)), dim3(256), 0, s, bounds, A, nchild, max_points_per_cell,
                           store_empty_leaves ? 1 : 0, adaptive_tree ? 1 : 0, flags, d_flags + 1);
        {
            size_t bytes = 0;
            TD_HIP(rocprim::exclusive_scan(nullptr, bytes, flags, pos, 0, static_cast<size_t>(nslots) + 1, rocprim::plus<int32_t>(), s));
            void *tmp = pool.get<uint8_t>(bytes);
            TD_PTR(tmp);
            TD_HIP(rocprim::exclusive_scan(tmp, bytes, flags, pos, 0, static_cast<size_t>(nslots) + 1, rocprim::plus<int32_t>(), s));
        }
        int32_t n_children = 0;
        TD_HIP(hipMemcpyAsync(&n_children, pos + nslots, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        TD_HIP(hipMemcpyAsync(h_flags + 1, d_flags + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        TD_HIP(hipStreamSynchronize(s));
        LevelArrays ch;
        if (!alloc_level(std::max(n_children, 1), &ch)) return -static_cast<int>(hipErrorOutOfMemory);
        hipLaunchKernelGGL(emit_kernel, dim3(blocks(nslots)), dim3(256), 0, s, act, A, nchild, d, par.dev.key, par.dev.b,
                           static_cast<int32_t>(cell_off), bounds, flags, pos, max_points_per_cell, store_empty_leaves ? 1 : 0,
                           adaptive_tree ? 1 : 0, static_cast<int>(child_level), ch);
        // cells subdivided at the next level, in order
        int32_t *next_act = pool.get<int32_t>(std::max(n_children, 1));
        TD_PTR(next_act);
        int32_t n_next = 0;
        if (n_children > 0) {
            size_t bytes = 0;
            TD_HIP(rocprim::select(nullptr, bytes, rocprim::counting_iterator<int32_t>(0), ch.next, next_act, d_flags + 2,
                                   static_cast<size_t>(n_children), s));
            void *tmp = pool.get<uint8_t>(bytes);
            TD_PTR(tmp);
            TD_HIP(rocprim::select(tmp, bytes, rocprim::counting_iterator<int32_t>(0), ch.next, next_act, d_flags + 2,
                                   static_cast<size_t>(n_children), s));
            TD_HIP(hipMemcpyAsync(&n_next, d_flags + 2, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            TD_HIP(hipStreamSynchronize(s));
        }
        cell_off += par.count;
        levels.push_back(HostLevel{ch, n_children});
        const bool should_subdivide = // linear_tree.rs:115-118
            adaptive_tree || (h_flags[1] != 0 && child_level < kMaximumLevel && child_level < optimal_depth);
        if (should_subdivide && n_next > 0) {
            act = next_act;
            n_active = n_next;
            current_level += 1;
            TD_HIP(hipMemsetAsync(d_flags + 1, 0, sizeof(int), s));
        } else {
            if (!adaptive_tree && n_children > 0)
                hipLaunchKernelGGL(mark_leaves_kernel, dim3(blocks(n_children)), dim3(256), 0, s, ch.next, n_children, ch.leaf);
            break;
        }
    }
    TD_HIP(hipGetLastError());

    // cells to the host, level after level = (level, key) order: the children of key-ordered parents come out
    // key-ordered (child key = parent code << d | digit)
    int64_t C = 0;
    for (const HostLevel &l : levels) C += l.count;
    cells->clear();
    cells->resize(static_cast<size_t>(C));
    std::vector<int32_t> seg_b, seg_e;
    {
        std::vector<uint64_t> hk;
        std::vector<int32_t> hb, he, hp;
        std::vector<uint8_t> hl;
        int64_t off = 0;
        for (size_t lv = 0; lv < levels.size(); ++lv) {
            const HostLevel &l = levels[lv];
            const size_t m = static_cast<size_t>(l.count);
            hk.resize(m), hb.resize(m), he.resize(m), hp.resize(m), hl.resize(m);
            if (m) {
                TD_HIP(hipMemcpyAsync(hk.data(), l.dev.key, m * 8, hipMemcpyDeviceToHost, s));
                TD_HIP(hipMemcpyAsync(hb.data(), l.dev.b, m * 4, hipMemcpyDeviceToHost, s));
                TD_HIP(hipMemcpyAsync(he.data(), l.dev.e, m * 4, hipMemcpyDeviceToHost, s));
                TD_HIP(hipMemcpyAsync(hp.data(), l.dev.parent, m * 4, hipMemcpyDeviceToHost, s));
                TD_HIP(hipMemcpyAsync(hl.data(), l.dev.leaf, m, hipMemcpyDeviceToHost, s));
                TD_HIP(hipStreamSynchronize(s));
            }
            for (size_t i = 0; i < m; ++i) {
                BuildCell &c = (*cells)[static_cast<size_t>(off) + i];
                c.key = hk[i];
                c.level = static_cast<int32_t>(lv);
                c.parent = hp[i];
                c.b = hb[i];
                c.e = he[i];
                c.leaf = hl[i] != 0;
                if (c.leaf && c.e - c.b > 1) {
                    seg_b.push_back(hb[i]);
                    seg_e.push_back(he[i]);
                }
            }
            off += l.count;
        }
    }
    // rows ascending inside a leaf, as the reference's level-by-level stable grouping leaves them
    // (linear_tree.rs:55-66): segmented sort of the row indices over the leaves
    uint32_t *order = idx;
    if (!seg_b.empty()) {
        const size_t ns = seg_b.size();
        int32_t *d_sb = pool.get<int32_t>(ns), *d_se = pool.get<int32_t>(ns);
        TD_PTR(d_sb);
        TD_PTR(d_se);
        TD_HIP(hipMemcpyAsync(d_sb, seg_b.data(), ns * 4, hipMemcpyHostToDevice, s));
        TD_HIP(hipMemcpyAsync(d_se, seg_e.data(), ns * 4, hipMemcpyHostToDevice, s));
        // rows outside the sorted segments (leaves of one point) must carry over: start from a copy
        TD_HIP(hipMemcpyAsync(idx_in, idx, static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        unsigned bits = 1;
        while ((int64_t(1) << bits) < n) ++bits;
        size_t bytes = 0;
        TD_HIP(rocprim::segmented_radix_sort_keys(nullptr, bytes, idx, idx_in, static_cast<size_t>(n), static_cast<unsigned>(ns),
                                                  d_sb, d_se, 0u, bits, s));
        void *tmp = pool.get<uint8_t>(bytes);
        TD_PTR(tmp);
        TD_HIP(rocprim::segmented_radix_sort_keys(tmp, bytes, idx, idx_in, static_cast<size_t>(n), static_cast<unsigned>(ns),
                                                  d_sb, d_se, 0u, bits, s));
        order = idx_in;
    }
    std::vector<uint32_t> h_order(static_cast<size_t>(n));
    TD_HIP(hipMemcpyAsync(h_order.data(), order, static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    TD_HIP(hipStreamSynchronize(s));

    HostTree &t = *out;
    t = HostTree();
    t.d = d;
    t.radius = radius;
    t.n_points = n;
    t.adaptive = adaptive_tree;
    for (int a = 0; a < d; ++a) t.center[a] = center[a];
    t.depth = static_cast<int>(current_level + 1); // linear_tree.rs:160
    t.order.resize(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) t.order[static_cast<size_t>(i)] = static_cast<int64_t>(h_order[static_cast<size_t>(i)]);
    if (dev_points) { // the caller gathers its sorted coordinates from these on the device
        free_dev_tree_points(dev_points);
        for (int a = 0; a < d; ++a) {
            dev_points->xyz[a] = dx[a];
            pool.release(dx[a]);
        }
        dev_points->order = order;
        pool.release(order);
        dev_points->n = n;
    }
    return 0;
}

} // namespace bbfmm
