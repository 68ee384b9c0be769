// Interaction lists on the device (linear_tree.rs:177-485): U / V / W / X of every cell and the
// M2L transfer index of every V pair, from the numbered cells, their geometry and the open-addressing key table
// (uploaded as the host built them).  One thread per cell walks exactly the candidates the reference walks
// (parents' colleagues' children for V; colleagues, their ancestors and their descendants for U and W), in two
// passes (count, scan, fill); rows are then sorted by cell index with a segmented radix sort, X is the transpose
// of W by one stable pair sort.  The host lists (tree.cpp) stay as the checker and the fallback.
#include "tree_device.hpp"

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace bbfmm {
namespace {

struct DevTree {
    int d, depth;
    int64_t C;
    const uint64_t *key;
    const uint8_t *is_leaf;
    const double *centers; // C x d
    const double *lengths;
    const int32_t *pt_begin, *pt_end; // regular lists: has_points
    const int64_t *child_ptr;
    const int32_t *child_idx;
    const uint64_t *tab_keys;
    const int32_t *tab_vals;
    uint64_t tab_mask;
    double center[3], radius;
};

__device__ inline uint64_t mix_d(uint64_t x) { // KeyTable::hash
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
__device__ inline int32_t find_d(const DevTree &t, uint64_t key) {
    uint64_t h = mix_d(key) & t.tab_mask;
    while (true) {
        const int32_t v = t.tab_vals[h];
        if (v < 0) return -1;
        if (t.tab_keys[h] == key) return v;
        h = (h + 1) & t.tab_mask;
    }
}
__device__ inline uint64_t spread_l(uint64_t v, int d) {
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
__device__ inline void decode_d(uint64_t key, int d, uint64_t *anchor, uint64_t *level) { // morton.rs:127-167
    *level = key & 0x7FFF;
    const uint64_t k = key >> 15;
    if (d == 1) {
        anchor[0] = k & 0xFFFF;
        return;
    }
    const int nbits = d == 3 ? 21 : 28;
    for (int a = 0; a < d; ++a) {
        uint64_t v = 0;
        for (int i = 0; i < nbits; ++i) v |= ((k >> (d * i + a)) & 1ull) << i;
        anchor[a] = v;
    }
}
__device__ inline uint64_t encode_d(const uint64_t *anchor, uint64_t level, int d) { // morton.rs:58-119
    uint64_t code = 0;
    for (int a = 0; a < d; ++a) code |= spread_l(anchor[a], d) << a;
    return (code << 15) | level;
}
__device__ inline bool parent_d(uint64_t key, int d, uint64_t *parent) { // morton.rs:170-190
    const uint64_t level = key & 0x7FFF;
    if (level == 0) return false;
    *parent = (((key >> 15) >> d) << 15) | (level - 1);
    return true;
}
__device__ inline uint64_t child_d(uint64_t key, int d, uint64_t s) { // morton.rs:266-297
    return ((((key >> 15) << d) | s) << 15) | ((key & 0x7FFF) + 1);
}
// same-level neighbours inside the root box (morton.rs:214-263); any order: the lists are sorted afterwards
__device__ inline int neighbours_d(uint64_t key, int d, uint64_t *out) {
    uint64_t anchor[3] = {0, 0, 0}, level;
    decode_d(key, d, anchor, &level);
    const int64_t nmax = int64_t(1) << level;
    int cnt = 0;
    const int n3 = d == 1 ? 3 : (d == 2 ? 9 : 27);
    for (int q = 0; q < n3; ++q) {
        int dv[3] = {q % 3 - 1, (q / 3) % 3 - 1, (q / 9) % 3 - 1};
        if (dv[0] == 0 && (d < 2 || dv[1] == 0) && (d < 3 || dv[2] == 0)) continue;
        uint64_t na[3];
        bool ok = true;
        for (int a = 0; a < d; ++a) {
            const int64_t v = static_cast<int64_t>(anchor[a]) + dv[a];
            if (v < 0 || v >= nmax) ok = false;
            na[a] = static_cast<uint64_t>(v);
        }
        if (ok) out[cnt++] = encode_d(na, level, d);
    }
    return cnt;
}
__device__ inline void center_length_d(const DevTree &t, uint64_t key, double *c, double *len) { // morton.rs:328-346
    uint64_t anchor[3], level;
    decode_d(key, t.d, anchor, &level);
    const double side = 2.0 * t.radius / static_cast<double>(uint64_t(1) << level);
    for (int a = 0; a < t.d; ++a) c[a] = (static_cast<double>(anchor[a]) + 0.5) * side + (t.center[a] - t.radius);
    *len = side;
}
__device__ inline bool adjacent_d(const double *ca, double la, const double *cb, double lb, int d) { // morton.rs:308-325
    const double length = 0.5 * (la + lb);
    for (int a = 0; a < d; ++a)
        if (!(fabs(cb[a] - ca[a]) <= 1e-6 + length)) return false;
    return true;
}

struct Sink { // count pass: pointers null
    int32_t *u, *v, *w;
    int64_t nu = 0, nv = 0, nw = 0;
    __device__ void add_u(int32_t j) {
        if (u) u[nu] = j;
        ++nu;
    }
    __device__ void add_v(int32_t j) {
        if (v) v[nv] = j;
        ++nv;
    }
    __device__ void add_w(int32_t j) {
        if (w) w[nw] = j;
        ++nw;
    }
};

// linear_tree.rs:177-395 for one cell
__device__ void lists_adaptive_cell(const DevTree &t, int64_t c, Sink &out) {
    const int d = t.d, nchild = 1 << d;
    const uint64_t key = t.key[c];
    uint64_t parent_key;
    if (!parent_d(key, d, &parent_key)) return; // root: all lists empty (277)
    const double *cc = t.centers + c * d;
    const double lc = t.lengths[c];
    auto adjacent_to = [&](int32_t j) { return adjacent_d(cc, lc, t.centers + static_cast<int64_t>(j) * d, t.lengths[j], d); };
    uint64_t nb[26];
    const int nnb = neighbours_d(parent_key, d, nb); // V: children of the parent's colleagues, existing, not adjacent (278-293)
    for (int i = 0; i < nnb; ++i)
        for (int s = 0; s < nchild; ++s) {
            const int32_t j = find_d(t, child_d(nb[i], d, static_cast<uint64_t>(s)));
            if (j >= 0 && !adjacent_to(j)) out.add_v(j);
        }
    if (!t.is_leaf[c]) return;
    uint64_t col[26];
    const int ncol = neighbours_d(key, d, col);
    // colleagues and their ancestors (302-328).  The parent of an adjacent cell is adjacent too, so every chain
    // climbs to the root; `visited` keeps the walk to the distinct ancestors (at most 26 + 8 per level).
    uint64_t queue[26 + 8 * 17 + 8];
    int qn = 0, qi = 0;
    for (int i = 0; i < ncol; ++i) queue[qn++] = col[i];
    while (qi < qn) {
        const uint64_t cur = queue[qi];
        bool seen = false;
        for (int k = 0; k < qi && !seen; ++k) seen = queue[k] == cur; // entries before qi that were processed
        ++qi;
        if (seen) continue;
        double cb[3], lb;
        center_length_d(t, cur, cb, &lb);
        if (adjacent_d(cc, lc, cb, lb, d)) {
            const int32_t j = find_d(t, cur);
            if (j >= 0 && t.is_leaf[j]) {
                out.add_u(j);
            } else {
                uint64_t par;
                if (parent_d(cur, d, &par)) {
                    bool dup = false; // (a parent already queued would be skipped as `visited` anyway)
                    for (int k = 0; k < qn && !dup; ++k) dup = queue[k] == par;
                    if (!dup && qn < static_cast<int>(sizeof(queue) / sizeof(queue[0]))) queue[qn++] = par;
                }
            }
        }
    }
    // descendants of the colleagues (330-362), depth first (the order is immaterial: rows are sorted afterwards)
    int32_t stack[8 * 18];
    int sn = 0;
    for (int i = 0; i < ncol; ++i) {
        for (int s = 0; s < nchild; ++s) {
            const int32_t j0 = find_d(t, child_d(col[i], d, static_cast<uint64_t>(s)));
            if (j0 < 0) continue;
            stack[sn++] = j0;
            while (sn > 0) {
                const int32_t j = stack[--sn];
                if (adjacent_to(j)) {
                    if (t.is_leaf[j]) {
                        out.add_u(j);
                    } else {
                        for (int s2 = 0; s2 < nchild; ++s2) {
                            const int32_t g = find_d(t, child_d(t.key[j], d, static_cast<uint64_t>(s2)));
                            if (g >= 0) stack[sn++] = g;
                        }
                    }
                } else {
                    out.add_w(j);
                }
            }
        }
    }
    out.add_u(static_cast<int32_t>(c)); // 364
}

// linear_tree.rs:397-485 for one cell
__device__ void lists_regular_cell(const DevTree &t, int64_t c, Sink &out) {
    const int d = t.d;
    const uint64_t key = t.key[c];
    uint64_t parent_key;
    if (!parent_d(key, d, &parent_key)) return;
    const double *cc = t.centers + c * d;
    const double lc = t.lengths[c];
    const bool leaf = t.is_leaf[c] != 0;
    auto has_points = [&](int32_t j) { return t.pt_end[j] > t.pt_begin[j]; };
    if (leaf) { // 453-461
        const int32_t p = find_d(t, parent_key);
        if (p >= 0)
            for (int64_t q = t.child_ptr[p]; q < t.child_ptr[p + 1]; ++q)
                if (has_points(t.child_idx[q])) out.add_u(t.child_idx[q]);
    }
    uint64_t nb[26];
    const int nnb = neighbours_d(parent_key, d, nb);
    for (int i = 0; i < nnb; ++i) { // 462-481
        const int32_t pc = find_d(t, nb[i]);
        if (pc < 0) continue;
        for (int64_t q = t.child_ptr[pc]; q < t.child_ptr[pc + 1]; ++q) {
            const int32_t j = t.child_idx[q];
            if (!has_points(j)) continue;
            if (adjacent_d(cc, lc, t.centers + static_cast<int64_t>(j) * d, t.lengths[j], d)) {
                if (leaf) out.add_u(j);
            } else {
                out.add_v(j);
            }
        }
    }
}

__global__ __launch_bounds__(128) void count_kernel(DevTree t, int adaptive, int32_t *__restrict__ nu, int32_t *__restrict__ nv,
                                                    int32_t *__restrict__ nw) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x;
    if (c >= t.C) return;
    Sink s{nullptr, nullptr, nullptr};
    if (adaptive) lists_adaptive_cell(t, c, s);
    else lists_regular_cell(t, c, s);
    nu[c] = static_cast<int32_t>(s.nu);
    nv[c] = static_cast<int32_t>(s.nv);
    nw[c] = static_cast<int32_t>(s.nw);
}

__global__ __launch_bounds__(128) void fill_kernel(DevTree t, int adaptive, const int64_t *__restrict__ pu,
                                                   const int64_t *__restrict__ pv, const int64_t *__restrict__ pw,
                                                   int32_t *__restrict__ u, int32_t *__restrict__ v, int32_t *__restrict__ w) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x;
    if (c >= t.C) return;
    Sink s{u + pu[c], v + pv[c], w + pw[c]};
    if (adaptive) lists_adaptive_cell(t, c, s);
    else lists_regular_cell(t, c, s);
}

// rows of the duplicate-free U lists: the reference de-duplicates (sort_unique); the walk above cannot produce
// duplicates, and the checker compares against the host lists
__global__ __launch_bounds__(256) void widen_kernel(const int32_t *__restrict__ cnt, int64_t C, int64_t *__restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i <= C) out[i] = i < C ? cnt[i] : 0;
}

__global__ __launch_bounds__(256) void narrow_kernel(const int64_t *__restrict__ in, int64_t n, unsigned *__restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = static_cast<unsigned>(in[i]);
}

// owner cell of every W entry (for the transpose)
__global__ __launch_bounds__(256) void owners_kernel(const int64_t *__restrict__ ptr, int64_t C, int32_t *__restrict__ owner) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (c >= C) return;
    for (int64_t q = ptr[c]; q < ptr[c + 1]; ++q) owner[q] = static_cast<int32_t>(c);
}

__global__ __launch_bounds__(256) void histogram_kernel(const int32_t *__restrict__ keys, int64_t n, int32_t *__restrict__ cnt) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[keys[i]], 1);
}

// M2L transfer index of every V pair (bbfmm.rs:872-888, 989-998)
__global__ __launch_bounds__(256) void tidx_kernel(DevTree t, const int64_t *__restrict__ pv, const int32_t *__restrict__ v,
                                                   int16_t *__restrict__ tidx) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (c >= t.C) return;
    const int d = t.d;
    for (int64_t q = pv[c]; q < pv[c + 1]; ++q) {
        const int32_t vv = v[q];
        int tix = 0;
        for (int a = 0; a < d; ++a) {
            const double r = round((t.centers[c * d + a] - t.centers[static_cast<int64_t>(vv) * d + a]) / t.lengths[c]);
            tix = tix * 7 + (static_cast<int>(r) + 3);
        }
        tidx[q] = static_cast<int16_t>(tix);
    }
}

inline unsigned blocks(int64_t n, int b = 256) { return static_cast<unsigned>((n + b - 1) / b); }

struct Pool {
    std::vector<void *> ptrs;
    ~Pool() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T> T *get(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return static_cast<T *>(p);
    }
    template <class T> T *up(const std::vector<T> &v, hipStream_t s) {
        T *p = get<T>(v.size());
        if (p && !v.empty() && hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
        return p;
    }
};

#define TL_HIP(expr)                                       \
    do {                                                   \
        const hipError_t e__ = (expr);                     \
        if (e__ != hipSuccess) return -static_cast<int>(e__); \
    } while (0)
#define TL_PTR(p) \
    if (!(p)) return -static_cast<int>(hipErrorOutOfMemory)

// counts (int32 per cell) -> ptr (int64, C + 1) on the device and on the host; returns the total
int scan_counts(Pool &pool, const int32_t *cnt, int64_t C, int64_t **d_ptr, std::vector<int64_t> *h_ptr, hipStream_t s) {
    int64_t *wide = pool.get<int64_t>(static_cast<size_t>(C) + 1), *ptr = pool.get<int64_t>(static_cast<size_t>(C) + 1);
    TL_PTR(wide);
    TL_PTR(ptr);
    hipLaunchKernelGGL(widen_kernel, dim3(blocks(C + 1)), dim3(256), 0, s, cnt, C, wide);
    size_t bytes = 0;
    TL_HIP(rocprim::exclusive_scan(nullptr, bytes, wide, ptr, int64_t(0), static_cast<size_t>(C) + 1, rocprim::plus<int64_t>(), s));
    void *tmp = pool.get<uint8_t>(bytes);
    TL_PTR(tmp);
    TL_HIP(rocprim::exclusive_scan(tmp, bytes, wide, ptr, int64_t(0), static_cast<size_t>(C) + 1, rocprim::plus<int64_t>(), s));
    h_ptr->resize(static_cast<size_t>(C) + 1);
    TL_HIP(hipMemcpyAsync(h_ptr->data(), ptr, (static_cast<size_t>(C) + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    TL_HIP(hipStreamSynchronize(s));
    *d_ptr = ptr;
    return 0;
}

// sort every row ascending; returns the buffer that holds the result
int sort_rows(Pool &pool, int32_t *in, int64_t total, const int64_t *d_ptr, int64_t C, int bits, int32_t **out, hipStream_t s) {
    *out = in;
    if (total == 0) return 0;
    int32_t *buf = pool.get<int32_t>(static_cast<size_t>(total));
    unsigned *off = pool.get<unsigned>(static_cast<size_t>(C) + 1); // totals are below 2^31
    TL_PTR(buf);
    TL_PTR(off);
    hipLaunchKernelGGL(narrow_kernel, dim3(blocks(C + 1)), dim3(256), 0, s, d_ptr, C + 1, off);
    size_t bytes = 0;
    TL_HIP(rocprim::segmented_radix_sort_keys(nullptr, bytes, in, buf, static_cast<size_t>(total), static_cast<unsigned>(C), off,
                                              off + 1, 0u, static_cast<unsigned>(bits), s));
    void *tmp = pool.get<uint8_t>(bytes);
    TL_PTR(tmp);
    TL_HIP(rocprim::segmented_radix_sort_keys(tmp, bytes, in, buf, static_cast<size_t>(total), static_cast<unsigned>(C), off,
                                              off + 1, 0u, static_cast<unsigned>(bits), s));
    *out = buf;
    return 0;
}

} // namespace

int build_lists_device(HostTree *tree, hipStream_t s) {
    HostTree &t = *tree;
    const int64_t C = t.n_cells();
    if (C < 1 || C >= (int64_t(1) << 31)) return 1;
    Pool pool;
    DevTree dt{};
    dt.d = t.d;
    dt.depth = t.depth;
    dt.C = C;
    dt.radius = t.radius;
    for (int a = 0; a < 3; ++a) dt.center[a] = t.center[a];
    std::vector<int32_t> pb(static_cast<size_t>(C)), pe(static_cast<size_t>(C));
    for (int64_t c = 0; c < C; ++c) {
        pb[static_cast<size_t>(c)] = static_cast<int32_t>(t.pt_begin[c]);
        pe[static_cast<size_t>(c)] = static_cast<int32_t>(t.pt_end[c]);
    }
    dt.key = pool.up(t.key, s);
    dt.is_leaf = pool.up(t.is_leaf, s);
    dt.centers = pool.up(t.centers, s);
    dt.lengths = pool.up(t.lengths, s);
    dt.pt_begin = pool.up(pb, s);
    dt.pt_end = pool.up(pe, s);
    dt.child_ptr = pool.up(t.children.ptr, s);
    dt.child_idx = pool.up(t.children.idx, s);
    dt.tab_keys = pool.up(t.table.raw_keys(), s);
    dt.tab_vals = pool.up(t.table.raw_vals(), s);
    dt.tab_mask = t.table.mask();
    if (!dt.key || !dt.is_leaf || !dt.centers || !dt.lengths || !dt.pt_begin || !dt.pt_end || !dt.child_ptr || !dt.child_idx ||
        !dt.tab_keys || !dt.tab_vals)
        return -static_cast<int>(hipErrorOutOfMemory);

    int32_t *nu = pool.get<int32_t>(static_cast<size_t>(C)), *nv = pool.get<int32_t>(static_cast<size_t>(C)),
            *nw = pool.get<int32_t>(static_cast<size_t>(C));
    TL_PTR(nu);
    TL_PTR(nv);
    TL_PTR(nw);
    const int adaptive = t.adaptive ? 1 : 0;
    hipLaunchKernelGGL(count_kernel, dim3(blocks(C, 128)), dim3(128), 0, s, dt, adaptive, nu, nv, nw);
    int64_t *pu = nullptr, *pv = nullptr, *pw = nullptr;
    int rc;
    if ((rc = scan_counts(pool, nu, C, &pu, &t.u.ptr, s))) return rc;
    if ((rc = scan_counts(pool, nv, C, &pv, &t.v.ptr, s))) return rc;
    if ((rc = scan_counts(pool, nw, C, &pw, &t.w.ptr, s))) return rc;
    const int64_t tu = t.u.ptr[C], tv = t.v.ptr[C], tw = t.w.ptr[C];
    if (std::max({tu, tv, tw}) >= (int64_t(1) << 31)) return 1;
    int32_t *u = pool.get<int32_t>(static_cast<size_t>(tu)), *v = pool.get<int32_t>(static_cast<size_t>(tv)),
            *w = pool.get<int32_t>(static_cast<size_t>(tw));
    TL_PTR(u);
    TL_PTR(v);
    TL_PTR(w);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks(C, 128)), dim3(128), 0, s, dt, adaptive, pu, pv, pw, u, v, w);
    TL_HIP(hipGetLastError());
    int bits = 1;
    while ((int64_t(1) << bits) < C) ++bits;
    int32_t *us = nullptr, *vs = nullptr, *wsrt = nullptr;
    if ((rc = sort_rows(pool, u, tu, pu, C, bits, &us, s))) return rc;
    if ((rc = sort_rows(pool, v, tv, pv, C, bits, &vs, s))) return rc;
    if ((rc = sort_rows(pool, w, tw, pw, C, bits, &wsrt, s))) return rc;
    t.u.idx.resize(static_cast<size_t>(tu));
    t.v.idx.resize(static_cast<size_t>(tv));
    t.w.idx.resize(static_cast<size_t>(tw));
    if (tu) TL_HIP(hipMemcpyAsync(t.u.idx.data(), us, static_cast<size_t>(tu) * 4, hipMemcpyDeviceToHost, s));
    if (tv) TL_HIP(hipMemcpyAsync(t.v.idx.data(), vs, static_cast<size_t>(tv) * 4, hipMemcpyDeviceToHost, s));
    if (tw) TL_HIP(hipMemcpyAsync(t.w.idx.data(), wsrt, static_cast<size_t>(tw) * 4, hipMemcpyDeviceToHost, s));
    // M2L transfer indices of the V pairs
    t.v_tidx.resize(static_cast<size_t>(tv));
    if (tv) {
        int16_t *tix = pool.get<int16_t>(static_cast<size_t>(tv));
        TL_PTR(tix);
        hipLaunchKernelGGL(tidx_kernel, dim3(blocks(C)), dim3(256), 0, s, dt, pv, vs, tix);
        TL_HIP(hipMemcpyAsync(t.v_tidx.data(), tix, static_cast<size_t>(tv) * 2, hipMemcpyDeviceToHost, s));
    }
    // X = transpose of W (388-392): pairs (W cell, owner) sorted by W cell, stable, owners ascending inside a row
    t.x.ptr.assign(static_cast<size_t>(C) + 1, 0);
    t.x.idx.clear();
    if (tw) {
        int32_t *owner = pool.get<int32_t>(static_cast<size_t>(tw)), *k2 = pool.get<int32_t>(static_cast<size_t>(tw)),
                *o2 = pool.get<int32_t>(static_cast<size_t>(tw)), *xc = pool.get<int32_t>(static_cast<size_t>(C));
        TL_PTR(owner);
        TL_PTR(k2);
        TL_PTR(o2);
        TL_PTR(xc);
        hipLaunchKernelGGL(owners_kernel, dim3(blocks(C)), dim3(256), 0, s, pw, C, owner);
        size_t bytes = 0;
        TL_HIP(rocprim::radix_sort_pairs(nullptr, bytes, wsrt, k2, owner, o2, static_cast<size_t>(tw), 0u, static_cast<unsigned>(bits), s));
        void *tmp = pool.get<uint8_t>(bytes);
        TL_PTR(tmp);
        TL_HIP(rocprim::radix_sort_pairs(tmp, bytes, wsrt, k2, owner, o2, static_cast<size_t>(tw), 0u, static_cast<unsigned>(bits), s));
        TL_HIP(hipMemsetAsync(xc, 0, static_cast<size_t>(C) * 4, s));
        hipLaunchKernelGGL(histogram_kernel, dim3(blocks(tw)), dim3(256), 0, s, wsrt, tw, xc);
        int64_t *px = nullptr;
        std::vector<int64_t> hx;
        if ((rc = scan_counts(pool, xc, C, &px, &hx, s))) return rc;
        t.x.ptr.swap(hx);
        t.x.idx.resize(static_cast<size_t>(tw));
        TL_HIP(hipMemcpyAsync(t.x.idx.data(), o2, static_cast<size_t>(tw) * 4, hipMemcpyDeviceToHost, s));
    }
    TL_HIP(hipStreamSynchronize(s));
    TL_HIP(hipGetLastError());
    return 0;
}

} // namespace bbfmm
