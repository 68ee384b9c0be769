// Host-side linear Morton tree and interaction lists.
// Restates ferreus_bbfmm/src/linear_tree.rs (citations inline) with deterministic
// containers.  See tree.hpp.
#include "tree.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "morton.hpp"
#include "parallel.hpp"

namespace bbfmm {

void KeyTable::build(const std::vector<uint64_t> &keys) {
    uint64_t cap = 16;
    while (cap < keys.size() * 2 + 2) cap <<= 1;
    mask_ = cap - 1;
    keys_.assign(cap, 0);
    vals_.assign(cap, -1);
    for (size_t i = 0; i < keys.size(); ++i) {
        uint64_t h = hash(keys[i]) & mask_;
        while (vals_[h] >= 0) {
            if (keys_[h] == keys[i]) break; // duplicate key: keep the first
            h = (h + 1) & mask_;
        }
        if (vals_[h] < 0) {
            keys_[h] = keys[i];
            vals_[h] = static_cast<int32_t>(i);
        }
    }
}

namespace {

using TmpCell = BuildCell;

inline uint64_t point_key(const double *pts, int64_t ld, int64_t i, int d, const double *disp,
                          double side, uint64_t level) {
    uint64_t anchor[3];
    for (int a = 0; a < d; ++a) anchor[a] = point_to_anchor_axis(pts[a * ld + i], disp[a], side);
    return encode_morton_point(anchor, level, d);
}

void flatten(const std::vector<std::vector<int32_t>> &rows, Csr *out) {
    const size_t n = rows.size();
    out->ptr.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) out->ptr[i + 1] = out->ptr[i] + static_cast<int64_t>(rows[i].size());
    out->idx.resize(static_cast<size_t>(out->ptr[n]));
    parallel_for(static_cast<int64_t>(n), 1024, [&](int64_t i) {
        std::copy(rows[i].begin(), rows[i].end(), out->idx.begin() + out->ptr[i]);
    });
}

inline void sort_unique(std::vector<int32_t> &v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
}

// linear_tree.rs:177-395
void interaction_lists_adaptive(HostTree &t) {
    const int d = t.d;
    const int64_t C = t.n_cells();
    const int nchild = 1 << d;
    std::vector<std::vector<int32_t>> U(C), V(C), W(C);

    parallel_for(C, 64, [&](int64_t c) {
        const uint64_t key = t.key[c];
        uint64_t parent_key;
        if (!get_parent(key, d, &parent_key)) return; // root: all lists empty (277)
        const double *cc = &t.centers[c * d];
        const double lc = t.lengths[c];
        auto adjacent_to_existing = [&](int32_t j) {
            return are_adjacent_cl(cc, lc, &t.centers[static_cast<int64_t>(j) * d], t.lengths[j], d);
        };
        // V list: children of the parent's colleagues, existing, not adjacent (278-293)
        uint64_t nb[26];
        const int nnb = get_neighbours(parent_key, d, nb);
        auto &vl = V[c];
        for (int i = 0; i < nnb; ++i)
            for (int s = 0; s < nchild; ++s) {
                const int32_t j = t.table.find(get_child(nb[i], d, static_cast<uint64_t>(s)));
                if (j >= 0 && !adjacent_to_existing(j)) vl.push_back(j);
            }
        sort_unique(vl);

        if (!t.is_leaf[c]) return;
        auto &ul = U[c];
        auto &wl = W[c];
        uint64_t colleagues[26];
        const int ncol = get_neighbours(key, d, colleagues);

        // colleagues and their ancestors (302-328)
        std::vector<uint64_t> queue(colleagues, colleagues + ncol);
        std::vector<uint64_t> visited;
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const uint64_t cur = queue[qi];
            if (std::find(visited.begin(), visited.end(), cur) != visited.end()) continue;
            visited.push_back(cur);
            double cb[3], lb;
            get_center_length(cur, t.center, t.radius, d, cb, &lb);
            if (are_adjacent_cl(cc, lc, cb, lb, d)) {
                const int32_t j = t.table.find(cur);
                if (j >= 0 && t.is_leaf[j]) {
                    ul.push_back(j);
                } else {
                    uint64_t par;
                    if (get_parent(cur, d, &par)) queue.push_back(par);
                }
            }
        }
        // descendants of the colleagues (330-362)
        std::vector<int32_t> dq;
        for (int i = 0; i < ncol; ++i)
            for (int s = 0; s < nchild; ++s) {
                const int32_t j = t.table.find(get_child(colleagues[i], d, static_cast<uint64_t>(s)));
                if (j >= 0) dq.push_back(j);
            }
        for (size_t qi = 0; qi < dq.size(); ++qi) {
            const int32_t j = dq[qi];
            if (adjacent_to_existing(j)) {
                if (t.is_leaf[j]) {
                    ul.push_back(j);
                } else {
                    for (int s = 0; s < nchild; ++s) {
                        const int32_t g = t.table.find(get_child(t.key[j], d, static_cast<uint64_t>(s)));
                        if (g >= 0) dq.push_back(g);
                    }
                }
            } else {
                wl.push_back(j);
            }
        }
        ul.push_back(static_cast<int32_t>(c)); // 364
        sort_unique(ul);
        sort_unique(wl);
    });

    // X = transpose of W (388-392)
    std::vector<std::vector<int32_t>> X(C);
    for (int64_t c = 0; c < C; ++c)
        for (int32_t w : W[c]) X[w].push_back(static_cast<int32_t>(c));
    flatten(U, &t.u);
    flatten(V, &t.v);
    flatten(W, &t.w);
    flatten(X, &t.x);
}

// linear_tree.rs:397-485
void interaction_lists_regular(HostTree &t) {
    const int d = t.d;
    const int64_t C = t.n_cells();
    std::vector<std::vector<int32_t>> U(C), V(C);
    parallel_for(C, 64, [&](int64_t c) {
        const uint64_t key = t.key[c];
        uint64_t parent_key;
        if (!get_parent(key, d, &parent_key)) return;
        const double *cc = &t.centers[c * d];
        const double lc = t.lengths[c];
        const bool leaf = t.is_leaf[c] != 0;
        auto has_points = [&](int32_t j) { return t.pt_end[j] > t.pt_begin[j]; };
        if (leaf) { // 453-461
            const int32_t p = t.table.find(parent_key);
            if (p >= 0)
                for (int64_t q = t.children.ptr[p]; q < t.children.ptr[p + 1]; ++q)
                    if (has_points(t.children.idx[q])) U[c].push_back(t.children.idx[q]);
        }
        uint64_t nb[26];
        const int nnb = get_neighbours(parent_key, d, nb);
        for (int i = 0; i < nnb; ++i) { // 462-481
            const int32_t pc = t.table.find(nb[i]);
            if (pc < 0) continue;
            for (int64_t q = t.children.ptr[pc]; q < t.children.ptr[pc + 1]; ++q) {
                const int32_t j = t.children.idx[q];
                if (!has_points(j)) continue;
                if (are_adjacent_cl(cc, lc, &t.centers[static_cast<int64_t>(j) * d], t.lengths[j], d)) {
                    if (leaf) U[c].push_back(j);
                } else {
                    V[c].push_back(j);
                }
            }
        }
        sort_unique(U[c]);
        sort_unique(V[c]);
    });
    flatten(U, &t.u);
    flatten(V, &t.v);
    t.w.ptr.assign(C + 1, 0);
    t.w.idx.clear();
    t.x.ptr.assign(C + 1, 0);
    t.x.idx.clear();
}

} // namespace

namespace {
struct TreeTimer { // BBFMM_VERBOSE=1: tree build stage times on stderr
    bool on = std::getenv("BBFMM_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm]   tree: %-22s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};
} // namespace

void build_tree(const double *pts, int64_t n, int64_t ld, int d, const double *center, double radius,
                int64_t max_points_per_cell, bool store_empty_leaves, bool adaptive_tree,
                HostTree *out) {
    TreeTimer timer;
    HostTree &t = *out;
    t = HostTree();
    t.d = d;
    t.radius = radius;
    t.n_points = n;
    t.adaptive = adaptive_tree;
    double disp[3] = {0, 0, 0};
    for (int a = 0; a < d; ++a) {
        t.center[a] = center[a];
        disp[a] = center[a] - radius; // linear_tree.rs:30
    }
    // linear_tree.rs:31-32 (`as u64` saturates: log2(0) = -inf -> 0)
    const uint64_t optimal_depth =
        f64_to_u64_saturating(std::ceil(std::log2(static_cast<double>(n)) / static_cast<double>(d)));
    const int nchild = 1 << d;

    t.order.resize(static_cast<size_t>(n));
    std::iota(t.order.begin(), t.order.end(), int64_t(0));
    std::vector<uint64_t> keys(static_cast<size_t>(n));
    std::vector<int64_t> tmp(static_cast<size_t>(n));

    // Fast path (every point inside the root box): the level-l anchor of a point is its level-16
    // anchor shifted right by 16 - l (the side lengths differ by exact powers of two), so one stable
    // sort by the interleaved level-16 code orders the points for every level at once and the
    // children of a cell are found by searching its range for the digit boundaries.  Inside a leaf
    // the row order of the level-by-level stable grouping is restored at the end.  Points outside
    // the box (explicit extents smaller than the data) take the level-by-level path below.
    std::vector<uint64_t> code16;
    bool fast = n > 0;
    {
        const double side16 = get_side_length(radius, kMaximumLevel);
        std::vector<uint8_t> bad_chunk(static_cast<size_t>((n + (1 << 15) - 1) >> 15) + 1, 0);
        code16.resize(static_cast<size_t>(n));
        parallel_for_chunks(n, 1 << 15, [&](int64_t lo, int64_t hi) {
            bool bad = false;
            for (int64_t i = lo; i < hi; ++i) {
                uint64_t code = 0;
                for (int a = 0; a < d; ++a) {
                    const double q = std::floor((pts[a * ld + i] - disp[a]) / side16);
                    if (!(q >= 0.0 && q < 65536.0)) bad = true;
                    code |= spread_bits(f64_to_u64_saturating(q), d) << a;
                }
                code16[i] = code;
            }
            if (bad) bad_chunk[static_cast<size_t>(lo >> 15)] = 1;
        });
        for (uint8_t b : bad_chunk) fast = fast && !b;
    }
    if (fast) {
        parallel_radix_sort_pairs(&code16, &t.order, 16 * d);
    } else {
        std::vector<uint64_t>().swap(code16);
    }

    std::vector<TmpCell> cells;
    cells.push_back(TmpCell{0, 0, -1, 0, n, false});
    std::vector<int32_t> active{0}, next;
    uint64_t current_level = 0;

    struct Group {
        uint64_t key;
        int64_t b, e;
    };

    while (!active.empty()) {
        next.clear();
        const uint64_t child_level = current_level + 1;
        const double side = get_side_length(radius, child_level); // linear_tree.rs:49
        bool any_child_exceeds = false;

        // child key of every point of every active cell (linear_tree.rs:56-61)
        if (!fast)
        for (int32_t ci : active) {
            const int64_t b = cells[ci].b, e = cells[ci].e;
            parallel_for_chunks(e - b, 1 << 15, [&](int64_t lo, int64_t hi) {
                for (int64_t i = b + lo; i < b + hi; ++i)
                    keys[i] = point_key(pts, ld, t.order[i], d, disp, side, child_level);
            });
        }
        // group the points of each active cell by child key, keeping row order.
        std::vector<std::vector<Group>> groups(active.size());
        parallel_for(static_cast<int64_t>(active.size()), 1, [&](int64_t ai) {
            const TmpCell cell = cells[active[ai]];
            const int64_t b = cell.b, e = cell.e;
            if (e == b) return;
            if (fast) { // the digit of the child level is non-decreasing over the sorted range
                const int shift = d * static_cast<int>(kMaximumLevel - child_level);
                auto &gf = groups[ai];
                int64_t lo = b;
                while (lo < e) {
                    const uint64_t dig = (code16[lo] >> shift) & static_cast<uint64_t>(nchild - 1);
                    const int64_t hi = std::partition_point(code16.begin() + lo, code16.begin() + e,
                                                            [&](uint64_t c) {
                                                                return ((c >> shift) & static_cast<uint64_t>(nchild - 1)) <= dig;
                                                            }) -
                                       code16.begin();
                    gf.push_back(Group{get_child(cell.key, d, dig), lo, hi});
                    lo = hi;
                }
                return;
            }
            bool proper = true;
            for (int64_t i = b; i < e && proper; ++i) {
                uint64_t par;
                proper = get_parent(keys[i], d, &par) && par == cell.key;
            }
            auto &g = groups[ai];
            if (proper) { // counting sort over the 2^d children (stable)
                int64_t cnt[9] = {0};
                for (int64_t i = b; i < e; ++i) ++cnt[get_child_index(keys[i], d) + 1];
                for (int s = 0; s < nchild; ++s) cnt[s + 1] += cnt[s];
                int64_t pos[8];
                for (int s = 0; s < nchild; ++s) pos[s] = b + cnt[s];
                for (int64_t i = b; i < e; ++i) tmp[pos[get_child_index(keys[i], d)]++] = t.order[i];
                std::copy(tmp.begin() + b, tmp.begin() + e, t.order.begin() + b);
                for (int s = 0; s < nchild; ++s)
                    if (cnt[s + 1] > cnt[s])
                        g.push_back(Group{get_child(cell.key, d, static_cast<uint64_t>(s)), b + cnt[s],
                                          b + cnt[s + 1]});
            } else { // sources outside the root box: arbitrary keys, stable sort
                std::vector<std::pair<uint64_t, int64_t>> kv;
                kv.reserve(static_cast<size_t>(e - b));
                for (int64_t i = b; i < e; ++i) kv.emplace_back(keys[i], t.order[i]);
                std::stable_sort(kv.begin(), kv.end(),
                                 [](const auto &x, const auto &y) { return x.first < y.first; });
                for (int64_t i = b; i < e; ++i) t.order[i] = kv[i - b].second;
                int64_t s = 0;
                for (int64_t i = 1; i <= e - b; ++i)
                    if (i == e - b || kv[i].first != kv[s].first) {
                        g.push_back(Group{kv[s].first, b + s, b + i});
                        s = i;
                    }
            }
        });

        for (size_t ai = 0; ai < active.size(); ++ai) {
            const int32_t ci = active[ai];
            const uint64_t cell_key = cells[ci].key;
            const auto &g = groups[ai];
            std::vector<Group> children;
            if (store_empty_leaves) { // linear_tree.rs:69-73: all 2^d children
                for (int s = 0; s < nchild; ++s) {
                    const uint64_t ck = get_child(cell_key, d, static_cast<uint64_t>(s));
                    Group grp{ck, cells[ci].b, cells[ci].b};
                    for (const auto &x : g)
                        if (x.key == ck) grp = x;
                    children.push_back(grp);
                }
            } else { // linear_tree.rs:74: occupied children only
                children = g;
            }
            for (const auto &ch : children) {
                TmpCell nc{ch.key, static_cast<int32_t>(child_level), ci, ch.b, ch.e, false};
                const int64_t cnt = ch.e - ch.b;
                const int32_t idx = static_cast<int32_t>(cells.size());
                if (cnt > 0) { // linear_tree.rs:87-102
                    if (adaptive_tree) {
                        if (cnt > max_points_per_cell && child_level < kMaximumLevel)
                            next.push_back(idx);
                        else
                            nc.leaf = true;
                    } else if (cnt > max_points_per_cell) {
                        any_child_exceeds = true;
                    }
                } else if (adaptive_tree && store_empty_leaves) { // 103-105
                    nc.leaf = true;
                }
                if (!adaptive_tree) next.push_back(idx); // 110-112
                cells.push_back(nc);
            }
        }

        const bool should_subdivide = // linear_tree.rs:115-118
            adaptive_tree ||
            (any_child_exceeds && child_level < kMaximumLevel && child_level < optimal_depth);
        if (should_subdivide && !next.empty()) {
            active.swap(next);
            current_level += 1;
        } else {
            if (!adaptive_tree)
                for (int32_t leaf : next) cells[leaf].leaf = true; // 123-130
            active.clear();
        }
    }
    t.depth = static_cast<int>(current_level + 1); // linear_tree.rs:160
    if (fast) // rows ascending inside a leaf, as the level-by-level stable grouping leaves them
        parallel_for(static_cast<int64_t>(cells.size()), 64, [&](int64_t ci) {
            const TmpCell &c = cells[ci];
            if (c.leaf && c.e - c.b > 1) std::sort(t.order.begin() + c.b, t.order.begin() + c.e);
        });
    timer.lap("subdivision");
    finish_tree(cells, out);
}

// Cell numbering, geometry, children, key table, interaction lists: everything after the subdivision
// (t.d, t.center, t.radius, t.n_points, t.adaptive, t.depth and t.order are set).  Shared by the host build
// above and the device build (tree_device.hip), whose cells arrive already in (level, key) order.
void finish_tree(const std::vector<BuildCell> &cells, HostTree *out, bool with_lists) {
    TreeTimer timer;
    HostTree &t = *out;
    const int d = t.d;
    const double radius = t.radius;
    // number the cells by (level, key)
    const int64_t C = static_cast<int64_t>(cells.size());
    std::vector<int32_t> perm(C);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
        if (cells[a].level != cells[b].level) return cells[a].level < cells[b].level;
        return cells[a].key < cells[b].key;
    });
    std::vector<int32_t> newidx(C);
    for (int64_t i = 0; i < C; ++i) newidx[perm[i]] = static_cast<int32_t>(i);

    t.key.resize(C);
    t.level.resize(C);
    t.parent.resize(C);
    t.octant.resize(C);
    t.is_leaf.resize(C);
    t.pt_begin.resize(C);
    t.pt_end.resize(C);
    t.centers.resize(static_cast<size_t>(C) * d);
    t.lengths.resize(C);
    for (int64_t i = 0; i < C; ++i) {
        const TmpCell &c = cells[perm[i]];
        t.key[i] = c.key;
        t.level[i] = c.level;
        t.parent[i] = c.parent < 0 ? -1 : newidx[c.parent];
        t.octant[i] = get_child_index(c.key, d);
        t.is_leaf[i] = c.leaf ? 1 : 0;
        t.pt_begin[i] = c.b;
        t.pt_end[i] = c.e;
        get_center_length(c.key, t.center, radius, d, &t.centers[i * d], &t.lengths[i]);
    }
    t.level_ptr.assign(static_cast<size_t>(t.depth) + 2, 0);
    for (int64_t i = 0; i < C; ++i) ++t.level_ptr[t.level[i] + 1];
    for (int l = 0; l <= t.depth; ++l) t.level_ptr[l + 1] += t.level_ptr[l];

    // children (cells are sorted, so each child list is sorted by key)
    t.children.ptr.assign(C + 1, 0);
    for (int64_t i = 0; i < C; ++i)
        if (t.parent[i] >= 0) ++t.children.ptr[t.parent[i] + 1];
    for (int64_t i = 0; i < C; ++i) t.children.ptr[i + 1] += t.children.ptr[i];
    t.children.idx.resize(static_cast<size_t>(t.children.ptr[C]));
    {
        std::vector<int64_t> pos(t.children.ptr.begin(), t.children.ptr.end() - 1);
        for (int64_t i = 0; i < C; ++i)
            if (t.parent[i] >= 0) t.children.idx[pos[t.parent[i]]++] = static_cast<int32_t>(i);
    }
    t.table.build(t.key);
    timer.lap("numbering, children");

    if (with_lists) build_lists_host(out);
}

// linear_tree.rs:177-485 and the M2L transfer index of every V pair
void build_lists_host(HostTree *out) {
    TreeTimer timer;
    HostTree &t = *out;
    const int d = t.d;
    const int64_t C = t.n_cells();
    if (t.adaptive)
        interaction_lists_adaptive(t);
    else
        interaction_lists_regular(t);
    timer.lap("interaction lists");

    // M2L transfer index of every V pair (bbfmm.rs:872-888, 989-998)
    t.v_tidx.resize(t.v.idx.size());
    parallel_for(C, 256, [&](int64_t c) {
        for (int64_t q = t.v.ptr[c]; q < t.v.ptr[c + 1]; ++q) {
            const int32_t v = t.v.idx[q];
            int tix = 0;
            for (int a = 0; a < d; ++a) {
                const double r =
                    std::round((t.centers[c * d + a] - t.centers[static_cast<int64_t>(v) * d + a]) / t.lengths[c]);
                tix = tix * 7 + (static_cast<int>(r) + 3);
            }
            t.v_tidx[q] = static_cast<int16_t>(tix);
        }
    });
}

int64_t points_to_leaves(const HostTree &t, const double *x, int64_t m, int64_t ldx,
                         int32_t *cell_out) {
    const int d = t.d;
    const uint64_t depth = static_cast<uint64_t>(t.depth);
    const double side = get_side_length(t.radius, depth); // linear_tree.rs:495
    double disp[3] = {0, 0, 0};
    for (int a = 0; a < d; ++a) disp[a] = t.center[a] - t.radius;
    std::atomic<int64_t> bad{-1};
    parallel_for_chunks(m, 4096, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            uint64_t cur = point_key(x, ldx, i, d, disp, side, depth);
            int32_t found = -1;
            while (true) { // linear_tree.rs:505-508
                const int32_t j = t.table.find(cur);
                if (j >= 0 && t.is_leaf[j]) {
                    found = j;
                    break;
                }
                uint64_t par;
                if (!get_parent(cur, d, &par)) break;
                cur = par;
            }
            cell_out[i] = found;
            if (found < 0) { // keep the smallest failing row (linear_tree.rs:514-517)
                int64_t prev = bad.load();
                while ((prev < 0 || i < prev) && !bad.compare_exchange_weak(prev, i)) {
                }
            }
        }
    });
    return bad.load();
}

} // namespace bbfmm
