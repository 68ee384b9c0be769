// See ddm.hpp.  Follows DDMTree::new (domain_decomposition.rs:67-347) with deterministic, threaded
// loops: the median splits of a level are processed breadth first (VecDeque order), the per-leaf
// coarse-point and overlap selection runs in parallel over the leaves (it only reads the
// neighbours' internal points, which no leaf changes).
#include "ddm.hpp"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "../../include/ferreus_bbfmm_hip.h"
#include "parallel.hpp"

namespace bbfmm {
namespace {

struct Pts {
    const double *p;
    int64_t ld;
    int d;
    double at(int64_t i, int a) const { return p[a * ld + i]; }
};

void extents_of(const Pts &P, const int64_t *idx, int64_t np, std::vector<double> *e) { // utils.rs:196-228
    const int d = P.d;
    e->assign(static_cast<size_t>(2 * d), 0.0);
    for (int a = 0; a < d; ++a) (*e)[a] = (*e)[a + d] = P.at(idx[0], a);
    for (int64_t k = 0; k < np; ++k)
        for (int a = 0; a < d; ++a) {
            const double v = P.at(idx[k], a);
            if (v < (*e)[a]) (*e)[a] = v;
            if (v > (*e)[a + d]) (*e)[a + d] = v;
        }
}

// The same over np points given by contiguous coordinate arrays x[a][0..np).
void extents_of(const double *const *x, int d, int64_t np, double *e) {
    for (int a = 0; a < 2 * d && np <= 0; ++a) e[a] = 0.0; // (an empty half of a one-point domain: leaf_threshold 1, quota 1)
    if (np <= 0) return;
    for (int a = 0; a < d; ++a) {
        const double *xa = x[a];
        double lo = xa[0], hi = xa[0];
        for (int64_t k = 0; k < np; ++k) {
            lo = xa[k] < lo ? xa[k] : lo;
            hi = xa[k] > hi ? xa[k] : hi;
        }
        e[a] = lo;
        e[a + d] = hi;
    }
}

// farthest_point_sampling (common.rs:246-288) over n points given by their coordinates (x[a][i], a contiguous copy
// made once per leaf: the points of a leaf are scattered over the input); returns positions
std::vector<int64_t> farthest_point_sampling(const double *const *x, int d, int64_t n, int64_t wanted, int64_t seed) {
    std::vector<int64_t> sel;
    sel.reserve(static_cast<size_t>(wanted));
    // Branch-free passes (they vectorise, square root included): a selected point keeps the distance -1, which no
    // update lowers and no maximum picks -- the reference skips selected points in both loops.  A missing axis reads
    // zeros (adds an exact 0.0 to the sum of squares).
    std::vector<double> min_d(static_cast<size_t>(n), INFINITY), zeros(d < 3 ? static_cast<size_t>(n) : 0, 0.0);
    const double *x0 = x[0], *x1 = d > 1 ? x[1] : zeros.data(), *x2 = d > 2 ? x[2] : zeros.data();
    double *md = min_d.data();
    sel.push_back(seed);
    md[seed] = -1.0;
    for (int64_t k = 1; k < wanted; ++k) {
        const int64_t last = sel.back();
        const double l0 = x0[last], l1 = x1[last], l2 = x2[last];
        for (int64_t i = 0; i < n; ++i) {
            const double t0 = l0 - x0[i], t1 = l1 - x1[i], t2 = l2 - x2[i];
            double s = 0.0; // get_distance, utils.rs:263-284
            s += t0 * t0;
            s += t1 * t1;
            s += t2 * t2;
            const double dd = std::sqrt(s);
            md[i] = dd < md[i] ? dd : md[i];
        }
        double mx = -1.0;
        for (int64_t i = 0; i < n; ++i) mx = md[i] > mx ? md[i] : mx;
        int64_t far = 0; // the first point at the largest distance
        if (mx > -1.0)
            for (int64_t i = 0; i < n; ++i)
                if (md[i] == mx) {
                    far = i;
                    break;
                }
        sel.push_back(far);
        md[far] = -1.0;
    }
    return sel;
}

int64_t large_domain_points() { // BBFMM_DDM_LARGE_DOMAIN=<points> (read per build: the tests force both passes)
    if (const char *e = std::getenv("BBFMM_DDM_LARGE_DOMAIN")) {
        const long long v = std::atoll(e);
        if (v > 0) return static_cast<int64_t>(v);
    }
    return int64_t(1) << 18;
}

// One domain during the median splits: a range of the level's working arrays -- point indices and, carried along,
// the points' coordinates, so that every pass over a domain streams through contiguous memory (gathering the
// coordinates of a deep domain through its indices costs a cache miss per point and pass).
struct Work {
    const int64_t *idx;
    const double *x[3];
    int64_t *idx_out;
    double *x_out[3];
};

// Stable two-way partition of np points about the cut value v of axis `axis`: the points below v and the first
// `ties_left` points equal to it go to out[0, mid), the others to out[mid, np); returns the coordinate of the first
// point of the upper part that equals v (element `mid` of the stable order: keeps the sign of a zero), else v.
double partition_range(const Work &w, int d, int axis, int64_t k0, int64_t k1, double v, int64_t ties, int64_t lpos,
                       int64_t rpos, bool *has_mid) {
    const double *va = w.x[axis];
    double mid_coord = v;
    *has_mid = false;
    for (int64_t k = k0; k < k1; ++k) {
        const double xv = va[k];
        int64_t o;
        if (xv < v) {
            o = lpos++;
        } else if (xv == v && ties > 0) {
            o = lpos++;
            --ties;
        } else {
            if (xv == v && !*has_mid) {
                mid_coord = xv;
                *has_mid = true;
            }
            o = rpos++;
        }
        w.idx_out[o] = w.idx[k];
        for (int a = 0; a < d; ++a) w.x_out[a][o] = w.x[a][k];
    }
    return mid_coord;
}

// Median split of one LARGE domain on T host threads: the same two halves and cut coordinate as the serial pass in
// build_ddm_tree (the first generations of a level are a handful of domains of millions of points each).
// The cut value is selected through a monotone bucketing of the coordinate range: per-chunk histograms find the
// bucket that holds element `mid` of the sorted order, and only that bucket is selected from serially.
double split_large_domain(const Work &w, int d, int axis, double lo, double hi, int64_t np, int64_t mid, int T) {
    const int64_t chunk = (np + T - 1) / T;
    constexpr int B = 4096;
    const double *vals = w.x[axis];
    const double scale = hi > lo ? static_cast<double>(B) / (hi - lo) : 0.0;
    auto bucket = [&](double x) { // monotone in x
        const double t = (x - lo) * scale;
        return t >= static_cast<double>(B - 1) ? B - 1 : (t > 0.0 ? static_cast<int>(t) : 0);
    };
    std::vector<int64_t> hist(static_cast<size_t>(T) * B, 0);
    parallel_for(T, 1, [&](int64_t th) {
        int64_t *h = &hist[static_cast<size_t>(th) * B];
        for (int64_t k = th * chunk; k < std::min(np, (th + 1) * chunk); ++k) ++h[bucket(vals[k])];
    });
    int bsel = 0;
    int64_t before = 0;
    for (; bsel < B; ++bsel) {
        int64_t c = 0;
        for (int th = 0; th < T; ++th) c += hist[static_cast<size_t>(th) * B + bsel];
        if (before + c > mid) break;
        before += c;
    }
    std::vector<std::vector<double>> part(static_cast<size_t>(T));
    parallel_for(T, 1, [&](int64_t th) {
        for (int64_t k = th * chunk; k < std::min(np, (th + 1) * chunk); ++k)
            if (bucket(vals[k]) == bsel) part[static_cast<size_t>(th)].push_back(vals[k]);
    });
    std::vector<double> sel;
    for (auto &pv : part) sel.insert(sel.end(), pv.begin(), pv.end());
    std::nth_element(sel.begin(), sel.begin() + (mid - before), sel.end());
    const double v = sel[static_cast<size_t>(mid - before)];
    // per chunk: values below the cut, ties; the ties fill the lower half in position order
    std::vector<int64_t> nb(static_cast<size_t>(T), 0), ne(static_cast<size_t>(T), 0);
    parallel_for(T, 1, [&](int64_t th) {
        int64_t b = 0, e = 0;
        for (int64_t k = th * chunk; k < std::min(np, (th + 1) * chunk); ++k) {
            b += vals[k] < v;
            e += vals[k] == v;
        }
        nb[static_cast<size_t>(th)] = b;
        ne[static_cast<size_t>(th)] = e;
    });
    int64_t below = 0;
    for (int th = 0; th < T; ++th) below += nb[static_cast<size_t>(th)];
    int64_t ties_left = mid - below;
    std::vector<int64_t> tl(static_cast<size_t>(T), 0), loff(static_cast<size_t>(T) + 1, 0), roff(static_cast<size_t>(T) + 1, 0);
    for (int th = 0; th < T; ++th) {
        const int64_t len = std::max<int64_t>(0, std::min(np, (th + 1) * chunk) - th * chunk);
        tl[static_cast<size_t>(th)] = std::min(ne[static_cast<size_t>(th)], ties_left);
        ties_left -= tl[static_cast<size_t>(th)];
        const int64_t nl = nb[static_cast<size_t>(th)] + tl[static_cast<size_t>(th)];
        loff[static_cast<size_t>(th) + 1] = loff[static_cast<size_t>(th)] + nl;
        roff[static_cast<size_t>(th) + 1] = roff[static_cast<size_t>(th)] + (len - nl);
    }
    // (loff[T] == mid: the lower half fills out[0, mid), the upper half follows)
    std::vector<uint8_t> has_mid(static_cast<size_t>(T), 0);
    std::vector<double> mid_of(static_cast<size_t>(T), v);
    parallel_for(T, 1, [&](int64_t th) {
        const int64_t k0 = th * chunk, k1 = std::min(np, (th + 1) * chunk);
        if (k0 >= k1) return;
        bool hm = false;
        mid_of[static_cast<size_t>(th)] = partition_range(w, d, axis, k0, k1, v, tl[static_cast<size_t>(th)], loff[static_cast<size_t>(th)],
                                                          mid + roff[static_cast<size_t>(th)], &hm);
        has_mid[static_cast<size_t>(th)] = hm ? 1 : 0;
    });
    for (int th = 0; th < T; ++th)
        if (has_mid[static_cast<size_t>(th)]) return mid_of[static_cast<size_t>(th)];
    return v;
}

void extents_of_large(const double *const *x, int d, int64_t np, double *e, int T) { // extents_of, threaded
    const int64_t chunk = (np + T - 1) / T;
    std::vector<double> part(static_cast<size_t>(T) * 6, 0.0);
    std::vector<uint8_t> used(static_cast<size_t>(T), 0);
    parallel_for(T, 1, [&](int64_t th) {
        const int64_t b = th * chunk, en = std::min(np, (th + 1) * chunk);
        if (b >= en) return;
        const double *xs[3] = {x[0] + b, d > 1 ? x[1] + b : nullptr, d > 2 ? x[2] + b : nullptr};
        extents_of(xs, d, en - b, &part[static_cast<size_t>(th) * 6]);
        used[static_cast<size_t>(th)] = 1;
    });
    bool first = true;
    for (int th = 0; th < T; ++th) {
        if (!used[static_cast<size_t>(th)]) continue;
        const double *q = &part[static_cast<size_t>(th) * 6];
        for (int a = 0; a < d; ++a) {
            if (first || q[a] < e[a]) e[a] = q[a];
            if (first || q[a + d] > e[a + d]) e[a + d] = q[a + d];
        }
        first = false;
    }
}

} // namespace

int build_ddm_tree(const double *pts, int64_t n, int d, int64_t ld, const DdmParams &prm, DdmTree *out) {
    if (!pts || n < 1 || d < 1 || d > 3 || ld < n || prm.leaf_threshold < 1 || prm.coarse_threshold < 1 ||
        !(prm.coarse_ratio > 0.0) || prm.overlap_quota < 0.0)
        return BBFMM_BAD_ARGUMENT;
    const Pts P{pts, ld, d};
    const int64_t kLargeDomain = large_domain_points();
    out->d = d;
    out->levels.clear();
    std::vector<int64_t> active(static_cast<size_t>(n));
    std::iota(active.begin(), active.end(), int64_t(0));

    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what, size_t level_no) {
        if (!verbose) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm]   decomposition level %zu: %-28s %7.3f s\n", level_no, what,
                     std::chrono::duration<double>(t - t_last).count());
        t_last = t;
    };
    while (static_cast<int64_t>(active.size()) > prm.coarse_threshold) {
        DdmLevel level;
        level.point_indices = active;
        // recursive median splits along the longest axis of a domain's own points (:96-162)
        // The reference pops a FIFO queue: every domain of one generation is split before any of the
        // next, and a leaf pair is appended when its parent is processed.  The same order results
        // from splitting a generation in parallel and collecting the children in parent order.
        // A domain is a range of the level's working arrays (indices and coordinates): its halves are written to the
        // same range of a second set of arrays (lower half first), the two sets change roles per generation, and only
        // the leaves get index vectors of their own.  The leaves' coordinates stay behind in `lx`, leaf after leaf,
        // for the coarse-point and overlap selection below.
        struct Range {
            int64_t b, e;
            std::vector<double> extents;
        };
        const int64_t na = static_cast<int64_t>(active.size());
        std::vector<int64_t> ibuf[2] = {active, std::vector<int64_t>(static_cast<size_t>(na))};
        PodDoubles xbuf[2][3], lx[3];
        for (int a = 0; a < d; ++a) {
            xbuf[0][a].resize(static_cast<size_t>(na));
            xbuf[1][a].resize(static_cast<size_t>(na));
            lx[a].resize(static_cast<size_t>(na));
        }
        parallel_for_chunks(na, 65536, [&](int64_t k0, int64_t k1) {
            for (int a = 0; a < d; ++a)
                for (int64_t k = k0; k < k1; ++k) xbuf[0][a][k] = P.at(active[k], a);
        });
        int cur_set = 0;
        std::vector<Range> gen(1);
        gen[0].b = 0;
        gen[0].e = na;
        gen[0].extents.assign(static_cast<size_t>(2 * d), 0.0);
        {
            const double *xs[3] = {xbuf[0][0].data(), d > 1 ? xbuf[0][1].data() : nullptr, d > 2 ? xbuf[0][2].data() : nullptr};
            if (na >= kLargeDomain) extents_of_large(xs, d, na, gen[0].extents.data(), host_threads());
            else extents_of(xs, d, na, gen[0].extents.data());
        }
        std::vector<int64_t> leaf_loc; // first entry of a leaf's coordinates in lx
        while (!gen.empty()) {
            const int64_t ng = static_cast<int64_t>(gen.size());
            std::vector<Range> left(static_cast<size_t>(ng)), right(static_cast<size_t>(ng));
            std::vector<uint8_t> is_leaf(static_cast<size_t>(ng), 0);
            // (the first generations are a few domains of millions of points: the host threads are shared among them)
            const int t_inner = static_cast<int>(host_threads() / ng);
            const int nxt_set = cur_set ^ 1;
            parallel_for(ng, 1, [&](int64_t g) {
                const Range &cur = gen[g];
                const int64_t np = cur.e - cur.b;
                Work w;
                w.idx = ibuf[cur_set].data() + cur.b;
                w.idx_out = ibuf[nxt_set].data() + cur.b;
                for (int a = 0; a < 3; ++a) {
                    w.x[a] = a < d ? xbuf[cur_set][a].data() + cur.b : nullptr;
                    w.x_out[a] = a < d ? xbuf[nxt_set][a].data() + cur.b : nullptr;
                }
                const bool large = np >= kLargeDomain && t_inner >= 2;
                double ext[6];
                if (large) extents_of_large(w.x, d, np, ext, t_inner);
                else extents_of(w.x, d, np, ext);
                int axis = 0; // argmax (utils.rs:147-170): first value greater than the running maximum, from 0
                double best = 0.0;
                for (int a = 0; a < d; ++a)
                    if (ext[a + d] - ext[a] > best) {
                        best = ext[a + d] - ext[a];
                        axis = a;
                    }
                // The reference argsorts the coordinate (stable) and cuts at np/2 (:118-147); only the two halves
                // as index sets (then sorted) and the coordinate of the first element of the upper half are
                // used.  The lower half of a stable argsort = every point below the median value v plus the
                // first ties (in position order) that fill it up, so one selection and one ordered pass
                // give the same sets -- already sorted, since the domain's indices are ascending.
                const int64_t mid = np / 2;
                double mid_coord;
                if (large) {
                    mid_coord = split_large_domain(w, d, axis, ext[axis], ext[axis + d], np, mid, t_inner);
                } else {
                    const double *vals = w.x[axis];
                    double v;
                    {
                        std::vector<double> sel(vals, vals + np);
                        std::nth_element(sel.begin(), sel.begin() + mid, sel.end());
                        v = sel[mid];
                    }
                    int64_t below = 0;
                    for (int64_t k = 0; k < np; ++k) below += vals[k] < v;
                    bool hm;
                    mid_coord = partition_range(w, d, axis, 0, np, v, mid - below, 0, mid, &hm);
                }
                Range &l = left[g], &r = right[g];
                l.b = cur.b;
                l.e = r.b = cur.b + mid;
                r.e = cur.e;
                l.extents = cur.extents;
                l.extents[axis + d] = mid_coord;
                r.extents = cur.extents;
                r.extents[axis] = mid_coord;
                if (!(static_cast<double>(np) + static_cast<double>(np) * prm.overlap_quota >=
                      2.0 * static_cast<double>(prm.leaf_threshold))) { // :150-162
                    is_leaf[g] = 1;
                    for (int a = 0; a < d; ++a) std::copy(w.x_out[a], w.x_out[a] + np, lx[a].data() + cur.b);
                }
            });
            std::vector<Range> next_gen;
            std::vector<Range *> new_leaves;
            for (int64_t g = 0; g < ng; ++g) {
                if (is_leaf[g]) {
                    new_leaves.push_back(&left[g]);
                    new_leaves.push_back(&right[g]);
                } else {
                    next_gen.push_back(std::move(left[g]));
                    next_gen.push_back(std::move(right[g]));
                }
            }
            const size_t l0 = level.leaves.size();
            level.leaves.resize(l0 + new_leaves.size());
            leaf_loc.resize(l0 + new_leaves.size());
            parallel_for(static_cast<int64_t>(new_leaves.size()), 16, [&](int64_t q) {
                Range *h = new_leaves[static_cast<size_t>(q)];
                DdmDomain &dom = level.leaves[l0 + static_cast<size_t>(q)];
                dom.idx.assign(ibuf[nxt_set].data() + h->b, ibuf[nxt_set].data() + h->e);
                dom.internal.assign(dom.idx.size(), 1);
                dom.extents = std::move(h->extents);
                leaf_loc[l0 + static_cast<size_t>(q)] = h->b;
            });
            if (verbose && std::getenv("BBFMM_VERBOSE_DDM_GENERATIONS")) {
                const auto tg = std::chrono::steady_clock::now();
                std::fprintf(stderr, "[bbfmm]     generation of %lld domains (threads per large domain %d): %.3f s since the level began\n",
                             static_cast<long long>(ng), t_inner, std::chrono::duration<double>(tg - t_last).count());
            }
            gen.swap(next_gen);
            cur_set = nxt_set;
        }
        lap("median splits", out->levels.size());
        const int64_t nl = static_cast<int64_t>(level.leaves.size());
        const int64_t num_coarse = static_cast<int64_t>(
            std::ceil(std::ceil(static_cast<double>(active.size()) * prm.coarse_ratio) / static_cast<double>(nl))); // :165-168

        // per leaf: coarse points (farthest point sampling from the point closest to the centroid)
        // and the overlap taken from the neighbouring leaves' internal points (:183-307)
        std::vector<std::vector<int64_t>> internal(static_cast<size_t>(nl)), coarse(static_cast<size_t>(nl)),
            overlap(static_cast<size_t>(nl));
        for (int64_t i = 0; i < nl; ++i) internal[i] = level.leaves[i].idx; // all internal at this point
        // (the leaves' own coordinates lie leaf after leaf in lx: every leaf is read again by each of its ~26 neighbours)
        const std::vector<int64_t> &loc = leaf_loc;
        std::vector<int64_t> n_overlap(static_cast<size_t>(nl), 0);
        parallel_for(nl, 1, [&](int64_t i) {
            const DdmDomain &dom = level.leaves[i];
            const std::vector<int64_t> &in = internal[i];
            const int64_t ni = static_cast<int64_t>(in.size());
            const double *x[3] = {nullptr, nullptr, nullptr};
            for (int a = 0; a < d; ++a) x[a] = lx[a].data() + loc[i];
            double c[3] = {0, 0, 0}; // get_centroid (:350-359)
            for (int a = 0; a < d; ++a) {
                double s = 0.0;
                for (int64_t k = 0; k < ni; ++k) s += x[a][k];
                c[a] = s / static_cast<double>(ni);
            }
            int64_t center = 0; // argmin (utils.rs:116-143): first minimum
            double dmin = 0.0;
            for (int64_t k = 0; k < ni; ++k) {
                double s = 0.0;
                for (int a = 0; a < d; ++a) {
                    const double t = c[a] - x[a][k];
                    s += t * t;
                }
                const double dd = std::sqrt(s);
                if (k == 0 || dd < dmin) {
                    dmin = dd;
                    center = k;
                }
            }
            const int64_t sample = std::min(ni, num_coarse);
            for (int64_t pos : farthest_point_sampling(x, d, ni, sample, center)) coarse[i].push_back(in[pos]);
            std::sort(coarse[i].begin(), coarse[i].end());
            // neighbours: leaves whose (closed) box intersects this one, self excluded (rtree.rs:76-88); their
            // internal points with the distance to this leaf's box (:268-287)
            std::vector<int64_t> cand;
            std::vector<double> bd;
            for (int64_t j = 0; j < nl; ++j) {
                if (j == i) continue;
                bool hit = true;
                for (int a = 0; a < d && hit; ++a)
                    hit = level.leaves[j].extents[a] <= dom.extents[a + d] &&
                          level.leaves[j].extents[a + d] >= dom.extents[a];
                if (!hit) continue;
                cand.insert(cand.end(), internal[j].begin(), internal[j].end());
                const int64_t nj = static_cast<int64_t>(internal[j].size());
                const size_t b0 = bd.size();
                bd.resize(b0 + static_cast<size_t>(nj));
                double *bo = bd.data() + b0;
                for (int a = 0; a < d; ++a) { // axis by axis, the same order of additions as the point-by-point sum
                    const double *xa = lx[a].data() + loc[j];
                    const double lo = dom.extents[a], hi = dom.extents[a + d];
                    if (a == 0)
                        for (int64_t k = 0; k < nj; ++k) {
                            const double cl = std::max(std::min(xa[k], hi), lo);
                            bo[k] = 0.0 + (xa[k] - cl) * (xa[k] - cl);
                        }
                    else
                        for (int64_t k = 0; k < nj; ++k) {
                            const double cl = std::max(std::min(xa[k], hi), lo);
                            bo[k] += (xa[k] - cl) * (xa[k] - cl);
                        }
                }
                for (int64_t k = 0; k < nj; ++k) bo[k] = std::sqrt(bo[k]);
            }
            n_overlap[i] = static_cast<int64_t>(std::ceil(static_cast<double>(dom.idx.size() * 2) * prm.overlap_quota));
            // the first `take` of the stable argsort by distance (:289-296): order by (distance, position),
            // select that prefix, sort only it
            std::vector<int64_t> ord(cand.size());
            std::iota(ord.begin(), ord.end(), int64_t(0));
            const size_t take = std::min<size_t>(static_cast<size_t>(n_overlap[i]), cand.size());
            auto before = [&](int64_t x_, int64_t y_) { return bd[x_] < bd[y_] || (bd[x_] == bd[y_] && x_ < y_); };
            if (take < ord.size()) std::nth_element(ord.begin(), ord.begin() + take, ord.end(), before);
            std::sort(ord.begin(), ord.begin() + take, before);
            for (size_t k = 0; k < take; ++k) overlap[i].push_back(cand[ord[k]]);
        });
        lap("coarse points + overlap", out->levels.size());
        parallel_for(nl, 16, [&](int64_t i) {
            DdmDomain &dom = level.leaves[i];
            dom.idx.insert(dom.idx.end(), overlap[i].begin(), overlap[i].end());
            dom.internal.resize(dom.idx.size(), 0); // the overlap is never internal
        });
        // the next level's points: the leaves' coarse points in ascending order.  They are distinct (a point is
        // internal to one leaf), so marking them in a flag array over all points and compacting it sorts them.
        std::vector<int64_t> next;
        {
            int64_t total = 0;
            for (int64_t i = 0; i < nl; ++i) total += static_cast<int64_t>(coarse[i].size());
            std::vector<uint8_t> mark(static_cast<size_t>(n), 0);
            parallel_for(nl, 16, [&](int64_t i) {
                for (int64_t g : coarse[i]) mark[static_cast<size_t>(g)] = 1;
            });
            constexpr int64_t kBlk = 1 << 16;
            const int64_t nblk = (n + kBlk - 1) / kBlk;
            std::vector<int64_t> cnt(static_cast<size_t>(nblk) + 1, 0);
            parallel_for(nblk, 1, [&](int64_t b) {
                int64_t c = 0;
                for (int64_t g = b * kBlk; g < std::min(n, (b + 1) * kBlk); ++g) c += mark[static_cast<size_t>(g)];
                cnt[static_cast<size_t>(b) + 1] = c;
            });
            for (int64_t b = 0; b < nblk; ++b) cnt[static_cast<size_t>(b) + 1] += cnt[static_cast<size_t>(b)];
            if (cnt[static_cast<size_t>(nblk)] == total) {
                next.resize(static_cast<size_t>(total));
                parallel_for(nblk, 1, [&](int64_t b) {
                    int64_t o = cnt[static_cast<size_t>(b)];
                    for (int64_t g = b * kBlk; g < std::min(n, (b + 1) * kBlk); ++g)
                        if (mark[static_cast<size_t>(g)]) next[static_cast<size_t>(o++)] = g;
                });
            } else { // (repeated indices: cannot happen with disjoint leaves; keep them as a sort would)
                for (int64_t i = 0; i < nl; ++i) next.insert(next.end(), coarse[i].begin(), coarse[i].end());
                std::sort(next.begin(), next.end());
            }
        }
        lap("merge", out->levels.size());
        out->levels.push_back(std::move(level));
        active.swap(next);
    }
    DdmLevel coarse_level; // one domain over the remaining points (:320-343)
    coarse_level.point_indices = active;
    DdmDomain cd;
    cd.idx = active;
    cd.internal.assign(active.size(), 1);
    extents_of(P, cd.idx.data(), static_cast<int64_t>(cd.idx.size()), &cd.extents);
    coarse_level.leaves.push_back(std::move(cd));
    out->levels.push_back(std::move(coarse_level));
    return BBFMM_OK;
}

} // namespace bbfmm
