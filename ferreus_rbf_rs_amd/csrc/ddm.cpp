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

void extents_of(const Pts &P, const std::vector<int64_t> &idx, std::vector<double> *e) { // utils.rs:196-228
    const int d = P.d;
    e->assign(static_cast<size_t>(2 * d), 0.0);
    for (int a = 0; a < d; ++a) (*e)[a] = (*e)[a + d] = P.at(idx[0], a);
    for (int64_t i : idx)
        for (int a = 0; a < d; ++a) {
            const double v = P.at(i, a);
            if (v < (*e)[a]) (*e)[a] = v;
            if (v > (*e)[a + d]) (*e)[a + d] = v;
        }
}

double dist(const Pts &P, int64_t i, int64_t j) { // get_distance, utils.rs:263-284
    double s = 0.0;
    for (int a = 0; a < P.d; ++a) {
        const double t = P.at(i, a) - P.at(j, a);
        s += t * t;
    }
    return std::sqrt(s);
}

// farthest_point_sampling (common.rs:246-288) over the points `ids`; returns positions in ids
std::vector<int64_t> farthest_point_sampling(const Pts &P, const std::vector<int64_t> &ids, int64_t wanted,
                                             int64_t seed) {
    const int64_t n = static_cast<int64_t>(ids.size());
    std::vector<int64_t> sel;
    sel.reserve(static_cast<size_t>(wanted));
    std::vector<uint8_t> is_sel(static_cast<size_t>(n), 0);
    std::vector<double> min_d(static_cast<size_t>(n), INFINITY);
    sel.push_back(seed);
    is_sel[seed] = 1;
    for (int64_t k = 1; k < wanted; ++k) {
        const int64_t last = sel.back();
        for (int64_t i = 0; i < n; ++i) {
            if (is_sel[i]) continue;
            const double dd = dist(P, ids[last], ids[i]);
            if (dd < min_d[i]) min_d[i] = dd;
        }
        int64_t far = 0;
        double mx = -1.0;
        for (int64_t i = 0; i < n; ++i)
            if (!is_sel[i] && min_d[i] > mx) {
                mx = min_d[i];
                far = i;
            }
        sel.push_back(far);
        is_sel[far] = 1;
    }
    return sel;
}

} // namespace

int build_ddm_tree(const double *pts, int64_t n, int d, int64_t ld, const DdmParams &prm, DdmTree *out) {
    if (!pts || n < 1 || d < 1 || d > 3 || ld < n || prm.leaf_threshold < 1 || prm.coarse_threshold < 1 ||
        !(prm.coarse_ratio > 0.0) || prm.overlap_quota < 0.0)
        return BBFMM_BAD_ARGUMENT;
    const Pts P{pts, ld, d};
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
        DdmDomain root;
        root.idx = active;
        extents_of(P, root.idx, &root.extents);
        // The reference pops a FIFO queue: every domain of one generation is split before any of the
        // next, and a leaf pair is appended when its parent is processed.  The same order results
        // from splitting a generation in parallel and collecting the children in parent order.
        std::vector<DdmDomain> gen;
        gen.push_back(std::move(root));
        while (!gen.empty()) {
            const int64_t ng = static_cast<int64_t>(gen.size());
            std::vector<DdmDomain> left(static_cast<size_t>(ng)), right(static_cast<size_t>(ng));
            std::vector<uint8_t> is_leaf(static_cast<size_t>(ng), 0);
            parallel_for(ng, 1, [&](int64_t g) {
                DdmDomain &cur = gen[g];
                const int64_t np = static_cast<int64_t>(cur.idx.size());
                std::vector<double> ext;
                extents_of(P, cur.idx, &ext);
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
                // give the same sets -- already sorted, since cur.idx is ascending.
                const int64_t mid = np / 2;
                std::vector<double> vals(static_cast<size_t>(np));
                for (int64_t k = 0; k < np; ++k) vals[k] = P.at(cur.idx[k], axis);
                double v;
                {
                    std::vector<double> sel(vals);
                    std::nth_element(sel.begin(), sel.begin() + mid, sel.end());
                    v = sel[mid];
                }
                int64_t below = 0;
                for (int64_t k = 0; k < np; ++k) below += vals[k] < v;
                int64_t ties_left = mid - below; // ties that still belong to the lower half
                DdmDomain &l = left[g], &r = right[g];
                l.idx.reserve(static_cast<size_t>(mid));
                r.idx.reserve(static_cast<size_t>(np - mid));
                double mid_coord = v;
                bool have_mid = false;
                for (int64_t k = 0; k < np; ++k) {
                    const double x = vals[k];
                    if (x < v) {
                        l.idx.push_back(cur.idx[k]);
                    } else if (x == v && ties_left > 0) {
                        l.idx.push_back(cur.idx[k]);
                        --ties_left;
                    } else {
                        if (x == v && !have_mid) { // element `mid` of the stable order (keeps the sign of a zero)
                            mid_coord = x;
                            have_mid = true;
                        }
                        r.idx.push_back(cur.idx[k]);
                    }
                }
                l.extents = cur.extents;
                l.extents[axis + d] = mid_coord;
                r.extents = cur.extents;
                r.extents[axis] = mid_coord;
                if (!(static_cast<double>(np) + static_cast<double>(np) * prm.overlap_quota >=
                      2.0 * static_cast<double>(prm.leaf_threshold))) { // :150-162
                    is_leaf[g] = 1;
                    l.internal.assign(l.idx.size(), 1);
                    r.internal.assign(r.idx.size(), 1);
                }
                std::vector<int64_t>().swap(cur.idx);
            });
            std::vector<DdmDomain> next_gen;
            for (int64_t g = 0; g < ng; ++g) {
                if (is_leaf[g]) {
                    level.leaves.push_back(std::move(left[g]));
                    level.leaves.push_back(std::move(right[g]));
                } else {
                    next_gen.push_back(std::move(left[g]));
                    next_gen.push_back(std::move(right[g]));
                }
            }
            gen.swap(next_gen);
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
        std::vector<int64_t> n_overlap(static_cast<size_t>(nl), 0);
        parallel_for(nl, 1, [&](int64_t i) {
            const DdmDomain &dom = level.leaves[i];
            const std::vector<int64_t> &in = internal[i];
            const int64_t ni = static_cast<int64_t>(in.size());
            double c[3] = {0, 0, 0}; // get_centroid (:350-359)
            for (int a = 0; a < d; ++a) {
                double s = 0.0;
                for (int64_t k = 0; k < ni; ++k) s += P.at(in[k], a);
                c[a] = s / static_cast<double>(ni);
            }
            int64_t center = 0; // argmin (utils.rs:116-143): first minimum
            double dmin = 0.0;
            for (int64_t k = 0; k < ni; ++k) {
                double s = 0.0;
                for (int a = 0; a < d; ++a) {
                    const double t = c[a] - P.at(in[k], a);
                    s += t * t;
                }
                const double dd = std::sqrt(s);
                if (k == 0 || dd < dmin) {
                    dmin = dd;
                    center = k;
                }
            }
            const int64_t sample = std::min(ni, num_coarse);
            for (int64_t pos : farthest_point_sampling(P, in, sample, center)) coarse[i].push_back(in[pos]);
            std::sort(coarse[i].begin(), coarse[i].end());
            // neighbours: leaves whose (closed) box intersects this one, self excluded (rtree.rs:76-88)
            std::vector<int64_t> cand;
            for (int64_t j = 0; j < nl; ++j) {
                if (j == i) continue;
                bool hit = true;
                for (int a = 0; a < d && hit; ++a)
                    hit = level.leaves[j].extents[a] <= dom.extents[a + d] &&
                          level.leaves[j].extents[a + d] >= dom.extents[a];
                if (!hit) continue;
                cand.insert(cand.end(), internal[j].begin(), internal[j].end());
            }
            n_overlap[i] = static_cast<int64_t>(std::ceil(static_cast<double>(dom.idx.size() * 2) * prm.overlap_quota));
            std::vector<double> bd(cand.size());
            for (size_t k = 0; k < cand.size(); ++k) { // distance to the box (:268-287)
                double s = 0.0;
                for (int a = 0; a < d; ++a) {
                    const double x = P.at(cand[k], a);
                    const double cl = std::max(std::min(x, dom.extents[a + d]), dom.extents[a]);
                    s += (x - cl) * (x - cl);
                }
                bd[k] = std::sqrt(s);
            }
            // the first `take` of the stable argsort by distance (:289-296): order by (distance, position),
            // select that prefix, sort only it
            std::vector<int64_t> ord(cand.size());
            std::iota(ord.begin(), ord.end(), int64_t(0));
            const size_t take = std::min<size_t>(static_cast<size_t>(n_overlap[i]), cand.size());
            auto before = [&](int64_t x, int64_t y) { return bd[x] < bd[y] || (bd[x] == bd[y] && x < y); };
            if (take < ord.size()) std::nth_element(ord.begin(), ord.begin() + take, ord.end(), before);
            std::sort(ord.begin(), ord.begin() + take, before);
            for (size_t k = 0; k < take; ++k) overlap[i].push_back(cand[ord[k]]);
        });
        lap("coarse points + overlap", out->levels.size());
        std::vector<int64_t> next;
        for (int64_t i = 0; i < nl; ++i) {
            DdmDomain &dom = level.leaves[i];
            dom.idx.insert(dom.idx.end(), overlap[i].begin(), overlap[i].end());
            dom.internal.resize(dom.idx.size(), 0); // the overlap is never internal
            next.insert(next.end(), coarse[i].begin(), coarse[i].end());
        }
        std::sort(next.begin(), next.end());
        lap("merge", out->levels.size());
        out->levels.push_back(std::move(level));
        active.swap(next);
    }
    DdmLevel coarse_level; // one domain over the remaining points (:320-343)
    coarse_level.point_indices = active;
    DdmDomain cd;
    cd.idx = active;
    cd.internal.assign(active.size(), 1);
    extents_of(P, cd.idx, &cd.extents);
    coarse_level.leaves.push_back(std::move(cd));
    out->levels.push_back(std::move(coarse_level));
    return BBFMM_OK;
}

} // namespace bbfmm
