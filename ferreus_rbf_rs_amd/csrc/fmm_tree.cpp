// FmmTree host orchestration.  See fmm_tree.hpp.
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

int FmmTree::fail(int code, const std::string &msg) {
    err_ = msg;
    return code;
}
int FmmTree::hip_fail(hipError_t e, const char *what) {
    err_ = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return BBFMM_DEVICE_ERROR;
}

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
void FmmTree::arena_begin() {
    arena_active_ = true;
    arena_used_ = 0;
    arena_need_ = 0;
}
int FmmTree::arena_end() { // every borrowed buffer has been released by now
    arena_active_ = false;
    if (arena_need_ > arena_.n) {
        dfree(&arena_);
        CHK(dalloc(&arena_, arena_need_ + arena_need_ / 4));
    }
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

FmmTree::~FmmTree() {
    if (!host_only_ && stream_) (void)hipStreamSynchronize(stream_);
    for (void *p : owned_) (void)hipFree(p);
    owned_.clear();
    for (const PendingPhase &pp : pending_) {
        (void)hipEventDestroy(pp.e0);
        (void)hipEventDestroy(pp.e1);
    }
    for (hipEvent_t e : event_pool_) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_out_) (void)hipEventDestroy(e);
    if (h_pin_) (void)hipHostFree(h_pin_);
    free_dev_tree_points(&dev_points_);
    if (ev_pack_) (void)hipEventDestroy(ev_pack_);
    if (ev_comm_) (void)hipEventDestroy(ev_comm_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

// Phase timing: hipEvent pairs recorded on the launch stream without synchronising, so the
// kernels of a timed region run back to back; collect_phase_times() resolves them later.
hipEvent_t FmmTree::get_event() {
    if (!event_pool_.empty()) {
        hipEvent_t e = event_pool_.back();
        event_pool_.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void FmmTree::phase_begin() {
    if (!profiling_) return;
    pending_begin_ = get_event();
    (void)hipEventRecord(pending_begin_, stream_);
}
void FmmTree::phase_end(int ph) {
    if (!profiling_ || !pending_begin_) return;
    hipEvent_t e1 = get_event();
    (void)hipEventRecord(e1, stream_);
    pending_.push_back(PendingPhase{ph, pending_begin_, e1});
    pending_begin_ = nullptr;
}
void FmmTree::collect_phase_times() {
    if (pending_.empty()) return;
    (void)hipStreamSynchronize(stream_);
    for (const PendingPhase &pp : pending_) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pp.e0, pp.e1) == hipSuccess) {
            phase_ms_[pp.phase] += ms;
            phase_count_[pp.phase] += 1;
        }
        event_pool_.push_back(pp.e0);
        event_pool_.push_back(pp.e1);
    }
    pending_.clear();
}

// Merge the point ranges of `cells` (those that hold points) into sorted runs.
static void merged_runs(const HostTree &t, const int32_t *cells, int64_t count, std::vector<int32_t> *runs) {
    std::vector<std::pair<int64_t, int64_t>> r;
    for (int64_t i = 0; i < count; ++i) {
        const int32_t c = cells[i];
        if (t.pt_end[c] > t.pt_begin[c]) r.emplace_back(t.pt_begin[c], t.pt_end[c]);
    }
    std::sort(r.begin(), r.end());
    size_t i = 0;
    while (i < r.size()) {
        int64_t b = r[i].first, e = r[i].second;
        size_t j = i + 1;
        while (j < r.size() && r[j].first == e) {
            e = r[j].second;
            ++j;
        }
        runs->push_back(static_cast<int32_t>(b));
        runs->push_back(static_cast<int32_t>(e));
        i = j;
    }
}

// BBFMM_FLAG_DIRECT_SMALL_W_LEAVES (extension): every W-list entry (B, w) whose cell w is a leaf with at most `n_nodes`
// points moves to the U lists of both cells, and B leaves the X list of w (X = W^T, linear_tree.rs:388-392).  Returns the
// number of entries moved.  Lists stay sorted by cell index inside a row.
static int64_t direct_small_w_leaves(HostTree *t, int64_t n_nodes) {
    const int64_t C = t->n_cells();
    std::vector<std::pair<int32_t, int32_t>> moved; // (B, w)
    Csr w2;
    w2.ptr.assign(static_cast<size_t>(C) + 1, 0);
    w2.idx.reserve(t->w.idx.size());
    for (int64_t B = 0; B < C; ++B) {
        for (int64_t q = t->w.ptr[B]; q < t->w.ptr[B + 1]; ++q) {
            const int32_t w = t->w.idx[q];
            const int64_t npts = t->pt_end[w] - t->pt_begin[w];
            if (t->is_leaf[w] && npts <= n_nodes) moved.emplace_back(static_cast<int32_t>(B), w);
            else w2.idx.push_back(w);
        }
        w2.ptr[B + 1] = static_cast<int64_t>(w2.idx.size());
    }
    if (moved.empty()) return 0;
    t->w = std::move(w2);
    // X = transpose of the remaining W (rows ascending in B because B ascends in the outer loop)
    Csr x2;
    x2.ptr.assign(static_cast<size_t>(C) + 1, 0);
    for (int32_t w : t->w.idx) ++x2.ptr[static_cast<size_t>(w) + 1];
    for (int64_t c = 0; c < C; ++c) x2.ptr[c + 1] += x2.ptr[c];
    x2.idx.resize(t->w.idx.size());
    {
        std::vector<int64_t> fill(x2.ptr.begin(), x2.ptr.end() - 1);
        for (int64_t B = 0; B < C; ++B)
            for (int64_t q = t->w.ptr[B]; q < t->w.ptr[B + 1]; ++q) x2.idx[fill[t->w.idx[q]]++] = static_cast<int32_t>(B);
    }
    t->x = std::move(x2);
    // U: both directions
    std::vector<std::pair<int32_t, int32_t>> add;
    add.reserve(2 * moved.size());
    for (const auto &m : moved) {
        add.emplace_back(m.first, m.second);
        add.emplace_back(m.second, m.first);
    }
    std::sort(add.begin(), add.end());
    add.erase(std::unique(add.begin(), add.end()), add.end());
    Csr u2;
    u2.ptr.assign(static_cast<size_t>(C) + 1, 0);
    u2.idx.reserve(t->u.idx.size() + add.size());
    size_t a = 0;
    for (int64_t c = 0; c < C; ++c) {
        const size_t row0 = u2.idx.size();
        for (int64_t q = t->u.ptr[c]; q < t->u.ptr[c + 1]; ++q) u2.idx.push_back(t->u.idx[q]);
        const size_t a0 = a;
        while (a < add.size() && add[a].first == c) u2.idx.push_back(add[a++].second);
        if (a > a0) {
            std::sort(u2.idx.begin() + row0, u2.idx.end());
            u2.idx.erase(std::unique(u2.idx.begin() + row0, u2.idx.end()), u2.idx.end());
        }
        u2.ptr[c + 1] = static_cast<int64_t>(u2.idx.size());
    }
    t->u = std::move(u2);
    return static_cast<int64_t>(moved.size());
}

int FmmTree::create(const double *pts, int64_t n, int d, int64_t ld, int order, int kernel_type, double base_range,
                    double total_sill, bool adaptive, bool sparse, const double *extents,
                    const bbfmm_params *params, uint32_t flags) {
    if (d < 1 || d > 3) // bbfmm.rs:293-298
        return fail(BBFMM_BAD_ARGUMENT, "Unsupported number of dimensions: " + std::to_string(d));
    if (!pts || n < 1 || ld < n) return fail(BBFMM_BAD_ARGUMENT, "source_points must hold at least one row");
    if (n >= (int64_t(1) << 31)) return fail(BBFMM_BAD_ARGUMENT, "more than 2^31-1 source points");
    if (!kernel_id_valid(kernel_type)) return fail(BBFMM_BAD_ARGUMENT, "unknown kernel_type");
    if (order < 2 || order > kMaxOrder)
        return fail(BBFMM_BAD_ARGUMENT, "interpolation_order must be in [2, " + std::to_string(kMaxOrder) + "]");
    if (!(base_range > 0.0)) return fail(BBFMM_BAD_ARGUMENT, "base_range must be positive"); // kernel_helpers.rs:69
    if (!l2p_order_supported(order, d)) return fail(BBFMM_UNSUPPORTED, "interpolation_order is not instantiated on the device");
    host_only_ = (flags & BBFMM_FLAG_HOST_ONLY) != 0;
    shared_basis_ = (flags & BBFMM_FLAG_M2L_SHARED_BASIS) != 0;
    deterministic_ = (flags & BBFMM_FLAG_DETERMINISTIC) != 0;
    if (shared_basis_ && host_only_) return fail(BBFMM_BAD_ARGUMENT, "BBFMM_FLAG_M2L_SHARED_BASIS needs a device");
    order_ = order;
    d_ = d;
    kernel_ = make_kernel_spec(kernel_type, base_range, total_sill);
    if (params)
        params_ = *params;
    else
        bbfmm_params_defaults(order, &params_);
    if (params_.max_points_per_cell < 1) return fail(BBFMM_BAD_ARGUMENT, "max_points_per_cell must be >= 1");
    if (params_.compression_type < 0 || params_.compression_type > 2)
        return fail(BBFMM_BAD_ARGUMENT, "unknown compression_type");

    StageTimer timer;
    pts_.resize(static_cast<size_t>(n) * d);
    for (int a = 0; a < d; ++a)
        parallel_for_chunks(n, int64_t(1) << 18, [&](int64_t b, int64_t e) {
            std::memcpy(&pts_[static_cast<size_t>(a) * n + b], pts + a * ld + b, static_cast<size_t>(e - b) * sizeof(double));
        });

    double ext[6];
    if (extents) {
        std::copy(extents, extents + 2 * d, ext);
    } else { // utils.rs:13-46 (min / max per axis: chunk results combined in order)
        constexpr int64_t kChunkE = int64_t(1) << 18;
        const int64_t nch = (n + kChunkE - 1) / kChunkE;
        std::vector<double> lo_c(static_cast<size_t>(nch)), hi_c(static_cast<size_t>(nch));
        for (int a = 0; a < d; ++a) {
            const double *col = &pts_[static_cast<size_t>(a) * n];
            parallel_for_chunks(n, kChunkE, [&](int64_t b, int64_t e) {
                for (int64_t c0 = b; c0 < e; c0 += kChunkE) {
                    double lo = col[c0], hi = col[c0];
                    for (int64_t i = c0 + 1; i < std::min(e, c0 + kChunkE); ++i) {
                        if (col[i] < lo) lo = col[i];
                        if (col[i] > hi) hi = col[i];
                    }
                    lo_c[static_cast<size_t>(c0 / kChunkE)] = lo;
                    hi_c[static_cast<size_t>(c0 / kChunkE)] = hi;
                }
            });
            double lo = lo_c[0], hi = hi_c[0];
            for (int64_t c = 1; c < nch; ++c) {
                if (lo_c[static_cast<size_t>(c)] < lo) lo = lo_c[static_cast<size_t>(c)];
                if (hi_c[static_cast<size_t>(c)] > hi) hi = hi_c[static_cast<size_t>(c)];
            }
            ext[a] = lo;
            ext[d + a] = hi;
        }
    }
    timer.lap("copy points, extents");
    double center[3] = {0, 0, 0}, radius = 0;
    calculate_tree_center_and_radius(ext, d, center, &radius);
    if (!(radius > 0.0) || !std::isfinite(radius)) return fail(BBFMM_BAD_ARGUMENT, "degenerate or non-finite extents");

    if (!host_only_) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(BBFMM_DEVICE_ERROR, "no HIP device available (the BBFMM passes have no CPU fallback)");
        {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                prop.multiProcessorCount > 0)
                n_cu_ = prop.multiProcessorCount;
            device_ = dev;
        }
        HIPCHK(hipStreamCreate(&stream_));
        HIPCHK(hipEventCreateWithFlags(&ev_pack_, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_comm_, hipEventDisableTiming));
        timer.lap("device, streams");
    }
    // The subdivision (Morton codes, sort, cells, per-leaf point lists) runs on the device when there is one and
    // every point lies inside the root box; the host build (tree.cpp) is the bit-exact checker and the fallback.
    static const bool tree_on_device = [] {
        const char *e = std::getenv("BBFMM_TREE_DEVICE");
        return !e || std::atoi(e) != 0;
    }();
    tree_built_on_device_ = false;
    if (!host_only_ && tree_on_device) {
        std::vector<BuildCell> cells;
        const int brc = build_tree_cells_device(pts_.data(), n, n, d, center, radius, params_.max_points_per_cell, !sparse,
                                                adaptive, &tree_, &cells, &dev_points_, stream_);
        if (brc < 0) return hip_fail(static_cast<hipError_t>(-brc), "device tree build");
        if (brc == 0) {
            timer.lap("tree: subdivision (device)");
            finish_tree(cells, &tree_, false);
            timer.lap("tree: numbering (host)");
            const int lrc = build_lists_device(&tree_, stream_);
            if (lrc < 0) return hip_fail(static_cast<hipError_t>(-lrc), "device interaction lists");
            if (lrc == 1) build_lists_host(&tree_);
            timer.lap(lrc == 0 ? "tree: lists (device)" : "tree: lists (host)");
            tree_built_on_device_ = true;
        }
    }
    if (!tree_built_on_device_)
        build_tree(pts_.data(), n, n, d, center, radius, params_.max_points_per_cell, !sparse, adaptive, &tree_);
    timer.lap("tree + interaction lists");
    if ((flags & BBFMM_FLAG_DIRECT_SMALL_W_LEAVES) && adaptive) {
        int64_t n_nodes = 1;
        for (int a = 0; a < d; ++a) n_nodes *= order;
        const int64_t moved = direct_small_w_leaves(&tree_, n_nodes);
        if (std::getenv("BBFMM_VERBOSE"))
            std::fprintf(stderr, "[bbfmm] direct small W leaves: %lld W-list entries became near field\n", static_cast<long long>(moved));
        timer.lap("W leaves -> near field");
    }
    precompute_operators(order, d, radius, tree_.depth, kernel_, params_.compression_type, params_.epsilon, &ops_);
    timer.lap("operators (ACA/SVD)");

    // ---- host-side work lists
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    level_cells_.assign(t.depth + 1, {});
    m2m_parents_.assign(t.depth + 1, {});
    for (int64_t c = 0; c < C; ++c) {
        level_cells_[t.level[c]].push_back(static_cast<int32_t>(c));
        if (t.children.ptr[c + 1] > t.children.ptr[c]) m2m_parents_[t.level[c]].push_back(static_cast<int32_t>(c));
        if (t.is_leaf[c] && t.pt_end[c] > t.pt_begin[c]) src_leaves_.push_back(static_cast<int32_t>(c));
    }
    // leaves in sorted-point (DFS Morton) order: spatially coherent job order
    std::sort(src_leaves_.begin(), src_leaves_.end(),
              [&](int32_t a, int32_t b) { return t.pt_begin[a] < t.pt_begin[b]; });
    u_runs_.ptr.assign(C + 1, 0);
    x_runs_.ptr.assign(C + 1, 0);
    { // per chunk of cells into local buffers (threads), concatenated in order
        constexpr int64_t kChunkR = 4096;
        const int64_t nch = (C + kChunkR - 1) / kChunkR;
        std::vector<std::vector<int32_t>> ub(static_cast<size_t>(nch)), xb(static_cast<size_t>(nch));
        parallel_for_chunks(C, kChunkR, [&](int64_t lo, int64_t hi) {
            for (int64_t c0 = lo; c0 < hi; c0 += kChunkR) {
                auto &uu = ub[static_cast<size_t>(c0 / kChunkR)];
                auto &xx = xb[static_cast<size_t>(c0 / kChunkR)];
                for (int64_t c = c0; c < std::min(hi, c0 + kChunkR); ++c) {
                    const size_t u0 = uu.size(), x0 = xx.size();
                    merged_runs(t, t.u.idx.data() + t.u.ptr[c], t.u.ptr[c + 1] - t.u.ptr[c], &uu);
                    merged_runs(t, t.x.idx.data() + t.x.ptr[c], t.x.ptr[c + 1] - t.x.ptr[c], &xx);
                    u_runs_.ptr[c + 1] = static_cast<int64_t>((uu.size() - u0) / 2); // counts, scanned below
                    x_runs_.ptr[c + 1] = static_cast<int64_t>((xx.size() - x0) / 2);
                }
            }
        });
        for (int64_t c = 0; c < C; ++c) {
            if (x_runs_.ptr[c + 1] > 0) x_cells_.push_back(static_cast<int32_t>(c));
            u_runs_.ptr[c + 1] += u_runs_.ptr[c];
            x_runs_.ptr[c + 1] += x_runs_.ptr[c];
        }
        u_runs_.idx.resize(static_cast<size_t>(2 * u_runs_.ptr[C]));
        x_runs_.idx.resize(static_cast<size_t>(2 * x_runs_.ptr[C]));
        parallel_for(nch, 1, [&](int64_t ch) {
            const int64_t c0 = ch * kChunkR;
            std::copy(ub[static_cast<size_t>(ch)].begin(), ub[static_cast<size_t>(ch)].end(), u_runs_.idx.begin() + 2 * u_runs_.ptr[c0]);
            std::copy(xb[static_cast<size_t>(ch)].begin(), xb[static_cast<size_t>(ch)].end(), x_runs_.idx.begin() + 2 * x_runs_.ptr[c0]);
        });
    }
    part_rows_.clear();
    timer.lap("run lists");

    CHK(build_m2l_tables());
    timer.lap("stacked M2L tables");
    if (!host_only_) {
        CHK(upload());
        CHK(build_source_target_set());
        HIPCHK(hipStreamSynchronize(stream_));
        timer.lap("upload");
    }
    return BBFMM_OK;
}

// Stacked operators of one (level, class): VtAll (n_pad x r_pad16) and UAll (k_pad x n_pad).
void FmmTree::fill_m2l_operator_arrays(const HostM2lClass &hc, std::vector<double> *vt_all,
                                       std::vector<double> *u_all) const {
    const int n_pad = round_up(ops_.n, 32);
    vt_all->resize(static_cast<size_t>(n_pad) * hc.r_pad16);
    u_all->resize(static_cast<size_t>(hc.k_pad) * n_pad);
    fill_m2l_operator_arrays(hc, vt_all->data(), u_all->data());
}

// vt_all: n_pad x r_pad16, u_all: k_pad x n_pad (both overwritten, padding zeroed)
void FmmTree::fill_m2l_operator_arrays(const HostM2lClass &hc, double *vt_all, double *u_all) const {
    const int n = ops_.n, n_pad = round_up(n, 32);
    const bool compressed = ops_.compression != kCompressionNone;
    const auto &lops = ops_.m2l[hc.level];
    auto zero = [](double *p, size_t len) {
        parallel_for_chunks(static_cast<int64_t>(len), int64_t(1) << 18, [&](int64_t b, int64_t e) {
            std::memset(p + b, 0, static_cast<size_t>(e - b) * sizeof(double));
        });
    };
    zero(vt_all, static_cast<size_t>(n_pad) * hc.r_pad16);
    zero(u_all, static_cast<size_t>(hc.k_pad) * n_pad);
    struct RowSrc {
        const M2lOperator *op;
        const int32_t *inv;
        int first_row;
    };
    std::vector<RowSrc> row_src;
    int row = 0;
    for (int tv : hc.src_tv) {
        const M2lOperator &op = lops[ops_.ref_lookup[tv]];
        row_src.push_back(RowSrc{&op, &ops_.invperm[static_cast<size_t>(ops_.perm_lookup[tv]) * n], row});
        row += round_up(op.rank, 2);
    }
    // c[kk] = sum_m Vt[kk][invperm[m]] * M_V[m]   (bbfmm.rs:924-930 folded)
    parallel_for(n, 8, [&](int64_t m) {
        double *dst = vt_all + static_cast<size_t>(m) * hc.r_pad16;
        for (const RowSrc &rs : row_src) {
            const int r = rs.op->rank;
            const int im = rs.inv[m];
            if (compressed) {
                const double *src = &rs.op->vt[static_cast<size_t>(im) * r];
                for (int kk = 0; kk < r; ++kk) dst[rs.first_row + kk] = src[kk];
            } else {
                dst[rs.first_row + im] = 1.0;
            }
        }
    });
    // L_B[i] += sum_kk U[invperm[i]][kk] * c[kk]   (bbfmm.rs:975-981 folded)
    parallel_for(static_cast<int64_t>(hc.tgt_tv.size()), 1, [&](int64_t pos) {
        const int tv = hc.tgt_tv[pos];
        const M2lOperator &op = lops[ops_.ref_lookup[tv]];
        const int32_t *inv = &ops_.invperm[static_cast<size_t>(ops_.perm_lookup[tv]) * n];
        for (int kk = 0; kk < op.rank; ++kk) {
            double *dst = u_all + static_cast<size_t>(hc.tgt_off[pos] + kk) * n_pad;
            const double *ucol = &op.u[static_cast<size_t>(kk) * n];
            for (int i = 0; i < n; ++i) dst[i] = ucol[inv[i]];
        }
    });
}

// ------------------------------------------------------------------ M2L tables
// Folds the reference's symmetry permutations (bbfmm.rs:910-931,964-982) into stacked
// per-octant-class operators; see device.hip "M2L".
// One workgroup per CU runs at a time, so a launch of T equal tiles takes ceil(T / CUs) rounds.
// When the last round is at most half full its tiles are halved (a workgroup whose upper four
// waves hold no cells runs one wave per SIMD and takes about half the time): the tail costs half a
// round instead of a whole one.
static void split_tile_tail(std::vector<M2lTileDesc> *tiles, int n_cu) {
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

int FmmTree::build_m2l_tables() {
    const HostTree &t = tree_;
    const int d = d_, n = ops_.n;
    const int ncls = 1 << d, nvec = ops_.n_vec;
    const bool compressed = ops_.compression != kCompressionNone;
    m2l_host_.clear();
    m2l_variants_.clear();
    m2l_tiles1_h_.clear();
    m2l_tile_idx1_h_.clear();
    m2l_classes_h_.clear();
    m2l_tiles_h_.clear();
    m2l_qlist_h_.clear();
    m2l_batches_.clear();
    m2l_batch_of_class_.clear();
    m2l_group_ops_.clear();
    cbuf_batch_len_ = 0;
    cbuf_total_len_ = 0;
    m2l_flops_k1_ = 0;
    m2l_flops_level_.clear();
    if (t.depth < 2) return BBFMM_OK;
    // budget of the intermediate: a sixteenth of the device's memory (18 GiB on a 288 GB MI355X: the slots of one
    // right-hand side of a 10M-point tree fit at orders 7 and 9), at least 4 GiB; 16 GiB without a device
    if (!host_only_) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0)
            m2l_budget_bytes_ = std::max<int64_t>(int64_t(4) << 30, static_cast<int64_t>(total_b / 16));
    }
    if (const char *e = std::getenv("BBFMM_M2L_CBUF_MB")) { // read per handle (tests vary it inside one process)
        const double mb = std::atof(e); // fractions allowed (the CPU tests cut small trees into groups)
        if (mb > 0) m2l_budget_bytes_ = static_cast<int64_t>(mb * 1048576.0);
    }

    auto comp = [&](int tv, int a) { return ops_.all_vecs[static_cast<size_t>(tv) * d + a]; };
    auto far = [&](int tv) {
        int mx = 0;
        for (int a = 0; a < d; ++a) mx = std::max(mx, std::abs(comp(tv, a)));
        return mx >= 2;
    };
    // admissible transfer vectors per class (B = V + t, both children of neighbouring parents)
    std::vector<std::vector<int>> tgt_list(ncls), src_list(ncls), tpos_tgt(ncls, std::vector<int>(nvec, -1)),
        tpos_src(ncls, std::vector<int>(nvec, -1));
    for (int o = 0; o < ncls; ++o)
        for (int tv = 0; tv < nvec; ++tv) {
            if (!far(tv)) continue;
            bool okt = true, oks = true;
            for (int a = 0; a < d; ++a) {
                const int oa = (o >> a) & 1, ta = comp(tv, a);
                okt = okt && ta >= oa - 3 && ta <= oa + 2;
                oks = oks && ta >= -2 - oa && ta <= 3 - oa;
            }
            if (okt) {
                tpos_tgt[o][tv] = static_cast<int>(tgt_list[o].size());
                tgt_list[o].push_back(tv);
            }
            if (oks) {
                tpos_src[o][tv] = static_cast<int>(src_list[o].size());
                src_list[o].push_back(tv);
            }
        }
    auto target_class = [&](int o, int tv) {
        int oc = 0;
        for (int a = 0; a < d; ++a) {
            const int v = ((o >> a) & 1) + comp(tv, a);
            oc |= (((v % 2) + 2) % 2) << a;
        }
        return oc;
    };

    std::vector<int32_t> pos_in_class(t.n_cells(), -1);
    int64_t bad_pairs = 0;
    // slot layout of a target of class o at `level`: one segment per admissible transfer vector
    auto slot_layout = [&](int level, std::vector<std::vector<int>> *off_tgt, std::vector<int> *k_pad) {
        const auto &lops = ops_.m2l[level];
        off_tgt->assign(ncls, {});
        k_pad->assign(ncls, 0);
        for (int o = 0; o < ncls; ++o) {
            int off = 0;
            for (int tv : tgt_list[o]) {
                (*off_tgt)[o].push_back(off);
                off += round_up(lops[ops_.ref_lookup[tv]].rank, 2); // 16-byte aligned segments
            }
            (*k_pad)[o] = round_up(std::max(off, 16), 16);
        }
    };
    // ---- batches.  The slots of all targets (one per cell, sum_t r_t doubles: 37 KB at order 7) are what the two
    // stages exchange -- 10.9 GB per right-hand side at 10M points, 76 GB for the finest level of an 80M-point tree.
    // They go through one buffer of at most m2l_budget_bytes_: consecutive levels share a batch while they fit; a
    // level that does not fit alone is cut into 2, 4 or 8 groups of target classes, and the sources of that level
    // get one stacked stage-1 operator per group (the transfer vectors that end in the group's classes: the same
    // tables over fewer transfer vectors, like the boundary variants below).
    std::vector<int> level_groups(static_cast<size_t>(t.depth) + 1, 1);
    std::vector<std::vector<int>> batch_of(static_cast<size_t>(t.depth) + 1, std::vector<int>(ncls, -1));
    {
        const int64_t budget = std::max<int64_t>(m2l_budget_bytes_ / 8 - 128, 1);
        std::vector<std::vector<int64_t>> class_len(static_cast<size_t>(t.depth) + 1, std::vector<int64_t>(ncls, 0));
        for (int level = 2; level <= t.depth; ++level) {
            std::vector<std::vector<int>> off_tgt;
            std::vector<int> k_pad;
            slot_layout(level, &off_tgt, &k_pad);
            for (int64_t c = t.level_ptr[level]; c < t.level_ptr[level + 1]; ++c) class_len[level][t.octant[c]] += k_pad[t.octant[c]];
        }
        int64_t cur_len = 0;
        for (int level = 2; level <= t.depth; ++level) {
            int64_t len = 0;
            for (int64_t v : class_len[level]) len += v;
            cbuf_total_len_ += len;
            int G = 1;
            if (len > budget)
                for (G = 2; G < ncls; G *= 2) {
                    int64_t worst = 0;
                    for (int g = 0; g < G; ++g) {
                        int64_t gl = 0;
                        for (int o = g * ncls / G; o < (g + 1) * ncls / G; ++o) gl += class_len[level][o];
                        worst = std::max(worst, gl);
                    }
                    if (worst <= budget) break;
                }
            if (G >= ncls) { // one group per class is the finest cut there is: the largest class may still not fit
                G = ncls;
                int64_t worst = 0;
                for (int64_t v : class_len[level]) worst = std::max(worst, v);
                if (worst > budget && std::getenv("BBFMM_VERBOSE"))
                    std::fprintf(stderr, "[bbfmm] warning: level %d needs %.1f MB of M2L intermediate per right-hand side even "
                                         "with one batch per target class; the budget of %.1f MB (BBFMM_M2L_CBUF_MB) is exceeded\n",
                                 level, worst * 8.0 / 1048576.0, m2l_budget_bytes_ / 1048576.0);
            }
            level_groups[level] = G;
            if (G == 1 && !m2l_batches_.empty() && m2l_batches_.back().groups == 1 && cur_len + len <= budget) {
                m2l_batches_.back().level_hi = level; // shares the batch of the level above
                cur_len += len;
                for (int o = 0; o < ncls; ++o) batch_of[level][o] = static_cast<int>(m2l_batches_.size()) - 1;
                continue;
            }
            for (int g = 0; g < G; ++g) {
                M2lBatch b;
                b.level_lo = b.level_hi = level;
                b.groups = G;
                b.group = g;
                for (int o = g * ncls / G; o < (g + 1) * ncls / G; ++o) batch_of[level][o] = static_cast<int>(m2l_batches_.size());
                m2l_batches_.push_back(b);
            }
            cur_len = len;
        }
    }
    std::vector<std::vector<M2lTileDesc>> tiles1_of_batch(m2l_batches_.size());
    std::vector<int32_t> variant_batch; // batch of every entry of m2l_variants_
    std::vector<std::vector<int32_t>> zero_of_batch(m2l_batches_.size()); // (slot / 2, length / 2) of the absent pairs' segments
    for (int level = 2; level <= t.depth; ++level) {
        const auto &lops = ops_.m2l[level];
        auto rank_of = [&](int tv) { return lops[ops_.ref_lookup[tv]].rank; };
        std::vector<std::vector<int>> off_tgt;
        std::vector<int> k_pad;
        slot_layout(level, &off_tgt, &k_pad);
        const size_t first_class = m2l_host_.size();
        // Stage-1 row tables of a class-o operator stacked over the transfer vectors `tvs` (the whole admissible
        // list for the class itself, the present ones for a boundary variant): every transfer vector's rows
        // start at an even stacked row (the scatter stores pairs of adjacent rows as 16 bytes).
        auto stage1_rows = [&](int o, const std::vector<int> &tvs, HostM2lClass *hcp) {
            HostM2lClass &hc = *hcp;
            hc.n_t = static_cast<int>(tvs.size());
            hc.n_rows = 0;
            for (int tv : tvs) hc.n_rows += round_up(rank_of(tv), 2);
            hc.r_pad16 = round_up(std::max(hc.n_rows, 1), kM2lS1Block);
            hc.row_tpos.assign(hc.r_pad16, -1);
            hc.row_off.assign(hc.r_pad16, 0);
            hc.src_tv = tvs;
            int row = 0;
            hc.src_row0.assign(tvs.size(), 0);
            hc.src_row1.assign(tvs.size(), 0);
            for (size_t pos = 0; pos < tvs.size(); ++pos) {
                const int tv = tvs[pos];
                const int oc = target_class(o, tv);
                const int base_off = off_tgt[oc][tpos_tgt[oc][tv]];
                hc.src_row0[pos] = row;
                for (int kk = 0; kk < rank_of(tv); ++kk, ++row) {
                    hc.row_tpos[row] = static_cast<int32_t>(pos);
                    hc.row_off[row] = base_off + kk;
                }
                hc.src_row1[pos] = row;
                row = round_up(row, 2); // the padding row keeps tpos -1 (never stored on its own)
            }
            // per column block: first transfer-vector position, and the packed row table
            const int n_blk = hc.r_pad16 / kM2lS1Block;
            hc.blk_t0.assign(n_blk, 0);
            hc.row_dst.assign(hc.r_pad16, -1);
            for (int b = 0; b < n_blk; ++b) {
                int t0 = -1, t1 = -1;
                for (int r = b * kM2lS1Block; r < (b + 1) * kM2lS1Block; ++r) {
                    if (hc.row_tpos[r] < 0) continue;
                    if (t0 < 0) t0 = hc.row_tpos[r];
                    t1 = hc.row_tpos[r];
                }
                if (t0 < 0) continue;
                hc.blk_t0[b] = t0;
                m2l_slot_t_ = std::max(m2l_slot_t_, t1 - t0 + 1);
                for (int r = b * kM2lS1Block; r < (b + 1) * kM2lS1Block; ++r) {
                    if (hc.row_tpos[r] < 0) continue;
                    if (hc.row_off[r] >= (1 << 24)) return false;
                    hc.row_dst[r] = ((hc.row_tpos[r] - t0) << 24) | hc.row_off[r];
                }
            }
            return true;
        };
        m2l_host_.resize(first_class + ncls);
        m2l_group_ops_.resize(first_class + ncls);
        for (int64_t c = t.level_ptr[level]; c < t.level_ptr[level + 1]; ++c)
            m2l_host_[first_class + t.octant[c]].cells.push_back(static_cast<int32_t>(c));
        // Order the cells of a class by their V-list pattern (complete lists first, equal patterns
        // together, Morton order inside a pattern): the 128-cell tiles then hold cells that miss
        // the same transfer vectors (domain boundary, coarse neighbours), which lets stage 2 skip
        // the contraction steps no cell of a tile needs.
        {
            std::vector<uint64_t> key(static_cast<size_t>(t.level_ptr[level + 1] - t.level_ptr[level]));
            const int64_t c_lo = t.level_ptr[level];
            for (int64_t c = c_lo; c < t.level_ptr[level + 1]; ++c) {
                uint64_t hsh = 1469598103934665603ull;
                const int64_t nv = t.v.ptr[c + 1] - t.v.ptr[c];
                uint64_t bits[6] = {0, 0, 0, 0, 0, 0}; // presence over the 7^d transfer vectors
                for (int64_t q = t.v.ptr[c]; q < t.v.ptr[c + 1]; ++q) {
                    const int tv = t.v_tidx[q];
                    if (tv >= 0 && tv < 384) bits[tv >> 6] |= 1ull << (tv & 63);
                }
                for (uint64_t b : bits) hsh = (hsh ^ b) * 1099511628211ull;
                // complete lists first; the hash only has to keep equal patterns together
                key[static_cast<size_t>(c - c_lo)] = (static_cast<uint64_t>(1023 - std::min<int64_t>(nv, 1023)) << 54) | (hsh >> 10);
            }
            parallel_for(ncls, 1, [&](int64_t o) { // the classes are disjoint cell sets
                auto &cells = m2l_host_[first_class + static_cast<size_t>(o)].cells;
                std::stable_sort(cells.begin(), cells.end(), [&](int32_t a, int32_t b) {
                    return key[static_cast<size_t>(a - c_lo)] < key[static_cast<size_t>(b - c_lo)];
                });
                for (size_t i = 0; i < cells.size(); ++i) pos_in_class[cells[i]] = static_cast<int32_t>(i);
            });
        }
        for (int o = 0; o < ncls; ++o) {
            HostM2lClass &hc = m2l_host_[first_class + o];
            hc.level = level;
            hc.octant = o;
            hc.n_t = static_cast<int>(src_list[o].size());
            hc.k_pad = k_pad[o];
            // stage 1 tall operator rows
            if (!stage1_rows(o, src_list[o], &hc)) return fail(BBFMM_BAD_ARGUMENT, "M2L slot too long for the packed row table");
            if (hc.cells.empty()) continue;
            hc.tgt_tv = tgt_list[o];
            hc.tgt_off = off_tgt[o];
            if (host_only_) fill_m2l_operator_arrays(hc, &hc.vt_all, &hc.u_all);
            hc.cbase.resize(hc.cells.size());
            int64_t &cursor = m2l_batches_[static_cast<size_t>(batch_of[level][o])].len; // slot addresses are relative to the batch
            for (size_t i = 0; i < hc.cells.size(); ++i) {
                hc.cbase[i] = cursor;
                cursor += hc.k_pad;
            }
            hc.cslot.resize(hc.cells.size() * static_cast<size_t>(hc.n_t));
            {
                int32_t *cs = hc.cslot.data();
                parallel_for_chunks(static_cast<int64_t>(hc.cslot.size()), int64_t(1) << 18, [&](int64_t b, int64_t e) {
                    std::fill(cs + b, cs + e, int32_t(-1));
                });
            }
        }
        // cslot: for every V pair (B <- V, t) the slot of B as seen from V
        // (threaded: every (V, t) slot has exactly one writer; flop and error counts are reduced per chunk)
        {
            const int64_t b0 = t.level_ptr[level], nb_cells = t.level_ptr[level + 1] - b0;
            constexpr int64_t kChunkB = 2048;
            const int64_t nch = (nb_cells + kChunkB - 1) / kChunkB;
            std::vector<double> flops_part(static_cast<size_t>(std::max<int64_t>(nch, 1)), 0.0);
            std::vector<int64_t> bad_part(static_cast<size_t>(std::max<int64_t>(nch, 1)), 0);
            parallel_for_chunks(nb_cells, kChunkB, [&](int64_t lo, int64_t hi) {
                double fl = 0.0;
                int64_t bad = 0;
                for (int64_t B = b0 + lo; B < b0 + hi; ++B) {
                    const HostM2lClass &hb = m2l_host_[first_class + t.octant[B]];
                    const int64_t base = hb.cbase[pos_in_class[B]];
                    for (int64_t q = t.v.ptr[B]; q < t.v.ptr[B + 1]; ++q) {
                        const int32_t V = t.v.idx[q];
                        const int tv = t.v_tidx[q];
                        HostM2lClass &hv = m2l_host_[first_class + t.octant[V]];
                        const int ps = (tv >= 0 && tv < nvec) ? tpos_src[t.octant[V]][tv] : -1;
                        if (ps < 0 || t.level[V] != level || tpos_tgt[t.octant[B]][tv] < 0) {
                            ++bad;
                            continue;
                        }
                        hv.cslot[static_cast<size_t>(pos_in_class[V]) * hv.n_t + ps] = static_cast<int32_t>(base / 2);
                        const int r = rank_of(tv);
                        fl += compressed ? 4.0 * n * r : 2.0 * n * static_cast<double>(n);
                    }
                }
                flops_part[static_cast<size_t>(lo / kChunkB)] = fl;
                bad_part[static_cast<size_t>(lo / kChunkB)] = bad;
            });
            double level_flops = 0.0;
            for (double f : flops_part) level_flops += f; // fixed order: same total on every run
            m2l_flops_k1_ += level_flops;
            if (m2l_flops_level_.size() <= static_cast<size_t>(level)) m2l_flops_level_.resize(static_cast<size_t>(level) + 1, 0.0);
            m2l_flops_level_[static_cast<size_t>(level)] = level_flops;
            for (int64_t b : bad_part) bad_pairs += b;
        }
        // Batches share one buffer, so a slot segment whose pair does not exist (domain boundary, coarser neighbour)
        // holds another batch's values when stage 2 reads it: such segments are zeroed before every pass
        // (launch_m2l_zero_segments).  A single batch keeps the zeros the buffer was allocated with.
        if (m2l_batches_.size() > 1)
            for (int o = 0; o < ncls; ++o) {
                const HostM2lClass &hc = m2l_host_[first_class + o];
                std::vector<int32_t> &zs = zero_of_batch[static_cast<size_t>(batch_of[level][o])];
                std::vector<uint8_t> present(tgt_list[o].size());
                for (size_t i = 0; i < hc.cells.size(); ++i) {
                    const int64_t B = hc.cells[i];
                    std::fill(present.begin(), present.end(), uint8_t(0));
                    for (int64_t q = t.v.ptr[B]; q < t.v.ptr[B + 1]; ++q) {
                        const int tv = t.v_tidx[q];
                        const int pos = tv >= 0 && tv < nvec ? tpos_tgt[o][tv] : -1;
                        if (pos >= 0) present[static_cast<size_t>(pos)] = 1;
                    }
                    for (size_t pos = 0; pos < present.size(); ++pos)
                        if (!present[pos]) {
                            zs.push_back(static_cast<int32_t>((hc.cbase[i] + off_tgt[o][pos]) / 2));
                            zs.push_back(round_up(rank_of(tgt_list[o][pos]), 2) / 2);
                        }
                }
            }
        for (int o = 0; o < ncls; ++o) {
            const HostM2lClass &hc = m2l_host_[first_class + o];
            const int nq = hc.k_pad / 16;
            const int64_t n_tiles_cls = (static_cast<int64_t>(hc.cells.size()) + kM2lTile - 1) / kM2lTile;
            std::vector<std::vector<uint16_t>> tile_q(static_cast<size_t>(n_tiles_cls));
            parallel_for(n_tiles_cls, 4, [&](int64_t ti) {
                const int32_t first = static_cast<int32_t>(ti * kM2lTile);
                const int32_t count = std::min<int32_t>(kM2lTile, static_cast<int32_t>(hc.cells.size()) - first);
                // contraction steps (16 slot entries each) that hold at least one V-list entry of the tile
                std::vector<uint8_t> act(static_cast<size_t>(nq), 0);
                for (int32_t i = 0; i < count; ++i) {
                    const int64_t B = hc.cells[first + i];
                    for (int64_t q = t.v.ptr[B]; q < t.v.ptr[B + 1]; ++q) {
                        const int tv = t.v_tidx[q];
                        const int pos = tpos_tgt[o][tv];
                        if (pos < 0) continue;
                        const int a = off_tgt[o][pos], b = a + rank_of(tv);
                        for (int sq = a / 16; sq <= (b - 1) / 16; ++sq) act[sq] = 1;
                    }
                }
                for (int sq = 0; sq < nq; ++sq)
                    if (act[sq]) tile_q[static_cast<size_t>(ti)].push_back(static_cast<uint16_t>(sq));
            });
            for (int64_t ti = 0; ti < n_tiles_cls; ++ti) {
                M2lTileDesc td;
                td.level_class = static_cast<int32_t>(first_class + o);
                td.first = static_cast<int32_t>(ti * kM2lTile);
                td.count = std::min<int32_t>(kM2lTile, static_cast<int32_t>(hc.cells.size()) - td.first);
                td.pad = 0;
                td.q_first = static_cast<int32_t>(m2l_qlist_h_.size());
                m2l_qlist_h_.insert(m2l_qlist_h_.end(), tile_q[static_cast<size_t>(ti)].begin(), tile_q[static_cast<size_t>(ti)].end());
                td.q_count = static_cast<int32_t>(m2l_qlist_h_.size()) - td.q_first;
                m2l_tiles_h_.push_back(td);
            }
        }
            // ---- stage-1 variants (boundary classes).  A source cell computes the compressed vectors of ALL admissible
        // transfer vectors of its class, also of those whose target does not exist (domain boundary, coarse
        // neighbours): 6 % of the stage-1 flops of a uniform cube, far more on clustered data.  Cells of a class
        // are sorted by V-list pattern, so cells that miss the same targets sit together: a run of at least four
        // full tiles of such cells gets its own stacked operator with the missing transfer vectors left out (the
        // same reference operators, gathered on the device); the remaining cells keep the class operator.  Only
        // the unrestricted stage 1 (the matvec) uses the variants; plans keep the class tables.
        // BBFMM_M2L_VARIANTS = 0: none; n > 0: runs of at least n full tiles (default 4: a variant costs setup
        // time -- tables, one more operator -- that only a long run of tiles earns back)
        const int variant_min_tiles = [] {
            const char *e = std::getenv("BBFMM_M2L_VARIANTS");
            return e ? std::atoi(e) : 4;
        }();
        const int G = level_groups[level];
        const bool variants_on = variant_min_tiles > 0 && G == 1;
        // A level cut into groups of target classes: per (group, source class) one stacked operator over the transfer
        // vectors whose targets lie in the group, all cells of the class as its tiles.
        for (int g = 0; g < G && G > 1; ++g) {
            const int o_lo = g * ncls / G, o_hi = (g + 1) * ncls / G;
            for (int o = 0; o < ncls; ++o) {
                const HostM2lClass &hc = m2l_host_[first_class + o];
                const size_t nc = hc.cells.size();
                if (nc == 0) continue;
                const int nt = hc.n_t;
                std::vector<int> tvs, keep;
                for (int ps = 0; ps < nt; ++ps) {
                    const int oc = target_class(o, hc.src_tv[ps]);
                    if (oc >= o_lo && oc < o_hi) {
                        tvs.push_back(hc.src_tv[ps]);
                        keep.push_back(ps);
                    }
                }
                if (tvs.empty()) continue;
                HostM2lClass v;
                v.level = level;
                v.octant = o;
                v.k_pad = 16;
                if (!stage1_rows(o, tvs, &v)) return fail(BBFMM_BAD_ARGUMENT, "M2L slot too long for the packed row table");
                v.cells = hc.cells;
                v.cslot.resize(nc * tvs.size());
                {
                    int32_t *dst = v.cslot.data();
                    const int32_t *src = hc.cslot.data();
                    const size_t nk = keep.size();
                    parallel_for_chunks(static_cast<int64_t>(nc), 4096, [&](int64_t lo, int64_t hi) {
                        for (int64_t k = lo; k < hi; ++k)
                            for (size_t q = 0; q < nk; ++q) dst[static_cast<size_t>(k) * nk + q] = src[static_cast<size_t>(k) * nt + keep[q]];
                    });
                }
                const int bidx = batch_of[level][o_lo];
                for (size_t f = 0; f < nc; f += kM2lTile) {
                    M2lTileDesc td;
                    std::memset(&td, 0, sizeof td);
                    td.level_class = -1 - static_cast<int32_t>(m2l_variants_.size()); // fixed up below
                    td.first = static_cast<int32_t>(f);
                    td.count = static_cast<int32_t>(std::min<size_t>(kM2lTile, nc - f));
                    td.pad = 0;
                    tiles1_of_batch[static_cast<size_t>(bidx)].push_back(td);
                }
                if (host_only_) fill_m2l_operator_arrays(v, &v.vt_all, &v.u_all);
                m2l_group_ops_[first_class + o].push_back(static_cast<int32_t>(m2l_variants_.size()));
                variant_batch.push_back(bidx);
                m2l_variants_.push_back(std::move(v));
            }
        }
        for (int o = 0; o < ncls && G == 1; ++o) {
            const HostM2lClass &hc = m2l_host_[first_class + o];
            const size_t nc = hc.cells.size();
            if (nc == 0) continue;
            const int nt = hc.n_t;
            std::vector<M2lTileDesc> &tiles1_out = tiles1_of_batch[static_cast<size_t>(batch_of[level][o])];
            // pattern signature per cell (which targets exist)
            std::vector<uint64_t> sig(nc);
            parallel_for(static_cast<int64_t>(nc), 256, [&](int64_t i) {
                uint64_t h = 1469598103934665603ull;
                const int32_t *row = &hc.cslot[static_cast<size_t>(i) * nt];
                uint64_t word = 0;
                for (int ps = 0; ps < nt; ++ps) {
                    word = (word << 1) | (row[ps] >= 0 ? 1u : 0u);
                    if ((ps & 63) == 63 || ps == nt - 1) {
                        h = (h ^ word) * 1099511628211ull;
                        word = 0;
                    }
                }
                sig[static_cast<size_t>(i)] = h;
            });
            auto same_pattern = [&](size_t a, size_t b) {
                if (sig[a] != sig[b]) return false;
                const int32_t *ra = &hc.cslot[a * nt], *rb = &hc.cslot[b * nt];
                for (int ps = 0; ps < nt; ++ps)
                    if ((ra[ps] >= 0) != (rb[ps] >= 0)) return false;
                return true;
            };
            std::vector<int32_t> rest; // class positions that keep the class operator
            size_t i = 0;
            while (i < nc) {
                size_t j = i + 1;
                while (j < nc && same_pattern(i, j)) ++j;
                size_t full = 0;
                if (variants_on && j - i >= static_cast<size_t>(variant_min_tiles) * kM2lTile) {
                    int present_rows = 0;
                    std::vector<int> tvs;
                    for (int ps = 0; ps < nt; ++ps)
                        if (hc.cslot[i * nt + ps] >= 0) {
                            tvs.push_back(hc.src_tv[ps]);
                            present_rows += round_up(rank_of(hc.src_tv[ps]), 2);
                        }
                    // worth a variant: at least one column block of 26 saved
                    if (!tvs.empty() && round_up(present_rows, kM2lS1Block) < hc.r_pad16) {
                        full = (j - i) / kM2lTile * kM2lTile;
                        HostM2lClass v;
                        v.level = level;
                        v.octant = o;
                        v.k_pad = 16;
                        if (!stage1_rows(o, tvs, &v)) return fail(BBFMM_BAD_ARGUMENT, "M2L slot too long for the packed row table");
                        v.cells.assign(hc.cells.begin() + static_cast<std::ptrdiff_t>(i), hc.cells.begin() + static_cast<std::ptrdiff_t>(i + full));
                        v.cslot.resize(full * tvs.size());
                        size_t pv = 0;
                        std::vector<int> keep;
                        for (int ps = 0; ps < nt; ++ps)
                            if (hc.cslot[i * nt + ps] >= 0) keep.push_back(ps);
                        for (size_t k = 0; k < full; ++k)
                            for (int ps : keep) v.cslot[pv++] = hc.cslot[(i + k) * nt + ps];
                        for (size_t f = 0; f < full; f += kM2lTile) {
                            M2lTileDesc td;
                            std::memset(&td, 0, sizeof td);
                            td.level_class = -1 - static_cast<int32_t>(m2l_variants_.size()); // fixed up below
                            td.first = static_cast<int32_t>(f);
                            td.count = kM2lTile;
                            td.pad = 0;
                            tiles1_out.push_back(td);
                        }
                        if (host_only_) fill_m2l_operator_arrays(v, &v.vt_all, &v.u_all);
                        variant_batch.push_back(batch_of[level][o]);
                        m2l_variants_.push_back(std::move(v));
                    }
                }
                for (size_t k = i + full; k < j; ++k) rest.push_back(static_cast<int32_t>(k));
                i = j;
            }
            for (size_t f = 0; f < rest.size(); f += kM2lTile) {
                M2lTileDesc td;
                std::memset(&td, 0, sizeof td);
                td.level_class = static_cast<int32_t>(first_class + o);
                td.first = static_cast<int32_t>(m2l_tile_idx1_h_.size() + f);
                td.count = static_cast<int32_t>(std::min<size_t>(kM2lTile, rest.size() - f));
                td.pad = 1; // first indexes the position list
                tiles1_out.push_back(td);
            }
            m2l_tile_idx1_h_.insert(m2l_tile_idx1_h_.end(), rest.begin(), rest.end());
        }
    }
    if (bad_pairs > 0)
        return fail(BBFMM_UNSUPPORTED,
                    "V-list pairs outside the admissible transfer-vector set (source points outside the root box?)");
    // device class table: level classes, then variants / group operators; every entry works for one batch
    m2l_batch_of_class_.assign(m2l_host_.size() + m2l_variants_.size(), 0);
    for (size_t lc = 0; lc < m2l_host_.size(); ++lc)
        m2l_batch_of_class_[lc] = batch_of[static_cast<size_t>(m2l_host_[lc].level)][static_cast<size_t>(m2l_host_[lc].octant)];
    for (size_t v = 0; v < m2l_variants_.size(); ++v) m2l_batch_of_class_[m2l_host_.size() + v] = variant_batch[v];
    // launch lists, batch by batch: stage 1 from the per-batch lists, stage 2 = the class tiles (classes of a batch
    // are consecutive) with the tail of every batch split
    m2l_tiles2_h_.clear();
    {
        size_t next = 0;
        for (size_t b = 0; b < m2l_batches_.size(); ++b) {
            M2lBatch &mb = m2l_batches_[b];
            mb.t1_first = static_cast<int32_t>(m2l_tiles1_h_.size());
            for (M2lTileDesc td : tiles1_of_batch[b]) {
                if (td.level_class < 0) td.level_class = static_cast<int32_t>(m2l_host_.size()) + (-1 - td.level_class);
                m2l_tiles1_h_.push_back(td);
            }
            mb.t1_count = static_cast<int32_t>(m2l_tiles1_h_.size()) - mb.t1_first;
            std::vector<M2lTileDesc> part;
            while (next < m2l_tiles_h_.size() && m2l_batch_of_class_[static_cast<size_t>(m2l_tiles_h_[next].level_class)] == static_cast<int32_t>(b))
                part.push_back(m2l_tiles_h_[next++]);
            split_tile_tail(&part, n_cu_);
            mb.t2_first = static_cast<int32_t>(m2l_tiles2_h_.size());
            mb.t2_count = static_cast<int32_t>(part.size());
            m2l_tiles2_h_.insert(m2l_tiles2_h_.end(), part.begin(), part.end());
            cbuf_batch_len_ = std::max(cbuf_batch_len_, mb.len);
        }
        if (next != m2l_tiles_h_.size()) return fail(BBFMM_DEVICE_ERROR, "internal: M2L tiles out of batch order");
    }
    m2l_zero_h_.clear();
    m2l_zero_ptr_.assign(m2l_batches_.size() + 1, 0);
    for (size_t b = 0; b < m2l_batches_.size(); ++b) {
        m2l_zero_h_.insert(m2l_zero_h_.end(), zero_of_batch[b].begin(), zero_of_batch[b].end());
        m2l_zero_ptr_[b + 1] = static_cast<int64_t>(m2l_zero_h_.size() / 2);
    }
    cbuf_batch_len_ += 128; // + dump area for the branch-free stage-1 scatter (never read)
    if (cbuf_batch_len_ / 2 >= (int64_t(1) << 31))
        return fail(BBFMM_UNSUPPORTED, "M2L intermediate buffer of one batch too large (raise the number of groups: lower BBFMM_M2L_CBUF_MB)");
    return BBFMM_OK;
}

// ------------------------------------------------------------------ upload
int FmmTree::upload() {
    const HostTree &t = tree_;
    const int d = d_;
    const int64_t N = t.n_points, C = t.n_cells();
    // Chebyshev tables
    DevCheb hc{};
    hc.p = order_;
    hc.d = d;
    hc.n = ops_.n;
    hc.n_pad = round_up(ops_.n, 32);
    std::copy(ops_.polyn.begin(), ops_.polyn.end(), hc.polyn);
    std::copy(ops_.nodes.begin(), ops_.nodes.end(), hc.nodes);
    std::copy(ops_.xfer.begin(), ops_.xfer.end(), hc.xfer);
    CHK(dalloc(&d_cheb_, 1));
    HIPCHK(hipMemcpy(d_cheb_.p, &hc, sizeof hc, hipMemcpyHostToDevice));
    cheb_ = ChebRef{d_cheb_.p, hc.p, hc.d, hc.n, hc.n_pad};

    // sorted sources (SoA); unused axes alias one zero array
    CHK(dalloc(&d_zero_axis_, static_cast<size_t>(N), true));
    if (dev_points_.order && dev_points_.n == N) {
        // the device tree build left the points and the hierarchical order in HBM: gather there
        for (int a = 0; a < 3; ++a) {
            if (a < d) {
                CHK(dalloc(&d_src_[a], static_cast<size_t>(N)));
                src_ptr_[a] = d_src_[a].p;
            } else {
                src_ptr_[a] = d_zero_axis_.p;
            }
        }
        launch_gather_targets(dev_points_.xyz[0], dev_points_.xyz[1], dev_points_.xyz[2],
                              reinterpret_cast<const int32_t *>(dev_points_.order), N, d > 0 ? d_src_[0].p : nullptr,
                              d > 1 ? d_src_[1].p : nullptr, d > 2 ? d_src_[2].p : nullptr, stream_);
        d_order_.p = reinterpret_cast<int32_t *>(dev_points_.order); // rows < 2^31: same bits as int32
        d_order_.n = static_cast<size_t>(N);
        owned_.push_back(dev_points_.order);
        dev_points_.order = nullptr;
        HIPCHK(hipStreamSynchronize(stream_));
        free_dev_tree_points(&dev_points_);
    } else {
        std::vector<double> tmp(static_cast<size_t>(N));
        for (int a = 0; a < 3; ++a) {
            if (a < d) {
                const double *col = &pts_[static_cast<size_t>(a) * N];
                parallel_for(N, 1 << 16, [&](int64_t i) { tmp[i] = col[t.order[i]]; });
                CHK(dupload(&d_src_[a], tmp));
                src_ptr_[a] = d_src_[a].p;
            } else {
                src_ptr_[a] = d_zero_axis_.p;
            }
        }
        std::vector<int32_t> ord(static_cast<size_t>(N));
        for (int64_t i = 0; i < N; ++i) ord[i] = static_cast<int32_t>(t.order[i]);
        CHK(dupload(&d_order_, ord));
    }
    {
        std::vector<double> c3(static_cast<size_t>(C) * 3, 0.0);
        for (int64_t c = 0; c < C; ++c)
            for (int a = 0; a < d; ++a) c3[c * 3 + a] = t.centers[c * d + a];
        CHK(dupload(&d_centers_, c3));
        CHK(dupload(&d_lengths_, t.lengths));
        std::vector<int32_t> b(C), e(C);
        for (int64_t c = 0; c < C; ++c) {
            b[c] = static_cast<int32_t>(t.pt_begin[c]);
            e[c] = static_cast<int32_t>(t.pt_end[c]);
        }
        CHK(dupload(&d_pt_begin_, b));
        CHK(dupload(&d_pt_end_, e));
        CHK(dupload(&d_parent_, t.parent));
        CHK(dupload(&d_octant_, t.octant));
        CHK(dupload(&d_child_ptr_, t.children.ptr));
        CHK(dupload(&d_child_idx_, t.children.idx));
        CHK(dupload(&d_src_leaves_, src_leaves_));
    }
    d_m2m_parents_.resize(level_cells_.size());
    d_level_cells_.resize(level_cells_.size());
    for (size_t l = 0; l < level_cells_.size(); ++l) {
        CHK(dupload(&d_m2m_parents_[l], m2m_parents_[l]));
        CHK(dupload(&d_level_cells_[l], level_cells_[l]));
    }
    CHK(dupload(&d_u_run_ptr_, u_runs_.ptr));
    CHK(dupload(&d_u_runs_, u_runs_.idx));
    CHK(dupload(&d_w_ptr_, t.w.ptr));
    CHK(dupload(&d_w_idx_, t.w.idx));
    {
        // P2L jobs: compact run_ptr over the cells that have an X list
        std::vector<int64_t> job_ptr(x_cells_.size() + 1, 0);
        std::vector<int32_t> job_runs;
        for (size_t j = 0; j < x_cells_.size(); ++j) {
            const int32_t c = x_cells_[j];
            for (int64_t r = x_runs_.ptr[c]; r < x_runs_.ptr[c + 1]; ++r) {
                job_runs.push_back(x_runs_.idx[2 * r]);
                job_runs.push_back(x_runs_.idx[2 * r + 1]);
            }
            job_ptr[j + 1] = static_cast<int64_t>(job_runs.size() / 2);
        }
        CHK(dupload(&d_x_cells_, x_cells_));
        CHK(dupload(&d_x_job_run_ptr_, job_ptr));
        CHK(dupload(&d_x_runs_, job_runs));
    }
    StageTimer ut;
    HIPCHK(hipStreamSynchronize(stream_));
    ut.lap("  upload: tree, run lists");
    // M2L tables
    m2l_classes_h_.resize(m2l_host_.size() + m2l_variants_.size());
    // The stacked operators (GBs at p = 9) are gathered on the device from the levels' reference operators and
    // the symmetry tables (MBs): per level one buffer [U_ref | Vt_ref] of all reference vectors.
    const bool compressed = ops_.compression != kCompressionNone;
    DevBuf<int32_t> d_invperm;
    CHK(dupload(&d_invperm, ops_.invperm));
    std::vector<DevBuf<double>> d_level_ops(ops_.m2l.size());
    std::vector<std::vector<int64_t>> u_off(ops_.m2l.size()), vt_off(ops_.m2l.size());
    for (size_t lv = 0; lv < ops_.m2l.size(); ++lv) {
        std::vector<double> buf;
        for (const M2lOperator &op : ops_.m2l[lv]) {
            u_off[lv].push_back(static_cast<int64_t>(buf.size()));
            buf.insert(buf.end(), op.u.begin(), op.u.end());
            vt_off[lv].push_back(static_cast<int64_t>(buf.size()));
            buf.insert(buf.end(), op.vt.begin(), op.vt.end());
        }
        if (!buf.empty()) CHK(dupload(&d_level_ops[lv], buf));
    }
    if (shared_basis_) {
        if (!compressed) return fail(BBFMM_BAD_ARGUMENT, "BBFMM_FLAG_M2L_SHARED_BASIS needs compressed M2L operators (ACA or SVD)");
        CHK(build_shared_basis(&d_level_ops));
    }
    const int m2l_len = shared_basis_ ? basis_pad_ : cheb_.n_pad; // contraction / output length of the stages
    std::vector<DevBuf<double>> natural_tmp;         // shared basis: the natural-frame operators, until projected
    std::vector<DevBuf<M2lAssembleTv>> assemble_tmp; // per-class tv tables: released once the kernels have run
    for (size_t i = 0; i < m2l_host_.size() + m2l_variants_.size(); ++i) {
        // (the boundary variants of stage 1 follow the level classes: same tables, fewer transfer vectors)
        HostM2lClass &h = i < m2l_host_.size() ? m2l_host_[i] : m2l_variants_[i - m2l_host_.size()];
        M2lClass &c = m2l_classes_h_[i];
        std::memset(&c, 0, sizeof c);
        c.n_rows = h.n_rows;
        c.r_pad16 = h.r_pad16;
        c.n_t = h.n_t;
        c.k_pad = h.k_pad;
        c.n_cells = static_cast<int32_t>(h.cells.size());
        if (h.cells.empty()) continue;
        DevBuf<double> vt, ua;
        DevBuf<int32_t> rt, ro, ce, cs;
        DevBuf<int64_t> cb;
        {
            const size_t nvt = static_cast<size_t>(cheb_.n_pad) * h.r_pad16, nu = static_cast<size_t>(h.k_pad) * cheb_.n_pad;
            CHK(dalloc(&vt, nvt));
            CHK(dalloc(&ua, nu));
            const auto &lops = ops_.m2l[h.level];
            std::vector<M2lAssembleTv> src_tv, tgt_tv;
            int row = 0, max_rank = 0;
            for (int tv : h.src_tv) {
                const int ref = ops_.ref_lookup[tv];
                const int rank = lops[ref].rank;
                src_tv.push_back(M2lAssembleTv{ops_.perm_lookup[tv], rank, row, vt_off[h.level][ref], u_off[h.level][ref]});
                row += round_up(rank, 2);
            }
            for (size_t pos = 0; pos < h.tgt_tv.size(); ++pos) {
                const int tv = h.tgt_tv[pos];
                const int ref = ops_.ref_lookup[tv];
                const int rank = lops[ref].rank;
                max_rank = std::max(max_rank, rank);
                tgt_tv.push_back(M2lAssembleTv{ops_.perm_lookup[tv], rank, h.tgt_off[pos], vt_off[h.level][ref], u_off[h.level][ref]});
            }
            DevBuf<M2lAssembleTv> d_src, d_tgt;
            CHK(dupload(&d_src, src_tv));
            CHK(dupload(&d_tgt, tgt_tv));
            const M2lAssembleClass ac{d_src.p, d_tgt.p, static_cast<int32_t>(src_tv.size()), static_cast<int32_t>(tgt_tv.size()),
                                      h.r_pad16, h.k_pad, max_rank};
            static const bool host_fill = std::getenv("BBFMM_M2L_ASSEMBLE_HOST") != nullptr; // checker: the host fill of round 1
            if (host_fill) {
                std::vector<double> hv, hu;
                fill_m2l_operator_arrays(h, &hv, &hu);
                HIPCHK(hipMemcpy(vt.p, hv.data(), hv.size() * sizeof(double), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(ua.p, hu.data(), hu.size() * sizeof(double), hipMemcpyHostToDevice));
            } else {
                launch_m2l_assemble(ac, ops_.n, cheb_.n_pad, compressed, d_level_ops[h.level].p, d_invperm.p, vt.p, ua.p, stream_);
            }
            assemble_tmp.push_back(d_src);
            assemble_tmp.push_back(d_tgt);
            if (shared_basis_) { // Vt' = W^T VtAll (m2l_len x r_pad16), U' = UAll W (k_pad x m2l_len)
                DevBuf<double> vt2, ua2;
                CHK(dalloc(&vt2, static_cast<size_t>(m2l_len) * h.r_pad16));
                CHK(dalloc(&ua2, static_cast<size_t>(h.k_pad) * m2l_len));
                const double *W = d_basis_c_[static_cast<size_t>(h.level)].p; // n_pad x m2l_len
                launch_small_gemm(true, m2l_len, h.r_pad16, cheb_.n_pad, W, m2l_len, vt.p, h.r_pad16, vt2.p, h.r_pad16, stream_);
                launch_small_gemm(false, h.k_pad, m2l_len, cheb_.n_pad, ua.p, cheb_.n_pad, W, m2l_len, ua2.p, m2l_len, stream_);
                natural_tmp.push_back(vt);
                natural_tmp.push_back(ua);
                vt = vt2;
                ua = ua2;
            }
        }
        CHK(dupload(&rt, h.row_dst));
        CHK(dupload(&ro, h.blk_t0));
        CHK(dupload(&ce, h.cells));
        CHK(dalloc(&cs, h.cslot.size()));
        if (!h.cslot.empty()) HIPCHK(hipMemcpy(cs.p, h.cslot.data(), h.cslot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        CHK(dupload(&cb, h.cbase));
        c.vt_all = vt.p;
        c.u_all = ua.p;
        c.row_dst = rt.p;
        c.blk_t0 = ro.p;
        c.cells = ce.p;
        c.cslot = cs.p;
        c.cbase = cb.p;
        decltype(h.cslot)().swap(h.cslot); // only needed for uploading
    }
    HIPCHK(hipStreamSynchronize(stream_)); // the assembly kernels have read their tables
    HIPCHK(hipGetLastError());
    for (auto &b : assemble_tmp) dfree(&b);
    for (auto &b : natural_tmp) dfree(&b);
    for (auto &b : d_level_ops) dfree(&b);
    dfree(&d_invperm);
    ut.lap("  upload: M2L operators, tables");
    CHK(dupload(&d_m2l_classes_, m2l_classes_h_));
    CHK(dupload(&d_m2l_tiles_, m2l_tiles_h_));
    CHK(dupload(&d_m2l_tiles1_, m2l_tiles1_h_));
    CHK(dupload(&d_tile_idx1_, m2l_tile_idx1_h_));
    CHK(dupload(&d_m2l_tiles2_, m2l_tiles2_h_));
    CHK(dupload(&d_m2l_zero_, m2l_zero_h_));
    decltype(m2l_zero_h_)().swap(m2l_zero_h_);
    CHK(dupload(&d_m2l_qlist_, m2l_qlist_h_));
    std::vector<uint8_t> act(static_cast<size_t>(C), 1);
    CHK(dupload(&d_active_, act));
    return BBFMM_OK;
}

// Shared-basis extension.  For every level: G = sum over the far transfer vectors t of K_t^T K_t + K_t K_t^T with
// K_t = U_t Vt_t the level's compressed operators in the natural node order (permuted copies of the reference
// operators), eigen-decomposition of G (n x n) on the device, W = the eigenvectors whose eigenvalues carry all but
// eps^2 of the trace (the operators' own cutoff rule, aca.rs:210-224, applied to the stack of all of them).
int FmmTree::build_shared_basis(std::vector<DevBuf<double>> *d_level_ops) {
    (void)d_level_ops;
    const int n = ops_.n, n_pad = cheb_.n_pad, d = ops_.d;
    const size_t n_levels = ops_.m2l.size();
    basis_rank_.assign(n_levels, 0);
    std::vector<std::vector<double>> basis(n_levels); // n x rank column-major, most important direction first
    const double eps_s = std::max(params_.epsilon, 1e-13);
    std::vector<int> tvs; // the far transfer vectors
    for (int tv = 0; tv < ops_.n_vec; ++tv) {
        int mx = 0;
        for (int a = 0; a < d; ++a) mx = std::max(mx, std::abs(static_cast<int>(ops_.all_vecs[static_cast<size_t>(tv) * d + a])));
        if (mx >= 2) tvs.push_back(tv);
    }
    // G = sum_t (K_t P)^T (K_t P) + (P K_t)(P K_t)^T over the level's operators in the natural node order, P = I - W1 W1^T
    // (k1 = 0: P = I).  K_t = Pi K_ref Pi^T with the symmetry permutations, and P commutes with them when W1 is a union
    // of whole eigenspaces of the undeflated G (which commutes with every Pi): the deflation is applied to the 16
    // reference factor pairs, K = U Vt:  (K P)^T (K P) = Vt'^T (U^T U) Vt',  (P K)(P K)^T = U' (Vt Vt^T) U'^T.
    auto gram = [&](size_t lv, const std::vector<double> &w1, int k1, std::vector<double> *G_out) {
        const auto &lops = ops_.m2l[lv];
        const int n_ref = static_cast<int>(lops.size());
        std::vector<std::vector<double>> gref(static_cast<size_t>(n_ref));
        parallel_for(n_ref, 1, [&](int64_t r) {
            const M2lOperator &op = lops[static_cast<size_t>(r)];
            const int rk = op.rank;
            std::vector<double> &g = gref[static_cast<size_t>(r)];
            g.assign(static_cast<size_t>(n) * n, 0.0);
            if (rk == 0) return;
            std::vector<double> tv(static_cast<size_t>(rk) * rk, 0.0), tu(static_cast<size_t>(rk) * rk, 0.0);
            for (int a = 0; a < rk; ++a)
                for (int b = 0; b < rk; ++b) {
                    double av = 0.0, au = 0.0;
                    for (int m = 0; m < n; ++m) {
                        av += op.vt[a + static_cast<size_t>(rk) * m] * op.vt[b + static_cast<size_t>(rk) * m];
                        au += op.u[m + static_cast<size_t>(n) * a] * op.u[m + static_cast<size_t>(n) * b];
                    }
                    tv[static_cast<size_t>(a) * rk + b] = av;
                    tu[static_cast<size_t>(a) * rk + b] = au;
                }
            std::vector<double> u(op.u), vt(op.vt); // u[i + n a], vt[a + rk m]
            if (k1 > 0) {
                std::vector<double> c(static_cast<size_t>(k1) * rk);
                for (int q = 0; q < k1; ++q) // W1^T U
                    for (int a2 = 0; a2 < rk; ++a2) {
                        double acc = 0.0;
                        for (int i = 0; i < n; ++i) acc += w1[i + static_cast<size_t>(n) * q] * op.u[i + static_cast<size_t>(n) * a2];
                        c[static_cast<size_t>(q) * rk + a2] = acc;
                    }
                for (int q = 0; q < k1; ++q)
                    for (int a2 = 0; a2 < rk; ++a2) {
                        const double cv = c[static_cast<size_t>(q) * rk + a2];
                        for (int i = 0; i < n; ++i) u[i + static_cast<size_t>(n) * a2] -= w1[i + static_cast<size_t>(n) * q] * cv;
                    }
                for (int q = 0; q < k1; ++q) // Vt W1
                    for (int a2 = 0; a2 < rk; ++a2) {
                        double acc = 0.0;
                        for (int m = 0; m < n; ++m) acc += op.vt[a2 + static_cast<size_t>(rk) * m] * w1[m + static_cast<size_t>(n) * q];
                        c[static_cast<size_t>(q) * rk + a2] = acc;
                    }
                for (int q = 0; q < k1; ++q)
                    for (int a2 = 0; a2 < rk; ++a2) {
                        const double cv = c[static_cast<size_t>(q) * rk + a2];
                        for (int m = 0; m < n; ++m) vt[a2 + static_cast<size_t>(rk) * m] -= cv * w1[m + static_cast<size_t>(n) * q];
                    }
            }
            std::vector<double> ut(static_cast<size_t>(n) * rk), vtt(static_cast<size_t>(n) * rk);
            for (int i = 0; i < n; ++i) // ut = U' (Vt Vt^T), vtt = Vt'^T (U^T U)
                for (int b = 0; b < rk; ++b) {
                    double au = 0.0, av = 0.0;
                    for (int a2 = 0; a2 < rk; ++a2) {
                        au += u[i + static_cast<size_t>(n) * a2] * tv[static_cast<size_t>(a2) * rk + b];
                        av += vt[a2 + static_cast<size_t>(rk) * i] * tu[static_cast<size_t>(a2) * rk + b];
                    }
                    ut[static_cast<size_t>(i) * rk + b] = au;
                    vtt[static_cast<size_t>(i) * rk + b] = av;
                }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0.0;
                    for (int a2 = 0; a2 < rk; ++a2)
                        acc += vtt[static_cast<size_t>(i) * rk + a2] * vt[a2 + static_cast<size_t>(rk) * j] +
                               ut[static_cast<size_t>(i) * rk + a2] * u[j + static_cast<size_t>(n) * a2];
                    g[static_cast<size_t>(i) * n + j] = acc;
                }
        });
        std::vector<double> &G = *G_out;
        G.assign(static_cast<size_t>(n) * n, 0.0);
        parallel_for(n, 1, [&](int64_t i) { // natural frame: entry (i, j) of K_t stems from (invperm[i], invperm[j]) of its reference
            double *row = &G[static_cast<size_t>(i) * n];
            for (int tv : tvs) {
                const int32_t *ip = &ops_.invperm[static_cast<size_t>(ops_.perm_lookup[tv]) * n];
                const double *g = &gref[static_cast<size_t>(ops_.ref_lookup[tv])][static_cast<size_t>(ip[i]) * n];
                for (int j = 0; j < n; ++j) row[j] += g[ip[j]];
            }
        });
        for (int i = 0; i < n; ++i) // exact symmetry for the solver
            for (int j = i + 1; j < n; ++j) {
                const double v = 0.5 * (G[static_cast<size_t>(i) * n + j] + G[static_cast<size_t>(j) * n + i]);
                G[static_cast<size_t>(i) * n + j] = G[static_cast<size_t>(j) * n + i] = v;
            }
    };
    // eigenvalues descending, eigenvectors as columns in the same order
    auto eigen = [&](const std::vector<double> &G, std::vector<double> *eval, std::vector<double> *evec) -> int {
        eval->assign(static_cast<size_t>(n), 0.0);
        evec->assign(static_cast<size_t>(n) * n, 0.0);
        std::vector<double> asc(static_cast<size_t>(n)), vasc(static_cast<size_t>(n) * n);
        DevBuf<double> d_g, d_ev;
        CHK(dupload(&d_g, G));
        CHK(dalloc(&d_ev, static_cast<size_t>(n)));
        int rc = std::getenv("BBFMM_BASIS_HOST_EIGEN") ? BBFMM_UNSUPPORTED : device_symmetric_eigen(n, d_g.p, d_ev.p, stream_);
        if (rc == BBFMM_OK) {
            HIPCHK(hipMemcpy(asc.data(), d_ev.p, static_cast<size_t>(n) * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(vasc.data(), d_g.p, static_cast<size_t>(n) * n * sizeof(double), hipMemcpyDeviceToHost));
            for (int j = 0; j < n; ++j) {
                (*eval)[static_cast<size_t>(j)] = asc[static_cast<size_t>(n - 1 - j)];
                std::copy(vasc.begin() + static_cast<size_t>(n - 1 - j) * n, vasc.begin() + static_cast<size_t>(n - j) * n,
                          evec->begin() + static_cast<size_t>(j) * n);
            }
        }
        dfree(&d_g);
        dfree(&d_ev);
        if (rc == BBFMM_UNSUPPORTED) { // no rocSOLVER: one-sided Jacobi on the host (slow at high orders, same result)
            std::vector<double> sv, vt;
            jacobi_svd(G, n, n, evec, &sv, &vt); // G symmetric positive semi-definite: singular values = eigenvalues, descending
            *eval = sv;
            rc = BBFMM_OK;
        }
        if (rc != BBFMM_OK) return fail(rc, "eigen-decomposition of the shared-basis Gram matrix failed");
        return BBFMM_OK;
    };
    // The operators of a level are permuted copies of each other, so the eigenvalues come in multiplets (the symmetry
    // group's irreducible dimensions, at most six members here): a cut inside one would keep an arbitrary part of
    // its eigenspace -- a different part with another solver, and not invariant under the permutations.  `count`
    // leading values of a descending list are extended to the end of their multiplet.
    auto whole_multiplet = [&](const std::vector<double> &ev, int count, double noise) {
        for (int extra = 0; extra < 8 && count > 0 && count < n; ++extra) {
            const double kept = ev[static_cast<size_t>(count - 1)], next = ev[static_cast<size_t>(count)];
            if (next > noise && kept - next <= 1e-3 * kept + 0.6 * noise) ++count;
            else break;
        }
        return count;
    };
    for (size_t lv = 2; lv < n_levels; ++lv) {
        if (ops_.m2l[lv].empty()) continue;
        // Pass 1.  G squares the singular values of the stack, and its f64 rounding noise shows as eigenvalues of
        // either sign around 1e-17 of the largest: values below 1e-16 of it are treated as 0 (hundreds of them would
        // otherwise add up past eps^2 of the trace and put the cut into the noise).
        std::vector<double> G, ev1, vec1;
        gram(lv, std::vector<double>(), 0, &G);
        CHK(eigen(G, &ev1, &vec1));
        const double lam_max = std::max(ev1[0], 0.0), noise1 = 1e-16 * lam_max;
        if (std::getenv("BBFMM_VERBOSE") && lv == 2) {
            std::fprintf(stderr, "[bbfmm] shared basis level %zu eigenvalues / largest (every 10th):", lv);
            for (int j = 0; j < n; j += 10) std::fprintf(stderr, " %.1e", ev1[static_cast<size_t>(j)] / ev1[0]);
            std::fprintf(stderr, "\n");
        }
        auto cut = [&](const std::vector<double> &ev, double head, double noise) { // values kept of a descending list
            double total = head;
            for (double v : ev)
                if (v > noise) total += v;
            double tail = 0.0;
            for (int j = n - 1; j >= 0; --j) {
                if (ev[static_cast<size_t>(j)] > noise) tail += ev[static_cast<size_t>(j)];
                if (!(tail < eps_s * eps_s * total)) return j + 1;
            }
            return 0;
        };
        int rank = 0;
        if (eps_s >= 1e-6) { // the cut lies far above the noise: one pass
            rank = whole_multiplet(ev1, std::max(1, cut(ev1, 0.0, noise1)), noise1);
            basis[lv].assign(vec1.begin(), vec1.begin() + static_cast<size_t>(rank) * n);
        } else {
            // Pass 2: everything above 1e-10 of the largest eigenvalue (whole multiplets) is accurate and kept; the
            // Gram matrix of the stack deflated by those directions carries the rest at its own scale, so the cut at
            // eps^2 of the trace is resolved down to eps ~ 1e-13 instead of 3e-8.
            int k1 = 0;
            while (k1 < n && ev1[static_cast<size_t>(k1)] > 1e-10 * lam_max) ++k1;
            k1 = whole_multiplet(ev1, std::max(1, k1), noise1);
            std::vector<double> w1(vec1.begin(), vec1.begin() + static_cast<size_t>(k1) * n), ev2, vec2;
            double head = 0.0;
            for (int j = 0; j < k1; ++j) head += ev1[static_cast<size_t>(j)];
            gram(lv, w1, k1, &G);
            CHK(eigen(G, &ev2, &vec2));
            const double noise2 = std::max(1e-16 * std::max(ev2[0], 0.0), 1e-30 * lam_max);
            int r2 = cut(ev2, head, noise2);
            if (r2 > 0) r2 = whole_multiplet(ev2, r2, noise2);
            r2 = std::min(r2, n - k1);
            rank = k1 + r2;
            basis[lv] = w1;
            basis[lv].insert(basis[lv].end(), vec2.begin(), vec2.begin() + static_cast<size_t>(r2) * n);
            for (int pass = 0; pass < 2; ++pass) // the second set is orthogonal to the first up to rounding: tidy up
                for (int j = k1; j < rank; ++j) {
                    double *cj = &basis[lv][static_cast<size_t>(j) * n];
                    for (int q = 0; q < j; ++q) {
                        const double *cq = &basis[lv][static_cast<size_t>(q) * n];
                        double dot = 0.0;
                        for (int i = 0; i < n; ++i) dot += cq[i] * cj[i];
                        for (int i = 0; i < n; ++i) cj[i] -= dot * cq[i];
                    }
                    double nn = 0.0;
                    for (int i = 0; i < n; ++i) nn += cj[i] * cj[i];
                    nn = nn > 0.0 ? 1.0 / std::sqrt(nn) : 0.0;
                    for (int i = 0; i < n; ++i) cj[i] *= nn;
                }
        }
        basis_rank_[lv] = rank;
    }
    int max_rank = 0;
    for (int r : basis_rank_) max_rank = std::max(max_rank, r);
    if (max_rank == 0) return fail(BBFMM_BAD_ARGUMENT, "shared basis: no M2L level");
    basis_pad_ = round_up(max_rank, 16);
    if (((basis_pad_ / 16) & 1) && basis_pad_ / 16 != 7) basis_pad_ += 16; // column-group plans: even counts, or 7
    basis_pad_ = std::min(basis_pad_, n_pad);
    if (basis_pad_ * 5 > n_pad * 3) { // the union of the operators fills most of the node space (e.g. Spheroidal3 with a
        // short range): the stages would not get cheaper -- the handle keeps the reference's arithmetic
        if (std::getenv("BBFMM_VERBOSE"))
            std::fprintf(stderr, "[bbfmm] shared basis: rank %d of %d nodes, not used\n", max_rank, n);
        shared_basis_ = false;
        basis_rank_.assign(n_levels, 0);
        basis_pad_ = 0;
        return BBFMM_OK;
    }
    d_basis_c_.assign(n_levels, DevBuf<double>());
    d_basis_e_.assign(n_levels, DevBuf<double>());
    std::vector<M2lClass> classes(2 * n_levels);
    std::vector<M2lTileDesc> tiles_c, tiles_e;
    const int64_t C = tree_.n_cells();
    double flops = 0.0;
    for (size_t lv = 2; lv < n_levels; ++lv) {
        std::memset(&classes[2 * lv], 0, 2 * sizeof(M2lClass));
        if (basis_rank_[lv] == 0 || lv >= level_cells_.size() || level_cells_[lv].empty()) continue;
        const int rank = basis_rank_[lv];
        std::vector<double> wc(static_cast<size_t>(n_pad) * basis_pad_, 0.0), we(static_cast<size_t>(basis_pad_) * n_pad, 0.0);
        for (int j = 0; j < rank; ++j) {
            const double *col = &basis[lv][static_cast<size_t>(j) * n];
            for (int m = 0; m < n; ++m) {
                wc[static_cast<size_t>(m) * basis_pad_ + j] = col[m];
                we[static_cast<size_t>(j) * n_pad + m] = col[m];
            }
        }
        CHK(dupload(&d_basis_c_[lv], wc));
        CHK(dupload(&d_basis_e_[lv], we));
        const int32_t nc = static_cast<int32_t>(level_cells_[lv].size());
        for (int e = 0; e < 2; ++e) {
            M2lClass &c = classes[2 * lv + e];
            c.u_all = e == 0 ? d_basis_c_[lv].p : d_basis_e_[lv].p;
            c.cells = d_level_cells_[lv].p;
            c.n_cells = nc;
        }
        for (int32_t first = 0; first < nc; first += kM2lTile) {
            const int32_t count = std::min<int32_t>(kM2lTile, nc - first);
            tiles_c.push_back(M2lTileDesc{static_cast<int32_t>(2 * lv), first, count, 0, 0, 0});
            tiles_e.push_back(M2lTileDesc{static_cast<int32_t>(2 * lv + 1), first, count, 0, 0, 0});
        }
        const double lf = lv < m2l_flops_level_.size() ? m2l_flops_level_[lv] : 0.0;
        flops += lf * rank / n + 4.0 * n * rank * nc; // the stages in the basis + the two changes of basis
    }
    m2l_flops_k1_ = flops;
    n_basis_tiles_ = static_cast<int>(tiles_c.size());
    CHK(dupload(&d_basis_classes_, classes));
    CHK(dupload(&d_basis_tiles_c_, tiles_c));
    CHK(dupload(&d_basis_tiles_e_, tiles_e));
    (void)C;
    if (std::getenv("BBFMM_VERBOSE")) {
        std::fprintf(stderr, "[bbfmm] shared basis: %d coordinates per cell (n = %d), ranks per level:", basis_pad_, n);
        for (size_t lv = 2; lv < n_levels; ++lv) std::fprintf(stderr, " %d", basis_rank_[lv]);
        std::fprintf(stderr, "\n");
    }
    return BBFMM_OK;
}

int FmmTree::ensure_rhs_capacity(int k) {
    if (k <= k_cap_) return BBFMM_OK;
    const int64_t N = tree_.n_points, C = tree_.n_cells();
    dfree(&d_w_sorted_);
    dfree(&d_M_);
    dfree(&d_L_);
    dfree(&d_cbuf_);
    dfree(&d_out_);
    dfree(&src_targets_.out);
    dfree(&src_targets_.grad);
    const size_t coef = static_cast<size_t>(k) * C * cheb_.n_pad;
    CHK(dalloc(&d_w_sorted_, static_cast<size_t>(k) * N));
    CHK(dalloc(&d_M_, coef, true));
    CHK(dalloc(&d_L_, coef, true));
    if (shared_basis_) {
        dfree(&d_Mc_);
        dfree(&d_Lc_);
        CHK(dalloc(&d_Mc_, static_cast<size_t>(k) * C * basis_pad_, true));
        CHK(dalloc(&d_Lc_, static_cast<size_t>(k) * C * basis_pad_, true));
    }
    // the M2L intermediate: as many right-hand sides per pass as fit the budget (at least one); absent pairs stay 0
    m2l_rhs_chunk_ = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(k, m2l_budget_bytes_ / 8 / std::max<int64_t>(cbuf_batch_len_, 1))));
    CHK(dalloc(&d_cbuf_, static_cast<size_t>(m2l_rhs_chunk_) * std::max<int64_t>(cbuf_batch_len_, 1), true));
    CHK(dalloc(&d_out_, static_cast<size_t>(k) * N));
    CHK(dalloc(&src_targets_.out, static_cast<size_t>(k) * N));
    k_cap_ = k;
    return BBFMM_OK;
}

// M2P jobs of one leaf: its W list cut into chunks so that a few big leaves still fill the chip.
// (whole = BBFMM_FLAG_DETERMINISTIC: one job per leaf, so that no two jobs add to the same target)
static void add_w_jobs(const HostTree &t, int32_t c, int32_t tb, int32_t te, std::vector<int32_t> *wtb,
                       std::vector<int32_t> *wte, std::vector<int64_t> *wb, std::vector<int64_t> *we, bool whole = false) {
    const int64_t kChunk = whole ? std::max<int64_t>(t.w.ptr[c + 1] - t.w.ptr[c], 1) : 8;
    for (int64_t q = t.w.ptr[c]; q < t.w.ptr[c + 1]; q += kChunk) {
        wtb->push_back(tb);
        wte->push_back(te);
        wb->push_back(q);
        we->push_back(std::min(q + kChunk, t.w.ptr[c + 1]));
    }
}

// targets = sources: jobs and ranges come straight from the tree
int FmmTree::build_source_target_set() {
    const HostTree &t = tree_;
    TargetSet &ts = src_targets_;
    ts.m = t.n_points;
    for (int a = 0; a < 3; ++a) ts.xyz_ptr[a] = src_ptr_[a];
    ts.perm = d_order_;
    std::vector<int32_t> jc, tb, te, wtb, wte;
    std::vector<int64_t> wb, we;
    for (int32_t c : src_leaves_) {
        jc.push_back(c);
        tb.push_back(static_cast<int32_t>(t.pt_begin[c]));
        te.push_back(static_cast<int32_t>(t.pt_end[c]));
        add_w_jobs(t, c, tb.back(), te.back(), &wtb, &wte, &wb, &we, deterministic_);
    }
    ts.n_jobs = static_cast<int>(jc.size());
    ts.n_w_jobs = static_cast<int>(wtb.size());
    CHK(dupload(&ts.job_cell, jc));
    CHK(dupload(&ts.tgt_begin, tb));
    CHK(dupload(&ts.tgt_end, te));
    CHK(dupload(&ts.w_tgt_begin, wtb));
    CHK(dupload(&ts.w_tgt_end, wte));
    CHK(dupload(&ts.w_begin, wb));
    CHK(dupload(&ts.w_end, we));
    CHK(build_sym_runs(&ts, jc, 0, t.n_points));
    return BBFMM_OK;
}

// Run lists of the symmetric P2P for the target leaves `job_cells` (in job order) when the targets are the
// sorted sources [pb, pe): the leaf itself and the U points outside [pb, pe) one-sided, the U points after
// the leaf inside the range two-sided; the U points before the leaf inside the range belong to those leaves'
// own jobs.  (U lists are symmetric: linear_tree.rs:295-364 collects adjacent leaves from both sides.)
int FmmTree::build_sym_runs(TargetSet *ts, const std::vector<int32_t> &job_cells, int64_t pb, int64_t pe,
                            const std::vector<uint8_t> *part_active) {
    const HostTree &t = tree_;
    // Two kinds of jobs over the same run lists: a leaf of at most p2p_sym_wave_rows() rows is ONE job of the
    // wave-per-job kernel (no barriers, columns in registers: faster where a leaf's work is small); bigger leaves go
    // in chunks of at most p2p_sym_rows_per_job() rows to the workgroup-per-job kernel, whose per-job overhead is
    // spread over eight waves (faster there).
    const int64_t max_rows = p2p_sym_rows_per_job(), wave_rows = p2p_sym_wave_rows();
    const int64_t nj_cells = static_cast<int64_t>(job_cells.size());
    // per chunk of leaves into local buffers (threads), concatenated in order
    constexpr int64_t kChunkS = 2048;
    const int64_t nch = (nj_cells + kChunkS - 1) / kChunkS;
    struct Part {
        std::vector<int32_t> runs, tb, te, wtb, wte;
        std::vector<int64_t> range, wrange; // run ranges relative to the part's first run
    };
    std::vector<Part> parts(static_cast<size_t>(std::max<int64_t>(nch, 1)));
    parallel_for_chunks(nj_cells, kChunkS, [&](int64_t lo, int64_t hi) {
        for (int64_t c0 = lo; c0 < hi; c0 += kChunkS) {
            Part &P = parts[static_cast<size_t>(c0 / kChunkS)];
            auto add = [&](int64_t b, int64_t e, int two) {
                if (e <= b) return;
                P.runs.push_back(static_cast<int32_t>(b));
                P.runs.push_back(static_cast<int32_t>(e));
                P.runs.push_back(two);
            };
            for (int64_t j = c0; j < std::min(hi, c0 + kChunkS); ++j) {
                const int32_t c = job_cells[static_cast<size_t>(j)];
                const int64_t a0 = t.pt_begin[c], a1 = t.pt_end[c];
                const int64_t first = static_cast<int64_t>(P.runs.size() / 3);
                add(a0, a1, 0); // self interaction included (bbfmm.rs:1162-1251)
                for (int64_t r = u_runs_.ptr[c]; r < u_runs_.ptr[c + 1]; ++r) {
                    const int64_t b = u_runs_.idx[2 * r], e = u_runs_.idx[2 * r + 1];
                    add(b, std::min({e, a0, pb}), 0);         // before the leaf, another rank's
                    add(std::max(b, a1), std::min(e, pe), 1); // after the leaf, inside the range
                    add(std::max({b, a1, pe}), e, 0);         // after the leaf, another rank's
                }
                const int64_t last = static_cast<int64_t>(P.runs.size() / 3);
                const int64_t na = a1 - a0;
                if (na <= wave_rows) { // one wave takes the whole leaf
                    P.wtb.push_back(static_cast<int32_t>(a0 - pb));
                    P.wte.push_back(static_cast<int32_t>(a1 - pb));
                    P.wrange.push_back(first);
                    P.wrange.push_back(last);
                } else { // the leaf's rows in equal chunks of at most max_rows
                    const int64_t nj = (na + max_rows - 1) / max_rows;
                    for (int64_t i = 0; i < nj; ++i) {
                        P.tb.push_back(static_cast<int32_t>(a0 - pb + na * i / nj));
                        P.te.push_back(static_cast<int32_t>(a0 - pb + na * (i + 1) / nj));
                        P.range.push_back(first);
                        P.range.push_back(last);
                    }
                }
            }
        }
    });
    std::vector<int64_t> range, wrange;
    std::vector<int32_t> runs, tb, te, wtb, wte;
    {
        size_t nr = 0, njobs = 0, nwjobs = 0;
        for (const Part &P : parts) nr += P.runs.size(), njobs += P.tb.size(), nwjobs += P.wtb.size();
        runs.reserve(nr);
        tb.reserve(njobs);
        te.reserve(njobs);
        range.reserve(2 * njobs);
        wtb.reserve(nwjobs);
        wte.reserve(nwjobs);
        wrange.reserve(2 * nwjobs);
        for (const Part &P : parts) {
            const int64_t base = static_cast<int64_t>(runs.size() / 3);
            runs.insert(runs.end(), P.runs.begin(), P.runs.end());
            tb.insert(tb.end(), P.tb.begin(), P.tb.end());
            te.insert(te.end(), P.te.begin(), P.te.end());
            for (int64_t v : P.range) range.push_back(base + v);
            wtb.insert(wtb.end(), P.wtb.begin(), P.wtb.end());
            wte.insert(wte.end(), P.wte.begin(), P.wte.end());
            for (int64_t v : P.wrange) wrange.push_back(base + v);
        }
    }
    ts->n_wx_jobs = 0;
    const bool whole = pb == 0 && pe == t.n_points;
    if ((whole || part_active) && !t.w.idx.empty() &&
        static_cast<int64_t>(t.n_cells()) * cheb_.n_pad < (int64_t(1) << 31)) { // M2P + P2L fused
        // Whole source set: every leaf with a W list.  A partition: its own leaves (row sums = M2P of its targets; the
        // column sums that fall on cells outside its subtree are never read) and the leaves outside whose W list holds
        // a cell of its subtree (column sums = P2L into that cell; their row sums are dropped by the kernel's output
        // window).  X = W^T (linear_tree.rs:388-392), so this covers the X lists of the partition's cells.
        std::vector<int32_t> wtb, wte;
        std::vector<int64_t> wr;
        const std::vector<int32_t> &cand = whole ? job_cells : src_leaves_;
        for (size_t j = 0; j < cand.size(); ++j) {
            const int32_t c = cand[j];
            if (t.w.ptr[c + 1] == t.w.ptr[c]) continue;
            if (!whole) {
                const bool own = t.pt_begin[c] >= pb && t.pt_begin[c] < pe;
                bool feeds = false;
                for (int64_t q = t.w.ptr[c]; q < t.w.ptr[c + 1] && !own && !feeds; ++q) feeds = (*part_active)[static_cast<size_t>(t.w.idx[q])] != 0;
                if (!own && !feeds) continue;
            }
            // jobs = (row chunk of the leaf) x (chunk of its W list): a nearly uniform tree has a few dozen coarse
            // leaves with long W lists (10M uniform points: 90 leaves of 250 points, about 100 W cells each), and whole-list
            // jobs would be a handful of long workgroups (0.96 ms for 0.6e9 kernel evaluations); both sums are atomic
            const int64_t max_rows_wx = wx_sym_rows_per_job(), max_cells_wx = 16;
            const int64_t a0 = t.pt_begin[c], na = t.pt_end[c] - a0, nj = (na + max_rows_wx - 1) / max_rows_wx;
            const int64_t w0 = t.w.ptr[c], nw = t.w.ptr[c + 1] - w0, nwj = (nw + max_cells_wx - 1) / max_cells_wx;
            for (int64_t i = 0; i < nj; ++i)
                for (int64_t jw = 0; jw < nwj; ++jw) {
                    wtb.push_back(static_cast<int32_t>(a0 + na * i / nj));
                    wte.push_back(static_cast<int32_t>(a0 + na * (i + 1) / nj));
                    wr.push_back(w0 + nw * jw / nwj);
                    wr.push_back(w0 + nw * (jw + 1) / nwj);
                }
        }
        ts->n_wx_jobs = static_cast<int>(wtb.size());
        CHK(dupload(&ts->wx_tb, wtb));
        CHK(dupload(&ts->wx_te, wte));
        CHK(dupload(&ts->wx_range, wr));
    }
    ts->n_symw_jobs = static_cast<int>(wtb.size());
    CHK(dupload(&ts->symw_tb, wtb));
    CHK(dupload(&ts->symw_te, wte));
    CHK(dupload(&ts->symw_ptr, wrange));
    ts->n_sym_jobs = static_cast<int>(tb.size());
    CHK(dupload(&ts->sym_tb, tb));
    CHK(dupload(&ts->sym_te, te));
    CHK(dupload(&ts->sym_ptr, range));
    CHK(dupload(&ts->sym_runs, runs));
    ts->sym = true;
    ts->sym_off = static_cast<int32_t>(pb);
    return BBFMM_OK;
}

int FmmTree::build_target_set(const double *x, int64_t m, int64_t ldx, TargetSet *ts, int64_t *bad_point_index,
                              std::vector<int32_t> *leaves_out) {
    static const int64_t min_rows = [] {
        const char *e = std::getenv("BBFMM_DEVICE_TARGETS_MIN");
        return e ? std::atoll(e) : int64_t(-1);
    }();
    if (m >= (min_rows >= 0 ? min_rows : device_targets_min_) && m > 0)
        return build_target_set_device(x, m, ldx, ts, bad_point_index, leaves_out);
    return build_target_set_host(x, m, ldx, ts, bad_point_index, leaves_out);
}

// points_to_leaves, the stable grouping by leaf and the coordinate gather as kernels (targets.hip); the
// host keeps the per-leaf part (M2P jobs from the W lists).  Same target set as the host path.
int FmmTree::build_target_set_device(const double *x, int64_t m, int64_t ldx, TargetSet *ts, int64_t *bad_point_index,
                                     std::vector<int32_t> *leaves_out) {
    const HostTree &t = tree_;
    if (!lk_ready_) {
        CHK(dupload(&d_tab_keys_, t.table.raw_keys()));
        CHK(dupload(&d_tab_vals_, t.table.raw_vals()));
        CHK(dupload(&d_is_leaf_, t.is_leaf));
        lk_.keys = d_tab_keys_.p;
        lk_.vals = d_tab_vals_.p;
        lk_.mask = t.table.mask();
        lk_.is_leaf = d_is_leaf_.p;
        lk_.d = d_;
        lk_.depth = t.depth;
        lk_.side = get_side_length(t.radius, static_cast<uint64_t>(t.depth)); // linear_tree.rs:495
        for (int a = 0; a < d_; ++a) lk_.disp[a] = t.center[a] - t.radius;
        lk_ready_ = true;
    }
    const int64_t C = t.n_cells();
    int end_bit = 1;
    while ((int64_t(1) << end_bit) < C) ++end_bit;
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t sm = static_cast<size_t>(m);
    const size_t o_x = 0, o_cell = o_x + up(sm * 8 * d_), o_sorted = o_cell + up(sm * 4), o_heads = o_sorted + up(sm * 4),
                 o_scal = o_heads + up(sm), o_temp = o_scal + 256;
    const size_t temp_bytes = group_targets_temp_bytes(m, end_bit);
    const size_t need = o_temp + temp_bytes;
    if (need > d_tscratch_.n) {
        dfree(&d_tscratch_);
        CHK(dalloc(&d_tscratch_, need + need / 4));
    }
    uint8_t *base = d_tscratch_.p;
    double *xin[3] = {nullptr, nullptr, nullptr};
    for (int a = 0; a < d_; ++a) {
        xin[a] = reinterpret_cast<double *>(base + o_x) + static_cast<size_t>(a) * sm;
        HIPCHK(hipMemcpyAsync(xin[a], x + a * ldx, sm * sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    int32_t *cell = reinterpret_cast<int32_t *>(base + o_cell), *sorted = reinterpret_cast<int32_t *>(base + o_sorted);
    uint8_t *heads = base + o_heads;
    unsigned long long *d_bad = reinterpret_cast<unsigned long long *>(base + o_scal);
    int32_t *d_runs = reinterpret_cast<int32_t *>(base + o_scal + 8);
    HIPCHK(hipMemsetAsync(d_bad, 0xFF, sizeof(unsigned long long), stream_));
    ts->m = m;
    const size_t cap = static_cast<size_t>(std::min<int64_t>(m, C));
    CHK(talloc(&ts->perm, sm));
    CHK(talloc(&ts->job_cell, cap));
    CHK(talloc(&ts->tgt_begin, cap));
    CHK(talloc(&ts->tgt_end, cap));
    for (int a = 0; a < 3; ++a) {
        if (a < d_) {
            CHK(talloc(&ts->xyz[a], sm));
            ts->xyz_ptr[a] = ts->xyz[a].p;
        } else if (sm > d_zero_axis_.n) {
            CHK(talloc(&ts->xyz[a], sm, true));
            ts->xyz_ptr[a] = ts->xyz[a].p;
        } else {
            ts->xyz_ptr[a] = d_zero_axis_.p;
        }
    }
    launch_points_to_leaves(lk_, xin[0], xin[1], xin[2], m, cell, d_bad, stream_);
    const int grc = group_targets(cell, m, end_bit, sorted, ts->perm.p, heads, ts->job_cell.p, ts->tgt_begin.p,
                                  ts->tgt_end.p, d_runs, base + o_temp, temp_bytes, stream_);
    if (grc != 0) return hip_fail(static_cast<hipError_t>(grc), "group targets by leaf");
    launch_gather_targets(xin[0], xin[1], xin[2], ts->perm.p, m, d_ > 0 ? ts->xyz[0].p : nullptr,
                          d_ > 1 ? ts->xyz[1].p : nullptr, d_ > 2 ? ts->xyz[2].p : nullptr, stream_);
    struct {
        unsigned long long bad;
        int32_t runs, pad;
    } scal;
    HIPCHK(hipMemcpyAsync(&scal, d_bad, 16, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    if (scal.bad != ~0ull) {
        if (bad_point_index) *bad_point_index = static_cast<int64_t>(scal.bad);
        return fail(BBFMM_POINT_OUTSIDE_TREE, "FMM evaluation failed: target point at row " + std::to_string(scal.bad) +
                                                  " lies outside the tree extents");
    }
    ts->n_jobs = scal.runs;
    std::vector<int32_t> wtb, wte;
    std::vector<int64_t> wb, we;
    if (leaves_out || !t.w.idx.empty()) {
        const size_t nj = static_cast<size_t>(scal.runs);
        std::vector<int32_t> jc(nj), tb(nj), te(nj);
        if (nj) {
            HIPCHK(hipMemcpyAsync(jc.data(), ts->job_cell.p, nj * 4, hipMemcpyDeviceToHost, stream_));
            if (!t.w.idx.empty()) {
                HIPCHK(hipMemcpyAsync(tb.data(), ts->tgt_begin.p, nj * 4, hipMemcpyDeviceToHost, stream_));
                HIPCHK(hipMemcpyAsync(te.data(), ts->tgt_end.p, nj * 4, hipMemcpyDeviceToHost, stream_));
            }
            HIPCHK(hipStreamSynchronize(stream_));
        }
        if (!t.w.idx.empty())
            for (size_t j = 0; j < nj; ++j) add_w_jobs(t, jc[j], tb[j], te[j], &wtb, &wte, &wb, &we, deterministic_);
        if (leaves_out) leaves_out->swap(jc);
    }
    ts->n_w_jobs = static_cast<int>(wtb.size());
    CHK(tupload(&ts->w_tgt_begin, wtb));
    CHK(tupload(&ts->w_tgt_end, wte));
    CHK(tupload(&ts->w_begin, wb));
    CHK(tupload(&ts->w_end, we));
    return BBFMM_OK;
}

int FmmTree::build_target_set_host(const double *x, int64_t m, int64_t ldx, TargetSet *ts, int64_t *bad_point_index,
                                   std::vector<int32_t> *leaves_out) {
    const HostTree &t = tree_;
    std::vector<int32_t> cell(static_cast<size_t>(m));
    const int64_t bad = points_to_leaves(t, x, m, ldx, cell.data());
    if (bad >= 0) {
        if (bad_point_index) *bad_point_index = bad;
        return fail(BBFMM_POINT_OUTSIDE_TREE, "FMM evaluation failed: target point at row " + std::to_string(bad) +
                                                  " lies outside the tree extents");
    }
    // group rows by leaf (ascending cell index), ascending rows inside a leaf (linear_tree.rs:522-534)
    const int64_t C = t.n_cells();
    std::vector<int32_t> leaves, jc, tb, te, wtb, wte;
    std::vector<int64_t> wb, we;
    std::vector<int32_t> perm(static_cast<size_t>(m));
    if (m * 8 < C) { // a small batch (isosurfacing): sort the rows instead of walking all cells
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return cell[a] < cell[b]; });
        for (int64_t i = 0; i < m;) {
            const int32_t c = cell[perm[i]];
            int64_t e = i + 1;
            while (e < m && cell[perm[e]] == c) ++e;
            leaves.push_back(c);
            jc.push_back(c);
            tb.push_back(static_cast<int32_t>(i));
            te.push_back(static_cast<int32_t>(e));
            add_w_jobs(t, c, tb.back(), te.back(), &wtb, &wte, &wb, &we, deterministic_);
            i = e;
        }
    } else {
        std::vector<int64_t> cnt(static_cast<size_t>(C) + 1, 0);
        for (int64_t i = 0; i < m; ++i) ++cnt[cell[i] + 1];
        for (int64_t c = 0; c < C; ++c)
            if (cnt[c + 1] > 0) leaves.push_back(static_cast<int32_t>(c));
        std::vector<int64_t> start(static_cast<size_t>(C), 0);
        int64_t cur = 0;
        for (int32_t c : leaves) {
            start[c] = cur;
            jc.push_back(c);
            tb.push_back(static_cast<int32_t>(cur));
            cur += cnt[c + 1];
            te.push_back(static_cast<int32_t>(cur));
            add_w_jobs(t, c, tb.back(), te.back(), &wtb, &wte, &wb, &we, deterministic_);
        }
        for (int64_t i = 0; i < m; ++i) perm[start[cell[i]]++] = static_cast<int32_t>(i);
    }
    ts->m = m;
    std::vector<double> tmp(static_cast<size_t>(m));
    for (int a = 0; a < 3; ++a) {
        if (a < d_) {
            for (int64_t i = 0; i < m; ++i) tmp[i] = x[a * ldx + perm[i]];
            CHK(tupload(&ts->xyz[a], tmp));
            ts->xyz_ptr[a] = ts->xyz[a].p;
        } else {
            if (static_cast<size_t>(m) > d_zero_axis_.n) {
                CHK(talloc(&ts->xyz[a], static_cast<size_t>(m), true));
                ts->xyz_ptr[a] = ts->xyz[a].p;
            } else {
                ts->xyz_ptr[a] = d_zero_axis_.p;
            }
        }
    }
    CHK(tupload(&ts->perm, perm));
    ts->n_jobs = static_cast<int>(jc.size());
    ts->n_w_jobs = static_cast<int>(wtb.size());
    CHK(tupload(&ts->job_cell, jc));
    CHK(tupload(&ts->tgt_begin, tb));
    CHK(tupload(&ts->tgt_end, te));
    CHK(tupload(&ts->w_tgt_begin, wtb));
    CHK(tupload(&ts->w_tgt_end, wte));
    CHK(tupload(&ts->w_begin, wb));
    CHK(tupload(&ts->w_end, we));
    if (leaves_out) leaves_out->swap(leaves);
    return BBFMM_OK;
}

void FmmTree::free_target_set(TargetSet *ts) {
    for (int a = 0; a < 3; ++a) dfree(&ts->xyz[a]);
    dfree(&ts->perm);
    dfree(&ts->job_cell);
    dfree(&ts->tgt_begin);
    dfree(&ts->tgt_end);
    dfree(&ts->w_tgt_begin);
    dfree(&ts->w_tgt_end);
    dfree(&ts->w_begin);
    dfree(&ts->w_end);
    dfree(&ts->out);
    dfree(&ts->grad);
    dfree(&ts->sym_tb);
    dfree(&ts->sym_te);
    dfree(&ts->sym_ptr);
    dfree(&ts->sym_runs);
    dfree(&ts->symw_tb);
    dfree(&ts->symw_te);
    dfree(&ts->symw_ptr);
    ts->n_symw_jobs = 0;
    dfree(&ts->wx_tb);
    dfree(&ts->wx_te);
    dfree(&ts->wx_range);
    ts->n_wx_jobs = 0;
    ts->sym = false;
}

int FmmTree::upload_weights(const double *w, int64_t rows, int k, int64_t ldw) {
    const int64_t N = tree_.n_points;
    if (!w || rows < N || ldw < rows || k < 1) return fail(BBFMM_BAD_ARGUMENT, "weights must be rows x k with rows >= N");
    CHK(ensure_rhs_capacity(k));
    if (static_cast<size_t>(k) * N > d_w_in_.n) {
        dfree(&d_w_in_);
        CHK(dalloc(&d_w_in_, static_cast<size_t>(k) * N));
    }
    // only rows < N are read (bbfmm.rs:704-708)
    HIPCHK(hipMemcpy2DAsync(d_w_in_.p, N * sizeof(double), w, ldw * sizeof(double), N * sizeof(double), k,
                            hipMemcpyHostToDevice, stream_));
    phase_begin();
    launch_gather_weights(d_w_in_.p, N, k, d_order_.p, N, d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    return BBFMM_OK;
}

int FmmTree::ensure_pinned(size_t n) {
    if (n <= h_pin_n_) return BBFMM_OK;
    if (h_pin_) (void)hipHostFree(h_pin_);
    h_pin_ = nullptr;
    h_pin_n_ = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h_pin_), n * sizeof(double), hipHostMallocDefault));
    h_pin_n_ = n;
    return BBFMM_OK;
}

// Host rows -> d_w_in_.  The caller's memory is pageable: it is staged through the pinned buffer in pieces, and
// the thread that staged a piece queues its copy to the device at once, so PCIe runs beside the staging of the
// other pieces (one memcpy pass + one 80 MB transfer at 10M points took 5 ms back to back).
int FmmTree::stage_weights_to_device(const double *w, int64_t n) {
    CHK(ensure_pinned(static_cast<size_t>(2 * tree_.n_points)));
    if (static_cast<size_t>(n) > d_w_in_.n) {
        dfree(&d_w_in_);
        CHK(dalloc(&d_w_in_, static_cast<size_t>(n)));
    }
    double *pin_in = h_pin_;
    double *dst = d_w_in_.p;
    std::atomic<int> err{0};
    parallel_for_chunks(n, kHostPiece, [&](int64_t b, int64_t e) {
        bind_device();
        std::memcpy(pin_in + b, w + b, static_cast<size_t>(e - b) * sizeof(double));
        const hipError_t r = hipMemcpyAsync(dst + b, pin_in + b, static_cast<size_t>(e - b) * sizeof(double), hipMemcpyHostToDevice, stream_);
        if (r != hipSuccess) err.store(static_cast<int>(r));
    });
    if (err.load() != 0) return hip_fail(static_cast<hipError_t>(err.load()), "hipMemcpyAsync(weights)");
    return BBFMM_OK;
}

// upward_pass (bbfmm.rs:666-688)
int FmmTree::upward(int k, const DownwardPlan *dp) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    const bool part = dp && dp->restrict_upward;
    // reset_multipole_coefficients (bbfmm.rs:619-624): M is zero-initialised once; P2M and M2M assign
    // every leaf with sources and every parent, the other entries are never written.  A partition computes
    // only its share (see DownwardPlan): the coarse prefix is zeroed first because the coarse leaves other ranks
    // own must enter the all-reduce as zeros; fine cells it does not need keep stale values nobody reads.
    phase_begin();
    if (part) {
        for (int j = 0; j < k && dp->coarse_cells > 0; ++j)
            HIPCHK(hipMemsetAsync(d_M_.p + static_cast<size_t>(j) * C * cheb_.n_pad, 0,
                                  static_cast<size_t>(dp->coarse_cells) * cheb_.n_pad * sizeof(double), stream_));
        launch_p2m(cheb_, src_ptr_, d_w_sorted_.p, t.n_points, k, C, dp->d_up_leaves.p, static_cast<int>(dp->up_leaves_h.size()),
                   d_pt_begin_.p, d_pt_end_.p, d_centers_.p, d_lengths_.p, d_M_.p, stream_);
    } else {
        launch_p2m(cheb_, src_ptr_, d_w_sorted_.p, t.n_points, k, C, d_src_leaves_.p, static_cast<int>(src_leaves_.size()),
                   d_pt_begin_.p, d_pt_end_.p, d_centers_.p, d_lengths_.p, d_M_.p, stream_);
    }
    phase_end(kPhP2M);
    phase_begin();
    for (int level = t.depth - 1; level >= 1; --level) { // (1..depth).rev(), bbfmm.rs:675
        const int rc = part ? launch_m2m(cheb_, k, C, dp->d_up_parents[level].p, static_cast<int>(dp->up_parents_h[level].size()),
                                         dp->d_part_child_ptr.p, dp->d_part_child_idx.p, d_octant_.p, d_M_.p, stream_)
                            : launch_m2m(cheb_, k, C, d_m2m_parents_[level].p, static_cast<int>(m2m_parents_[level].size()),
                                         d_child_ptr_.p, d_child_idx_.p, d_octant_.p, d_M_.p, stream_);
        if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "M2M: dynamic LDS attribute");
    }
    phase_end(kPhM2M);
    HIPCHK(hipGetLastError());
    return BBFMM_OK;
}

// downward_pass (bbfmm.rs:778-857).  All cells are treated as "with targets": locals of
// cells without targets are never read by the leaf pass, so results are unchanged.
int FmmTree::downward(int k, const DownwardPlan *dp, const TargetSet *wx) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    // reset_local_coefficients (bbfmm.rs:627-632): tiles without any V-list entry are not written by
    // stage 2, and P2L / L2L add onto L
    HIPCHK(hipMemsetAsync(d_L_.p, 0, static_cast<size_t>(k) * C * cheb_.n_pad * sizeof(double), stream_));
    // a plan runs stage 1 on compact tiles of the sources its targets need, stage 2 on the tiles that
    // hold a cell with targets, P2L / L2L on the cells with targets (cells_with_targets, bbfmm.rs:468-480)
    const int m2l_len = shared_basis_ ? basis_pad_ : cheb_.n_pad;
    const double *m_in = shared_basis_ ? d_Mc_.p : d_M_.p;
    double *l_out = shared_basis_ ? d_Lc_.p : d_L_.p;
    if (shared_basis_) { // coordinates of every multipole in its level's basis; stage 2 leaves untouched tiles at 0
        phase_begin();
        launch_m2l_basis(d_basis_classes_.p, d_basis_tiles_c_.p, n_basis_tiles_, cheb_.n_pad, basis_pad_, k, C, d_M_.p, d_Mc_.p, stream_);
        HIPCHK(hipMemsetAsync(d_Lc_.p, 0, static_cast<size_t>(k) * C * basis_pad_ * sizeof(double), stream_));
        phase_end(kPhM2L1);
    }
    // The batches go through the bounded intermediate one after another (stage 1 fills the slots of the batch's
    // targets, stage 2 contracts them into L), m2l_rhs_chunk_ right-hand sides per pass.
    const int nb = static_cast<int>(m2l_batches_.size());
    for (int k0 = 0; k0 < k; k0 += m2l_rhs_chunk_) {
        const int kb = std::min(m2l_rhs_chunk_, k - k0);
        const double *m_chunk = m_in + static_cast<size_t>(k0) * C * m2l_len;
        double *l_chunk = l_out + static_cast<size_t>(k0) * C * m2l_len;
        for (int b = 0; b < nb; ++b) {
            const M2lBatch &mb = m2l_batches_[static_cast<size_t>(b)];
            const int t1_first = dp ? dp->batch_t1[4 * b] : mb.t1_first, t1_count = dp ? dp->batch_t1[4 * b + 1] : mb.t1_count;
            const int t2_first = dp ? dp->batch_t2[2 * b] : mb.t2_first, t2_count = dp ? dp->batch_t2[2 * b + 1] : mb.t2_count;
            if (t2_count == 0) continue; // no target of this batch is active: nobody reads its slots
            phase_begin();
            if (nb > 1)
                launch_m2l_zero_segments(d_m2l_zero_.p + 2 * m2l_zero_ptr_[static_cast<size_t>(b)],
                                         m2l_zero_ptr_[static_cast<size_t>(b) + 1] - m2l_zero_ptr_[static_cast<size_t>(b)], kb, d_cbuf_.p,
                                         cbuf_batch_len_, stream_);
            if (dp) { // whole-operator tiles, then the tiles of single column blocks (sources in the halo of the target set)
                launch_m2l_stage1(d_m2l_classes_.p, dp->d_tiles1.p + t1_first, dp->d_tile_idx.p, t1_count, m2l_len, m2l_slot_t_, kb,
                                  C, m_chunk, d_cbuf_.p, cbuf_batch_len_, stream_, false);
                launch_m2l_stage1(d_m2l_classes_.p, dp->d_tiles1.p + dp->batch_t1[4 * b + 2], dp->d_tile_idx.p, dp->batch_t1[4 * b + 3],
                                  m2l_len, m2l_slot_t_, kb, C, m_chunk, d_cbuf_.p, cbuf_batch_len_, stream_, true);
            } else
                launch_m2l_stage1(d_m2l_classes_.p, d_m2l_tiles1_.p + t1_first, d_tile_idx1_.p, t1_count, m2l_len, m2l_slot_t_, kb, C,
                                  m_chunk, d_cbuf_.p, cbuf_batch_len_, stream_, false);
            phase_end(kPhM2L1);
            phase_begin();
            if (dp)
                launch_m2l_stage2(d_m2l_classes_.p, dp->d_tiles2.p + t2_first, dp->d_tile_idx.p, t2_count, m2l_len, kb, C, d_cbuf_.p,
                                  cbuf_batch_len_, dp->d_qlist.p, l_chunk, stream_, !deterministic_);
            else
                launch_m2l_stage2(d_m2l_classes_.p, d_m2l_tiles2_.p + t2_first, nullptr, t2_count, m2l_len, kb, C, d_cbuf_.p,
                                  cbuf_batch_len_, d_m2l_qlist_.p, l_chunk, stream_, !deterministic_);
            phase_end(kPhM2L2);
        }
    }
    if (shared_basis_) { // back to the node values (every cell of level >= 2; cells above keep the zeros)
        phase_begin();
        launch_m2l_basis(d_basis_classes_.p, d_basis_tiles_e_.p, n_basis_tiles_, basis_pad_, cheb_.n_pad, k, C, d_Lc_.p, d_L_.p, stream_);
        phase_end(kPhM2L2);
    }
    phase_begin();
    if (t.adaptive && wx) { // targets = all sources: P2L and M2P share their kernel evaluations (X = W^T)
        launch_wx_sym(kernel_, cheb_, wx->n_wx_jobs, wx->wx_tb.p, wx->wx_te.p, wx->wx_range.p, d_w_idx_.p, d_centers_.p,
                      d_lengths_.p, src_ptr_, d_w_sorted_.p, t.n_points, k, d_M_.p, d_L_.p, C * cheb_.n_pad, wx->out.p,
                      static_cast<int64_t>(wx->m), wx->sym_off, static_cast<int>(wx->m), stream_);
    } else if (t.adaptive) {
        if (dp)
            launch_p2l(kernel_, cheb_, dp->n_x_jobs, dp->d_x_cells.p, dp->d_x_ptr.p, dp->d_x_runs.p, d_centers_.p,
                       d_lengths_.p, src_ptr_, d_w_sorted_.p, t.n_points, k, C, d_L_.p, stream_);
        else
            launch_p2l(kernel_, cheb_, static_cast<int>(x_cells_.size()), d_x_cells_.p, d_x_job_run_ptr_.p,
                       d_x_runs_.p, d_centers_.p, d_lengths_.p, src_ptr_, d_w_sorted_.p, t.n_points, k, C, d_L_.p,
                       stream_);
    }
    phase_end(kPhP2L);
    phase_begin();
    for (int level = 2; level <= t.depth; ++level) { // children of level-1.. cells (bbfmm.rs:834-856)
        const int rc = launch_l2l(cheb_, k, C, d_level_cells_[level].p, static_cast<int>(level_cells_[level].size()), d_parent_.p,
                                  d_octant_.p, dp ? dp->d_active.p : d_active_.p, d_L_.p, stream_);
        if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "L2L: dynamic LDS attribute");
    }
    phase_end(kPhL2L);
    HIPCHK(hipGetLastError());
    have_locals_ = dp == nullptr; // the whole-tree expansions are in L
    return BBFMM_OK;
}

// leaf_pass (bbfmm.rs:1089-1159) into ts.out / ts.grad (sorted order)
int FmmTree::leaf_pass(const TargetSet &ts, int k, bool with_grads) {
    CHK(leaf_pass_near(ts, k, with_grads, stream_, 3));
    return leaf_pass_far(ts, k, with_grads);
}

// P2P + M2P: need the weights and the multipoles only (can run beside the downward pass)
// parts: 1 = zero the outputs + P2P (needs the sorted weights), 2 = M2P (needs the multipoles)
int FmmTree::leaf_pass_near(const TargetSet &ts, int k, bool with_grads, hipStream_t st, int parts, bool wx_done) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    double *grad = with_grads ? ts.grad.p : nullptr;
    const bool timed = st == stream_;
    if (parts & 1) {
        if (!wx_done) // (the fused M2P + P2L pass has already added into the zeroed output)
            HIPCHK(hipMemsetAsync(ts.out.p, 0, static_cast<size_t>(k) * ts.m * sizeof(double), st));
        if (with_grads) HIPCHK(hipMemsetAsync(grad, 0, static_cast<size_t>(k) * d_ * ts.m * sizeof(double), st));
        DirectJobs jobs{ts.n_jobs, ts.job_cell.p, ts.tgt_begin.p, ts.tgt_end.p, d_u_run_ptr_.p, d_u_runs_.p};
        static const bool sym_on = [] {
            const char *e = std::getenv("BBFMM_P2P_SYM"); // 0: every ordered pair, as the reference loops
            return !e || std::atoi(e) != 0;
        }();
        if (timed) phase_begin();
        if (ts.sym && sym_on && !deterministic_ && !with_grads) // targets = sources: every unordered pair once, for all rhs
            launch_p2p_sym(kernel_, ts.n_sym_jobs, ts.sym_tb.p, ts.sym_te.p, ts.sym_ptr.p, ts.n_symw_jobs, ts.symw_tb.p, ts.symw_te.p,
                           ts.symw_ptr.p, ts.sym_runs.p, ts.sym_off, src_ptr_, d_w_sorted_.p, t.n_points, k, ts.out.p, ts.m, st);
        else
            launch_p2p(kernel_, d_, jobs, ts.xyz_ptr, ts.m, src_ptr_, d_w_sorted_.p, t.n_points, k, ts.out.p, grad, st);
        if (timed) phase_end(kPhP2P);
    }
    if (parts & 2) {
        if (timed) phase_begin();
        if (t.adaptive && !wx_done)
            launch_m2p(kernel_, cheb_, ts.n_w_jobs, ts.w_tgt_begin.p, ts.w_tgt_end.p, ts.w_begin.p, ts.w_end.p,
                       d_w_idx_.p, d_centers_.p, d_lengths_.p, ts.xyz_ptr, ts.m, k, C, d_M_.p, ts.out.p, grad, st);
        if (timed) phase_end(kPhM2P);
    }
    HIPCHK(hipGetLastError());
    return BBFMM_OK;
}

// L2P: needs the finished local expansions
int FmmTree::leaf_pass_far(const TargetSet &ts, int k, bool with_grads) {
    const int64_t C = tree_.n_cells();
    double *grad = with_grads ? ts.grad.p : nullptr;
    phase_begin();
    launch_l2p(cheb_, ts.n_jobs, ts.job_cell.p, ts.tgt_begin.p, ts.tgt_end.p, d_centers_.p, d_lengths_.p, ts.xyz_ptr,
               ts.m, k, C, d_L_.p, ts.out.p, grad, stream_);
    phase_end(kPhL2P);
    HIPCHK(hipGetLastError());
    return BBFMM_OK;
}

int FmmTree::set_weights(const double *w, int64_t rows, int k, int64_t ldw) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    CHK(upload_weights(w, rows, k, ldw));
    nrhs_ = k; // bbfmm.rs:384
    have_locals_ = locals_requested_ = false; // the stored local expansions belong to the old weights
    CHK(upward(k));
    HIPCHK(hipStreamSynchronize(stream_));
    return BBFMM_OK;
}

int FmmTree::set_local_coefficients(const double *w, int64_t rows, int k, int64_t ldw) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    if (nrhs_ == 0) return fail(BBFMM_BAD_ARGUMENT, "set_weights must be called first");
    if (k != nrhs_) return fail(BBFMM_BAD_ARGUMENT, "weights must have the column count given to set_weights");
    CHK(upload_weights(w, rows, k, ldw));
    CHK(downward(k));
    HIPCHK(hipStreamSynchronize(stream_));
    locals_requested_ = true;
    return BBFMM_OK;
}

int FmmTree::evaluate(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx,
                      double *out, int64_t ldo, double *grad, int64_t ldg, bool with_grads, bool leaves_only,
                      int64_t *bad_point_index) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    if (nrhs_ == 0) return fail(BBFMM_BAD_ARGUMENT, "set_weights must be called first");
    if (k != nrhs_) return fail(BBFMM_BAD_ARGUMENT, "weights must have the column count given to set_weights");
    if (m < 0 || (m > 0 && (!x || !out || ldx < m || ldo < m))) return fail(BBFMM_BAD_ARGUMENT, "bad target/output arrays");
    if (with_grads && m > 0 && (!grad || ldg < m)) return fail(BBFMM_BAD_ARGUMENT, "bad gradient array");
    if (leaves_only && !have_locals_) return fail(BBFMM_BAD_ARGUMENT, "set_local_coefficients must be called first");
    if (m >= (int64_t(1) << 31)) return fail(BBFMM_BAD_ARGUMENT, "more than 2^31-1 target points");
    TargetSet ts;
    std::vector<int32_t> target_leaves;
    arena_begin();
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    int rc = build_target_set(x, m, ldx, &ts, bad_point_index, leaves_only ? nullptr : &target_leaves); // points_to_keys, bbfmm.rs:455-465
    const auto t_targets = std::chrono::steady_clock::now();
    // w == NULL (leaves-only entry points): keep the weights already on the device, i.e. those the stored
    // local coefficients were computed from -- no N x k host-to-device copy per batch
    if (rc == BBFMM_OK && (w || !leaves_only)) rc = upload_weights(w, rows, k, ldw);
    DownwardPlan dplan;
    if (rc == BBFMM_OK && !leaves_only) {
        // downward pass over cells_with_targets (bbfmm.rs:468-480) when the targets are few; many
        // targets touch every cell anyway and the whole-tree pass needs no plan.  Once
        // set_local_coefficients has stored the whole-tree expansions (Leaves mode, rbf.rs:836-838)
        // the whole-tree pass is kept, so that evaluate_leaves stays valid afterwards.
        // (planning costs host time per call: measured break-even near N/100 targets at 10M sources)
        const bool restricted = m * 128 < tree_.n_points && !locals_requested_;
        if (restricted) rc = build_downward_plan(target_leaves, &dplan);
        if (rc == BBFMM_OK) rc = downward(k, restricted ? &dplan : nullptr);
    }
    if (rc == BBFMM_OK && with_grads && !kernel_supports_gradients(kernel_.id)) // bbfmm.rs:634-658
        rc = fail(BBFMM_KERNEL_NO_GRADIENTS,
                  "FMM evaluation failed: gradient evaluation requested but kernel does not support gradients");
    DevBuf<double> o_dev, g_dev;
    if (rc == BBFMM_OK && m > 0) {
        rc = talloc(&ts.out, static_cast<size_t>(k) * m);
        if (rc == BBFMM_OK && with_grads) rc = talloc(&ts.grad, static_cast<size_t>(k) * d_ * m);
        if (rc == BBFMM_OK) rc = leaf_pass(ts, k, with_grads);
        if (rc == BBFMM_OK) rc = talloc(&o_dev, static_cast<size_t>(k) * m);
        if (rc == BBFMM_OK) {
            phase_begin();
            launch_scatter_output(ts.out.p, m, k, ts.perm.p, o_dev.p, m, 0, stream_);
            phase_end(kPhScatter);
            hipError_t e = hipMemcpy2DAsync(out, ldo * sizeof(double), o_dev.p, m * sizeof(double), m * sizeof(double), k,
                                            hipMemcpyDeviceToHost, stream_);
            if (e != hipSuccess) rc = hip_fail(e, "copy values to host");
        }
        if (rc == BBFMM_OK && with_grads) {
            rc = talloc(&g_dev, static_cast<size_t>(k) * d_ * m);
            if (rc == BBFMM_OK) {
                launch_scatter_output(ts.grad.p, m, k * d_, ts.perm.p, g_dev.p, m, 0, stream_);
                hipError_t e = hipMemcpy2DAsync(grad, ldg * sizeof(double), g_dev.p, m * sizeof(double),
                                                m * sizeof(double), static_cast<size_t>(k) * d_, hipMemcpyDeviceToHost,
                                                stream_);
                if (e != hipSuccess) rc = hip_fail(e, "copy gradients to host");
            }
        }
    }
    if (stream_) {
        hipError_t e = hipStreamSynchronize(stream_);
        if (rc == BBFMM_OK && e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
    }
    dfree(&o_dev);
    dfree(&g_dev);
    free_target_set(&ts);
    free_downward_plan(&dplan);
    {
        const int arc = arena_end();
        if (rc == BBFMM_OK) rc = arc;
    }
    if (verbose) {
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm] evaluate: %lld targets, target set %.3f ms, rest (weights, passes, copies) %.3f ms\n",
                     static_cast<long long>(m), std::chrono::duration<double, std::milli>(t_targets - t_begin).count(),
                     std::chrono::duration<double, std::milli>(t_end - t_targets).count());
    }
    return rc;
}

int FmmTree::matvec_device(const double *d_w, int64_t ldw, int k, double *d_out, int64_t ldo, bool sync) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    const int64_t N = tree_.n_points;
    if (!d_w || !d_out || k < 1 || ldw < N || ldo < N) return fail(BBFMM_BAD_ARGUMENT, "bad device matvec arguments");
    CHK(ensure_rhs_capacity(k));
    nrhs_ = k;
    phase_begin();
    launch_gather_weights(d_w, ldw, k, d_order_.p, N, d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    if (have_part_ && part_targets_.out.n < static_cast<size_t>(k) * part_targets_.m) {
        dfree(&part_targets_.out);
        CHK(dalloc(&part_targets_.out, static_cast<size_t>(k) * part_targets_.m));
    }
    const TargetSet &ts = have_part_ ? part_targets_ : src_targets_;
    const DownwardPlan *plan = have_part_ ? &part_plan_ : nullptr;
    static const bool wx_on = [] {
        const char *e = std::getenv("BBFMM_WX_FUSED"); // 0: separate P2L and M2P kernels
        const char *e2 = std::getenv("BBFMM_P2P_SYM");
        return (!e || std::atoi(e) != 0) && (!e2 || std::atoi(e2) != 0);
    }();
    const bool wx = wx_on && !deterministic_ && !have_part_ && ts.sym && ts.n_wx_jobs > 0;
    // (a partitioned handle called on its own, without the exchange of matvec_partition_upward / _finish,
    // needs every multipole: the whole upward pass)
    CHK(upward(k, nullptr));
    if (wx) HIPCHK(hipMemsetAsync(ts.out.p, 0, static_cast<size_t>(k) * ts.m * sizeof(double), stream_));
    CHK(downward(k, plan, wx ? &ts : nullptr));
    CHK(leaf_pass_near(ts, k, false, stream_, 3, wx));
    CHK(leaf_pass_far(ts, k, false));
    phase_begin();
    launch_scatter_output(ts.out.p, ts.m, k, ts.perm.p, d_out, ldo, 0, stream_);
    phase_end(kPhScatter);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(stream_));
    return BBFMM_OK;
}

// Doubles per right-hand side of the coarse multipoles a partition exchanges (0: nothing to exchange).
int64_t FmmTree::partition_coarse_count() const {
    return have_part_ || (host_only_ && part_world_ > 1) ? part_plan_.coarse_cells * static_cast<int64_t>(round_up(ops_.n, 32)) : 0;
}

// First half of the partitioned matvec: set_weights restricted to this rank's share of the upward pass
// (bbfmm.rs:383-401, 666-772 split by subtree), partial coarse multipoles packed rhs-major into d_coarse.
int FmmTree::matvec_partition_upward(const double *d_w, int64_t ldw, int k, double *d_coarse, hipStream_t comm_stream) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (!have_part_) return fail(BBFMM_BAD_ARGUMENT, "bbfmm_set_partition (world > 1) must be called first");
    const int64_t N = tree_.n_points, C = tree_.n_cells();
    const int64_t cnt = partition_coarse_count();
    if (!d_w || k < 1 || ldw < N || (cnt > 0 && !d_coarse)) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    CHK(ensure_rhs_capacity(k));
    nrhs_ = k;
    have_locals_ = locals_requested_ = false;
    phase_begin(); // only the weights this rank reads: its subtree and halo (the rest of w_sorted keeps stale values)
    launch_gather_weights_subset(d_w, ldw, k, d_order_.p, part_plan_.d_gather_pos.p, static_cast<int64_t>(part_plan_.gather_pos_h.size()), N,
                                 d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    if (part_targets_.out.n < static_cast<size_t>(k) * part_targets_.m) {
        dfree(&part_targets_.out);
        CHK(dalloc(&part_targets_.out, static_cast<size_t>(k) * part_targets_.m));
    }
    CHK(upward(k, &part_plan_));
    phase_begin();
    for (int j = 0; j < k && cnt > 0; ++j)
        HIPCHK(hipMemcpyAsync(d_coarse + static_cast<size_t>(j) * cnt, d_M_.p + static_cast<size_t>(j) * C * cheb_.n_pad,
                              static_cast<size_t>(cnt) * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    phase_end(kPhM2M);
    // The collective may start as soon as the packed multipoles are there: the caller's communication stream waits for
    // that point only, and the near field of the owned targets (P2P: needs the weights, not the multipoles) is queued
    // behind it, so that the all-reduce runs beside it instead of in front of it.
    if (comm_stream) {
        HIPCHK(hipEventRecord(ev_pack_, stream_));
        HIPCHK(hipStreamWaitEvent(comm_stream, ev_pack_, 0));
    }
    CHK(leaf_pass_near(part_targets_, k, false, stream_, 1));
    part_pending_k_ = k;
    return BBFMM_OK;
}

// Second half: d_coarse holds the sum over all ranks of what matvec_partition_upward packed (an all-reduce on this
// handle's stream, or any stream ordered with it); evaluate at the owned targets (bbfmm.rs:444-507 over this
// rank's cells_with_targets), owned rows of d_out written, the others untouched.
int FmmTree::partition_finish_core(const double *d_coarse, hipStream_t comm_stream, int *k_out) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (!have_part_ || part_pending_k_ < 1) return fail(BBFMM_BAD_ARGUMENT, "bbfmm_matvec_partition_upward must be called first");
    const int64_t C = tree_.n_cells();
    const int64_t cnt = partition_coarse_count();
    const int k = part_pending_k_;
    if (cnt > 0 && !d_coarse) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    part_pending_k_ = 0;
    if (comm_stream) { // the summed multipoles are ready when the communication stream gets here
        HIPCHK(hipEventRecord(ev_comm_, comm_stream));
        HIPCHK(hipStreamWaitEvent(stream_, ev_comm_, 0));
    }
    phase_begin();
    for (int j = 0; j < k && cnt > 0; ++j)
        HIPCHK(hipMemcpyAsync(d_M_.p + static_cast<size_t>(j) * C * cheb_.n_pad, d_coarse + static_cast<size_t>(j) * cnt,
                              static_cast<size_t>(cnt) * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    phase_end(kPhM2M);
    const TargetSet &ts = part_targets_;
    static const bool wx_on = [] {
        const char *e = std::getenv("BBFMM_WX_FUSED"); // 0: separate P2L and M2P kernels
        const char *e2 = std::getenv("BBFMM_P2P_SYM");
        return (!e || std::atoi(e) != 0) && (!e2 || std::atoi(e2) != 0);
    }();
    // one rhs: M2P of the owned targets and P2L into the subtree's cells share their kernel evaluations, as in the
    // unpartitioned matvec (the outputs were zeroed and P2P ran in the first half; the fused pass adds to them)
    const bool wx = wx_on && !deterministic_ && ts.sym && ts.n_wx_jobs > 0;
    CHK(downward(k, &part_plan_, wx ? &ts : nullptr));
    CHK(leaf_pass_near(ts, k, false, stream_, 2, wx)); // M2P unless the fused pass has done it
    CHK(leaf_pass_far(ts, k, false));
    *k_out = k;
    return BBFMM_OK;
}

int FmmTree::matvec_partition_finish(const double *d_coarse, double *d_out, int64_t ldo, bool sync, hipStream_t comm_stream) {
    if (!d_out || ldo < tree_.n_points) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    int k = 0;
    CHK(partition_finish_core(d_coarse, comm_stream, &k));
    const TargetSet &ts = part_targets_;
    phase_begin();
    launch_scatter_output(ts.out.p, ts.m, k, ts.perm.p, d_out, ldo, 0, stream_);
    phase_end(kPhScatter);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(stream_));
    return BBFMM_OK;
}

int FmmTree::matvec_partition_finish_sorted(const double *d_coarse, double *d_seg, int64_t ld, hipStream_t comm_stream) {
    if (!d_seg || ld < part_targets_.m) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    int k = 0;
    CHK(partition_finish_core(d_coarse, comm_stream, &k));
    const TargetSet &ts = part_targets_;
    phase_begin();
    if (ts.m > 0)
        HIPCHK(hipMemcpy2DAsync(d_seg, static_cast<size_t>(ld) * sizeof(double), ts.out.p, static_cast<size_t>(ts.m) * sizeof(double),
                                static_cast<size_t>(ts.m) * sizeof(double), static_cast<size_t>(k), hipMemcpyDeviceToDevice, stream_));
    phase_end(kPhScatter);
    HIPCHK(hipGetLastError());
    return BBFMM_OK;
}

int FmmTree::partition_scatter(const double *d_all, int first_part, int n_parts, int64_t m_max, int k, double *d_out, int64_t ldo) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (part_world_ < 2 || part_bounds_.size() != static_cast<size_t>(part_world_) + 1)
        return fail(BBFMM_BAD_ARGUMENT, "the handle has no partition");
    if (!d_all || !d_out || k < 1 || ldo < tree_.n_points || first_part < 0 || n_parts < 1 || first_part + n_parts > part_world_)
        return fail(BBFMM_BAD_ARGUMENT, "bad arguments of the gathered scatter");
    if (n_parts > kMaxScatterParts) return fail(BBFMM_UNSUPPORTED, "more gathered parts than the scatter kernel takes");
    ScatterParts sp;
    sp.n = n_parts;
    for (int r = 0; r <= n_parts; ++r) sp.bound[r] = part_bounds_[static_cast<size_t>(first_part + r)];
    for (int r = 0; r < n_parts; ++r)
        if (sp.bound[r + 1] - sp.bound[r] > m_max) return fail(BBFMM_BAD_ARGUMENT, "a part holds more rows than m_max");
    phase_begin();
    launch_scatter_parts(d_all, sp, m_max, k, d_order_.p, d_out, ldo, stream_);
    phase_end(kPhScatter);
    HIPCHK(hipGetLastError());
    return BBFMM_OK;
}

// target_indices = 0, 1, ..., N-1 (the finest Schwarz level): the plain all-rows product serves it
bool FmmTree::is_identity_subset(const int64_t *idx, int64_t n_idx) const {
    if (n_idx != tree_.n_points) return false;
    std::atomic<bool> same{true};
    parallel_for_chunks(n_idx, int64_t(1) << 18, [&](int64_t b, int64_t e) {
        for (int64_t j = b; j < e; ++j)
            if (idx[j] != j) {
                same.store(false, std::memory_order_relaxed);
                return;
            }
    });
    return same.load();
}

int FmmTree::prepare_target_subset(const int64_t *target_indices, int64_t n_target_indices) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (!target_indices || n_target_indices < 0) return fail(BBFMM_BAD_ARGUMENT, "bad target index array");
    CHK(ensure_pinned(static_cast<size_t>(2 * tree_.n_points)));
    CHK(ensure_rhs_capacity(1));
    if (is_identity_subset(target_indices, n_target_indices)) return BBFMM_OK;
    SubsetPlan *sp = nullptr;
    return subset_plan(target_indices, n_target_indices, &sp);
}

int FmmTree::fast_matrix_vector_product(const double *w, int64_t rows, int64_t basis_size,
                                        const int64_t *target_indices, int64_t n_target_indices, const double *poly,
                                        int64_t ldp, double nugget, double *result) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    const int64_t N = tree_.n_points;
    if (!w || !result || basis_size < 0 || rows != N + basis_size)
        return fail(BBFMM_BAD_ARGUMENT, "weights must have N + basis_size rows");
    if (poly && ldp < N) return fail(BBFMM_BAD_ARGUMENT, "polynomial matrix needs N rows");
    if (target_indices && is_identity_subset(target_indices, n_target_indices)) target_indices = nullptr; // all rows in order
    if (!target_indices) {
        // All sources (the FGMRES matvec, rbf.rs:105-117): the targets already live on the device.
        // Host traffic goes through one pinned staging buffer (pageable copies run at a fraction of
        // the PCIe rate), and the host loops are threaded.
        CHK(stage_weights_to_device(w, N));
        double *pin_out = h_pin_ + N;
        CHK(ensure_rhs_capacity(1));
        nrhs_ = 1;
        if (have_part_) // a partitioned handle fills its owned rows only; the others read as 0
            HIPCHK(hipMemsetAsync(d_out_.p, 0, static_cast<size_t>(N) * sizeof(double), stream_));
        CHK(matvec_device(d_w_in_.p, N, 1, d_out_.p, N, false)); // set_weights + evaluate, rbf.rs:1357-1364
        // The way back in pieces as well: an event behind each piece's copy, and the host threads add the nugget
        // and polynomial terms (rbf.rs:1366-1376) of a piece as soon as it has landed.
        const int64_t n_pieces = (N + kHostPiece - 1) / kHostPiece;
        while (static_cast<int64_t>(ev_out_.size()) < n_pieces) {
            hipEvent_t ev;
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ev_out_.push_back(ev);
        }
        for (int64_t pc = 0; pc < n_pieces; ++pc) {
            const int64_t b = pc * kHostPiece, e = std::min(N, b + kHostPiece);
            HIPCHK(hipMemcpyAsync(pin_out + b, d_out_.p + b, static_cast<size_t>(e - b) * sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipEventRecord(ev_out_[static_cast<size_t>(pc)], stream_));
        }
        HIPCHK(hipEventSynchronize(ev_out_[0])); // the product is done and the first piece is here: start the team
        std::atomic<int> err{0};
        parallel_for_chunks(N, kHostPiece, [&](int64_t b, int64_t e) { // chunks start at piece boundaries, ascending
            bind_device();
            for (int64_t pb = b; pb < e; pb += kHostPiece) {
                const hipError_t r = hipEventSynchronize(ev_out_[static_cast<size_t>(pb / kHostPiece)]);
                if (r != hipSuccess) {
                    err.store(static_cast<int>(r));
                    return;
                }
                const int64_t pe = std::min(e, pb + kHostPiece);
                for (int64_t i = pb; i < pe; ++i) {
                    double v = pin_out[i] + w[i] * nugget;
                    if (poly) {
                        double sacc = 0.0;
                        for (int64_t q = 0; q < basis_size; ++q) sacc += poly[q * ldp + i] * w[N + q];
                        v += sacc;
                    }
                    result[i] = v;
                }
            }
        });
        HIPCHK(hipStreamSynchronize(stream_));
        if (err.load() != 0) return hip_fail(static_cast<hipError_t>(err.load()), "hipEventSynchronize(result piece)");
        std::fill(result + N, result + rows, 0.0); // the last basis_size rows stay 0 (rbf.rs:1346)
        return BBFMM_OK;
    }
    // A subset of the sources (matvec_partial, rbf.rs:119-133): cached sorted targets, downward pass
    // restricted to the cells that carry them; everything else as above.
    SubsetPlan *sp = nullptr;
    CHK(subset_plan(target_indices, n_target_indices, &sp));
    const int64_t m = n_target_indices;
    std::fill(result, result + rows, 0.0); // rbf.rs:1346
    if (m == 0) return BBFMM_OK;
    CHK(stage_weights_to_device(w, N));
    double *pin_out = h_pin_ + N;
    CHK(ensure_rhs_capacity(1));
    nrhs_ = 1;
    phase_begin();
    launch_gather_weights(d_w_in_.p, N, 1, d_order_.p, N, d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    CHK(upward(1));            // set_weights: all sources contribute (rbf.rs:1357)
    CHK(downward(1, &sp->dp)); // evaluate at the subset (rbf.rs:1359-1364)
    CHK(leaf_pass(sp->ts, 1, false));
    launch_scatter_output(sp->ts.out.p, m, 1, sp->ts.perm.p, d_out_.p, m, 0, stream_);
    HIPCHK(hipMemcpyAsync(pin_out, d_out_.p, m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    parallel_for_chunks(m, int64_t(1) << 16, [&](int64_t b, int64_t e) { // rbf.rs:1366-1376
        for (int64_t j = b; j < e; ++j) {
            const int64_t i = target_indices[j];
            double r = pin_out[j] + w[i] * nugget;
            if (poly) {
                double sacc = 0.0;
                for (int64_t q = 0; q < basis_size; ++q) sacc += poly[q * ldp + i] * w[N + q];
                r += sacc;
            }
            result[i] = r;
        }
    });
    return BBFMM_OK;
}

// Restriction of the downward pass to the cells that carry targets (cells_with_targets,
// bbfmm.rs:468-480): M2L stage 2 on the tiles that hold such a cell, stage 1 on compact tiles (lists
// of class positions, 128 per tile) of the cells that are a V-list source of one, P2L on such cells.
// The host part also runs on BBFMM_FLAG_HOST_ONLY handles.
int FmmTree::build_downward_plan(const std::vector<int32_t> &target_leaves, DownwardPlan *dp, bool restrict_upward,
                                 int64_t own_b, int64_t own_e) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    dp->active.assign(static_cast<size_t>(C), 0);
    for (int32_t leaf : target_leaves) {
        int32_t c = leaf;
        while (c >= 0 && !dp->active[c]) {
            dp->active[c] = 1;
            c = t.parent[c];
        }
    }
    const std::vector<uint8_t> &active = dp->active;
    // class and position of every cell with M2L work; group of its class inside its level's batches
    std::vector<int32_t> cls_of(static_cast<size_t>(C), -1), pos_of(static_cast<size_t>(C), -1);
    for (size_t lc = 0; lc < m2l_host_.size(); ++lc) {
        const HostM2lClass &hc = m2l_host_[lc];
        for (size_t i = 0; i < hc.cells.size(); ++i) {
            cls_of[hc.cells[i]] = static_cast<int32_t>(lc);
            pos_of[hc.cells[i]] = static_cast<int32_t>(i);
        }
    }
    auto group_of_class = [&](int32_t lc) { return m2l_batches_[static_cast<size_t>(m2l_batch_of_class_[static_cast<size_t>(lc)])].group; };
    // needed[V]: V is a V-list source of an active cell; bit g: of an active cell whose class lies in group g of the level
    std::vector<uint8_t> needed(static_cast<size_t>(C), 0);
    auto flag = [](uint8_t *p) { __atomic_store_n(p, uint8_t(1), __ATOMIC_RELAXED); }; // threads may set the same flag
    parallel_for_chunks(C, 4096, [&](int64_t lo, int64_t hi) {
        for (int64_t B = lo; B < hi; ++B) {
            if (!active[B] || t.level[B] < 2 || cls_of[B] < 0) continue;
            const uint8_t bit = static_cast<uint8_t>(1u << group_of_class(cls_of[B]));
            for (int64_t q = t.v.ptr[B]; q < t.v.ptr[B + 1]; ++q) __atomic_fetch_or(&needed[t.v.idx[q]], bit, __ATOMIC_RELAXED);
        }
    });
    dp->tiles2_h.clear();
    dp->tiles1_h.clear();
    dp->tile_idx_h.clear();
    dp->qlist_h.clear();
    const size_t nb = m2l_batches_.size();
    std::vector<std::vector<M2lTileDesc>> t1b(nb), t2b(nb);
    // stage-1 source operators a plan uses: the class operator, or -- on a level cut into groups -- the group operators
    struct SrcOp {
        int32_t dev_class;
        const HostM2lClass *h;
        uint8_t bit;
    };
    std::vector<SrcOp> sops;
    std::vector<std::vector<int32_t>> sops_of_class(m2l_host_.size());
    for (size_t lc = 0; lc < m2l_host_.size(); ++lc) {
        if (m2l_host_[lc].cells.empty()) continue;
        if (m2l_group_ops_[lc].empty()) {
            sops_of_class[lc].push_back(static_cast<int32_t>(sops.size()));
            sops.push_back(SrcOp{static_cast<int32_t>(lc), &m2l_host_[lc], uint8_t(0xff)});
        } else {
            // (a class whose transfer vectors miss a group has no operator for it: index by the operator's group)
            sops_of_class[lc].assign(8, -1);
            for (int32_t v : m2l_group_ops_[lc]) {
                const int32_t dev = static_cast<int32_t>(m2l_host_.size()) + v;
                const int g = m2l_batches_[static_cast<size_t>(m2l_batch_of_class_[static_cast<size_t>(dev)])].group;
                sops_of_class[lc][static_cast<size_t>(g)] = static_cast<int32_t>(sops.size());
                sops.push_back(SrcOp{dev, &m2l_variants_[static_cast<size_t>(v)], static_cast<uint8_t>(1u << g)});
            }
        }
    }
    auto add_tiles = [&](const std::vector<uint8_t> &flags, uint8_t bit, int32_t dev_class, const std::vector<int32_t> &cells,
                         std::vector<M2lTileDesc> *tiles) {
        const size_t start = dp->tile_idx_h.size();
        for (size_t i = 0; i < cells.size(); ++i)
            if (flags[cells[i]] & bit) dp->tile_idx_h.push_back(static_cast<int32_t>(i));
        for (size_t f = start; f < dp->tile_idx_h.size(); f += kM2lTile) {
            M2lTileDesc td;
            std::memset(&td, 0, sizeof td);
            td.level_class = dev_class;
            td.first = static_cast<int32_t>(f);
            td.count = static_cast<int32_t>(std::min<size_t>(kM2lTile, dp->tile_idx_h.size() - f));
            td.pad = 1; // first indexes tile_idx (class positions)
            tiles->push_back(td);
        }
    };
    std::vector<int> tpos_of(static_cast<size_t>(ops_.n_vec));
    for (size_t lc = 0; lc < m2l_host_.size(); ++lc) {
        const HostM2lClass &hc = m2l_host_[lc];
        if (hc.cells.empty()) continue;
        std::vector<M2lTileDesc> &tiles2 = t2b[static_cast<size_t>(m2l_batch_of_class_[lc])];
        const size_t t2 = tiles2.size();
        add_tiles(active, uint8_t(0xff), static_cast<int32_t>(lc), hc.cells, &tiles2);
        // contraction steps (16 slot entries each) that hold a V-list entry of some cell of the tile
        std::fill(tpos_of.begin(), tpos_of.end(), -1);
        for (size_t pos = 0; pos < hc.tgt_tv.size(); ++pos) tpos_of[hc.tgt_tv[pos]] = static_cast<int>(pos);
        const auto &lops = ops_.m2l[hc.level];
        const int nq = hc.k_pad / 16;
        const int64_t n_t2 = static_cast<int64_t>(tiles2.size() - t2);
        std::vector<std::vector<uint16_t>> tile_q(static_cast<size_t>(n_t2));
        parallel_for(n_t2, 4, [&](int64_t k) {
            const M2lTileDesc &td = tiles2[t2 + static_cast<size_t>(k)];
            std::vector<uint8_t> act_k(static_cast<size_t>(nq), 0);
            for (int32_t i = 0; i < td.count; ++i) {
                const int64_t B = hc.cells[dp->tile_idx_h[td.first + i]];
                for (int64_t q = t.v.ptr[B]; q < t.v.ptr[B + 1]; ++q) {
                    const int tv = t.v_tidx[q];
                    const int pos = tv >= 0 && tv < ops_.n_vec ? tpos_of[tv] : -1;
                    if (pos < 0) continue;
                    const int a = hc.tgt_off[pos], b = a + lops[ops_.ref_lookup[tv]].rank;
                    for (int sq = a / 16; sq <= (b - 1) / 16; ++sq) act_k[sq] = 1;
                }
            }
            for (int sq = 0; sq < nq; ++sq)
                if (act_k[sq]) tile_q[static_cast<size_t>(k)].push_back(static_cast<uint16_t>(sq));
        });
        for (int64_t k = 0; k < n_t2; ++k) {
            M2lTileDesc &td = tiles2[t2 + static_cast<size_t>(k)];
            td.q_first = static_cast<int32_t>(dp->qlist_h.size());
            dp->qlist_h.insert(dp->qlist_h.end(), tile_q[static_cast<size_t>(k)].begin(), tile_q[static_cast<size_t>(k)].end());
            td.q_count = static_cast<int32_t>(dp->qlist_h.size()) - td.q_first;
        }
    }
    for (auto &tl : t2b) split_tile_tail(&tl, n_cu_);
    // Stage-1 tiles.  A needed source cell needs only the column blocks (kM2lS1Block stacked rows = a few transfer
    // vectors) that hold a transfer vector towards an ACTIVE target: all of them inside the target set, about half in
    // its three-cell halo (a partition's subtree: 31 % of the needed cells at eight ranks), a handful for scattered
    // targets.  The plan holds one tile per (column block, the sources that need it).  Measured at 10M points (stage 1
    // of one rank of 2 / 3 / 4 / 8): whole-operator tiles for every needed cell 9.79 / - / 5.64 / 3.28 ms; whole
    // operators for the cells that need at least 80 % of their blocks and per-block tiles for the rest 10.03 / 6.96 /
    // 5.34 / 2.94; per-block tiles throughout 9.75 / 6.72 / 5.12 / 2.72 -- thousands of short tiles leave no launch
    // tail, and a persistent walk over the blocks buys almost nothing (the unrestricted 10M-point stage 1 as per-block
    // tiles: 17.25 against 17.14 ms).  The analysis runs for partitions and for target sets under half of the cells;
    // denser sets take whole operators.
    std::vector<std::vector<M2lTileDesc>> t1s(nb); // per-block tiles
    int64_t n_active = 0;
    for (uint8_t a : active) n_active += a;
    const bool analyse = restrict_upward || n_active * 2 < C;
    if (!analyse) {
        for (const SrcOp &so : sops)
            add_tiles(needed, so.bit, so.dev_class, so.h->cells, &t1b[static_cast<size_t>(m2l_batch_of_class_[static_cast<size_t>(so.dev_class)])]);
    } else {
        std::vector<int64_t> bm_off(sops.size() + 1, 0); // per source operator: n_blk x n_cells flags
        for (size_t si = 0; si < sops.size(); ++si)
            bm_off[si + 1] = bm_off[si] + static_cast<int64_t>(sops[si].h->r_pad16 / kM2lS1Block) * static_cast<int64_t>(sops[si].h->cells.size());
        std::vector<uint8_t> bm(static_cast<size_t>(bm_off.back()), 0);
        std::vector<std::vector<int32_t>> spos(sops.size());
        for (size_t si = 0; si < sops.size(); ++si) {
            spos[si].assign(static_cast<size_t>(ops_.n_vec), -1);
            for (size_t pos = 0; pos < sops[si].h->src_tv.size(); ++pos) spos[si][sops[si].h->src_tv[pos]] = static_cast<int32_t>(pos);
        }
        parallel_for_chunks(C, 4096, [&](int64_t lo, int64_t hi) {
            for (int64_t B = lo; B < hi; ++B) {
                if (!active[B] || t.level[B] < 2 || cls_of[B] < 0) continue;
                const int gB = group_of_class(cls_of[B]);
                for (int64_t q = t.v.ptr[B]; q < t.v.ptr[B + 1]; ++q) {
                    const int32_t S = t.v.idx[q];
                    const int tv = t.v_tidx[q];
                    const int32_t lc = cls_of[S];
                    if (lc < 0 || tv < 0 || tv >= ops_.n_vec) continue;
                    const auto &cand = sops_of_class[static_cast<size_t>(lc)];
                    const int32_t si = cand.size() == 1 ? cand[0] : cand[static_cast<size_t>(gB)];
                    if (si < 0) continue;
                    const int32_t sp = spos[static_cast<size_t>(si)][tv];
                    if (sp < 0) continue;
                    const HostM2lClass &hs = *sops[static_cast<size_t>(si)].h;
                    if (hs.src_row1[sp] <= hs.src_row0[sp]) continue;
                    const int64_t nc = static_cast<int64_t>(hs.cells.size());
                    for (int zb = hs.src_row0[sp] / kM2lS1Block; zb <= (hs.src_row1[sp] - 1) / kM2lS1Block; ++zb)
                        flag(&bm[static_cast<size_t>(bm_off[static_cast<size_t>(si)] + zb * nc + pos_of[S])]);
                }
            }
        });
        for (size_t si = 0; si < sops.size(); ++si) {
            const HostM2lClass &hs = *sops[si].h;
            const int64_t nc = static_cast<int64_t>(hs.cells.size());
            const int n_blk = hs.r_pad16 / kM2lS1Block;
            const size_t bidx = static_cast<size_t>(m2l_batch_of_class_[static_cast<size_t>(sops[si].dev_class)]);
            // one tile per (column block, up to 128 of the sources that need it)
            for (int zb = 0; zb < n_blk; ++zb) {
                const size_t start = dp->tile_idx_h.size();
                const uint8_t *f = &bm[static_cast<size_t>(bm_off[si] + zb * nc)];
                for (int64_t i = 0; i < nc; ++i)
                    if (f[i]) dp->tile_idx_h.push_back(static_cast<int32_t>(i));
                for (size_t fst = start; fst < dp->tile_idx_h.size(); fst += kM2lTile) {
                    M2lTileDesc td;
                    std::memset(&td, 0, sizeof td);
                    td.level_class = sops[si].dev_class;
                    td.first = static_cast<int32_t>(fst);
                    td.count = static_cast<int32_t>(std::min<size_t>(kM2lTile, dp->tile_idx_h.size() - fst));
                    td.q_first = zb;
                    td.q_count = 1;
                    td.pad = 2;
                    t1s[bidx].push_back(td);
                }
            }
        }
    }
    dp->batch_t1.assign(4 * nb, 0); // per batch: whole-operator tiles (first, count), per-block tiles (first, count)
    dp->batch_t2.assign(2 * nb, 0);
    dp->n_tiles1_blocks = 0;
    for (size_t b = 0; b < nb; ++b) {
        dp->batch_t1[4 * b] = static_cast<int32_t>(dp->tiles1_h.size());
        dp->batch_t1[4 * b + 1] = static_cast<int32_t>(t1b[b].size());
        dp->tiles1_h.insert(dp->tiles1_h.end(), t1b[b].begin(), t1b[b].end());
        dp->batch_t1[4 * b + 2] = static_cast<int32_t>(dp->tiles1_h.size());
        dp->batch_t1[4 * b + 3] = static_cast<int32_t>(t1s[b].size());
        dp->tiles1_h.insert(dp->tiles1_h.end(), t1s[b].begin(), t1s[b].end());
        dp->n_tiles1_blocks += static_cast<int64_t>(t1s[b].size());
        dp->batch_t2[2 * b] = static_cast<int32_t>(dp->tiles2_h.size());
        dp->batch_t2[2 * b + 1] = static_cast<int32_t>(t2b[b].size());
        dp->tiles2_h.insert(dp->tiles2_h.end(), t2b[b].begin(), t2b[b].end());
    }
    std::vector<int32_t> xc, xruns;
    std::vector<int64_t> xptr(1, 0);
    for (int32_t c : x_cells_) {
        if (!active[c]) continue;
        xc.push_back(c);
        for (int64_t r = x_runs_.ptr[c]; r < x_runs_.ptr[c + 1]; ++r) {
            xruns.push_back(x_runs_.idx[2 * r]);
            xruns.push_back(x_runs_.idx[2 * r + 1]);
        }
        xptr.push_back(static_cast<int64_t>(xruns.size() / 2));
    }
    dp->n_x_jobs = static_cast<int>(xc.size());
    dp->restrict_upward = restrict_upward;
    dp->up_leaves_h.clear();
    dp->up_parents_h.assign(static_cast<size_t>(t.depth) + 1, {});
    dp->coarse_level = 0;
    dp->coarse_cells = 0;
    dp->part_child_ptr_h.clear();
    dp->part_child_idx_h.clear();
    if (restrict_upward) {
        // Coarse level: the deepest level whose prefix of M (levels 0..Lc, per rhs) stays under 32 MB -- level 4 of
        // a uniform tree at orders 7 and 9 (4,681 cells: 13 / 28 MB); one more level is eight times that.
        // BBFMM_PART_COARSE_LEVEL overrides (0: no exchange, every needed multipole is recomputed).
        int Lc = 0;
        {
            const char *e = std::getenv("BBFMM_PART_COARSE_LEVEL");
            const int64_t n_pad = round_up(ops_.n, 32);
            if (e) {
                Lc = std::max(0, std::min(std::atoi(e), t.depth - 1));
            } else {
                for (int l = 1; l <= t.depth - 1; ++l)
                    if (t.level_ptr[static_cast<size_t>(l) + 1] * n_pad * 8 <= (int64_t(32) << 20)) Lc = l;
            }
            if (t.depth < 2) Lc = 0;
        }
        dp->coarse_level = Lc;
        dp->coarse_cells = Lc > 0 ? t.level_ptr[static_cast<size_t>(Lc) + 1] : 0;
        auto owned = [&](int64_t c) { return t.pt_end[c] > t.pt_begin[c] && t.pt_begin[c] >= own_b && t.pt_begin[c] < own_e; };
        // complete multipoles above the coarse level: V-list sources of the active cells (stage 1), W-list cells of
        // the target leaves (M2P), the cells of level Lc + 1 this rank owns (the rank that holds a cell's first
        // point; they feed the partial sums), and everything below them.  Cells are numbered by (level, key):
        // parents come first.
        std::vector<uint8_t> up(static_cast<size_t>(C), 0);
        for (int64_t c = 0; c < C; ++c)
            if (needed[c] && t.level[c] > Lc) up[c] = 1;
        for (int32_t leaf : target_leaves)
            for (int64_t q = t.w.ptr[leaf]; q < t.w.ptr[leaf + 1]; ++q)
                if (t.level[t.w.idx[q]] > Lc) up[t.w.idx[q]] = 1;
        if (Lc > 0 && Lc + 1 <= t.depth)
            for (int64_t c = t.level_ptr[static_cast<size_t>(Lc) + 1]; c < t.level_ptr[static_cast<size_t>(Lc) + 2]; ++c)
                if (owned(c)) up[c] = 1;
        for (int64_t c = 0; c < C; ++c)
            if (up[c])
                for (int64_t q = t.children.ptr[c]; q < t.children.ptr[c + 1]; ++q) up[t.children.idx[q]] = 1;
        for (int32_t c : src_leaves_)
            if (up[c] || (t.level[c] <= Lc && owned(c))) dp->up_leaves_h.push_back(c);
        for (int level = 1; level < t.depth; ++level)
            for (int32_t c : m2m_parents_[level])
                if (level <= Lc || up[c]) dp->up_parents_h[level].push_back(c); // coarse parents: all (partial sums, maybe zero)
        dp->reads_h.assign(static_cast<size_t>(C), 0);
        for (int64_t c = 0; c < C; ++c) dp->reads_h[static_cast<size_t>(c)] = needed[static_cast<size_t>(c)] ? 1 : 0;
        for (int32_t leaf : target_leaves)
            for (int64_t q = t.w.ptr[leaf]; q < t.w.ptr[leaf + 1]; ++q) dp->reads_h[t.w.idx[q]] = 1;
        // weights read: the points of the anterpolated leaves, of the U lists of the target leaves (near field) and of
        // the X lists of the active cells (P2L) -- as sorted positions, for the restricted gather
        {
            std::vector<uint8_t> leaf_read(static_cast<size_t>(C), 0);
            for (int32_t c : dp->up_leaves_h) leaf_read[static_cast<size_t>(c)] = 1;
            for (int32_t leaf : target_leaves)
                for (int64_t q = t.u.ptr[leaf]; q < t.u.ptr[leaf + 1]; ++q) leaf_read[static_cast<size_t>(t.u.idx[q])] = 1;
            for (int32_t c : x_cells_)
                if (active[c])
                    for (int64_t q = t.x.ptr[c]; q < t.x.ptr[c + 1]; ++q) leaf_read[static_cast<size_t>(t.x.idx[q])] = 1;
            dp->gather_pos_h.clear();
            for (int32_t c : src_leaves_) // sorted by position
                if (leaf_read[static_cast<size_t>(c)])
                    for (int64_t i = t.pt_begin[c]; i < t.pt_end[c]; ++i) dp->gather_pos_h.push_back(static_cast<int32_t>(i));
        }
        // children lists: the parents of level Lc sum the children they own only
        dp->part_child_ptr_h.assign(static_cast<size_t>(C) + 1, 0);
        dp->part_child_idx_h.reserve(t.children.idx.size());
        for (int64_t c = 0; c < C; ++c) {
            for (int64_t q = t.children.ptr[c]; q < t.children.ptr[c + 1]; ++q) {
                const int32_t ch = t.children.idx[q];
                if (Lc > 0 && t.level[c] == Lc && !owned(ch)) continue;
                dp->part_child_idx_h.push_back(ch);
            }
            dp->part_child_ptr_h[static_cast<size_t>(c) + 1] = static_cast<int64_t>(dp->part_child_idx_h.size());
        }
    }
    if (host_only_) return BBFMM_OK;
    if (restrict_upward) {
        CHK(tupload(&dp->d_up_leaves, dp->up_leaves_h));
        dp->d_up_parents.resize(dp->up_parents_h.size());
        for (size_t l = 0; l < dp->up_parents_h.size(); ++l) CHK(tupload(&dp->d_up_parents[l], dp->up_parents_h[l]));
        CHK(tupload(&dp->d_gather_pos, dp->gather_pos_h));
        CHK(tupload(&dp->d_part_child_ptr, dp->part_child_ptr_h));
        CHK(tupload(&dp->d_part_child_idx, dp->part_child_idx_h));
    }
    CHK(tupload(&dp->d_active, dp->active));
    CHK(tupload(&dp->d_tiles2, dp->tiles2_h));
    CHK(tupload(&dp->d_tiles1, dp->tiles1_h));
    CHK(tupload(&dp->d_tile_idx, dp->tile_idx_h));
    CHK(tupload(&dp->d_qlist, dp->qlist_h));
    CHK(tupload(&dp->d_x_cells, xc));
    CHK(tupload(&dp->d_x_ptr, xptr));
    CHK(tupload(&dp->d_x_runs, xruns));
    return BBFMM_OK;
}

void FmmTree::free_downward_plan(DownwardPlan *dp) {
    dfree(&dp->d_up_leaves);
    for (auto &b : dp->d_up_parents) dfree(&b);
    dfree(&dp->d_part_child_ptr);
    dfree(&dp->d_part_child_idx);
    dfree(&dp->d_gather_pos);
    dfree(&dp->d_active);
    dfree(&dp->d_tiles2);
    dfree(&dp->d_tiles1);
    dfree(&dp->d_tile_idx);
    dfree(&dp->d_qlist);
    dfree(&dp->d_x_cells);
    dfree(&dp->d_x_ptr);
    dfree(&dp->d_x_runs);
    *dp = DownwardPlan();
}

// Target subset of a partial matvec (IterativeSolver::matvec_partial, rbf.rs:119-133): the Schwarz
// preconditioner asks for the same index sets (its levels' points) in every iteration, so the
// sorted targets and the restricted downward pass are built once per distinct index set and kept
// (8 sets, least recently used evicted).
int FmmTree::subset_plan(const int64_t *idx, int64_t n_idx, SubsetPlan **out) {
    const int64_t N = tree_.n_points;
    // hash of the index set: chunk hashes computed in parallel, combined in order
    uint64_t h = 1469598103934665603ull ^ static_cast<uint64_t>(n_idx);
    {
        constexpr int64_t kChunk = int64_t(1) << 16;
        const int64_t nch = (n_idx + kChunk - 1) / kChunk;
        std::vector<uint64_t> part(static_cast<size_t>(std::max<int64_t>(nch, 1)), 0);
        parallel_for_chunks(n_idx, kChunk, [&](int64_t b, int64_t e) {
            for (int64_t c = b; c < e; c += kChunk) { // (a single-threaded host gets one call for everything)
                uint64_t hc = 1469598103934665603ull;
                for (int64_t j = c; j < std::min(e, c + kChunk); ++j) hc = (hc ^ static_cast<uint64_t>(idx[j])) * 1099511628211ull;
                part[static_cast<size_t>(c / kChunk)] = hc;
            }
        });
        for (uint64_t hc : part) h = (h ^ hc) * 1099511628211ull;
    }
    ++subset_clock_;
    for (auto &sp : subset_plans_)
        if (sp->key == h && sp->n_idx == n_idx &&
            (n_idx == 0 || std::memcmp(sp->idx.data(), idx, static_cast<size_t>(n_idx) * sizeof(int64_t)) == 0)) {
            sp->last_use = subset_clock_;
            *out = sp.get();
            return BBFMM_OK;
        }
    for (int64_t j = 0; j < n_idx; ++j)
        if (idx[j] < 0 || idx[j] >= N) return fail(BBFMM_BAD_ARGUMENT, "target index out of range");
    if (subset_plans_.size() >= 8) {
        size_t victim = 0;
        for (size_t i = 1; i < subset_plans_.size(); ++i)
            if (subset_plans_[i]->last_use < subset_plans_[victim]->last_use) victim = i;
        free_target_set(&subset_plans_[victim]->ts);
        free_downward_plan(&subset_plans_[victim]->dp);
        subset_plans_.erase(subset_plans_.begin() + static_cast<std::ptrdiff_t>(victim));
    }
    std::unique_ptr<SubsetPlan> sp(new SubsetPlan());
    sp->key = h;
    sp->last_use = subset_clock_;
    CHK(fill_subset_plan(idx, n_idx, sp.get()));
    *out = sp.get();
    subset_plans_.push_back(std::move(sp));
    return BBFMM_OK;
}

// Sorted targets + restricted downward pass of one index set (rows validated by the caller).
int FmmTree::fill_subset_plan(const int64_t *idx, int64_t n_idx, SubsetPlan *sp) {
    const int64_t N = tree_.n_points;
    sp->n_idx = n_idx;
    sp->idx.assign(idx, idx + n_idx);
    const int64_t m = n_idx;
    std::vector<double> x(static_cast<size_t>(std::max<int64_t>(m, 1)) * d_); // select_mat_rows, rbf.rs:1359-1360
    for (int a = 0; a < d_; ++a)
        parallel_for_chunks(m, int64_t(1) << 16, [&](int64_t b, int64_t e) {
            for (int64_t j = b; j < e; ++j) x[static_cast<size_t>(a) * m + j] = pts_[static_cast<size_t>(a) * N + idx[j]];
        });
    std::vector<int32_t> leaves;
    int64_t bad = -1;
    const auto t_0 = std::chrono::steady_clock::now();
    int rc = build_target_set(x.data(), m, std::max<int64_t>(m, 1), &sp->ts, &bad, &leaves);
    const auto t_1 = std::chrono::steady_clock::now();
    if (rc == BBFMM_OK) rc = build_downward_plan(leaves, &sp->dp);
    if (std::getenv("BBFMM_VERBOSE"))
        std::fprintf(stderr, "[bbfmm] subset plan: %lld rows, target set %.3f s, downward plan %.3f s (%zu stage-1 tiles%s)\n",
                     static_cast<long long>(m), std::chrono::duration<double>(t_1 - t_0).count(),
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_1).count(), sp->dp.tiles1_h.size(),
                     sp->dp.n_tiles1_blocks > 0 ? ", of which some cover one column block each" : "");
    if (rc == BBFMM_OK) rc = dalloc(&sp->ts.out, static_cast<size_t>(std::max<int64_t>(m, 1)));
    if (rc != BBFMM_OK) {
        free_target_set(&sp->ts);
        free_downward_plan(&sp->dp);
    }
    return rc;
}

// A registered index set (bbfmm_target_subset_create): like a cached plan, but named by an id and kept for the
// life of the handle -- the Schwarz sweep names its levels once and then calls by id, without passing (and
// comparing) millions of indices per product.  id -1 = all rows in order.
int FmmTree::register_subset(const int64_t *idx, int64_t n_idx, int *id_out) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (!idx || n_idx < 0 || !id_out) return fail(BBFMM_BAD_ARGUMENT, "bad target index array");
    CHK(ensure_rhs_capacity(1));
    if (is_identity_subset(idx, n_idx)) {
        *id_out = -1;
        return BBFMM_OK;
    }
    const int64_t N = tree_.n_points;
    for (int64_t j = 0; j < n_idx; ++j)
        if (idx[j] < 0 || idx[j] >= N) return fail(BBFMM_BAD_ARGUMENT, "target index out of range");
    std::unique_ptr<SubsetPlan> sp(new SubsetPlan());
    CHK(fill_subset_plan(idx, n_idx, sp.get()));
    *id_out = static_cast<int>(registered_plans_.size());
    registered_plans_.push_back(std::move(sp));
    return BBFMM_OK;
}

// d_y[j] = sum_i phi(x_idx[j], x_i) d_w[i]: set_weights + evaluate at the registered rows
// (IterativeSolver::matvec_partial, rbf.rs:119-133, without the nugget / polynomial terms), all on the device.
int FmmTree::matvec_subset_device(int id, const double *d_w, double *d_y, bool sync) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    const int64_t N = tree_.n_points;
    if (!d_w || !d_y) return fail(BBFMM_BAD_ARGUMENT, "bad device matvec arguments");
    if (id == -1) return matvec_device(d_w, N, 1, d_y, N, sync);
    if (id < 0 || id >= static_cast<int>(registered_plans_.size())) return fail(BBFMM_BAD_ARGUMENT, "unknown subset id");
    SubsetPlan *sp = registered_plans_[static_cast<size_t>(id)].get();
    CHK(ensure_rhs_capacity(1));
    nrhs_ = 1;
    phase_begin();
    launch_gather_weights(d_w, N, 1, d_order_.p, N, d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    CHK(upward(1));
    CHK(downward(1, &sp->dp));
    if (sp->n_idx > 0) {
        CHK(leaf_pass(sp->ts, 1, false));
        phase_begin();
        launch_scatter_output(sp->ts.out.p, sp->n_idx, 1, sp->ts.perm.p, d_y, sp->n_idx, 0, stream_);
        phase_end(kPhScatter);
    }
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(stream_));
    return BBFMM_OK;
}

// Multi-GPU: own a contiguous range of the leaves in sorted-point (Morton DFS) order.  The
// host part (owned rows, active cells, M2L tiles) also runs on BBFMM_FLAG_HOST_ONLY handles so
// that the N > 1 bookkeeping is testable without a device.
int FmmTree::set_partition(int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return fail(BBFMM_BAD_ARGUMENT, "bad rank/world");
    const HostTree &t = tree_;
    const int64_t N = t.n_points;
    part_rank_ = rank;
    part_world_ = world;
    if (have_part_) {
        free_target_set(&part_targets_);
        have_part_ = false;
    }
    free_downward_plan(&part_plan_);
    part_pending_k_ = 0;
    part_rows_.clear();
    part_bounds_.clear();
    if (world == 1) return BBFMM_OK;
    // balance the leaf-pass + M2L work proxy: P2P pair count + a per-point share of the far field
    const size_t nl = src_leaves_.size();
    std::vector<double> work(nl);
    double total = 0;
    const double far_per_point = m2l_flops_k1_ / std::max<double>(1.0, static_cast<double>(N)) / 30.0;
    for (size_t i = 0; i < nl; ++i) {
        const int32_t c = src_leaves_[i];
        double ns = 0;
        for (int64_t r = u_runs_.ptr[c]; r < u_runs_.ptr[c + 1]; ++r) ns += u_runs_.idx[2 * r + 1] - u_runs_.idx[2 * r];
        const double nt = static_cast<double>(t.pt_end[c] - t.pt_begin[c]);
        work[i] = nt * ns + nt * far_per_point;
        total += work[i];
    }
    auto cut = [&](int r) {
        const double goal = total * r / world;
        double acc = 0;
        size_t i = 0;
        while (i < nl && acc + 0.5 * work[i] < goal) acc += work[i++];
        return i;
    };
    const size_t lb = rank == 0 ? 0 : cut(rank), le = rank == world - 1 ? nl : cut(rank + 1);
    // every part's range of the sorted points (the same on every rank: the gathered potentials are scattered by it)
    part_bounds_.assign(static_cast<size_t>(world) + 1, 0);
    for (int r = 1; r < world; ++r) {
        const size_t l = cut(r);
        part_bounds_[static_cast<size_t>(r)] = l < nl ? t.pt_begin[src_leaves_[l]] : N;
    }
    part_bounds_[static_cast<size_t>(world)] = N;
    std::vector<int32_t> owned_leaves(src_leaves_.begin() + static_cast<std::ptrdiff_t>(lb),
                                      src_leaves_.begin() + static_cast<std::ptrdiff_t>(le));
    // owned targets: one contiguous range of the sorted sources
    const int64_t pb = lb < le ? t.pt_begin[src_leaves_[lb]] : 0;
    const int64_t pe = lb < le ? t.pt_end[src_leaves_[le - 1]] : 0;
    CHK(build_downward_plan(owned_leaves, &part_plan_, true, pb, pe));
    part_rows_.resize(static_cast<size_t>(pe - pb));
    for (int64_t i = 0; i < pe - pb; ++i) part_rows_[i] = t.order[pb + i];
    part_empty_ = pe == pb;
    if (host_only_) return BBFMM_OK;

    TargetSet &ts = part_targets_;
    ts.m = pe - pb;
    for (int a = 0; a < 3; ++a) ts.xyz_ptr[a] = src_ptr_[a] + pb;
    std::vector<int32_t> perm(static_cast<size_t>(ts.m)), jc, tb, te, wtb, wte;
    std::vector<int64_t> wb, we;
    for (int64_t i = 0; i < ts.m; ++i) perm[i] = static_cast<int32_t>(t.order[pb + i]);
    for (size_t i = lb; i < le; ++i) {
        const int32_t c = src_leaves_[i];
        jc.push_back(c);
        tb.push_back(static_cast<int32_t>(t.pt_begin[c] - pb));
        te.push_back(static_cast<int32_t>(t.pt_end[c] - pb));
        add_w_jobs(t, c, tb.back(), te.back(), &wtb, &wte, &wb, &we, deterministic_);
    }
    ts.n_jobs = static_cast<int>(jc.size());
    ts.n_w_jobs = static_cast<int>(wtb.size());
    CHK(dupload(&ts.perm, perm));
    CHK(dupload(&ts.job_cell, jc));
    CHK(dupload(&ts.tgt_begin, tb));
    CHK(dupload(&ts.tgt_end, te));
    CHK(dupload(&ts.w_tgt_begin, wtb));
    CHK(dupload(&ts.w_tgt_end, wte));
    CHK(dupload(&ts.w_begin, wb));
    CHK(dupload(&ts.w_end, we));
    CHK(dalloc(&ts.out, static_cast<size_t>(std::max(k_cap_, 1)) * std::max<int64_t>(ts.m, 1)));
    CHK(build_sym_runs(&ts, jc, pb, pe, &part_plan_.active));
    have_part_ = true;
    return BBFMM_OK;
}

void FmmTree::stats(bbfmm_tree_stats *out) const {
    const HostTree &t = tree_;
    std::memset(out, 0, sizeof *out);
    out->d = d_;
    out->order = order_;
    out->n_nodes = ops_.n;
    out->depth = t.depth;
    out->n_points = t.n_points;
    out->n_cells = t.n_cells();
    for (int64_t c = 0; c < t.n_cells(); ++c) out->n_leaves += t.is_leaf[c];
    out->n_u = static_cast<int64_t>(t.u.idx.size());
    out->n_v = static_cast<int64_t>(t.v.idx.size());
    out->n_w = static_cast<int64_t>(t.w.idx.size());
    out->n_x = static_cast<int64_t>(t.x.idx.size());
    int64_t pairs = 0, tile_bytes = 0;
    const int64_t per_pt = 8 * d_ + 8; // coordinates + one weight
    for (int32_t c : src_leaves_) {
        const int64_t nt = t.pt_end[c] - t.pt_begin[c];
        int64_t ns = 0;
        for (int64_t r = u_runs_.ptr[c]; r < u_runs_.ptr[c + 1]; ++r) ns += u_runs_.idx[2 * r + 1] - u_runs_.idx[2 * r];
        pairs += nt * ns;
        tile_bytes += (nt + ns) * per_pt;
    }
    out->p2p_pairs = pairs;
    out->p2p_tile_bytes_k1 = tile_bytes;
    for (int32_t c : src_leaves_) {
        const int64_t nt = t.pt_end[c] - t.pt_begin[c], nw = t.w.ptr[c + 1] - t.w.ptr[c];
        out->wx_pairs += nt * nw * ops_.n;
        out->wx_tile_bytes_k1 += nw * (nt * per_pt + 2 * static_cast<int64_t>(ops_.n) * 8);
    }
    out->m2l_flops_k1 = m2l_flops_k1_;
    for (int r : basis_rank_) out->m2l_basis_rank = std::max<int32_t>(out->m2l_basis_rank, r);
    out->m2l_basis_len = shared_basis_ ? basis_pad_ : 0;
    out->m2l_batches = static_cast<int32_t>(m2l_batches_.size());
    out->m2l_rhs_per_pass = m2l_rhs_chunk_;
    out->m2l_slots_bytes_per_rhs = cbuf_total_len_ * 8;
    out->m2l_intermediate_bytes = static_cast<int64_t>(d_cbuf_.n) * 8;
    for (int a = 0; a < d_; ++a) out->center[a] = t.center[a];
    out->radius = t.radius;
}

int FmmTree::debug_get_coefficients(char which, int k, double *out) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (k < 1 || k > k_cap_ || !out) return fail(BBFMM_BAD_ARGUMENT, "bad coefficient request");
    const double *src = (which == 'M' || which == 'm') ? d_M_.p : d_L_.p;
    const int64_t C = tree_.n_cells();
    const int n = ops_.n, n_pad = cheb_.n_pad;
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipMemcpy2D(out, n * sizeof(double), src, n_pad * sizeof(double), n * sizeof(double),
                       static_cast<size_t>(k) * C, hipMemcpyDeviceToHost));
    return BBFMM_OK;
}

// Test hook: the partition's upward plan walked with point counts in place of multipoles (see the header).
int FmmTree::debug_partition_upward_counts(int64_t *counts_out, uint8_t *reads_out, int64_t *info_out) const {
    const HostTree &t = tree_;
    const DownwardPlan &dp = part_plan_;
    if (part_world_ < 2 || !dp.restrict_upward) return BBFMM_BAD_ARGUMENT;
    const int64_t C = t.n_cells();
    std::fill(counts_out, counts_out + C, int64_t(-1));
    std::fill(counts_out, counts_out + dp.coarse_cells, int64_t(0)); // the memset of the coarse prefix
    for (int32_t c : dp.up_leaves_h) counts_out[c] = t.pt_end[c] - t.pt_begin[c];
    int64_t n_parents = 0;
    for (int level = t.depth - 1; level >= 1; --level)
        for (int32_t c : dp.up_parents_h[static_cast<size_t>(level)]) {
            int64_t sum = 0;
            for (int64_t q = dp.part_child_ptr_h[static_cast<size_t>(c)]; q < dp.part_child_ptr_h[static_cast<size_t>(c) + 1]; ++q) {
                const int64_t v = counts_out[dp.part_child_idx_h[static_cast<size_t>(q)]];
                if (v < 0) return BBFMM_UNSUPPORTED; // a child that was never computed: the plan is broken
                sum += v;
            }
            counts_out[c] = sum;
            ++n_parents;
        }
    std::copy(dp.reads_h.begin(), dp.reads_h.end(), reads_out);
    info_out[0] = dp.coarse_level;
    info_out[1] = dp.coarse_cells;
    info_out[2] = static_cast<int64_t>(dp.up_leaves_h.size());
    info_out[3] = n_parents;
    return BBFMM_OK;
}

// Test hook: apply the stacked M2L tables on the host (plain loops).  Validates the table
// construction without a GPU; never reached from a compute entry point.
int FmmTree::debug_apply_m2l_tables_host(const double *M, double *L) const {
    const int n = ops_.n, n_pad = round_up(n, 32);
    // exactly as the unrestricted device launches walk them: batch by batch through ONE buffer of the largest
    // batch's length (slot addresses are relative to the batch), stage 1 over the batch's tile list (boundary
    // variants, group operators), stage 2 over the classes of the batch.  The buffer is NOT cleared between batches:
    // what the zero-fill lists do not reset is left as the previous batch wrote it, like on the device.
    std::vector<double> cbuf(static_cast<size_t>(std::max<int64_t>(cbuf_batch_len_, 1)), 0.0);
    std::vector<int32_t> seen(static_cast<size_t>(tree_.n_cells()), 0);
    const bool have_zero_lists = m2l_batches_.size() > 1 && !m2l_zero_h_.empty();
    for (size_t b = 0; b < m2l_batches_.size(); ++b) {
        const M2lBatch &mb = m2l_batches_[b];
        if (have_zero_lists)
            for (int64_t z = m2l_zero_ptr_[b]; z < m2l_zero_ptr_[b + 1]; ++z)
                std::fill(cbuf.begin() + 2 * static_cast<int64_t>(m2l_zero_h_[static_cast<size_t>(2 * z)]),
                          cbuf.begin() + 2 * (static_cast<int64_t>(m2l_zero_h_[static_cast<size_t>(2 * z)]) + m2l_zero_h_[static_cast<size_t>(2 * z + 1)]), 0.0);
        for (int32_t ti = mb.t1_first; ti < mb.t1_first + mb.t1_count; ++ti) {
            const M2lTileDesc &td = m2l_tiles1_h_[static_cast<size_t>(ti)];
            const bool variant = static_cast<size_t>(td.level_class) >= m2l_host_.size();
            const HostM2lClass &hc = variant ? m2l_variants_[static_cast<size_t>(td.level_class) - m2l_host_.size()]
                                             : m2l_host_[static_cast<size_t>(td.level_class)];
            if (hc.vt_all.empty()) return BBFMM_UNSUPPORTED; // tables were released after upload
            if (m2l_batch_of_class_[static_cast<size_t>(td.level_class)] != static_cast<int32_t>(b)) return BBFMM_BAD_ARGUMENT;
            for (int32_t q = 0; q < td.count; ++q) {
                const size_t pos = static_cast<size_t>(td.pad ? m2l_tile_idx1_h_[static_cast<size_t>(td.first + q)] : td.first + q);
                ++seen[hc.cells[pos]];
                const double *Mv = M + static_cast<size_t>(hc.cells[pos]) * n;
                for (int row = 0; row < hc.n_rows; ++row) {
                    if (hc.row_tpos[row] < 0) continue; // padding row
                    const int32_t slot = hc.cslot[pos * hc.n_t + hc.row_tpos[row]];
                    if (slot < 0) continue;
                    double s = 0.0;
                    for (int m = 0; m < n; ++m) s += hc.vt_all[static_cast<size_t>(m) * hc.r_pad16 + row] * Mv[m];
                    cbuf[static_cast<size_t>(slot) * 2 + hc.row_off[row]] = s;
                }
            }
        }
        for (size_t lc = 0; lc < m2l_host_.size(); ++lc) {
            if (m2l_batch_of_class_[lc] != static_cast<int32_t>(b)) continue;
            const HostM2lClass &hc = m2l_host_[lc];
            for (size_t pos = 0; pos < hc.cells.size(); ++pos) {
                double *Lb = L + static_cast<size_t>(hc.cells[pos]) * n;
                const double *cc = &cbuf[static_cast<size_t>(hc.cbase[pos])];
                for (int i = 0; i < n; ++i) {
                    double s = 0.0;
                    for (int k = 0; k < hc.k_pad; ++k) s += hc.u_all[static_cast<size_t>(k) * n_pad + i] * cc[k];
                    Lb[i] += s;
                }
            }
        }
    }
    // every source cell of a level with M2L work belongs to exactly one stage-1 tile per batch of its level
    for (const HostM2lClass &hc : m2l_host_)
        for (int32_t c : hc.cells) {
            // (on a level cut into groups a class has one operator per group its transfer vectors reach -- a class
            // whose vectors miss a group has none for it, as build_downward_plan anticipates)
            const size_t lc = static_cast<size_t>(&hc - m2l_host_.data());
            const M2lBatch &mb = m2l_batches_[static_cast<size_t>(m2l_batch_of_class_[lc])];
            const int want = mb.groups == 1 || m2l_group_ops_[lc].empty() ? mb.groups : static_cast<int>(m2l_group_ops_[lc].size());
            if (seen[c] != want) return BBFMM_BAD_ARGUMENT;
        }
    return BBFMM_OK;
}

} // namespace bbfmm
