// FmmTree host orchestration: creation, upload, the passes and the entry points.  See fmm_tree.hpp.
#include "fmm_tree_impl.hpp"

namespace bbfmm {

int FmmTree::fail(int code, const std::string &msg) {
    err_ = msg;
    return code;
}
int FmmTree::hip_fail(hipError_t e, const char *what) {
    err_ = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return BBFMM_DEVICE_ERROR;
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

    { // A NaN coordinate would slip through every comparison below (the reference's saturating casts send it to cell 0:
      // not a behaviour worth keeping); infinities are caught with it instead of by the radius check further down.
        std::atomic<int64_t> bad{-1};
        parallel_for_chunks(static_cast<int64_t>(pts_.size()), int64_t(1) << 18, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i)
                if (!std::isfinite(pts_[static_cast<size_t>(i)])) {
                    int64_t expect = -1;
                    bad.compare_exchange_strong(expect, i);
                    return;
                }
        });
        if (bad.load() >= 0)
            return fail(BBFMM_BAD_ARGUMENT, "source_points hold a non-finite coordinate (row " + std::to_string(bad.load() % n) + ", column " +
                                                std::to_string(bad.load() / n) + ")");
    }
    double ext[6];
    solver_tree_ = sparse && !extents;
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

int FmmTree::ensure_rhs_capacity(int k) {
    if (k <= k_cap_) return BBFMM_OK;
    const int64_t N = tree_.n_points, C = tree_.n_cells();
    pin_w_k_ = 0;
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

int FmmTree::upload_weights(const double *w, int64_t rows, int k, int64_t ldw) {
    const int64_t N = tree_.n_points;
    if (!w || rows < N || ldw < rows || k < 1) return fail(BBFMM_BAD_ARGUMENT, "weights must be rows x k with rows >= N");
    pin_w_k_ = 0;
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
    pin_w_k_ = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h_pin_), n * sizeof(double), hipHostMallocPortable)); // (every device of a group reads it)
    h_pin_n_ = n;
    return BBFMM_OK;
}

// Host rows -> d_w_in_ (k columns of N rows, leading dimension ldw on the host, N on the device).  The caller's
// memory is pageable: it is staged through the pinned buffer in pieces, and the thread that staged a piece queues
// its copy to the device at once, so PCIe runs beside the staging of the other pieces (one memcpy pass + one 80 MB
// transfer at 10M points took 5 ms back to back).  The staged copy stays in h_pin_[0, k N): see pin_w_k_.
int FmmTree::stage_weights_to_device(const double *w, int64_t n, int k, int64_t ldw) {
    pin_w_k_ = 0;
    CHK(ensure_pinned(static_cast<size_t>(2) * k * n));
    if (static_cast<size_t>(k) * n > d_w_in_.n) {
        dfree(&d_w_in_);
        CHK(dalloc(&d_w_in_, static_cast<size_t>(k) * n));
    }
    double *pin_in = h_pin_;
    double *dst = d_w_in_.p;
    std::atomic<int> err{0};
    for (int j = 0; j < k; ++j) {
        const double *col = w + static_cast<size_t>(j) * ldw;
        const int64_t off = static_cast<int64_t>(j) * n;
        parallel_for_chunks(n, kHostPiece, [&](int64_t b, int64_t e) {
            bind_device();
            std::memcpy(pin_in + off + b, col + b, static_cast<size_t>(e - b) * sizeof(double));
            hipError_t r = hipMemcpyAsync(dst + off + b, pin_in + off + b, static_cast<size_t>(e - b) * sizeof(double),
                                          hipMemcpyHostToDevice, stream_);
            if (r != hipSuccess) err.store(static_cast<int>(r));
            for (const WeightMirror &mr : mirrors_) { // the other devices of a group: the same piece over their own links
                (void)hipSetDevice(mr.device);
                r = hipMemcpyAsync(mr.dst + off + b, pin_in + off + b, static_cast<size_t>(e - b) * sizeof(double), hipMemcpyHostToDevice,
                                   mr.stream);
                if (r != hipSuccess) err.store(static_cast<int>(r));
            }
            if (!mirrors_.empty()) bind_device();
        });
    }
    if (err.load() != 0) return hip_fail(static_cast<hipError_t>(err.load()), "hipMemcpyAsync(weights)");
    return BBFMM_OK;
}

// Device rows (contiguous, `total` doubles) -> the host through the pinned buffer at pin_out, in pieces: an event
// behind each piece's copy, and the host threads hand a piece to consume(begin, end) -- flattened positions, the
// values at pin_out[begin, end) -- as soon as it has landed.  Returns with the stream idle.
template <class F> int FmmTree::download_pieces(const double *d_src, int64_t total, double *pin_out, F &&consume) {
    const int64_t n_pieces = (total + kHostPiece - 1) / kHostPiece;
    if (n_pieces == 0) {
        HIPCHK(hipStreamSynchronize(stream_));
        return BBFMM_OK;
    }
    while (static_cast<int64_t>(ev_out_.size()) < n_pieces) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ev_out_.push_back(ev);
    }
    for (int64_t pc = 0; pc < n_pieces; ++pc) {
        const int64_t b = pc * kHostPiece, e = std::min(total, b + kHostPiece);
        HIPCHK(hipMemcpyAsync(pin_out + b, d_src + b, static_cast<size_t>(e - b) * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipEventRecord(ev_out_[static_cast<size_t>(pc)], stream_));
    }
    HIPCHK(hipEventSynchronize(ev_out_[0])); // the product is done and the first piece is here: start the team
    std::atomic<int> err{0};
    parallel_for_chunks(total, kHostPiece, [&](int64_t b, int64_t e) { // chunks start at piece boundaries, ascending
        bind_device();
        for (int64_t pb = b; pb < e; pb += kHostPiece) {
            const hipError_t r = hipEventSynchronize(ev_out_[static_cast<size_t>(pb / kHostPiece)]);
            if (r != hipSuccess) {
                err.store(static_cast<int>(r));
                return;
            }
            consume(pb, std::min(e, pb + kHostPiece));
        }
    });
    HIPCHK(hipStreamSynchronize(stream_));
    if (err.load() != 0) return hip_fail(static_cast<hipError_t>(err.load()), "hipEventSynchronize(result piece)");
    return BBFMM_OK;
}


// Weights of set_weights / evaluate: rows < N of k columns -> d_w_sorted_.  The pinned staging buffer keeps the last
// staged weights; a caller that hands the same values again (rbf.rs:1357-1364: set_weights(w), evaluate(w, ..)) is
// recognised by a threaded bit-for-bit comparison and pays no second transfer.
int FmmTree::put_weights(const double *w, int64_t rows, int k, int64_t ldw) {
    const int64_t N = tree_.n_points;
    if (group_weights_resident_ && k == nrhs_) return BBFMM_OK; // (a device group has checked them against its staged copy)
    if (!w || rows < N || ldw < rows || k < 1) return fail(BBFMM_BAD_ARGUMENT, "weights must be rows x k with rows >= N");
    if (weights_match_staged(w, k, ldw)) return BBFMM_OK;
    if (static_cast<size_t>(2) * k * N > kMaxPinnedDoubles) {
        pin_w_k_ = 0;
        return upload_weights(w, rows, k, ldw);
    }
    CHK(ensure_rhs_capacity(k));
    CHK(stage_weights_to_device(w, N, k, ldw));
    phase_begin();
    launch_gather_weights(d_w_in_.p, N, k, d_order_.p, N, d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    pin_w_k_ = k;
    return BBFMM_OK;
}

bool FmmTree::weights_match_staged(const double *w, int k, int64_t ldw) const {
    const int64_t N = tree_.n_points;
    if (pin_w_k_ != k || !h_pin_ || k < 1) return false;
    std::atomic<bool> same{true};
    for (int j = 0; j < k && same.load(std::memory_order_relaxed); ++j) {
        const double *a = w + static_cast<size_t>(j) * ldw, *b = h_pin_ + static_cast<size_t>(j) * N;
        parallel_for_chunks(N, kHostPiece, [&](int64_t lo, int64_t hi) {
            if (same.load(std::memory_order_relaxed) && std::memcmp(a + lo, b + lo, static_cast<size_t>(hi - lo) * sizeof(double)) != 0)
                same.store(false, std::memory_order_relaxed);
        });
    }
    return same.load();
}

// Hash of one point's coordinate bits (column-major x, leading dimension ld, row i).
uint64_t FmmTree::point_hash(const double *x, int64_t ld, int64_t i) const {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int a = 0; a < d_; ++a) {
        uint64_t b;
        std::memcpy(&b, x + static_cast<size_t>(a) * ld + i, sizeof b);
        h ^= b + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 33;
    }
    return h;
}

// Targets that are ROWS of the sources, bit for bit: the unchanged caller of matvec_partial (rbf.rs:119-133 ->
// 1357-1364 with Some(target_indices)) evaluates at select_mat_rows(source_points, idx).  Every target is looked up in a
// table over the source points (built once, threaded; two rows with equal coordinates are interchangeable: a potential
// depends on the coordinates only); one target that is no source point ends the attempt.
bool FmmTree::targets_are_rows_of_sources(const double *x, int64_t m, int64_t ldx, std::vector<int64_t> *rows) {
    const int64_t N = tree_.n_points;
    if (!x || m < 1 || ldx < m || pts_.size() != static_cast<size_t>(N) * d_) return false;
    if (src_row_table_.empty()) {
        uint64_t cap = 1;
        while (cap < static_cast<uint64_t>(2 * N)) cap <<= 1;
        src_row_mask_ = cap - 1;
        src_row_table_.resize(cap);
        parallel_for_chunks(static_cast<int64_t>(cap), int64_t(1) << 18, [&](int64_t b, int64_t e) {
            std::fill(src_row_table_.begin() + b, src_row_table_.begin() + e, -1);
        });
        int32_t *tab = src_row_table_.data();
        const double *p = pts_.data();
        parallel_for_chunks(N, int64_t(1) << 16, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                uint64_t s = point_hash(p, N, i) & src_row_mask_;
                for (;;) {
                    int32_t cur = __atomic_load_n(&tab[s], __ATOMIC_RELAXED);
                    if (cur == -1) {
                        int32_t expect = -1;
                        if (__atomic_compare_exchange_n(&tab[s], &expect, static_cast<int32_t>(i), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                        cur = expect;
                    }
                    bool same = true; // a row with the same coordinates is already in: nothing to add
                    for (int a = 0; a < d_ && same; ++a)
                        same = std::memcmp(p + static_cast<size_t>(a) * N + i, p + static_cast<size_t>(a) * N + cur, sizeof(double)) == 0;
                    if (same) break;
                    s = (s + 1) & src_row_mask_;
                }
            }
        });
    }
    rows->resize(static_cast<size_t>(m));
    int64_t *out = rows->data();
    const int32_t *tab = src_row_table_.data();
    const double *p = pts_.data();
    std::atomic<bool> all{true};
    parallel_for_chunks(m, int64_t(1) << 14, [&](int64_t b, int64_t e) {
        for (int64_t j = b; j < e; ++j) {
            if (!all.load(std::memory_order_relaxed)) return;
            uint64_t s = point_hash(x, ldx, j) & src_row_mask_;
            int64_t found = -1;
            for (;;) {
                const int32_t cur = tab[s];
                if (cur == -1) break;
                bool same = true;
                for (int a = 0; a < d_ && same; ++a)
                    same = std::memcmp(x + static_cast<size_t>(a) * ldx + j, p + static_cast<size_t>(a) * N + cur, sizeof(double)) == 0;
                if (same) {
                    found = cur;
                    break;
                }
                s = (s + 1) & src_row_mask_;
            }
            if (found < 0) {
                all.store(false, std::memory_order_relaxed);
                return;
            }
            out[j] = found;
        }
    });
    return all.load();
}

// m == N targets that are the handle's own source points, row for row and bit for bit (the unchanged caller of
// rbf.rs:1359-1360 passes select_mat_rows(source_points, all rows)).  One differing coordinate, a swapped pair of
// rows, -0.0 for 0.0: not the sources, and the caller takes the general path.
bool FmmTree::targets_are_sources(const double *x, int64_t m, int64_t ldx) const {
    const int64_t N = tree_.n_points;
    if (!x || m != N || ldx < m || pts_.size() != static_cast<size_t>(N) * d_) return false;
    std::atomic<bool> same{true};
    for (int a = 0; a < d_ && same.load(std::memory_order_relaxed); ++a) {
        const double *xa = x + static_cast<size_t>(a) * ldx, *pa = &pts_[static_cast<size_t>(a) * N];
        parallel_for_chunks(N, kHostPiece, [&](int64_t lo, int64_t hi) {
            if (same.load(std::memory_order_relaxed) && std::memcmp(xa + lo, pa + lo, static_cast<size_t>(hi - lo) * sizeof(double)) != 0)
                same.store(false, std::memory_order_relaxed);
        });
    }
    return same.load();
}

// upward_pass (bbfmm.rs:666-688)
int FmmTree::upward(int k, const DownwardPlan *dp) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    const bool part = dp && dp->restrict_upward;
    multipoles_partial_ = part; // (a partition's share: nothing but the partitioned downward pass may read M until a whole upward pass has run)
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
    CHK(downward_m2l(k, dp));
    return downward_tail(k, dp, wx);
}

// The M2L part: reads the multipoles only -- neither the sorted weights nor any target set -- so evaluate() queues it
// before it knows which targets it was given.
int FmmTree::downward_m2l(int k, const DownwardPlan *dp) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    have_locals_ = false; // L is being rewritten; downward_tail says what it holds at the end
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
                                  C, m_chunk, d_cbuf_.p, cbuf_batch_len_, stream_, false, m2l_max_blocks_);
                launch_m2l_stage1(d_m2l_classes_.p, dp->d_tiles1.p + dp->batch_t1[4 * b + 2], dp->d_tile_idx.p, dp->batch_t1[4 * b + 3],
                                  m2l_len, m2l_slot_t_, kb, C, m_chunk, d_cbuf_.p, cbuf_batch_len_, stream_, true, m2l_max_blocks_);
            } else
                launch_m2l_stage1(d_m2l_classes_.p, d_m2l_tiles1_.p + t1_first, d_tile_idx1_.p, t1_count, m2l_len, m2l_slot_t_, kb, C,
                                  m_chunk, d_cbuf_.p, cbuf_batch_len_, stream_, false, m2l_max_blocks_);
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
    HIPCHK(hipGetLastError());
    (void)t;
    return BBFMM_OK;
}

// P2L (fused with M2P when wx is given) and L2L behind the M2L part
int FmmTree::downward_tail(int k, const DownwardPlan *dp, const TargetSet *wx) {
    const HostTree &t = tree_;
    const int64_t C = t.n_cells();
    phase_begin();
    if (t.adaptive && wx) { // targets = all sources: P2L and M2P share their kernel evaluations (X = W^T)
        launch_wx_sym(kernel_, cheb_, wx->n_wx_jobs, wx->wx_tb.p, wx->wx_te.p, wx->wx_range.p, wx->n_wxl_jobs, wx->wxl_tb.p, wx->wxl_te.p,
                      wx->wxl_range.p, d_w_idx_.p, d_centers_.p,
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
// m x ncols values (column after column on the device) to the caller's host array with leading dimension ld: one copy when the
// columns are adjacent there too, one per column otherwise.  NOT hipMemcpy2DAsync: on ROCm 7.2 a per-call device buffer that
// was the source of a 2-D copy to pageable host memory and is freed right behind it (hipFree returns success) never goes
// back to the device -- with it the buffers the last kernels read; 4 MB per handle that had evaluated 80k targets, without
// bound (found with scripts/group_lifecycle_check.py; the same copy as a 1-D one leaves nothing behind).
hipError_t FmmTree::columns_to_host(double *dst, int64_t ld, const double *d_src, int64_t m, int ncols) {
    if (m <= 0 || ncols <= 0) return hipSuccess;
    if (ld == m) return hipMemcpyAsync(dst, d_src, static_cast<size_t>(m) * ncols * sizeof(double), hipMemcpyDeviceToHost, stream_);
    for (int j = 0; j < ncols; ++j) {
        const hipError_t e = hipMemcpyAsync(dst + static_cast<size_t>(j) * ld, d_src + static_cast<size_t>(j) * m,
                                            static_cast<size_t>(m) * sizeof(double), hipMemcpyDeviceToHost, stream_);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

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
            launch_p2p_sym(kernel_, ts.n_sym_jobs, ts.sym_tb.p, ts.sym_te.p, ts.sym_ptr.p, ts.n_syml_jobs, ts.syml_tb.p, ts.syml_te.p,
                           ts.syml_ptr.p, ts.n_symw_jobs, ts.symw_tb.p, ts.symw_te.p,
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
    pin_w_k_ = 0;
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    CHK(put_weights(w, rows, k, ldw));
    const auto t_1 = std::chrono::steady_clock::now();
    nrhs_ = k; // bbfmm.rs:384
    have_locals_ = locals_requested_ = false; // the stored local expansions belong to the old weights
    CHK(upward(k));
    const auto t_2 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(stream_));
    if (verbose) {
        const auto t_3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[bbfmm] set_weights: weights staged and their copies queued %.3f ms, upward pass queued %.3f ms, waited for the device %.3f ms\n",
                     ms(t_0, t_1), ms(t_1, t_2), ms(t_2, t_3));
    }
    return BBFMM_OK;
}

// After bbfmm_matvec_partition_upward the multipoles of this handle are its own subtree's share (+ the coarse prefix): a call of
// the evaluator behind it would read them as if they were whole.
static const char kPartialMultipoles[] =
    "the multipoles are one partition's share (bbfmm_matvec_partition_upward ran last): bbfmm_set_weights must be called first";

int FmmTree::set_local_coefficients(const double *w, int64_t rows, int k, int64_t ldw) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    part_pending_k_ = 0; // M, the sorted weights or the partition's outputs are rewritten: a half-done partitioned matvec is void
    if (nrhs_ == 0) return fail(BBFMM_BAD_ARGUMENT, "set_weights must be called first");
    if (multipoles_partial_) return fail(BBFMM_BAD_ARGUMENT, kPartialMultipoles);
    if (k != nrhs_) return fail(BBFMM_BAD_ARGUMENT, "weights must have the column count given to set_weights");
    CHK(put_weights(w, rows, k, ldw));
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
    if (multipoles_partial_) return fail(BBFMM_BAD_ARGUMENT, kPartialMultipoles);
    if (k != nrhs_) return fail(BBFMM_BAD_ARGUMENT, "weights must have the column count given to set_weights");
    if (m < 0 || (m > 0 && (!x || !out || ldx < m || ldo < m))) return fail(BBFMM_BAD_ARGUMENT, "bad target/output arrays");
    if (with_grads && m > 0 && (!grad || ldg < m)) return fail(BBFMM_BAD_ARGUMENT, "bad gradient array");
    if (leaves_only && !have_locals_) return fail(BBFMM_BAD_ARGUMENT, "set_local_coefficients must be called first");
    if (m >= (int64_t(1) << 31)) return fail(BBFMM_BAD_ARGUMENT, "more than 2^31-1 target points");
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    // The unchanged caller of the matvec (rbf.rs:1357-1364) evaluates at select_mat_rows(source_points, all rows):
    // N targets that ARE the sources.  The whole-tree M2L reads the multipoles only, so it is queued first and the host
    // compares targets and weights bit for bit beside it; equal targets take the resident sorted target set of
    // matvec_device (unordered-pair near field, fused M2P + P2L) -- anything else falls through to the general path
    // with the M2L already under way.
    static const bool sources_fast = [] {
        const char *e = std::getenv("BBFMM_EVAL_SOURCES_FAST"); // 0: always the general path (checker)
        return !e || std::atoi(e) != 0;
    }();
    bool m2l_queued = false;
    last_eval_at_sources_ = false;
    // (not in Leaves mode: the early M2L rewrites L, and a call that then fails -- a target outside the tree -- would leave the
    // stored expansions half written)
    if (sources_fast && !leaves_only && !with_grads && !have_part_ && !locals_requested_ && m == tree_.n_points && src_targets_.m == m && w && rows >= m &&
        ldw >= rows && static_cast<size_t>(2) * k * m <= kMaxPinnedDoubles) {
        CHK(ensure_rhs_capacity(k));
        CHK(downward_m2l(k, nullptr));
        m2l_queued = true;
        if (targets_are_sources(x, m, ldx)) {
            const int64_t N = m;
            const TargetSet &ts = src_targets_;
            const auto t_cmp = std::chrono::steady_clock::now();
            CHK(put_weights(w, rows, k, ldw)); // the weights of set_weights again: recognised, nothing moves
            const auto t_put = std::chrono::steady_clock::now();
            static const bool wx_on = [] {
                const char *e = std::getenv("BBFMM_WX_FUSED");
                const char *e2 = std::getenv("BBFMM_P2P_SYM");
                return (!e || std::atoi(e) != 0) && (!e2 || std::atoi(e2) != 0);
            }();
            const bool wx = wx_on && !deterministic_ && ts.sym && ts.n_wx_jobs > 0;
            if (wx) HIPCHK(hipMemsetAsync(ts.out.p, 0, static_cast<size_t>(k) * ts.m * sizeof(double), stream_));
            CHK(downward_tail(k, nullptr, wx ? &ts : nullptr));
            CHK(leaf_pass_near(ts, k, false, stream_, 3, wx));
            CHK(leaf_pass_far(ts, k, false));
            phase_begin();
            launch_scatter_output(ts.out.p, ts.m, k, ts.perm.p, d_out_.p, N, 0, stream_);
            phase_end(kPhScatter);
            HIPCHK(hipGetLastError());
            double *pin_out = h_pin_ + static_cast<size_t>(k) * N; // behind the staged weights (put_weights sized the buffer)
            const auto t_queued = std::chrono::steady_clock::now();
            if (verbose) { // (the device's share, by itself: costs the overlap of the copies with the last kernels)
                HIPCHK(hipStreamSynchronize(stream_));
            }
            const auto t_dev = std::chrono::steady_clock::now();
            CHK(download_pieces(d_out_.p, static_cast<int64_t>(k) * N, pin_out, [&](int64_t pb, int64_t pe) {
                while (pb < pe) { // a piece may run over a column boundary
                    const int64_t col = pb / N, row = pb - col * N, len = std::min(pe - pb, N - row);
                    std::memcpy(out + col * ldo + row, pin_out + pb, static_cast<size_t>(len) * sizeof(double));
                    pb += len;
                }
            }));
            last_eval_at_sources_ = true;
            if (verbose) {
                const auto t_end = std::chrono::steady_clock::now();
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                std::fprintf(stderr, "[bbfmm] evaluate: the %lld targets are the sources (M2L queued and targets compared in %.3f ms beside it, weights "
                                     "compared %.3f ms, rest queued %.3f ms, waited for the device %.3f ms, potentials home %.3f ms); total %.3f ms\n",
                             static_cast<long long>(m), ms(t_begin, t_cmp), ms(t_cmp, t_put), ms(t_put, t_queued), ms(t_queued, t_dev), ms(t_dev, t_end),
                             ms(t_begin, t_end));
            }
            return BBFMM_OK;
        }
    }
    // The unchanged caller of matvec_partial (rbf.rs:119-133: fast_matrix_vector_product with Some(target_indices), i.e.
    // set_weights(w) then evaluate(w, select_mat_rows(source_points, idx))): targets that are rows of the sources get the
    // cached plan of bbfmm_fast_matrix_vector_product(target_indices) -- sorted targets and restricted downward pass are
    // built once per index set instead of once per call (10M points, 19.5k rows: 88 -> 38 ms).  Only on a tree made the
    // way the solver makes its own (sparse, extents from the data: rbf.rs:456-467) -- an evaluator's tree (explicit
    // extents, not sparse: rbf.rs:677-690) never builds the table -- and not for batches below N / 2048 rows.
    last_eval_rows_of_sources_ = false;
    if (sources_fast && solver_tree_ && !m2l_queued && !leaves_only && !with_grads && (!have_part_ || group_primary_) && !locals_requested_ && k == 1 && w &&
        m < tree_.n_points && m >= std::max<int64_t>(1024, tree_.n_points / 2048) && m <= tree_.n_points / 2) {
        std::vector<int64_t> rows_of;
        // A plan (sorted targets + restricted downward pass) is built on the SECOND sighting of an index set: a caller that
        // never repeats one pays the lookup only and keeps the plans of those who do.
        bool known = false;
        if (tree_.n_points <= (int64_t(1) << 26) && targets_are_rows_of_sources(x, m, ldx, &rows_of)) {
            const uint64_t key = subset_key(rows_of.data(), m);
            known = subset_plan_cached(rows_of.data(), m, key) || key == last_subset_miss_;
            if (!known) last_subset_miss_ = key;
        }
        if (known) {
            SubsetPlan *sp = nullptr;
            CHK(subset_plan(rows_of.data(), m, &sp));
            CHK(put_weights(w, rows, 1, ldw));
            CHK(downward(1, &sp->dp));
            CHK(leaf_pass(sp->ts, 1, false));
            phase_begin();
            launch_scatter_output(sp->ts.out.p, m, 1, sp->ts.perm.p, d_out_.p, m, 0, stream_);
            phase_end(kPhScatter);
            HIPCHK(hipGetLastError());
            CHK(ensure_pinned(static_cast<size_t>(2) * tree_.n_points));
            double *pin_out = h_pin_ + tree_.n_points;
            CHK(download_pieces(d_out_.p, m, pin_out, [&](int64_t pb, int64_t pe) {
                std::memcpy(out + pb, pin_out + pb, static_cast<size_t>(pe - pb) * sizeof(double));
            }));
            last_eval_rows_of_sources_ = true;
            return BBFMM_OK;
        }
    }
    TargetSet ts;
    std::vector<int32_t> target_leaves;
    arena_begin();
    int rc = build_target_set(x, m, ldx, &ts, bad_point_index, leaves_only ? nullptr : &target_leaves); // points_to_keys, bbfmm.rs:455-465
    const auto t_targets = std::chrono::steady_clock::now();
    // w == NULL (leaves-only entry points): keep the weights already on the device, i.e. those the stored
    // local coefficients were computed from -- no N x k host-to-device copy per batch
    if (rc == BBFMM_OK && (w || !leaves_only)) rc = put_weights(w, rows, k, ldw);
    DownwardPlan dplan;
    if (rc == BBFMM_OK && !leaves_only) {
        // downward pass over cells_with_targets (bbfmm.rs:468-480) when the targets are few; many
        // targets touch every cell anyway and the whole-tree pass needs no plan.  Once
        // set_local_coefficients has stored the whole-tree expansions (Leaves mode, rbf.rs:836-838)
        // the whole-tree pass is kept, so that evaluate_leaves stays valid afterwards.
        // (planning costs host time per call: measured break-even near N/100 targets at 10M sources)
        const bool restricted = m * 128 < tree_.n_points && !locals_requested_ && !m2l_queued;
        if (restricted) rc = build_downward_plan(target_leaves, &dplan);
        if (rc == BBFMM_OK) rc = m2l_queued ? downward_tail(k, nullptr, nullptr) : downward(k, restricted ? &dplan : nullptr);
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
            hipError_t e = columns_to_host(out, ldo, o_dev.p, m, k);
            if (e != hipSuccess) rc = hip_fail(e, "copy values to host");
        }
        if (rc == BBFMM_OK && with_grads) {
            rc = talloc(&g_dev, static_cast<size_t>(k) * d_ * m);
            if (rc == BBFMM_OK) {
                launch_scatter_output(ts.grad.p, m, k * d_, ts.perm.p, g_dev.p, m, 0, stream_);
                hipError_t e = columns_to_host(grad, ldg, g_dev.p, m, k * d_);
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
    pin_w_k_ = 0; // the sorted weights come from the caller's device buffer now
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
int FmmTree::matvec_partition_upward(const double *d_w, int64_t ldw, int k, double *d_coarse, hipStream_t comm_stream, bool near_field) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (!have_part_) return fail(BBFMM_BAD_ARGUMENT, "bbfmm_set_partition (world > 1) must be called first");
    const int64_t N = tree_.n_points, C = tree_.n_cells();
    const int64_t cnt = partition_coarse_count();
    if (!d_w || k < 1 || ldw < N || (cnt > 0 && !d_coarse)) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    CHK(ensure_rhs_capacity(k));
    nrhs_ = k;
    pin_w_k_ = 0;
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
    if (near_field) CHK(leaf_pass_near(part_targets_, k, false, stream_, 1));
    part_pending_k_ = near_field ? k : -k; // (negative: upward done, the owned targets' near field not queued)
    return BBFMM_OK;
}

int FmmTree::partition_subset_finish(const double *d_coarse, const int64_t *idx, int64_t n_idx, double *h_out, hipStream_t comm_stream) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    if (!have_part_ || (part_pending_k_ != 1 && part_pending_k_ != -1)) return fail(BBFMM_BAD_ARGUMENT, "bbfmm_matvec_partition_upward (one rhs) must be called first");
    if (n_idx < 0 || (n_idx > 0 && (!idx || !h_out))) return fail(BBFMM_BAD_ARGUMENT, "bad partial product arguments");
    const int64_t C = tree_.n_cells();
    const int64_t cnt = partition_coarse_count();
    if (cnt > 0 && !d_coarse) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    part_pending_k_ = 0;
    SubsetPlan *sp = nullptr;
    if (n_idx > 0) CHK(subset_plan(idx, n_idx, &sp)); // (before anything is queued: a bad index leaves the handle as it was)
    if (comm_stream) {
        HIPCHK(hipEventRecord(ev_comm_, comm_stream));
        HIPCHK(hipStreamWaitEvent(stream_, ev_comm_, 0));
    }
    phase_begin();
    if (cnt > 0)
        HIPCHK(hipMemcpyAsync(d_M_.p, d_coarse, static_cast<size_t>(cnt) * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    phase_end(kPhM2M);
    (void)C;
    if (n_idx == 0) return BBFMM_OK;
    CHK(downward(1, &sp->dp));
    CHK(leaf_pass(sp->ts, 1, false));
    phase_begin();
    launch_scatter_output(sp->ts.out.p, n_idx, 1, sp->ts.perm.p, d_out_.p, n_idx, 0, stream_);
    HIPCHK(hipMemcpyAsync(h_out, d_out_.p, static_cast<size_t>(n_idx) * sizeof(double), hipMemcpyDeviceToHost, stream_));
    phase_end(kPhScatter);
    HIPCHK(hipGetLastError());
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

int FmmTree::matvec_partition_finish_host(const double *d_coarse, double *h_seg, int64_t ld, hipStream_t comm_stream) {
    if (!h_seg || ld < part_targets_.m) return fail(BBFMM_BAD_ARGUMENT, "bad partitioned matvec arguments");
    int k = 0;
    CHK(partition_finish_core(d_coarse, comm_stream, &k));
    const TargetSet &ts = part_targets_;
    phase_begin();
    if (ts.m > 0)
        HIPCHK(hipMemcpy2DAsync(h_seg, static_cast<size_t>(ld) * sizeof(double), ts.out.p, static_cast<size_t>(ts.m) * sizeof(double),
                                static_cast<size_t>(ts.m) * sizeof(double), static_cast<size_t>(k), hipMemcpyDeviceToHost, stream_));
    phase_end(kPhScatter);
    HIPCHK(hipGetLastError());
    return BBFMM_OK;
}

int FmmTree::ensure_w_in(int k) {
    const size_t need = static_cast<size_t>(k) * tree_.n_points;
    if (need > d_w_in_.n) {
        dfree(&d_w_in_);
        CHK(dalloc(&d_w_in_, need));
    }
    return BBFMM_OK;
}

int FmmTree::complete_upward_from_staged(int k) {
    if (k < 1 || static_cast<size_t>(k) * tree_.n_points > d_w_in_.n) return fail(BBFMM_BAD_ARGUMENT, "no staged weights");
    return complete_upward_from(d_w_in_.p, k);
}

int FmmTree::complete_upward_from(const double *d_w, int k) {
    if (host_only_) return fail(BBFMM_DEVICE_ERROR, "handle was created with BBFMM_FLAG_HOST_ONLY");
    const int64_t N = tree_.n_points;
    if (k < 1 || !d_w) return fail(BBFMM_BAD_ARGUMENT, "no staged weights");
    part_pending_k_ = 0;
    CHK(ensure_rhs_capacity(k));
    phase_begin();
    launch_gather_weights(d_w, N, k, d_order_.p, N, d_w_sorted_.p, stream_);
    phase_end(kPhGather);
    nrhs_ = k;
    have_locals_ = locals_requested_ = false;
    CHK(upward(k));
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
        CHK(ensure_rhs_capacity(1));
        CHK(stage_weights_to_device(w, N, 1, N));
        double *pin_out = h_pin_ + N;
        nrhs_ = 1;
        if (have_part_) // a partitioned handle fills its owned rows only; the others read as 0
            HIPCHK(hipMemsetAsync(d_out_.p, 0, static_cast<size_t>(N) * sizeof(double), stream_));
        CHK(matvec_device(d_w_in_.p, N, 1, d_out_.p, N, false)); // set_weights + evaluate, rbf.rs:1357-1364
        // The way back in pieces as well; the host threads add the nugget and polynomial terms (rbf.rs:1366-1376)
        // of a piece as soon as it has landed.
        CHK(download_pieces(d_out_.p, N, pin_out, [&](int64_t pb, int64_t pe) {
            for (int64_t i = pb; i < pe; ++i) {
                double v = pin_out[i] + w[i] * nugget;
                if (poly) {
                    double sacc = 0.0;
                    for (int64_t q = 0; q < basis_size; ++q) sacc += poly[q * ldp + i] * w[N + q];
                    v += sacc;
                }
                result[i] = v;
            }
        }));
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
    CHK(ensure_rhs_capacity(1));
    CHK(stage_weights_to_device(w, N, 1, N));
    double *pin_out = h_pin_ + N;
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

} // namespace bbfmm
