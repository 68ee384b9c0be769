// FmmTree: target sets (sources as targets, arbitrary targets on host or device), the unordered-pair run lists, restricted
// downward plans, registered target subsets and partitions.  See fmm_tree.hpp.
#include "fmm_tree_impl.hpp"

namespace bbfmm {

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
    const int64_t max_rows = p2p_sym_rows_per_job(), leaf_rows = p2p_sym3_rows_per_job();
    const int64_t nj_cells = static_cast<int64_t>(job_cells.size());
    int64_t wave_rows = p2p_sym_wave_rows();
    if (wave_rows > 0) { // a small tree (or a thin part of one): too few wave jobs to fill the chip -- workgroups take every leaf
        int64_t n_wave = 0;
        for (const int32_t c : job_cells) n_wave += (t.pt_end[c] - t.pt_begin[c]) <= wave_rows;
        if (n_wave < p2p_sym_wave_min_jobs()) wave_rows = 0;
    }
    // per chunk of leaves into local buffers (threads), concatenated in order
    constexpr int64_t kChunkS = 2048;
    const int64_t nch = (nj_cells + kChunkS - 1) / kChunkS;
    struct Part {
        std::vector<int32_t> runs, tb, te, wtb, wte, ltb, lte;
        std::vector<int64_t> range, wrange, lrange; // run ranges relative to the part's first run
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
                    if (leaf_rows > 0) { // and whole (one rhs): as few equal parts as the kernel's accumulators allow
                        const int64_t nl = (na + leaf_rows - 1) / leaf_rows;
                        for (int64_t i = 0; i < nl; ++i) {
                            P.ltb.push_back(static_cast<int32_t>(a0 - pb + na * i / nl));
                            P.lte.push_back(static_cast<int32_t>(a0 - pb + na * (i + 1) / nl));
                            P.lrange.push_back(first);
                            P.lrange.push_back(last);
                        }
                    }
                }
            }
        }
    });
    std::vector<int64_t> range, wrange, lrange;
    std::vector<int32_t> runs, tb, te, wtb, wte, ltb, lte;
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
            ltb.insert(ltb.end(), P.ltb.begin(), P.ltb.end());
            lte.insert(lte.end(), P.lte.begin(), P.lte.end());
            for (int64_t v : P.lrange) lrange.push_back(base + v);
        }
    }
    static const int order_mask = [] {
        const char *e = std::getenv("BBFMM_SYM_JOB_ORDER");
        return e ? std::atoi(e) : 15;
    }();
    ts->n_wx_jobs = 0;
    const bool whole = pb == 0 && pe == t.n_points;
    if ((whole || part_active) && !t.w.idx.empty() &&
        static_cast<int64_t>(t.n_cells()) * cheb_.n_pad < (int64_t(1) << 31)) { // M2P + P2L fused
        // Whole source set: every leaf with a W list.  A partition: its own leaves (row sums = M2P of its targets; the
        // column sums that fall on cells outside its subtree are never read) and the leaves outside whose W list holds
        // a cell of its subtree (column sums = P2L into that cell; their row sums are dropped by the kernel's output
        // window).  X = W^T (linear_tree.rs:388-392), so this covers the X lists of the partition's cells.
        std::vector<int32_t> wtb, wte, ltb2, lte2;
        std::vector<int64_t> wr, lr2;
        std::vector<int32_t> wx_leaves; // the leaves that take part
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
            wx_leaves.push_back(c);
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
        // whole-leaf jobs for one rhs: all rows of a leaf (bigger ones than the kernel takes in equal parts) x a chunk of its W
        // list; the chunks shrink until there are a few jobs per CU (a nearly uniform tree has a few dozen leaves with long lists)
        const int64_t leaf_rows_wx = wx_sym3_rows_per_job();
        if (leaf_rows_wx > 0 && !wx_leaves.empty()) {
            int64_t cells_per_job = 16;
            for (;;) {
                int64_t nj_total = 0;
                for (int32_t c : wx_leaves) {
                    const int64_t na = t.pt_end[c] - t.pt_begin[c], nw = t.w.ptr[c + 1] - t.w.ptr[c];
                    nj_total += ((na + leaf_rows_wx - 1) / leaf_rows_wx) * ((nw + cells_per_job - 1) / cells_per_job);
                }
                if (nj_total >= 4 * static_cast<int64_t>(n_cu_) || cells_per_job == 1) break;
                cells_per_job = std::max<int64_t>(1, cells_per_job / 2);
            }
            for (int32_t c : wx_leaves) {
                const int64_t a0 = t.pt_begin[c], na = t.pt_end[c] - a0, nj = (na + leaf_rows_wx - 1) / leaf_rows_wx;
                const int64_t w0 = t.w.ptr[c], nw = t.w.ptr[c + 1] - w0, nwj = (nw + cells_per_job - 1) / cells_per_job;
                for (int64_t i = 0; i < nj; ++i)
                    for (int64_t jw = 0; jw < nwj; ++jw) {
                        ltb2.push_back(static_cast<int32_t>(a0 + na * i / nj));
                        lte2.push_back(static_cast<int32_t>(a0 + na * (i + 1) / nj));
                        lr2.push_back(w0 + nw * jw / nwj);
                        lr2.push_back(w0 + nw * (jw + 1) / nwj);
                    }
            }
        }
        auto wx_longest_first = [&](std::vector<int32_t> *b, std::vector<int32_t> *e, std::vector<int64_t> *rg) { // rows x W cells
            const size_t n = b->size();
            std::vector<int32_t> perm(n), b2(n), e2(n);
            std::vector<int64_t> r2(2 * n);
            for (size_t i = 0; i < n; ++i) perm[i] = static_cast<int32_t>(i);
            std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) {
                return static_cast<int64_t>((*e)[x] - (*b)[x]) * ((*rg)[2 * x + 1] - (*rg)[2 * x]) >
                       static_cast<int64_t>((*e)[y] - (*b)[y]) * ((*rg)[2 * y + 1] - (*rg)[2 * y]);
            });
            for (size_t i = 0; i < n; ++i) {
                const size_t s0 = static_cast<size_t>(perm[i]);
                b2[i] = (*b)[s0], e2[i] = (*e)[s0], r2[2 * i] = (*rg)[2 * s0], r2[2 * i + 1] = (*rg)[2 * s0 + 1];
            }
            b->swap(b2), e->swap(e2), rg->swap(r2);
        };
        if (order_mask & 8) wx_longest_first(&ltb2, &lte2, &lr2);
        ts->n_wxl_jobs = static_cast<int>(ltb2.size());
        CHK(dupload(&ts->wxl_tb, ltb2));
        CHK(dupload(&ts->wxl_te, lte2));
        CHK(dupload(&ts->wxl_range, lr2));
        if (order_mask & 8) wx_longest_first(&wtb, &wte, &wr);
        ts->n_wx_jobs = static_cast<int>(wtb.size());
        CHK(dupload(&ts->wx_tb, wtb));
        CHK(dupload(&ts->wx_te, wte));
        CHK(dupload(&ts->wx_range, wr));
    }
    // Job order.  The lists above are in Morton order, and the hardware starts workgroups in index order: the duration of
    // a job varies with its place in that order (a leaf's two-sided runs are the neighbours AFTER it), and whole 153-row
    // leaves in Morton order left the SQs idle for a quarter of the launch (5M Spheroidal3 points: 13.5 ms; shuffled 10.6;
    // longest first 10.5 -- profiles/r06_e_*).  Longest first (rows x columns) is the classic remedy for such a tail.
    // BBFMM_SYM_JOB_ORDER: bit 0 whole-leaf jobs, bit 1 chunk jobs, bit 2 wave jobs, bit 3 the fused M2P + P2L jobs (default 15: all).
    auto longest_first = [&](std::vector<int32_t> *b, std::vector<int32_t> *e, std::vector<int64_t> *rg) {
        const size_t n = b->size();
        std::vector<int64_t> work(n);
        parallel_for(static_cast<int64_t>(n), 1024, [&](int64_t i) {
            int64_t cols = 0;
            for (int64_t r = (*rg)[2 * i]; r < (*rg)[2 * i + 1]; ++r) cols += runs[3 * r + 1] - runs[3 * r];
            work[static_cast<size_t>(i)] = cols * ((*e)[static_cast<size_t>(i)] - (*b)[static_cast<size_t>(i)]);
        });
        std::vector<int32_t> perm(n);
        for (size_t i = 0; i < n; ++i) perm[i] = static_cast<int32_t>(i);
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return work[static_cast<size_t>(x)] > work[static_cast<size_t>(y)]; });
        std::vector<int32_t> b2(n), e2(n);
        std::vector<int64_t> r2(2 * n);
        for (size_t i = 0; i < n; ++i) {
            const size_t s0 = static_cast<size_t>(perm[i]);
            b2[i] = (*b)[s0], e2[i] = (*e)[s0], r2[2 * i] = (*rg)[2 * s0], r2[2 * i + 1] = (*rg)[2 * s0 + 1];
        }
        b->swap(b2), e->swap(e2), rg->swap(r2);
    };
    if (order_mask & 1) longest_first(&ltb, &lte, &lrange);
    if (order_mask & 2) longest_first(&tb, &te, &range);
    if (order_mask & 4) longest_first(&wtb, &wte, &wrange);
    ts->n_symw_jobs = static_cast<int>(wtb.size());
    CHK(dupload(&ts->symw_tb, wtb));
    CHK(dupload(&ts->symw_te, wte));
    CHK(dupload(&ts->symw_ptr, wrange));
    ts->n_syml_jobs = static_cast<int>(ltb.size());
    CHK(dupload(&ts->syml_tb, ltb));
    CHK(dupload(&ts->syml_te, lte));
    CHK(dupload(&ts->syml_ptr, lrange));
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
    dfree(&ts->syml_tb);
    dfree(&ts->syml_te);
    dfree(&ts->syml_ptr);
    ts->n_syml_jobs = 0;
    dfree(&ts->wx_tb);
    dfree(&ts->wx_te);
    dfree(&ts->wx_range);
    ts->n_wx_jobs = 0;
    dfree(&ts->wxl_tb);
    dfree(&ts->wxl_te);
    dfree(&ts->wxl_range);
    ts->n_wxl_jobs = 0;
    ts->sym = false;
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
// hash of an index set: chunk hashes computed in parallel, combined in order
uint64_t FmmTree::subset_key(const int64_t *idx, int64_t n_idx) const {
    uint64_t h = 1469598103934665603ull ^ static_cast<uint64_t>(n_idx);
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
    return h;
}

bool FmmTree::subset_plan_cached(const int64_t *idx, int64_t n_idx, uint64_t key) const {
    for (const auto &sp : subset_plans_)
        if (sp->key == key && sp->n_idx == n_idx &&
            (n_idx == 0 || std::memcmp(sp->idx.data(), idx, static_cast<size_t>(n_idx) * sizeof(int64_t)) == 0))
            return true;
    return false;
}

int FmmTree::subset_plan(const int64_t *idx, int64_t n_idx, SubsetPlan **out) {
    const int64_t N = tree_.n_points;
    const uint64_t h = subset_key(idx, n_idx);
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
    pin_w_k_ = 0;
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

} // namespace bbfmm
