// FmmTree: the stacked, permutation-folded M2L tables, their bounded batches, the shared-basis
// extension, and the host walk of the tables the tests check.  See fmm_tree.hpp.
#include "fmm_tree_impl.hpp"

namespace bbfmm {

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
            m2l_max_blocks_ = std::max(m2l_max_blocks_, n_blk);
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
