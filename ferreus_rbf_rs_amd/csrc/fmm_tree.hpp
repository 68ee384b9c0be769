// FmmTree: host-side mirror of ferreus_rbf_utils::FmmTree (utils.rs:383-494), i.e. of
// ferreus_bbfmm::FmmTree<K> for the closed kernel set (bbfmm.rs:194-616).  Owns the
// host tree / operators and all device state; every pass runs on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <memory>
#include <vector>

#include "device.hpp"
#include "targets.hpp"
#include "ferreus_bbfmm_hip.h"
#include "kernels.hpp"
#include "operators.hpp"
#include "parallel.hpp"
#include "tree.hpp"
#include "tree_device.hpp"

namespace bbfmm {

enum Phase {
    kPhGather = 0, kPhP2M, kPhM2M, kPhM2L1, kPhM2L2, kPhP2L, kPhL2L, kPhP2P, kPhM2P, kPhL2P, kPhScatter,
    kNumPhases
};
static_assert(kNumPhases == BBFMM_N_PHASES, "phase table");

// RAII-free device buffer helper (freed in ~FmmTree).
template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool borrowed = false; // carved from the per-call arena of evaluate(): nothing to free
};

// Host copy of one (level, octant class) M2L table set; see device.hpp M2lClass.
struct HostM2lClass {
    int level = 0, octant = 0;
    int n_rows = 0, r_pad16 = 16, n_t = 0, k_pad = 16;
    std::vector<double> vt_all, u_all; // kept only on BBFMM_FLAG_HOST_ONLY handles
    std::vector<int> src_tv, tgt_tv, tgt_off; // transfer vectors (source / target side), slot offsets
    std::vector<int32_t> row_tpos, row_off, row_dst, blk_t0, cells;
    std::vector<int32_t, DefaultInitAllocator<int32_t>> cslot; // cells x n_t, filled by parallel loops
    std::vector<int32_t> src_row0, src_row1; // stacked rows [row0, row1) of each source-side transfer vector
    std::vector<int64_t> cbase;
};

// One pass of stage 1 + stage 2 through the bounded intermediate buffer (cbuf): whole levels, or -- for a level
// whose slots alone exceed the budget -- a group of its target classes.  Slot addresses (HostM2lClass::cbase,
// cslot) are relative to the batch, so the buffer is as long as the largest batch, whatever the tree.
struct M2lBatch {
    int level_lo = 0, level_hi = 0;      // levels of the batch (level_lo == level_hi when groups > 1)
    int groups = 1, group = 0;           // the level is cut into `groups` ranges of target classes; this is range `group`
    int64_t len = 0;                     // doubles of its slots (without the dump area)
    int32_t t1_first = 0, t1_count = 0;  // its tiles in the unrestricted stage-1 / stage-2 launch lists
    int32_t t2_first = 0, t2_count = 0;
};

struct TargetSet { // targets sorted by leaf, resident on the device
    int64_t m = 0;
    DevBuf<double> xyz[3];
    const double *xyz_ptr[3] = {nullptr, nullptr, nullptr};
    DevBuf<int32_t> perm;      // sorted position -> caller's row
    DevBuf<int32_t> job_cell, tgt_begin, tgt_end; // leaves with targets
    int n_jobs = 0;
    DevBuf<int32_t> w_tgt_begin, w_tgt_end;          // M2P jobs: (leaf, chunk of its W list)
    DevBuf<int64_t> w_begin, w_end;
    int n_w_jobs = 0;
    DevBuf<double> out, grad;  // K x m, K*d x m (sorted order)
    // targets that are a contiguous sorted range of the sources (the matvec, a partition): run lists of the
    // symmetric P2P (device.hpp launch_p2p_sym); sym_off = sorted source index of target position 0
    bool sym = false;
    int32_t sym_off = 0;
    int n_sym_jobs = 0;
    DevBuf<int32_t> sym_tb, sym_te; // jobs: row chunks of the leaves
    DevBuf<int64_t> sym_ptr;        // 2 per job: run range of the job's leaf
    DevBuf<int32_t> sym_runs;
    int n_syml_jobs = 0;            // the big leaves again as whole-leaf jobs (one rhs; device.hpp launch_p2p_sym)
    DevBuf<int32_t> syml_tb, syml_te;
    DevBuf<int64_t> syml_ptr;
    int n_symw_jobs = 0;            // whole small leaves: one wave each (device.hpp launch_p2p_sym)
    DevBuf<int32_t> symw_tb, symw_te;
    DevBuf<int64_t> symw_ptr;
    // M2P + P2L fused (whole source set only): row chunks of the leaves with a W list and their W ranges
    int n_wx_jobs = 0;
    DevBuf<int32_t> wx_tb, wx_te;
    DevBuf<int64_t> wx_range;
    int n_wxl_jobs = 0;             // the same as whole-leaf jobs (one rhs; device.hpp launch_wx_sym)
    DevBuf<int32_t> wxl_tb, wxl_te;
    DevBuf<int64_t> wxl_range;
};

// The part of the downward pass a set of target leaves needs (a partition of the sources, or the
// target subset of a partial matvec): cells that carry targets, the M2L tiles and P2L jobs for them.
struct DownwardPlan {
    std::vector<uint8_t> active;            // cells with targets (leaves and their ancestors)
    std::vector<M2lTileDesc> tiles2_h;      // compact stage-2 tiles over the active cells (tail split), batch by batch
    std::vector<M2lTileDesc> tiles1_h;      // compact stage-1 tiles over the V-list sources of active cells, batch by batch
    std::vector<int32_t> batch_t1, batch_t2; // per batch: stage 1 (first, count) of the whole-operator tiles and of the
                                             // per-block tiles (4 ints); stage 2 (first, count)
    int64_t n_tiles1_blocks = 0;            // stage-1 tiles that cover one column block (sources needing few blocks)
    std::vector<int32_t> tile_idx_h;        // class positions of the cells of both tile lists
    std::vector<uint16_t> qlist_h;          // active contraction steps of the stage-2 tiles
    int n_x_jobs = 0;
    // A partition also splits the upward pass (bbfmm.rs:666-772).  Levels <= coarse_level: every rank sums the
    // multipoles of the sources it owns only (P2M of its own coarse leaves, M2M over the children it owns at
    // level coarse_level + 1, over all children above), and one all-reduce (sum) of that prefix of M -- cells are
    // numbered by level, so it is contiguous -- gives every rank the complete coarse multipoles (M2M is linear).
    // Levels > coarse_level: the rank computes, complete, the cells it owns at level coarse_level + 1, the V-list
    // sources of its active cells and the W-list cells of its target leaves, with everything below them (a halo
    // of three cells per level around its subtree).
    bool restrict_upward = false;
    int coarse_level = 0;          // 0: nothing to exchange (every multipole the rank reads is computed locally)
    int64_t coarse_cells = 0;      // cells of levels 0..coarse_level = level_ptr[coarse_level + 1]
    std::vector<int32_t> up_leaves_h;
    std::vector<std::vector<int32_t>> up_parents_h; // per level
    std::vector<int64_t> part_child_ptr_h;          // children CSR with the rows of level coarse_level cut to the
    std::vector<int32_t> part_child_idx_h;          // children this rank owns
    std::vector<uint8_t> reads_h;                   // cells whose multipoles the rank's downward / leaf pass reads
    std::vector<int32_t> gather_pos_h;              // sorted source positions whose weights the rank reads (P2M of its
    DevBuf<int32_t> d_gather_pos;                   // share, near field and X lists of its targets)
    DevBuf<int32_t> d_up_leaves;
    std::vector<DevBuf<int32_t>> d_up_parents;
    DevBuf<int64_t> d_part_child_ptr;
    DevBuf<int32_t> d_part_child_idx;
    DevBuf<uint8_t> d_active;
    DevBuf<M2lTileDesc> d_tiles2, d_tiles1;
    DevBuf<int32_t> d_tile_idx, d_x_cells, d_x_runs;
    DevBuf<int64_t> d_x_ptr;
    DevBuf<uint16_t> d_qlist;
};

struct SubsetPlan { // cached target subset of bbfmm_fast_matrix_vector_product (rbf.rs:119-133)
    uint64_t key = 0;
    int64_t n_idx = 0;
    std::vector<int64_t> idx; // the index set itself: a hash hit is confirmed by comparing it
    uint64_t last_use = 0;
    TargetSet ts;
    DownwardPlan dp;
};

class DeviceGroup;

class FmmTree {
    friend class DeviceGroup; // one handle over several devices (device_group.hpp): drives the parts' passes directly
  public:
    FmmTree() = default;
    ~FmmTree();
    FmmTree(const FmmTree &) = delete;
    FmmTree &operator=(const FmmTree &) = delete;

    // FmmTree::new (bbfmm.rs:272-353)
    int create(const double *pts, int64_t n, int d, int64_t ld, int order, int kernel_type, double base_range,
               double total_sill, bool adaptive, bool sparse, const double *extents, const bbfmm_params *params,
               uint32_t flags);
    int set_weights(const double *w, int64_t rows, int k, int64_t ldw);                 // bbfmm.rs:383-401
    int set_local_coefficients(const double *w, int64_t rows, int k, int64_t ldw);      // bbfmm.rs:518-524
    int evaluate(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx,
                 double *out, int64_t ldo, double *grad, int64_t ldg, bool with_grads, bool leaves_only,
                 int64_t *bad_point_index);                                             // bbfmm.rs:444-616
    int prepare_target_subset(const int64_t *target_indices, int64_t n_target_indices);
    bool is_identity_subset(const int64_t *idx, int64_t n_idx) const;
    int fast_matrix_vector_product(const double *w, int64_t rows, int64_t basis_size, const int64_t *target_indices,
                                   int64_t n_target_indices, const double *poly, int64_t ldp, double nugget,
                                   double *result);                                     // rbf.rs:1338-1379
    int matvec_device(const double *d_w, int64_t ldw, int k, double *d_out, int64_t ldo, bool sync);
    // The partitioned matvec in two calls around the exchange of the coarse multipoles:
    // upward: gather + own P2M / M2M, packs k x partition_coarse_count() partial multipoles into d_coarse;
    // finish: takes their sum over the ranks, runs the downward and leaf passes of the owned targets.
    int64_t partition_coarse_count() const;
    // comm_stream (may be null): the stream the caller runs the all-reduce on.  It is made to wait for the packed
    // multipoles only, so that the collective overlaps the near field queued behind the pack; finish makes the handle's
    // stream wait for it.  Null: the caller orders the collective after the handle's stream itself.
    // near_field = false: the near field of the owned targets is not queued (a device group's partial product evaluates a
    // subset of them: partition_subset_finish)
    int matvec_partition_upward(const double *d_w, int64_t ldw, int k, double *d_coarse, hipStream_t comm_stream, bool near_field = true);
    // A partition's share of a partial matvec (rbf.rs:119-133) behind matvec_partition_upward and the exchange: the
    // restricted downward pass and the leaf pass of the rows `idx` (rows this part owns; plan cached by index set), the
    // n_idx values in the order of idx copied to h_out (pinned) asynchronously on the handle's stream.  One rhs.
    int partition_subset_finish(const double *d_coarse, const int64_t *idx, int64_t n_idx, double *h_out, hipStream_t comm_stream);
    int matvec_partition_finish(const double *d_coarse, double *d_out, int64_t ldo, bool sync, hipStream_t comm_stream);
    // the same, but the owned potentials stay in the tree's sorted order: k rows of d_seg (ld >= the owned count), the
    // block a rank sends to the all-gather; partition_scatter then writes gathered blocks of parts [first, first + n)
    // -- d_all[part][k][m_max] -- to their rows of d_out in one pass over the tree's permutation
    int matvec_partition_finish_sorted(const double *d_coarse, double *d_seg, int64_t ld, hipStream_t comm_stream);
    int partition_scatter(const double *d_all, int first_part, int n_parts, int64_t m_max, int k, double *d_out, int64_t ldo);
    // the same for a device group: the owned potentials go straight to (pinned) host memory, k rows of ld doubles, as one
    // asynchronous 2-D copy behind the passes on the handle's stream
    int matvec_partition_finish_host(const double *d_coarse, double *h_seg, int64_t ld, hipStream_t comm_stream);
    const std::vector<int64_t> &partition_bounds() const { return part_bounds_; } // world + 1 offsets into the sorted points
    int register_subset(const int64_t *idx, int64_t n_idx, int *id_out);
    int matvec_subset_device(int id, const double *d_w, double *d_y, bool sync);
    int set_partition(int rank, int world);

    const HostTree &tree() const { return tree_; }
    const Operators &ops() const { return ops_; }
    const PodDoubles &source_points() const { return pts_; } // n x d column-major, ld = n
    int order() const { return order_; }
    const char *last_error() const { return err_.c_str(); }
    hipStream_t stream() const { return stream_; }
    bool host_only() const { return host_only_; }
    // The handle may be used from any host thread (a solver's worker threads start with device 0 current):
    // the C entry points call this first.
    void bind_device() const {
        if (device_ >= 0) (void)hipSetDevice(device_);
    }
    int device() const { return device_; }
    bool tree_built_on_device() const { return tree_built_on_device_; }
    bool last_evaluate_at_sources() const { return last_eval_at_sources_; }
    int last_evaluate_path() const { return last_eval_at_sources_ ? 1 : (last_eval_rows_of_sources_ ? 2 : 0); }
    // targets that are rows of the sources, bit for bit: their row numbers (host only; builds the point table on first use)
    bool targets_are_rows_of_sources(const double *x, int64_t m, int64_t ldx, std::vector<int64_t> *rows);
    // m == N targets that are the handle's source points bit for bit, row for row (what evaluate() asks; host only)
    bool targets_are_sources(const double *x, int64_t m, int64_t ldx) const;
    void stats(bbfmm_tree_stats *out) const;
    void set_profiling(bool on) { profiling_ = on; }
    // Resolves the recorded event pairs (synchronises the stream) and returns the totals.
    const double *phase_ms() { collect_phase_times(); return phase_ms_; }
    const int64_t *phase_count() { collect_phase_times(); return phase_count_; }
    void reset_phase_ms() {
        collect_phase_times();
        for (double &v : phase_ms_) v = 0.0;
        for (int64_t &v : phase_count_) v = 0;
    }
    const std::vector<int64_t> &partition_rows() const { return part_rows_; }
    bool partitioned() const { return part_world_ > 1; }
    int partition_rank() const { return part_rank_; }
    const std::vector<HostM2lClass> &m2l_host() const { return m2l_host_; }
    void m2l_variant_stats(int64_t *n_variants, int64_t *n_cells) const {
        *n_variants = static_cast<int64_t>(m2l_variants_.size());
        *n_cells = 0;
        for (const HostM2lClass &v : m2l_variants_) *n_cells += static_cast<int64_t>(v.cells.size());
    }
    // Test hook (host loops over the stacked M2L tables; needs BBFMM_FLAG_HOST_ONLY).
    // M, L: n_cells x n (cell-major, one rhs).  L is accumulated into.
    int debug_apply_m2l_tables_host(const double *M, double *L) const;
    int debug_get_coefficients(char which, int k, double *out);
    int debug_partition_upward_counts(int64_t *counts_out, uint8_t *reads_out, int64_t *info_out) const;

  private:
    int fail(int code, const std::string &msg);
    int hip_fail(hipError_t e, const char *what);
    int upload();
    int build_m2l_tables();
    void fill_m2l_operator_arrays(const HostM2lClass &hc, std::vector<double> *vt_all,
                                  std::vector<double> *u_all) const;
    void fill_m2l_operator_arrays(const HostM2lClass &hc, double *vt_all, double *u_all) const;
    int ensure_rhs_capacity(int k);
    int upward(int k, const DownwardPlan *dp = nullptr); // P2M + M2M from w_sorted_ (a partition's plan: needed cells only)
    // M2L + P2L + L2L into L_ (restricted by a plan); wx: run P2L fused with M2P into wx->out (zeroed by the caller)
    int downward(int k, const DownwardPlan *dp = nullptr, const TargetSet *wx = nullptr);
    int downward_m2l(int k, const DownwardPlan *dp);                          // its M2L part (reads M only)
    int downward_tail(int k, const DownwardPlan *dp, const TargetSet *wx);    // P2L (+ M2P when wx) and L2L
    int leaf_pass(const TargetSet &ts, int k, bool with_grads);
    int leaf_pass_near(const TargetSet &ts, int k, bool with_grads, hipStream_t st, int parts, bool wx_done = false);
    int leaf_pass_far(const TargetSet &ts, int k, bool with_grads);
    int build_target_set(const double *x, int64_t m, int64_t ldx, TargetSet *ts, int64_t *bad_point_index,
                         std::vector<int32_t> *leaves_out = nullptr);
    // the same on the device (targets.hip) for batches of at least device_targets_min_ rows
    int build_target_set_device(const double *x, int64_t m, int64_t ldx, TargetSet *ts, int64_t *bad_point_index,
                                std::vector<int32_t> *leaves_out);
    int build_target_set_host(const double *x, int64_t m, int64_t ldx, TargetSet *ts, int64_t *bad_point_index,
                              std::vector<int32_t> *leaves_out);
    int build_source_target_set();
    int build_sym_runs(TargetSet *ts, const std::vector<int32_t> &job_cells, int64_t pb, int64_t pe, const std::vector<uint8_t> *part_active = nullptr);
    void free_target_set(TargetSet *ts);
    hipError_t columns_to_host(double *dst, int64_t ld, const double *d_src, int64_t m, int ncols);
    int upload_weights(const double *w, int64_t rows, int k, int64_t ldw);
    void phase_begin();
    void phase_end(int ph);
    void collect_phase_times();
    hipEvent_t get_event();

    // ---- host state
    std::string err_;
    bool host_only_ = false;
    bool deterministic_ = false; // BBFMM_FLAG_DETERMINISTIC: no f64 atomics anywhere (ordered-pair kernels, one M2P job per leaf)
    bool tree_built_on_device_ = false;
    DevTreePoints dev_points_; // left by the device tree build until upload() has gathered the sorted sources
    int order_ = 0, d_ = 0;
    KernelSpec kernel_{};
    bbfmm_params params_{};
    PodDoubles pts_;
    HostTree tree_;
    Operators ops_;
    int nrhs_ = 0;           // set by set_weights (bbfmm.rs:384); 0 = no weights yet
    bool have_locals_ = false;
    bool locals_requested_ = false; // set_local_coefficients was called for the current weights (Leaves mode)
    // leaf-pass run lists per cell (merged sorted-source ranges)
    Csr u_runs_;             // ptr per cell, idx = 2 ints per run
    Csr x_runs_;
    std::vector<int32_t> src_leaves_;                   // leaves with sources
    std::vector<std::vector<int32_t>> m2m_parents_;     // per level: cells with children
    std::vector<std::vector<int32_t>> level_cells_;     // per level
    std::vector<int32_t> x_cells_;                      // cells with an X list
    // M2L tables (host copies kept for stats / tests)
    std::vector<HostM2lClass> m2l_host_;
    std::vector<HostM2lClass> m2l_variants_;       // stage-1 boundary variants (fmm_m2l_tables.cpp build_m2l_tables)
    std::vector<M2lTileDesc> m2l_tiles1_h_;        // unrestricted stage-1 launch: variant tiles + class tiles over the rest
    std::vector<int32_t> m2l_tile_idx1_h_;         // class positions of the `rest` tiles
    std::vector<M2lClass> m2l_classes_h_;
    std::vector<M2lTileDesc> m2l_tiles_h_;
    std::vector<uint16_t> m2l_qlist_h_;
    // cbuf: bounded intermediate of the two M2L stages.  The batches run one after another through one buffer of
    // cbuf_batch_len_ doubles per right-hand side (the largest batch + the dump area); m2l_rhs_chunk_ right-hand sides
    // go through it per pass.  Budget: BBFMM_M2L_CBUF_MB (default: a sixteenth of the device memory, at least 4096).
    std::vector<M2lBatch> m2l_batches_;
    std::vector<int32_t> m2l_batch_of_class_;          // per device class (level classes, then variants / group operators)
    std::vector<std::vector<int32_t>> m2l_group_ops_;  // per level class: its stage-1 group operators (indices into
                                                       // m2l_variants_, one per batch of the level), empty when groups == 1
    std::vector<int32_t> m2l_zero_h_;                  // (slot / 2, length / 2) of the slot segments of absent pairs, batch by batch
    std::vector<int64_t> m2l_zero_ptr_;                // per batch: range of its segments (only filled with more than one batch)
    int64_t cbuf_batch_len_ = 0;
    int64_t cbuf_total_len_ = 0;                       // sum of all slots (what one unbounded buffer would hold)
    int64_t m2l_budget_bytes_ = int64_t(16384) << 20;
    int m2l_rhs_chunk_ = 1;
    double m2l_flops_k1_ = 0;
    int n_cu_ = 256;      // compute units of the device (tail splitting of the tile lists)
    int device_ = -1;     // the HIP device that was current in create(): every entry point binds its thread to it
    std::vector<M2lTileDesc> m2l_tiles2_h_; // stage-2 launch list: m2l_tiles_h_ with the tail of every batch split
    int m2l_slot_t_ = 1; // most transfer vectors any stage-1 column block touches
    int m2l_max_blocks_ = 1; // most stage-1 column blocks any class has (how far a small launch may split the walk)
    // partition
    int part_rank_ = 0, part_world_ = 1;
    std::vector<int64_t> part_rows_, part_bounds_;
    int partition_finish_core(const double *d_coarse, hipStream_t comm_stream, int *k_out);
    bool part_empty_ = false;
    DownwardPlan part_plan_;                     // partition: restricted downward pass
    std::vector<std::unique_ptr<SubsetPlan>> subset_plans_; // partial matvecs, least recently used evicted
    uint64_t subset_clock_ = 0;
    // restrict_upward: the plan of a partition that owns the sorted sources [own_b, own_e)
    int build_downward_plan(const std::vector<int32_t> &target_leaves, DownwardPlan *dp, bool restrict_upward = false,
                            int64_t own_b = 0, int64_t own_e = 0);
    int part_pending_k_ = 0; // rhs count of a matvec_partition_upward that still waits for its finish
    bool multipoles_partial_ = false; // the last upward pass was a partition's share: the evaluator's entry points refuse to read M
    void free_downward_plan(DownwardPlan *dp);
    int subset_plan(const int64_t *idx, int64_t n_idx, SubsetPlan **out);
    uint64_t subset_key(const int64_t *idx, int64_t n_idx) const;
    bool subset_plan_cached(const int64_t *idx, int64_t n_idx, uint64_t key) const;
    uint64_t last_subset_miss_ = 0; // key of the last index set evaluate() saw without a plan (a plan is built on the second sighting)
    int fill_subset_plan(const int64_t *idx, int64_t n_idx, SubsetPlan *sp);
    std::vector<std::unique_ptr<SubsetPlan>> registered_plans_; // bbfmm_target_subset_create: kept for the handle's life

    // ---- device state
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_pack_ = nullptr, ev_comm_ = nullptr; // partitioned matvec: packed multipoles ready / all-reduce done
    struct PendingPhase {
        int phase;
        hipEvent_t e0, e1;
    };
    std::vector<PendingPhase> pending_;
    std::vector<hipEvent_t> event_pool_;
    hipEvent_t pending_begin_ = nullptr;
    bool profiling_ = false;
    double phase_ms_[kNumPhases] = {0};
    int64_t phase_count_[kNumPhases] = {0};
    std::vector<void *> owned_; // every hipMalloc'd pointer, freed in the destructor
    template <class T> int dalloc(DevBuf<T> *b, size_t n, bool zero = false);
    template <class T> int dupload(DevBuf<T> *b, const std::vector<T> &v);
    // per-call buffers of evaluate(): from a grow-only arena while one is active (no hipMalloc / hipFree per
    // call, no scrubbing of re-used memory by the driver), else as dalloc / dupload
    template <class T> int talloc(DevBuf<T> *b, size_t n, bool zero = false);
    template <class T> int tupload(DevBuf<T> *b, const std::vector<T> &v);
    void arena_begin();
    int arena_end();
    template <class T> void dfree(DevBuf<T> *b);

    ChebRef cheb_{};
    DevBuf<DevCheb> d_cheb_;
    DevBuf<double> d_src_[3];
    const double *src_ptr_[3] = {nullptr, nullptr, nullptr};
    DevBuf<double> d_zero_axis_;
    DevBuf<uint8_t> arena_;
    size_t arena_used_ = 0, arena_need_ = 0;
    bool arena_active_ = false;
    // device target grouping: the tree's key table, leaf flags, grow-only scratch
    DevBuf<uint64_t> d_tab_keys_;
    DevBuf<int32_t> d_tab_vals_;
    DevBuf<uint8_t> d_is_leaf_;
    DevBuf<uint8_t> d_tscratch_;
    DevLeafLookup lk_;
    bool lk_ready_ = false;
    int64_t device_targets_min_ = 4096;
    DevBuf<int32_t> d_order_;
    DevBuf<double> d_centers_, d_lengths_;
    DevBuf<int32_t> d_pt_begin_, d_pt_end_, d_parent_, d_octant_;
    DevBuf<int64_t> d_child_ptr_;
    DevBuf<int32_t> d_child_idx_;
    DevBuf<int32_t> d_src_leaves_;
    std::vector<DevBuf<int32_t>> d_m2m_parents_, d_level_cells_;
    DevBuf<int64_t> d_u_run_ptr_, d_x_run_ptr_, d_w_ptr_;
    DevBuf<int32_t> d_u_runs_, d_x_runs_, d_w_idx_, d_x_cells_;
    DevBuf<int64_t> d_x_job_run_ptr_;
    DevBuf<M2lClass> d_m2l_classes_;
    DevBuf<M2lTileDesc> d_m2l_tiles_, d_m2l_tiles2_, d_m2l_tiles1_;
    DevBuf<int32_t> d_tile_idx1_;
    DevBuf<int32_t> d_m2l_zero_;
    DevBuf<uint16_t> d_m2l_qlist_;

    DevBuf<uint8_t> d_active_;
    // per-rhs-capacity buffers
    int k_cap_ = 0;
    DevBuf<double> d_w_in_, d_w_sorted_, d_M_, d_L_, d_cbuf_, d_out_;
    // ---- shared-basis M2L (BBFMM_FLAG_M2L_SHARED_BASIS, an extension beyond the reference) ----
    // Per level one orthonormal basis W (n x rank) of the space the level's M2L operators read from and write to;
    // the stages run on coordinates in that basis: Mc = W^T M (stage 3 of the GEMM kernel), stage 1 / 2 with the
    // projected operators Vt_t W and W^T U_t (contraction / output length basis_pad_ instead of n_pad), L = W Lc.
    bool shared_basis_ = false;
    int basis_pad_ = 0;                  // coordinates per cell (multiple of 16; the largest level's rank, padded)
    std::vector<int> basis_rank_;        // per level
    std::vector<double> m2l_flops_level_; // per level: sum_pairs 4 n r (to restate m2l_flops_k1_ in the basis)
    std::vector<DevBuf<double>> d_basis_c_, d_basis_e_; // per level: n_pad x basis_pad_ (W), basis_pad_ x n_pad (W^T)
    DevBuf<M2lClass> d_basis_classes_;   // 2 per level >= 2: [compress, expand], cells = the level's cells
    DevBuf<M2lTileDesc> d_basis_tiles_c_, d_basis_tiles_e_;
    int n_basis_tiles_ = 0;
    DevBuf<double> d_Mc_, d_Lc_;         // k x C x basis_pad_
    int build_shared_basis(std::vector<DevBuf<double>> *d_level_ops);
    double *h_pin_ = nullptr; // pinned staging for the host-buffer matvec (N doubles up, N down)
    size_t h_pin_n_ = 0;
    int ensure_pinned(size_t n);
    // A device group stages the caller's weights once (in the primary's pinned buffer) and sends every piece to the other
    // devices of the group as well: (device, stream, destination of k x N doubles) per mirror.
    struct WeightMirror {
        int device;
        hipStream_t stream;
        double *dst;
    };
    std::vector<WeightMirror> mirrors_;
    int ensure_w_in(int k); // d_w_in_ holds k x N doubles
    // the whole upward pass from the weights staged in d_w_in_ (a device group's primary before a product the group does
    // not partition: arbitrary targets, target subsets, stored local expansions)
    int complete_upward_from_staged(int k);
    // the same from a copy of the staged weights that lives elsewhere on this device (k rows of N, the group's owner part)
    int complete_upward_from(const double *d_w, int k);
    // the sorted weights on the device ARE the caller's (a device group has compared them with the staged copy): evaluate /
    // set_local_coefficients skip their transfer while this is set
    bool group_weights_resident_ = false;
    int stage_weights_to_device(const double *w, int64_t n, int k, int64_t ldw); // host rows -> d_w_in_, staging and PCIe overlapped
    template <class F> int download_pieces(const double *d_src, int64_t total, double *pin_out, F &&consume);
    // h_pin_[0, pin_w_k_ * N) holds exactly the host weights d_w_sorted_ was gathered from (0: no such copy)
    int pin_w_k_ = 0;
    int put_weights(const double *w, int64_t rows, int k, int64_t ldw);
    bool weights_match_staged(const double *w, int k, int64_t ldw) const;
    bool last_eval_at_sources_ = false;
    bool last_eval_rows_of_sources_ = false;
    bool group_primary_ = false; // part 0 of a device group: its own partition does not stand in the way of the cached-subset path
    bool solver_tree_ = false; // created as the solver creates its tree (rbf.rs:456-467: sparse, extents from the data)
    // open-addressing table over the source points keyed by the bits of their coordinates (value: a row with those
    // coordinates, -1: empty); built by the first evaluate() that could be a matvec_partial of the unchanged caller
    std::vector<int32_t, DefaultInitAllocator<int32_t>> src_row_table_;
    uint64_t src_row_mask_ = 0;
    uint64_t point_hash(const double *x, int64_t ld, int64_t i) const;
    // Host buffers above this size (weights in + values out) are not mirrored in pinned memory: pageable copies instead.
    static constexpr size_t kMaxPinnedDoubles = size_t(1) << 28; // 2 GiB
    static constexpr int64_t kHostPiece = int64_t(1) << 18;  // rows per piece of the host <-> device pipelines (2 MB)
    std::vector<hipEvent_t> ev_out_;                         // per piece of the pipelined copy back
    TargetSet src_targets_;  // targets = sources (the matvec)
    TargetSet part_targets_; // sources owned by this rank (multi-GPU)
    bool have_part_ = false;
};

} // namespace bbfmm
