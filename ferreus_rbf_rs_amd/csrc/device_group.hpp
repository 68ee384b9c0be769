// DeviceGroup: ONE handle, several devices, one process.
//
// The reference keeps a single FmmTree behind a Mutex (ferreus_rbf/src/rbf.rs:85-133) and its FGMRES is not an SPMD
// program, so a drop-in that is to use the GPUs of a node has to do so behind the unchanged method set of
// ferreus_rbf_utils::FmmTree (utils.rs:392-449).  A group owns G parts -- one FmmTree per entry of the device list, each
// holding the tree on its device and one subtree partition (FmmTree::set_partition) -- and runs the partitioned matvec of
// SURVEY 8(e) inside set_weights + evaluate / fast_matrix_vector_product / matvec_device:
//
//   weights     staged once in pinned memory, every piece sent to every device of the group over its own link
//   upward      each part anterpolates its own subtree (+ halo) and packs its partial coarse multipoles
//   exchange    every part copies its partial sums to a slot on every device (peer copies on a second stream per
//               part, beside the near field); each device adds the slots in part order: same bits everywhere
//   downward    restricted M2L / P2L / L2L + leaf pass of the owned targets
//   potentials  host callers: each device copies its block (contiguous in the tree's sorted order) to pinned memory over
//               its own link and the host threads write the caller's rows through the inverse permutation;
//               device callers: blocks peer-copied to the primary, one scatter pass there
//
// No collective is supplied by the caller and no second process exists.  Parts may share a device ("0,0,0": logical
// parts, the one-GPU rehearsal of the N-device path; peer copies degenerate to device copies).  Arbitrary targets (values,
// gradients, Leaves mode) with the weights of set_weights are sharded by target rows: every part completes its own
// multipoles from its device's copy of the staged weights and evaluates a contiguous share.  Row subsets (matvec_partial) are
// dealt to the parts that own the rows.  What remains (few targets, other weights than set_weights') is served by part 0
// alone after it has completed its multipoles.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "fmm_tree.hpp"

namespace bbfmm {

struct GroupCreateArgs {
    const double *pts;
    int64_t n;
    int d;
    int64_t ld;
    int order, kernel_type;
    double base_range, total_sill;
    bool adaptive, sparse;
    const double *extents;
    const bbfmm_params *params;
    uint32_t flags;
};

class DeviceGroup {
  public:
    DeviceGroup() = default;
    ~DeviceGroup();
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;

    // primary: part 0, created by the caller on devices[0]; the group creates the others and partitions all of them
    int init(FmmTree *primary, const GroupCreateArgs &a, const std::vector<int> &devices);
    int n_parts() const { return static_cast<int>(parts_.size()); }
    int part_device(int g) const { return parts_[static_cast<size_t>(g)].device; }
    FmmTree *part(int g) const { return parts_[static_cast<size_t>(g)].t; }
    const std::vector<int64_t> &bounds() const { return bounds_; }
    const char *last_error() const { return err_.c_str(); }

    int set_weights(const double *w, int64_t rows, int k, int64_t ldw);
    // *handled = true: the targets are the sources and the weights those of set_weights -- out is written;
    // false: nothing done, the caller serves the call on the primary (after prepare_primary)
    int evaluate_at_sources(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx, double *out,
                            int64_t ldo, bool *handled);
    int fast_matrix_vector_product(const double *w, int64_t rows, int64_t basis_size, const double *poly, int64_t ldp, double nugget,
                                   double *result);
    int matvec_device(const double *d_w, int64_t ldw, int k, double *d_out, int64_t ldo, bool sync);
    // Arbitrary targets (Full mode, gradients, Leaves mode) with the weights of set_weights: part g evaluates the g-th
    // contiguous share of the target rows on its device from complete multipoles of its own (every part runs the whole
    // upward pass from its device's copy of the staged weights: replicated, side by side).  *handled = false: too few
    // targets, other weights, or no stored expansions on the parts -- the caller serves the call on the primary.
    int evaluate_sharded(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx, double *out,
                         int64_t ldo, double *grad, int64_t ldg, bool with_grads, bool leaves_only, int64_t *bad_point_index,
                         bool *handled);
    // set_local_coefficients on every part (Leaves mode over the group); *handled = false: other weights than set_weights'
    int set_local_coefficients_all(const double *w, int64_t rows, int k, int64_t ldw, bool *handled);
    // matvec_partial (rbf.rs:119-133) over the group: the rows `idx` are dealt to the parts that own them (the split is cached
    // by index set), every part runs its share of the upward pass, the exchange, and the restricted downward + leaf pass of
    // ITS rows of the set.  result: N + basis_size values, zero except the rows of idx (rbf.rs:1346, 1366-1376).
    int fast_matvec_subset(const double *w, int64_t rows, int64_t basis_size, const int64_t *idx, int64_t n_idx, const double *poly,
                           int64_t ldp, double nugget, double *result);
    // the unchanged caller's form of it: set_weights(w), then evaluate(w, select_mat_rows(source_points, idx)) -- targets that
    // are rows of the sources (found bit for bit by the primary's table), one rhs, the weights of set_weights.
    // *handled = false: not such a call.
    int evaluate_rows_of_sources(const double *w, int64_t rows, int k, int64_t ldw, const double *x, int64_t m, int64_t ldx, double *out,
                                 bool *handled);
    // Before a call that part 0 serves alone.  same_weights: the call brings the weights of set_weights (or none).
    int prepare_primary(bool same_weights);
    // the primary has been given other weights / another product behind the group's back: nothing staged is valid any more
    void primary_state_changed() { staged_k_ = dev_staged_k_ = 0; pending_k_ = 0; primary_complete_ = true; all_complete_ = all_locals_ = false; }
    bool weights_match_staged(const double *w, int64_t rows, int k, int64_t ldw) const;
    int last_path() const { return last_path_; } // 1: the last evaluate ran partitioned over the group, 0: on the primary
    void set_profiling(bool on);
    int part_phase_ms(int g, double *ms_out, int64_t *count_out);
    void reset_phase_ms();

  private:
    struct Part {
        FmmTree *t = nullptr;
        std::unique_ptr<FmmTree> own;
        int device = 0;
        int owner = 0;             // first part on the same device: holds the device's copy of the staged weights
        hipStream_t comm = nullptr;
        hipEvent_t ev_w = nullptr;    // (owners) the staged weights have arrived
        hipEvent_t ev_up = nullptr;   // upward pass + near field queued: the part has read the staged weights
        hipEvent_t ev_sent = nullptr; // its partial sums are in every slot
        hipEvent_t ev_sum = nullptr;  // it has added its slots
        hipEvent_t ev_done = nullptr; // its potentials have left
        double *d_send = nullptr, *d_slots = nullptr, *d_sum = nullptr, *d_seg = nullptr;
        int64_t pb = 0, m = 0;        // owned range of the sorted points
    };
    int fail(int code, const std::string &msg);
    int hip_fail(hipError_t e, const char *what);
    int part_fail(const Part &p, int rc);
    int ensure_capacity(int k, bool device_blocks);
    int stage(const double *w, int64_t rows, int k, int64_t ldw);
    // upward pass of every part + the exchange, from the weights at d_w (leading dimension ld) on the primary's device and
    // from the owners' staged copies elsewhere (d_w == nullptr: staged copies everywhere)
    int run_upward(int k, const double *d_w_primary, int64_t ld_primary, bool near_field = true);
    template <class F> int finish_to_host(int k, F &&consume);
    template <class F> int for_parts(F &&fn);
    void free_buffers();

    std::vector<Part> parts_;
    std::vector<int64_t> bounds_;
    std::vector<int32_t> inv_order_; // row -> sorted position
    std::string err_;
    int64_t n_ = 0, cnt_ = 0, m_max_ = 0;
    int k_cap_ = 0;
    bool have_blocks_ = false;
    double *d_all_ = nullptr;       // primary's device: gathered blocks of a device-resident product
    hipEvent_t ev_in_ = nullptr;
    int staged_k_ = 0;              // the pinned buffer of the primary and every owner's d_w_in_ hold the weights of set_weights
    int dev_staged_k_ = 0;          // every owner's d_w_in_ (the primary's included) holds the weights of the last product: set_weights',
                                    // or the device-resident ones of matvec_device (which the pinned buffer does not hold)
    int pending_k_ = 0;             // upward + exchange queued for them, not consumed yet
    bool primary_complete_ = false; // the primary holds the complete multipoles of the staged weights
    bool all_complete_ = false;     // every part does
    bool all_locals_ = false;       // every part holds the whole-tree local expansions of the staged weights (Leaves mode)
    int64_t shard_min_rows_ = 16384; // targets per part below which a call is not worth sharding (BBFMM_GROUP_SHARD_MIN)
    int complete_all(int k);
    struct SubsetSplit { // an index set dealt to the parts that own its rows
        uint64_t key = 0, last_use = 0;
        std::vector<int64_t> idx;                  // the set itself (a hash hit is confirmed by comparing it)
        std::vector<std::vector<int64_t>> rows;    // per part: its rows of the set ...
        std::vector<std::vector<int64_t>> where;   // ... and their positions in idx
        std::vector<int64_t> offset;               // per part: first value of its block in the pinned buffer
    };
    std::vector<std::unique_ptr<SubsetSplit>> splits_;
    uint64_t split_clock_ = 0, last_rows_miss_ = 0;
    int subset_split(const int64_t *idx, int64_t n_idx, SubsetSplit **out);
    template <class F> int subset_product(SubsetSplit *sp, bool upward_pending, F &&consume);
    bool threads_ = true;
    int last_path_ = 0;
};

// Device list of a handle: "0,1,2,3", "all", or repeated ids for logical parts on one device ("0,0").  Empty / unset: none.
// Returns false on a malformed list (message in *err).
bool parse_device_list(const char *text, int n_devices, std::vector<int> *out, std::string *err);

} // namespace bbfmm
