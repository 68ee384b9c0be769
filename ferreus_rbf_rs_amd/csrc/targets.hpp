// Target -> leaf assignment and grouping on the device (the per-evaluate part:
// points_to_keys / points_to_leaves, ferreus_bbfmm/src/linear_tree.rs:487-534, bbfmm.rs:455-465).
// Same arithmetic as the host path (tree.cpp points_to_leaves): f64 floor((x - disp) / side) anchors,
// Morton key at the tree depth, walk up the parents until a leaf of the tree is found; rows are then
// grouped by leaf in ascending cell index with ascending rows inside a leaf (stable radix sort), so
// the target set is identical to the one the host builds.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace bbfmm {

struct DevLeafLookup {
    const uint64_t *keys = nullptr; // open-addressing table of the tree's cells (KeyTable)
    const int32_t *vals = nullptr;
    uint64_t mask = 0;
    const uint8_t *is_leaf = nullptr;
    int d = 3, depth = 0;
    double disp[3] = {0, 0, 0};
    double side = 1.0;
};

// cell[i] = leaf of row i or -1; *bad_row = min(*bad_row, first row outside the tree)
void launch_points_to_leaves(const DevLeafLookup &lk, const double *x0, const double *x1, const double *x2, int64_t m,
                             int32_t *cell, unsigned long long *bad_row, hipStream_t s);

// scratch bytes group_targets needs for m rows (keys of end_bit bits)
size_t group_targets_temp_bytes(int64_t m, int end_bit);

// cell (m) -> cell_sorted, perm (sorted position -> row); heads (m bytes) scratch;
// runs: job_cell / tgt_begin / tgt_end (capacity >= min(m, cells)), *n_runs on the device.
int group_targets(const int32_t *cell, int64_t m, int end_bit, int32_t *cell_sorted, int32_t *perm, uint8_t *heads,
                  int32_t *job_cell, int32_t *tgt_begin, int32_t *tgt_end, int32_t *n_runs, void *temp, size_t temp_bytes,
                  hipStream_t s);

// out[a][i] = in[a][perm[i]] for the axes with non-null pointers
void launch_gather_targets(const double *in0, const double *in1, const double *in2, const int32_t *perm, int64_t m,
                           double *out0, double *out1, double *out2, hipStream_t s);

} // namespace bbfmm
