// Source tree build on the device (linear_tree.rs:20-175): Morton codes, one radix sort,
// level-by-level subdivision by binary searches over the sorted codes, per-leaf row order by a segmented
// sort.  Produces exactly the cells and per-leaf point lists of the host build (tree.cpp build_tree), which
// stays as the bit-exact checker (tests/test_gpu_tree_build.py) and as the fallback for point sets that
// reach outside the root box (explicit extents smaller than the data).
#pragma once
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "tree.hpp"

namespace bbfmm {

// What the build leaves on the device for the caller to adopt (or free with free_dev_tree_points).
struct DevTreePoints {
    double *xyz[3] = {nullptr, nullptr, nullptr}; // the caller's points, caller's row order (n each)
    uint32_t *order = nullptr;                    // hierarchical order: sorted position -> row (n)
    int64_t n = 0;
};
void free_dev_tree_points(DevTreePoints *p);

// Returns 0 when the tree was built (cells in (level, key) order in *cells, out->order / depth / header
// fields set: continue with finish_tree), 1 when the device path does not apply (a point outside the root
// box, n >= 2^31): nothing was changed, use the host build; < 0: HIP error (-hipError_t).
int build_tree_cells_device(const double *pts, int64_t n, int64_t ld, int d, const double *center, double radius,
                            int64_t max_points_per_cell, bool store_empty_leaves, bool adaptive_tree, HostTree *out,
                            std::vector<BuildCell> *cells, DevTreePoints *dev_points, hipStream_t s);

// U / V / W / X lists and v_tidx of a numbered tree (finish_tree(..., with_lists = false) has run) on the device.
// Returns 0 on success, 1 when the device path does not apply (more than 2^31 - 1 entries in a list: use
// build_lists_host), < 0 on a HIP error.
int build_lists_device(HostTree *tree, hipStream_t s);

} // namespace bbfmm
