// Host-side linear Morton tree + interaction lists (stays on the host per the
// north-star: "the octree build ... stay[s] on the host").
//
// Restates ferreus_bbfmm/src/linear_tree.rs with sorted vectors and an
// open-addressing key table instead of Rust HashSet/HashMap, so that the cell
// numbering is deterministic: cells are numbered by (level, Morton key).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bbfmm {

// Open-addressing u64 -> int32 table (keys are Morton codes; value = cell index).
class KeyTable {
  public:
    void build(const std::vector<uint64_t> &keys);
    int32_t find(uint64_t key) const {
        if (mask_ == 0) return -1;
        uint64_t h = hash(key) & mask_;
        while (true) {
            const int32_t v = vals_[h];
            if (v < 0) return -1;
            if (keys_[h] == key) return v;
            h = (h + 1) & mask_;
        }
    }

    // raw table for the device lookup (targets.hip probes it with the same hash)
    const std::vector<uint64_t> &raw_keys() const { return keys_; }
    const std::vector<int32_t> &raw_vals() const { return vals_; }
    uint64_t mask() const { return mask_; }

  private:
    static uint64_t hash(uint64_t x) {
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdull;
        x ^= x >> 33;
        x *= 0xc4ceb9fe1a85ec53ull;
        x ^= x >> 33;
        return x;
    }
    std::vector<uint64_t> keys_;
    std::vector<int32_t> vals_;
    uint64_t mask_ = 0;
};

struct Csr {
    std::vector<int64_t> ptr; // n_cells + 1
    std::vector<int32_t> idx; // cell indices, sorted by key inside a row
};

struct HostTree {
    int d = 3;
    double center[3] = {0, 0, 0};
    double radius = 0;
    int depth = 0; // linear_tree.rs:160
    int64_t n_points = 0;
    bool adaptive = true;

    // cells in (level, key) order
    std::vector<uint64_t> key;
    std::vector<int32_t> level;
    std::vector<int64_t> level_ptr; // depth + 2 entries: cells of level l are [level_ptr[l], level_ptr[l+1])
    std::vector<int32_t> parent;    // generating cell (children map of the reference), -1 for root
    std::vector<int32_t> octant;    // morton::get_child_index
    std::vector<uint8_t> is_leaf;
    Csr children;
    std::vector<double> centers; // n_cells x d (row-major)
    std::vector<double> lengths;

    // points in hierarchical order: every cell owns order[pt_begin, pt_end); leaves
    // list their source rows in ascending order (linear_tree.rs:55-66).
    std::vector<int64_t> order;
    std::vector<int64_t> pt_begin, pt_end;

    Csr u, v, w, x;
    std::vector<int16_t> v_tidx; // M2L transfer index per V entry (bbfmm.rs:989-998)

    KeyTable table;

    int64_t n_cells() const { return static_cast<int64_t>(key.size()); }
};

// A cell as the subdivision produces it (before numbering).
struct BuildCell {
    uint64_t key;
    int32_t level;
    int32_t parent; // index into the same array, -1 for the root
    int64_t b, e;   // range of the hierarchically ordered points
    bool leaf;
};
void finish_tree(const std::vector<BuildCell> &cells, HostTree *out, bool with_lists = true);
void build_lists_host(HostTree *out); // U / V / W / X + v_tidx (linear_tree.rs:177-485)

// linear_tree.rs:20-175 (+ 177-485 for the lists).  pts: n x d column-major (ld).
void build_tree(const double *pts, int64_t n, int64_t ld, int d, const double *center, double radius,
                int64_t max_points_per_cell, bool store_empty_leaves, bool adaptive_tree,
                HostTree *out);

// linear_tree.rs:487-520.  Returns -1 on success or the smallest row whose point has
// no leaf ancestor (FmmError::PointOutsideTree).  cell_out: leaf cell index per row.
int64_t points_to_leaves(const HostTree &t, const double *x, int64_t m, int64_t ldx,
                         int32_t *cell_out);

} // namespace bbfmm
