// Multi-level overlapping domain decomposition of ferreus_rbf's Schwarz preconditioner
// (ferreus_rbf/src/preconditioning/domain_decomposition.rs:67-347), host side: which points form
// which leaf domain on which level.  SURVEY.md 8(f)-1; the local factorisations and the device
// apply build on these index sets.
#pragma once
#include <cstdint>
#include <vector>

namespace bbfmm {

struct DdmParams { // config.rs:42-69
    int64_t leaf_threshold = 1024;
    double overlap_quota = 0.5;
    double coarse_ratio = 0.125;
    int64_t coarse_threshold = 4096;
};

struct DdmDomain {                 // domain.rs:86-117 (bookkeeping part)
    std::vector<int64_t> idx;      // overlapping_point_indices: internal points, then the overlap
    std::vector<uint8_t> internal; // internal_points_mask (as long as idx)
    std::vector<double> extents;   // [mins..., maxs...]
};

struct DdmLevel {
    std::vector<int64_t> point_indices; // union of the internal points of the level's leaves
    std::vector<DdmDomain> leaves;
};

struct DdmTree {
    int d = 0;
    std::vector<DdmLevel> levels; // finest first; the last level is the single coarse domain
};

// pts: n x d column-major (ld).  Returns 0, or a bbfmm_status code.
int build_ddm_tree(const double *pts, int64_t n, int d, int64_t ld, const DdmParams &params, DdmTree *out);

} // namespace bbfmm
