// Local solvers of the Schwarz preconditioner on the device:
// Domain::factorise / Domain::solve (ferreus_rbf/src/domain.rs:153-475) for all leaf domains of a
// level at once.  Host: monomials, pivoted QR, special points and Beatson's Q per domain
// (rank <= 10 columns).  Device: assembly of Q^T A Q from the points, blocked Cholesky one
// workgroup per domain, forward / back substitution per right-hand side.
#pragma once
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "ddm.hpp"
#include "kernels.hpp"

namespace bbfmm {

struct DomainPrep { // host result of the polynomial part of Domain::factorise (domain.rs:163-310)
    int k = 0;                   // rank of the monomial basis on the domain = number of special points
    std::vector<int> cols;       // the unisolvent monomial columns (ascending)
    std::vector<double> q;       // Q top block, k x m row-major (m = points - k): -(L(x_j))_a
    std::vector<double> sp_mono; // k x k special-point monomials, row-major (point, monomial)
    double tr[3] = {0, 0, 0}, sc[3] = {1, 1, 1};
    std::vector<double> xyz;     // the domain's coordinates in its (reordered) point order, axis-major d x n; handed to the
                                 // level's packed arrays and released (gathered once: the points are scattered)
};

// Reorders dom->idx / dom->internal so that the special points come first (domain.rs:250-279).
// scaling: NULL -> the cube scaling of the domain's own points (get_cheb_cube_scaling_factors of the
// domain, domain.rs:171-172), else {translation[3], scale[3]} to use instead.
int prepare_domain(const double *pts, int64_t ld, int d, int degree, int basis_size, DdmDomain *dom, DomainPrep *out,
                   const double *scaling = nullptr);

// All leaf domains of one level, factorised and resident on the device.
struct DdmLevelSolver {
    int d = 0;
    int64_t n_dom = 0, n_entries = 0;
    std::vector<int64_t> dom_off;  // n_dom + 1: first entry (point) of a domain
    std::vector<int32_t> k;        // special points per domain
    std::vector<int64_t> q_off;    // n_dom + 1: offsets into q / scratch (k*m doubles each)
    std::vector<int64_t> fac_off;  // n_dom + 1: offsets into fac (packed lower triangles, m(m+1)/2 doubles each)
    std::vector<DomainPrep> prep;  // host copies (polynomial recovery of the coarse domain)
    std::vector<int64_t> gidx_h;   // global index per entry
    bool solve_for_poly = false;
    // device
    double *d_xyz[3] = {nullptr, nullptr, nullptr}; // entries, domain order
    int64_t *d_gidx = nullptr, *d_dom_off = nullptr, *d_q_off = nullptr, *d_fac_off = nullptr;
    int32_t *d_k = nullptr;
    uint8_t *d_internal = nullptr;
    double *d_q = nullptr, *d_t = nullptr, *d_g = nullptr, *d_fac = nullptr;
    double *d_linv = nullptr;  // one large domain: inverses of the 64 x 64 diagonal blocks of the factor
    double *d_binv = nullptr;  // ... and of its 1024 x 1024 diagonal blocks (row-major, full squares): the substitutions
                               // walk blocks of 1024 (two launches each) instead of 64
    uint8_t *d_mode = nullptr; // per domain: 1 = fac holds the packed symmetric inverse (host fallback)
    double *d_tmp = nullptr;   // n_entries scratch for those domains
    int n_fallback = 0;
    double *d_work = nullptr; // n_entries: rhs / solution per entry (3x for one large domain: + z, gamma)
    // one large domain whose Cholesky factorisation failed (domain.rs:60-68: the reference switches to LBL^T): the
    // full m x m matrix factorised by rocSOLVER's pivoted LU, loaded on demand (ddm_solver.cpp big_lu_*)
    double *d_lu = nullptr;
    int *d_ipiv = nullptr;
    void *lu_handle = nullptr; // rocblas_handle
    int max_m = 0;
};

// One large domain (the coarse level above a few thousand points): factorisation and substitutions
// run as multi-workgroup launch sequences instead of one workgroup per domain.
inline bool ddm_level_is_big(const DdmLevelSolver &lv) { return lv.n_dom == 1 && lv.max_m > 2048; }

int ddm_level_build(const double *pts, int64_t ld, int d, DdmLevel *level, const KernelSpec &ks, double nugget,
                    int degree, int basis_size, bool solve_for_poly, hipStream_t s, DdmLevelSolver *out,
                    const double *scaling = nullptr);
void ddm_level_free(DdmLevelSolver *lv);
// values: global vector on the device (n_total); out: global vector on the device, rows of internal
// points (all points when `all_points`) receive the domain coefficients, other rows are left alone.
int ddm_level_solve(const DdmLevelSolver &lv, const double *d_values, double *d_out, bool all_points, hipStream_t s);

// launchers (ddm_kernels.hip)
void launch_ddm_prep(const KernelSpec &ks, double nugget, int d, const DdmLevelSolver &lv, hipStream_t s);
void launch_ddm_assemble(const KernelSpec &ks, double nugget, int d, const DdmLevelSolver &lv, hipStream_t s);
void launch_ddm_cholesky(const DdmLevelSolver &lv, int *d_fail, hipStream_t s);
int launch_ddm_solve(const DdmLevelSolver &lv, const double *d_values, double *d_out, bool all_points, hipStream_t s);
// one large domain, after a successful Cholesky: lv.d_binv = inverses of the factor's 1024 x 1024 diagonal blocks
constexpr int kBigSolveBlock = 1024;
void launch_ddm_big_block_inverses(const DdmLevelSolver &lv, hipStream_t s);
// packed lower triangle (column by column) -> full symmetric m x m column-major
void launch_ddm_unpack_symmetric(const double *packed, int m, double *full, hipStream_t s);
// gamma = (Q^T A Q)^-1 y through the LU factors of a large domain (y, gamma: m doubles on the device, in place)
int big_lu_solve(const DdmLevelSolver &lv, double *d_rhs, hipStream_t s);
// symmetric eigen-decomposition on the device (rocSOLVER, bound on demand; BBFMM_UNSUPPORTED without it): d_a n x n
// column-major -> eigenvectors, d_eval ascending.  Used by the shared-basis M2L extension (fmm_m2l_tables.cpp).
int device_symmetric_eigen(int n, double *d_a, double *d_eval, hipStream_t s);
int big_lu_factor(DdmLevelSolver *lv, hipStream_t s); // lv->d_fac holds the assembled packed matrix
void big_lu_release(DdmLevelSolver *lv);

} // namespace bbfmm
