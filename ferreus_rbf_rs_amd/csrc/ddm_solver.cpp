// See ddm_solver.hpp.
#include "ddm_solver.hpp"
#include "ddm_monomials.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "../../include/ferreus_bbfmm_hip.h"
#include "parallel.hpp"

namespace bbfmm {
namespace {

// Column-pivoted Householder QR of a (rows x cols, row-major) copy: LAPACK dgeqp3's rule, the
// remaining column of largest norm next (first of equals).  Returns the pivot order and |R_jj|.
void pivoted_qr(std::vector<double> a, int rows, int cols, std::vector<int> *piv, std::vector<double> *rdiag) {
    piv->resize(static_cast<size_t>(cols));
    std::iota(piv->begin(), piv->end(), 0);
    const int steps = std::min(rows, cols);
    rdiag->assign(static_cast<size_t>(steps), 0.0);
    auto A = [&](int r, int c) -> double & { return a[static_cast<size_t>(r) * cols + c]; };
    for (int j = 0; j < steps; ++j) {
        int best = j;
        double bn = -1.0;
        for (int c = j; c < cols; ++c) {
            double s = 0.0;
            for (int r = j; r < rows; ++r) s += A(r, c) * A(r, c);
            if (s > bn) {
                bn = s;
                best = c;
            }
        }
        if (best != j) {
            for (int r = 0; r < rows; ++r) std::swap(A(r, j), A(r, best));
            std::swap((*piv)[j], (*piv)[best]);
        }
        double norm = 0.0;
        for (int r = j; r < rows; ++r) norm += A(r, j) * A(r, j);
        norm = std::sqrt(norm);
        (*rdiag)[j] = norm;
        if (norm == 0.0) continue;
        const double alpha = A(j, j) > 0.0 ? -norm : norm;
        std::vector<double> v(static_cast<size_t>(rows - j));
        for (int r = j; r < rows; ++r) v[r - j] = A(r, j);
        v[0] -= alpha;
        double vn = 0.0;
        for (double x : v) vn += x * x;
        if (vn == 0.0) continue;
        for (int c = j; c < cols; ++c) {
            double dotp = 0.0;
            for (int r = j; r < rows; ++r) dotp += v[r - j] * A(r, c);
            const double f = 2.0 * dotp / vn;
            for (int r = j; r < rows; ++r) A(r, c) -= f * v[r - j];
        }
    }
}

// inverse of a small k x k matrix (row-major) by Gaussian elimination with partial pivoting
bool invert_small(const std::vector<double> &m, int k, std::vector<double> *inv) {
    std::vector<double> a(m);
    inv->assign(static_cast<size_t>(k) * k, 0.0);
    for (int i = 0; i < k; ++i) (*inv)[static_cast<size_t>(i) * k + i] = 1.0;
    for (int c = 0; c < k; ++c) {
        int p = c;
        for (int r = c + 1; r < k; ++r)
            if (std::fabs(a[static_cast<size_t>(r) * k + c]) > std::fabs(a[static_cast<size_t>(p) * k + c])) p = r;
        if (a[static_cast<size_t>(p) * k + c] == 0.0) return false;
        if (p != c)
            for (int x = 0; x < k; ++x) {
                std::swap(a[static_cast<size_t>(p) * k + x], a[static_cast<size_t>(c) * k + x]);
                std::swap((*inv)[static_cast<size_t>(p) * k + x], (*inv)[static_cast<size_t>(c) * k + x]);
            }
        const double piv = a[static_cast<size_t>(c) * k + c];
        for (int x = 0; x < k; ++x) {
            a[static_cast<size_t>(c) * k + x] /= piv;
            (*inv)[static_cast<size_t>(c) * k + x] /= piv;
        }
        for (int r = 0; r < k; ++r) {
            if (r == c) continue;
            const double f = a[static_cast<size_t>(r) * k + c];
            if (f == 0.0) continue;
            for (int x = 0; x < k; ++x) {
                a[static_cast<size_t>(r) * k + x] -= f * a[static_cast<size_t>(c) * k + x];
                (*inv)[static_cast<size_t>(r) * k + x] -= f * (*inv)[static_cast<size_t>(c) * k + x];
            }
        }
    }
    return true;
}

} // namespace

int prepare_domain(const double *pts, int64_t ld, int d, int degree, int basis_size, DdmDomain *dom, DomainPrep *out,
                   const double *scaling) {
    *out = DomainPrep();
    if (basis_size == 0) return BBFMM_OK;
    const int n = static_cast<int>(dom->idx.size());
    std::vector<double> loc(static_cast<size_t>(d) * n); // the domain's coordinates, gathered once
    for (int a = 0; a < d; ++a)
        for (int i = 0; i < n; ++i) loc[static_cast<size_t>(a) * n + i] = pts[a * ld + dom->idx[i]];
    // get_cheb_cube_scaling_factors (common.rs:299-322) of the domain's points
    for (int a = 0; a < d; ++a) {
        const double *xa = &loc[static_cast<size_t>(a) * n];
        double lo = xa[0], hi = lo;
        for (int i = 0; i < n; ++i) {
            lo = std::min(lo, xa[i]);
            hi = std::max(hi, xa[i]);
        }
        out->tr[a] = (hi + lo) / 2.0;
        out->sc[a] = (hi - lo) / 2.0;
        if (out->sc[a] == 0.0) out->sc[a] = 1.0;
        if (scaling) {
            out->tr[a] = scaling[a];
            out->sc[a] = scaling[3 + a];
        }
    }
    // evaluate_monomials (polynomials.rs:30-74), n x basis_size row-major
    std::vector<double> mono(static_cast<size_t>(n) * basis_size, 0.0);
    for (int i = 0; i < n; ++i) {
        double sx[3] = {0, 0, 0};
        for (int a = 0; a < d; ++a) sx[a] = (loc[static_cast<size_t>(a) * n + i] - out->tr[a]) / out->sc[a];
        monomial_row(sx, d, degree, &mono[static_cast<size_t>(i) * basis_size], 1);
    }
    // rank and unisolvent columns (domain.rs:186-212)
    std::vector<int> piv;
    std::vector<double> rd;
    pivoted_qr(mono, n, basis_size, &piv, &rd);
    int rank = 0;
    for (size_t j = 0; j < rd.size(); ++j)
        if (std::fabs(rd[j]) > 1e-10 * std::fabs(rd[0])) ++rank;
    if (rank == 0) return BBFMM_BAD_ARGUMENT;
    out->k = rank;
    out->cols.assign(piv.begin(), piv.begin() + rank);
    std::sort(out->cols.begin(), out->cols.end());
    // special points: pivoted QR of the transposed reduced monomial matrix (domain.rs:214-224)
    std::vector<double> mt(static_cast<size_t>(rank) * n);
    for (int a = 0; a < rank; ++a)
        for (int i = 0; i < n; ++i) mt[static_cast<size_t>(a) * n + i] = mono[static_cast<size_t>(i) * basis_size + out->cols[a]];
    std::vector<int> pivr;
    pivoted_qr(mt, rank, n, &pivr, &rd);
    std::vector<int> special(pivr.begin(), pivr.begin() + rank);
    std::sort(special.begin(), special.end());
    std::vector<uint8_t> is_sp(static_cast<size_t>(n), 0);
    for (int sidx : special) is_sp[sidx] = 1;
    std::vector<int> order(special);
    for (int i = 0; i < n; ++i)
        if (!is_sp[i]) order.push_back(i);
    std::vector<int64_t> nidx(static_cast<size_t>(n));
    std::vector<uint8_t> nint(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        nidx[i] = dom->idx[order[i]];
        nint[i] = dom->internal[order[i]];
    }
    dom->idx.swap(nidx);
    dom->internal.swap(nint);
    out->xyz.resize(static_cast<size_t>(d) * n);
    for (int a = 0; a < d; ++a)
        for (int i = 0; i < n; ++i) out->xyz[static_cast<size_t>(a) * n + i] = loc[static_cast<size_t>(a) * n + order[i]];
    // Lagrange coefficients on the special points and Q = -(N_ns lag)^T (domain.rs:296-307)
    out->sp_mono.resize(static_cast<size_t>(rank) * rank);
    for (int a = 0; a < rank; ++a)
        for (int c = 0; c < rank; ++c)
            out->sp_mono[static_cast<size_t>(a) * rank + c] = mono[static_cast<size_t>(order[a]) * basis_size + out->cols[c]];
    std::vector<double> lag;
    if (!invert_small(out->sp_mono, rank, &lag)) return BBFMM_BAD_ARGUMENT;
    const int m = n - rank;
    out->q.assign(static_cast<size_t>(rank) * m, 0.0);
    for (int j = 0; j < m; ++j) {
        const double *row = &mono[static_cast<size_t>(order[rank + j]) * basis_size];
        for (int a = 0; a < rank; ++a) {
            double s = 0.0;
            for (int c = 0; c < rank; ++c) s += row[out->cols[c]] * lag[static_cast<size_t>(c) * rank + a];
            out->q[static_cast<size_t>(a) * m + j] = -s;
        }
    }
    return BBFMM_OK;
}

#define DHIP(x)                                                                                                      \
    do {                                                                                                             \
        if ((x) != hipSuccess) return BBFMM_DEVICE_ERROR;                                                            \
    } while (0)

template <class T, class A> static int up(T **dst, const std::vector<T, A> &v, hipStream_t s) {
    *dst = nullptr;
    if (v.empty()) return BBFMM_OK;
    DHIP(hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(T)));
    DHIP(hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return BBFMM_OK;
}

// Fallback for a local system whose Cholesky factorisation fails: the reference switches that domain to a
// Bunch-Kaufman LBL^T solve (DomainSolver::new, domain.rs:60-68; linalg.rs:514-616).  Here the host
// assembles the same Q^T A Q (the formula of ddm_assemble_kernel), inverts it by Gauss-Jordan
// elimination with partial pivoting and stores the symmetrised inverse packed like a factor; the solve
// kernel multiplies by it.  Meant for the occasional small domain, not for speed.
static int host_domain_inverse(const PodDoubles *xyz, int64_t o, int k, int m, const double *q,
                               const KernelSpec &ks, double nugget, std::vector<double> *packed) {
    auto phi = [&](int64_t a, int64_t b) {
        const double dx = xyz[0][a] - xyz[0][b], dy = xyz[1][a] - xyz[1][b], dz = xyz[2][a] - xyz[2][b];
        return kernel_value_r2_rt(ks, dx * dx + dy * dy + dz * dz);
    };
    const size_t M = static_cast<size_t>(m);
    std::vector<double> T(static_cast<size_t>(k) * M), G(static_cast<size_t>(k) * M), a11(static_cast<size_t>(k) * k);
    for (int a = 0; a < k; ++a) {
        for (int b = 0; b < k; ++b) a11[static_cast<size_t>(a) * k + b] = phi(o + a, o + b) + (a == b ? nugget : 0.0);
        for (int j = 0; j < m; ++j) T[a * M + j] = phi(o + a, o + k + j);
    }
    for (int a = 0; a < k; ++a)
        for (int j = 0; j < m; ++j) {
            double g = T[a * M + j];
            for (int b = 0; b < k; ++b) g += a11[static_cast<size_t>(a) * k + b] * q[b * M + j];
            G[a * M + j] = g;
        }
    std::vector<double> A(M * M), inv(M * M, 0.0);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = phi(o + k + i, o + k + j) + (i == j ? nugget : 0.0);
            for (int a = 0; a < k; ++a) s += q[a * M + i] * G[a * M + j] + T[a * M + i] * q[a * M + j];
            A[i * M + j] = s;
            A[j * M + i] = s;
        }
    for (int i = 0; i < m; ++i) inv[i * M + i] = 1.0;
    for (int c = 0; c < m; ++c) {
        int p = c;
        for (int r = c + 1; r < m; ++r)
            if (std::fabs(A[r * M + c]) > std::fabs(A[p * M + c])) p = r;
        if (!(std::fabs(A[p * M + c]) > 0.0)) return BBFMM_UNSUPPORTED; // singular local system
        if (p != c)
            for (int j = 0; j < m; ++j) {
                std::swap(A[p * M + j], A[c * M + j]);
                std::swap(inv[p * M + j], inv[c * M + j]);
            }
        const double piv = 1.0 / A[c * M + c];
        for (int j = 0; j < m; ++j) {
            A[c * M + j] *= piv;
            inv[c * M + j] *= piv;
        }
        for (int r = 0; r < m; ++r) {
            if (r == c) continue;
            const double f = A[r * M + c];
            if (f == 0.0) continue;
            for (int j = 0; j < m; ++j) {
                A[r * M + j] -= f * A[c * M + j];
                inv[r * M + j] -= f * inv[c * M + j];
            }
        }
    }
    packed->resize(M * (M + 1) / 2);
    size_t e = 0;
    for (int c = 0; c < m; ++c) // packed lower triangle, column by column (ddm_kernels.hip pk)
        for (int r = c; r < m; ++r) (*packed)[e++] = 0.5 * (inv[r * M + c] + inv[c * M + r]);
    return BBFMM_OK;
}

// ---- pivoted LU of one large domain through rocSOLVER, bound at run time (dlopen): the library is only needed
// when a coarse system of more than 2,048 points is not positive definite (e.g. a negative nugget), where the
// reference switches from Cholesky to a Bunch-Kaufman LBL^T factorisation (domain.rs:60-68, linalg.rs:514-616).
namespace {
struct RocSolverApi {
    void *lib_solver = nullptr, *lib_blas = nullptr;
    int (*create_handle)(void **) = nullptr;
    int (*destroy_handle)(void *) = nullptr;
    int (*set_stream)(void *, hipStream_t) = nullptr;
    int (*dgetrf)(void *, int, int, double *, int, int *, int *) = nullptr;
    int (*dgetrs)(void *, int, int, int, double *, int, const int *, double *, int) = nullptr;
    int (*dsyevd)(void *, int, int, int, double *, int, double *, double *, int *) = nullptr;
    bool ok = false;
};
RocSolverApi &rocsolver_api() {
    static RocSolverApi api = [] {
        RocSolverApi a;
        for (const char *name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"})
            if (!a.lib_blas) a.lib_blas = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        for (const char *name : {"librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so"})
            if (!a.lib_solver) a.lib_solver = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!a.lib_solver) return a;
        auto sym = [&](const char *n) {
            void *p = dlsym(a.lib_solver, n);
            if (!p && a.lib_blas) p = dlsym(a.lib_blas, n);
            return p;
        };
        a.create_handle = reinterpret_cast<int (*)(void **)>(sym("rocblas_create_handle"));
        a.destroy_handle = reinterpret_cast<int (*)(void *)>(sym("rocblas_destroy_handle"));
        a.set_stream = reinterpret_cast<int (*)(void *, hipStream_t)>(sym("rocblas_set_stream"));
        a.dgetrf = reinterpret_cast<int (*)(void *, int, int, double *, int, int *, int *)>(sym("rocsolver_dgetrf"));
        a.dgetrs = reinterpret_cast<int (*)(void *, int, int, int, double *, int, const int *, double *, int)>(sym("rocsolver_dgetrs"));
        a.dsyevd = reinterpret_cast<int (*)(void *, int, int, int, double *, int, double *, double *, int *)>(sym("rocsolver_dsyevd"));
        a.ok = a.create_handle && a.destroy_handle && a.set_stream && a.dgetrf && a.dgetrs;
        return a;
    }();
    return api;
}
} // namespace

// Eigen-decomposition of a symmetric n x n matrix on the device (rocSOLVER dsyevd, bound on demand): d_a (column-major,
// lower triangle read) is overwritten with the eigenvectors, d_eval gets the eigenvalues in ascending order.
// BBFMM_UNSUPPORTED when rocSOLVER is not on this machine (the caller falls back to the host Jacobi sweep).
int device_symmetric_eigen(int n, double *d_a, double *d_eval, hipStream_t s) {
    RocSolverApi &api = rocsolver_api();
    if (!api.ok || !api.dsyevd) return BBFMM_UNSUPPORTED;
    void *handle = nullptr;
    if (api.create_handle(&handle) != 0) return BBFMM_DEVICE_ERROR;
    int rc = BBFMM_OK;
    double *d_e = nullptr;
    int *d_info = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_e), static_cast<size_t>(n) * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&d_info), sizeof(int)) != hipSuccess)
        rc = BBFMM_DEVICE_ERROR;
    if (rc == BBFMM_OK && api.set_stream(handle, s) != 0) rc = BBFMM_DEVICE_ERROR;
    if (rc == BBFMM_OK &&
        api.dsyevd(handle, 211 /* rocblas_evect_original */, 122 /* rocblas_fill_lower */, n, d_a, n, d_eval, d_e, d_info) != 0)
        rc = BBFMM_DEVICE_ERROR;
    int info = 0;
    if (rc == BBFMM_OK && (hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
                           hipStreamSynchronize(s) != hipSuccess))
        rc = BBFMM_DEVICE_ERROR;
    if (rc == BBFMM_OK && info != 0) rc = BBFMM_UNSUPPORTED; // no convergence: the caller takes the host path
    if (d_e) (void)hipFree(d_e);
    if (d_info) (void)hipFree(d_info);
    (void)api.destroy_handle(handle);
    return rc;
}

int big_lu_factor(DdmLevelSolver *lv, hipStream_t s) {
    RocSolverApi &api = rocsolver_api();
    if (!api.ok) return BBFMM_UNSUPPORTED; // no rocSOLVER on this machine: the large indefinite system cannot be solved
    const int m = lv->max_m;
    DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_lu), static_cast<size_t>(m) * m * sizeof(double)));
    DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_ipiv), (static_cast<size_t>(m) + 1) * sizeof(int)));
    launch_ddm_unpack_symmetric(lv->d_fac, m, lv->d_lu, s);
    if (api.create_handle(&lv->lu_handle) != 0) {
        lv->lu_handle = nullptr;
        return BBFMM_DEVICE_ERROR;
    }
    int *d_info = lv->d_ipiv + m;
    if (api.set_stream(lv->lu_handle, s) != 0 || api.dgetrf(lv->lu_handle, m, m, lv->d_lu, m, lv->d_ipiv, d_info) != 0) {
        big_lu_release(lv);
        return BBFMM_DEVICE_ERROR;
    }
    int info = 0;
    DHIP(hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    return info == 0 ? BBFMM_OK : BBFMM_UNSUPPORTED; // info > 0: exactly singular
}

int big_lu_solve(const DdmLevelSolver &lv, double *d_rhs, hipStream_t s) {
    RocSolverApi &api = rocsolver_api();
    if (!api.ok || !lv.lu_handle) return BBFMM_DEVICE_ERROR;
    const int m = lv.max_m;
    if (api.set_stream(lv.lu_handle, s) != 0) return BBFMM_DEVICE_ERROR;
    return api.dgetrs(lv.lu_handle, 111 /* rocblas_operation_none */, m, 1, lv.d_lu, m, lv.d_ipiv, d_rhs, m) == 0 ? BBFMM_OK
                                                                                                               : BBFMM_DEVICE_ERROR;
}

void big_lu_release(DdmLevelSolver *lv) {
    if (lv->lu_handle && rocsolver_api().ok) (void)rocsolver_api().destroy_handle(lv->lu_handle);
    lv->lu_handle = nullptr;
}

int ddm_level_build(const double *pts, int64_t ld, int d, DdmLevel *level, const KernelSpec &ks, double nugget,
                    int degree, int basis_size, bool solve_for_poly, hipStream_t s, DdmLevelSolver *lv,
                    const double *scaling) {
    *lv = DdmLevelSolver();
    lv->d = d;
    lv->solve_for_poly = solve_for_poly;
    static const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what, bool sync) { // BBFMM_VERBOSE: stage times (with a stream sync per stage)
        if (!verbose) return;
        if (sync) (void)hipStreamSynchronize(s);
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm]   level build: %-22s %8.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
        t_last = t;
    };
    const int64_t nd = static_cast<int64_t>(level->leaves.size());
    lv->n_dom = nd;
    lv->prep.resize(static_cast<size_t>(nd));
    std::vector<int> rcs(static_cast<size_t>(nd), BBFMM_OK);
    parallel_for(nd, 1, [&](int64_t i) {
        rcs[i] = prepare_domain(pts, ld, d, degree, basis_size, &level->leaves[i], &lv->prep[i], scaling);
    });
    for (int rc : rcs)
        if (rc != BBFMM_OK) return rc;
    lap("host prep (QR, Q)", false);
    lv->dom_off.assign(static_cast<size_t>(nd) + 1, 0);
    lv->q_off.assign(static_cast<size_t>(nd) + 1, 0);
    lv->fac_off.assign(static_cast<size_t>(nd) + 1, 0);
    lv->k.resize(static_cast<size_t>(nd));
    for (int64_t i = 0; i < nd; ++i) {
        const int64_t n = static_cast<int64_t>(level->leaves[i].idx.size());
        const int64_t kk = lv->prep[i].k, m = n - kk;
        lv->k[i] = static_cast<int32_t>(kk);
        lv->dom_off[i + 1] = lv->dom_off[i] + n;
        lv->q_off[i + 1] = lv->q_off[i] + kk * m;
        lv->fac_off[i + 1] = lv->fac_off[i] + m * (m + 1) / 2; // packed lower triangle (ddm_kernels.hip pk)
        lv->max_m = std::max<int>(lv->max_m, static_cast<int>(m));
    }
    lv->n_entries = lv->dom_off[nd];
    // (20M entries and 80M Q values on level 0 of 10M points: no serial zero fill, every element is written below)
    PodDoubles xyz[3];
    std::vector<uint8_t> internal(static_cast<size_t>(lv->n_entries));
    lv->gidx_h.resize(static_cast<size_t>(lv->n_entries));
    for (int a = 0; a < 3; ++a) xyz[a].resize(static_cast<size_t>(lv->n_entries));
    PodDoubles q(static_cast<size_t>(lv->q_off[nd]));
    parallel_for(nd, 1, [&](int64_t i) {
        const DdmDomain &dom = level->leaves[i];
        const int64_t o = lv->dom_off[i];
        const size_t ne = dom.idx.size();
        std::vector<double> &pxyz = lv->prep[i].xyz; // (empty without a polynomial part: gather here)
        for (size_t e = 0; e < ne; ++e) {
            lv->gidx_h[o + e] = dom.idx[e];
            internal[o + e] = dom.internal[e];
            for (int a = 0; a < 3; ++a)
                xyz[a][o + e] = a >= d ? 0.0 : (pxyz.empty() ? pts[a * ld + dom.idx[e]] : pxyz[static_cast<size_t>(a) * ne + e]);
        }
        std::vector<double>().swap(pxyz);
        std::copy(lv->prep[i].q.begin(), lv->prep[i].q.end(), q.begin() + lv->q_off[i]);
    });
    lap("host pack", false);
    int rc;
    for (int a = 0; a < 3; ++a)
        if ((rc = up(&lv->d_xyz[a], xyz[a], s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_gidx, lv->gidx_h, s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_dom_off, lv->dom_off, s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_q_off, lv->q_off, s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_fac_off, lv->fac_off, s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_k, lv->k, s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_internal, internal, s)) != BBFMM_OK) return rc;
    if ((rc = up(&lv->d_q, q, s)) != BBFMM_OK) return rc;
    if (!q.empty()) {
        DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_t), q.size() * sizeof(double)));
        DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_g), q.size() * sizeof(double)));
    }
    DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_fac), static_cast<size_t>(std::max<int64_t>(lv->fac_off[nd], 1)) * sizeof(double)));
    DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_work), static_cast<size_t>(std::max<int64_t>(lv->n_entries, 1)) * (ddm_level_is_big(*lv) ? 3 : 1) * sizeof(double)));
    if (ddm_level_is_big(*lv)) {
        DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_linv), static_cast<size_t>((lv->max_m + 63) / 64) * 4096 * sizeof(double)));
        DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_binv),
                       static_cast<size_t>((lv->max_m + kBigSolveBlock - 1) / kBigSolveBlock) * kBigSolveBlock * kBigSolveBlock * sizeof(double)));
    }
    int *d_fail = nullptr; // one flag per domain
    DHIP(hipMalloc(reinterpret_cast<void **>(&d_fail), static_cast<size_t>(std::max<int64_t>(nd, 1)) * sizeof(int)));
    DHIP(hipMemsetAsync(d_fail, 0, static_cast<size_t>(std::max<int64_t>(nd, 1)) * sizeof(int), s));
    lap("alloc + upload", true);
    launch_ddm_prep(ks, nugget, d, *lv, s);
    launch_ddm_assemble(ks, nugget, d, *lv, s);
    lap("assemble", true);
    launch_ddm_cholesky(*lv, d_fail, s);
    lap("cholesky", true);
    std::vector<int> fail(static_cast<size_t>(std::max<int64_t>(nd, 1)), 0);
    DHIP(hipMemcpyAsync(fail.data(), d_fail, fail.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    (void)hipFree(d_fail);
    DHIP(hipGetLastError());
    std::vector<int64_t> failed;
    for (int64_t i = 0; i < nd; ++i)
        if (fail[static_cast<size_t>(i)]) failed.push_back(i);
    if (failed.empty()) {
        if (ddm_level_is_big(*lv)) {
            launch_ddm_big_block_inverses(*lv, s);
            lap("inverses of the 1024-blocks", true);
            DHIP(hipGetLastError());
        }
        return BBFMM_OK;
    }
    // local systems that are not positive definite: host fallback per domain; the one large coarse matrix
    // (which would take the host hours) is assembled again, unpacked and factorised by pivoted LU on the device
    if (ddm_level_is_big(*lv)) {
        launch_ddm_assemble(ks, nugget, d, *lv, s); // the failed factorisation overwrote the packed matrix
        rc = big_lu_factor(lv, s);
        lap("pivoted LU of the coarse domain (LBLT role)", true);
        return rc;
    }
    lv->n_fallback = static_cast<int>(failed.size());
    std::vector<uint8_t> mode(static_cast<size_t>(nd), 0);
    std::vector<int> frc(failed.size(), BBFMM_OK);
    std::vector<std::vector<double>> inv(failed.size());
    parallel_for(static_cast<int64_t>(failed.size()), 1, [&](int64_t f) {
        const int64_t i = failed[static_cast<size_t>(f)];
        const int kk = lv->k[i], m = static_cast<int>(lv->dom_off[i + 1] - lv->dom_off[i]) - kk;
        frc[f] = host_domain_inverse(xyz, lv->dom_off[i], kk, m, q.data() + lv->q_off[i], ks, nugget, &inv[f]);
    });
    for (size_t f = 0; f < failed.size(); ++f) {
        if (frc[f] != BBFMM_OK) return frc[f];
        const int64_t i = failed[f];
        mode[static_cast<size_t>(i)] = 1;
        DHIP(hipMemcpyAsync(lv->d_fac + lv->fac_off[i], inv[f].data(), inv[f].size() * sizeof(double), hipMemcpyHostToDevice, s));
    }
    if ((rc = up(&lv->d_mode, mode, s)) != BBFMM_OK) return rc;
    DHIP(hipMalloc(reinterpret_cast<void **>(&lv->d_tmp), static_cast<size_t>(std::max<int64_t>(lv->n_entries, 1)) * sizeof(double)));
    DHIP(hipStreamSynchronize(s));
    lap("host fallback (LBLT role)", false);
    return BBFMM_OK;
}

void ddm_level_free(DdmLevelSolver *lv) {
    for (int a = 0; a < 3; ++a) (void)hipFree(lv->d_xyz[a]);
    (void)hipFree(lv->d_gidx);
    (void)hipFree(lv->d_dom_off);
    (void)hipFree(lv->d_q_off);
    (void)hipFree(lv->d_fac_off);
    (void)hipFree(lv->d_k);
    (void)hipFree(lv->d_internal);
    (void)hipFree(lv->d_q);
    (void)hipFree(lv->d_t);
    (void)hipFree(lv->d_g);
    (void)hipFree(lv->d_fac);
    (void)hipFree(lv->d_work);
    (void)hipFree(lv->d_mode);
    (void)hipFree(lv->d_linv);
    (void)hipFree(lv->d_binv);
    (void)hipFree(lv->d_tmp);
    (void)hipFree(lv->d_lu);
    (void)hipFree(lv->d_ipiv);
    big_lu_release(lv);
    *lv = DdmLevelSolver();
}

int ddm_level_solve(const DdmLevelSolver &lv, const double *d_values, double *d_out, bool all_points, hipStream_t s) {
    const int rc = launch_ddm_solve(lv, d_values, d_out, all_points, s);
    if (rc != BBFMM_OK) return rc;
    return hipGetLastError() == hipSuccess ? BBFMM_OK : BBFMM_DEVICE_ERROR;
}

} // namespace bbfmm
