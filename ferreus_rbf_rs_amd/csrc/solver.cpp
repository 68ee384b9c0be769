// Host iterative solvers of ferreus_rbf (iterative_solvers.rs:38-281) behind the C ABI:
// restarted flexible GMRES (right-preconditioned, modified Gram-Schmidt, LAPACK-style Givens
// rotations), the stationary Schwarz iteration, and the RBF system operator that feeds them
// from the device matvec (rbf.rs:105-133 -> bbfmm_fast_matrix_vector_product).
// The operators are C callbacks, so the reference's closures (FMM matvec, Schwarz
// preconditioner) plug in unchanged.  Vectors live on the host as in the reference.
#include "../../include/ferreus_bbfmm_hip.h"
#include "parallel.hpp"

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

namespace {

using bbfmm::parallel_for_chunks;
constexpr int64_t kChunk = 1 << 16;

// Deterministic chunked reductions (fixed chunk boundaries, partials combined in index order).
template <class F> double reduce_chunks(int64_t n, bool take_max, F &&f) {
    const int64_t nch = (n + kChunk - 1) / kChunk;
    std::vector<double> part(static_cast<size_t>(std::max<int64_t>(nch, 1)), 0.0);
    parallel_for_chunks(n, kChunk, [&](int64_t b, int64_t e) { part[static_cast<size_t>(b / kChunk)] = f(b, e); });
    double r = 0.0;
    for (double p : part) r = take_max ? std::max(r, p) : r + p;
    return r;
}
double dot(const double *a, const double *b, int64_t n) {
    return reduce_chunks(n, false, [&](int64_t lo, int64_t hi) {
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) s += a[i] * b[i];
        return s;
    });
}
double norm_l2(const double *a, int64_t n) { return std::sqrt(dot(a, a, n)); }
double norm_max(const double *a, int64_t n) {
    return reduce_chunks(n, true, [&](int64_t lo, int64_t hi) {
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) s = std::max(s, std::fabs(a[i]));
        return s;
    });
}
void axpy(double alpha, const double *x, double *y, int64_t n) { // y += alpha x
    parallel_for_chunks(n, kChunk, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) y[i] += alpha * x[i];
    });
}
void scale_to(double alpha, const double *x, double *y, int64_t n) { // y = alpha x
    parallel_for_chunks(n, kChunk, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) y[i] = alpha * x[i];
    });
}
void sub_to(const double *a, const double *b, double *y, int64_t n) { // y = a - b
    parallel_for_chunks(n, kChunk, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) y[i] = a[i] - b[i];
    });
}

double progress_from_rel(double cur, double start, double target) { // progress.rs:124-130
    if (cur <= target) return 1.0;
    return (std::log10(start) - std::log10(cur)) / (std::log10(start) - std::log10(target));
}

// get_solution (iterative_solvers.rs:174-183): x += Z[:, :i] (H[:i,:i]^-1 g[:i])
template <class ZBasis>
void add_solution(const std::vector<double> &h, int ldh, const std::vector<double> &g, const ZBasis &z, int i, double *x,
                  int64_t n) {
    std::vector<double> y(g.begin(), g.begin() + i);
    for (int r = i - 1; r >= 0; --r) {
        double s = y[r];
        for (int c = r + 1; c < i; ++c) s -= h[static_cast<size_t>(r) + static_cast<size_t>(c) * ldh] * y[c];
        y[r] = s / h[static_cast<size_t>(r) + static_cast<size_t>(r) * ldh];
    }
    for (int c = 0; c < i; ++c) axpy(y[c], z[c].data(), x, n);
}

} // namespace

extern "C" {

void bbfmm_givens_rotation(double f, double g, double *c, double *s, double *r) { // iterative_solvers.rs:185-227
    const double safmin = DBL_MIN, safmax = DBL_MAX;
    const double rtmin = std::sqrt(safmin), rtmax = std::sqrt(safmax / 2.0);
    if (g == 0.0) {
        *c = 1.0, *s = 0.0, *r = f;
        return;
    }
    if (f == 0.0) {
        *c = 0.0, *s = std::copysign(1.0, g), *r = std::fabs(g);
        return;
    }
    const double f1 = std::fabs(f), g1 = std::fabs(g);
    if (f1 >= rtmin && f1 < rtmax && g1 >= rtmin && g1 < rtmax) {
        const double rr = std::copysign(std::sqrt(f * f + g * g), f);
        *c = f1 / std::fabs(rr), *s = g / rr, *r = rr;
    } else {
        const double u = std::min(std::max(std::max(f1, g1), safmin), safmax);
        const double fs = f / u, gs = g / u;
        const double mag = std::sqrt(fs * fs + gs * gs);
        *c = std::fabs(fs) / mag, *s = gs / mag, *r = std::copysign(mag, f) * u;
    }
}

int bbfmm_fgmres(int64_t n, bbfmm_apply_fn a, void *a_user, const double *b, bbfmm_apply_fn m, void *m_user,
                 const double *x0, int32_t max_outer_iterations, int32_t max_inner_iterations,
                 int32_t tolerance_type, double tolerance, bbfmm_iteration_fn callback, void *cb_user, double *x,
                 int64_t *iterations, double *final_residual) {
    if (n < 1 || !a || !b || !x || max_outer_iterations < 0 || max_inner_iterations < 1 ||
        (tolerance_type != BBFMM_ACCURACY_ABSOLUTE && tolerance_type != BBFMM_ACCURACY_RELATIVE))
        return BBFMM_BAD_ARGUMENT;
    const int mi = max_inner_iterations, ldh = mi + 1;
    const bool absolute = tolerance_type == BBFMM_ACCURACY_ABSOLUTE;
    // BBFMM_VERBOSE: where the solve's wall time goes (operator, preconditioner, host vector work)
    struct Clock {
        double t_a = 0, t_m = 0;
        std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
        ~Clock() {
            if (!std::getenv("BBFMM_VERBOSE")) return;
            const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
            std::fprintf(stderr, "[bbfmm] fgmres: total %.3f s = operator %.3f + preconditioner %.3f + host vectors %.3f\n", total,
                         t_a, t_m, total - t_a - t_m);
        }
    } clock;
    auto timed = [](double &acc, bbfmm_apply_fn f, void *user, const double *in, double *out, int64_t len) {
        const auto t0 = std::chrono::steady_clock::now();
        const int r = f(user, in, out, len);
        acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return r;
    };
    if (x0) std::memcpy(x, x0, static_cast<size_t>(n) * sizeof(double));
    else std::memset(x, 0, static_cast<size_t>(n) * sizeof(double));
    std::vector<double> r(static_cast<size_t>(n)), w(static_cast<size_t>(n)), wj(static_cast<size_t>(n));
    int rc = timed(clock.t_a, a, a_user, x, wj.data(), n);
    if (rc) return rc;
    sub_to(b, wj.data(), r.data(), n);
    const double beta = absolute ? norm_max(r.data(), n) : norm_l2(r.data(), n);
    int64_t iteration = 1;
    double res_norm = absolute ? beta : 1.0;
    if (iterations) *iterations = 0;
    if (final_residual) *final_residual = res_norm;
    // Krylov and preconditioned bases: allocated without initialisation -- every vector is written in full
    // (threaded) before it is read, so the pages are first touched in parallel instead of by one thread
    struct Basis {
        std::vector<std::unique_ptr<double[]>> col;
        Basis(int count, int64_t len) {
            for (int i = 0; i < count; ++i) col.emplace_back(new double[static_cast<size_t>(len)]);
        }
        struct Col {
            double *p;
            double *data() const { return p; }
        };
        Col operator[](size_t i) const { return Col{col[i].get()}; }
    };
    Basis v(mi + 1, n), z(mi, n);
    std::vector<double> h(static_cast<size_t>(ldh) * mi), g(static_cast<size_t>(mi + 1)), cs(mi), sn(mi);
    for (int outer = 0; outer < max_outer_iterations; ++outer) {
        std::fill(h.begin(), h.end(), 0.0);
        std::fill(g.begin(), g.end(), 0.0);
        const double r_norm = norm_l2(r.data(), n);
        // The reference divides by r_norm unguarded (NaNs when b = A x0 exactly); a zero residual
        // is returned as converged here.
        if (r_norm == 0.0) break;
        scale_to(1.0 / r_norm, r.data(), v[0].data(), n);
        g[0] = r_norm;
        for (int j = 0; j < mi; ++j) {
            if (m) {
                rc = timed(clock.t_m, m, m_user, v[j].data(), w.data(), n);
                if (rc) return rc;
            } else {
                std::memcpy(w.data(), v[j].data(), static_cast<size_t>(n) * sizeof(double));
            }
            std::memcpy(z[j].data(), w.data(), static_cast<size_t>(n) * sizeof(double));
            rc = timed(clock.t_a, a, a_user, w.data(), wj.data(), n);
            if (rc) return rc;
            for (int i = 0; i <= j; ++i) { // modified Gram-Schmidt
                const double hij = dot(v[i].data(), wj.data(), n);
                h[static_cast<size_t>(i) + static_cast<size_t>(j) * ldh] = hij;
                axpy(-hij, v[i].data(), wj.data(), n);
            }
            const double norm = norm_l2(wj.data(), n);
            auto H = [&](int rr, int cc) -> double & { return h[static_cast<size_t>(rr) + static_cast<size_t>(cc) * ldh]; };
            H(j + 1, j) = norm;
            for (int i = 0; i < j; ++i) { // previous rotations
                const double temp = cs[i] * H(i, j) + sn[i] * H(i + 1, j);
                H(i + 1, j) = -sn[i] * H(i, j) + cs[i] * H(i + 1, j);
                H(i, j) = temp;
            }
            double c, s, rr;
            bbfmm_givens_rotation(H(j, j), H(j + 1, j), &c, &s, &rr);
            H(j, j) = c * H(j, j) + s * H(j + 1, j);
            H(j + 1, j) = 0.0;
            const double temp = c * g[j] + s * g[j + 1];
            g[j + 1] = -s * g[j] + c * g[j + 1];
            g[j] = temp;
            cs[j] = c, sn[j] = s;
            if (norm != 0.0) scale_to(1.0 / norm, wj.data(), v[j + 1].data(), n);
            else scale_to(0.0, wj.data(), v[j + 1].data(), n);
            res_norm = absolute ? std::fabs(g[j + 1]) : std::fabs(g[j + 1]) / beta;
            if (iterations) *iterations = iteration;
            if (final_residual) *final_residual = res_norm;
            if (callback) callback(cb_user, iteration, res_norm, progress_from_rel(res_norm, beta, tolerance));
            if (res_norm < tolerance) {
                add_solution(h, ldh, g, z, j + 1, x, n);
                return BBFMM_OK;
            }
            ++iteration;
        }
        add_solution(h, ldh, g, z, mi, x, n); // restart update
        rc = timed(clock.t_a, a, a_user, x, wj.data(), n);
        if (rc) return rc;
        sub_to(b, wj.data(), r.data(), n);
        res_norm = absolute ? norm_max(r.data(), n) : norm_l2(r.data(), n) / beta;
        if (final_residual) *final_residual = res_norm;
        if (res_norm < tolerance) break;
    }
    if (iterations) *iterations = iteration - 1;
    return BBFMM_OK;
}

int bbfmm_schwarz_ddm_solver(int64_t n, bbfmm_apply_fn matvec, void *a_user, const double *rhs, bbfmm_apply_fn m,
                             void *m_user, int32_t max_iterations, int32_t tolerance_type, double tolerance,
                             bbfmm_iteration_fn callback, void *cb_user, double *x, int64_t *iterations,
                             double *final_residual) {
    if (n < 1 || !matvec || !rhs || !x || max_iterations < 0 ||
        (tolerance_type != BBFMM_ACCURACY_ABSOLUTE && tolerance_type != BBFMM_ACCURACY_RELATIVE))
        return BBFMM_BAD_ARGUMENT;
    const bool absolute = tolerance_type == BBFMM_ACCURACY_ABSOLUTE;
    std::vector<double> rg(rhs, rhs + n), t(static_cast<size_t>(n));
    std::memset(x, 0, static_cast<size_t>(n) * sizeof(double));
    const double beta = absolute ? norm_max(rg.data(), n) : norm_l2(rg.data(), n);
    double res_norm = beta;
    int64_t iteration = 0;
    if (m) { // without a preconditioner the reference returns the zero vector (iterative_solvers.rs:256)
        while (res_norm > tolerance && iteration < max_iterations) {
            int rc = m(m_user, rg.data(), t.data(), n);
            if (rc) return rc;
            axpy(1.0, t.data(), x, n);
            rc = matvec(a_user, x, t.data(), n);
            if (rc) return rc;
            sub_to(rhs, t.data(), rg.data(), n);
            res_norm = absolute ? norm_max(rg.data(), n) : norm_l2(rg.data(), n) / beta;
            ++iteration;
            if (callback) callback(cb_user, iteration, res_norm, progress_from_rel(res_norm, beta, tolerance));
        }
    }
    if (iterations) *iterations = iteration;
    if (final_residual) *final_residual = res_norm;
    return BBFMM_OK;
}

// IterativeSolver::matvec (rbf.rs:105-117) as a bbfmm_apply_fn: user = bbfmm_rbf_system*.
int bbfmm_rbf_system_apply(void *user, const double *x, double *y, int64_t n) {
    const bbfmm_rbf_system *sys = static_cast<const bbfmm_rbf_system *>(user);
    if (!sys || !sys->tree) return BBFMM_BAD_ARGUMENT;
    return bbfmm_fast_matrix_vector_product(sys->tree, x, n, sys->basis_size, nullptr, 0, sys->monomial_matrix,
                                            sys->ld_monomial, sys->nugget, y);
}

} // extern "C"
