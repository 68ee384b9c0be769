// The overlapping Schwarz preconditioner of ferreus_rbf (preconditioning/schwarz.rs:32-155) behind the
// C ABI: restricted additive Schwarz inside a level (local solves on the device, ddm_solver.hpp),
// multiplicative between the levels with the coarse domain as smoother, two partial matvecs per fine
// level through the BBFMM tree (IterativeSolver::precon, rbf.rs:140-155).  SURVEY.md 8(f)-1.
// Vectors are host arrays (the FGMRES driver is a host driver); the residual of a level goes to the
// device for the batched local solves and the correction comes back.
#include "../../include/ferreus_bbfmm_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "ddm.hpp"
#include "ddm_solver.hpp"
#include "device.hpp"
#include "parallel.hpp"

namespace {
using namespace bbfmm;

struct Schwarz {
    bbfmm_handle *tree = nullptr;
    int64_t n = 0;
    int d = 0, degree = -1, basis = 0;
    double nugget = 0.0;
    KernelSpec ks{};
    DdmTree ddm;
    std::vector<DdmLevelSolver> levels;
    std::vector<double> mono;  // N x basis column-major, the solver's global monomial matrix (rbf.rs:485-491)
    std::vector<double> ortho; // N x basis column-major, thin Q of mono (rbf.rs:493-495)
    std::vector<double> a_special; // coarse domain: k x n_coarse rows of A (domain.rs:352-355), row-major
    std::vector<double> coarse_xyz; // coarse domain points (domain order), 3 x n_c
    hipStream_t stream = nullptr;
    double *d_in = nullptr, *d_out = nullptr;
    // levels well under N points exchange only their own rows with the device: per level the point indices on
    // the device, one pinned buffer [values in | coefficients out] and its device twin
    std::vector<int32_t *> d_lidx;
    double *h_comp = nullptr, *d_comp = nullptr;
    int64_t comp_cap = 0;
    std::vector<double> res, tmp, s1;
    double t_matvec = 0, t_solve = 0, t_host = 0; // BBFMM_VERBOSE: seconds per apply
    std::vector<double> t_level, t_level_mv;      // per level: local solves, partial matvecs
    ~Schwarz() {
        for (auto &lv : levels) ddm_level_free(&lv);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        for (int32_t *p : d_lidx)
            if (p) (void)hipFree(p);
        if (d_comp) (void)hipFree(d_comp);
        if (h_comp) (void)hipHostFree(h_comp);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

int partial_matvec(Schwarz &S, const double *w, const std::vector<int64_t> &idx, double *y) {
    return bbfmm_fast_matrix_vector_product(S.tree, w, S.n + S.basis, S.basis, idx.data(),
                                            static_cast<int64_t>(idx.size()), S.basis ? S.mono.data() : nullptr,
                                            S.n, S.nugget, y);
}

// res = rg - matvec(sl, rows idx); then the level's local solves into s1 (zero elsewhere)
int level_correction(Schwarz &S, size_t li, const double *rg, const double *sl, bool coarse, bool add_poly) {
    const int64_t nt = S.n + S.basis;
    const DdmLevel &L = S.ddm.levels[li];
    int rc = BBFMM_OK;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    auto t0 = now();
    if (sl) {
        rc = partial_matvec(S, sl, L.point_indices, S.tmp.data());
        if (rc) return rc;
        S.t_matvec += secs(t0, now());
        if (S.t_level_mv.size() < S.levels.size()) S.t_level_mv.resize(S.levels.size(), 0.0);
        S.t_level_mv[li] += secs(t0, now());
        t0 = now();
    }
    const int64_t nl = static_cast<int64_t>(L.point_indices.size());
    const bool compact = li < S.d_lidx.size() && S.d_lidx[li] != nullptr;
    if (compact) {
        // only the level's rows matter to its domains (they read nothing else, and every row of the level is
        // internal to exactly one domain): residual on those rows, up, solve, coefficients of those rows back
        const int64_t *idx = L.point_indices.data();
        parallel_for_chunks(nl, 1 << 15, [&](int64_t b, int64_t e) {
            for (int64_t j = b; j < e; ++j) {
                const double v = sl ? rg[idx[j]] - S.tmp[idx[j]] : rg[idx[j]];
                S.h_comp[j] = v;
                S.res[idx[j]] = v; // (the coarse domain's polynomial recovery reads its special points here)
            }
        });
        if (hipMemcpyAsync(S.d_comp, S.h_comp, static_cast<size_t>(nl) * sizeof(double), hipMemcpyHostToDevice, S.stream) != hipSuccess)
            return BBFMM_DEVICE_ERROR;
        launch_scatter_output(S.d_comp, nl, 1, S.d_lidx[li], S.d_in, S.n, 0, S.stream);
        rc = ddm_level_solve(S.levels[li], S.d_in, S.d_out, coarse, S.stream);
        if (rc) return rc;
        launch_gather_rows(S.d_out, S.n, 1, S.d_lidx[li], nl, S.d_comp + S.comp_cap, nl, S.stream);
        if (hipMemcpyAsync(S.h_comp + S.comp_cap, S.d_comp + S.comp_cap, static_cast<size_t>(nl) * sizeof(double),
                           hipMemcpyDeviceToHost, S.stream) != hipSuccess)
            return BBFMM_DEVICE_ERROR;
        parallel_for_chunks(nt, 1 << 18, [&](int64_t b, int64_t e) { // (beside the device work)
            std::memset(S.s1.data() + b, 0, static_cast<size_t>(e - b) * sizeof(double));
        });
        if (hipStreamSynchronize(S.stream) != hipSuccess) return BBFMM_DEVICE_ERROR;
        parallel_for_chunks(nl, 1 << 15, [&](int64_t b, int64_t e) {
            for (int64_t j = b; j < e; ++j) S.s1[idx[j]] = S.h_comp[S.comp_cap + j];
        });
    } else {
        if (sl) {
            parallel_for_chunks(nt, 1 << 16, [&](int64_t b, int64_t e) {
                for (int64_t i = b; i < e; ++i) S.res[i] = rg[i] - S.tmp[i];
            });
        } else { // debug entry: solve the level for rg itself
            std::copy(rg, rg + nt, S.res.begin());
        }
        if (hipMemcpyAsync(S.d_in, S.res.data(), static_cast<size_t>(S.n) * sizeof(double), hipMemcpyHostToDevice, S.stream) != hipSuccess ||
            hipMemsetAsync(S.d_out, 0, static_cast<size_t>(S.n) * sizeof(double), S.stream) != hipSuccess)
            return BBFMM_DEVICE_ERROR;
        rc = ddm_level_solve(S.levels[li], S.d_in, S.d_out, coarse, S.stream);
        if (rc) return rc;
        if (hipMemcpyAsync(S.s1.data(), S.d_out, static_cast<size_t>(S.n) * sizeof(double), hipMemcpyDeviceToHost, S.stream) != hipSuccess ||
            hipStreamSynchronize(S.stream) != hipSuccess)
            return BBFMM_DEVICE_ERROR;
        for (int64_t i = S.n; i < nt; ++i) S.s1[i] = 0.0;
    }
    S.t_solve += secs(t0, now());
    if (S.t_level.size() < S.levels.size()) S.t_level.resize(S.levels.size(), 0.0);
    S.t_level[li] += secs(t0, now());
    t0 = now();
    struct HostTime { // the rest of this function is host work
        Schwarz &s;
        std::chrono::steady_clock::time_point t;
        ~HostTime() { s.t_host += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
    } host_time{S, t0};
    if (!coarse) {
        if (S.basis) { // orthogonalise against the global polynomial basis (schwarz.rs:113-126)
            std::vector<double> proj(static_cast<size_t>(S.basis), 0.0);
            constexpr int64_t kChunkP = 1 << 16; // fixed chunks, partial sums combined in order: deterministic
            const int64_t nch = (S.n + kChunkP - 1) / kChunkP;
            std::vector<double> part(static_cast<size_t>(nch) * S.basis, 0.0);
            parallel_for_chunks(S.n, kChunkP, [&](int64_t lo, int64_t hi) {
                for (int b = 0; b < S.basis; ++b) {
                    double s = 0.0;
                    const double *q = &S.ortho[static_cast<size_t>(b) * S.n];
                    for (int64_t i = lo; i < hi; ++i) s += q[i] * S.s1[i];
                    part[static_cast<size_t>(lo / kChunkP) * S.basis + b] = s;
                }
            });
            for (int64_t c = 0; c < nch; ++c)
                for (int b = 0; b < S.basis; ++b) proj[b] += part[static_cast<size_t>(c) * S.basis + b];
            parallel_for_chunks(S.n, 1 << 16, [&](int64_t bb, int64_t e) {
                for (int64_t i = bb; i < e; ++i) {
                    double s = 0.0;
                    for (int b = 0; b < S.basis; ++b) s += S.ortho[static_cast<size_t>(b) * S.n + i] * proj[b];
                    S.s1[i] -= s;
                }
            });
        }
        return BBFMM_OK;
    }
    // coarse domain: polynomial 'tail' (schwarz.rs:145-151, domain.rs:452-472)
    const DdmLevelSolver &lv = S.levels[li];
    if (lv.solve_for_poly && add_poly && S.basis) {
        const DomainPrep &pp = lv.prep[0];
        const int k = pp.k;
        const int64_t nc = static_cast<int64_t>(lv.gidx_h.size());
        std::vector<double> r(static_cast<size_t>(k));
        for (int a = 0; a < k; ++a) {
            double s = S.res[lv.gidx_h[a]];
            const double *row = &S.a_special[static_cast<size_t>(a) * nc];
            for (int64_t j = 0; j < nc; ++j) s -= row[j] * S.s1[lv.gidx_h[j]];
            r[a] = s;
        }
        // solve sp_mono * poly = r (k x k, partial pivoting)
        std::vector<double> a(pp.sp_mono), x(r);
        for (int c = 0; c < k; ++c) {
            int p = c;
            for (int rr = c + 1; rr < k; ++rr)
                if (std::fabs(a[static_cast<size_t>(rr) * k + c]) > std::fabs(a[static_cast<size_t>(p) * k + c])) p = rr;
            if (p != c) {
                for (int q = 0; q < k; ++q) std::swap(a[static_cast<size_t>(p) * k + q], a[static_cast<size_t>(c) * k + q]);
                std::swap(x[p], x[c]);
            }
            for (int rr = c + 1; rr < k; ++rr) {
                const double f = a[static_cast<size_t>(rr) * k + c] / a[static_cast<size_t>(c) * k + c];
                for (int q = c; q < k; ++q) a[static_cast<size_t>(rr) * k + q] -= f * a[static_cast<size_t>(c) * k + q];
                x[rr] -= f * x[c];
            }
        }
        for (int c = k - 1; c >= 0; --c) {
            double s = x[c];
            for (int q = c + 1; q < k; ++q) s -= a[static_cast<size_t>(c) * k + q] * x[q];
            x[c] = s / a[static_cast<size_t>(c) * k + c];
        }
        // sc.subrows_mut(idx_offset, num_poly) <- poly coefficients (schwarz.rs:146-151)
        for (int a2 = 0; a2 < k; ++a2) S.s1[nt - k + a2] = x[a2];
    }
    return BBFMM_OK;
}

} // namespace

struct bbfmm_schwarz {
    Schwarz s;
};

extern "C" {

int bbfmm_schwarz_create(bbfmm_handle *tree, const double *points, int64_t n, int32_t d, int64_t ld,
                         const bbfmm_interpolant *settings, const bbfmm_ddm_params *params, bbfmm_schwarz **out) {
    if (!out) return BBFMM_BAD_ARGUMENT;
    *out = nullptr;
    if (!tree || !points || !settings || n < 1 || d < 1 || d > 3 || ld < n) return BBFMM_BAD_ARGUMENT;
    if (settings->kernel_type < 0 || settings->kernel_type > 6 || settings->polynomial_degree < -1 ||
        settings->polynomial_degree > 2)
        return BBFMM_BAD_ARGUMENT;
    std::unique_ptr<bbfmm_schwarz> h(new (std::nothrow) bbfmm_schwarz());
    if (!h) return BBFMM_DEVICE_ERROR;
    Schwarz &S = h->s;
    S.tree = tree;
    S.n = n;
    S.d = d;
    S.degree = settings->polynomial_degree;
    S.nugget = settings->nugget;
    S.ks = make_kernel_spec(settings->kernel_type, settings->base_range, settings->total_sill);
    const int kk = S.degree + 1; // set_basis_size, interpolant_config.rs:150-178
    S.basis = S.degree < 0 ? 0 : (d == 1 ? kk : (d == 2 ? kk * (kk + 1) / 2 : kk * (kk + 1) * (kk + 2) / 6));
    DdmParams p;
    if (params) {
        p.leaf_threshold = params->leaf_threshold;
        p.overlap_quota = params->overlap_quota;
        p.coarse_ratio = params->coarse_ratio;
        p.coarse_threshold = params->coarse_threshold;
    }
    const bool verbose = std::getenv("BBFMM_VERBOSE") != nullptr; // stage times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what, int level) {
        if (!verbose) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bbfmm] schwarz: %-26s %2d %8.3f s\n", what, level, std::chrono::duration<double>(t1 - t_last).count());
        t_last = t1;
    };
    int rc = build_ddm_tree(points, n, d, ld, p, &S.ddm);
    if (rc) return rc;
    lap("domain decomposition", -1);
    if (hipStreamCreate(&S.stream) != hipSuccess) return BBFMM_DEVICE_ERROR;
    // global monomial matrix on the cube-scaled points and its thin Q (rbf.rs:418-421, 476-495)
    double gscale[6] = {0, 0, 0, 1, 1, 1}; // translation, scale of the global monomial basis
    if (S.basis) {
        double *tr = gscale, *sc = gscale + 3;
        for (int a = 0; a < d; ++a) {
            double lo = points[a * ld], hi = lo;
            for (int64_t i = 0; i < n; ++i) {
                lo = std::min(lo, points[a * ld + i]);
                hi = std::max(hi, points[a * ld + i]);
            }
            tr[a] = (hi + lo) / 2.0;
            sc[a] = (hi - lo) / 2.0;
            if (sc[a] == 0.0) sc[a] = 1.0;
        }
        S.mono.assign(static_cast<size_t>(n) * S.basis, 0.0);
        for (int64_t i = 0; i < n; ++i) {
            double sx[3] = {0, 0, 0};
            for (int a = 0; a < d; ++a) sx[a] = (points[a * ld + i] - tr[a]) / sc[a];
            S.mono[i] = 1.0;
            if (S.degree >= 1)
                for (int a = 0; a < d; ++a) S.mono[static_cast<size_t>(1 + a) * n + i] = sx[a];
            if (S.degree == 2) {
                int c = 1 + d;
                for (int a = 0; a < d; ++a)
                    for (int b = a; b < d; ++b) S.mono[static_cast<size_t>(c++) * n + i] = sx[a] * sx[b];
            }
        }
        S.ortho = S.mono; // modified Gram-Schmidt, twice
        for (int pass = 0; pass < 2; ++pass)
            for (int b = 0; b < S.basis; ++b) {
                double *qb = &S.ortho[static_cast<size_t>(b) * n];
                for (int c = 0; c < b; ++c) {
                    const double *qc = &S.ortho[static_cast<size_t>(c) * n];
                    double s = 0.0;
                    for (int64_t i = 0; i < n; ++i) s += qc[i] * qb[i];
                    for (int64_t i = 0; i < n; ++i) qb[i] -= s * qc[i];
                }
                double nn = 0.0;
                for (int64_t i = 0; i < n; ++i) nn += qb[i] * qb[i];
                nn = std::sqrt(nn);
                if (nn == 0.0) return BBFMM_BAD_ARGUMENT;
                for (int64_t i = 0; i < n; ++i) qb[i] /= nn;
            }
    }
    S.levels.resize(S.ddm.levels.size());
    for (size_t li = 0; li < S.ddm.levels.size(); ++li) {
        const bool coarse = li + 1 == S.ddm.levels.size();
        // The coarse domain returns the polynomial tail of the correction.  The reference scales that
        // domain's monomials by the extents of its own points (domain.rs:171-172) although the system's
        // monomial matrix is scaled by the extents of all points (rbf.rs:418-421, 485-491), so the tail
        // belongs to a slightly different basis (relative extent mismatch of the coarse sample: the
        // coarse solve then leaves a residual of 5e-5 .. 4e-4 on its own points for a linear drift, 1e-12
        // with the global scaling).  Default: as the reference; BBFMM_FLAG_GLOBAL_SCALING: the global one.
        // (Measured: the FGMRES histories at 3M points are the same either way.)
        const bool global_scaling = (settings->flags & BBFMM_FLAG_GLOBAL_SCALING) != 0;
        rc = ddm_level_build(points, ld, d, &S.ddm.levels[li], S.ks, S.nugget, S.degree, S.basis, coarse && S.basis != 0,
                             S.stream, &S.levels[li], (coarse && S.basis != 0 && global_scaling) ? gscale : nullptr);
        if (rc) return rc;
        lap("level prepared + factorised", static_cast<int>(li));
    }
    if (S.basis) { // rows of A for the coarse domain's special points (domain.rs:352-355)
        const DdmLevelSolver &lv = S.levels.back();
        const int k = lv.prep[0].k;
        const int64_t nc = static_cast<int64_t>(lv.gidx_h.size());
        S.a_special.assign(static_cast<size_t>(k) * nc, 0.0);
        for (int a = 0; a < k; ++a)
            for (int64_t j = 0; j < nc; ++j) {
                double r2 = 0.0;
                for (int ax = 0; ax < d; ++ax) {
                    const double t = points[ax * ld + lv.gidx_h[a]] - points[ax * ld + lv.gidx_h[j]];
                    r2 += t * t;
                }
                S.a_special[static_cast<size_t>(a) * nc + j] = kernel_value_r2_rt(S.ks, r2) + (a == j ? S.nugget : 0.0);
            }
    }
    if (hipMalloc(reinterpret_cast<void **>(&S.d_in), static_cast<size_t>(n) * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&S.d_out), static_cast<size_t>(n) * sizeof(double)) != hipSuccess)
        return BBFMM_DEVICE_ERROR;
    S.res.assign(static_cast<size_t>(n + S.basis), 0.0);
    S.tmp.assign(static_cast<size_t>(n + S.basis), 0.0);
    S.s1.assign(static_cast<size_t>(n + S.basis), 0.0);
    S.d_lidx.assign(S.ddm.levels.size(), nullptr);
    for (size_t li = 0; li < S.ddm.levels.size(); ++li) {
        const auto &pi = S.ddm.levels[li].point_indices;
        if (static_cast<int64_t>(pi.size()) * 2 >= n || pi.empty()) continue; // (the finest level moves whole vectors)
        std::vector<int32_t> idx32(pi.begin(), pi.end());
        if (hipMalloc(reinterpret_cast<void **>(&S.d_lidx[li]), idx32.size() * sizeof(int32_t)) != hipSuccess ||
            hipMemcpy(S.d_lidx[li], idx32.data(), idx32.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
            return BBFMM_DEVICE_ERROR;
        S.comp_cap = std::max<int64_t>(S.comp_cap, static_cast<int64_t>(pi.size()));
    }
    if (S.comp_cap > 0 &&
        (hipHostMalloc(reinterpret_cast<void **>(&S.h_comp), static_cast<size_t>(2 * S.comp_cap) * sizeof(double), hipHostMallocDefault) != hipSuccess ||
         hipMalloc(reinterpret_cast<void **>(&S.d_comp), static_cast<size_t>(2 * S.comp_cap) * sizeof(double)) != hipSuccess))
        return BBFMM_DEVICE_ERROR;
    // the target-subset plans of the levels' partial matvecs belong to the setup, not to the first apply
    for (const DdmLevel &L : S.ddm.levels) {
        rc = bbfmm_prepare_target_subset(tree, L.point_indices.data(), static_cast<int64_t>(L.point_indices.size()));
        if (rc) return rc;
    }
    lap("target-subset plans", -1);
    *out = h.release();
    return BBFMM_OK;
}

void bbfmm_schwarz_destroy(bbfmm_schwarz *h) { delete h; }

int64_t bbfmm_schwarz_basis_size(const bbfmm_schwarz *h) { return h ? h->s.basis : -1; }
int32_t bbfmm_schwarz_num_levels(const bbfmm_schwarz *h) { return h ? static_cast<int32_t>(h->s.ddm.levels.size()) : 0; }
const double *bbfmm_schwarz_monomial_matrix(const bbfmm_schwarz *h) { return (h && h->s.basis) ? h->s.mono.data() : nullptr; }

int64_t bbfmm_schwarz_level_size(const bbfmm_schwarz *h, int32_t level) {
    if (!h || level < 0 || level >= static_cast<int32_t>(h->s.ddm.levels.size())) return -1;
    return static_cast<int64_t>(h->s.ddm.levels[static_cast<size_t>(level)].point_indices.size());
}
int bbfmm_schwarz_level_points(const bbfmm_schwarz *h, int32_t level, int64_t *out) {
    if (!h || !out || level < 0 || level >= static_cast<int32_t>(h->s.ddm.levels.size())) return BBFMM_BAD_ARGUMENT;
    const auto &p = h->s.ddm.levels[static_cast<size_t>(level)].point_indices;
    std::copy(p.begin(), p.end(), out);
    return BBFMM_OK;
}
// solve_fine_level / solve_coarse_level (schwarz.rs:84-155) of one level for a given residual
int bbfmm_schwarz_debug_level_solve(bbfmm_schwarz *h, int32_t level, const double *residual, double *out,
                                    int32_t add_poly) {
    if (!h || !residual || !out || level < 0 || level >= static_cast<int32_t>(h->s.ddm.levels.size()))
        return BBFMM_BAD_ARGUMENT;
    Schwarz &S = h->s;
    const bool coarse = static_cast<size_t>(level) + 1 == S.ddm.levels.size();
    const int rc = level_correction(S, static_cast<size_t>(level), residual, nullptr, coarse, add_poly != 0);
    if (rc) return rc;
    std::copy(S.s1.begin(), S.s1.end(), out);
    return BBFMM_OK;
}

// schwarz_preconditioner (schwarz.rs:32-82) as a bbfmm_apply_fn: user = bbfmm_schwarz*, n = N + basis
int bbfmm_schwarz_apply(void *user, const double *rg, double *sl, int64_t n) {
    bbfmm_schwarz *h = static_cast<bbfmm_schwarz *>(user);
    if (!h || !rg || !sl || n != h->s.n + h->s.basis) return BBFMM_BAD_ARGUMENT;
    Schwarz &S = h->s;
    std::memset(sl, 0, static_cast<size_t>(n) * sizeof(double));
    const size_t coarse = S.ddm.levels.size() - 1;
    auto add = [&]() {
        parallel_for_chunks(n, 1 << 16, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) sl[i] += S.s1[i];
        });
    };
    int rc;
    if (coarse > 0) {
        for (size_t i = 0; i < coarse; ++i) {
            if ((rc = level_correction(S, i, rg, sl, false, false))) return rc;
            add();
            if ((rc = level_correction(S, coarse, rg, sl, true, i == coarse - 1))) return rc;
            add();
        }
    } else {
        if ((rc = level_correction(S, coarse, rg, sl, true, true))) return rc;
        add();
    }
    if (std::getenv("BBFMM_VERBOSE")) {
        std::fprintf(stderr, "[bbfmm] schwarz apply: partial matvecs %.3f s, level solves (incl. PCIe) %.3f s, host %.3f s\n",
                     S.t_matvec, S.t_solve, S.t_host);
        for (size_t l = 0; l < S.t_level.size(); ++l) {
            std::fprintf(stderr, "[bbfmm]   level %zu: %lld domains, %lld entries, max m %d: solves %.3f s, matvecs %.3f s\n", l,
                         (long long)S.levels[l].n_dom, (long long)S.levels[l].n_entries, S.levels[l].max_m, S.t_level[l],
                         l < S.t_level_mv.size() ? S.t_level_mv[l] : 0.0);
            S.t_level[l] = 0;
            if (l < S.t_level_mv.size()) S.t_level_mv[l] = 0;
        }
        S.t_matvec = S.t_solve = S.t_host = 0;
    }
    return BBFMM_OK;
}

} // extern "C"
