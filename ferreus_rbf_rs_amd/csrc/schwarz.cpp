// The overlapping Schwarz preconditioner of ferreus_rbf (preconditioning/schwarz.rs:32-155) behind the
// C ABI: restricted additive Schwarz inside a level (local solves on the device, ddm_solver.hpp),
// multiplicative between the levels with the coarse domain as smoother, two partial matvecs per fine
// level through the BBFMM tree (IterativeSolver::precon, rbf.rs:140-155).  SURVEY.md 8(f)-1.
// The FGMRES driver is a host driver, so an apply takes and returns host vectors -- but inside an apply
// every vector stays in HBM: the residual goes up once, the running correction `sl`, the level
// residuals, the partial products (bbfmm_matvec_subset_device), the local solves, the write-back and the
// orthogonalisation against the polynomial basis all run on the tree's stream, and the correction comes
// down once.  The first correction of a sweep skips its partial product: `sl` is still zero there and
// K * 0 = 0 exactly (schwarz.rs:53-59 with sl = 0).
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
#include "ddm_monomials.hpp"
#include "parallel.hpp"
#include "schwarz_kernels.hpp"

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
    PodDoubles mono;           // N x basis column-major, the solver's global monomial matrix (rbf.rs:485-491)
    PodDoubles ortho;          // N x basis column-major, thin Q of mono (rbf.rs:493-495)
    std::vector<double> a_special; // coarse domain: k x n_coarse rows of A (domain.rs:352-355), row-major
    hipStream_t stream = nullptr;  // the tree's stream (not owned): products and level work stay in order
    int device = -1;               // the tree's device: the entry points bind the calling thread to it
    // device vectors of a sweep (N doubles each): incoming residual, running correction, level residual,
    // local-solve output, partial product; the orthonormal polynomial basis; projection scratch
    double *d_rg = nullptr, *d_sl = nullptr, *d_res = nullptr, *d_out = nullptr, *d_y = nullptr;
    double *d_ortho = nullptr, *d_part = nullptr, *d_proj = nullptr;
    double *d_small = nullptr;          // coarse tail: coefficients of the coarse entries + k residuals
    double *h_pin = nullptr;            // pinned staging: N doubles (+ the coarse tail download)
    int64_t small_cap = 0;
    std::vector<int32_t *> d_lidx;      // per level: its rows on the device (nullptr: all rows in order)
    std::vector<int32_t> subset_id;     // per level: registered target subset of the tree (-1: all rows)
    // Factors sharded over the ranks of a job (bbfmm_schwarz_create_sharded): this rank factorises and solves the
    // domains [dom_first, dom_first + count) of every fine level; a level's corrections -- disjoint rows, zeros for the
    // domains of the other ranks -- are summed over the ranks in the caller's exchange buffer.  The coarse domain and the
    // partial products are replicated.
    int rank = 0, world = 1;
    double *d_xchg = nullptr;   // caller's device buffer (not owned), >= the largest fine level
    int64_t xchg_cap = 0;
    bbfmm_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    std::vector<int64_t> dom_total, dom_first; // per level: domains of the decomposition, first one owned here
    double t_exchange = 0;
    double t_matvec = 0, t_solve = 0, t_host = 0; // BBFMM_VERBOSE: seconds per apply (stream synchronised per stage)
    std::vector<double> t_level, t_level_mv;      // per level: local solves, partial matvecs
    bool verbose = false;
    ~Schwarz() {
        for (auto &lv : levels) ddm_level_free(&lv);
        for (double *p : {d_rg, d_sl, d_res, d_out, d_y, d_ortho, d_part, d_proj, d_small})
            if (p) (void)hipFree(p);
        for (int32_t *p : d_lidx)
            if (p) (void)hipFree(p);
        if (h_pin) (void)hipHostFree(h_pin);
    }
};

#define HIPOK(expr)                                      \
    do {                                                 \
        if ((expr) != hipSuccess) return BBFMM_DEVICE_ERROR; \
    } while (0)

// One level of the sweep on the device vectors: d_sl += solve(d_rg - K d_sl | level rows).  have_sl = false:
// d_sl is known to be zero, the product is skipped.  The coarse domain's polynomial tail (k values) is
// returned through tail when asked for.
int level_step(Schwarz &S, size_t li, bool have_sl, bool coarse, bool add_poly, std::vector<double> *tail) {
    const DdmLevel &L = S.ddm.levels[li];
    const DdmLevelSolver &lv = S.levels[li];
    const int64_t nl = static_cast<int64_t>(L.point_indices.size());
    const int32_t *rows = S.d_lidx[li];
    auto now = [&] {
        if (S.verbose) (void)hipStreamSynchronize(S.stream);
        return std::chrono::steady_clock::now();
    };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    auto t0 = now();
    const double *src = S.d_rg;
    const bool sharded = !coarse && S.world > 1;
    // A sharded level ends in a collective: a rank that failed on the way there must still take part in it, or its peers
    // wait forever.  Its contribution is poisoned (NaN) instead, and every rank reads the verdict out of the sum.
    int local_rc = BBFMM_OK;
    if (have_sl) { // rg - matvec_partial(sl, level rows)  (schwarz.rs:53-59, 63-67)
        const int rc = bbfmm_matvec_subset_device(S.tree, S.subset_id[li], S.d_sl, S.d_y, 0);
        if (rc && !sharded) return rc;
        if (rc) local_rc = rc;
        launch_schwarz_residual(S.d_rg, S.d_y, S.d_sl, S.nugget, rows, nl, S.d_res, S.stream);
        src = S.d_res;
        if (S.verbose) {
            const double dt = secs(t0, now());
            S.t_matvec += dt;
            S.t_level_mv[li] += dt;
            t0 = now();
        }
    }
    if (sharded) launch_schwarz_scatter_rows(nullptr, rows, nl, S.d_out, S.stream); // rows of the other ranks' domains: 0
    int rc = (lv.n_dom > 0 && local_rc == BBFMM_OK) ? ddm_level_solve(lv, src, S.d_out, coarse, S.stream) : BBFMM_OK;
    if (rc && !sharded) return rc;
    if (rc) local_rc = rc;
    if (sharded) { // the level's correction = the sum over the ranks of their domains' rows (disjoint: exact)
        if (nl > S.xchg_cap) return BBFMM_BAD_ARGUMENT; // (the same on every rank: they hold the same decomposition)
        if (local_rc == BBFMM_OK && hipGetLastError() != hipSuccess) local_rc = BBFMM_DEVICE_ERROR;
        if (local_rc == BBFMM_OK)
            launch_schwarz_gather_rows(S.d_out, rows, nl, S.d_xchg, S.stream);
        else // all-ones bytes = NaN: whatever the peers add to it stays NaN
            (void)hipMemsetAsync(S.d_xchg, 0xFF, static_cast<size_t>(nl) * sizeof(double), S.stream);
        if (hipStreamSynchronize(S.stream) != hipSuccess && local_rc == BBFMM_OK) local_rc = BBFMM_DEVICE_ERROR;
        const auto tx = std::chrono::steady_clock::now();
        if (S.allreduce(S.allreduce_user, nl) != 0) return BBFMM_DEVICE_ERROR; // (the collective itself failed: nothing to agree through)
        S.t_exchange += std::chrono::duration<double>(std::chrono::steady_clock::now() - tx).count();
        double first = 0.0;
        if (nl > 0) HIPOK(hipMemcpy(&first, S.d_xchg, sizeof(double), hipMemcpyDeviceToHost));
        if (local_rc != BBFMM_OK) return local_rc;
        if (std::isnan(first)) return BBFMM_DEVICE_ERROR; // a peer failed in this level (or the iteration has diverged)
        launch_schwarz_scatter_rows(S.d_xchg, rows, nl, S.d_out, S.stream);
    }
    if (!coarse) { // solve_fine_level, schwarz.rs:84-126: internal points written back, then orthogonalised
        if (S.basis)
            launch_schwarz_project(S.d_ortho, S.n, S.basis, S.d_out, rows, nl, S.d_part, kSchwarzProjectBlocks, S.d_proj,
                                   S.stream);
        launch_schwarz_add_rows(S.d_out, rows, nl, S.d_sl, S.stream);
        if (S.basis) launch_schwarz_subtract_projection(S.d_ortho, S.n, S.basis, S.d_proj, S.d_sl, S.stream);
    } else { // solve_coarse_level, schwarz.rs:134-155
        launch_schwarz_add_rows(S.d_out, rows, nl, S.d_sl, S.stream);
    }
    HIPOK(hipGetLastError());
    if (S.verbose) {
        const double dt = secs(t0, now());
        S.t_solve += dt;
        S.t_level[li] += dt;
        t0 = now();
    }
    if (coarse && lv.solve_for_poly && add_poly && S.basis && tail) {
        // polynomial 'tail' (schwarz.rs:145-151, domain.rs:452-472): k x k system on the special points
        const DomainPrep &pp = lv.prep[0];
        const int k = pp.k;
        const int64_t nc = static_cast<int64_t>(lv.gidx_h.size());
        launch_gather_rows64(S.d_out, lv.d_gidx, nc, S.d_small, S.stream);
        launch_gather_rows64(src, lv.d_gidx, k, S.d_small + nc, S.stream);
        double *h = S.h_pin + S.n;
        HIPOK(hipMemcpyAsync(h, S.d_small, static_cast<size_t>(nc + k) * sizeof(double), hipMemcpyDeviceToHost, S.stream));
        HIPOK(hipStreamSynchronize(S.stream));
        const auto th = std::chrono::steady_clock::now();
        std::vector<double> r(static_cast<size_t>(k));
        for (int a = 0; a < k; ++a) {
            double s = h[nc + a];
            const double *row = &S.a_special[static_cast<size_t>(a) * nc];
            for (int64_t j = 0; j < nc; ++j) s -= row[j] * h[j];
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
            double sacc = x[c];
            for (int q = c + 1; q < k; ++q) sacc -= a[static_cast<size_t>(c) * k + q] * x[q];
            x[c] = sacc / a[static_cast<size_t>(c) * k + c];
        }
        tail->assign(x.begin(), x.end()); // sc.subrows_mut(idx_offset, num_poly) <- poly coefficients (schwarz.rs:146-151)
        S.t_host += std::chrono::duration<double>(std::chrono::steady_clock::now() - th).count();
    }
    return BBFMM_OK;
}

int upload_residual(Schwarz &S, const double *rg) {
    parallel_for_chunks(S.n, int64_t(1) << 18, [&](int64_t b, int64_t e) {
        std::memcpy(S.h_pin + b, rg + b, static_cast<size_t>(e - b) * sizeof(double));
    });
    HIPOK(hipMemcpyAsync(S.d_rg, S.h_pin, static_cast<size_t>(S.n) * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPOK(hipMemsetAsync(S.d_sl, 0, static_cast<size_t>(S.n) * sizeof(double), S.stream));
    return BBFMM_OK;
}

// d_sl -> out[0 .. N), tail -> the last `basis` rows (zero when no tail was produced)
int download_correction(Schwarz &S, const std::vector<double> &tail, double *out) {
    // (the staging buffer is free again: the upload was consumed before the first kernel of the sweep ran)
    HIPOK(hipMemcpyAsync(S.h_pin, S.d_sl, static_cast<size_t>(S.n) * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIPOK(hipStreamSynchronize(S.stream));
    parallel_for_chunks(S.n, int64_t(1) << 18, [&](int64_t b, int64_t e) {
        std::memcpy(out + b, S.h_pin + b, static_cast<size_t>(e - b) * sizeof(double));
    });
    for (int b = 0; b < S.basis; ++b) out[S.n + b] = 0.0;
    const int64_t nt = S.n + S.basis;
    for (size_t a = 0; a < tail.size(); ++a) out[nt - static_cast<int64_t>(tail.size()) + static_cast<int64_t>(a)] = tail[a];
    return BBFMM_OK;
}

} // namespace

struct bbfmm_schwarz {
    Schwarz s;
};

namespace {
int schwarz_create_impl(bbfmm_handle *tree, const double *points, int64_t n, int32_t d, int64_t ld,
                        const bbfmm_interpolant *settings, const bbfmm_ddm_params *params, Schwarz &S) {
    // (S.rank / S.world / the exchange were set by the caller for a sharded preconditioner)
    S.tree = tree;
    S.n = n;
    S.d = d;
    S.degree = settings->polynomial_degree;
    S.nugget = settings->nugget;
    S.ks = make_kernel_spec(settings->kernel_type, settings->base_range, settings->total_sill);
    S.basis = monomial_basis_size(d, S.degree);
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
    S.stream = static_cast<hipStream_t>(bbfmm_stream(tree)); // (binds this thread to the tree's device)
    if (!S.stream) return BBFMM_DEVICE_ERROR;
    if (hipGetDevice(&S.device) != hipSuccess) return BBFMM_DEVICE_ERROR;
    // global monomial matrix on the cube-scaled points and its thin Q (rbf.rs:418-421, 476-495)
    double gscale[6] = {0, 0, 0, 1, 1, 1}; // translation, scale of the global monomial basis
    if (S.basis) {
        double *tr = gscale, *sc = gscale + 3;
        // Vectors of n entries on all host threads.  Sums run over fixed blocks of 65,536 entries whose partial
        // results are added in block order: the same bits whatever the thread count.
        constexpr int64_t kBlk = 65536;
        const int64_t nblk = (n + kBlk - 1) / kBlk;
        std::vector<double> part(static_cast<size_t>(nblk) * 2);
        auto dot = [&](const double *x, const double *y) {
            parallel_for(nblk, 1, [&](int64_t b) {
                double sum = 0.0;
                for (int64_t i = b * kBlk; i < std::min(n, (b + 1) * kBlk); ++i) sum += x[i] * y[i];
                part[static_cast<size_t>(b)] = sum;
            });
            double sum = 0.0;
            for (int64_t b = 0; b < nblk; ++b) sum += part[static_cast<size_t>(b)];
            return sum;
        };
        for (int a = 0; a < d; ++a) {
            const double *x = points + a * ld;
            parallel_for(nblk, 1, [&](int64_t b) {
                double lo = x[b * kBlk], hi = lo;
                for (int64_t i = b * kBlk; i < std::min(n, (b + 1) * kBlk); ++i) {
                    lo = std::min(lo, x[i]);
                    hi = std::max(hi, x[i]);
                }
                part[static_cast<size_t>(2 * b)] = lo;
                part[static_cast<size_t>(2 * b + 1)] = hi;
            });
            double lo = part[0], hi = part[1];
            for (int64_t b = 1; b < nblk; ++b) {
                lo = std::min(lo, part[static_cast<size_t>(2 * b)]);
                hi = std::max(hi, part[static_cast<size_t>(2 * b + 1)]);
            }
            tr[a] = (hi + lo) / 2.0;
            sc[a] = (hi - lo) / 2.0;
            if (sc[a] == 0.0) sc[a] = 1.0;
        }
        S.mono.resize(static_cast<size_t>(n) * S.basis);
        parallel_for_chunks(n, kBlk, [&](int64_t i0, int64_t i1) {
            for (int64_t i = i0; i < i1; ++i) {
                double sx[3] = {0, 0, 0};
                for (int a = 0; a < d; ++a) sx[a] = (points[a * ld + i] - tr[a]) / sc[a];
                monomial_row(sx, d, S.degree, &S.mono[static_cast<size_t>(i)], static_cast<size_t>(n));
            }
        });
        S.ortho.resize(S.mono.size()); // modified Gram-Schmidt, twice
        parallel_for_chunks(static_cast<int64_t>(S.mono.size()), kBlk, [&](int64_t i0, int64_t i1) {
            std::copy(S.mono.begin() + i0, S.mono.begin() + i1, S.ortho.begin() + i0);
        });
        for (int pass = 0; pass < 2; ++pass)
            for (int b = 0; b < S.basis; ++b) {
                double *qb = &S.ortho[static_cast<size_t>(b) * n];
                for (int c = 0; c < b; ++c) {
                    const double *qc = &S.ortho[static_cast<size_t>(c) * n];
                    const double sdot = dot(qc, qb);
                    parallel_for_chunks(n, kBlk, [&](int64_t i0, int64_t i1) {
                        for (int64_t i = i0; i < i1; ++i) qb[i] -= sdot * qc[i];
                    });
                }
                const double nn = std::sqrt(dot(qb, qb));
                if (nn == 0.0) return BBFMM_BAD_ARGUMENT;
                parallel_for_chunks(n, kBlk, [&](int64_t i0, int64_t i1) {
                    for (int64_t i = i0; i < i1; ++i) qb[i] /= nn;
                });
            }
    }
    S.levels.resize(S.ddm.levels.size());
    S.dom_total.assign(S.ddm.levels.size(), 0);
    S.dom_first.assign(S.ddm.levels.size(), 0);
    for (size_t li = 0; li < S.ddm.levels.size(); ++li) {
        const bool coarse = li + 1 == S.ddm.levels.size();
        {
            auto &leaves = S.ddm.levels[li].leaves;
            const int64_t nd = static_cast<int64_t>(leaves.size());
            S.dom_total[li] = nd;
            if (!coarse && S.world > 1) { // this rank's contiguous share of the level's domains; the others are dropped here
                const int64_t b = nd * S.rank / S.world, e = nd * (S.rank + 1) / S.world;
                S.dom_first[li] = b;
                std::vector<DdmDomain> own(std::make_move_iterator(leaves.begin() + b), std::make_move_iterator(leaves.begin() + e));
                leaves.swap(own);
                if (static_cast<int64_t>(S.ddm.levels[li].point_indices.size()) > S.xchg_cap) return BBFMM_BAD_ARGUMENT;
            }
        }
        // The coarse domain returns the polynomial tail of the correction.  The reference scales that
        // domain's monomials by the extents of its own points (domain.rs:171-172) although the system's
        // monomial matrix is scaled by the extents of all points (rbf.rs:418-421, 485-491), so the tail
        // belongs to a slightly different basis (relative extent mismatch of the coarse sample: the
        // coarse solve then leaves a residual of 5e-5 .. 4e-4 on its own points for a linear drift, 1e-12
        // with the global scaling).  Default: as the reference; BBFMM_FLAG_GLOBAL_SCALING: the global one.
        // (Measured: the FGMRES histories at 3M points are the same either way.)
        const bool global_scaling = (settings->flags & BBFMM_FLAG_GLOBAL_SCALING) != 0;
        if (S.ddm.levels[li].leaves.empty()) continue; // (a level with fewer domains than ranks: nothing to factorise here)
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
    // device vectors of the sweep, the polynomial basis, staging
    const size_t nb = static_cast<size_t>(n) * sizeof(double);
    for (double **pp : {&S.d_rg, &S.d_sl, &S.d_res, &S.d_out, &S.d_y}) HIPOK(hipMalloc(reinterpret_cast<void **>(pp), nb));
    HIPOK(hipMemsetAsync(S.d_out, 0, nb, S.stream));
    S.small_cap = static_cast<int64_t>(S.levels.back().gidx_h.size()) + 16;
    HIPOK(hipMalloc(reinterpret_cast<void **>(&S.d_small), static_cast<size_t>(S.small_cap) * sizeof(double)));
    HIPOK(hipHostMalloc(reinterpret_cast<void **>(&S.h_pin), static_cast<size_t>(n + S.small_cap) * sizeof(double), hipHostMallocDefault));
    if (S.basis) {
        HIPOK(hipMalloc(reinterpret_cast<void **>(&S.d_ortho), nb * static_cast<size_t>(S.basis)));
        HIPOK(hipMemcpy(S.d_ortho, S.ortho.data(), nb * static_cast<size_t>(S.basis), hipMemcpyHostToDevice));
        HIPOK(hipMalloc(reinterpret_cast<void **>(&S.d_part), static_cast<size_t>(kSchwarzProjectBlocks) * S.basis * sizeof(double)));
        HIPOK(hipMalloc(reinterpret_cast<void **>(&S.d_proj), 16 * sizeof(double)));
    }
    // the levels' rows: registered with the tree once (sorted targets + restricted downward pass belong to the
    // setup, not to the first apply) and resident on the device for the vector kernels
    S.d_lidx.assign(S.ddm.levels.size(), nullptr);
    S.subset_id.assign(S.ddm.levels.size(), -1);
    for (size_t li = 0; li < S.ddm.levels.size(); ++li) {
        const auto &pi = S.ddm.levels[li].point_indices;
        int32_t id = -1;
        rc = bbfmm_target_subset_create(tree, pi.data(), static_cast<int64_t>(pi.size()), &id);
        if (rc) return rc;
        S.subset_id[li] = id;
        if (id == -1) continue; // all rows in order
        std::vector<int32_t> idx32(pi.begin(), pi.end());
        HIPOK(hipMalloc(reinterpret_cast<void **>(&S.d_lidx[li]), std::max<size_t>(idx32.size(), 1) * sizeof(int32_t)));
        HIPOK(hipMemcpy(S.d_lidx[li], idx32.data(), idx32.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    S.t_level.assign(S.levels.size(), 0.0);
    S.t_level_mv.assign(S.levels.size(), 0.0);
    S.verbose = verbose;
    lap("target-subset plans", -1);
    return BBFMM_OK;
}
} // namespace

// Nothing may unwind through the C ABI (the N-sized host vectors can throw std::bad_alloc / length_error).
#define SCHWARZ_GUARD try {
#define SCHWARZ_BIND(h) \
    if ((h)->s.device >= 0) (void)hipSetDevice((h)->s.device);
#define SCHWARZ_END_GUARD                                   \
    }                                                       \
    catch (const std::bad_alloc &) { return BBFMM_DEVICE_ERROR; } \
    catch (...) { return BBFMM_BAD_ARGUMENT; }

extern "C" {

int bbfmm_schwarz_create_sharded(bbfmm_handle *tree, const double *points, int64_t n, int32_t d, int64_t ld,
                                 const bbfmm_interpolant *settings, const bbfmm_ddm_params *params, int32_t rank,
                                 int32_t world, double *d_exchange, int64_t exchange_capacity,
                                 bbfmm_allreduce_fn allreduce, void *allreduce_user, bbfmm_schwarz **out) {
    if (!out) return BBFMM_BAD_ARGUMENT;
    *out = nullptr;
    if (!tree || !points || !settings || n < 1 || d < 1 || d > 3 || ld < n) return BBFMM_BAD_ARGUMENT;
    if (settings->kernel_type < 0 || settings->kernel_type > 6 || settings->polynomial_degree < -1 ||
        settings->polynomial_degree > 2)
        return BBFMM_BAD_ARGUMENT;
    if (world < 1 || rank < 0 || rank >= world) return BBFMM_BAD_ARGUMENT;
    if (world > 1 && (!d_exchange || !allreduce || exchange_capacity < 1)) return BBFMM_BAD_ARGUMENT;
    SCHWARZ_GUARD
    std::unique_ptr<bbfmm_schwarz> h(new bbfmm_schwarz());
    h->s.rank = rank;
    h->s.world = world;
    h->s.d_xchg = d_exchange;
    h->s.xchg_cap = exchange_capacity;
    h->s.allreduce = allreduce;
    h->s.allreduce_user = allreduce_user;
    const int rc = schwarz_create_impl(tree, points, n, d, ld, settings, params, h->s);
    if (rc) return rc;
    *out = h.release();
    return BBFMM_OK;
    SCHWARZ_END_GUARD
}

int bbfmm_schwarz_create(bbfmm_handle *tree, const double *points, int64_t n, int32_t d, int64_t ld,
                         const bbfmm_interpolant *settings, const bbfmm_ddm_params *params, bbfmm_schwarz **out) {
    return bbfmm_schwarz_create_sharded(tree, points, n, d, ld, settings, params, 0, 1, nullptr, 0, nullptr, nullptr, out);
}

int64_t bbfmm_schwarz_factor_bytes(const bbfmm_schwarz *h) {
    if (!h) return -1;
    int64_t bytes = 0;
    for (const DdmLevelSolver &lv : h->s.levels)
        if (!lv.fac_off.empty()) bytes += lv.fac_off.back() * static_cast<int64_t>(sizeof(double));
    return bytes;
}
int64_t bbfmm_schwarz_domains_owned(const bbfmm_schwarz *h, int32_t level, int64_t *first, int64_t *total) {
    if (!h || level < 0 || level >= static_cast<int32_t>(h->s.levels.size())) return -1;
    if (first) *first = h->s.dom_first[static_cast<size_t>(level)];
    if (total) *total = h->s.dom_total[static_cast<size_t>(level)];
    return h->s.levels[static_cast<size_t>(level)].n_dom;
}

void bbfmm_schwarz_destroy(bbfmm_schwarz *h) { delete h; }

int64_t bbfmm_schwarz_basis_size(const bbfmm_schwarz *h) { return h ? h->s.basis : -1; }
int32_t bbfmm_schwarz_num_levels(const bbfmm_schwarz *h) { return h ? static_cast<int32_t>(h->s.ddm.levels.size()) : 0; }
int bbfmm_debug_evaluate_monomials(const double *points, int64_t n, int32_t d, int64_t ld, int32_t degree,
                                   const double *translation, const double *scale, double *out) {
    if (!points || !out || n < 0 || d < 1 || d > 3 || ld < n || degree < 0 || degree > 2) return BBFMM_BAD_ARGUMENT;
    for (int64_t i = 0; i < n; ++i) {
        double sx[3] = {0, 0, 0};
        for (int a = 0; a < d; ++a) sx[a] = (points[a * ld + i] - (translation ? translation[a] : 0.0)) / (scale ? scale[a] : 1.0);
        bbfmm::monomial_row(sx, d, degree, out + i, static_cast<size_t>(n));
    }
    return BBFMM_OK;
}

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
int bbfmm_schwarz_debug_level_solve(bbfmm_schwarz *h, int32_t level, const double *residual, double *out, int64_t n,
                                    int32_t add_poly) {
    if (!h || !residual || !out || level < 0 || level >= static_cast<int32_t>(h->s.ddm.levels.size()) ||
        n != h->s.n + h->s.basis)
        return BBFMM_BAD_ARGUMENT;
    SCHWARZ_GUARD
    SCHWARZ_BIND(h)
    Schwarz &S = h->s;
    const bool coarse = static_cast<size_t>(level) + 1 == S.ddm.levels.size();
    int rc = upload_residual(S, residual);
    if (rc) return rc;
    std::vector<double> tail;
    if ((rc = level_step(S, static_cast<size_t>(level), false, coarse, add_poly != 0, &tail))) return rc;
    return download_correction(S, tail, out);
    SCHWARZ_END_GUARD
}

// schwarz_preconditioner (schwarz.rs:32-82) as a bbfmm_apply_fn: user = bbfmm_schwarz*, n = N + basis
int bbfmm_schwarz_apply(void *user, const double *rg, double *sl, int64_t n) {
    bbfmm_schwarz *h = static_cast<bbfmm_schwarz *>(user);
    if (!h || !rg || !sl || n != h->s.n + h->s.basis) return BBFMM_BAD_ARGUMENT;
    SCHWARZ_GUARD
    SCHWARZ_BIND(h)
    Schwarz &S = h->s;
    const auto t_begin = std::chrono::steady_clock::now();
    int rc = upload_residual(S, rg);
    if (rc) return rc;
    const size_t coarse = S.ddm.levels.size() - 1;
    std::vector<double> tail;
    bool have_sl = false; // sl = 0 until the first correction has been added
    if (coarse > 0) {
        for (size_t i = 0; i < coarse; ++i) {
            if ((rc = level_step(S, i, have_sl, false, false, nullptr))) return rc;
            have_sl = true;
            if ((rc = level_step(S, coarse, true, true, i == coarse - 1, &tail))) return rc;
        }
    } else {
        if ((rc = level_step(S, coarse, false, true, true, &tail))) return rc;
    }
    if ((rc = download_correction(S, tail, sl))) return rc;
    if (S.verbose) {
        std::fprintf(stderr, "[bbfmm] schwarz apply %.3f s: partial matvecs %.3f s, level solves %.3f s (of which exchange %.3f s), host %.3f s\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), S.t_matvec, S.t_solve,
                     S.t_exchange, S.t_host);
        S.t_exchange = 0;
        for (size_t l = 0; l < S.t_level.size(); ++l) {
            std::fprintf(stderr, "[bbfmm]   level %zu: %lld domains, %lld entries, max m %d: solves %.3f s, matvecs %.3f s\n", l,
                         (long long)S.levels[l].n_dom, (long long)S.levels[l].n_entries, S.levels[l].max_m, S.t_level[l],
                         S.t_level_mv[l]);
            S.t_level[l] = S.t_level_mv[l] = 0;
        }
        S.t_matvec = S.t_solve = S.t_host = 0;
    }
    return BBFMM_OK;
    SCHWARZ_END_GUARD
}

} // extern "C"
