/*
 * oracle/passes.c -- CPU restatement of the per-matvec passes of ferreus_bbfmm.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Used by tests/, by
 * __graft_entry__.smoke() as the checker and by bench.py's cpu_baseline leg.
 * The shipped product never links or loads this file.
 *
 * PARITY UNPINNED by reference tests (no numeric golden vectors exist for this
 * path upstream); pinned against the dense direct sum (oracle_dense_sum below)
 * and the known-answer fixtures in tests/golden/.
 *
 * Each function cites the reference lines (relative to /root/reference) whose
 * arithmetic it follows.  Parallelism mirrors the reference's rayon loops: one
 * OpenMP task per leaf / per cell.
 *
 * Layouts (all f64, all "column-major" like faer::Mat):
 *   points   row-major  N x d   (pts[i*d + a])
 *   weights  w[k*ldw + i]              (N' x K col-major)
 *   M, L     coef[(k*C + c)*n + node]  (n x C*K col-major, column = c + k*C,
 *                                        ferreus_bbfmm/src/bbfmm.rs:724-725)
 *   out      out[k*ldo + i]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Per-thread scratch that survives the parallel regions (grow-only, never freed: test infrastructure).  The passes
 * used to malloc their work arrays at the top of every parallel region; the M2L ones are ~1 MB each, i.e. an mmap, a
 * page fault per 4 KB and a munmap per thread, region and level -- with 128-256 threads those serialise in the kernel
 * and the port got slower with every doubling of the thread count (2 x 64-core host of a GPU box). */
#define TLS_SLOTS 8
static __thread void *tls_ptr[TLS_SLOTS];
static __thread size_t tls_len[TLS_SLOTS];
static void *tls_buf(int slot, size_t bytes)
{
    if (tls_len[slot] < bytes) {
        free(tls_ptr[slot]);
        tls_ptr[slot] = malloc(bytes);
        tls_len[slot] = tls_ptr[slot] ? bytes : 0;
    }
    return tls_ptr[slot];
}

/* ------------------------------------------------------------------ kernels */
/* KernelType order: ferreus_rbf_utils/src/utils.rs:558-571.  100/101 are
 * extension kernels of this repo (NOT in the reference; BASELINE.json configs
 * name them): Gaussian exp(-(r/base_range)^2), Multiquadric sqrt(1+(r/base_range)^2). */
enum {
    K_LINEAR = 0, K_TPS = 1, K_CUBIC = 2, K_SPH3 = 3, K_SPH5 = 4, K_SPH7 = 5, K_SPH9 = 6,
    K_LAPLACIAN = 7, K_ONE_OVER_R2 = 8, K_ONE_OVER_R4 = 9,
    K_GAUSSIAN_EXT = 100, K_MULTIQUADRIC_EXT = 101
};

typedef struct {
    int id;
    double base_range, total_sill;
    /* spheroidal derived values, rbf_kernels.rs:232-243 */
    double s2, ip2, near_slope, far_coef;
    int sph_pow;
    double inv_br2;
} kspec_t;

/* ferreus_rbf_utils/src/constants.rs:21-50 */
static const double SPH_CONST[4][4] = {
    /* inflexion_point, linear_slope, range_scaling, inv_y_intercept */
    {0.5000000000, 0.7500000000, 2.6798340586, 0.8734640537},
    {0.4082482905, 1.0206207262, 1.5822795750, 0.8575980168},
    {0.3535533906, 1.2374368671, 1.2008676644, 0.8494862533},
    {0.3162277660, 1.4230249471, 1.0000000000, 0.8445585690},
};

static kspec_t make_kspec(int id, double base_range, double total_sill)
{
    kspec_t k;
    memset(&k, 0, sizeof k);
    k.id = id;
    k.base_range = base_range;
    k.total_sill = total_sill;
    if (id >= K_SPH3 && id <= K_SPH9) {
        const double *c = SPH_CONST[id - K_SPH3];
        double s = c[2] / base_range;          /* rbf_kernels.rs:234 */
        k.s2 = s * s;
        k.ip2 = c[0] * c[0];
        k.near_slope = total_sill * c[1] * s;
        k.far_coef = total_sill * c[3];
        k.sph_pow = id - K_SPH3 + 1;           /* POW 1..4, rbf_kernels.rs:178-205 */
    }
    k.inv_br2 = 1.0 / (base_range * base_range);
    return k;
}

static inline double powi_(double x, int n)
{
    double r = 1.0;
    for (int i = 0; i < n; ++i) r *= x;
    return r;
}

/* value of the kernel given r^2 = distance_sq (utils.rs:230-237). */
static inline double kval_r2(const kspec_t *k, double r2)
{
    switch (k->id) {
    case K_LINEAR: /* rbf_kernels.rs:25-36 */
        return -sqrt(r2);
    case K_TPS: { /* rbf_kernels.rs:69-84 */
        double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : (r * r) * log(r);
    }
    case K_CUBIC: { /* rbf_kernels.rs:118-130 */
        double r = sqrt(r2);
        return r * r * r;
    }
    case K_SPH3: case K_SPH5: case K_SPH7: case K_SPH9: { /* rbf_kernels.rs:245-256 */
        double sr2 = k->s2 * r2;
        if (sr2 <= k->ip2) {
            return k->total_sill - k->near_slope * sqrt(r2);
        } else {
            double t = 1.0 + sr2;
            return k->far_coef / (powi_(t, k->sph_pow) * sqrt(t));
        }
    }
    case K_LAPLACIAN: { /* non_rbf_kernels.rs:20-37 */
        double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : 1.0 / r;
    }
    case K_ONE_OVER_R2: { /* non_rbf_kernels.rs:70-86 */
        double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : 1.0 / (r * r);
    }
    case K_ONE_OVER_R4: { /* non_rbf_kernels.rs:121-137 */
        double r = sqrt(r2);
        return (fabs(r) < DBL_EPSILON) ? 0.0 : 1.0 / ((r * r) * (r * r));
    }
    case K_GAUSSIAN_EXT:
        return exp(-r2 * k->inv_br2);
    case K_MULTIQUADRIC_EXT:
        return sqrt(1.0 + r2 * k->inv_br2);
    default:
        return NAN;
    }
}

static inline double dist2(const double *t, const double *s, int d)
{
    double r2 = 0.0;
    for (int a = 0; a < d; ++a) {
        double df = t[a] - s[a];
        r2 += df * df;
    }
    return r2;
}

/* value + gradient wrt the target; evaluate_value_gradient of each kernel
 * (rbf_kernels.rs:38-57,86-106,132-152,266-300; non_rbf_kernels.rs:39-58,
 * 88-109,139-157).  g receives d entries. */
static inline double kval_grad(const kspec_t *k, const double *t, const double *s, int d, double *g)
{
    double r2 = 0.0;
    for (int a = 0; a < d; ++a) {
        double df = t[a] - s[a];
        g[a] = df;
        r2 += df * df;
    }
    double factor = 0.0, value = 0.0;
    int zero = (r2 <= DBL_EPSILON);
    switch (k->id) {
    case K_LINEAR:
        if (zero) { value = -sqrt(r2); break; }
        { double r = sqrt(r2); factor = -1.0 / r; value = -r; }
        break;
    case K_TPS:
        if (zero) { value = 0.0; break; }
        { double r = sqrt(r2); factor = 2.0 * log(r) + 1.0; value = r2 * log(r); }
        break;
    case K_CUBIC:
        if (zero) { value = 0.0; break; }
        { double r = sqrt(r2); factor = 3.0 * r; value = r2 * r; }
        break;
    case K_SPH3: case K_SPH5: case K_SPH7: case K_SPH9:
        value = kval_r2(k, r2);
        if (zero) break;
        {
            double sr2 = k->s2 * r2;
            if (sr2 <= k->ip2) {
                factor = -k->near_slope * (1.0 / sqrt(r2));
            } else {
                double t1 = 1.0 + sr2;
                double p = (double)k->sph_pow + 0.5;
                factor = -2.0 * p * k->s2 * k->far_coef / pow(t1, p + 1.0);
            }
        }
        break;
    case K_LAPLACIAN:
        if (zero) { value = 0.0; break; }
        { double ir = 1.0 / sqrt(r2); factor = -(ir * ir * ir); value = ir; }
        break;
    case K_ONE_OVER_R2:
        if (zero) { value = 0.0; break; }
        { factor = -2.0 * (1.0 / (r2 * r2)); value = 1.0 / r2; }
        break;
    case K_ONE_OVER_R4:
        if (zero) { value = 0.0; break; }
        { factor = -4.0 * (1.0 / (r2 * r2 * r2)); value = 1.0 / (r2 * r2); }
        break;
    case K_GAUSSIAN_EXT:
        value = exp(-r2 * k->inv_br2);
        factor = -2.0 * k->inv_br2 * value;
        zero = 0;
        break;
    case K_MULTIQUADRIC_EXT:
        value = sqrt(1.0 + r2 * k->inv_br2);
        factor = k->inv_br2 / value;
        zero = 0;
        break;
    default:
        value = NAN;
    }
    if (zero) {
        for (int a = 0; a < d; ++a) g[a] = 0.0;
    } else {
        for (int a = 0; a < d; ++a) g[a] *= factor;
    }
    return value;
}

/* 1 if the kernel provides gradients (all ten reference kernels do; the
 * KernelFunction default returns None, traits.rs:26-33). */
int oracle_kernel_has_gradient(int id)
{
    return (id >= 0 && id <= 9) || id == K_GAUSSIAN_EXT || id == K_MULTIQUADRIC_EXT;
}

double oracle_kernel_phi_r2(int id, double base_range, double total_sill, double r2)
{
    kspec_t k = make_kspec(id, base_range, total_sill);
    return kval_r2(&k, r2);
}

/* Dense kernel block a[i + j*m] = K(target_i, source_j)   (col-major m x n),
 * ferreus_bbfmm/src/utils.rs:64-88 (get_a_matrix) / 91-119 (subset). */
void oracle_kernel_block(int id, double base_range, double total_sill, int d,
                         int64_t m, const double *tgt, int64_t n, const double *src, double *a)
{
    kspec_t k = make_kspec(id, base_range, total_sill);
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < m; ++i)
            a[i + j * m] = kval_r2(&k, dist2(tgt + i * d, src + j * d, d));
}

/* Ground truth: out[k*ldo + i] = sum_j K(t_i, s_j) w[k*ldw + j]; optional
 * gradients grad[(k*d + a)*ldg + i] (layout [rhs0_dx, rhs0_dy, rhs0_dz, rhs1_dx..],
 * bbfmm.rs:431-433).  Dense builder in the reference:
 * ferreus_rbf_utils/src/utils.rs:288-312. */
/* Kernel matrix A[i + j*ldo] = phi(t_i, s_j) (get_a_matrix, ferreus_rbf_utils/src/utils.rs:288-312) */
void oracle_kernel_matrix(int id, double base_range, double total_sill, int d, int64_t nt, const double *tgt,
                          int64_t ns, const double *src, double *out, int64_t ldo)
{
    kspec_t k = make_kspec(id, base_range, total_sill);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ns; ++j)
        for (int64_t i = 0; i < nt; ++i) out[j * ldo + i] = kval_r2(&k, dist2(tgt + i * d, src + j * d, d));
}

/* at most 16 right-hand sides */
void oracle_dense_sum(int id, double base_range, double total_sill, int d,
                      int64_t nt, const double *tgt, int64_t ns, const double *src,
                      int K, const double *w, int64_t ldw, double *out, int64_t ldo,
                      double *grad, int64_t ldg)
{
    kspec_t k = make_kspec(id, base_range, total_sill);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nt; ++i) {
        double acc[16], gacc[16 * 3];
        for (int r = 0; r < K; ++r) acc[r] = 0.0;
        for (int r = 0; r < K * d; ++r) gacc[r] = 0.0;
        for (int64_t j = 0; j < ns; ++j) {
            if (grad) {
                double g[3];
                double v = kval_grad(&k, tgt + i * d, src + j * d, d, g);
                for (int r = 0; r < K; ++r) {
                    double wj = w[r * ldw + j];
                    acc[r] += v * wj;
                    for (int a = 0; a < d; ++a) gacc[r * d + a] += g[a] * wj;
                }
            } else {
                double v = kval_r2(&k, dist2(tgt + i * d, src + j * d, d));
                for (int r = 0; r < K; ++r) acc[r] += v * w[r * ldw + j];
            }
        }
        for (int r = 0; r < K; ++r) out[r * ldo + i] = acc[r];
        if (grad)
            for (int r = 0; r < K * d; ++r) grad[r * ldg + i] = gacc[r];
    }
}

/* ------------------------------------------------------- Chebyshev helpers */
/* T_k(x), k=0..p-1 and optionally T'_k(x): chebyshev.rs:47-110. */
static inline void cheb_T(int p, double x, double *T, double *dT)
{
    T[0] = 1.0;
    if (dT) dT[0] = 0.0;
    if (p > 1) {
        T[1] = x;
        if (dT) dT[1] = 1.0;
    }
    for (int j = 2; j < p; ++j) {
        T[j] = 2.0 * x * T[j - 1] - T[j - 2];
        if (dT) dT[j] = 2.0 * T[j - 1] + 2.0 * x * dT[j - 1] - dT[j - 2];
    }
}

/* S_j(x) = (2 * sum_k T_k(x) T_k(node_j) - 1) / p   (calculate_sn, chebyshev.rs:114-127)
 * dS_j(x) = (2/p) * sum_k T'_k(x) T_k(node_j)       (calculate_dsn_dx, 130-142)
 * polyn[j*p + k] = T_k(node_j). */
static inline void cheb_S(int p, double x, const double *polyn, double *S, double *dS)
{
    double T[32], dT[32];
    cheb_T(p, x, T, dS ? dT : NULL);
    for (int j = 0; j < p; ++j) {
        double s = 0.0, ds = 0.0;
        for (int k = 0; k < p; ++k) {
            s += T[k] * polyn[j * p + k];
            if (dS) ds += dT[k] * polyn[j * p + k];
        }
        S[j] = (s * 2.0 - 1.0) / (double)p;
        if (dS) dS[j] = ds * (2.0 / (double)p);
    }
}

static inline int ipow(int b, int e)
{
    int r = 1;
    for (int i = 0; i < e; ++i) r *= b;
    return r;
}

/* Row of the tensor-product anterpolation matrix for one point:
 * get_approximation_coefficients, chebyshev.rs:831-927.
 * vals[n]; grads[d*n] (g*n + col), already scaled by 2/length (862-869). */
static void cheb_row(int p, int d, const double *pt, const double *center, double length,
                     const double *polyn, double *vals, double *grads)
{
    double S[3][32], dS[3][32];
    for (int a = 0; a < d; ++a) {
        double x = (pt[a] - center[a]) / (length * 0.5); /* 841-845 */
        cheb_S(p, x, polyn, S[a], grads ? dS[a] : NULL);
        if (grads)
            for (int j = 0; j < p; ++j) dS[a][j] *= 2.0 / length;
    }
    int n = ipow(p, d);
    for (int col = 0; col < n; ++col) {
        int mi[3] = {0, 0, 0}, rem = col;
        for (int a = d - 1; a >= 0; --a) { /* 896-900: most significant axis first */
            mi[a] = rem % p;
            rem /= p;
        }
        double v = 1.0;
        for (int a = 0; a < d; ++a) v *= S[a][mi[a]];
        vals[col] = v;
        if (grads) {
            for (int g = 0; g < d; ++g) {
                double gv = dS[g][mi[g]];
                for (int a = 0; a < d; ++a)
                    if (a != g) gv *= S[a][mi[a]];
                grads[g * n + col] = gv;
            }
        }
    }
}

/* ------------------------------------------------------------------ P2M */
/* particle_to_multipole, bbfmm.rs:691-739, for every leaf flagged active
 * (upward_pass, bbfmm.rs:669-673). */
void oracle_p2m(int p, int d, int64_t C, int K,
                const int64_t *leaf_cells, int64_t nleaf,
                const double *centers, const double *lengths,
                const int64_t *src_ptr, const int64_t *src_idx,
                const double *pts, const double *w, int64_t ldw,
                const double *polyn, double *M)
{
    int n = ipow(p, d);
#pragma omp parallel
    {
        double *row = (double *)tls_buf(0, sizeof(double) * n);
#pragma omp for schedule(dynamic, 4)
        for (int64_t li = 0; li < nleaf; ++li) {
            int64_t c = leaf_cells[li];
            for (int64_t q = src_ptr[c]; q < src_ptr[c + 1]; ++q) {
                int64_t i = src_idx[q];
                cheb_row(p, d, pts + i * d, centers + c * d, lengths[c], polyn, row, NULL);
                for (int k = 0; k < K; ++k) {
                    double wi = w[k * ldw + i];
                    double *Mc = M + ((int64_t)k * C + c) * n;
                    for (int j = 0; j < n; ++j) Mc[j] += wi * row[j];
                }
            }
        }
    }
}

/* ------------------------------------------------------------ M2M / L2L */
/* multipole_to_multipole, bbfmm.rs:742-772: M_parent += T[ci] * M_child for the
 * parents listed (one level).  m2m[ci][row*n + col] (row = parent node). */
void oracle_m2m(int n, int64_t C, int K, const int64_t *parents, int64_t nparents,
                const int64_t *child_ptr, const int64_t *child_idx, const int32_t *child_octant,
                const double *m2m, double *M)
{
#pragma omp parallel for schedule(dynamic, 2)
    for (int64_t pi = 0; pi < nparents; ++pi) {
        int64_t P = parents[pi];
        for (int k = 0; k < K; ++k) {
            double *Mp = M + ((int64_t)k * C + P) * n;
            for (int64_t q = child_ptr[P]; q < child_ptr[P + 1]; ++q) {
                int64_t ch = child_idx[q];
                const double *T = m2m + (int64_t)child_octant[ch] * n * n;
                const double *Mc = M + ((int64_t)k * C + ch) * n;
                for (int r = 0; r < n; ++r) {
                    double s = 0.0;
                    const double *Tr = T + (int64_t)r * n;
                    for (int j = 0; j < n; ++j) s += Tr[j] * Mc[j];
                    Mp[r] += s;
                }
            }
        }
    }
}

/* local_to_local, bbfmm.rs:1051-1086: L_child += T[ci]^T * L_parent for
 * children flagged active (cells_with_targets). */
void oracle_l2l(int n, int64_t C, int K, const int64_t *parents, int64_t nparents,
                const int64_t *child_ptr, const int64_t *child_idx, const int32_t *child_octant,
                const uint8_t *active, const double *m2m, double *L)
{
#pragma omp parallel for schedule(dynamic, 2)
    for (int64_t pi = 0; pi < nparents; ++pi) {
        int64_t P = parents[pi];
        for (int64_t q = child_ptr[P]; q < child_ptr[P + 1]; ++q) {
            int64_t ch = child_idx[q];
            if (!active[ch]) continue;
            const double *T = m2m + (int64_t)child_octant[ch] * n * n;
            for (int k = 0; k < K; ++k) {
                const double *Lp = L + ((int64_t)k * C + P) * n;
                double *Lc = L + ((int64_t)k * C + ch) * n;
                /* (T^T Lp)[j] = sum_r T[r][j] Lp[r] */
                for (int r = 0; r < n; ++r) {
                    const double *Tr = T + (int64_t)r * n;
                    double lp = Lp[r];
                    for (int j = 0; j < n; ++j) Lc[j] += Tr[j] * lp;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ M2L */
/* multipole_to_local, bbfmm.rs:864-986, for the listed target cells of ONE level.
 * v_tidx = calculate_m2l_transfer_index (bbfmm.rs:989-998) per V pair.
 * Operators of the level: u_off[ref], vt_off[ref] offsets into opbuf, rank[ref].
 *   U  : n x r col-major;  Vt : r x n col-major.  compressed=0 -> U is n x n, no Vt.
 * perm / invperm: nperm x n int32 (chebyshev.rs:544-560); perm_lookup / ref_lookup:
 * 7^d entries (chebyshev.rs:574-577). */
void oracle_m2l(int n, int64_t C, int K, const int64_t *cells, int64_t ncells,
                const int64_t *v_ptr, const int64_t *v_idx, const int32_t *v_tidx,
                int nref, const int64_t *u_off, const int64_t *vt_off, const int32_t *rank,
                const double *opbuf, int compressed,
                const int32_t *perm, const int32_t *invperm,
                const int32_t *perm_lookup, const int32_t *ref_lookup,
                const double *M, double *L)
{
#pragma omp parallel
    {
        int maxv = 7 * 7 * 7;
        double *X = (double *)tls_buf(1, sizeof(double) * (size_t)n * maxv);  /* permuted multipoles, n x kk */
        double *Cm = (double *)tls_buf(2, sizeof(double) * (size_t)n * maxv); /* r x kk */
        double *Y = (double *)tls_buf(3, sizeof(double) * (size_t)n * maxv);  /* n x kk */
        int64_t *grp = (int64_t *)tls_buf(4, sizeof(int64_t) * maxv);
#pragma omp for schedule(dynamic, 1)
        for (int64_t ci = 0; ci < ncells; ++ci) {
            int64_t B = cells[ci];
            int64_t v0 = v_ptr[B], v1 = v_ptr[B + 1];
            if (v1 == v0) continue;
            for (int ref = 0; ref < nref; ++ref) {
                int kk = 0;
                for (int64_t q = v0; q < v1; ++q)
                    if (ref_lookup[v_tidx[q]] == ref) grp[kk++] = q;
                if (!kk) continue;
                int r = rank[ref];
                const double *U = opbuf + u_off[ref];
                const double *Vt = compressed ? opbuf + vt_off[ref] : NULL;
                for (int k = 0; k < K; ++k) {
                    /* gather + permute, bbfmm.rs:910-931 */
                    for (int c = 0; c < kk; ++c) {
                        int64_t q = grp[c];
                        const int32_t *pi = perm + (int64_t)perm_lookup[v_tidx[q]] * n;
                        const double *Mv = M + ((int64_t)k * C + v_idx[q]) * n;
                        double *Xc = X + (int64_t)c * n;
                        for (int j = 0; j < n; ++j) Xc[j] = Mv[pi[j]];
                    }
                    const double *rhs = X;
                    int inner = n;
                    if (compressed) { /* Cm = Vt * X, bbfmm.rs:953-954 */
                        for (int c = 0; c < kk; ++c) {
                            double *Cc = Cm + (int64_t)c * r;
                            for (int a = 0; a < r; ++a) Cc[a] = 0.0;
                            const double *Xc = X + (int64_t)c * n;
                            for (int j = 0; j < n; ++j) {
                                double xj = Xc[j];
                                const double *Vj = Vt + (int64_t)j * r;
                                for (int a = 0; a < r; ++a) Cc[a] += Vj[a] * xj;
                            }
                        }
                        rhs = Cm;
                        inner = r;
                    }
                    /* Y = U * rhs, bbfmm.rs:955-960 */
                    for (int c = 0; c < kk; ++c) {
                        double *Yc = Y + (int64_t)c * n;
                        for (int i = 0; i < n; ++i) Yc[i] = 0.0;
                        const double *Rc = rhs + (int64_t)c * inner;
                        for (int a = 0; a < inner; ++a) {
                            double ra = Rc[a];
                            const double *Ua = U + (int64_t)a * n;
                            for (int i = 0; i < n; ++i) Yc[i] += Ua[i] * ra;
                        }
                    }
                    /* inverse permute + accumulate, bbfmm.rs:964-982 */
                    double *Lb = L + ((int64_t)k * C + B) * n;
                    for (int c = 0; c < kk; ++c) {
                        const int32_t *ip = invperm + (int64_t)perm_lookup[v_tidx[grp[c]]] * n;
                        const double *Yc = Y + (int64_t)c * n;
                        for (int i = 0; i < n; ++i) Lb[i] += Yc[ip[i]];
                    }
                }
            }
        }
    }
}

/* The same pass with the reference's SHAPE of arithmetic: per (target cell, reference vector) the gathered k x n block
 * goes through two small GEMMs (bbfmm.rs:953-960 calls faer's, i.e. register-blocked FMA micro-kernels) instead of the
 * per-column multiply-add loops above.  Used by bench.py's cpu_baseline ("port" = what a CPU does with this algorithm);
 * oracle_m2l stays the checker of this one (tests/test_oracle_vs_dense.py: 1e-13) and the pass the parity tests run. */
typedef double v4d __attribute__((vector_size(32)));
typedef double v4du __attribute__((vector_size(32), aligned(8)));

/* Cc[m x nc] (ldc) = A[m x k] (lda) * B[k x nc] (ldb), all column-major; 8 x 4 register tiles, scalar edges */
__attribute__((optimize("fp-contract=fast")))
static void gemm_nn(int m, int nc, int k, const double *A, int lda, const double *B, int ldb, double *Cc, int ldc)
{
    int i = 0;
    for (; i + 8 <= m; i += 8) {
        int c = 0;
        for (; c + 4 <= nc; c += 4) {
            v4d a00 = {0, 0, 0, 0}, a01 = a00, a02 = a00, a03 = a00, a10 = a00, a11 = a00, a12 = a00, a13 = a00;
            const double *b0 = B + (int64_t)c * ldb, *b1 = b0 + ldb, *b2 = b1 + ldb, *b3 = b2 + ldb;
            const double *Ai = A + i;
            for (int q = 0; q < k; ++q) {
                v4d x0 = *(const v4du *)(Ai + (int64_t)q * lda), x1 = *(const v4du *)(Ai + (int64_t)q * lda + 4);
                v4d y0 = {b0[q], b0[q], b0[q], b0[q]}, y1 = {b1[q], b1[q], b1[q], b1[q]};
                v4d y2 = {b2[q], b2[q], b2[q], b2[q]}, y3 = {b3[q], b3[q], b3[q], b3[q]};
                a00 += x0 * y0; a10 += x1 * y0;
                a01 += x0 * y1; a11 += x1 * y1;
                a02 += x0 * y2; a12 += x1 * y2;
                a03 += x0 * y3; a13 += x1 * y3;
            }
            double *o = Cc + (int64_t)c * ldc + i;
            *(v4du *)(o) = a00; *(v4du *)(o + 4) = a10;
            *(v4du *)(o + ldc) = a01; *(v4du *)(o + ldc + 4) = a11;
            *(v4du *)(o + 2 * (int64_t)ldc) = a02; *(v4du *)(o + 2 * (int64_t)ldc + 4) = a12;
            *(v4du *)(o + 3 * (int64_t)ldc) = a03; *(v4du *)(o + 3 * (int64_t)ldc + 4) = a13;
        }
        for (; c < nc; ++c) { /* column edge: 8 x 1 */
            v4d a0 = {0, 0, 0, 0}, a1 = a0;
            const double *b = B + (int64_t)c * ldb, *Ai = A + i;
            for (int q = 0; q < k; ++q) {
                v4d y = {b[q], b[q], b[q], b[q]};
                a0 += *(const v4du *)(Ai + (int64_t)q * lda) * y;
                a1 += *(const v4du *)(Ai + (int64_t)q * lda + 4) * y;
            }
            *(v4du *)(Cc + (int64_t)c * ldc + i) = a0;
            *(v4du *)(Cc + (int64_t)c * ldc + i + 4) = a1;
        }
    }
    for (; i < m; ++i) /* row edge */
        for (int c = 0; c < nc; ++c) {
            double acc = 0.0;
            const double *b = B + (int64_t)c * ldb;
            for (int q = 0; q < k; ++q) acc += A[(int64_t)q * lda + i] * b[q];
            Cc[(int64_t)c * ldc + i] = acc;
        }
}

void oracle_m2l_gemm(int n, int64_t C, int K, const int64_t *cells, int64_t ncells,
                     const int64_t *v_ptr, const int64_t *v_idx, const int32_t *v_tidx,
                     int nref, const int64_t *u_off, const int64_t *vt_off, const int32_t *rank,
                     const double *opbuf, int compressed,
                     const int32_t *perm, const int32_t *invperm,
                     const int32_t *perm_lookup, const int32_t *ref_lookup,
                     const double *M, double *L)
{
    /* the level's operators once, rows padded to whole register tiles (zero rows: the edge loops of gemm_nn never run) */
    const int np8 = (n + 7) & ~7;
    double **Up = (double **)calloc((size_t)nref, sizeof(double *)), **Vp = (double **)calloc((size_t)nref, sizeof(double *));
    int *rp8 = (int *)calloc((size_t)nref, sizeof(int));
    for (int ref = 0; ref < nref; ++ref) {
        int r = compressed ? rank[ref] : n;
        rp8[ref] = (r + 7) & ~7;
        Up[ref] = (double *)calloc((size_t)np8 * r, sizeof(double));
        for (int a = 0; a < r; ++a) memcpy(Up[ref] + (size_t)a * np8, opbuf + u_off[ref] + (size_t)a * n, sizeof(double) * n);
        if (compressed) {
            Vp[ref] = (double *)calloc((size_t)rp8[ref] * n, sizeof(double));
            for (int j = 0; j < n; ++j) memcpy(Vp[ref] + (size_t)j * rp8[ref], opbuf + vt_off[ref] + (size_t)j * r, sizeof(double) * r);
        }
    }
#pragma omp parallel
    {
        int maxv = 7 * 7 * 7;
        double *X = (double *)tls_buf(1, sizeof(double) * (size_t)np8 * maxv);
        double *Cm = (double *)tls_buf(2, sizeof(double) * (size_t)np8 * maxv);
        double *Y = (double *)tls_buf(3, sizeof(double) * (size_t)np8 * maxv);
        int64_t *grp = (int64_t *)tls_buf(4, sizeof(int64_t) * maxv);
        /* cells of a level come in (level, key) order: neighbours in the list share most of their V-list sources, so a
         * thread keeps a contiguous block (static) -- no queue shared by all threads, no accumulator shared by any two */
#pragma omp for schedule(static)
        for (int64_t ci = 0; ci < ncells; ++ci) {
            int64_t B = cells[ci];
            int64_t v0 = v_ptr[B], v1 = v_ptr[B + 1];
            if (v1 == v0) continue;
            for (int ref = 0; ref < nref; ++ref) {
                int kk = 0;
                for (int64_t q = v0; q < v1; ++q)
                    if (ref_lookup[v_tidx[q]] == ref) grp[kk++] = q;
                if (!kk) continue;
                int r = rank[ref], rp = rp8[ref];
                for (int k = 0; k < K; ++k) {
                    for (int c = 0; c < kk; ++c) { /* gather + permute, bbfmm.rs:910-931 */
                        int64_t q = grp[c];
                        const int32_t *pi = perm + (int64_t)perm_lookup[v_tidx[q]] * n;
                        const double *Mv = M + ((int64_t)k * C + v_idx[q]) * n;
                        double *Xc = X + (int64_t)c * n;
                        for (int j = 0; j < n; ++j) Xc[j] = Mv[pi[j]];
                    }
                    if (compressed) {
                        gemm_nn(rp, kk, n, Vp[ref], rp, X, n, Cm, rp);      /* Cm = Vt * X, bbfmm.rs:953-954 */
                        gemm_nn(np8, kk, r, Up[ref], np8, Cm, rp, Y, np8);  /* Y = U * Cm, bbfmm.rs:955-960 */
                    } else {
                        gemm_nn(np8, kk, n, Up[ref], np8, X, n, Y, np8);
                    }
                    double *Lb = L + ((int64_t)k * C + B) * n; /* inverse permute + accumulate, bbfmm.rs:964-982 */
                    for (int c = 0; c < kk; ++c) {
                        const int32_t *ip = invperm + (int64_t)perm_lookup[v_tidx[grp[c]]] * n;
                        const double *Yc = Y + (int64_t)c * np8;
                        for (int i = 0; i < n; ++i) Lb[i] += Yc[ip[i]];
                    }
                }
            }
        }
    }
    for (int ref = 0; ref < nref; ++ref) { free(Up[ref]); free(Vp[ref]); }
    free(Up); free(Vp); free(rp8);
}

/* ------------------------------------------------------------------ P2L */
/* particle_to_local, bbfmm.rs:1001-1048 with nodes scaled by
 * scale_cheb_nodes_to_cell (chebyshev.rs:951-968). */
void oracle_p2l(int id, double base_range, double total_sill, int n, int d, int64_t C, int K,
                const int64_t *cells, int64_t ncells,
                const double *centers, const double *lengths, const double *nodes_nd,
                const int64_t *x_ptr, const int64_t *x_idx,
                const int64_t *src_ptr, const int64_t *src_idx,
                const double *pts, const double *w, int64_t ldw, double *L)
{
    kspec_t ks = make_kspec(id, base_range, total_sill);
#pragma omp parallel
    {
        double *nodes = (double *)tls_buf(5, sizeof(double) * (size_t)n * d);
#pragma omp for schedule(dynamic, 1)
        for (int64_t ci = 0; ci < ncells; ++ci) {
            int64_t B = cells[ci];
            if (x_ptr[B + 1] == x_ptr[B]) continue;
            for (int j = 0; j < n; ++j)
                for (int a = 0; a < d; ++a)
                    nodes[j * d + a] = centers[B * d + a] + (lengths[B] * 0.5) * nodes_nd[j * d + a];
            for (int64_t q = x_ptr[B]; q < x_ptr[B + 1]; ++q) {
                int64_t X = x_idx[q];
                for (int64_t s = src_ptr[X]; s < src_ptr[X + 1]; ++s) {
                    int64_t si = src_idx[s];
                    for (int j = 0; j < n; ++j) {
                        double v = kval_r2(&ks, dist2(nodes + j * d, pts + si * d, d));
                        for (int k = 0; k < K; ++k)
                            L[((int64_t)k * C + B) * n + j] += v * w[k * ldw + si];
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------ leaf pass */
/* leaf_pass_mode, bbfmm.rs:1113-1159: for each leaf that holds targets: P2P over
 * the U-list (1162-1251), M2P over the W-list (1254-1355), L2P (1358-1440).
 * tgt_ptr/tgt_idx: per-cell CSR of target rows (get_points_to_leaves_map,
 * linear_tree.rs:522-534).  flags: bit0 P2P, bit1 M2P, bit2 L2P (for per-phase
 * checks and timing). */
void oracle_leaf_pass(int id, double base_range, double total_sill, int p, int d, int64_t C, int K,
                      const int64_t *leaves, int64_t nleaves,
                      const double *centers, const double *lengths, const double *nodes_nd,
                      const double *polyn,
                      const int64_t *u_ptr, const int64_t *u_idx,
                      const int64_t *w_ptr, const int64_t *w_idx,
                      const int64_t *src_ptr, const int64_t *src_idx,
                      const int64_t *tgt_ptr, const int64_t *tgt_idx,
                      const double *pts, const double *tpts,
                      const double *w, int64_t ldw,
                      const double *M, const double *L,
                      double *out, int64_t ldo, double *grad, int64_t ldg, int flags)
{
    kspec_t ks = make_kspec(id, base_range, total_sill);
    int n = ipow(p, d);
#pragma omp parallel
    {
        double *row = (double *)tls_buf(0, sizeof(double) * n);
        double *grow = (double *)tls_buf(6, sizeof(double) * (size_t)n * d);
        double *nodes = (double *)tls_buf(5, sizeof(double) * (size_t)n * d);
#pragma omp for schedule(dynamic, 1)
        for (int64_t li = 0; li < nleaves; ++li) {
            int64_t B = leaves[li];
            int64_t t0 = tgt_ptr[B], t1 = tgt_ptr[B + 1];
            if (t1 == t0) continue;
            /* ---- P2P */
            if ((flags & 1) && (flags & 8) && !grad && d == 3 && (id == K_LINEAR || id == K_CUBIC)) {
                /* cpu_baseline only: the reference's loop (bbfmm.rs:1162-1251) copies the points and weights of every
                 * U-list leaf next to each other first; on the copies the pair loop of a kernel without branches is a
                 * plain reduction the compiler vectorises -- as LLVM can for the monomorphised Rust loop.  Sums run over
                 * the same sources in the same order per target as the loop below, in SIMD partial sums. */
                int64_t ns = 0;
                for (int64_t q = u_ptr[B]; q < u_ptr[B + 1]; ++q) ns += src_ptr[u_idx[q] + 1] - src_ptr[u_idx[q]];
                double *gx = (double *)tls_buf(7, sizeof(double) * (size_t)ns * (3 + (size_t)K));
                double *gy = gx + ns, *gz = gy + ns, *gw = gz + ns;
                int64_t at = 0;
                for (int64_t q = u_ptr[B]; q < u_ptr[B + 1]; ++q) {
                    int64_t U = u_idx[q];
                    for (int64_t s = src_ptr[U]; s < src_ptr[U + 1]; ++s, ++at) {
                        int64_t si = src_idx[s];
                        gx[at] = pts[si * 3]; gy[at] = pts[si * 3 + 1]; gz[at] = pts[si * 3 + 2];
                        for (int k = 0; k < K; ++k) gw[(int64_t)k * ns + at] = w[k * ldw + si];
                    }
                }
                for (int64_t tq = t0; tq < t1; ++tq) {
                    int64_t ti = tgt_idx[tq];
                    double tx = tpts[ti * 3], ty = tpts[ti * 3 + 1], tz = tpts[ti * 3 + 2];
                    for (int k = 0; k < K; ++k) {
                        const double *wk = gw + (int64_t)k * ns;
                        double acc = 0.0;
                        if (id == K_LINEAR) {
#pragma omp simd reduction(+ : acc)
                            for (int64_t s = 0; s < ns; ++s) {
                                double dx = tx - gx[s], dy = ty - gy[s], dz = tz - gz[s];
                                acc += -sqrt(dx * dx + dy * dy + dz * dz) * wk[s];
                            }
                        } else {
#pragma omp simd reduction(+ : acc)
                            for (int64_t s = 0; s < ns; ++s) {
                                double dx = tx - gx[s], dy = ty - gy[s], dz = tz - gz[s];
                                double r2 = dx * dx + dy * dy + dz * dz;
                                acc += r2 * sqrt(r2) * wk[s];
                            }
                        }
                        out[k * ldo + ti] += acc;
                    }
                }
            } else if (flags & 1) {
                for (int64_t q = u_ptr[B]; q < u_ptr[B + 1]; ++q) {
                    int64_t U = u_idx[q];
                    for (int64_t tq = t0; tq < t1; ++tq) {
                        int64_t ti = tgt_idx[tq];
                        const double *tp = tpts + ti * d;
                        for (int64_t s = src_ptr[U]; s < src_ptr[U + 1]; ++s) {
                            int64_t si = src_idx[s];
                            if (grad) {
                                double g[3];
                                double v = kval_grad(&ks, tp, pts + si * d, d, g);
                                for (int k = 0; k < K; ++k) {
                                    double wk = w[k * ldw + si];
                                    out[k * ldo + ti] += v * wk;
                                    for (int a = 0; a < d; ++a)
                                        grad[((int64_t)k * d + a) * ldg + ti] += g[a] * wk;
                                }
                            } else {
                                double v = kval_r2(&ks, dist2(tp, pts + si * d, d));
                                for (int k = 0; k < K; ++k) out[k * ldo + ti] += v * w[k * ldw + si];
                            }
                        }
                    }
                }
            }
            /* ---- M2P */
            if ((flags & 2) && w_ptr) {
                for (int64_t q = w_ptr[B]; q < w_ptr[B + 1]; ++q) {
                    int64_t W = w_idx[q];
                    for (int j = 0; j < n; ++j)
                        for (int a = 0; a < d; ++a)
                            nodes[j * d + a] =
                                centers[W * d + a] + (lengths[W] * 0.5) * nodes_nd[j * d + a];
                    for (int64_t tq = t0; tq < t1; ++tq) {
                        int64_t ti = tgt_idx[tq];
                        const double *tp = tpts + ti * d;
                        for (int j = 0; j < n; ++j) {
                            if (grad) {
                                double g[3];
                                double v = kval_grad(&ks, tp, nodes + j * d, d, g);
                                for (int k = 0; k < K; ++k) {
                                    double cf = M[((int64_t)k * C + W) * n + j];
                                    out[k * ldo + ti] += v * cf;
                                    for (int a = 0; a < d; ++a)
                                        grad[((int64_t)k * d + a) * ldg + ti] += g[a] * cf;
                                }
                            } else {
                                double v = kval_r2(&ks, dist2(tp, nodes + j * d, d));
                                for (int k = 0; k < K; ++k)
                                    out[k * ldo + ti] += v * M[((int64_t)k * C + W) * n + j];
                            }
                        }
                    }
                }
            }
            /* ---- L2P */
            if (flags & 4) {
                for (int64_t tq = t0; tq < t1; ++tq) {
                    int64_t ti = tgt_idx[tq];
                    cheb_row(p, d, tpts + ti * d, centers + B * d, lengths[B], polyn, row,
                             grad ? grow : NULL);
                    for (int k = 0; k < K; ++k) {
                        const double *Lb = L + ((int64_t)k * C + B) * n;
                        double s = 0.0;
                        for (int j = 0; j < n; ++j) s += row[j] * Lb[j];
                        out[k * ldo + ti] += s;
                        if (grad) {
                            for (int a = 0; a < d; ++a) {
                                double gs = 0.0;
                                for (int j = 0; j < n; ++j) gs += grow[a * n + j] * Lb[j];
                                grad[((int64_t)k * d + a) * ldg + ti] += gs;
                            }
                        }
                    }
                }
            }
        }
    }
}

/* Zero fill by all threads (static blocks): the coefficient arrays are kept between matvecs in the cpu_baseline mode --
 * a fresh calloc per pass means a page fault per 4 KB from every thread at once, and those serialise in the kernel
 * (measured on the 2 x 64-core host of a GPU box: the passes got SLOWER with every doubling of the thread count). */
void oracle_zero(double *p, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (n + 65535) / 65536; ++b) {
        int64_t lo = b * 65536, hi = lo + 65536 < n ? lo + 65536 : n;
        memset(p + lo, 0, (size_t)(hi - lo) * sizeof(double));
    }
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Thread count of the following passes (bench.py's cpu_baseline tries a few and reports the fastest). */
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
