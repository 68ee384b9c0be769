// Host precomputation of Chebyshev / M2L operators.  See operators.hpp.
#include "operators.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "parallel.hpp"

namespace bbfmm {
namespace {

int ipow(int b, int e) {
    int r = 1;
    for (int i = 0; i < e; ++i) r *= b;
    return r;
}

// chebyshev.rs:47-110 (values only)
void cheb_T(int p, double x, double *T) {
    T[0] = 1.0;
    if (p > 1) T[1] = x;
    for (int j = 2; j < p; ++j) T[j] = 2.0 * x * T[j - 1] - T[j - 2];
}

// ferreus_bbfmm/src/utils.rs:123-134: row i, column j of the cartesian product of
// `base` values repeated ncols times (axis 0 slowest) -> index into values.
inline int cart_index(int i, int j, int base, int ncols) {
    return (i / ipow(base, ncols - j - 1)) % base;
}

// ferreus_bbfmm/src/utils.rs:138-146 (stable)
template <class T> std::vector<int> argsort_stable(const std::vector<T> &v) {
    std::vector<int> idx(v.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return v[a] < v[b]; });
    return idx;
}

// chebyshev.rs:300-315 (alpha is 1-based)
int map_multi_index_to_k(const int *alpha, int d, int p) {
    int m = 0;
    for (int a = 0; a < d; ++a) m = m * p + (alpha[a] - 1);
    return m;
}

// chebyshev.rs:486-585
void permutation_lookups(Operators &o) {
    const int d = o.d, p = o.p, n = o.n;
    // axis order permutations in lexicographic order (itertools::permutations)
    std::vector<std::vector<int>> order_perms;
    {
        std::vector<int> ax(d);
        std::iota(ax.begin(), ax.end(), 0);
        do order_perms.push_back(ax);
        while (std::next_permutation(ax.begin(), ax.end()));
    }
    const int n_sign = 1 << d;
    const int n_order = static_cast<int>(order_perms.size());
    auto sign_of = [&](int row, int axis) { return cart_index(row, axis, 2, d) == 0 ? -1 : 1; };

    std::vector<std::vector<int>> diag(n_order, std::vector<int>(n)), axial(n_sign, std::vector<int>(n));
    for (int j = 0; j < n; ++j) {
        int alpha[3];
        for (int a = 0; a < d; ++a) alpha[a] = cart_index(j, a, p, d) + 1; // multi_indices row j
        for (int b = 0; b < n_order; ++b) { // 339-342
            int ap[3];
            for (int a = 0; a < d; ++a) ap[a] = alpha[order_perms[b][a]];
            diag[b][map_multi_index_to_k(ap, d, p)] = j;
        }
        for (int s = 0; s < n_sign; ++s) { // 318-336
            int ap[3];
            for (int a = 0; a < d; ++a) ap[a] = sign_of(s, a) < 0 ? p - (alpha[a] - 1) : alpha[a];
            axial[s][map_multi_index_to_k(ap, d, p)] = j;
        }
    }
    o.n_perm = n_sign * n_order;
    o.perm.resize(static_cast<size_t>(o.n_perm) * n);
    o.invperm.resize(static_cast<size_t>(o.n_perm) * n);
    for (int a = 0; a < n_sign; ++a)
        for (int b = 0; b < n_order; ++b) { // 536-560
            const int c = a * n_order + b;
            std::vector<int> combo(n);
            for (int i = 0; i < n; ++i) combo[i] = axial[a][diag[b][i]];
            const std::vector<int> inv = argsort_stable(combo);
            for (int i = 0; i < n; ++i) {
                o.perm[static_cast<size_t>(c) * n + i] = combo[i];
                o.invperm[static_cast<size_t>(c) * n + i] = inv[i];
            }
        }
    // lookups (379-483, 562-577)
    o.perm_lookup.assign(o.n_vec, 0);
    o.ref_lookup.assign(o.n_vec, 0);
    std::vector<std::vector<int>> sorted_refs(o.n_ref);
    for (int r = 0; r < o.n_ref; ++r) {
        sorted_refs[r].assign(o.ref_vecs.begin() + r * d, o.ref_vecs.begin() + (r + 1) * d);
        std::sort(sorted_refs[r].begin(), sorted_refs[r].end());
    }
    for (int v = 0; v < o.n_vec; ++v) {
        const int32_t *vec = &o.all_vecs[static_cast<size_t>(v) * d];
        int axial_case = 0;
        for (int s = 0; s < n_sign; ++s) {
            bool ok = true;
            for (int a = 0; a < d; ++a) ok = ok && (sign_of(s, a) == (vec[a] < 0 ? -1 : 1));
            if (ok) { axial_case = s; break; }
        }
        std::vector<int> neg_abs(d);
        for (int a = 0; a < d; ++a) neg_abs[a] = -std::abs(vec[a]);
        const std::vector<int> sorted_axes = argsort_stable(neg_abs);
        int diag_case = 0;
        for (int b = 0; b < n_order; ++b)
            if (order_perms[b] == sorted_axes) { diag_case = b; break; }
        o.perm_lookup[v] = axial_case * n_order + diag_case;
        std::vector<int> sv(d);
        for (int a = 0; a < d; ++a) sv[a] = std::abs(vec[a]);
        std::sort(sv.begin(), sv.end());
        for (int r = 0; r < o.n_ref; ++r)
            if (sorted_refs[r] == sv) { o.ref_lookup[v] = r; break; }
    }
}

// aca.rs:146-161
int argmax_masked(const double *data, const uint8_t *mask, int n) {
    int mi = 0;
    double mv = 0.0;
    for (int i = 0; i < n; ++i) {
        const double w = std::fabs(data[i]) * static_cast<double>(mask[i]);
        if (w > mv) { mv = w; mi = i; }
    }
    return mi;
}

// aca.rs:23-136.  entry(i, j) evaluates A[i, j].  Returns k; u is rows x k, v is cols x k.
template <class F>
int aca_partial_pivoting(int rows, int cols, F &&entry, double epsilon, std::vector<double> *u_out,
                         std::vector<double> *v_out) {
    const int max_it = std::min(rows, cols);
    const double tol = epsilon * epsilon;
    std::vector<uint8_t> unused_rows(rows, 1), unused_cols(cols, 1);
    std::vector<double> u(static_cast<size_t>(rows) * max_it, 0.0), v(static_cast<size_t>(cols) * max_it, 0.0);
    std::vector<double> vrow(cols), ucol(rows), p1(max_it), p2(max_it);
    double residual_norm = 0.0, sum_k = 0.0;
    int i = 0, k = 0;
    for (int it = 0; it < max_it; ++it) {
        for (int j = 0; j < cols; ++j) vrow[j] = entry(i, j); // 61
        unused_rows[i] = 0;
        for (int q = 0; q < k; ++q) { // 67-70
            const double uiq = u[static_cast<size_t>(q) * rows + i];
            const double *vq = &v[static_cast<size_t>(q) * cols];
            for (int j = 0; j < cols; ++j) vrow[j] -= uiq * vq[j];
        }
        const int j = argmax_masked(vrow.data(), unused_cols.data(), cols);
        if (vrow[j] == 0.0 && k > 0) {
            // Exactly zero residual row: the cross approximation is already exact.  The
            // reference divides by zero here (aca.rs:76) and propagates NaN; documented
            // deviation (also taken by the oracle): stop with the current rank.
            break;
        }
        const double pivot = 1.0 / vrow[j];
        for (int c = 0; c < cols; ++c) vrow[c] *= pivot;
        for (int r = 0; r < rows; ++r) ucol[r] = entry(r, j); // 81
        unused_cols[j] = 0;
        for (int q = 0; q < k; ++q) { // 87-90
            const double vjq = v[static_cast<size_t>(q) * cols + j];
            const double *uq = &u[static_cast<size_t>(q) * rows];
            for (int r = 0; r < rows; ++r) ucol[r] -= vjq * uq[r];
        }
        i = argmax_masked(ucol.data(), unused_rows.data(), rows);
        if (k > 0) { // 96-112: sum_q <u_q,u_k><v_q,v_k>
            sum_k = 0.0;
            for (int q = 0; q < k; ++q) {
                double a = 0.0, b = 0.0;
                const double *uq = &u[static_cast<size_t>(q) * rows];
                const double *vq = &v[static_cast<size_t>(q) * cols];
                for (int r = 0; r < rows; ++r) a += uq[r] * ucol[r];
                for (int c = 0; c < cols; ++c) b += vq[c] * vrow[c];
                sum_k += a * b;
            }
        }
        double nu = 0.0, nv = 0.0;
        for (int r = 0; r < rows; ++r) nu += ucol[r] * ucol[r];
        for (int c = 0; c < cols; ++c) nv += vrow[c] * vrow[c];
        const double norm_u_v_2 = nu * nv; // 115-116
        residual_norm += norm_u_v_2 + 2.0 * sum_k;
        std::copy(ucol.begin(), ucol.end(), u.begin() + static_cast<size_t>(k) * rows);
        std::copy(vrow.begin(), vrow.end(), v.begin() + static_cast<size_t>(k) * cols);
        ++k;
        if (norm_u_v_2 <= tol * residual_norm) break; // 129
    }
    u.resize(static_cast<size_t>(rows) * k);
    v.resize(static_cast<size_t>(cols) * k);
    *u_out = std::move(u);
    *v_out = std::move(v);
    return k;
}

} // namespace

int singular_values_cutoff(const std::vector<double> &sigma, double epsilon) {
    const int n = static_cast<int>(sigma.size());
    std::vector<double> cum(n);
    double acc = 0.0;
    for (int i = n - 1; i >= 0; --i) { // aca.rs:234-247
        acc += sigma[i] * sigma[i];
        cum[i] = acc;
    }
    if (n == 0) return 0;
    const double eps_qr = cum[0] * epsilon * epsilon; // aca.rs:215
    for (int i = 0; i < n; ++i)
        if (cum[i] < eps_qr) return i;
    return n;
}

void thin_qr(const std::vector<double> &a, int m, int k, std::vector<double> *q_out,
             std::vector<double> *r_out) {
    std::vector<double> w = a; // m x k
    std::vector<double> vs(static_cast<size_t>(m) * k, 0.0), betas(k, 0.0);
    std::vector<double> &r = *r_out;
    r.assign(static_cast<size_t>(k) * k, 0.0);
    for (int j = 0; j < k; ++j) {
        double *col = &w[static_cast<size_t>(j) * m];
        double norm = 0.0;
        for (int i = j; i < m; ++i) norm += col[i] * col[i];
        norm = std::sqrt(norm);
        double *v = &vs[static_cast<size_t>(j) * m];
        if (norm == 0.0) {
            betas[j] = 0.0;
            continue;
        }
        const double alpha = col[j] >= 0 ? -norm : norm;
        for (int i = j; i < m; ++i) v[i] = col[i];
        v[j] -= alpha;
        double vn = 0.0;
        for (int i = j; i < m; ++i) vn += v[i] * v[i];
        betas[j] = vn == 0.0 ? 0.0 : 2.0 / vn;
        for (int c = j; c < k; ++c) { // apply H to the trailing columns
            double *cc = &w[static_cast<size_t>(c) * m];
            double dot = 0.0;
            for (int i = j; i < m; ++i) dot += v[i] * cc[i];
            dot *= betas[j];
            for (int i = j; i < m; ++i) cc[i] -= dot * v[i];
        }
    }
    for (int c = 0; c < k; ++c)
        for (int i = 0; i <= c; ++i) r[static_cast<size_t>(c) * k + i] = w[static_cast<size_t>(c) * m + i];
    // Q = H_0 ... H_{k-1} * [I_k; 0]
    std::vector<double> &q = *q_out;
    q.assign(static_cast<size_t>(m) * k, 0.0);
    for (int c = 0; c < k; ++c) q[static_cast<size_t>(c) * m + c] = 1.0;
    for (int j = k - 1; j >= 0; --j) {
        const double *v = &vs[static_cast<size_t>(j) * m];
        if (betas[j] == 0.0) continue;
        for (int c = 0; c < k; ++c) {
            double *qc = &q[static_cast<size_t>(c) * m];
            double dot = 0.0;
            for (int i = j; i < m; ++i) dot += v[i] * qc[i];
            dot *= betas[j];
            for (int i = j; i < m; ++i) qc[i] -= dot * v[i];
        }
    }
}

void jacobi_svd(const std::vector<double> &a, int m, int k, std::vector<double> *u_out,
                std::vector<double> *s_out, std::vector<double> *vt_out) {
    std::vector<double> w = a;                               // m x k
    std::vector<double> v(static_cast<size_t>(k) * k, 0.0);  // k x k
    for (int i = 0; i < k; ++i) v[static_cast<size_t>(i) * k + i] = 1.0;
    const double eps = 1e-15;
    for (int sweep = 0; sweep < 80; ++sweep) {
        bool rotated = false;
        for (int i = 0; i < k - 1; ++i) {
            double *wi = &w[static_cast<size_t>(i) * m];
            for (int j = i + 1; j < k; ++j) {
                double *wj = &w[static_cast<size_t>(j) * m];
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int r = 0; r < m; ++r) {
                    alpha += wi[r] * wi[r];
                    beta += wj[r] * wj[r];
                    gamma += wi[r] * wj[r];
                }
                if (gamma == 0.0 || std::fabs(gamma) <= eps * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < m; ++r) {
                    const double x = wi[r], y = wj[r];
                    wi[r] = c * x - s * y;
                    wj[r] = s * x + c * y;
                }
                double *vi = &v[static_cast<size_t>(i) * k], *vj = &v[static_cast<size_t>(j) * k];
                for (int r = 0; r < k; ++r) {
                    const double x = vi[r], y = vj[r];
                    vi[r] = c * x - s * y;
                    vj[r] = s * x + c * y;
                }
            }
        }
        if (!rotated) break;
    }
    std::vector<double> sig(k);
    for (int j = 0; j < k; ++j) {
        double nn = 0.0;
        const double *wj = &w[static_cast<size_t>(j) * m];
        for (int r = 0; r < m; ++r) nn += wj[r] * wj[r];
        sig[j] = std::sqrt(nn);
    }
    std::vector<int> order(k);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sig[x] > sig[y]; });
    u_out->assign(static_cast<size_t>(m) * k, 0.0);
    s_out->assign(k, 0.0);
    vt_out->assign(static_cast<size_t>(k) * k, 0.0);
    for (int jj = 0; jj < k; ++jj) {
        const int j = order[jj];
        (*s_out)[jj] = sig[j];
        const double inv = sig[j] > 0.0 ? 1.0 / sig[j] : 0.0;
        for (int r = 0; r < m; ++r) (*u_out)[static_cast<size_t>(jj) * m + r] = w[static_cast<size_t>(j) * m + r] * inv;
        for (int r = 0; r < k; ++r) (*vt_out)[static_cast<size_t>(r) * k + jj] = v[static_cast<size_t>(j) * k + r]; // vt[jj, r]
    }
}

void dense_m2m_matrix(const Operators &o, int ci, std::vector<double> *out) {
    const int n = o.n, p = o.p, d = o.d;
    out->assign(static_cast<size_t>(n) * n, 0.0);
    for (int pr = 0; pr < n; ++pr)     // parent node (row)
        for (int ch = 0; ch < n; ++ch) { // child node (col)
            double v = 1.0;
            for (int a = 0; a < d; ++a) {
                const int ia = cart_index(pr, a, p, d), ca = cart_index(ch, a, p, d);
                const int side = (ci >> a) & 1; // chebyshev.rs:183-192: bit a <-> axis a
                v *= o.xfer[static_cast<size_t>(side) * p * p + ca * p + ia];
            }
            (*out)[static_cast<size_t>(ch) * n + pr] = v; // column-major
        }
}

void precompute_operators(int p, int d, double radius, int depth, const KernelSpec &kernel,
                          int compression, double epsilon, Operators *out) {
    Operators &o = *out;
    o = Operators();
    o.p = p;
    o.d = d;
    const int n = o.n = ipow(p, d);
    o.compression = compression;

    o.nodes.resize(p); // chebyshev.rs:32-40
    for (int i = 0; i < p; ++i) {
        const int ii = p - 1 - i;
        o.nodes[i] = std::cos(M_PI * (static_cast<double>(ii) + 0.5) / static_cast<double>(p));
    }
    o.polyn.resize(static_cast<size_t>(p) * p);
    for (int j = 0; j < p; ++j) cheb_T(p, o.nodes[j], &o.polyn[static_cast<size_t>(j) * p]);
    o.nodes_nd.resize(static_cast<size_t>(n) * d);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < d; ++a) o.nodes_nd[static_cast<size_t>(i) * d + a] = o.nodes[cart_index(i, a, p, d)];

    // chebyshev.rs:146-180: S at the 2p child nodes
    o.xfer.resize(static_cast<size_t>(2) * p * p);
    for (int row = 0; row < 2 * p; ++row) {
        const double child = (row < p ? o.nodes[row] - 1.0 : o.nodes[row - p] + 1.0) * 0.5;
        double T[64];
        cheb_T(p, child, T);
        for (int j = 0; j < p; ++j) {
            double s = 0.0;
            for (int k = 0; k < p; ++k) s += T[k] * o.polyn[static_cast<size_t>(j) * p + k];
            o.xfer[static_cast<size_t>(row) * p + j] = (s * 2.0 - 1.0) / static_cast<double>(p);
        }
    }

    // chebyshev.rs:267-297
    o.n_vec = ipow(7, d);
    o.all_vecs.resize(static_cast<size_t>(o.n_vec) * d);
    for (int v = 0; v < o.n_vec; ++v)
        for (int a = 0; a < d; ++a) o.all_vecs[static_cast<size_t>(v) * d + a] = cart_index(v, a, 7, d) - 3;
    o.ref_vecs.clear();
    for (int b = 0; b < ipow(4, d); ++b) {
        int row[3];
        for (int a = 0; a < d; ++a) row[a] = cart_index(b, a, 4, d);
        bool valid = row[0] >= 2;
        for (int a = 1; a < d && valid; ++a) valid = row[a] <= row[a - 1];
        if (valid)
            for (int a = 0; a < d; ++a) o.ref_vecs.push_back(row[a]);
    }
    o.n_ref = static_cast<int>(o.ref_vecs.size()) / d;
    permutation_lookups(o);

    // chebyshev.rs:697-791: reference operators for levels 2..=depth
    o.m2l.assign(static_cast<size_t>(depth) + 1, {});
    for (int level = 2; level <= depth; ++level) o.m2l[level].resize(o.n_ref);
    const int n_tasks = depth >= 2 ? (depth - 1) * o.n_ref : 0;
    parallel_for(n_tasks, 1, [&](int64_t task) {
        const int level = 2 + static_cast<int>(task) / o.n_ref;
        const int ref = static_cast<int>(task) % o.n_ref;
        const double length = radius / static_cast<double>(uint64_t(1) << (level - 1)); // 702
        // 588-627: nodes of the target cell (origin) and of the cell at +ref
        std::vector<double> tp(static_cast<size_t>(n) * d), sp(static_cast<size_t>(n) * d);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < d; ++a) {
                const double node = o.nodes[cart_index(i, a, p, d)];
                tp[static_cast<size_t>(i) * d + a] = node * (0.5 * length);
                sp[static_cast<size_t>(i) * d + a] =
                    (static_cast<double>(o.ref_vecs[static_cast<size_t>(ref) * d + a]) + node * 0.5) * length;
            }
        // A[i, j] = K(sp_i, tp_j): rows <-> cell at +ref, cols <-> cell at the origin (728-746)
        auto entry = [&](int i, int j) {
            double r2 = 0.0;
            for (int a = 0; a < d; ++a) {
                const double df = sp[static_cast<size_t>(i) * d + a] - tp[static_cast<size_t>(j) * d + a];
                r2 += df * df;
            }
            return kernel_value_r2_rt(kernel, r2);
        };
        M2lOperator &op = o.m2l[level][ref];
        if (compression == kCompressionAca) {
            std::vector<double> u, v;
            const int k = aca_partial_pivoting(n, n, entry, epsilon, &u, &v);
            // recompress_aca, aca.rs:173-200
            std::vector<double> qu, ru, qv, rv;
            thin_qr(u, n, k, &qu, &ru);
            thin_qr(v, n, k, &qv, &rv);
            std::vector<double> core(static_cast<size_t>(k) * k, 0.0); // ru * rv^T
            for (int c = 0; c < k; ++c)
                for (int r = 0; r < k; ++r) {
                    double s = 0.0;
                    for (int q = 0; q < k; ++q) s += ru[static_cast<size_t>(q) * k + r] * rv[static_cast<size_t>(q) * k + c];
                    core[static_cast<size_t>(c) * k + r] = s;
                }
            std::vector<double> ur, sr, vrt;
            jacobi_svd(core, k, k, &ur, &sr, &vrt);
            const int rank = singular_values_cutoff(sr, epsilon);
            op.rank = rank;
            op.u.assign(static_cast<size_t>(n) * rank, 0.0);
            op.vt.assign(static_cast<size_t>(rank) * n, 0.0);
            for (int c = 0; c < rank; ++c)      // U = Qu * (Ur[:, :rank] * diag(s))
                for (int q = 0; q < k; ++q) {
                    const double f = ur[static_cast<size_t>(c) * k + q] * sr[c];
                    const double *quq = &qu[static_cast<size_t>(q) * n];
                    double *uc = &op.u[static_cast<size_t>(c) * n];
                    for (int r = 0; r < n; ++r) uc[r] += quq[r] * f;
                }
            for (int j = 0; j < n; ++j)         // Vt = Vrt[:rank] * Qv^T
                for (int a = 0; a < rank; ++a) {
                    double s = 0.0;
                    for (int q = 0; q < k; ++q) s += vrt[static_cast<size_t>(q) * k + a] * qv[static_cast<size_t>(q) * n + j];
                    op.vt[static_cast<size_t>(j) * rank + a] = s;
                }
        } else {
            std::vector<double> a(static_cast<size_t>(n) * n);
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) a[static_cast<size_t>(j) * n + i] = entry(i, j);
            if (compression == kCompressionSvd) { // 760-779
                std::vector<double> ur, sr, vrt;
                jacobi_svd(a, n, n, &ur, &sr, &vrt);
                const int rank = singular_values_cutoff(sr, epsilon);
                op.rank = rank;
                op.u.assign(ur.begin(), ur.begin() + static_cast<size_t>(n) * rank);
                op.vt.assign(static_cast<size_t>(rank) * n, 0.0);
                for (int j = 0; j < n; ++j)
                    for (int r = 0; r < rank; ++r)
                        op.vt[static_cast<size_t>(j) * rank + r] = sr[r] * vrt[static_cast<size_t>(j) * n + r];
            } else { // 780-785
                op.rank = n;
                op.u = std::move(a);
                op.vt.clear();
            }
        }
    });
}

} // namespace bbfmm
