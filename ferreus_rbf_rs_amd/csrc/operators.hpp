// Host-side precomputation of the Chebyshev / M2L operators (one-time setup; the
// device applies them).  Restates ferreus_bbfmm/src/chebyshev.rs:32-814 and
// ferreus_bbfmm/src/aca.rs with hand-written dense linear algebra (Householder QR,
// one-sided Jacobi SVD) in place of the faer 0.23.2 crate.
#pragma once
#include <cstdint>
#include <vector>

#include "kernels.hpp"

namespace bbfmm {

enum Compression : int { kCompressionNone = 0, kCompressionSvd = 1, kCompressionAca = 2 };

struct M2lOperator {
    int rank = 0;
    std::vector<double> u;  // n x rank, column-major  (PrecomputeOperators.u, bbfmm.rs:167-168)
    std::vector<double> vt; // rank x n, column-major  (PrecomputeOperators.vt, 170-171); empty if uncompressed
};

struct Operators {
    int p = 0, d = 0, n = 0;
    std::vector<double> nodes;  // p, ascending (chebyshev.rs:32-40)
    std::vector<double> polyn;  // p x p row-major: polyn[j*p + k] = T_k(node_j) (chebyshev.rs:669-670)
    std::vector<double> nodes_nd; // n x d row-major, axis 0 slowest (chebyshev.rs:666)
    // 1-D child transfer S(child node a, parent node i), one p x p block per side
    // (chebyshev.rs:146-180): xfer[side][a*p + i].  The n x n M2M matrices of the
    // reference are Kronecker products of these (chebyshev.rs:196-241).
    std::vector<double> xfer;
    int n_vec = 0;  // 7^d
    int n_ref = 0;  // 2 / 7 / 16
    int n_perm = 0; // 2^d * d!
    std::vector<int32_t> all_vecs;   // n_vec x d (chebyshev.rs:268-269)
    std::vector<int32_t> ref_vecs;   // n_ref x d (chebyshev.rs:272-294)
    std::vector<int32_t> perm;       // n_perm x n (permutation_indices, 544-555)
    std::vector<int32_t> invperm;    // n_perm x n (inverse_permutations, 557-560)
    std::vector<int32_t> perm_lookup; // n_vec (permutation_lookups, 574-575)
    std::vector<int32_t> ref_lookup;  // n_vec (reference_vector_lookups, 577)
    // level -> ref -> operator; levels 2..=depth (chebyshev.rs:697-699)
    std::vector<std::vector<M2lOperator>> m2l;
    int compression = kCompressionAca;
};

void precompute_operators(int p, int d, double radius, int depth, const KernelSpec &kernel,
                          int compression, double epsilon, Operators *out);

// Dense n x n M2M matrix of child `ci` (row = parent node, col = child node), as the
// reference stores it (chebyshev.rs:216-240).  Only used by tests.
void dense_m2m_matrix(const Operators &ops, int ci, std::vector<double> *out);

// ---- small dense helpers (column-major), exposed for tests ----
// Thin Householder QR of a (m x k, m >= k): q is m x k, r is k x k upper triangular.
void thin_qr(const std::vector<double> &a, int m, int k, std::vector<double> *q,
             std::vector<double> *r);
// SVD of a (m x k, m >= k) by one-sided Jacobi: a = u * diag(s) * vt, s descending,
// u m x k, vt k x k.
void jacobi_svd(const std::vector<double> &a, int m, int k, std::vector<double> *u,
                std::vector<double> *s, std::vector<double> *vt);
// aca.rs:210-247
int singular_values_cutoff(const std::vector<double> &sigma, double epsilon);

} // namespace bbfmm
