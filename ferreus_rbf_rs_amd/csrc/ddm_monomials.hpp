// evaluate_monomials of ferreus_rbf (polynomials.rs:30-74): one row of the monomial matrix, 1 | x_a | x_a x_b (a <= b),
// on the point already translated and scaled.  One definition for the solver's global matrix (schwarz.cpp), the
// per-domain matrices (ddm_solver.cpp) and the entry point the reference's own known answers are checked through
// (bbfmm_debug_evaluate_monomials; polynomials.rs:163-242 -> tests/golden/reference_monomials.json).
#pragma once
#include <cstddef>

namespace bbfmm {

// row[j * stride], j = 0 .. basis_size - 1; degree -1: nothing, 0: constant, 1: linear, 2: quadratic
inline void monomial_row(const double *sx, int d, int degree, double *row, size_t stride) {
    if (degree < 0) return;
    row[0] = 1.0;
    if (degree >= 1)
        for (int a = 0; a < d; ++a) row[static_cast<size_t>(1 + a) * stride] = sx[a];
    if (degree == 2) {
        int c = 1 + d;
        for (int a = 0; a < d; ++a)
            for (int b = a; b < d; ++b) row[static_cast<size_t>(c++) * stride] = sx[a] * sx[b];
    }
}

inline int monomial_basis_size(int d, int degree) { // set_basis_size, interpolant_config.rs:150-178
    const int k = degree + 1;
    return degree < 0 ? 0 : (d == 1 ? k : (d == 2 ? k * (k + 1) / 2 : k * (k + 1) * (k + 2) / 6));
}

} // namespace bbfmm
