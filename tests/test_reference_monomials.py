"""The reference's own known answers for the polynomial part of the system: the nine `monomials_*` unit tests of
ferreus_rbf/src/polynomials.rs:163-242 (values in tests/golden/reference_monomials.json, extracted by
tests/golden/make_reference_monomials.py).  evaluate_monomials (polynomials.rs:30-74) builds the matrix P of
`y_i += P[i,:] lambda` in the FGMRES matvec (rbf.rs:1366-1376, 476-491) and every domain's polynomial block
(domain.rs:171-212).  Checked: the oracle's restatement and the product's one definition (csrc/ddm_monomials.hpp, through
bbfmm_debug_evaluate_monomials: host code, no GPU needed) at the reference's tolerance (atol 1e-12 + rtol 1e-10); and
that the solver's scaled matrix is that definition on the cube-scaled points."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import ddm as D
from ferreus_rbf_rs_amd import _lib as L

with open(os.path.join(ROOT, "tests", "golden", "reference_monomials.json")) as f:
    CASES = json.load(f)["cases"]


def product_monomials(points, degree, translation=None, scale=None):
    pts = np.asfortranarray(np.asarray(points, dtype=np.float64))
    n, d = pts.shape
    basis = {1: degree + 1, 2: (degree + 1) * (degree + 2) // 2, 3: (degree + 1) * (degree + 2) * (degree + 3) // 6}[d]
    out = np.zeros((n, basis), order="F")
    tr = None if translation is None else np.ascontiguousarray(translation, dtype=np.float64)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
    rc = L.load().bbfmm_debug_evaluate_monomials(pts.ctypes.data, n, d, n, degree, None if tr is None else tr.ctypes.data,
                                                 None if sc is None else sc.ctypes.data, out.ctypes.data)
    assert rc == L.OK
    return out


def test_fixture_is_the_nine_reference_cases():
    assert len(CASES) == 9 and sorted({(len(c["points"][0]), c["degree"]) for c in CASES}) == \
        [(d, g) for d in (1, 2, 3) for g in (0, 1, 2)]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_monomials_equal_the_reference_known_answers(case):
    pts, want, degree = np.array(case["points"]), np.array(case["expected"]), case["degree"]
    d = pts.shape[1]
    tol = 1e-12 + 1e-10 * max(np.abs(want).max(), 1.0)                    # assert_mat_close, polynomials.rs:137-142
    got_oracle = D.evaluate_monomials(pts, degree, want.shape[1], np.zeros(d), np.ones(d))
    assert got_oracle.shape == want.shape and np.abs(got_oracle - want).max() <= tol
    got_product = product_monomials(pts, degree)
    assert got_product.shape == want.shape and np.abs(got_product - want).max() <= tol
    assert np.array_equal(got_product, got_oracle)                       # products of two doubles: bit for bit


def test_scaled_monomials_are_the_same_definition_on_scaled_points():
    rng = np.random.default_rng(9)
    for d in (1, 2, 3):
        pts = rng.random((50, d)) * 7.0 - 2.0
        tr, sc = D.cheb_cube_scaling_factors(pts)
        for degree in (0, 1, 2):
            a = product_monomials(pts, degree, tr, sc)
            b = D.evaluate_monomials(pts, degree, a.shape[1], tr, sc)
            assert np.array_equal(a, b)
            assert np.abs(a[:, 1:]).max(initial=0.0) <= 1.0 + 1e-15    # the Chebyshev cube: monomials of magnitude <= 1
    assert L.load().bbfmm_debug_evaluate_monomials(None, 1, 3, 1, 1, None, None, None) == L.BAD_ARGUMENT
