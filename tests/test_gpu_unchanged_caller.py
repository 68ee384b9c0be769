"""The unchanged caller of the FGMRES matvec: ferreus_rbf/src/rbf.rs:1357-1364 calls `set_weights(w)` and then
`evaluate(w, select_mat_rows(source_points, all rows))`.  bbfmm_evaluate recognises N targets that are the handle's
sources bit for bit and row for row and serves them from the resident sorted target set (what bbfmm_matvec_device
runs); anything else takes the general path.  Both against the oracle at 1e-11 (same host-computed operators: only
the summation order differs), against each other and against the patched entry point at 1e-12."""
import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points, inject_product_operators, relerr
from oracle import bbfmm_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-11


@pytest.fixture(scope="module")
def case():
    rng = np.random.default_rng(77)
    pts = np.vstack([rng.random((120000, 3)), clustered_points(rng, 10000, 3)])   # mixed levels: W / X lists live
    pts[5, 1] = 0.0                                                                # a coordinate whose sign bit can flip
    t = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
    r = O.FmmTree(pts, 7, 0, True, True)
    inject_product_operators(t, r)
    assert t.stats().n_w > 0
    return rng, pts, t, r


def test_targets_equal_sources_take_the_resident_path(case):
    rng, pts, t, r = case
    n = len(pts)
    w = np.asfortranarray(rng.standard_normal((n + 4, 1)))       # N + basis_size rows, as the solver's vectors (rbf.rs:1344)
    t.set_weights(w)
    y = t.evaluate(w, pts.copy())                                # a fresh copy of the rows: select_mat_rows, rbf.rs:1359-1360
    assert t.last_evaluate_at_sources()
    r.set_weights(w[:n])
    yr = r.evaluate(w[:n], pts)
    assert relerr(y, yr) < TOL
    assert relerr(t.debug_get_coefficients("L", 1), r.L) < TOL
    ym = t.fast_matrix_vector_product(w[:, 0].copy(), basis_size=4)          # the patched caller
    assert relerr(y[:, 0], ym[:n]) < 1e-12
    # again on the same handle, behind an entry point that replaced the staged weights
    t.set_weights(w)
    y2 = t.evaluate(w, pts)
    assert t.last_evaluate_at_sources() and relerr(y2, y) < 1e-12


def test_two_rhs_and_strided_weights(case):
    rng, pts, t, r = case
    n = len(pts)
    w = np.asfortranarray(rng.standard_normal((n + 10, 2)))
    t.set_weights(w)
    y = t.evaluate(w, pts)
    assert t.last_evaluate_at_sources() and y.shape == (n, 2)
    r.set_weights(w[:n])
    assert relerr(y, r.evaluate(w[:n], pts)) < TOL


def test_one_perturbed_target_takes_the_general_path(case):
    rng, pts, t, r = case
    n = len(pts)
    w = rng.standard_normal((n, 1))
    x = pts.copy()
    x[n // 2, 2] = np.nextafter(x[n // 2, 2], 0.0)                # one bit of one coordinate
    t.set_weights(w)
    y = t.evaluate(w, x)
    assert not t.last_evaluate_at_sources()
    r.set_weights(w)
    assert relerr(y, r.evaluate(w, x)) < TOL
    y0 = t.evaluate(w, pts)                                       # and the two paths agree where the targets agree
    assert t.last_evaluate_at_sources()
    keep = np.arange(n) != n // 2
    assert relerr(y[keep], y0[keep]) < 1e-12


def test_row_permuted_sources_are_not_mistaken_for_the_sources(case):
    rng, pts, t, r = case
    n = len(pts)
    w = rng.standard_normal((n, 1))
    perm = np.arange(n)
    perm[[10, n - 3]] = perm[[n - 3, 10]]                         # the same point set, two rows swapped
    t.set_weights(w)
    y = t.evaluate(w, pts[perm])
    assert not t.last_evaluate_at_sources()
    y0 = t.evaluate(w, pts)
    assert t.last_evaluate_at_sources()
    assert relerr(y, y0[perm]) < 1e-12
    assert abs(y0[10, 0] - y0[n - 3, 0]) > 1e-6 * np.abs(y0).max()   # the swap is visible in the values
    full = rng.permutation(n)                                     # and a full shuffle
    yf = t.evaluate(w, pts[full])
    assert not t.last_evaluate_at_sources() and relerr(yf, y0[full]) < 1e-12


def test_negative_zero_is_not_positive_zero(case):
    rng, pts, t, r = case
    n = len(pts)
    w = rng.standard_normal((n, 1))
    x = pts.copy()
    x[5, 1] = -0.0                                                # equal as a number, different as bits: general path, same values
    t.set_weights(w)
    y = t.evaluate(w, x)
    assert not t.last_evaluate_at_sources()
    y0 = t.evaluate(w, pts)
    assert t.last_evaluate_at_sources() and relerr(y, y0) < 1e-12


def test_other_weights_in_evaluate_than_in_set_weights(case):
    """The reference reads the multipoles left by set_weights and the weights handed to evaluate (bbfmm.rs:444-507: P2P
    and P2L take the argument): a caller that passes different vectors gets exactly that mixture on both paths."""
    rng, pts, t, r = case
    n = len(pts)
    w1, w2 = rng.standard_normal((n, 1)), rng.standard_normal((n, 1))
    r.set_weights(w1)
    yr = r.evaluate(w2, pts)
    t.set_weights(w1)
    y = t.evaluate(w2, pts)
    assert t.last_evaluate_at_sources() and relerr(y, yr) < TOL
    x = pts.copy()
    x[0, 0] = np.nextafter(x[0, 0], 1.0)
    t.set_weights(w1)
    yg = t.evaluate(w2, x)
    assert not t.last_evaluate_at_sources()
    assert relerr(yg[1:], yr[1:]) < 1e-9                          # (row 0 moved by one ulp)
    # the staged copy follows the weights that are on the device: w2 now, so w1 must be transferred again
    y1 = t.evaluate(w1, pts)
    r.set_weights(w1)
    assert relerr(y1, r.evaluate(w1, pts)) < TOL


def test_deterministic_handle_and_leaves_after_the_resident_path():
    rng = np.random.default_rng(5)
    pts = rng.random((60000, 3))
    w = rng.standard_normal((60000, 1))
    t = F.FmmTree(pts, 5, F.KernelParams(F.FmmKernelType.CubicRbf), True, True, deterministic=True)
    t.set_weights(w)
    a = t.evaluate(w, pts)
    assert t.last_evaluate_at_sources()
    t.set_weights(w)
    b = t.evaluate(w, pts)
    assert np.array_equal(a, b)                                   # ordered pairs, no atomics: bit for bit
    x = pts[:100] * (1 - 1e-9)
    t.set_local_coefficients(w)
    z = t.evaluate_leaves(w, x)
    assert relerr(z, t.evaluate(w, x)) < 1e-12


def test_gradients_at_the_sources_keep_the_general_path(case):
    rng, pts, t, r = case
    n = len(pts)
    t2 = F.FmmTree(pts[:30000], 6, F.KernelParams(F.FmmKernelType.CubicRbf), True, True)
    w = rng.standard_normal((30000, 1))
    t2.set_weights(w)
    y, g = t2.evaluate_with_gradients(w, pts[:30000])
    assert not t2.last_evaluate_at_sources()
    y0 = t2.evaluate(w, pts[:30000])
    assert t2.last_evaluate_at_sources() and relerr(y, y0) < 1e-12 and g.shape == (30000, 3)


def test_rows_of_the_sources_take_the_cached_subset_plan(case):
    """The unchanged caller's matvec_partial (rbf.rs:119-133 with Some(target_indices)): set_weights(w), then
    evaluate(w, select_mat_rows(source_points, idx)).  Targets that are rows of the sources are recognised (a table over
    the source points, bit for bit) and served by the plan bbfmm_fast_matrix_vector_product(target_indices) caches."""
    rng, pts, t, r = case
    n = len(pts)
    w = rng.standard_normal((n, 1))
    idx = np.sort(rng.choice(n, n // 8, replace=False))
    x = pts[idx]                                                  # select_mat_rows
    t.set_weights(w)
    y_first = t.evaluate(w, x)                                    # first sighting of the index set: recognised, but no plan is
    assert t.last_evaluate_path() == 0                            # built for a caller that may never come back (round 6)
    y = t.evaluate(w, x)                                          # second sighting: the plan is built and kept
    assert t.last_evaluate_path() == 2 and not t.last_evaluate_at_sources()
    assert relerr(y_first, y) < 1e-12
    r.set_weights(w)
    yr = r.evaluate(w, x)
    assert relerr(y, yr) < TOL
    ym = t.fast_matrix_vector_product(w[:, 0].copy(), target_indices=idx)      # the patched caller: the same plan
    assert relerr(y[:, 0], ym[idx]) < 1e-12
    y2 = t.evaluate(w, x)                                         # (the plan is cached now)
    assert t.last_evaluate_path() == 2 and relerr(y2, y) < 1e-12
    # unsorted rows with a repeated one: still rows of the sources, values follow the rows
    idx2 = rng.permutation(idx)[: n // 16]
    idx2[3] = idx2[7]
    t.evaluate(w, pts[idx2])
    y3 = t.evaluate(w, pts[idx2])
    assert t.last_evaluate_path() == 2
    full = t.evaluate(w, pts)[:, 0]
    assert relerr(y3[:, 0], full[idx2]) < 1e-12
    # one target that is no source point: the general path, same values elsewhere
    x4 = x.copy()
    x4[11, 0] = np.nextafter(x4[11, 0], 1.0)
    y4 = t.evaluate(w, x4)
    assert t.last_evaluate_path() == 0
    keep = np.arange(len(idx)) != 11
    assert relerr(y4[keep], y[keep]) < 1e-12 and relerr(y4, r.evaluate(w, x4)) < TOL
    # small batches (an evaluator's grid) are not looked up at all
    t.evaluate(w, pts[:500])
    assert t.last_evaluate_path() == 0
    # nor is anything on a tree made the evaluator's way (explicit extents, not sparse: rbf.rs:677-690)
    te = F.FmmTree(pts, 5, F.KernelParams(F.FmmKernelType.LinearRbf), True, False, extents=[0, 0, 0, 1, 1, 1])
    te.set_weights(w)
    ze = te.evaluate(w, x)
    assert te.last_evaluate_path() == 0 and relerr(ze, y) < 1e-4          # (order 5: the method's accuracy)


def test_leading_dimensions_larger_than_the_row_counts(case):
    """faer views may carry a column stride larger than their row count (utils.rs:425-429): targets with ldx > m, weights
    with ldw > rows, output with ldo > m -- through the C ABI directly, on the resident path and on the row-subset path."""
    import ctypes
    from ferreus_rbf_rs_amd import _lib as L
    rng, pts, t, r = case
    lib = L.load()
    n = len(pts)
    wbuf = np.zeros((n + 9, 1), order="F")
    wbuf[:n, 0] = rng.standard_normal(n)
    bad = ctypes.c_int64(-1)

    def run(xrows):
        m = len(xrows)
        xbuf = np.full((m + 7, 3), np.nan, order="F")
        xbuf[:m] = pts[xrows]
        obuf = np.full((m + 5, 1), np.nan, order="F")
        assert lib.bbfmm_set_weights(t._h, wbuf.ctypes.data, n + 3, 1, n + 9) == 0           # rows n + 3 of a buffer with ld n + 9
        rc = lib.bbfmm_evaluate(t._h, wbuf.ctypes.data, n + 3, 1, n + 9, xbuf.ctypes.data, m, m + 7, obuf.ctypes.data, m + 5,
                                ctypes.byref(bad))
        assert rc == 0 and np.isnan(obuf[m:]).all() and not np.isnan(obuf[:m]).any()         # nothing written past m
        return obuf[:m].copy()

    y_all = run(np.arange(n))
    assert t.last_evaluate_path() == 1
    t.set_weights(wbuf[:n])
    assert relerr(y_all, t.evaluate(wbuf[:n], pts)) < 1e-12
    idx = np.sort(rng.choice(n, n // 10, replace=False))
    run(idx)
    y_sub = run(idx)                                              # (second sighting of the row set: the cached-plan path)
    assert t.last_evaluate_path() == 2 and relerr(y_sub, y_all[idx]) < 1e-12
