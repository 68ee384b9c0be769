"""The oracle's BBFMM against the dense direct sum it approximates (the ground truth of this
path: the reference's tests hold no numeric check).  Expected agreement
at the default epsilon = 10^-order is ~10^-order * O(10).  No GPU."""
import numpy as np
import pytest

from conftest import clustered_points, relerr
from oracle import bbfmm_oracle as O

K = O.KERNEL_IDS


def run(pts, kid, order=7, adaptive=True, sparse=True, extents=None, params=None, nrhs=1, grads=False,
        targets=None, br=1.0, sill=1.0, seed=0):
    rng = np.random.default_rng(seed)
    n = pts.shape[0]
    w = rng.random((n, nrhs))
    t = O.FmmTree(pts, order, kid, adaptive, sparse, extents, params, base_range=br, total_sill=sill)
    t.set_weights(w)
    tp = pts if targets is None else targets
    idx = rng.choice(len(tp), min(len(tp), 400), replace=False)
    if grads:
        y, g = t.evaluate_with_gradients(w, tp)
        yd, gd = O.dense_sum(kid, br, sill, tp[idx], pts, w, True)
        return t, relerr(y[idx], yd), relerr(g[idx], gd)
    y = t.evaluate(w, tp)
    yd = O.dense_sum(kid, br, sill, tp[idx], pts, w)
    return t, relerr(y[idx], yd), 0.0


def test_uniform_3d_linear_two_rhs():
    t, e, _ = run(np.random.default_rng(1).random((20000, 3)), K["LinearRbf"], nrhs=2)
    assert t.depth == 3 and e < 1e-6


def test_mixed_levels_exercise_w_and_x_lists():
    t, e, _ = run(np.random.default_rng(2).random((40000, 3)), K["LinearRbf"],
                  params=O.FmmParams(80, O.COMPRESSION_ACA, 1e-7, 1024))
    assert len(t.w_idx) > 0 and len(t.x_idx) == len(t.w_idx) and e < 1e-6


def test_clustered_thin_plate_spline_order_9():
    t, e, _ = run(clustered_points(np.random.default_rng(3), 15000, 3), K["ThinPlateSplineRbf"], order=9)
    assert t.depth >= 4 and e < 1e-7


@pytest.mark.parametrize("name,br,sill,tol", [("CubicRbf", 1, 1, 1e-6), ("Spheroidal3Rbf", 0.5, 0.4, 5e-6),
                                              ("Spheroidal9Rbf", 0.5, 0.4, 5e-6), ("Laplacian", 1, 1, 1e-6),
                                              ("OneOverR2", 1, 1, 5e-6), ("OneOverR4", 1, 1, 5e-5),
                                              ("GaussianExt", 1.0, 1.0, 1e-5), ("MultiquadricExt", 0.3, 0.3, 1e-6)])
def test_kernels(name, br, sill, tol):
    _, e, _ = run(np.random.default_rng(4).random((8000, 3)), K[name], br=br, sill=sill)
    assert e < tol


def test_gradients_cubic():
    _, e, ge = run(np.random.default_rng(5).random((12000, 3)), K["CubicRbf"], grads=True)
    assert e < 1e-6 and ge < 1e-5


def test_two_and_one_dimensional_trees():
    _, e2, _ = run(np.random.default_rng(6).random((10000, 2)), K["LinearRbf"])
    _, e1, _ = run(np.random.default_rng(7).random((3000, 1)), K["LinearRbf"])
    assert e2 < 1e-6 and e1 < 1e-10


def test_non_sparse_tree_with_explicit_extents_and_outside_targets():
    # the evaluator flow of the reference's third doctest (ferreus_bbfmm/src/lib.rs:236-293)
    rng = np.random.default_rng(8)
    pts = rng.random((8000, 3)) * 2 - 1
    tg = rng.random((1000, 3)) * 4 - 2
    _, e, _ = run(pts, K["LinearRbf"], sparse=False, extents=[-2, -2, -2, 2, 2, 2], nrhs=2, targets=tg)
    assert e < 1e-6


def test_regular_tree_and_other_compressions():
    pts = np.random.default_rng(9).random((9000, 3))
    assert run(pts, K["LinearRbf"], adaptive=False)[1] < 1e-6
    assert run(pts, K["LinearRbf"], adaptive=False, sparse=False)[1] < 1e-6
    assert run(pts, K["CubicRbf"], params=O.FmmParams(256, O.COMPRESSION_SVD, 1e-7, 1024))[1] < 1e-6
    assert run(pts, K["CubicRbf"], params=O.FmmParams(256, O.COMPRESSION_NONE, 1e-7, 1024))[1] < 1e-6


def test_leaf_only_evaluator_and_matvec_caller():
    rng = np.random.default_rng(10)
    pts = rng.random((6000, 3))
    w = rng.random((6000, 1))
    t = O.FmmTree(pts, 7, K["LinearRbf"], True, False, [0, 0, 0, 1, 1, 1])
    t.set_weights(w)
    t.set_local_coefficients(w)                      # bbfmm.rs:518-524
    x = rng.random((500, 3))
    assert relerr(t.evaluate_leaves(w, x), O.dense_sum(0, 1, 1, x, pts, w)) < 1e-6
    # fast_matrix_vector_product, ferreus_rbf/src/rbf.rs:1338-1379
    wf = rng.random(6004)
    P = np.hstack([np.ones((6000, 1)), pts])
    sub = rng.choice(6000, 700, replace=False)
    r = O.fast_matrix_vector_product(t, wf, 4, sub, P, 0.05)
    ref = np.zeros(6004)
    ref[sub] = O.dense_sum(0, 1, 1, pts[sub], pts, wf[:6000, None])[:, 0] + 0.05 * wf[sub] + P[sub] @ wf[6000:]
    assert relerr(r, ref) < 1e-6 and np.all(r[6000:] == 0.0)
    mask = np.ones(6004, bool)
    mask[sub] = False
    assert np.all(r[mask] == 0.0)


def test_gemm_shaped_port_equals_the_plain_loop_passes():
    """bench.py's cpu_baseline runs the oracle with M2L as gather + two register-blocked FMA GEMMs +
    permuted scatter (oracle_m2l_gemm; the shape of bbfmm.rs:910-982 with faer's GEMMs) and the near field on gathered
    copies in vectorised loops.  The plain-loop passes every parity test uses are its checker: 1e-13 on L and on the
    potentials, compressed and uncompressed operators, two right-hand sides, mixed levels (P2L / M2P run too)."""
    rng = np.random.default_rng(12)
    pts = np.vstack([rng.random((26000, 3)), np.clip(rng.normal(size=(4000, 3)) * 0.03 + 0.4, 0.0, 0.999)])
    w = rng.standard_normal((len(pts), 2))
    for kid, params in ((O.KERNEL_IDS["LinearRbf"], None), (O.KERNEL_IDS["CubicRbf"], O.FmmParams(64, O.COMPRESSION_NONE, 1e-5, 1024)),
                        (O.KERNEL_IDS["Spheroidal3Rbf"], None)):
        t = O.FmmTree(pts, 5, kid, True, True, None, params, base_range=0.5, total_sill=0.4)
        got = {}
        for mode in (False, True):
            t.gemm_shaped = mode
            t.set_weights(w)
            got[mode] = (t.evaluate(w, pts), t.L.copy())
        assert np.abs(got[True][1] - got[False][1]).max() <= 1e-13 * np.abs(got[False][1]).max(), kid
        assert np.abs(got[True][0] - got[False][0]).max() <= 1e-13 * np.abs(got[False][0]).max(), kid
