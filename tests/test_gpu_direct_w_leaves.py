"""BBFMM_FLAG_DIRECT_SMALL_W_LEAVES on the device -- an extension beyond the reference (off by default): W-list leaves with
no more points than nodes are summed directly (U lists, both ways) instead of through M2P / P2L.  The direct sum is exact
where the reference approximates, so the flagged handle may differ from the default one by the reference's own M2P / P2L
error (about epsilon) and must be at least as close to the dense sum."""
import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points, relerr
from oracle import bbfmm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kid,order,br,sill,shared", [(0, 7, 1.0, 1.0, False), (3, 7, 0.3, 0.2, False), (1, 6, 1.0, 1.0, False), (2, 7, 1.0, 1.0, True)])
def test_direct_w_leaves_match_default_and_dense(kid, order, br, sill, shared):
    rng = np.random.default_rng(500 + kid)
    n = 130000
    pts = np.vstack([rng.random((n // 2, 3)), clustered_points(rng, n - n // 2, 3)])
    pts = np.unique(pts, axis=0)
    n = pts.shape[0]
    kp = F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill)
    a = F.FmmTree(pts, order, kp, True, True)
    b = F.FmmTree(pts, order, kp, True, True, direct_small_w_leaves=True, m2l_shared_basis=shared)
    sa, sb = a.stats(), b.stats()
    assert sa.n_w > 0 and sb.n_w < sa.n_w and sb.n_x == sb.n_w and sb.n_u == sa.n_u + 2 * (sa.n_w - sb.n_w)
    eps = 10.0 ** -order
    w = rng.standard_normal((n, 2))
    a.set_weights(w)
    b.set_weights(w)
    ya, yb = a.evaluate(w, pts), b.evaluate(w, pts)
    a.set_weights(np.abs(w))
    scale = max(np.abs(ya).max(), np.abs(a.evaluate(np.abs(w), pts)).max())   # the sum without cancellation
    assert np.abs(yb - ya).max() / scale < 10 * eps
    rows = rng.choice(n, 64, replace=False)
    dense = O.dense_sum(kid, br, sill, pts[rows], pts, w)
    ea, eb = np.abs(ya[rows] - dense).max() / scale, np.abs(yb[rows] - dense).max() / scale
    if shared:
        assert eb < 10 * eps, (ea, eb)                                        # (the other extension's own error, about eps)
    else:
        assert eb < 1.2 * ea + 1e-13, (ea, eb)                                # exact where the reference approximates


def test_direct_w_leaves_other_flows():
    """Arbitrary targets with gradients, the leaves-only evaluator, a row subset from host buffers, a two-way partition."""
    rng = np.random.default_rng(81)
    n = 100000
    pts = np.unique(np.vstack([rng.random((n // 2, 3)), clustered_points(rng, n // 2, 3)]), axis=0)
    n = pts.shape[0]
    kp = F.KernelParams(F.KernelType(2))
    a = F.FmmTree(pts, 6, kp, True, True)
    b = F.FmmTree(pts, 6, kp, True, True, direct_small_w_leaves=True)
    w = rng.standard_normal((n, 1))
    tg = np.clip(pts[rng.choice(n, 6000, replace=False)] + 1e-3 * rng.standard_normal((6000, 3)), pts.min(0), pts.max(0))
    a.set_weights(np.abs(w))
    sy, sg = a.evaluate_with_gradients(np.abs(w), tg)
    sy, sg = np.abs(sy).max(), np.abs(sg).max()
    a.set_weights(w)
    b.set_weights(w)
    (ya, ga), (yb, gb) = a.evaluate_with_gradients(w, tg), b.evaluate_with_gradients(w, tg)
    assert np.abs(yb - ya).max() / sy < 1e-5 and np.abs(gb - ga).max() / sg < 1e-4          # eps = 1e-6
    a.set_local_coefficients(w)
    b.set_local_coefficients(w)
    assert np.abs(b.evaluate_leaves(w, tg) - a.evaluate_leaves(w, tg)).max() / sy < 1e-5
    idx = np.sort(rng.choice(n, 9000, replace=False)).astype(np.int64)
    w1 = w[:, 0].copy()
    assert np.abs(b.fast_matrix_vector_product(w1, target_indices=idx) - a.fast_matrix_vector_product(w1, target_indices=idx)).max() / sy < 1e-5
    full = b.fast_matrix_vector_product(w1)
    parts = np.zeros(n)
    for rank in range(2):
        b.set_partition(rank, 2)
        rows = b.partition_rows()
        parts[rows] = b.fast_matrix_vector_product(w1)[rows]
    b.set_partition(0, 1)
    assert relerr(parts, full) < 1e-12
