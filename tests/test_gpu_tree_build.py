"""The source tree built on the device (csrc/tree_device.hip; SURVEY.md 8(f)-4, linear_tree.rs:20-175) against the
oracle and against the host build (csrc/tree.cpp, the bit-exact checker): keys, leaf flags, per-leaf point lists
(row order included) and the U / V / W / X lists derived from them -- exactly equal.  Same cases as
tests/test_host_structure.py, plus full-size ones compared array for array with the host build."""
import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points
from oracle import bbfmm_oracle as O
from test_host_structure import CASES, assert_same_structure

pytestmark = pytest.mark.gpu


def _trees(pts, order=5, kid=0, adaptive=True, sparse=True, extents=None, params=None):
    fp = None if params is None else F.FmmParams(*params)
    kp = F.KernelParams(F.KernelType(kid))
    dev = F.FmmTree(pts, order, kp, adaptive, sparse, extents=extents, params=fp)
    host = F.FmmTree(pts, order, kp, adaptive, sparse, extents=extents, params=fp, host_only=True)
    return dev, host


def _assert_identical(dev, host):
    kd, ld = dev.cells()
    kh, lh = host.cells()
    assert np.array_equal(kd, kh) and np.array_equal(ld, lh)
    pd_, id_ = dev.leaf_sources()
    ph, ih = host.leaf_sources()
    assert np.array_equal(pd_, ph) and np.array_equal(id_, ih)          # same rows, same order inside every leaf
    for name in "UVWX":
        a, b = dev.interaction_list(name), host.interaction_list(name)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), name
    sd, sh = dev.stats(), host.stats()
    assert (sd.depth, sd.n_cells, sd.n_leaves, sd.p2p_pairs) == (sh.depth, sh.n_cells, sh.n_leaves, sh.p2p_pairs)


@pytest.mark.parametrize("name", list(CASES))
def test_device_built_tree_equals_oracle_and_host_build(name):
    c = dict(CASES[name])
    pts = c.pop("pts")(np.random.default_rng(11))
    order = c.pop("order", 5)
    c.setdefault("params", (256, 2, 1e-5, 1024))
    dev, host = _trees(pts, order=order, **c)
    assert dev.tree_built_on_device() and not host.tree_built_on_device()
    r = O.FmmTree(pts, order, 0, c.get("adaptive", True), c.get("sparse", True), c.get("extents"), O.FmmParams(*c["params"]))
    assert_same_structure(dev, r)
    _assert_identical(dev, host)


@pytest.mark.parametrize("n,d,kw", [
    (1, 3, {}), (7, 3, {}), (300, 2, dict(adaptive=False)), (257, 1, {}),
    (200000, 3, {}), (200000, 3, dict(sparse=False, extents=[-0.5] * 3 + [1.5] * 3)),
    (150000, 3, dict(adaptive=False)), (100000, 2, dict(adaptive=False, sparse=False)),
    (120000, 3, dict(params=(17, 2, 1e-5, 1024))), (50000, 1, dict(params=(5, 2, 1e-5, 1024))),
])
def test_device_build_edge_and_larger_cases(n, d, kw):
    rng = np.random.default_rng(n + d)
    pts = rng.random((n, d)) if n < 1000 or kw else clustered_points(rng, n, d)
    if n == 200000 and not kw:
        pts[:5000] = pts[0]                                              # coincident points: subdivision to level 16
        pts[5000:5010, 0] = [0.0, 1.0 - 2 ** -53, 0.5, 0.25, 0.75, 2 ** -20, 0.5 - 2 ** -30, 0.5 + 2 ** -30, 0.125, 0.875]
    dev, host = _trees(pts, **kw)
    assert dev.tree_built_on_device()
    _assert_identical(dev, host)


def test_points_outside_the_root_box_fall_back_to_the_host_build():
    """Explicit extents smaller than the data (sources outside the root box take arbitrary per-level keys,
    linear_tree.rs:56-61): the device path declines and the host build runs -- same result as ever."""
    rng = np.random.default_rng(5)
    pts = rng.random((5000, 3)) * 2 - 0.5
    dev, host = _trees(pts, sparse=False, extents=[0.0] * 3 + [1.0] * 3, params=(40, 2, 1e-5, 1024))
    assert not dev.tree_built_on_device()
    _assert_identical(dev, host)


def test_full_size_10m_device_and_host_builds_agree():
    n = 10_000_000
    pts = np.random.default_rng(42).random((n, 3))
    dev, host = _trees(pts, order=4)
    assert dev.tree_built_on_device()
    _assert_identical(dev, host)
    # and the matvec on the device-built tree reproduces sampled dense rows
    import torch
    w = torch.rand((1, n), dtype=torch.float64, device="cuda")
    y = torch.zeros_like(w)
    dev.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
    idx = np.random.default_rng(2).choice(n, 32, replace=False)
    yd = O.dense_sum(0, 1.0, 1.0, pts[idx], pts, w.cpu().numpy().T.copy())
    assert np.abs(y.cpu().numpy().T[idx] - yd).max() < 2e-3 * np.abs(yd).max()      # order 4: ~1e-4
