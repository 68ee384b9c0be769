"""Host part of the Schwarz preconditioner: the C++ domain decomposition behind the
C ABI against the numpy restatement of domain_decomposition.rs -- index sets compared for exact
equality (no GPU)."""
import numpy as np
import pytest

from ferreus_rbf_rs_amd.ddm import DDMParams, DDMTree
from oracle import ddm as D


def _oracle_levels(pts, prm):
    class NoFactor(D.Domain):                     # the decomposition only: skip the local factorisations
        def factorise(self, *a, **k):
            pass
    saved = D.Domain
    D.Domain = NoFactor
    try:
        st = D.InterpolantSettings(3, pts.shape[1])
        return D.build_ddm_tree(pts, st, D.DDMParams(prm.leaf_threshold, prm.overlap_quota, prm.coarse_ratio,
                                                     prm.coarse_threshold))
    finally:
        D.Domain = saved


def _clustered(rng, n, d):
    c = rng.random((6, d))
    return np.clip(c[rng.integers(0, 6, n)] + 0.05 * rng.standard_normal((n, d)), 0.0, 1.0)


def _same_tree(tree, ref):
    assert len(tree.levels) == len(ref) >= 2
    for lv, (a, b) in enumerate(zip(tree.levels, ref)):
        assert list(a.point_indices) == list(b.point_indices), f"level {lv}"
        assert len(a.leaf_domains) == len(b.leaf_domains)
        for da, db in zip(a.leaf_domains, b.leaf_domains):
            k = len(db.overlapping_point_indices)
            assert list(da.overlapping_point_indices) == list(db.overlapping_point_indices)
            assert list(da.internal_points_mask) == [bool(m) for m in db.internal_points_mask[:k]]
            if db.extents is not None:            # (the reference leaves the coarse domain's extents empty)
                np.testing.assert_array_equal(da.extents, db.extents)
                assert np.array_equal(np.signbit(da.extents), np.signbit(db.extents))


@pytest.mark.parametrize("dim,n,prm,clustered", [
    (1, 700, DDMParams(20, 0.5, 0.25, 60), False),
    (2, 3000, DDMParams(50, 0.5, 0.125, 200), False),
    (3, 6000, DDMParams(64, 0.5, 0.125, 256), False),
    (3, 5000, DDMParams(100, 0.25, 0.2, 300), True),
    (2, 2500, DDMParams(40, 1.0, 0.3, 100), True),
])
def test_decomposition_equals_the_restatement(dim, n, prm, clustered):
    rng = np.random.default_rng(100 * dim + n)
    pts = _clustered(rng, n, dim) if clustered else rng.random((n, dim))
    tree = DDMTree(pts, prm)
    _same_tree(tree, _oracle_levels(pts, prm))
    # the reference's structural test (domain_decomposition.rs:378-596) on the product tree
    for lvl in tree.levels:
        union = sorted(int(g) for dm in lvl.leaf_domains
                       for g, m in zip(dm.overlapping_point_indices, dm.internal_points_mask) if m)
        assert union == sorted(int(g) for g in lvl.point_indices)


def test_defaults_and_bad_arguments():
    prm = DDMParams()
    assert (prm.leaf_threshold, prm.overlap_quota, prm.coarse_ratio, prm.coarse_threshold) == (1024, 0.5, 0.125, 4096)
    pts = np.random.default_rng(1).random((5000, 3))
    t = DDMTree(pts)                              # defaults: 5000 > 4096 -> one fine level + the coarse domain
    assert len(t.levels) == 2 and len(t.levels[1].leaf_domains) == 1
    assert len(t.levels[1].point_indices) <= 4096
    with pytest.raises(ValueError):
        DDMTree(np.zeros((10, 4)))


def test_median_selection_split_equals_the_stable_argsort_with_ties():
    """ddm.cpp cuts a domain by selecting the median value and filling the lower half with the first ties
    in position order; that must be the stable argsort cut of the reference (domain_decomposition.rs:118-147)
    as the restatement does it -- duplicated coordinates and signed zeros included."""
    rng = np.random.default_rng(9)
    pts = np.round(rng.random((4000, 3)), 2) - 0.5          # many equal coordinates
    pts[::97] = 0.0
    pts[1::97, 0] = -0.0
    prm = DDMParams(50, 0.5, 0.125, 200)
    _same_tree(DDMTree(pts, prm), _oracle_levels(pts, prm))


@pytest.mark.parametrize("case", ["ties", "uniform3", "clustered2", "line1"])
def test_threaded_split_of_large_domains_equals_the_serial_pass(case, monkeypatch):
    """Domains above BBFMM_DDM_LARGE_DOMAIN points are cut on all host threads (bucketed selection of the cut value,
    per-chunk counts, ordered scatter); with the threshold at 64 points every split of these small trees takes that
    pass, and the result must still be the restatement's -- ties, signed zeros and degenerate ranges included."""
    rng = np.random.default_rng(17)
    if case == "ties":
        pts = np.round(rng.random((4000, 3)), 2) - 0.5
        pts[::97] = 0.0
        pts[1::97, 0] = -0.0
        prm = DDMParams(50, 0.5, 0.125, 200)
    elif case == "uniform3":
        pts = rng.random((6000, 3))
        prm = DDMParams(64, 0.5, 0.125, 256)
    elif case == "clustered2":
        pts = _clustered(rng, 2500, 2)
        prm = DDMParams(40, 1.0, 0.3, 100)
    else:
        pts = np.repeat(rng.random((350, 1)), 2, axis=0)        # every coordinate twice
        prm = DDMParams(20, 0.5, 0.25, 60)
    monkeypatch.setenv("BBFMM_DDM_LARGE_DOMAIN", "64")
    forced = DDMTree(pts, prm)
    monkeypatch.setenv("BBFMM_DDM_LARGE_DOMAIN", "1000000000")
    serial = DDMTree(pts, prm)
    _same_tree(forced, _oracle_levels(pts, prm))
    for a, b in zip(forced.levels, serial.levels):
        for da, db in zip(a.leaf_domains, b.leaf_domains):
            assert list(da.overlapping_point_indices) == list(db.overlapping_point_indices)
            np.testing.assert_array_equal(da.extents, db.extents)


def test_concurrent_builds_share_the_host_pool():
    """The host loops run on one persistent pool of helper threads, one loop at a time; a loop that finds the pool
    taken (another host thread's build, a nested loop) runs on threads of its own.  Four builds at once must give the
    tree of a build on its own."""
    import threading
    rng = np.random.default_rng(23)
    pts = rng.random((30000, 3))
    prm = DDMParams(64, 0.5, 0.125, 512)
    alone = DDMTree(pts, prm)
    got = [None] * 4

    def work(i):
        got[i] = DDMTree(pts, prm)                      # (ctypes drops the GIL for the call)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for tree in got:
        assert tree is not None and len(tree.levels) == len(alone.levels)
        for a, b in zip(tree.levels, alone.levels):
            assert list(a.point_indices) == list(b.point_indices)
            for da, db in zip(a.leaf_domains, b.leaf_domains):
                assert list(da.overlapping_point_indices) == list(db.overlapping_point_indices)


def test_forked_child_starts_its_own_host_pool():
    """The helper threads do not exist in a fork()ed child: it must start a pool of its own instead of waiting for
    the parent's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from ferreus_rbf_rs_amd.ddm import DDMParams, DDMTree\n"
        "pts = np.random.default_rng(1).random((30000, 3)); prm = DDMParams(64, 0.5, 0.125, 512)\n"
        "a = DDMTree(pts, prm)\n"
        "pid = os.fork()\n"
        "if pid == 0:\n"
        "    b = DDMTree(pts, prm)\n"
        "    same = all(list(x.point_indices) == list(y.point_indices) for x, y in zip(a.levels, b.levels))\n"
        "    os._exit(0 if same else 3)\n"
        "_, st = os.waitpid(pid, 0)\n"
        "c = DDMTree(pts, prm)\n"
        "sys.exit(0 if os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0 and len(c.levels) == len(a.levels) else 4)\n")
    assert subprocess.run([sys.executable, "-c", code], timeout=300).returncode == 0


def test_params_for_points_keep_three_fine_levels():
    """bbfmm_ddm_params_for_points (extension): defaults below ~2.1M points, above that a coarse threshold of
    n/470 + 1 (a level keeps at most N (1/8 + 1/341) points), so that the hierarchy ends after three fine levels."""
    for n in (10, 5000, 1_900_000):
        p = DDMParams.for_points(n)
        assert (p.leaf_threshold, p.overlap_quota, p.coarse_ratio, p.coarse_threshold) == (1024, 0.5, 0.125, 4096)
    for n in (3_000_000, 10_000_000, 40_000_000):
        p = DDMParams.for_points(n)
        assert p.coarse_threshold == n // 470 + 1 and n * (1 / 8 + 1 / 341) ** 3 <= p.coarse_threshold < n / 8 ** 2


def test_params_for_points_on_a_real_hierarchy():
    n = 2_400_000 // 8                                   # 300k points: defaults give 3 levels either way
    pts = np.random.default_rng(3).random((n, 3))
    t = DDMTree(pts, DDMParams.for_points(n))
    assert len(t.levels) <= 4 and len(t.levels[-1].leaf_domains) == 1
