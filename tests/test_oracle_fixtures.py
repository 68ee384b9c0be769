"""Pins the CPU oracle: known answers derivable from the reference text, the committed golden
vectors (tests/golden/small_cases.json, generator alongside) and structural invariants of the
tree / interaction lists.  No GPU."""
import json
import math
import os

import numpy as np
import pytest

from conftest import ROOT, relerr
from oracle import bbfmm_oracle as O

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "small_cases.json")))


# ---- known answers from the reference text -------------------------------------------------
def test_tree_center_and_radius_known_answer():
    # morton.rs:349-373 with the unit test's extents [0, 1] (bbfmm.rs:1473)
    c, r = O.calculate_tree_center_and_radius([0.0, 1.0])
    assert c == [0.5] and r == pytest.approx(0.501, abs=1e-15)
    c, r = O.calculate_tree_center_and_radius([-0.3, 0.2, 0.1, 0.9, 1.7, 0.4])
    assert c == [0.0, 1.0, 0.5] and r == pytest.approx(1.001, abs=1e-15)


def test_reference_unit_test_point_outside_tree():
    # ferreus_bbfmm/src/bbfmm.rs:1464-1500: 1-D, one source at 0.5, extents [0,1], order 3,
    # adaptive, non-sparse; targets [0.5, 10.0] -> PointOutsideTree{point_index: 1}
    t = O.FmmTree(np.array([[0.5]]), 3, O.KERNEL_IDS["LinearRbf"], True, False, [0.0, 1.0])
    t.set_weights(np.array([[1.0]]))
    with pytest.raises(O.PointOutsideTree) as e:
        t.evaluate(np.array([[1.0]]), np.array([[0.5], [10.0]]))
    assert e.value.point_index == 1
    assert "target point at row 1 lies outside the tree extents" in str(e.value)   # bbfmm.rs:32-36
    ok = t.evaluate(np.array([[1.0]]), np.array([[0.5], [0.25]]))
    assert ok[0, 0] == 0.0 and ok[1, 0] == pytest.approx(-0.25)


def test_reference_vectors_known_answer():
    # chebyshev.rs:245-266: 2 / 7 / 16 reference vectors; the 3-D list
    assert O.get_m2l_vectors(1)[1].shape[0] == 2
    assert O.get_m2l_vectors(2)[1].shape[0] == 7
    allv, ref = O.get_m2l_vectors(3)
    assert allv.shape == (343, 3)
    expect = {(2, 0, 0), (2, 1, 0), (2, 2, 0), (3, 0, 0), (3, 1, 0), (3, 2, 0), (3, 3, 0), (2, 1, 1), (2, 2, 1),
              (3, 1, 1), (3, 2, 1), (3, 3, 1), (2, 2, 2), (3, 2, 2), (3, 3, 2), (3, 3, 3)}
    assert {tuple(int(v) for v in r) for r in ref} == expect
    # transfer index formula, bbfmm.rs:989-998: sum 7^(d-1-i) (t_i + 3)
    for idx in (0, 171, 342, 57):
        t = allv[idx]
        assert idx == (t[0] + 3) * 49 + (t[1] + 3) * 7 + (t[2] + 3)


def test_cartesian_product_doctest():
    # ferreus_rbf_utils/src/utils.rs:58-75: cartesian_product(&[0, 1], 2)
    assert O.cartesian_product(np.array([0, 1]), 2).tolist() == [[0, 0], [0, 1], [1, 0], [1, 1]]


def test_remaining_utils_doctests():
    """The other three doctests of ferreus_rbf_utils/src/utils.rs, as known answers of the restatement's equivalents:
    argsort (92-102: [30, 10, 20] -> [1, 2, 0]; the permutation tables are built with it), select_mat_rows (19-43: the
    target subset of the matvec, rbf.rs:1359-1360) and get_distance (263-280: (1, 2) to (4, 6) is 5) -- the latter through
    the dense sum with the linear kernel phi(r) = -r."""
    assert list(O.argsort_stable([30, 10, 20])) == [1, 2, 0]
    m = np.array([[0.0, 1.0], [1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])
    assert O.select_mat_rows(m, [0, 2]).tolist() == [[0.0, 1.0], [2.0, 2.0]]    # the function the oracle's matvec calls
    assert O.select_mat_rows(m, [3, 0, 3]).tolist() == [[3.0, 3.0], [0.0, 1.0], [3.0, 3.0]]
    pts = np.array([[1.0, 2.0], [4.0, 6.0]])
    y = O.dense_sum(O.KERNEL_IDS["LinearRbf"], 1.0, 1.0, pts, pts, np.array([[1.0], [1.0]]))
    assert y[:, 0].tolist() == [-5.0, -5.0]                               # -get_distance, exactly
    assert float(O.kernel_phi(O.KERNEL_IDS["LinearRbf"], 5.0)) == -5.0


def test_pointarray_extents_doctest():
    # ferreus_rbf_utils/src/utils.rs:181-194
    pts = np.array([[1.0, 5.0], [3.0, 2.0], [-1.0, 4.0]])
    assert O.get_pointarray_extents(pts) == [-1.0, 2.0, 3.0, 5.0]


def test_chebyshev_nodes_ascending_and_transfer_partition_of_unity():
    nodes = O.generate_chebyshev_nodes(7)                      # chebyshev.rs:32-40
    assert np.all(np.diff(nodes) > 0) and nodes[0] == pytest.approx(-math.cos(math.pi / 14))
    polyn, _ = O.evaluate_chebyshev_polynomials(7, nodes)
    s = O.calculate_sn(O.evaluate_chebyshev_polynomials(7, np.linspace(-1, 1, 9))[0], polyn, 7)
    assert np.allclose(s.sum(axis=1), 1.0, atol=1e-13)          # interpolation reproduces constants
    mats, _ = O.get_m2m_transfer_matrices(7, nodes, polyn, 3)
    assert len(mats) == 8 and mats[0].shape == (343, 343)
    assert np.allclose(mats[5].sum(axis=0), 1.0, atol=1e-12)    # each child node distributes a unit mass


def test_permutation_tables_are_consistent():
    allv, ref = O.get_m2l_vectors(3)
    perm, inv, pl, rl = O.get_permutation_lookups(3, 4, allv, ref)
    assert perm.shape == (48, 64)                               # 2^d * d!
    for c in range(48):
        assert sorted(perm[c]) == list(range(64))
        assert np.array_equal(perm[c][inv[c]], np.arange(64))   # inverse = argsort
    assert rl[int((2 + 3) * 49 + 3 * 7 + 3)] == 0 and rl[342] == 15   # [2,0,0] -> 0, [3,3,3] -> 15


def test_morton_primitives_bit_exact():
    # key = (interleave(x, y, z) << 15) | level with x in bit 0 (morton.rs:58-119)
    assert O.encode_morton_point((1, 0, 0), 1, 3) == (1 << 15) | 1
    assert O.encode_morton_point((0, 1, 0), 1, 3) == (2 << 15) | 1
    assert O.encode_morton_point((0, 0, 1), 1, 3) == (4 << 15) | 1
    assert O.encode_morton_point((3, 5), 3, 2) == (0b100111 << 15) | 3
    rng = np.random.default_rng(0)
    for _ in range(200):
        lvl = int(rng.integers(1, 17))
        a = tuple(int(v) for v in rng.integers(0, 1 << lvl, size=3))
        key = O.encode_morton_point(a, lvl, 3)
        assert O.decode_key(key, 3) == (a, lvl)
        par = O.get_parent(key, 3)
        assert key in O.get_children(par, 3)
        assert O.get_child_index(key, 3) == (a[0] & 1) | ((a[1] & 1) << 1) | ((a[2] & 1) << 2)
    assert O.get_parent(0, 3) is None
    assert len(O.get_neighbours(O.encode_morton_point((1, 1, 1), 2, 3), 3)) == 26
    assert len(O.get_neighbours(O.encode_morton_point((0, 0, 0), 2, 3), 3)) == 7
    # vectorised point -> key agrees with the scalar path, including saturation of negatives
    pts = np.array([[0.1, 0.2, 0.3], [-5.0, 0.5, 0.5], [0.999, 0.999, 0.999]])
    keys = O.points_to_anchor_keys(pts, 3, [0.5, 0.5, 0.5], 0.501)
    side = O.get_side_length(0.501, 3)
    for p, k in zip(pts, keys):
        anc = [max(0, int(math.floor((v - (0.5 - 0.501)) / side))) for v in p]
        assert int(k) == O.encode_morton_point(anc, 3, 3)


def test_kernel_known_values():
    # phi(0): 0 for Linear/TPS/Cubic, sill for spheroidal, 0 for the singular kernels (guards at
    # non_rbf_kernels.rs:24-29); continuity of the two spheroidal branches at s*r = inflexion
    K = O.KERNEL_IDS
    for name in ("LinearRbf", "ThinPlateSplineRbf", "CubicRbf", "Laplacian", "OneOverR2", "OneOverR4"):
        assert O.kernel_phi(K[name], 0.0) == 0.0
    assert O.kernel_phi(K["Spheroidal5Rbf"], 0.0, 2.0, 1.5) == 1.5
    assert O.kernel_phi(K["LinearRbf"], 2.5) == -2.5
    assert O.kernel_phi(K["CubicRbf"], 2.0) == 8.0
    assert O.kernel_phi(K["ThinPlateSplineRbf"], 2.0) == pytest.approx(4.0 * math.log(2.0), rel=1e-15)
    assert O.kernel_phi(K["Laplacian"], 4.0) == 0.25
    consts = {3: (0.5, 2.6798340586), 5: (0.4082482905, 1.5822795750), 7: (0.3535533906, 1.2008676644),
              9: (0.3162277660, 1.0)}
    for order, (ip, scaling) in consts.items():
        kid = K[f"Spheroidal{order}Rbf"]
        r = ip / scaling                                       # base_range = 1
        lo, hi = O.kernel_phi(kid, r * (1 - 1e-9)), O.kernel_phi(kid, r * (1 + 1e-9))
        assert lo == pytest.approx(hi, rel=2e-8)               # constants carry 10 digits


# ---- golden vectors -------------------------------------------------------------------------
@pytest.mark.parametrize("case", GOLD["dense"]["cases"], ids=lambda c: c["kernel"])
def test_dense_sum_matches_numpy_golden(case):
    src = np.array(GOLD["dense"]["sources"])
    tgt = np.array(GOLD["dense"]["targets"])
    w = np.array(GOLD["dense"]["weights"])
    kw = case["params"]
    y = O.dense_sum(O.KERNEL_IDS[case["kernel"]], kw.get("base_range", 1.0), kw.get("total_sill", 1.0), tgt, src, w)
    assert relerr(y, np.array(case["y"])) < 1e-13


def test_tree_and_lists_match_golden_and_invariants():
    g = GOLD["tree"]
    pts = np.array(g["points"])
    t = O.FmmTree(pts, 4, 0, True, True, params=O.FmmParams(g["max_points_per_cell"], O.COMPRESSION_ACA, 1e-4, 1024))
    st = t.structure()
    assert st["depth"] == g["depth"] and st["center"] == g["center"] and st["radius"] == g["radius"]
    assert [str(k) for k in st["tree"]] == g["tree"]
    assert [str(k) for k in st["leaves"]] == g["leaves"]
    assert {str(k): v for k, v in st["leaf_source_indices"].items()} == g["leaf_source_indices"]
    for name in "uvwx":
        assert {str(k): [str(x) for x in v] for k, v in st[name].items()} == g[name], name
    # invariants (linear_tree.rs:189-220): every point in exactly one leaf; U symmetric and
    # reflexive on leaves; V same level, symmetric, non adjacent; X = transpose of W
    leaves = set(st["leaves"])
    seen = sorted(i for v in st["leaf_source_indices"].values() for i in v)
    assert seen == list(range(len(pts)))
    for k, lst in st["leaf_source_indices"].items():
        assert len(lst) <= g["max_points_per_cell"] or (k & O.LEVEL_MASK) == 16
    for b, ul in st["u"].items():
        assert b in ul and set(ul) <= leaves
        for u in ul:
            assert b in st["u"][u]
            assert O.are_adjacent(b, u, t.center, t.radius, 3)
    for b, vl in st["v"].items():
        for v in vl:
            assert (v & O.LEVEL_MASK) == (b & O.LEVEL_MASK) and b in st["v"][v]
            assert not O.are_adjacent(b, v, t.center, t.radius, 3)
            assert O.are_adjacent(O.get_parent(b, 3), O.get_parent(v, 3), t.center, t.radius, 3)
    assert sum(len(v) for v in st["w"].values()) > 0
    for b, wl in st["w"].items():
        for w in wl:
            assert b in st["x"][w] and (w & O.LEVEL_MASK) > (b & O.LEVEL_MASK)
            assert not O.are_adjacent(b, w, t.center, t.radius, 3)
            assert O.are_adjacent(b, O.get_parent(w, 3), t.center, t.radius, 3)
    assert sum(len(v) for v in st["x"].values()) == sum(len(v) for v in st["w"].values())


def test_fmm_on_golden_dense_problem():
    """The oracle's FMM (order 7, defaults) reproduces the numpy golden sums to the BBFMM accuracy."""
    src = np.array(GOLD["dense"]["sources"])
    tgt = np.array(GOLD["dense"]["targets"])
    w = np.array(GOLD["dense"]["weights"])
    case = GOLD["dense"]["cases"][0]
    t = O.FmmTree(src, 7, 0, True, True, extents=[0, 0, 0, 1, 1, 1], params=O.FmmParams(16, O.COMPRESSION_ACA, 1e-7, 1024))
    t.set_weights(w)
    assert relerr(t.evaluate(w, tgt), np.array(case["y"])) < 5e-7
