"""Host logic of the product (C++ tree / lists / operators / stacked M2L tables behind the C ABI,
created with BBFMM_FLAG_HOST_ONLY) against the oracle.  Integer structure is compared for exact
equality as sets (SURVEY.md section 0 finding 5: the reference's own column numbering is
hash-order dependent); operators to the compression tolerance.  No GPU."""
import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points, inject_product_operators, relerr
from oracle import bbfmm_oracle as O


def both(pts, order=7, kid=0, adaptive=True, sparse=True, extents=None, params=None, br=1.0, sill=1.0):
    fp = None if params is None else F.FmmParams(*params)
    op = None if params is None else O.FmmParams(*params)
    t = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), adaptive, sparse,
                  extents=extents, params=fp, host_only=True)
    r = O.FmmTree(pts, order, kid, adaptive, sparse, extents, op, base_range=br, total_sill=sill)
    return t, r


def assert_same_structure(t, r):
    st = r.structure()
    keys, leaf = t.cells()
    s = t.stats()
    assert s.depth == st["depth"] and s.radius == st["radius"]
    assert [s.center[a] for a in range(s.d)] == st["center"]
    assert keys.tolist() == sorted(st["tree"], key=lambda k: (k & O.LEVEL_MASK, k))   # (level, key) order
    assert sorted(keys[leaf == 1].tolist()) == st["leaves"]
    k2i = {int(k): i for i, k in enumerate(keys)}
    ptr, idx = t.leaf_sources()
    got = {int(keys[c]): idx[ptr[c]:ptr[c + 1]].tolist() for c in range(len(keys)) if ptr[c + 1] > ptr[c]}
    assert got == {k: v for k, v in st["leaf_source_indices"].items() if v}
    for name in "UVWX":
        p, i = t.interaction_list(name)
        got = {int(keys[c]): sorted(int(keys[j]) for j in i[p[c]:p[c + 1]]) for c in range(len(keys)) if p[c + 1] > p[c]}
        ref = {k: v for k, v in st[name.lower()].items() if v}
        assert got == ref, f"{name} lists differ"
    return keys, k2i


CASES = {
    "uniform3d": dict(pts=lambda g: g.random((6000, 3))),
    "mixed3d": dict(pts=lambda g: g.random((9000, 3)), params=(60, 2, 1e-7, 1024)),
    "clustered3d": dict(pts=lambda g: clustered_points(g, 6000, 3), params=(40, 2, 1e-7, 1024)),
    "nonsparse_extents": dict(pts=lambda g: g.random((3000, 3)) * 2 - 1, sparse=False, extents=[-2, -2, -2, 2, 2, 2],
                              params=(50, 2, 1e-7, 1024)),
    "regular": dict(pts=lambda g: g.random((5000, 3)), adaptive=False, params=(100, 2, 1e-7, 1024)),
    "regular_nonsparse": dict(pts=lambda g: g.random((3000, 3)), adaptive=False, sparse=False, params=(200, 2, 1e-7, 1024)),
    "planar2d": dict(pts=lambda g: g.random((5000, 2)), params=(30, 2, 1e-7, 1024)),
    "line1d": dict(pts=lambda g: g.random((2000, 1)), params=(20, 2, 1e-7, 1024)),
    "duplicates_deep": dict(pts=lambda g: np.vstack([g.random((500, 3)), np.tile(g.random((1, 3)), (300, 1))]),
                            params=(64, 2, 1e-5, 1024), order=5),
}


@pytest.mark.parametrize("name", list(CASES))
def test_tree_and_lists_identical(name):
    c = dict(CASES[name])
    pts = c.pop("pts")(np.random.default_rng(11))
    order = c.pop("order", 5)
    c.setdefault("params", (256, 2, 1e-5, 1024))
    t, r = both(pts, order=order, **c)
    assert_same_structure(t, r)
    if name == "duplicates_deep":
        assert t.stats().depth == 16 + 0 or t.stats().depth > 8   # coincident points descend to MAXIMUM_LEVEL


def test_target_assignment_and_error_index():
    rng = np.random.default_rng(12)
    pts = rng.random((4000, 3)) * 2 - 1
    t, r = both(pts, order=4, sparse=False, extents=[-2, -2, -2, 2, 2, 2], params=(50, 2, 1e-4, 1024))
    keys, _ = t.cells()
    x = rng.random((3000, 3)) * 4 - 2
    cells = t.points_to_leaves(x)
    ok = O.points_to_keys(x, r.tl.leaves, r.depth, r.center, r.radius, 3)
    assert keys[cells].tolist() == ok.tolist()
    assert keys[t.points_to_leaves(pts)].tolist() == O.points_to_keys(pts, r.tl.leaves, r.depth, r.center, r.radius, 3).tolist()
    x[[17, 900]] = [[9.0, 0.0, 0.0], [0.0, -7.5, 0.0]]
    with pytest.raises(F.PointOutsideTree) as e:
        t.points_to_leaves(x)
    assert e.value.point_index == 17               # smallest offending row (linear_tree.rs:514-517)
    with pytest.raises(O.PointOutsideTree) as e2:
        O.points_to_keys(x, r.tl.leaves, r.depth, r.center, r.radius, 3)
    assert e2.value.point_index == 17


def test_reference_unit_test_tree_through_the_abi():
    # bbfmm.rs:1464-1500 (host part: the tree of the 1-D unit test and the failing target)
    t = F.FmmTree(np.array([[0.5]]), 3, F.KernelParams(F.FmmKernelType.LinearRbf), True, False, extents=[0.0, 1.0],
                  host_only=True)
    s = t.stats()
    assert (s.d, s.depth, s.n_cells, s.n_leaves) == (1, 1, 3, 2) and s.radius == pytest.approx(0.501)
    with pytest.raises(F.PointOutsideTree) as e:
        t.points_to_leaves(np.array([[0.5], [10.0]]))
    assert e.value.point_index == 1


@pytest.mark.parametrize("d,order", [(3, 5), (2, 6), (1, 7)])
def test_symmetry_tables_identical(d, order):
    pts = np.random.default_rng(13).random((600, d))
    t, r = both(pts, order=order, params=(64, 2, 1e-4, 1024))
    perm, inv, pl, rl = t.permutation_tables()
    assert np.array_equal(perm, r.ops.perm) and np.array_equal(inv, r.ops.invperm)
    assert np.array_equal(pl, r.ops.perm_lookup) and np.array_equal(rl, r.ops.ref_lookup)
    for ci in range(1 << d):
        assert np.abs(t.debug_dense_m2m(ci) - r.ops.m2m[ci]).max() < 1e-14


@pytest.mark.parametrize("kid,comp,br,sill", [(0, 2, 1, 1), (1, 2, 1, 1), (3, 2, 0.5, 0.4), (2, 1, 1, 1), (2, 0, 1, 1),
                                             (7, 2, 1, 1), (100, 2, 1.0, 1.0)])
def test_m2l_operators_accuracy_and_ranks(kid, comp, br, sill):
    """U*Vt approximates the dense kernel block (aca.rs / chebyshev.rs:697-791) exactly as well as
    the oracle's independent ACA + LAPACK recompression does, with the same ranks.  (The ACA
    stopping rule, aca.rs:114-131, is a heuristic: for singular kernels the achieved Frobenius
    error exceeds epsilon in both implementations alike.)"""
    eps = 1e-6
    pts = np.random.default_rng(14).random((3000, 3))
    t, r = both(pts, order=5, kid=kid, params=(64, comp, eps, 1024), br=br, sill=sill)
    ranks = t.m2l_ranks()
    n = 125
    nodes = r.ops.nodes
    idx = O.cartesian_product(np.arange(5), 3)
    for lvl in range(2, r.depth + 1):
        length = r.radius / float(2 ** (lvl - 1))
        tp = r.ops.nodes_nd * (0.5 * length)
        for ref in (0, 3, 7, 15):
            sp = (r.ops.ref_vecs[ref][None, :] + nodes[idx] * 0.5) * length
            A = O.kernel_block(kid, br, sill, sp, tp)
            P = t.m2l_operator(lvl, ref)
            err = np.linalg.norm(P - A) / np.linalg.norm(A)
            Q = r.ops.u[lvl][ref] if comp == 0 else r.ops.u[lvl][ref] @ r.ops.vt[lvl][ref]
            err_oracle = np.linalg.norm(Q - A) / np.linalg.norm(A)
            assert err < (1e-13 if comp == 0 else 100 * eps), (lvl, ref, err)
            assert err <= 1.05 * err_oracle + 1e-13, (lvl, ref, err, err_oracle)
            want = r.ops.u[lvl][ref].shape[1]
            assert abs(int(ranks[lvl, ref]) - want) <= 1
            assert ranks[lvl, ref] == (n if comp == 0 else ranks[lvl, ref])


def test_stacked_m2l_tables_reproduce_the_reference_grouping():
    """The per-class stacked operators the MFMA kernels consume (permutations folded in) give the
    same local expansions as the reference's gather / GEMM / GEMM / scatter (bbfmm.rs:864-986),
    here applied by the oracle with the same U, Vt."""
    rng = np.random.default_rng(15)
    for pts, order, params in [(rng.random((5000, 3)), 5, (40, 2, 1e-6, 1024)),
                               (clustered_points(rng, 3000, 3), 4, (30, 2, 1e-5, 1024)),
                               (rng.random((3000, 2)), 6, (30, 2, 1e-6, 1024)),
                               (rng.random((2500, 3)), 4, (60, 0, 1e-6, 1024))]:
        t, r = both(pts, order=order, params=params)
        inject_product_operators(t, r)
        r.set_weights(rng.random((pts.shape[0], 1)))
        M = r.M[0].copy()
        Lp = t.debug_apply_m2l_tables_host(M)
        r.L = np.zeros_like(r.M)
        lib = O.lib()
        compressed = 0 if params[1] == 0 else 1
        for level in range(2, r.depth + 1):
            cells = np.ascontiguousarray(r.level_cells[level])
            buf, u_off, vt_off, rank = r.opbuf[level]
            lib.oracle_m2l(O.I32(r.ops.n), O.I64(r.C), O.I32(1), O._p(cells), O.I64(len(cells)), O._p(r.v_ptr),
                           O._p(r.v_idx), O._p(r.v_tidx), O.I32(len(rank)), O._p(u_off), O._p(vt_off), O._p(rank),
                           O._p(buf), O.I32(compressed), O._p(r.ops.perm), O._p(r.ops.invperm),
                           O._p(r.ops.perm_lookup), O._p(r.ops.ref_lookup), O._p(r.M), O._p(r.L))
        assert relerr(Lp, r.L[0]) < 1e-12


def test_stage1_boundary_variants_cover_every_cell_once_and_reproduce_m2l(monkeypatch):
    """A tree deep enough for runs of >= 128 same-class cells that miss the same targets (the faces of a 32^3
    level): those get stacked stage-1 operators without the transfer vectors towards the missing targets.  The
    host walk of the SAME tile list the device launch uses (bbfmm_debug_apply_m2l_tables_host) must visit every
    source cell exactly once and reproduce the reference's M2L."""
    rng = np.random.default_rng(17)
    g = (np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing="ij"), -1).reshape(-1, 3)[:, None, :]
         + 0.15 + 0.7 * rng.random((32 ** 3, 2, 3))).reshape(-1, 3) / 32.0
    params = (6, 2, 1e-3, 1024)
    monkeypatch.setenv("BBFMM_M2L_VARIANTS", "1")      # runs of one full tile qualify (default: four; the faces here hold 256 cells)
    t, r = both(g, order=3, params=params)
    nv, nc = t.debug_m2l_variants()
    assert nv >= 6 * 8 and nc >= 128 * nv            # at least the six faces of each of the eight classes of level 5
    inject_product_operators(t, r)
    r.set_weights(rng.random((g.shape[0], 1)))
    M = r.M[0].copy()
    Lp = t.debug_apply_m2l_tables_host(M)
    r.L = np.zeros_like(r.M)
    lib = O.lib()
    for level in range(2, r.depth + 1):
        cells = np.ascontiguousarray(r.level_cells[level])
        buf, u_off, vt_off, rank = r.opbuf[level]
        lib.oracle_m2l(O.I32(r.ops.n), O.I64(r.C), O.I32(1), O._p(cells), O.I64(len(cells)), O._p(r.v_ptr),
                       O._p(r.v_idx), O._p(r.v_tidx), O.I32(len(rank)), O._p(u_off), O._p(vt_off), O._p(rank),
                       O._p(buf), O.I32(1), O._p(r.ops.perm), O._p(r.ops.invperm),
                       O._p(r.ops.perm_lookup), O._p(r.ops.ref_lookup), O._p(r.M), O._p(r.L))
    assert relerr(Lp, r.L[0]) < 1e-12


def test_partition_rows_cover_all_points_once():
    pts = np.random.default_rng(16).random((20000, 3))
    t = F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            t.set_partition(rank, world)
            parts.append(t.partition_rows())
        allr = np.concatenate(parts)
        assert len(allr) == 20000 and len(np.unique(allr)) == 20000
        sizes = [len(p) for p in parts]
        assert max(sizes) < 1.5 * min(sizes)                  # balanced for a uniform cloud
    t.set_partition(0, 1)
    assert np.array_equal(t.partition_rows(), np.arange(20000))


def test_tree_statistics_definitions():
    # P2P pair count and tile bytes as defined there
    pts = np.random.default_rng(17).random((5000, 3))
    t, r = both(pts, order=4, params=(50, 2, 1e-4, 1024))
    s = t.stats()
    pairs = 0
    tile_bytes = 0
    for c in r.leaf_cells:
        nt = r.src_ptr[c + 1] - r.src_ptr[c]
        ns = sum(r.src_ptr[u + 1] - r.src_ptr[u] for u in r.u_idx[r.u_ptr[c]:r.u_ptr[c + 1]])
        pairs += nt * ns
        if nt:
            tile_bytes += (nt + ns) * (8 * 3 + 8)
    assert s.p2p_pairs == pairs and s.p2p_tile_bytes_k1 == tile_bytes
    flops = sum(4.0 * 64 * r.ops.u[int(r.cell_level[c])][int(r.ops.ref_lookup[t_])].shape[1]
                for c in range(r.C) for t_ in r.v_tidx[r.v_ptr[c]:r.v_ptr[c + 1]])
    assert s.m2l_flops_k1 == pytest.approx(flops, rel=1e-12)
    assert s.n_v == len(r.v_idx) and s.n_u == len(r.u_idx) and s.n_w == len(r.w_idx) and s.n_x == len(r.x_idx)


@pytest.mark.parametrize("budget_mb,order,n,leaf", [(None, 4, 6000, 40), (3.0, 4, 6000, 40), (0.8, 4, 6000, 40), (0.25, 4, 6000, 40),
                                                    (0.1, 3, 5000, 30), (0.4, 5, 3000, 30)])
def test_bounded_m2l_intermediate_batches_reproduce_m2l(monkeypatch, budget_mb, order, n, leaf):
    """The slots of the two M2L stages go through ONE buffer of bounded length (BBFMM_M2L_CBUF_MB): whole levels
    while they fit, else 2 / 4 / 8 groups of a level's target classes with one stacked stage-1 operator per (group,
    source class).  The host walk of the very tables and tile lists the device launches use -- one buffer, batch
    after batch, zero-fill lists for the absent pairs -- must reproduce the reference's M2L (bbfmm.rs:864-986)
    whatever the cut, on a mixed-level clustered tree (boundary cells, absent pairs everywhere)."""
    rng = np.random.default_rng(23)
    pts = np.vstack([rng.random((n, 3)), clustered_points(rng, n // 2, 3)])
    if budget_mb is None:
        monkeypatch.delenv("BBFMM_M2L_CBUF_MB", raising=False)
    else:
        monkeypatch.setenv("BBFMM_M2L_CBUF_MB", str(budget_mb))
    params = (leaf, 2, 1e-5, 1024)
    t, r = both(pts, order=order, params=params)
    st = t.stats()
    if budget_mb is None:
        assert st.m2l_batches == 1
    else:
        assert st.m2l_batches > 1
        if budget_mb <= 0.25:
            assert st.m2l_batches > st.depth - 1                 # some level was cut into groups of classes
    inject_product_operators(t, r)
    r.set_weights(rng.random((pts.shape[0], 1)))
    M = r.M[0].copy()
    Lp = t.debug_apply_m2l_tables_host(M)
    r.L = np.zeros_like(r.M)
    lib = O.lib()
    for level in range(2, r.depth + 1):
        cells = np.ascontiguousarray(r.level_cells[level])
        buf, u_off, vt_off, rank = r.opbuf[level]
        lib.oracle_m2l(O.I32(r.ops.n), O.I64(r.C), O.I32(1), O._p(cells), O.I64(len(cells)), O._p(r.v_ptr),
                       O._p(r.v_idx), O._p(r.v_tidx), O.I32(len(rank)), O._p(u_off), O._p(vt_off), O._p(rank),
                       O._p(buf), O.I32(1), O._p(r.ops.perm), O._p(r.ops.invperm),
                       O._p(r.ops.perm_lookup), O._p(r.ops.ref_lookup), O._p(r.M), O._p(r.L))
    assert relerr(Lp, r.L[0]) < 1e-12
    assert st.m2l_slots_bytes_per_rhs > 0


def test_targets_are_sources_comparison_is_bit_for_bit_and_row_for_row():
    """What bbfmm_evaluate asks before it serves the unchanged caller (rbf.rs:1357-1364) from the resident target set:
    the threaded host comparison alone, on a host-only handle (the device side: tests/test_gpu_unchanged_caller.py)."""
    rng = np.random.default_rng(31)
    n = 700_000                                                   # several 2 MB pieces per axis: every helper thread compares
    pts = rng.random((n, 3))
    pts[123, 2] = 0.0
    t = F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
    assert t.debug_targets_are_sources(pts) and t.debug_targets_are_sources(pts.copy())
    x = pts.copy()
    x[n - 1, 2] = np.nextafter(x[n - 1, 2], 2.0)                  # the very last value, one ulp
    assert not t.debug_targets_are_sources(x)
    x = pts.copy()
    x[[5, 600_000]] = x[[600_000, 5]]                             # the same set of rows, two swapped
    assert not t.debug_targets_are_sources(x)
    x = pts.copy()
    x[123, 2] = -0.0                                              # equal as a number, not as bits
    assert not t.debug_targets_are_sources(x)
    assert not t.debug_targets_are_sources(pts[:-1]) and not t.debug_targets_are_sources(np.vstack([pts, pts[:1]]))
    with pytest.raises(ValueError):                               # another dimension: refused by the binding (the C ABI reads
        t.debug_targets_are_sources(pts[:, :2])                   # d columns off a pointer -- ASan caught the first version of this line)


def test_rows_of_the_sources_lookup():
    """The table over the source points behind the unchanged caller's matvec_partial (rbf.rs:119-133 ->
    select_mat_rows(source_points, idx)): built once by all host threads (compare-and-swap inserts), looked up per target."""
    rng = np.random.default_rng(32)
    n = 300_000
    pts = rng.random((n, 3))
    pts[1000] = pts[7]                                             # two rows with the same coordinates
    t = F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
    idx = rng.choice(n, 40_000, replace=False)
    rows = t.debug_rows_of_sources(pts[idx])
    assert rows is not None and np.array_equal(pts[rows], pts[idx])          # a row with exactly these coordinates
    keep = (idx != 7) & (idx != 1000)
    assert np.array_equal(rows[keep], idx[keep])                               # unique points: the row itself
    both = t.debug_rows_of_sources(pts[[7, 1000]])
    assert both[0] == both[1] and both[0] in (7, 1000)                          # equal points are interchangeable
    x = pts[idx].copy()
    x[-1, 1] = np.nextafter(x[-1, 1], 2.0)
    assert t.debug_rows_of_sources(x) is None                                  # one target that is no source point
    assert t.debug_rows_of_sources(rng.random((100, 3))) is None
    t2 = F.FmmTree(pts[:, :2].copy(), 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
    r2 = t2.debug_rows_of_sources(pts[idx, :2])
    assert r2 is not None and np.array_equal(pts[r2, :2], pts[idx, :2])


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_source_coordinates_are_refused_with_a_message(bad):
    """A NaN passes every comparison of the extents and the tree build (the reference's saturating casts would send it to
    cell 0): bbfmm_create says which row and column instead (a NaN coordinate was accepted silently)."""
    pts = np.random.default_rng(3).random((500, 3))
    pts[123, 1] = bad
    with pytest.raises(ValueError, match=r"non-finite coordinate \(row 123, column 1\)"):
        F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
