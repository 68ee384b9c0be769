"""The reference's own example workloads on the reference's own data.

Data: tests/golden/albatite_SD_points.npz = the 35,801 x 4 values of datasets/albatite_SD_points.csv (drill-hole
samples; coordinates of 3.3e5 / 7.7e6 / +-4e2 -- NOT recentred by these tests), made by tests/golden/make_albatite_fixture.py,
which also restates the reference's duplicate removal (rbf.rs:1430-1467: nothing is removed, smallest spacing 6e-3 against
a cutoff of 1e-13).

Workloads: ferreus_rbf/examples/isosurface_spheroidal.rs:86-115 (Spheroidal order 3, base_range 50, sill 10, no drift,
absolute tolerance 0.01) and isosurface_linear.rs:88-107 (Linear, its minimum drift = constant, absolute tolerance 0.01);
both with the defaults of Params (config.rs:141-150: FGMRES 20 x 5, DDMParams::default, FmmParams::new_defaults = order
7, 256 points per cell, ACA, eps 1e-7), adaptive sparse tree, extents from the data (rbf.rs:456-467).

  * CPU (`-m "not gpu"`): fixture integrity; the oracle's matvec on the raw coordinates against dense rows; the
    oracle's solve of the spheroidal example converges (FGMRES + Schwarz restatements).
  * GPU: (i) the matvec -- device against the oracle on the same operators at 1e-11 (M, L, potentials; through
    bbfmm_evaluate as the unchanged caller calls it, and through bbfmm_fast_matrix_vector_product) and against dense rows;
    (ii) the solves -- device FGMRES + Schwarz against oracle/solvers.py + oracle/ddm.py + the oracle's products:
    residual histories equal to 1e-8, converged, fitted values within 0.01 of the data; the iteration counts go to
    gpurun_out/albatite_solve_histories.json (committed copy under profiles/).
The reference documents one residual history of its own (py_ferreus_rbf/docs/api/progress.md:53-70: 8 iterations,
1.8e2 -> 2.5e-3, on a 26,988-point set that is not in its repository): a different data set, qualitative only."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, inject_product_operators, relerr
from oracle import bbfmm_oracle as O
from oracle import ddm as D
from oracle import solvers as OS

FIXTURE = os.path.join(ROOT, "tests", "golden", "albatite_SD_points.npz")
EXAMPLES = {   # name: kernel id, base_range, total_sill, dense-row tolerance (the BBFMM's own accuracy at order 7)
    "spheroidal": (O.KERNEL_IDS["Spheroidal3Rbf"], 50.0, 10.0, 1e-5),
    "linear": (O.KERNEL_IDS["LinearRbf"], 1.0, 1.0, 1e-6),
}
ORDER = 7           # get_default_fmm_interpolation_order, config.rs:200-207 (Linear and Spheroidal: 7)
TOLERANCE = 0.01    # FittingAccuracyType::Absolute


@pytest.fixture(scope="module")
def data():
    z = np.load(FIXTURE)
    rows = z["rows"]
    assert rows.shape == (35801, 4)
    for key in ("keep_spheroidal3", "keep_linear"):
        assert np.array_equal(z[key], np.arange(35801))          # remove_duplicates keeps every row
    return np.ascontiguousarray(rows[:, :3]), rows[:, 3].copy()


def test_fixture_is_the_reference_data_set(data):
    pts, vals = data
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "albatite_SD_points.json")))
    assert meta["rows"] == 35801 and meta["kernels"]["Spheroidal3Rbf"]["kept"] == 35801
    assert pts[0].tolist() == [329314.1, 7744801.47, 406.0] and vals[0] == 224.768     # first data row of the CSV
    assert 3.29e5 < pts[:, 0].min() and pts[:, 1].max() < 7.75e6                          # raw coordinates
    assert len(np.unique(pts, axis=0)) == 35801


@pytest.mark.parametrize("name", list(EXAMPLES))
def test_oracle_matvec_on_the_raw_coordinates_against_dense_rows(data, name):
    pts, _ = data
    kid, br, sill, tol = EXAMPLES[name]
    r = O.FmmTree(pts, ORDER, kid, True, True, None, None, base_range=br, total_sill=sill)
    assert r.depth >= 4
    w = np.random.default_rng(1).standard_normal((len(pts), 1))
    r.set_weights(w)
    y = r.evaluate(w, pts)
    idx = np.random.default_rng(2).choice(len(pts), 400, replace=False)
    assert relerr(y[idx], O.dense_sum(kid, br, sill, pts[idx], pts, w)) < tol


def _oracle_solve(pts, vals, kid, br, sill, otree):
    ost = D.InterpolantSettings(kid, 3, None, 0.0, br, sill)      # drift None: the kernel's minimum (get_min_drift)
    m = ost.basis_size
    mono = ortho = None
    if m:
        tr, sc = D.cheb_cube_scaling_factors(pts)
        mono, ortho = D.orthonormal_poly(pts, ost, tr, sc)
    levels = D.build_ddm_tree(pts, ost, D.DDMParams())
    rhs = np.concatenate([vals, np.zeros(m)])
    mv = lambda w: O.fast_matrix_vector_product(otree, w, m, None, mono, 0.0)
    pv = lambda w, idx: O.fast_matrix_vector_product(otree, w, m, idx, mono, 0.0)
    pre = lambda v: D.schwarz_preconditioner(v, levels, pv, ost, ortho)
    x, hist = OS.fgmres(mv, rhs, pre, None, 20, 5, OS.ABSOLUTE, TOLERANCE)
    return x, [float(h[1]) for h in hist], levels, mv, m


@pytest.mark.timeout(900)
def test_oracle_solves_the_spheroidal_example(data):
    pts, vals = data
    kid, br, sill, _ = EXAMPLES["spheroidal"]
    otree = O.FmmTree(pts, ORDER, kid, True, True, None, None, base_range=br, total_sill=sill)
    x, hist, levels, mv, m = _oracle_solve(pts, vals, kid, br, sill, otree)
    assert [len(lv.point_indices) for lv in levels] == [35801, 4480, 560]      # DDMParams::default on this set
    assert hist[-1] < TOLERANCE and len(hist) <= 8
    assert np.abs(mv(x)[:len(pts)] - vals).max() < TOLERANCE                    # the interpolant fits the data


# ------------------------------------------------------------------------------------------------ on the device
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(EXAMPLES))
def test_device_matvec_parity_on_the_reference_data(data, name):
    import ferreus_rbf_rs_amd as F
    pts, _ = data
    n = len(pts)
    kid, br, sill, tol = EXAMPLES[name]
    t = F.FmmTree(pts, ORDER, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, True)
    r = O.FmmTree(pts, ORDER, kid, True, True, None, None, base_range=br, total_sill=sill)
    inject_product_operators(t, r)
    assert t.stats().depth == r.depth and t.stats().n_cells == len(r.cell_keys)
    w = np.random.default_rng(1).standard_normal((n, 1))
    t.set_weights(w)
    r.set_weights(w)
    assert relerr(t.debug_get_coefficients("M", 1), r.M) < 1e-11
    y = t.evaluate(w, pts)                                       # the unchanged caller's sequence (rbf.rs:1357-1364)
    assert t.last_evaluate_at_sources()
    yr = r.evaluate(w, pts)
    assert relerr(t.debug_get_coefficients("L", 1), r.L) < 1e-11
    assert relerr(y, yr) < 1e-11
    ym = t.fast_matrix_vector_product(w[:, 0].copy())
    assert relerr(ym, yr[:, 0]) < 1e-11
    x = pts.copy()
    x[0, 0] = np.nextafter(x[0, 0], 0.0)                         # and the general path (ordered pairs) on the same data
    yg = t.evaluate(w, x)
    assert not t.last_evaluate_at_sources() and relerr(yg[1:], yr[1:]) < 1e-11
    idx = np.random.default_rng(2).choice(n, 400, replace=False)
    assert relerr(y[idx], O.dense_sum(kid, br, sill, pts[idx], pts, w)) < tol


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("name", list(EXAMPLES))
def test_device_solve_of_the_reference_example_equals_the_restatement(data, name):
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd import solvers as S
    from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
    pts, vals = data
    n = len(pts)
    kid, br, sill, _ = EXAMPLES[name]
    tree = F.FmmTree(pts, ORDER, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, True)
    otree = O.FmmTree(pts, ORDER, kid, True, True, None, None, base_range=br, total_sill=sill)
    inject_product_operators(tree, otree)
    xo, ho, levels, mv, m = _oracle_solve(pts, vals, kid, br, sill, otree)
    st = InterpolantSettings(kid, 3, None, 0.0, br, sill)
    assert st.basis_size == m == (0 if name == "spheroidal" else 1)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams())
    assert pre.num_levels == len(levels)
    for lv in range(len(levels)):
        assert np.array_equal(pre.level_points(lv), np.asarray(levels[lv].point_indices)), lv
    op = S.RbfSystemOperator(tree, m, pre.monomial_matrix, 0.0)
    rhs = np.concatenate([vals, np.zeros(m)])
    x, hist = S.fgmres(op, rhs, pre, None, 20, 5, S.FittingAccuracy(TOLERANCE, S.FittingAccuracyType.Absolute))
    hd = [float(h[1]) for h in hist]
    assert len(hd) == len(ho) and hd[-1] < TOLERANCE                         # converged, in the same number of iterations
    diff = max(abs(a - b) / b for a, b in zip(hd, ho))
    assert diff < 1e-8, (hd, ho)
    fit = float(np.abs(op(x)[:n] - vals).max())
    assert fit < TOLERANCE                                                    # fitted values within 0.01 of the data
    assert relerr(x[:n], xo[:n]) < 1e-6                                       # and the coefficients themselves
    # The examples go on to evaluate the interpolant on a 5 m grid inside the data's bounding box (build_isosurface,
    # examples/isosurface_spheroidal.rs:118-129): a second tree over the same points, adaptive, NOT sparse, with explicit
    # extents (RBFInterpolator::_setup_fmmtree / evaluate, rbf.rs:590-630, 677-690), weights = the solved coefficients.
    # Full mode and Leaves mode (set_local_coefficients + evaluate_leaves, rbf.rs:830-838), kernel part only.
    ext = np.concatenate([pts.min(0), pts.max(0)])
    rng = np.random.default_rng(9)
    grid = np.floor(ext[:3] / 5.0) * 5.0 + 5.0 * rng.integers(1, ((ext[3:] - ext[:3]) // 5.0).astype(int) - 1, size=(6000, 3))
    assert np.all(grid >= ext[:3]) and np.all(grid <= ext[3:])
    coef = x[:n, None].copy()
    te = F.FmmTree(pts, ORDER, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, False, extents=list(ext))
    oe = O.FmmTree(pts, ORDER, kid, True, False, list(ext), None, base_range=br, total_sill=sill)
    inject_product_operators(te, oe)
    te.set_weights(coef)
    oe.set_weights(coef)
    v, vo = te.evaluate(coef, grid), oe.evaluate(coef, grid)
    assert relerr(v, vo) < 1e-11
    te.set_local_coefficients(coef)
    assert relerr(te.evaluate_leaves(coef, grid), vo) < 1e-11
    sub = rng.choice(len(grid), 300, replace=False)
    assert relerr(v[sub], O.dense_sum(kid, br, sill, grid[sub], pts, coef)) < 1e-4   # the BBFMM's accuracy on the solved weights
    vg, gg = te.evaluate_with_gradients(coef, grid)
    _, go = oe.evaluate_with_gradients(coef, grid)
    assert relerr(vg, vo) < 1e-11 and relerr(gg, go) < 1e-9
    out = os.path.join(ROOT, "gpurun_out", "albatite_solve_histories.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    rec = json.load(open(out)) if os.path.exists(out) else {}
    rec[name] = {"points": n, "kernel": F.KernelType(kid).name, "base_range": br, "total_sill": sill, "basis_size": m,
                 "tolerance": TOLERANCE, "tolerance_type": "Absolute", "ddm_level_sizes": [len(lv.point_indices) for lv in levels],
                 "iterations": len(hd), "device_history": hd, "oracle_history": ho, "max_rel_diff_of_histories": diff,
                 "max_abs_misfit_at_the_data": fit,
                 "reference_documented_log": "py_ferreus_rbf/docs/api/progress.md:53-70: 8 iterations 1.808e2 -> 2.452e-3, Spheroidal, "
                                             "26,988 points, absolute 0.01 -- DIFFERENT DATA SET (not in the repository), qualitative only"}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
