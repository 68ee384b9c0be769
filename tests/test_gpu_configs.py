"""BASELINE.json configs 2, 3 and 4 on a real MI355X ("configs not exercised by the driver-run
-m gpu suite").

  * config 4's code paths: more right-hand sides than one chunk of the direct kernels holds (nrhs = 5: one
    full chunk + a partial one, nrhs = 8: the config itself) on a mixed-level tree, values, gradients,
    leaves-only evaluation and the device-resident matvec, against the oracle at 1e-11 (reference loops over
    the right-hand sides: bbfmm.rs:720,745,906,1031,1060,1184,1279,1402); the 10M x 8 size through
    size-independent properties;
  * config 3: 3-D thin-plate spline, order 9 (the solver's default for it), linear drift, FGMRES 20 x 5
    with the Schwarz preconditioner (rbf.rs:536-554) against the dense restatement (oracle/ddm.py +
    oracle/solvers.py) and a direct dense solve;
  * config 2: 1M points, Spheroidal3 (reference-native) and the multiquadric extension, mixed-level tree:
    linearity, symmetry and sampled dense rows.
"""
import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points, inject_product_operators, relerr
from oracle import bbfmm_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-11


def _pair(pts, kid, order=7, br=1.0, sill=1.0, params=None, inject=True):
    fp = None if params is None else F.FmmParams(*params)
    op = None if params is None else O.FmmParams(*params)
    t = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, True, params=fp)
    r = O.FmmTree(pts, order, kid, True, True, None, op, base_range=br, total_sill=sill)
    if inject:
        inject_product_operators(t, r)
    return t, r


@pytest.mark.parametrize("nrhs,kid", [(5, 0), (8, 2)])
def test_many_right_hand_sides_on_a_mixed_level_tree(nrhs, kid):
    """DIRECT_KB / P2M_KB chunk loops with k0 > 0, the zero-filled partial chunk (nrhs = 5), M2L with more
    than two rhs: every pass's output against the oracle."""
    import torch
    rng = np.random.default_rng(300 + nrhs)
    n = 130000
    pts = rng.random((n, 3))
    t, r = _pair(pts, kid)
    s = t.stats()
    assert s.n_w > 0 and s.n_x == s.n_w                                        # W / X lists are live
    w = rng.standard_normal((n, nrhs))
    t.set_weights(w)
    r.set_weights(w)
    assert relerr(t.debug_get_coefficients("M", nrhs), r.M) < TOL              # P2M + M2M, all rhs
    y, yr = t.evaluate(w, pts), r.evaluate(w, pts)
    assert relerr(t.debug_get_coefficients("L", nrhs), r.L) < TOL              # M2L + P2L + L2L
    assert relerr(y, yr) < TOL
    for j in range(nrhs):                                                       # no rhs hides behind a bigger one
        assert relerr(y[:, j], yr[:, j]) < TOL
    # gradients (P2P / M2P / L2P gradient accumulators per rhs chunk)
    tg = pts[rng.choice(n, 20000, replace=False)] + 1e-4 * rng.standard_normal((20000, 3))
    tg = np.clip(tg, pts.min(0), pts.max(0))
    yg, g = t.evaluate_with_gradients(w, tg)
    ygr, gr = r.evaluate_with_gradients(w, tg)
    assert g.shape == (20000, 3 * nrhs)
    assert relerr(yg, ygr) < TOL and relerr(g, gr) < 1e-9
    # leaves-only evaluation after set_local_coefficients
    t.set_local_coefficients(w)
    r.set_local_coefficients(w)
    assert relerr(t.evaluate_leaves(w, tg), r.evaluate_leaves(w, tg)) < TOL
    yl, gl = t.evaluate_leaves_with_gradients(w, tg[:5000])
    ylr, glr = r.evaluate_leaves_with_gradients(w, tg[:5000])
    assert relerr(yl, ylr) < TOL and relerr(gl, glr) < 1e-9
    # device-resident matvec (what bench.py times), same weights
    dw = torch.from_numpy(np.ascontiguousarray(w.T)).cuda()
    out = torch.zeros((nrhs, n), dtype=torch.float64, device="cuda")
    t.matvec_device(dw.data_ptr(), n, nrhs, out.data_ptr(), n, True)
    assert relerr(out.cpu().numpy().T, yr) < TOL
    # dense rows
    idx = rng.choice(n, 200, replace=False)
    yd = O.dense_sum(kid, 1.0, 1.0, pts[idx], pts, w)
    assert relerr(y[idx], yd) < 1e-6


def test_config3_tps_order9_linear_drift_fgmres_schwarz_end_to_end():
    """3-D thin-plate spline, p = 9, linear drift: FGMRES(20 x 5) + Schwarz on the device against (1) the dense
    restatement of the same solver + preconditioner and (2) a direct solve of the dense saddle system."""
    from ferreus_rbf_rs_amd import solvers as S
    from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
    from oracle import ddm as D
    from oracle import solvers as OS
    rng = np.random.default_rng(333)
    n, dim, kid, drift = 8000, 3, 1, 1
    pts = rng.random((n, dim))
    prm = (640, 0.5, 0.125, 200)                            # 8000 -> 1024 -> 128 points: two fine levels + coarse
    st = InterpolantSettings(kid, dim, drift=drift)
    ost = D.InterpolantSettings(kid, dim, drift=drift)
    assert st.basis_size == ost.basis_size == 4
    tree = F.FmmTree(pts, 9, F.KernelParams(F.KernelType(kid)), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
    levels = D.build_ddm_tree(pts, ost, D.DDMParams(*prm))
    assert pre.num_levels == len(levels) >= 3
    for lv in range(len(levels)):
        assert np.array_equal(pre.level_points(lv), np.asarray(levels[lv].point_indices))
    # dense system of the restatement
    A = D.a_matrix(pts, ost)
    tr, sc = D.cheb_cube_scaling_factors(pts)
    mono, ortho = D.orthonormal_poly(pts, ost, tr, sc)
    np.testing.assert_allclose(pre.monomial_matrix, mono, rtol=0, atol=1e-14)
    m = ost.basis_size

    def dense_matvec(w):                                    # rbf.rs:1338-1379
        y = np.zeros(n + m)
        y[:n] = A @ w[:n] + mono @ w[n:]
        return y

    def dense_partial(w, idx):
        y = np.zeros(n + m)
        idx = np.asarray(idx)
        y[idx] = A[idx] @ w[:n] + mono[idx] @ w[n:]
        return y

    op = S.RbfSystemOperator(tree, m, pre.monomial_matrix, 0.0)
    wt = rng.standard_normal(n + m)
    yd = dense_matvec(wt)
    assert np.abs(op(wt) - yd).max() < 1e-7 * np.abs(yd).max()             # the operator (p = 9: ~1e-9)
    # one application of the preconditioner against the restatement on the dense partial products
    r0 = rng.standard_normal(n + m)
    r0[n:] = 0.0
    z, zo = pre(r0), D.schwarz_preconditioner(r0, levels, dense_partial, ost, ortho)
    assert np.abs(z - zo).max() < 2e-5 * np.abs(zo).max()
    # the solve, as the solver calls it (rbf.rs:545-554)
    vals = np.sin(4 * pts[:, 0]) * np.cos(3 * pts[:, 1]) + pts[:, 2] ** 2 + 0.5 * pts[:, 0]
    rhs = np.concatenate([vals, np.zeros(m)])
    x, hist = S.fgmres(op, rhs, pre, None, 20, 5, S.FittingAccuracy(1e-6))
    assert hist[-1][1] < 1e-6 and len(hist) <= 30
    pre_o = lambda v: D.schwarz_preconditioner(v, levels, dense_partial, ost, ortho)
    xo, histo = OS.fgmres(dense_matvec, rhs, pre_o, None, 20, 5, OS.RELATIVE, 1e-6)
    assert histo[-1][1] < 1e-6
    assert abs(len(hist) - len(histo)) <= 1                                   # same convergence
    k = min(len(hist), len(histo)) - 1
    np.testing.assert_allclose([h[1] for h in hist[:k]], [h[1] for h in histo[:k]], rtol=0.05)
    # both reach the same residual of the dense system, and the interpolant reproduces the data
    res = np.linalg.norm(dense_matvec(x) - rhs) / np.linalg.norm(rhs)
    reso = np.linalg.norm(dense_matvec(xo) - rhs) / np.linalg.norm(rhs)
    assert res < 2e-6 and reso < 2e-6
    assert np.abs(dense_matvec(x)[:n] - vals).max() < 1e-4 * np.abs(vals).max()
    # direct solve of [A P; P^T 0][lambda; c] = [f; 0] (the system FGMRES approximates, in the orthogonal
    # complement of the polynomials): same fitted surface away from the data
    Ks = np.block([[A, mono], [mono.T, np.zeros((m, m))]])
    xs = np.linalg.solve(Ks, rhs)
    q = rng.random((500, 3))
    Pq = D.evaluate_monomials(q, ost.polynomial_degree, m, tr, sc)
    Phi = np.array(O.kernel_matrix(kid, 1.0, 1.0, q, pts))
    fs = Phi @ xs[:n] + Pq @ xs[n:]
    # FGMRES keeps lambda orthogonal to the polynomials only approximately and returns the tail from the
    # coarse solves; evaluate with its own coefficients
    ff = Phi @ x[:n] + Pq @ x[n:]
    assert np.abs(ff - fs).max() < 1e-3 * np.abs(fs).max()
    # A hierarchy whose coarse domain is too small for its level-0 domains (128 coarse points for 128 domains;
    # DESIGN.md section 9, "hierarchy depth"): the sweep of schwarz.rs then stagnates -- in the restatement and on
    # the device alike, at the same residuals.  (This is what the reference's default coarse_threshold does to
    # thin-plate-spline problems above ~2M points.)
    prm2 = (160, 0.5, 0.125, 600)
    pre2 = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm2))
    levels2 = D.build_ddm_tree(pts, ost, D.DDMParams(*prm2))
    assert pre2.num_levels == len(levels2) == 3 and len(levels2[0].leaf_domains) >= 100
    x2, hist2 = S.fgmres(op, rhs, pre2, None, 3, 5, S.FittingAccuracy(1e-6))
    pre2_o = lambda v: D.schwarz_preconditioner(v, levels2, dense_partial, ost, ortho)
    x2o, hist2o = OS.fgmres(dense_matvec, rhs, pre2_o, None, 3, 5, OS.RELATIVE, 1e-6)
    assert len(hist2) == len(hist2o) == 15 and hist2o[-1][1] > 1e-2          # stagnation, not convergence
    np.testing.assert_allclose([h[1] for h in hist2], [h[1] for h in hist2o], rtol=0.05)


@pytest.mark.parametrize("kid,br,sill", [(3, 0.1, 0.1), (101, 0.1, 0.1)])
def test_config2_one_million_points_mixed_level_tree(kid, br, sill):
    """BASELINE.json configs[1] (Spheroidal3 base_range 0.1 / sill 0.1, and the labelled
    multiquadric extension): 1M uniform points, whose adaptive tree mixes levels 4 and 5 (W / X lists live)."""
    import torch
    n = 1_000_000
    pts = np.random.default_rng(42).random((n, 3))
    t = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, True)
    s = t.stats()
    assert s.n_points == n and s.n_w > 0 and s.n_x == s.n_w
    g = torch.Generator(device="cuda").manual_seed(7)
    w = torch.rand((2, n), dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.zeros_like(w)
    t.matvec_device(w.data_ptr(), n, 2, y.data_ptr(), n, True)
    a, b = 1.25, -0.5
    wc = (a * w[0] + b * w[1]).reshape(1, n).contiguous()
    yc = torch.zeros_like(wc)
    t.matvec_device(wc.data_ptr(), n, 1, yc.data_ptr(), n, True)
    lin = (yc[0] - (a * y[0] + b * y[1])).abs().max() / yc.abs().max()
    assert float(lin) < 1e-12
    # K = K^T up to the far-field approximation (order 7 on these short-range kernels: ~1e-6, the dense rows below)
    sym = abs(float(torch.dot(w[1], y[0]) - torch.dot(w[0], y[1]))) / float(y[0].norm() * w[1].norm())
    assert sym < 2e-5
    idx = np.random.default_rng(3).choice(n, 64, replace=False)
    wh = w.cpu().numpy().T.copy()
    yd = O.dense_sum(kid, br, sill, pts[idx], pts, wh)
    assert relerr(y.cpu().numpy().T[idx], yd) < 5e-6
    # host-buffer entry points on the same tree: same numbers as the device-resident call
    yh = t.fast_matrix_vector_product(wh[:, 0].copy())
    assert relerr(yh, y[0].cpu().numpy()) < 1e-13


def test_config4_ten_million_points_eight_rhs_properties():
    """BASELINE.json configs[3] at full size: 10M points, 8 right-hand sides in one batched matvec."""
    import torch
    n, k = 10_000_000, 8
    pts = np.random.default_rng(42).random((n, 3))
    t = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
    g = torch.Generator(device="cuda").manual_seed(11)
    w = torch.rand((k, n), dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.zeros_like(w)
    t.matvec_device(w.data_ptr(), n, k, y.data_ptr(), n, True)
    # linearity: a random combination of the 8 columns as a single-rhs product
    coef = torch.tensor([0.5, -1.0, 2.0, 0.25, -0.75, 1.5, -2.0, 1.0], dtype=torch.float64, device="cuda")
    wc = (coef[:, None] * w).sum(0, keepdim=True).contiguous()
    yc = torch.zeros_like(wc)
    t.matvec_device(wc.data_ptr(), n, 1, yc.data_ptr(), n, True)
    lin = (yc[0] - (coef[:, None] * y).sum(0)).abs().max() / yc.abs().max()
    assert float(lin) < 1e-12
    # every column equals its own single-rhs product (the batch does not mix columns)
    for j in (0, 4, 7):
        wj = w[j:j + 1].contiguous()
        yj = torch.zeros_like(wj)
        t.matvec_device(wj.data_ptr(), n, 1, yj.data_ptr(), n, True)
        assert float((yj[0] - y[j]).abs().max() / yj.abs().max()) < 1e-13
    # symmetry of the kernel matrix across two columns, sampled dense rows for all eight
    sym = abs(float(torch.dot(w[5], y[2]) - torch.dot(w[2], y[5]))) / float(y[2].norm() * w[5].norm())
    assert sym < 1e-7
    idx = np.random.default_rng(5).choice(n, 32, replace=False)
    yd = O.dense_sum(0, 1.0, 1.0, pts[idx], pts, w.cpu().numpy().T.copy())
    assert relerr(y.cpu().numpy().T[idx], yd) < 1e-6


@pytest.mark.parametrize("kid,order,br,sill,tol", [(1, 9, 1.0, 1.0, 1e-7), (3, 7, 0.4, 0.3, 5e-6)])
def test_independent_operators_on_a_mixed_level_tree(kid, order, br, sill, tol):
    """No operator injection: the oracle builds its own ACA + LAPACK operators, the product its own ACA +
    Householder + Jacobi ones; they agree to the compression tolerance (thin-plate spline p = 9 = config 3's
    operator; Spheroidal3 = config 2's kernel)."""
    rng = np.random.default_rng(400 + kid)
    n = 130000 if order == 7 else 60000
    pts = rng.random((n, 3)) if order == 7 else clustered_points(rng, n, 3)
    t, r = _pair(pts, kid, order=order, br=br, sill=sill, inject=False)
    assert t.stats().n_w > 0
    w = rng.standard_normal((n, 1))
    t.set_weights(w)
    r.set_weights(w)
    y, yr = t.evaluate(w, pts), r.evaluate(w, pts)
    assert relerr(y, yr) < tol
    idx = rng.choice(n, 200, replace=False)
    yd = O.dense_sum(kid, br, sill, pts[idx], pts, w)
    assert relerr(y[idx], yd) < 10 * tol and relerr(yr[idx], yd) < 10 * tol


@pytest.mark.parametrize("kid,d,mpc,nrhs", [(0, 3, 256, 1), (1, 3, 256, 2), (3, 3, 20, 1), (7, 3, 256, 3), (2, 2, 256, 1),
                                           (0, 1, 64, 1), (0, 3, 256, 5), (0, 3, 40, 8), (3, 3, 256, 8), (1, 3, 64, 11),
                                           (2, 2, 30, 4)])
@pytest.mark.parametrize("jobs", ["wave_per_small_leaf", "size_rule"])
def test_symmetric_p2p_equals_the_ordered_pair_loops(kid, d, mpc, nrhs, jobs, monkeypatch):
    """The matvec evaluates every unordered near-field pair once (launch_p2p_sym); `evaluate` at the same points
    goes through the ordered-pair kernel the reference's loops correspond to (bbfmm.rs:1162-1251).  Leaves of up
    to 256 points (several register groups per wave, several source tiles per leaf) and of a few points.  Round 4:
    up to four right-hand sides share one kernel evaluation per unordered pair (kernel instances for 1, 2, 4; three runs
    the 4-slot instance; eleven = 4 + 4 + 3), in the near field and in the fused M2P + P2L."""
    import torch
    # leaves of at most 64 points: one wave each (what a tree with enough of them gets: conftest.py), or -- the library's own
    # choice for trees this small -- workgroup jobs like the bigger leaves (BBFMM_P2P_SYM_WAVE_MIN is read when a plan is built)
    if jobs == "size_rule":
        monkeypatch.delenv("BBFMM_P2P_SYM_WAVE_MIN", raising=False)
    else:
        monkeypatch.setenv("BBFMM_P2P_SYM_WAVE_MIN", "0")
    rng = np.random.default_rng(600 + kid + d)
    n = 150000 if d == 3 else 40000
    pts = clustered_points(rng, n, d)
    pts = np.unique(pts, axis=0)
    n = pts.shape[0]
    params = (mpc, O.COMPRESSION_ACA, 1e-7, 1024)
    t = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(kid), base_range=0.5, total_sill=0.4), True, True,
                  params=F.FmmParams(*params))
    r = O.FmmTree(pts, 7, kid, True, True, None, O.FmmParams(*params), base_range=0.5, total_sill=0.4)
    inject_product_operators(t, r)
    w = rng.standard_normal((n, nrhs))
    dw = torch.from_numpy(np.ascontiguousarray(w.T)).cuda()
    out = torch.zeros((nrhs, n), dtype=torch.float64, device="cuda")
    t.matvec_device(dw.data_ptr(), n, nrhs, out.data_ptr(), n, True)
    y_sym = out.cpu().numpy().T
    L_sym = t.debug_get_coefficients("L", nrhs)          # P2L ran fused with M2P (X = W^T), atomically into L
    t.set_weights(w)
    y_ord = t.evaluate(w, pts)
    r.set_weights(w)
    yr = r.evaluate(w, pts)
    assert relerr(y_sym, y_ord) < 1e-12
    assert relerr(y_sym, yr) < TOL
    assert relerr(L_sym, r.L) < TOL
    if d == 3 and kid != 7:
        assert t.stats().n_w > 0                          # the W / X lists are live in these trees
    # a 3-way partition of the same product: one- and two-sided runs at the partition boundaries
    acc = torch.full((nrhs, n), float("nan"), dtype=torch.float64, device="cuda")
    for rank in range(3):
        t.set_partition(rank, 3)
        rows = torch.from_numpy(t.partition_rows()).cuda()
        tmp = torch.zeros((nrhs, n), dtype=torch.float64, device="cuda")
        t.matvec_device(dw.data_ptr(), n, nrhs, tmp.data_ptr(), n, True)
        acc[:, rows] = tmp[:, rows]
    t.set_partition(0, 1)
    assert relerr(acc.cpu().numpy().T, yr) < TOL


def test_config5_forty_million_points_eight_way_partition():
    """BASELINE.json configs[4] at full size on one GPU: 40M points, Spheroidal3, the matvec
    partitioned by target subtree into 8 shares that are run one after another and reassembled from their owned
    rows -- everything of the 8-GPU run (own-subtree upward pass, summed coarse multipoles, restricted downward and
    leaf passes) except the two RCCL collectives themselves (tests/test_gpu_two_ranks.py runs those with two real
    processes)."""
    import torch
    n, br, sill = 40_000_000, 0.1, 0.1
    pts = np.random.default_rng(42).random((n, 3))
    t = F.FmmTree(pts, 7, F.KernelParams(F.KernelType.Spheroidal3Rbf, base_range=br, total_sill=sill), True, True)
    assert t.tree_built_on_device() and t.stats().n_points == n
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.rand((1, n), dtype=torch.float64, device="cuda", generator=g) - 0.5
    ref = torch.zeros_like(w)
    t.matvec_device(w.data_ptr(), n, 1, ref.data_ptr(), n, True)
    acc = torch.full((1, n), float("nan"), dtype=torch.float64, device="cuda")
    tmp = torch.zeros_like(w)
    owned = 0
    # the split upward pass: every share anterpolates its own subtree (+ halo); the sum of the partial coarse
    # multipoles is what the all-reduce delivers
    total = None
    for rank in range(8):
        t.set_partition(rank, 8)
        c = torch.zeros((1, t.partition_coarse_count()), dtype=torch.float64, device="cuda")
        t.matvec_partition_upward(w.data_ptr(), n, 1, c.data_ptr())
        torch.cuda.synchronize()
        total = c if total is None else total + c
    assert total.numel() == 4681 * 352                                  # levels 0..4 of the uniform tree, n_pad = 352
    scratch = torch.zeros_like(total)
    for rank in range(8):
        t.set_partition(rank, 8)
        rows = torch.from_numpy(t.partition_rows()).cuda()
        assert 0.08 * n < rows.numel() < 0.18 * n                       # balanced shares
        tmp.zero_()
        t.matvec_partition_upward(w.data_ptr(), n, 1, scratch.data_ptr())
        t.matvec_partition_finish(total.data_ptr(), tmp.data_ptr(), n, True)
        acc[:, rows] = tmp[:, rows]
        owned += rows.numel()
        del rows
    t.set_partition(0, 1)
    assert owned == n and not bool(torch.isnan(acc).any())
    assert float((acc - ref).abs().max() / ref.abs().max()) < 1e-12     # the shares reassemble the product
    idx = np.random.default_rng(6).choice(n, 16, replace=False)
    yd = O.dense_sum(3, br, sill, pts[idx], pts, w.cpu().numpy().T.copy())
    assert relerr(ref.cpu().numpy().T[idx], yd) < 5e-6


@pytest.mark.timeout(1800)
def test_eighty_million_points_depth_seven_tree():
    """A size the unbounded intermediate could not hold: 80M uniform points give a depth-7 tree whose finest level
    alone has 77 GB of M2L slots per right-hand side (and slot indices beyond 2^31 in one buffer).  Bounded, the
    level goes through the buffer in 8 groups of target classes.  Checked through dense rows, linearity and the
    symmetry of the kernel matrix (size-independent properties; the oracle does not reach this size in seconds)."""
    import torch
    n = 80_000_000
    pts = np.random.default_rng(42).random((n, 3))
    t = F.FmmTree(pts, 7, F.KernelParams(F.KernelType.LinearRbf), True, True)
    st = t.stats()
    assert st.depth == 7 and t.tree_built_on_device()
    assert st.m2l_slots_bytes_per_rhs > 80e9 and st.m2l_batches >= 9
    g = torch.Generator(device="cuda").manual_seed(8)
    w = torch.rand((2, n), dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.zeros_like(w)
    t.matvec_device(w.data_ptr(), n, 1, y[0].data_ptr(), n, True)
    assert t.stats().m2l_intermediate_bytes < 20e9                         # the budget (a sixteenth of the device memory), not 87 GB
    t.matvec_device(w[1].data_ptr(), n, 1, y[1].data_ptr(), n, True)
    a, b = 0.75, -1.25
    wc = (a * w[0] + b * w[1]).contiguous()
    yc = torch.zeros_like(wc)
    t.matvec_device(wc.data_ptr(), n, 1, yc.data_ptr(), n, True)
    assert float((yc - (a * y[0] + b * y[1])).abs().max() / yc.abs().max()) < 1e-11          # linearity
    sym = abs(float(torch.dot(w[1], y[0]) - torch.dot(w[0], y[1]))) / abs(float(torch.dot(w[1], y[0])))
    assert sym < 1e-6                                                                      # K = K^T up to the far field
    idx = np.random.default_rng(9).choice(n, 32, replace=False)
    pd = torch.from_numpy(pts).cuda()
    r2 = torch.zeros((32, n), dtype=torch.float64, device="cuda")
    for ax in range(3):
        r2 += (pd[idx, ax, None] - pd[None, :, ax]) ** 2
    yd = -(torch.sqrt(r2) * w[0]).sum(1)
    got = y[0][torch.from_numpy(idx).cuda()]
    assert float((got - yd).abs().max() / yd.abs().max()) < 1e-6


def test_the_evaluator_refuses_a_partition_s_share_of_the_multipoles():
    """After bbfmm_matvec_partition_upward a handle's multipoles are its own subtree's share (+ the coarse prefix): evaluate and
    set_local_coefficients say so instead of reading them as if they were whole; set_weights makes them whole again.  (A device
    group completes its first part's multipoles by itself: test_gpu_device_group.py.)"""
    import torch
    rng = np.random.default_rng(77)
    pts = rng.random((60000, 3))
    n = len(pts)
    t = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(0)), True, True)
    w = rng.standard_normal((n, 1))
    dw = torch.from_numpy(np.ascontiguousarray(w.T)).cuda()
    x = rng.random((50, 3))
    t.set_weights(w)
    y0 = t.evaluate(w, x)
    t.set_partition(0, 2)
    c = torch.zeros((1, max(t.partition_coarse_count(), 1)), dtype=torch.float64, device="cuda")
    t.matvec_partition_upward(dw.data_ptr(), n, 1, c.data_ptr())
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="partition's share"):
        t.evaluate(w, x)
    with pytest.raises(ValueError, match="partition's share"):
        t.set_local_coefficients(w)
    t.set_partition(0, 1)
    with pytest.raises(ValueError, match="partition's share"):
        t.evaluate(w, x)
    t.set_weights(w)
    assert relerr(t.evaluate(w, x), y0) < 1e-13
