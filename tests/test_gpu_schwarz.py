"""The Schwarz preconditioner with device-batched local solves against the numpy
restatement of schwarz.rs / domain.rs (oracle/ddm.py, dense system) and end to end inside FGMRES."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
from oracle import ddm as D
from oracle import solvers as OS


def _dense_partial(pts, st):
    n = pts.shape[0]
    A = D.a_matrix(pts, st)
    tr, sc = D.cheb_cube_scaling_factors(pts)
    P = D.evaluate_monomials(pts, st.polynomial_degree, st.basis_size, tr, sc) if st.basis_size else None
    m = st.basis_size

    def partial(w, idx):
        y = np.zeros(n + m)
        idx = np.asarray(idx)
        y[idx] = A[idx] @ w[:n] + (P[idx] @ w[n:] if m else 0.0)
        return y
    return A, P, partial


CASES = [  # kernel id, dims, drift, nugget, base_range/sill
    (0, 3, None, 0.0, 1.0),
    (0, 3, 1, 0.0, 1.0),
    (1, 2, None, 0.0, 1.0),
    (2, 3, None, 0.0, 1.0),
    (3, 3, None, 0.02, 0.3),
]


@pytest.mark.parametrize("kid,dim,drift,nugget,rng_", CASES)
def test_apply_matches_the_restatement(kid, dim, drift, nugget, rng_):
    rng = np.random.default_rng(50 + kid + dim)
    n = 2500
    pts = rng.random((n, dim))
    prm = DDMParams(80, 0.5, 0.125, 250)
    st = InterpolantSettings(kid, dim, drift=drift, nugget=nugget, base_range=rng_, total_sill=rng_)
    tree = F.FmmTree(pts, 8, F.KernelParams(F.KernelType(kid), base_range=rng_, total_sill=rng_), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, prm)
    ost = D.InterpolantSettings(kid, dim, drift=drift, nugget=nugget, base_range=rng_, total_sill=rng_)
    assert pre.basis_size == ost.basis_size
    levels = D.build_ddm_tree(pts, ost, D.DDMParams(80, 0.5, 0.125, 250))
    assert pre.num_levels == len(levels) >= 2
    A, P, partial = _dense_partial(pts, ost)
    ortho = None
    if ost.basis_size:
        tr, sc = D.cheb_cube_scaling_factors(pts)
        mono, ortho = D.orthonormal_poly(pts, ost, tr, sc)
        np.testing.assert_allclose(pre.monomial_matrix, mono, rtol=0, atol=1e-14)
    r = rng.standard_normal(n + ost.basis_size)
    r[n:] = 0.0
    z = pre(r)
    zo = D.schwarz_preconditioner(r, levels, partial, ost, ortho)
    # the FMM matvec inside (order 8, ~1e-8) is the only difference between the two; the cubic kernel's
    # local systems (condition ~1e9) amplify it most
    assert np.abs(z - zo).max() < (1e-3 if kid == 2 else 2e-5) * np.abs(zo).max()


@pytest.mark.parametrize("kid,dim,drift,nugget,rng_", [CASES[0], CASES[2], CASES[4]])
def test_fgmres_with_schwarz_preconditioner_solves_the_interpolation_problem(kid, dim, drift, nugget, rng_):
    rng = np.random.default_rng(70 + kid)
    n = 6000
    pts = rng.random((n, dim))
    st = InterpolantSettings(kid, dim, drift=drift, nugget=nugget, base_range=rng_, total_sill=rng_)
    tree = F.FmmTree(pts, 8, F.KernelParams(F.KernelType(kid), base_range=rng_, total_sill=rng_), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(128, 0.5, 0.125, 512))
    op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, nugget)
    vals = np.sin(4 * pts[:, 0]) * np.cos(3 * pts[:, -1]) + pts[:, 0]
    rhs = np.concatenate([vals, np.zeros(st.basis_size)])
    x, hist = S.fgmres(op, rhs, pre, None, 20, 5, S.FittingAccuracy(1e-6))        # the solver's call, rbf.rs:545-554
    assert hist[-1][1] < 1e-6 and len(hist) <= 40
    x0, hist0 = S.fgmres(op, rhs, None, None, 4, 5, S.FittingAccuracy(1e-6))
    assert hist0[-1][1] > 10 * hist[min(len(hist), len(hist0)) - 1][1]           # the preconditioner is why
    fitted = op(x)[:n]
    assert np.abs(fitted - vals).max() < 1e-4 * np.abs(vals).max()


def _oracle_level_solve(levels, lv, r, n, ost, ortho):
    """one level's correction as schwarz.rs computes it (solve_fine_level / solve_coarse_level)"""
    if lv < len(levels) - 1:
        s1 = np.zeros_like(r)
        for dom in levels[lv].leaf_domains:
            coef, _ = dom.solve(r[:, None])
            for local, (g, m) in enumerate(zip(dom.overlapping_point_indices, dom.internal_points_mask)):
                if m:
                    s1[g] = coef[local, 0]
        if ost.basis_size:
            s1[:n] -= ortho @ (ortho.T @ s1[:n])
        return s1
    dom = levels[lv].leaf_domains[0]
    coef, poly = dom.solve(r[:, None])
    s1 = np.zeros_like(r)
    s1[np.asarray(dom.overlapping_point_indices)] = coef[:, 0]
    if poly is not None:
        s1[n:] = poly[:, 0]
    return s1


@pytest.mark.parametrize("kid,drift", [(3, None), (3, 1)])
def test_local_systems_that_are_not_positive_definite_take_the_fallback(kid, drift):
    """DomainSolver::new (domain.rs:60-68): when the Cholesky factorisation of a domain fails the reference
    solves that domain with a Bunch-Kaufman LBL^T factorisation; here the host inverts such a system and
    the device multiplies by the inverse.  A negative nugget makes the local systems indefinite."""
    rng = np.random.default_rng(77)
    n, dim, nugget, br = 2000, 3, -0.02, 0.3
    pts = rng.random((n, dim))
    prm = (80, 0.5, 0.125, 250)
    st = InterpolantSettings(kid, dim, drift=drift, nugget=nugget, base_range=br, total_sill=br)
    ost = D.InterpolantSettings(kid, dim, drift=drift, nugget=nugget, base_range=br, total_sill=br)
    tree = F.FmmTree(pts, 6, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=br), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
    levels = D.build_ddm_tree(pts, ost, D.DDMParams(*prm))
    n_indef = sum(dom.indefinite is not None for lvl in levels for dom in lvl.leaf_domains)
    assert n_indef > 0                                   # the case really exercises the fallback
    ortho = None
    if ost.basis_size:
        tr, sc = D.cheb_cube_scaling_factors(pts)
        _, ortho = D.orthonormal_poly(pts, ost, tr, sc)
    r = rng.standard_normal(n + ost.basis_size)
    r[n:] = 0.0
    for lv in range(len(levels)):
        z = pre.debug_level_solve(lv, r, True)
        zo = _oracle_level_solve(levels, lv, r, n, ost, ortho)
        assert np.abs(z - zo).max() < 1e-7 * np.abs(zo).max(), f"level {lv}"


@pytest.mark.parametrize("kid", [1, 0])
def test_reference_sized_domains_and_a_large_coarse_domain(kid):
    """The reference's default leaf size (1024 -> here domains of 750 points: the blocked MFMA Cholesky runs a
    dozen block columns) and a coarse domain of more than 2,048 points (factorised and solved as multi-workgroup
    launch sequences): every level's correction against the restatement, no FMM in between."""
    rng = np.random.default_rng(5 + kid)
    n, dim = 24000, 3
    pts = rng.random((n, dim))
    prm = (1024, 0.5, 0.125, 3200)
    st = InterpolantSettings(kid, dim)
    ost = D.InterpolantSettings(kid, dim)
    tree = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(kid)), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
    levels = D.build_ddm_tree(pts, ost, D.DDMParams(*prm))
    assert pre.num_levels == len(levels) == 2
    assert len(levels[1].point_indices) > 2048 and len(levels[0].leaf_domains[0].overlapping_point_indices) > 700
    tr, sc = D.cheb_cube_scaling_factors(pts)
    _, ortho = D.orthonormal_poly(pts, ost, tr, sc)
    r = rng.standard_normal(n + ost.basis_size)
    r[n:] = 0.0
    for lv in range(2):
        z = pre.debug_level_solve(lv, r, True)
        zo = _oracle_level_solve(levels, lv, r, n, ost, ortho)
        assert np.abs(z - zo).max() < 1e-8 * np.abs(zo).max(), f"level {lv}"


@pytest.mark.parametrize("n", [16600, 20500, 24900, 33500])
def test_large_coarse_domain_block_substitutions_at_block_edges(n):
    """The one large coarse domain is solved in blocks of 1,024 columns with the diagonal blocks' inverses
    (ddm_kernels.hip).  Coarse sizes 2,080 (a third block of 32 columns), 2,592 (mid-block), 3,136 and 4,224 (short fourth /
    fifth blocks) against numpy's solve of the same domain."""
    rng = np.random.default_rng(n)
    dim, kid = 3, 0
    pts = rng.random((n, dim))
    prm = (1024, 0.5, 0.125, 6000)
    st = InterpolantSettings(kid, dim)
    ost = D.InterpolantSettings(kid, dim)
    tree = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(kid)), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
    levels = D.build_ddm_tree(pts, ost, D.DDMParams(*prm))
    assert pre.num_levels == len(levels) == 2
    m = len(levels[1].point_indices)
    assert m > 2048, m
    tr, sc = D.cheb_cube_scaling_factors(pts)
    _, ortho = D.orthonormal_poly(pts, ost, tr, sc)
    r = rng.standard_normal(n + ost.basis_size)
    r[n:] = 0.0
    z = pre.debug_level_solve(1, r, True)
    zo = _oracle_level_solve(levels, 1, r, n, ost, ortho)
    assert np.abs(z - zo).max() < 1e-8 * np.abs(zo).max(), (n, m)


def test_large_coarse_domain_that_is_not_positive_definite_takes_the_pivoted_lu():
    """domain.rs:60-68 for the one large coarse domain (> 2,048 points, factorised and solved as launch sequences
    over the whole chip): a negative nugget makes Q^T A Q indefinite, the blocked Cholesky reports the failure and
    the level is re-assembled and factorised by pivoted LU (rocSOLVER, bound at run time) -- the role of the
    reference's Bunch-Kaufman LBL^T.  Every level's correction against the restatement."""
    rng = np.random.default_rng(91)
    n, dim, kid, nugget, br = 20000, 3, 3, -0.02, 0.3
    pts = rng.random((n, dim))
    prm = (1024, 0.5, 0.125, 2600)
    st = InterpolantSettings(kid, dim, nugget=nugget, base_range=br, total_sill=br)
    ost = D.InterpolantSettings(kid, dim, nugget=nugget, base_range=br, total_sill=br)
    tree = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=br), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
    levels = D.build_ddm_tree(pts, ost, D.DDMParams(*prm))
    assert pre.num_levels == len(levels) == 2 and len(levels[1].point_indices) > 2048
    assert levels[1].leaf_domains[0].indefinite is not None          # the coarse system really is indefinite
    r = rng.standard_normal(n)
    for lv in range(2):
        z = pre.debug_level_solve(lv, r, True)
        zo = _oracle_level_solve(levels, lv, r, n, ost, None)
        assert np.abs(z - zo).max() < 1e-7 * np.abs(zo).max(), f"level {lv}"


def test_the_factors_are_released_with_the_last_reference():
    """Round 5: the preconditioner's operator object used to hold the preconditioner (a reference cycle), so `del pre`
    kept the factors -- 95 GB of HBM at 10M points -- until the cyclic collector ran; a sweep over hierarchies at 10M
    points ran out of device memory on its third preconditioner (tests/checks/ddm_depth_sweep_10M.py)."""
    import gc
    import weakref
    import torch
    from ferreus_rbf_rs_amd import solvers as S
    rng = np.random.default_rng(3)
    pts = rng.random((60000, 3))
    tree = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(0)), True, True)
    st = InterpolantSettings(0, 3)
    gc.disable()
    try:
        free0 = torch.cuda.mem_get_info()[0]
        pre = SchwarzPreconditioner(tree, pts, st, DDMParams())
        op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, 0.0)
        rhs = np.concatenate([np.sin(pts[:, 0]), np.zeros(st.basis_size)])
        S.fgmres(op, rhs, pre, None, 1, 3, S.FittingAccuracy(1e-6))
        held = free0 - torch.cuda.mem_get_info()[0]
        assert held > 50e6                                        # the factors of ~100 domains of ~1,200 points
        ref = weakref.ref(pre)
        del pre
        assert ref() is None                                      # no collector pass needed
        assert free0 - torch.cuda.mem_get_info()[0] < 0.2 * held  # and the device memory is back
    finally:
        gc.enable()
