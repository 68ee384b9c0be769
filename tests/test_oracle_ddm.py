"""The numpy restatement of the reference's Schwarz / domain-decomposition preconditioner
(oracle/ddm.py) against what the reference's own tests assert for those files and
against dense solves.  Test infrastructure for the next row of SURVEY 8(f); no GPU."""
import numpy as np
import pytest

from oracle import bbfmm_oracle as O
from oracle import ddm as D
from oracle import solvers as OS


def _system(pts, st):
    """dense RBF system operator as the reference's matvec / matvec_partial (rbf.rs:105-133, 1338-1379)"""
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


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_hierarchy_invariants(dim):
    """domain_decomposition.rs:378-596: union(internal) == level points, coarse points come from the
    level, overlap never marked internal, every level shrinks"""
    pts = np.random.default_rng(42).random((1200, dim))
    st = D.InterpolantSettings(3, dim, nugget=1e-6, base_range=0.2, total_sill=0.2)   # spheroidal, no polynomial
    levels = D.build_ddm_tree(pts, st, D.DDMParams(leaf_threshold=40, overlap_quota=0.5, coarse_ratio=0.25,
                                                   coarse_threshold=100))
    assert len(levels) >= 2 and len(levels[-1].leaf_domains) == 1
    for li, lvl in enumerate(levels):
        union = sorted(g for dom in lvl.leaf_domains for g in dom.internal_indices())
        assert union == sorted(lvl.point_indices), f"level {li}"
        for dom in lvl.leaf_domains:
            k = len(dom.overlapping_point_indices)
            assert len(set(dom.overlapping_point_indices)) == k
            assert len(dom.internal_indices()) <= 40 + 40                     # leaves below the threshold
        if li + 1 < len(levels):
            assert set(levels[li + 1].point_indices) <= set(lvl.point_indices)
            assert len(levels[li + 1].point_indices) < len(lvl.point_indices)
    assert len(levels[-1].point_indices) <= 100


@pytest.mark.parametrize("kid,drift", [(0, 0), (0, 1), (1, 1), (2, 1), (2, 2), (3, None)])
def test_domain_solve_matches_naive_augmented_solve(kid, drift):
    """domain.rs:732-763: the Q-formulation solve == the saddle-point solve [A P; P^T 0]"""
    rng = np.random.default_rng(7)
    pts = rng.random((150, 3))
    st = D.InterpolantSettings(kid, 3, drift=drift, nugget=0.0 if kid != 3 else 0.01)
    dom = D.Domain(range(150))
    dom.internal_points_mask = [True] * 150
    dom.factorise(pts, st, st.basis_size != 0)
    vals = rng.standard_normal((150, 2))
    coef, poly = dom.solve(vals)
    lam = np.zeros((150, 2))
    lam[np.asarray(dom.overlapping_point_indices)] = coef
    A = D.a_matrix(pts, st)
    if st.basis_size:
        tr, sc = D.cheb_cube_scaling_factors(pts)
        P = D.evaluate_monomials(pts, st.polynomial_degree, st.basis_size, tr, sc)
        m = st.basis_size
        K = np.block([[A, P], [P.T, np.zeros((m, m))]])
        ref = np.linalg.solve(K, np.vstack([vals, np.zeros((m, 2))]))
        np.testing.assert_allclose(lam, ref[:150], rtol=1e-7, atol=1e-7 * np.abs(ref).max())
        np.testing.assert_allclose(poly, ref[150:], rtol=1e-6, atol=1e-7 * np.abs(ref).max())
        assert np.abs(P.T @ lam).max() < 1e-8 * np.abs(lam).max()            # side condition
    else:
        np.testing.assert_allclose(lam, np.linalg.solve(A, vals), rtol=1e-8, atol=1e-10)
        assert poly is None


def test_non_unisolvent_points_on_a_plane():
    """domain.rs:718-730: all points in one plane: the linear basis loses a column, the solve stands"""
    rng = np.random.default_rng(8)
    pts = np.column_stack([rng.random(80), rng.random(80), np.full(80, 0.3)])
    st = D.InterpolantSettings(1, 3)                          # thin-plate spline, linear drift
    dom = D.Domain(range(80))
    dom.internal_points_mask = [True] * 80
    dom.factorise(pts, st, True)
    assert dom.n_special == 3                                 # 1, x, y
    vals = rng.standard_normal((80, 1))
    coef, poly = dom.solve(vals)
    lam = np.zeros(80)
    lam[np.asarray(dom.overlapping_point_indices)] = coef[:, 0]
    A = D.a_matrix(pts, st)
    tr, sc = D.cheb_cube_scaling_factors(pts)
    P = D.evaluate_monomials(pts, 1, 4, tr, sc)[:, [0, 1, 2]]
    assert np.abs(A @ lam + P @ poly[:, 0] - vals[:, 0]).max() < 1e-8


@pytest.mark.parametrize("kid,dim,nugget,tol", [(0, 3, 0.0, 1e-8), (1, 2, 0.0, 1e-8), (1, 3, 0.0, 1e-6),
                                                (2, 3, 0.0, 1e-3), (3, 3, 0.02, 1e-8)])
def test_schwarz_preconditioned_fgmres_converges_to_the_dense_solution(kid, dim, nugget, tol):
    # (60-point leaves and 192 coarse points: the smoother kernels in 3-D converge more slowly than with
    # the reference's defaults of 1024 / 4096, hence their looser targets)
    rng = np.random.default_rng(11 + kid)
    n = 1500
    pts = rng.random((n, dim))
    st = D.InterpolantSettings(kid, dim, nugget=nugget, base_range=0.3, total_sill=0.3)
    A, P, partial = _system(pts, st)
    m = st.basis_size
    levels = D.build_ddm_tree(pts, st, D.DDMParams(leaf_threshold=60, overlap_quota=0.5, coarse_ratio=0.125,
                                                   coarse_threshold=200))
    assert len(levels) >= 2
    ortho = None
    if m:
        tr, sc = D.cheb_cube_scaling_factors(pts)
        _, ortho = D.orthonormal_poly(pts, st, tr, sc)
    vals = np.sin(5 * pts[:, 0]) + pts[:, -1] ** 2
    rhs = np.concatenate([vals, np.zeros(m)])
    matvec = lambda w: partial(w, np.arange(n))
    precon = lambda r: D.schwarz_preconditioner(r, levels, partial, st, ortho)
    x, hist = OS.fgmres(matvec, rhs, precon, None, 20, 5, OS.RELATIVE, tol)
    x0, hist0 = OS.fgmres(matvec, rhs, None, None, 20, 5, OS.RELATIVE, tol)
    assert hist[-1][1] < tol and len(hist) <= 60                              # converges in a few cycles
    assert len(hist) < len(hist0) or hist0[-1][1] > tol                       # and the preconditioner is why
    fitted = A @ x[:n] + (P @ x[n:] if m else 0.0)
    assert np.abs(fitted - vals).max() < 1e3 * tol * max(1.0, np.abs(vals).max())
    if m:
        assert np.abs(P.T @ x[:n]).max() < 1e-6 * np.abs(x[:n]).max()


def test_indefinite_local_system_takes_the_symmetric_indefinite_solver():
    """DomainSolver::new (domain.rs:60-68): a failed Cholesky falls back to the LBL^T solver; the domain
    solve still equals the naive solve of the same (now indefinite) system."""
    rng = np.random.default_rng(17)
    pts = rng.random((120, 3))
    st = D.InterpolantSettings(3, 3, nugget=-0.05, base_range=0.3, total_sill=0.3)
    dom = D.Domain(range(120))
    dom.internal_points_mask = [True] * 120
    dom.factorise(pts, st, False)
    assert dom.indefinite is not None and dom.chol is None
    vals = rng.standard_normal((120, 1))
    coef, poly = dom.solve(vals)
    lam = np.zeros((120, 1))
    lam[np.asarray(dom.overlapping_point_indices)] = coef
    np.testing.assert_allclose(lam, np.linalg.solve(D.a_matrix(pts, st), vals), rtol=1e-8, atol=1e-10)
