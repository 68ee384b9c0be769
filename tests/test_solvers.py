"""Host iterative solvers: the C-ABI drivers against the numpy restatement of
ferreus_rbf/src/iterative_solvers.rs (oracle/solvers.py) and against dense numpy solves.
No GPU needed: the operators are callbacks."""
import math

import numpy as np
import pytest

from oracle import solvers as OS
from ferreus_rbf_rs_amd import solvers as S


def _spd(n, seed, cond=50.0):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (q * np.geomspace(1.0, cond, n)) @ q.T


@pytest.mark.parametrize("f,g", [(3.0, 4.0), (-3.0, 4.0), (3.0, -4.0), (0.0, 2.5), (0.0, -2.5), (7.0, 0.0),
                                 (1e-200, 1e-180), (1e200, -1e190), (5e-324, 1.0), (1.0, 5e-324)])
def test_givens_rotation_matches_restatement_and_dlartg_properties(f, g):
    c, s, r = S.givens_rotation(f, g)
    co, so, ro = OS.givens_rotation(f, g)
    assert (c, s, r) == (co, so, ro)                     # same arithmetic, bit for bit
    assert abs(c * c + s * s - 1.0) < 4e-16
    assert abs(-s * f + c * g) <= 4e-16 * max(abs(f), abs(g))          # second row annihilated
    assert math.isclose(c * f + s * g, r, rel_tol=4e-16)
    assert c >= 0.0


@pytest.mark.parametrize("tol_type", [S.FittingAccuracyType.Relative, S.FittingAccuracyType.Absolute])
@pytest.mark.parametrize("precond", [False, True])
def test_fgmres_matches_restatement(tol_type, precond):
    n = 300
    A = _spd(n, 1, cond=200.0) + 0.05 * np.random.default_rng(2).standard_normal((n, n))
    b = np.random.default_rng(3).standard_normal(n)
    Minv = np.diag(1.0 / np.diag(A)) if precond else None
    a = lambda x: A @ x
    m = (lambda x: Minv @ x) if precond else None
    tol = S.FittingAccuracy(1e-8, tol_type)
    x, hist = S.fgmres(a, b, m, None, 20, 5, tol)
    xo, histo = OS.fgmres(a, b, m, None, 20, 5, int(tol_type), 1e-8)
    assert len(hist) == len(histo) and [h[0] for h in hist] == [h[0] for h in histo]
    np.testing.assert_allclose([h[1] for h in hist], [h[1] for h in histo], rtol=1e-6)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-9 * np.abs(xo).max())
    # iteration numbering: 1, 2, 3, ... across restarts (iterative_solvers.rs:60,156)
    assert [h[0] for h in hist] == list(range(1, len(hist) + 1))


def test_fgmres_converges_to_dense_solution_and_reports_events():
    n = 200
    A = _spd(n, 5, cond=30.0)
    b = np.random.default_rng(6).standard_normal(n)
    events = []
    x, hist = S.fgmres(lambda v: A @ v, b, None, None, 50, 5, S.FittingAccuracy(1e-10),
                       callback=lambda it, res, prog: events.append((it, res, prog)))
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 2e-10
    assert [(e[0], e[1]) for e in events] == hist
    assert events[-1][2] == 1.0 and all(0.0 <= e[2] <= 1.0 for e in events)   # progress_from_rel
    assert hist[-1][1] < 1e-10


def test_fgmres_initial_guess_and_restart_only_exit():
    n = 120
    A = _spd(n, 8, cond=10.0)
    xs = np.random.default_rng(9).standard_normal(n)
    b = A @ xs
    x0 = xs + 1e-3 * np.random.default_rng(10).standard_normal(n)
    x, hist = S.fgmres(lambda v: A @ v, b, None, x0, 20, 5, S.FittingAccuracy(1e-9))
    xo, histo = OS.fgmres(lambda v: A @ v, b, None, x0, 20, 5, OS.RELATIVE, 1e-9)
    np.testing.assert_allclose(x, xo, atol=1e-10)
    assert len(hist) == len(histo)
    # exact solution as the initial guess: zero residual returns at once (documented deviation)
    x, hist = S.fgmres(lambda v: A @ v, b * 0.0, None, None, 3, 5)
    assert np.all(x == 0.0) and hist == []


def test_fgmres_flexible_preconditioner_changes_every_call():
    """The 'F' of FGMRES: the preconditioner may differ per iteration (z_j is stored)."""
    n = 150
    A = _spd(n, 11, cond=100.0)
    b = np.random.default_rng(12).standard_normal(n)
    d = 1.0 / np.diag(A)
    calls = []

    def m(v):
        calls.append(1)
        return d * v * (1.0 + 0.1 * (len(calls) % 3))
    x, hist = S.fgmres(lambda v: A @ v, b, m, None, 40, 5, S.FittingAccuracy(1e-9))
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-8
    assert len(calls) == len(hist)


@pytest.mark.parametrize("tol_type", [S.FittingAccuracyType.Relative, S.FittingAccuracyType.Absolute])
def test_schwarz_ddm_solver_matches_restatement(tol_type):
    n = 100
    A = np.eye(n) * 4.0 + 0.5 * _spd(n, 13, cond=3.0)
    b = np.random.default_rng(14).standard_normal(n)
    d = 1.0 / np.diag(A)
    x, hist = S.schwarz_ddm_solver(lambda v: A @ v, b, lambda r: d * r, 100, S.FittingAccuracy(1e-9, tol_type))
    xo, histo = OS.schwarz_ddm_solver(lambda v: A @ v, b, lambda r: d * r, 100, int(tol_type), 1e-9)
    assert len(hist) == len(histo) and len(hist) < 100
    np.testing.assert_allclose([h[1] for h in hist], [h[1] for h in histo], rtol=1e-9)
    np.testing.assert_allclose(x, xo, atol=1e-12)
    assert np.linalg.norm(A @ x - b) < 1e-7
    # without a preconditioner the reference returns zeros (iterative_solvers.rs:256)
    x, hist = S.schwarz_ddm_solver(lambda v: A @ v, b, None)
    assert np.all(x == 0.0) and hist == []


def test_operator_errors_propagate():
    def bad(v):
        raise ValueError("boom")
    with pytest.raises(ValueError, match="boom"):
        S.fgmres(bad, np.ones(4))
    with pytest.raises(ValueError, match="returned"):
        S.fgmres(lambda v: np.ones(3), np.ones(4))
