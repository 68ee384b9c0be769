"""FGMRES around the device matvec: bbfmm_fgmres + bbfmm_rbf_system_apply against
the numpy restatement driven by the dense system matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from oracle import bbfmm_oracle as O
from oracle import solvers as OS


def _dense_system(pts, kid, nugget, poly):
    n = pts.shape[0]
    d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=2)
    Kmat = np.vectorize(lambda r: O.kernel_phi(kid, r))(d) if kid != 0 else -d
    Kmat = Kmat + nugget * np.eye(n)
    m = 0 if poly is None else poly.shape[1]

    def matvec(w):                                   # rbf.rs:1338-1379: tail rows stay zero
        y = np.zeros(n + m)
        y[:n] = Kmat @ w[:n] + (poly @ w[n:] if m else 0.0)
        return y
    return matvec


@pytest.mark.parametrize("with_poly", [False, True])
def test_fgmres_on_device_matvec_matches_dense_restatement(with_poly):
    n = 3000
    rng = np.random.default_rng(21)
    pts = rng.random((n, 3))
    poly = np.column_stack([np.ones(n), pts]) if with_poly else None
    m = 4 if with_poly else 0
    nugget = 0.05
    tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
    op = S.RbfSystemOperator(tree, m, poly, nugget)
    dense = _dense_system(pts, 0, nugget, poly)
    w = rng.standard_normal(n + m)
    yd = dense(w)
    assert np.abs(op(w) - yd).max() < 1e-6 * np.abs(yd).max()          # the operator itself
    b = np.concatenate([np.sin(4 * pts[:, 0]) + pts[:, 1] * pts[:, 2], np.zeros(m)])
    tol = S.FittingAccuracy(1e-12)                                       # never reached: 3 x 5 iterations
    x, hist = S.fgmres(op, b, None, None, 3, 5, tol)
    # (1) the driver: the restatement run on the SAME device operator follows it step for step
    xs, hists = OS.fgmres(op, b, None, None, 3, 5, OS.RELATIVE, 1e-12)
    assert [h[0] for h in hist] == list(range(1, 16)) == [h[0] for h in hists]
    np.testing.assert_allclose([h[1] for h in hist], [h[1] for h in hists], rtol=1e-5)
    assert np.abs(x - xs).max() < 1e-5 * np.abs(xs).max()
    # (2) the solve: against the dense system the first cycle agrees to the FMM accuracy; later
    # cycles amplify the 1e-7 operator difference (GMRES restarts), so they are compared loosely
    xo, histo = OS.fgmres(dense, b, None, None, 3, 5, OS.RELATIVE, 1e-12)
    np.testing.assert_allclose([h[1] for h in hist[:5]], [h[1] for h in histo[:5]], rtol=1e-4)
    np.testing.assert_allclose([h[1] for h in hist], [h[1] for h in histo], rtol=0.1)
    res = np.linalg.norm(dense(x) - b) / np.linalg.norm(b)
    reso = np.linalg.norm(dense(xo) - b) / np.linalg.norm(b)
    assert abs(res - reso) < 0.1 * reso and res < 0.1                   # same reduction of the true residual


def test_schwarz_iteration_with_python_preconditioner_on_device_matvec():
    n = 2000
    rng = np.random.default_rng(22)
    pts = rng.random((n, 3))
    tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
    nugget = 40.0                                    # diagonally dominant: Jacobi converges
    op = S.RbfSystemOperator(tree, 0, None, nugget)
    dense = _dense_system(pts, 0, nugget, None)
    b = rng.standard_normal(n)
    pre = lambda r: r / nugget
    x, hist = S.schwarz_ddm_solver(op, b, pre, 6, S.FittingAccuracy(1e-14))
    xo, histo = OS.schwarz_ddm_solver(dense, b, pre, 6, OS.RELATIVE, 1e-14)
    assert len(hist) == len(histo) == 6
    np.testing.assert_allclose([h[1] for h in hist], [h[1] for h in histo], rtol=1e-4)
    assert np.abs(x - xo).max() < 1e-5 * np.abs(xo).max()
