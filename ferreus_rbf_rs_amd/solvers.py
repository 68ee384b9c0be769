"""Host iterative solvers of ferreus_rbf behind the C ABI (ferreus_rbf/src/iterative_solvers.rs).

`fgmres` / `schwarz_ddm_solver` take Python callables for the operator and the right
preconditioner exactly like the reference takes closures; `RbfSystemOperator` is the
reference's `IterativeSolver::matvec` (rbf.rs:105-117) on the device tree, bound natively (no
Python in the matvec path).  Everything computes in libferreus_bbfmm_hip.so.
"""
from __future__ import annotations

import ctypes
import enum
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _lib as L


class FittingAccuracyType(enum.IntEnum):
    """interpolant_config.rs:54-63"""
    Absolute = L.ACCURACY_ABSOLUTE
    Relative = L.ACCURACY_RELATIVE


class FittingAccuracy:
    """interpolant_config.rs:80-92 (defaults 1e-6, Relative)."""

    def __init__(self, tolerance: float = 1e-6,
                 tolerance_type: FittingAccuracyType = FittingAccuracyType.Relative):
        self.tolerance = float(tolerance)
        self.tolerance_type = FittingAccuracyType(tolerance_type)


def givens_rotation(f: float, g: float) -> Tuple[float, float, float]:
    """iterative_solvers.rs:185-227"""
    c, s, r = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    L.load().bbfmm_givens_rotation(f, g, ctypes.byref(c), ctypes.byref(s), ctypes.byref(r))
    return c.value, s.value, r.value


class _Operator:
    """(function pointer, user pointer) pair for a bbfmm_apply_fn; keeps its referents alive."""

    def __init__(self, fn_ptr, user_ptr, keep):
        self.fn_ptr, self.user_ptr, self._keep = fn_ptr, user_ptr, keep


def _wrap_callable(f: Callable[[np.ndarray], np.ndarray]) -> _Operator:
    err: List[BaseException] = []

    def tramp(_user, x, y, n):
        try:
            xa = np.ctypeslib.as_array(x, shape=(n,))
            out = np.asarray(f(xa), dtype=np.float64).reshape(-1)
            if out.size != n:
                raise ValueError(f"operator returned {out.size} values for a vector of {n}")
            np.ctypeslib.as_array(y, shape=(n,))[:] = out
            return L.OK
        except BaseException as e:  # noqa: BLE001 -- must not unwind through C
            err.append(e)
            return L.BAD_ARGUMENT

    cb = L.APPLY_FN(tramp)
    op = _Operator(ctypes.cast(cb, ctypes.c_void_p), None, (cb, err))
    op.errors = err
    return op


class RbfSystemOperator:
    """IterativeSolver::matvec (rbf.rs:105-117): y = fast_matrix_vector_product(tree, x, basis_size,
    all sources, monomial_matrix, nugget) with the native entry point as the callback."""

    def __init__(self, tree, basis_size: int = 0, monomial_matrix=None, nugget: float = 0.0):
        self.tree = tree
        self.n = tree.n_points + int(basis_size)
        self._poly = None if monomial_matrix is None else np.asfortranarray(monomial_matrix, dtype=np.float64)
        self._sys = L.RbfSystem(tree._h, int(basis_size),
                                None if self._poly is None else self._poly.ctypes.data,
                                0 if self._poly is None else self._poly.shape[0], float(nugget))
        lib = L.load()
        self._op = _Operator(ctypes.cast(lib.bbfmm_rbf_system_apply, ctypes.c_void_p),
                             ctypes.cast(ctypes.pointer(self._sys), ctypes.c_void_p), (self._sys, self._poly, tree))
        self._op.errors = []

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        y = np.zeros_like(x)
        rc = L.load().bbfmm_rbf_system_apply(self._op.user_ptr, x.ctypes.data, y.ctypes.data, x.size)
        if rc != L.OK:
            raise RuntimeError(f"bbfmm_rbf_system_apply failed with status {rc}")
        return y


def _as_operator(f) -> Optional[_Operator]:
    if f is None:
        return None
    if isinstance(f, RbfSystemOperator) or isinstance(getattr(f, "_op", None), _Operator):
        return f._op                      # natively bound operators (system matvec, Schwarz preconditioner)
    return _wrap_callable(f)


def _run(entry, n, a, m, args_mid, callback):
    hist: List[Tuple[int, float]] = []

    def on_iter(_user, it, res, progress):
        hist.append((int(it), float(res)))
        if callback is not None:
            callback(int(it), float(res), float(progress))

    cb = L.ITERATION_FN(on_iter)
    x = np.zeros(n)
    iters, res = ctypes.c_int64(0), ctypes.c_double(0.0)
    rc = entry(n, a.fn_ptr, a.user_ptr, *args_mid(m), cb, None, x.ctypes.data, ctypes.byref(iters),
               ctypes.byref(res))
    for op in (a, m):
        if op is not None and getattr(op, "errors", None):
            raise op.errors[0]
    if rc != L.OK:
        raise RuntimeError(f"solver failed with status {rc}")
    return x, hist


def fgmres(a, b, m=None, x0=None, max_outer_iterations: int = 20, max_inner_iterations: int = 5,
           tolerance: Optional[FittingAccuracy] = None, callback=None):
    """fgmres (iterative_solvers.rs:38-172).  a, m: callables on 1-D float64 arrays (or a
    RbfSystemOperator).  Returns (x, [(iteration, residual), ...])."""
    tol = tolerance or FittingAccuracy()
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
    ao, mo = _as_operator(a), _as_operator(m)
    lib = L.load()

    def mid(mo_):
        return (b.ctypes.data, None if mo_ is None else mo_.fn_ptr, None if mo_ is None else mo_.user_ptr,
                None if x0a is None else x0a.ctypes.data, int(max_outer_iterations), int(max_inner_iterations),
                int(tol.tolerance_type), tol.tolerance)

    return _run(lib.bbfmm_fgmres, b.size, ao, mo, mid, callback)


def schwarz_ddm_solver(matvec, rhs, m=None, max_iterations: int = 100,
                       tolerance: Optional[FittingAccuracy] = None, callback=None):
    """schwarz_ddm_solver (iterative_solvers.rs:229-281)."""
    tol = tolerance or FittingAccuracy()
    rhs = np.ascontiguousarray(rhs, dtype=np.float64).reshape(-1)
    ao, mo = _as_operator(matvec), _as_operator(m)
    lib = L.load()

    def mid(mo_):
        return (rhs.ctypes.data, None if mo_ is None else mo_.fn_ptr, None if mo_ is None else mo_.user_ptr,
                int(max_iterations), int(tol.tolerance_type), tol.tolerance)

    return _run(lib.bbfmm_schwarz_ddm_solver, rhs.size, ao, mo, mid, callback)
