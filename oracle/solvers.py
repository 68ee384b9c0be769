"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's iterative solvers
(ferreus_rbf/src/iterative_solvers.rs).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product (ferreus_rbf_rs_amd) never does.

Parity unpinned: the reference has no numeric test of these routines and cannot be built here;
the restatement follows the text of the file line by line and is checked against dense numpy
solves and LAPACK dlartg properties (tests/test_solvers.py).
"""
from __future__ import annotations

import math

import numpy as np

ABSOLUTE, RELATIVE = 0, 1   # FittingAccuracyType (interpolant_config.rs:54-63); this repo's C ABI ids


def progress_from_rel(current_res, start_res, target_res):
    """progress.rs:124-130"""
    if current_res <= target_res:
        return 1.0
    return (math.log10(start_res) - math.log10(current_res)) / (math.log10(start_res) - math.log10(target_res))


def givens_rotation(f, g):
    """iterative_solvers.rs:185-227 (a port of LAPACK's dlartg): [c s; -s c][f; g] = [r; 0]."""
    safmin = np.finfo(np.float64).tiny
    safmax = np.finfo(np.float64).max
    rtmin = math.sqrt(safmin)
    rtmax = math.sqrt(safmax / 2.0)
    if g == 0.0:
        return 1.0, 0.0, f
    if f == 0.0:
        return 0.0, math.copysign(1.0, g), abs(g)
    f1, g1 = abs(f), abs(g)
    if rtmin <= f1 < rtmax and rtmin <= g1 < rtmax:
        r = math.copysign(math.sqrt(f * f + g * g), f)
        return f1 / abs(r), g / r, r
    u = min(max(max(f1, g1), safmin), safmax)
    fs, gs = f / u, g / u
    mag = math.sqrt(fs * fs + gs * gs)
    return abs(fs) / mag, gs / mag, math.copysign(mag, f) * u


def _get_solution(h, g, z, i):
    """iterative_solvers.rs:174-183: x-update = Z[:, :i] * (H[:i, :i]^-1 g[:i]) (upper triangular)."""
    y = np.array(g[:i], dtype=np.float64)
    for r in range(i - 1, -1, -1):
        y[r] = (y[r] - h[r, r + 1:i] @ y[r + 1:i]) / h[r, r]
    return z[:, :i] @ y


def fgmres(a, b, m=None, x0=None, max_outer_iterations=20, max_inner_iterations=5,
           tolerance_type=RELATIVE, tolerance=1e-6, callback=None):
    """iterative_solvers.rs:38-172.  a, m: callables on 1-D arrays.  Returns (x, history) where
    history holds the (iteration, residual) pairs the reference emits as SolverIteration events."""
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    n = b.size
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64).reshape(-1)
    r = b - a(x)
    beta = np.abs(r).max() if tolerance_type == ABSOLUTE else np.linalg.norm(r)
    iteration = 1
    mi = max_inner_iterations
    history = []
    for outer in range(max_outer_iterations):
        v = np.zeros((n, mi + 1))
        h = np.zeros((mi + 1, mi))
        z = np.zeros((n, mi))
        g = np.zeros(mi + 1)
        cs = np.zeros(mi)
        sn = np.zeros(mi)
        r_norm = np.linalg.norm(r)
        v[:, 0] = r / r_norm
        g[0] = r_norm
        for j in range(mi):
            w = np.array(v[:, j]) if m is None else np.asarray(m(v[:, j]), dtype=np.float64).reshape(-1)
            z[:, j] = w
            wj = np.array(a(w), dtype=np.float64).reshape(-1)
            for i in range(j + 1):                      # modified Gram-Schmidt
                hij = float(v[:, i] @ wj)
                h[i, j] = hij
                wj -= v[:, i] * hij
            norm = np.linalg.norm(wj)
            h[j + 1, j] = norm
            for i in range(j):                          # previous rotations
                temp = cs[i] * h[i, j] + sn[i] * h[i + 1, j]
                h[i + 1, j] = -sn[i] * h[i, j] + cs[i] * h[i + 1, j]
                h[i, j] = temp
            c, s, _ = givens_rotation(h[j, j], h[j + 1, j])
            h[j, j] = c * h[j, j] + s * h[j + 1, j]
            h[j + 1, j] = 0.0
            temp = c * g[j] + s * g[j + 1]
            g[j + 1] = -s * g[j] + c * g[j + 1]
            g[j] = temp
            cs[j], sn[j] = c, s
            if norm != 0.0:
                v[:, j + 1] = wj / norm
            res_norm = abs(g[j + 1]) if tolerance_type == ABSOLUTE else abs(g[j + 1]) / beta
            history.append((iteration, res_norm))
            if callback is not None:
                callback(iteration, res_norm, progress_from_rel(res_norm, beta, tolerance))
            if res_norm < tolerance:
                x = x + _get_solution(h, g, z, j + 1)
                return x, history
            iteration += 1
        x = x + _get_solution(h, g, z, mi)
        r = b - a(x)
        res_norm = np.abs(r).max() if tolerance_type == ABSOLUTE else np.linalg.norm(r) / beta
        if res_norm < tolerance:
            break
    return x, history


def schwarz_ddm_solver(matvec, rhs, m=None, max_iterations=100, tolerance_type=RELATIVE, tolerance=1e-6,
                       callback=None):
    """iterative_solvers.rs:229-281: stationary iteration s += M(r); r = rhs - A s."""
    rhs = np.asarray(rhs, dtype=np.float64).reshape(-1)
    rg = np.array(rhs)
    sg = np.zeros_like(rhs)
    beta = np.abs(rg).max() if tolerance_type == ABSOLUTE else np.linalg.norm(rg)
    res_norm = beta
    iteration = 0
    history = []
    if m is not None:
        while res_norm > tolerance and iteration < max_iterations:
            sg = sg + np.asarray(m(rg), dtype=np.float64).reshape(-1)
            rg = rhs - matvec(sg)
            res_norm = np.abs(rg).max() if tolerance_type == ABSOLUTE else np.linalg.norm(rg) / beta
            iteration += 1
            history.append((iteration, res_norm))
            if callback is not None:
                callback(iteration, res_norm, progress_from_rel(res_norm, beta, tolerance))
    return sg, history
