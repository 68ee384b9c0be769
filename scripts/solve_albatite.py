#!/usr/bin/env python3
"""The reference's two example solves (ferreus_rbf/examples/isosurface_spheroidal.rs, isosurface_linear.rs) on its own
data set (tests/golden/albatite_SD_points.npz: 35,801 points, raw coordinates), timed on the device: tree, preconditioner
setup, FGMRES 20 x 5 to absolute 0.01 with the default DDMParams -- what `RBFInterpolator::builder(..).build()` does
(rbf.rs:411-574).  The reference documents one timing of its own: 'Took 2.870149s to solve RBF for 26988 points'
(py_ferreus_rbf/docs/api/progress.md:66-70; Spheroidal, absolute 0.01; ANOTHER data set, unknown CPU): qualitative only."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner

rows = np.load(os.path.join(ROOT, "tests", "golden", "albatite_SD_points.npz"))["rows"]
pts, vals = np.ascontiguousarray(rows[:, :3]), rows[:, 3].copy()
n = len(pts)
out = {"points": n, "reference_documented": "2.870149 s for 26,988 points, Spheroidal, absolute 0.01, 8 iterations (progress.md:53-70; different data set and an unknown CPU: qualitative only)"}
F.FmmTree(pts[:2000], 7, F.KernelParams(F.KernelType(0)), True, True)          # (first-use costs of the device out of the way)
for name, kid, br, sill in (("spheroidal", 3, 50.0, 10.0), ("linear", 0, 1.0, 1.0)):
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, True)
        t1 = time.perf_counter()
        st = InterpolantSettings(kid, 3, None, 0.0, br, sill)
        pre = SchwarzPreconditioner(tree, pts, st, DDMParams())
        t2 = time.perf_counter()
        op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, 0.0)
        rhs = np.concatenate([vals, np.zeros(st.basis_size)])
        x, hist = S.fgmres(op, rhs, pre, None, 20, 5, S.FittingAccuracy(0.01, S.FittingAccuracyType.Absolute))
        t3 = time.perf_counter()
        rec = {"tree_s": round(t1 - t0, 4), "preconditioner_setup_s": round(t2 - t1, 4), "solve_s": round(t3 - t2, 4),
               "total_s": round(t3 - t0, 4), "iterations": len(hist), "final_residual": float(hist[-1][1]),
               "max_abs_misfit_at_the_data": float(np.abs(op(x)[:n] - vals).max())}
        if best is None or rec["total_s"] < best["total_s"]:
            best = rec
        pre.close()
        del pre, op, tree
    out[name] = best
print(json.dumps(out))
