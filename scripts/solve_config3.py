#!/usr/bin/env python3
"""BASELINE.json config 3 ("TPS, N points, FGMRES+DDM, 1 GPU"): thin-plate spline interpolation of a
smooth function, linear drift, FGMRES 20 x 5 right-preconditioned by the multi-level Schwarz method,
matvecs and local solves on the GPU."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--kernel", default="ThinPlateSplineRbf")
ap.add_argument("--order", type=int, default=9)
ap.add_argument("--tol", type=float, default=1e-6)
ap.add_argument("--nugget", type=float, default=0.0)
ap.add_argument("--coarse-threshold", type=int, default=4096,
                help="0: DDMParams.for_points(n) (the extension that keeps three fine levels)")
ap.add_argument("--leaf-threshold", type=int, default=1024)
ap.add_argument("--precon-order", type=int, default=0,
                help="interpolation order of a second tree that serves only the preconditioner's partial matvecs "
                     "(0: the reference's arrangement, one tree for both)")
ap.add_argument("--precon-shared-basis", action="store_true",
                help="extension: the preconditioner's partial matvecs run on a second tree with BBFMM_FLAG_M2L_SHARED_BASIS "
                     "(FGMRES is flexible; the operator keeps the reference's arithmetic)")
ap.add_argument("--shared-basis", action="store_true",
                help="extension: ONE tree with BBFMM_FLAG_M2L_SHARED_BASIS for the operator and the preconditioner (the "
                     "operator's products then carry the basis' truncation, about epsilon)")
ap.add_argument("--clustered", action="store_true", help="points from a mixture of 12 Gaussian clusters instead of uniform")
a = ap.parse_args()
kid = {"LinearRbf": 0, "ThinPlateSplineRbf": 1, "CubicRbf": 2, "Spheroidal3Rbf": 3}[a.kernel]
n = a.points
rng = np.random.default_rng(42)
pts = rng.random((n, 3))
if a.clustered:
    c = rng.random((12, 3)); sg = 0.01 + 0.08 * rng.random(12); which = rng.integers(0, 12, n)
    pts = np.clip(c[which] + rng.normal(size=(n, 3)) * sg[which, None], 0.0, 0.999)
    pts = np.unique(pts, axis=0); n = pts.shape[0]          # (clipping can duplicate points on the box faces)
vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.5 * pts[:, 2] ** 2          # smooth test function
t0 = time.time()
tree = F.FmmTree(pts, a.order, F.KernelParams(F.KernelType(kid)), True, True, m2l_shared_basis=a.shared_basis)
t_tree = time.time() - t0
st = InterpolantSettings(kid, 3, nugget=a.nugget)
t0 = time.time()
ptree = tree
if (a.precon_order and a.precon_order != a.order) or a.precon_shared_basis:
    # FGMRES is flexible: the preconditioner may use cheaper (less accurate) products than the operator
    ptree = F.FmmTree(pts, a.precon_order or a.order, F.KernelParams(F.KernelType(kid)), True, True,
                      m2l_shared_basis=a.precon_shared_basis)
ct = a.coarse_threshold or DDMParams.for_points(n).coarse_threshold
pre = SchwarzPreconditioner(ptree, pts, st, DDMParams(leaf_threshold=a.leaf_threshold, coarse_threshold=ct))
t_ddm = time.time() - t0
op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, a.nugget)
rhs = np.concatenate([vals, np.zeros(st.basis_size)])
marks = []
t0 = time.time()
x, hist = S.fgmres(op, rhs, pre, None, 20, 5, S.FittingAccuracy(a.tol), callback=lambda it, r, p: marks.append(time.time()))
t_solve = time.time() - t0
idx = rng.choice(n, 2000, replace=False)
fit = op(x)[idx]
print(json.dumps({"points": n, "kernel": a.kernel, "order": a.order, "precon_order": a.precon_order or a.order, "precon_shared_basis": bool(a.precon_shared_basis), "shared_basis": bool(a.shared_basis), "levels": pre.num_levels, "basis": st.basis_size,
                  "fmm_tree_build_s": round(t_tree, 2), "ddm_build_and_factor_s": round(t_ddm, 2),
                  "solve_s": round(t_solve, 2), "iterations": len(hist),
                  "s_per_iteration": round(t_solve / max(len(hist), 1), 3),
                  "residual_history": [float("%.3e" % r) for _, r in hist],
                  "max_fit_error_on_sample": float(np.abs(fit - vals[idx]).max())}))
