#!/usr/bin/env python3
"""Exactness of the coarse-domain solve at scale: z = coarse solve of r, then the system product of z
on the coarse points must reproduce r there (the coarse solve is a direct solve of that block)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
n = int(sys.argv[1]); ct = int(sys.argv[2]); kid = int(sys.argv[3]) if len(sys.argv) > 3 else 1
order = 9 if kid == 1 else 7
rng = np.random.default_rng(42)
pts = rng.random((n, 3))
tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
st = InterpolantSettings(kid, 3)
pre = SchwarzPreconditioner(tree, pts, st, DDMParams(coarse_threshold=ct), global_scaling=True)
lv = pre.num_levels - 1
cidx = pre.level_points(lv)
r = np.zeros(n + st.basis_size); r[cidx] = rng.standard_normal(len(cidx))
z = pre.debug_level_solve(lv, r, True)
from oracle import bbfmm_oracle as O      # dense check of the coarse block (the FMM product would add its own error times |z|)
Acc = np.array(O.kernel_matrix(kid, 1.0, 1.0, pts[cidx], pts[cidx]))
y = np.zeros_like(r)
y[cidx] = Acc @ z[cidx] + (pre.monomial_matrix[cidx] @ z[n:] if st.basis_size else 0.0)
err = np.abs(y[cidx] - r[cidx]).max() / np.abs(r).max()
ptl = np.abs(pre.monomial_matrix.T @ z[:n]).max() / np.abs(z[:n]).max()
print(json.dumps({"points": n, "levels": pre.num_levels, "coarse_points": int(len(cidx)), "coarse_solve_residual": float(err),
                  "side_condition": float(ptl), "|z|max": float(np.abs(z).max())}))
