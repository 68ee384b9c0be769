#!/usr/bin/env python3
"""Device-memory leak check: handles (tree, Schwarz preconditioner) created, used and destroyed repeatedly;
free device memory must return to where it was."""
import gc, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
n = 200_000
pts = np.random.default_rng(1).random((n, 3))
vals = np.sin(3 * pts[:, 0]) + pts[:, 2] ** 2
free = []
for rep in range(12):
    tree = F.FmmTree(pts, 9, F.KernelParams(F.KernelType(1)), True, True)
    st = InterpolantSettings(1, 3)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams.for_points(n))
    op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, 0.0)
    x, hist = S.fgmres(op, np.concatenate([vals, np.zeros(st.basis_size)]), pre, None, 20, 5, S.FittingAccuracy(1e-6))
    tree.evaluate(x[:n, None].copy(), pts[:5000])
    tree.set_local_coefficients(x[:n, None].copy()); tree.evaluate_leaves(None, pts[:100000])
    del op, pre, tree
    gc.collect(); torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0])
print(json.dumps({"iterations": len(hist), "free_bytes_after_each_round": free, "drift_bytes_last_six_rounds": free[-1] - free[-7]}))
