#!/usr/bin/env python3
"""Phase times of the partial matvecs of a Schwarz sweep (IterativeSolver::matvec_partial, rbf.rs:119-133): 10M
sources, random target subsets of the sizes of the sweep's levels.  args: points kernel_id order"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
order = int(sys.argv[3]) if len(sys.argv) > 3 else 9
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
w = np.random.default_rng(43).standard_normal(n)
for m in (n // 8, n // 64, n // 512):
    idx = np.sort(np.random.default_rng(7).choice(n, m, replace=False)).astype(np.int64)
    tree.fast_matrix_vector_product(w, target_indices=idx)
    tree.set_profiling(True); tree.phase_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(4):
        tree.fast_matrix_vector_product(w, target_indices=idx)
    ms = (time.perf_counter() - t0) / 4 * 1e3
    ph = tree.phase_ms(); tree.set_profiling(False)
    print(json.dumps({"points": n, "order": order, "subset": m, "ms": round(ms, 2), "phases": {k: round(v / 4, 2) for k, v in ph.items() if v / 4 > 0.05}}), flush=True)
