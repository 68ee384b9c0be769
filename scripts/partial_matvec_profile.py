#!/usr/bin/env python3
"""One target subset of matvec_partial, repeated: run under rocprofv3 --kernel-trace --stats to see which
kernels a Schwarz-level partial matvec spends its time in.  args: points subset kernel_id order"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]); m = int(sys.argv[2]); kid = int(sys.argv[3]); order = int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
w = np.random.default_rng(43).standard_normal(n)
idx = np.sort(np.random.default_rng(7).choice(n, m, replace=False)).astype(np.int64)
tree.fast_matrix_vector_product(w, target_indices=idx)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); tree.fast_matrix_vector_product(w, target_indices=idx); ts.append(time.perf_counter() - t0)
print(json.dumps({"points": n, "subset": m, "kernel": kid, "order": order, "ms": round(float(np.median(ts)) * 1e3, 2)}))
