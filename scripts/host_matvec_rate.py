#!/usr/bin/env python3
"""PCIe-inclusive matvec rate: bbfmm_fast_matrix_vector_product on host buffers (10M points)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
op = F.solvers.RbfSystemOperator(tree, 0, None, 0.0)
w = np.random.default_rng(43).random(n)
op(w)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); op(w); ts.append(time.perf_counter() - t0)
print(json.dumps({"points": n, "host_buffer_matvec_ms": [round(t * 1e3, 2) for t in ts],
                  "median_ms": round(sorted(ts)[2] * 1e3, 2), "matvecs_per_s": round(1 / sorted(ts)[2], 2)}))
