#!/usr/bin/env python3
"""PCIe-inclusive matvec rate: bbfmm_fast_matrix_vector_product on host buffers (10M points),
through the Python wrapper (allocates the result each call) and with a preallocated result."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
op = F.solvers.RbfSystemOperator(tree, 0, None, 0.0)
w = np.random.default_rng(43).random(n)
op(w)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); op(w); ts.append(time.perf_counter() - t0)
y = np.zeros(n); lib = L.load(); ts2 = []
for _ in range(5):
    t0 = time.perf_counter()
    rc = lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n, 0, None, 0, None, 0, 0.0, y.ctypes.data)
    ts2.append(time.perf_counter() - t0)
print(json.dumps({"points": n, "wrapper_median_ms": round(sorted(ts)[2] * 1e3, 2),
                  "preallocated_result_median_ms": round(sorted(ts2)[2] * 1e3, 2),
                  "matvecs_per_s_preallocated": round(1 / sorted(ts2)[2], 2)}))
