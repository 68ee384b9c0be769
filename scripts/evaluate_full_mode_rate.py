#!/usr/bin/env python3
"""evaluate() in Full mode (no set_local_coefficients): per call the downward pass over cells_with_targets
(bbfmm.rs:468-480) is planned on the host and run; small and large target batches.  args: sources"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(5)
pts = rng.random((n, 3))
w = rng.random((n, 1))
tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
tree.set_weights(w)
out = {"sources": n}
for m in (1000, 100_000, 1_000_000, 4_000_000):
    x = np.asfortranarray(pts[rng.choice(n, m, replace=False)] * 0.999)
    tree.evaluate(w, x)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); tree.evaluate(w, x); ts.append(time.perf_counter() - t0)
    out[f"evaluate_{m}"] = {"ms": round(min(ts) * 1e3, 2)}
print(json.dumps(out))
