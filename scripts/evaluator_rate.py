#!/usr/bin/env python3
"""Evaluator modes: evaluate / evaluate_leaves on arbitrary targets, large and
small batches (what isosurfacing does: set_local_coefficients once, then many evaluate_leaves)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(5)
pts = rng.random((n, 3)) * 2 - 1
w = rng.random((n, 1))
t0 = time.time()
tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, False, extents=[-2, -2, -2, 2, 2, 2])
build = time.time() - t0
tree.set_weights(w); tree.set_local_coefficients(w)
out = {"sources": n, "build_s": round(build, 2)}
for m in (1000, 100_000, 2_000_000):
    x = np.asfortranarray(rng.random((m, 3)) * 2 - 1)   # column-major, as the C ABI takes it (no wrapper copy)
    for name, fn in (("evaluate", tree.evaluate), ("evaluate_leaves", tree.evaluate_leaves),
                     ("evaluate_leaves_with_gradients", tree.evaluate_leaves_with_gradients)):
        fn(w, x)
        reps = 20 if m <= 1000 else 3
        t0 = time.perf_counter()
        for _ in range(reps): fn(w, x)
        dt = (time.perf_counter() - t0) / reps
        out[f"{name}_{m}"] = {"ms": round(dt * 1e3, 3), "Mtargets_per_s": round(m / dt * 1e-6, 3)}
    tree.evaluate_leaves(None, x)                       # weights left on the device (header: bbfmm_evaluate_leaves)
    reps = 50 if m <= 1000 else 3
    t0 = time.perf_counter()
    for _ in range(reps): tree.evaluate_leaves(None, x)
    dt = (time.perf_counter() - t0) / reps
    out[f"evaluate_leaves_resident_weights_{m}"] = {"ms": round(dt * 1e3, 3), "Mtargets_per_s": round(m / dt * 1e-6, 3)}
print(json.dumps(out))
