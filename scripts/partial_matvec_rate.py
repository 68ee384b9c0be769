#!/usr/bin/env python3
"""matvec_partial (rbf.rs:119-133) at 10M sources: target subsets of the sizes the Schwarz levels use."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
w = np.random.default_rng(43).random(n)
out = {"points": n}
full = tree.fast_matrix_vector_product(w)
for frac in (1 / 8, 1 / 64, 1 / 512, 1 / 4096):
    m = int(n * frac)
    idx = np.sort(np.random.default_rng(int(1 / frac)).choice(n, m, replace=False)).astype(np.int64)
    t0 = time.perf_counter(); y = tree.fast_matrix_vector_product(w, target_indices=idx); first = time.perf_counter() - t0
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); y = tree.fast_matrix_vector_product(w, target_indices=idx); ts.append(time.perf_counter() - t0)
    err = np.abs(y[idx] - full[idx]).max() / np.abs(full).max()
    nz = np.count_nonzero(y) - np.count_nonzero(y[idx])
    out[f"subset_{m}"] = {"first_ms": round(first * 1e3, 1), "cached_ms": round(sorted(ts)[1] * 1e3, 2),
                          "max_diff_vs_full": float(err), "rows_outside_subset_nonzero": int(nz)}
print(json.dumps(out))
