#!/usr/bin/env python3
"""matvec_partial (rbf.rs:119-133: fast_matrix_vector_product with target_indices) on HOST buffers, both ways a caller can
reach it: patched -- bbfmm_fast_matrix_vector_product(target_indices) (cached subset plans) -- and UNCHANGED --
bbfmm_set_weights(w) then bbfmm_evaluate(w, select_mat_rows(source_points, idx)) (rbf.rs:1357-1364).  10M uniform points,
thin-plate spline order 9 (config 3's products), row subsets of N/8, N/64, N/512 (the Schwarz levels).  args: [points]"""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
order, kid = (9, 1) if len(sys.argv) < 3 else (int(sys.argv[2]), int(sys.argv[3]))
rng = np.random.default_rng(42)
pts = np.asfortranarray(rng.random((n, 3)))
lib = L.load()
tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
w = rng.random(n)
out = {"points": n, "order": order, "kernel": F.KernelType(kid).name, "subsets": {}}
bad = ctypes.c_int64(-1)
for frac in (8, 64, 512):
    idx = np.sort(rng.choice(n, n // frac, replace=False)).astype(np.int64)
    m = len(idx)
    x = np.asfortranarray(pts[idx])                      # select_mat_rows: the caller's copy (not timed)
    y = np.zeros(n)
    z = np.zeros(m)

    def patched():
        return lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n, 0, idx.ctypes.data, m, None, 0, 0.0, y.ctypes.data)

    def unchanged():
        rc = lib.bbfmm_set_weights(tree._h, w.ctypes.data, n, 1, n)
        return rc or lib.bbfmm_evaluate(tree._h, w.ctypes.data, n, 1, n, x.ctypes.data, m, m, z.ctypes.data, m, ctypes.byref(bad))

    rec = {"rows": m}
    for name, fn in (("patched_ms", patched), ("unchanged_ms", unchanged)):
        assert fn() == 0
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        rec[name] = round(sorted(ts)[2] * 1e3, 2)
    rec["rel_diff"] = float(np.abs(y[idx] - z).max() / np.abs(z).max())
    rec["ratio"] = round(rec["unchanged_ms"] / rec["patched_ms"], 3)
    out["subsets"]["N/%d" % frac] = rec
print(json.dumps(out))
