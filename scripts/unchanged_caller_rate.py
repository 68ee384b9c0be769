#!/usr/bin/env python3
"""The unchanged caller of the matvec (rbf.rs:1357-1364: set_weights(w), then evaluate(w, the source rows)) on HOST
buffers, timed against the library found under the package root given as argv[1] (default: this checkout) -- so that
the same script gives the round-4 figure (a copy of the round-4 sources built under gpurun_variants/) and today's.
args: [package root] [points]"""
import ctypes, json, os, sys, time
import numpy as np
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L
assert os.path.abspath(L.LIB_PATH).startswith(root), (L.LIB_PATH, root)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
pts = np.asfortranarray(np.random.default_rng(42).random((n, 3)))
lib = L.load()
tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(0)), True, True)
w = np.random.default_rng(43).random(n)
y = np.zeros(n)
bad = ctypes.c_int64(-1)


def unchanged():
    rc = lib.bbfmm_set_weights(tree._h, w.ctypes.data, n, 1, n)
    return rc or lib.bbfmm_evaluate(tree._h, w.ctypes.data, n, 1, n, pts.ctypes.data, n, n, y.ctypes.data, n, ctypes.byref(bad))


def patched():
    return lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n, 0, None, 0, None, 0, 0.0, y.ctypes.data)


out = {"library": os.path.relpath(L.LIB_PATH, os.getcwd()), "points": n}
for name, fn in (("unchanged_caller_ms", unchanged), ("patched_caller_ms", patched)):
    assert fn() == 0
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    out[name] = round(sorted(ts)[2] * 1e3, 2)
print(json.dumps(out))
