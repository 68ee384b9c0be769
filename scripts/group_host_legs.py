#!/usr/bin/env python3
"""The unchanged caller (bbfmm_set_weights + bbfmm_evaluate at the source rows) on HOST buffers through a device group,
with the library's own leg times (BBFMM_VERBOSE): staging, queueing, comparison, passes + way back, host row writes.
args: [device list, default 0,0] [points, default 10M]"""
import ctypes, json, os, sys, time
os.environ["BBFMM_VERBOSE"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L
devs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,0").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
pts = np.asfortranarray(np.random.default_rng(42).random((n, 3)))
lib = L.load()
tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(0)), True, True, devices=devs if len(devs) > 1 else None)
w = np.random.default_rng(43).random(n)
y = np.zeros(n)
bad = ctypes.c_int64(-1)
ts = []
for i in range(6):
    sys.stderr.write(f"--- call {i}\n")
    t0 = time.perf_counter()
    rc = lib.bbfmm_set_weights(tree._h, w.ctypes.data, n, 1, n)
    t1 = time.perf_counter()
    rc = rc or lib.bbfmm_evaluate(tree._h, w.ctypes.data, n, 1, n, pts.ctypes.data, n, n, y.ctypes.data, n, ctypes.byref(bad))
    t2 = time.perf_counter()
    assert rc == 0
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
ts = sorted(ts[1:], key=lambda t: t[0] + t[1])
print(json.dumps({"devices": devs, "points": n, "set_weights_ms": round(ts[2][0], 2), "evaluate_ms": round(ts[2][1], 2),
                  "total_ms": round(ts[2][0] + ts[2][1], 2)}))
