#!/usr/bin/env python3
"""PCIe-inclusive matvec rate: bbfmm_fast_matrix_vector_product on host buffers, preallocated result;
LinearRbf p = 7 without polynomial part and thin-plate spline p = 9 with the linear polynomial tail
(what each FGMRES iteration of config 3 calls).  args: points"""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = np.random.default_rng(42).random((n, 3))
lib = L.load()
out = {"points": n}
for name, kid, order, basis in (("linear_p7", 0, 7, 0), ("tps_p9_linear_drift", 1, 9, 4)):
    tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
    w = np.random.default_rng(43).random(n + basis)
    P = np.asfortranarray(np.hstack([np.ones((n, 1)), pts])) if basis else None
    y = np.zeros(n + basis)
    def call():
        return lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n + basis, basis, None, 0,
                                                    P.ctypes.data if basis else None, n if basis else 0, 0.0, y.ctypes.data)
    assert call() == 0
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    tree.set_profiling(True); tree.phase_ms(reset=True); call(); dev = sum(tree.phase_ms().values()); tree.set_profiling(False)
    out[name] = {"host_buffers_ms": round(sorted(ts)[2] * 1e3, 2), "device_phases_ms": round(dev, 2)}
    del tree
print(json.dumps(out))
