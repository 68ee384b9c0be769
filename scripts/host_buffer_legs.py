#!/usr/bin/env python3
"""The PCIe legs of the drop-in matvec on HOST buffers, each by itself (measure before hiding anything):
H2D and D2H of N doubles from / to pinned memory, the host's staging copy and its row permutation on the library's own
thread pool (bbfmm_debug_host_copy_rates), and the three end-to-end figures they sit between -- device-resident vectors,
the patched caller, the unchanged caller.  args: [points=10000000]"""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lib = L.load()
dev = torch.device("cuda", 0)
out = {"points": n}
hp = torch.empty(n, dtype=torch.float64).pin_memory()
dd = torch.empty(n, dtype=torch.float64, device=dev)


def med(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


out["h2d_pinned_ms"] = med(lambda: dd.copy_(hp, non_blocking=True))
out["d2h_pinned_ms"] = med(lambda: hp.copy_(dd, non_blocking=True))
out["h2d_gbps"] = n * 8 / out["h2d_pinned_ms"] * 1e-6
out["d2h_gbps"] = n * 8 / out["d2h_pinned_ms"] * 1e-6
rates = (ctypes.c_double * 4)()
lib.bbfmm_debug_host_copy_rates(ctypes.c_int64(n), rates)
out["host_memcpy_pool_ms"], out["host_row_gather_pool_ms"], out["host_row_scatter_pool_ms"], out["host_threads"] = [float(x) for x in rates]
pts = np.asfortranarray(np.random.default_rng(42).random((n, 3)))
tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(0)), True, True)
w = np.random.default_rng(43).random(n)
y = np.zeros(n)
bad = ctypes.c_int64(-1)
wd = torch.from_numpy(w).to(dev)
yd = torch.zeros(n, dtype=torch.float64, device=dev)
out["device_resident_ms"] = med(lambda: tree.matvec_device(wd.data_ptr(), n, 1, yd.data_ptr(), n, sync=True))
out["patched_caller_ms"] = med(lambda: lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n, 0, None, 0, None, 0, 0.0, y.ctypes.data))


def unchanged():
    lib.bbfmm_set_weights(tree._h, w.ctypes.data, n, 1, n)
    lib.bbfmm_evaluate(tree._h, w.ctypes.data, n, 1, n, pts.ctypes.data, n, n, y.ctypes.data, n, ctypes.byref(bad))


out["unchanged_caller_ms"] = med(unchanged)
ph = None
tree.set_profiling(True)
tree.phase_ms(reset=True)
for _ in range(5):
    tree.matvec_device(wd.data_ptr(), n, 1, yd.data_ptr(), n, sync=True)
ph = tree.phase_ms()
out["phase_ms"] = {k: round(v / 5, 3) for k, v in ph.items()}
out["exposed_over_device_resident_ms"] = {"patched": out["patched_caller_ms"] - out["device_resident_ms"],
                                          "unchanged": out["unchanged_caller_ms"] - out["device_resident_ms"]}
print(json.dumps(out))
