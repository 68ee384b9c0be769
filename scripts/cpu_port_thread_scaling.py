#!/usr/bin/env python3
"""Thread scaling of the GEMM-shaped CPU port on the bench's sample (1.25M points), with and without OpenMP placement
(OMP_PLACES=cores OMP_PROC_BIND=spread: one thread per physical core): does 'all cores' beat a quarter of the hardware
threads once SMT siblings are left alone?  Run as:  [OMP_PLACES=cores OMP_PROC_BIND=spread] python scripts/cpu_port_thread_scaling.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bbfmm_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
pts = np.random.default_rng(42).random((n, 3)); w = np.random.default_rng(43).random((n, 1))
tree = O.FmmTree(pts, 7, 0, True, True); tree.gemm_shaped = True
hw = int(O.lib().oracle_num_threads())
out = {"points": n, "hardware_threads": hw, "cpus_in_affinity_mask": len(os.sched_getaffinity(0)),
       "OMP_PLACES": os.environ.get("OMP_PLACES"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "seconds_per_matvec": {}}
for th in sorted({hw, hw // 2, hw // 4, hw // 8}, reverse=True):
    if th < 1: continue
    O.lib().oracle_set_num_threads(th)
    tree.set_weights(w); tree.evaluate(w, pts)
    ts = []
    for _ in range(3):
        t0 = time.time(); tree.set_weights(w); tree.evaluate(w, pts); ts.append(time.time() - t0)
    out["seconds_per_matvec"][str(th)] = round(float(np.median(ts)), 3)
print(json.dumps(out))
