#!/usr/bin/env python3
"""bbfmm_create stage times (BBFMM_VERBOSE=1 prints them to stderr): python scripts/build_stages.py [points] [order]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BBFMM_VERBOSE"] = "1"
import numpy as np
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
order = int(sys.argv[2]) if len(sys.argv) > 2 else 7
pts = np.random.default_rng(42).random((n, 3))
for rep in range(2):
    t0 = time.time()
    t = F.FmmTree(pts, order, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
    print(f"bbfmm_create #{rep}: {time.time() - t0:.3f} s", file=sys.stderr)
    del t
