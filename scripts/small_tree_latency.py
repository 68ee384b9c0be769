#!/usr/bin/env python3
"""Where a SMALL matvec goes (the reference's own examples are 35,801 points; BASELINE config 1 is 50k): wall time per
device-resident matvec against the sum of its kernels' times (hipEvent pairs per phase) and the number of launches.
args: [sizes=35801,50000,200000]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ferreus_rbf_rs_amd as F
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "35801,50000,200000").split(",")]
dev = torch.device("cuda", 0)
for n in sizes:
    pts = np.random.default_rng(42).random((n, 3))
    tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(0)), True, True)
    w = torch.rand((1, n), dtype=torch.float64, device=dev)
    out = torch.zeros_like(w)
    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    for _ in range(5):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, True)
    reps = 200
    torch.cuda.synchronize(); stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, False)
    stream.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, True)
    wall_sync = (time.perf_counter() - t0) / reps * 1e3
    tree.set_profiling(True); tree.phase_ms(reset=True)
    for _ in range(20):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, True)
    ph, cnt = tree.phase_ms(counts=True)
    tree.set_profiling(False)
    st = tree.stats()
    print(json.dumps({"points": n, "depth": st.depth, "cells": st.n_cells, "wall_ms_back_to_back": round(wall, 4),
                      "wall_ms_each_synchronised": round(wall_sync, 4), "sum_of_phase_ms": round(sum(ph.values()) / 20, 4),
                      "phase_intervals_per_matvec": sum(cnt.values()) // 20,
                      "phases_ms": {k: round(v / 20, 4) for k, v in ph.items() if v > 0}}))
