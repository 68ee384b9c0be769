#!/usr/bin/env python3
"""Matvec time against the number of points (uniform cloud, LinearRbf, order 7, one rhs, device-resident weights):
create time, ms per matvec, points per second, tree depth, 16 dense rows.  -> one JSON line per size."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ferreus_rbf_rs_amd as F
from oracle import bbfmm_oracle as O           # the dense-row checker only

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [50_000, 200_000, 1_000_000, 2_000_000, 5_000_000,
                                                                          10_000_000, 20_000_000, 40_000_000]
for n in sizes:
    pts = np.random.default_rng(42).random((n, 3))
    t0 = time.time()
    tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(0)), True, True)
    t_create = time.time() - t0
    w = torch.rand((1, n), dtype=torch.float64, device="cuda")
    y = torch.zeros_like(w)
    for _ in range(2):
        tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
    steps = 10 if n <= 10_000_000 else 4
    t0 = time.perf_counter()
    for _ in range(steps):
        tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, False)
    tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
    ms = (time.perf_counter() - t0) / (steps + 1) * 1e3
    rows = np.random.default_rng(2).choice(n, 16, replace=False)
    wd = w.cpu().numpy().T.copy()
    yd = O.dense_sum(0, 1.0, 1.0, pts[rows], pts, wd)
    err = float(np.abs(y.cpu().numpy()[0, rows] - yd[:, 0]).max() / np.abs(yd).max())
    st = tree.stats()
    print(json.dumps({"points": n, "create_s": round(t_create, 3), "ms_per_matvec": round(ms, 3),
                      "million_points_per_s": round(n / ms / 1e3, 1), "depth": st.depth, "cells": st.n_cells,
                      "dense_rows_rel_err": err}), flush=True)
    del tree, w, y
