#!/usr/bin/env python3
"""One BASELINE.json config on the GPU: matvec timing + sampled rows against the dense direct sum.

  python tests/checks/check_config.py --points 10000000 --kernel ThinPlateSplineRbf --order 9 --nrhs 1
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--kernel", default="LinearRbf")
ap.add_argument("--order", type=int, default=7)
ap.add_argument("--nrhs", type=int, default=1)
ap.add_argument("--base-range", type=float, default=1.0)
ap.add_argument("--total-sill", type=float, default=1.0)
ap.add_argument("--samples", type=int, default=64)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--shared-basis", action="store_true", help="extension BBFMM_FLAG_M2L_SHARED_BASIS")
ap.add_argument("--direct-w-leaves", action="store_true", help="extension BBFMM_FLAG_DIRECT_SMALL_W_LEAVES")
a = ap.parse_args()

import torch
import ferreus_rbf_rs_amd as F
from oracle import bbfmm_oracle as O

N, K = a.points, a.nrhs
kid = O.KERNEL_IDS[a.kernel]
pts = np.random.default_rng(42).random((N, 3))
t0 = time.time()
tree = F.FmmTree(pts, a.order, F.KernelParams(F.KernelType(kid), base_range=a.base_range, total_sill=a.total_sill), True, True,
               m2l_shared_basis=a.shared_basis, direct_small_w_leaves=a.direct_w_leaves)
build = time.time() - t0
dev = torch.device("cuda")
w_h = np.random.default_rng(43).random((K, N))
w = torch.from_numpy(w_h).to(dev)
y = torch.zeros_like(w)
tree.matvec_device(w.data_ptr(), N, K, y.data_ptr(), N, True)
torch.cuda.synchronize()
tree.set_profiling(True); tree.phase_ms(reset=True)
t0 = time.perf_counter()
for _ in range(a.steps):
    tree.matvec_device(w.data_ptr(), N, K, y.data_ptr(), N, False)
tree.matvec_device(w.data_ptr(), N, K, y.data_ptr(), N, True)
ms = (time.perf_counter() - t0) / (a.steps + 1) * 1e3
ph = tree.phase_ms()
yh = y.cpu().numpy()
idx = np.random.default_rng(7).choice(N, a.samples, replace=False)
ref = O.dense_sum(kid, a.base_range, a.total_sill, pts[idx], pts, w_h.T).T   # C/OpenMP dense rows
err = np.abs(yh[:, idx] - ref).max() / np.abs(ref).max()
s = tree.stats()
m2l_ms = (ph.get("M2L_stage1", 0) + ph.get("M2L_stage2", 0)) / (a.steps + 1)
print(json.dumps({"config": vars(a), "ms_per_matvec": ms, "m2l_flops_k1": s.m2l_flops_k1, "v_pairs": s.n_v,
                  "m2l_tflops_algorithmic": (s.m2l_flops_k1 * K / (m2l_ms * 1e-3) * 1e-12) if m2l_ms > 0 else None, "matvecs_per_s": 1e3 / ms, "build_s": build,
                  "rel_err_vs_dense_sampled": err, "depth": s.depth, "cells": s.n_cells, "leaves": s.n_leaves,
                  "phase_ms": {k: round(v / (a.steps + 1), 3) for k, v in ph.items() if v > 0}}))
