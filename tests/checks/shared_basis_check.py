#!/usr/bin/env python3
"""Shared-basis M2L extension (BBFMM_FLAG_M2L_SHARED_BASIS) against the default path and dense rows.
args: points [kernel_id order]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ferreus_rbf_rs_amd as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
kid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
order = int(sys.argv[3]) if len(sys.argv) > 3 else 7
rng = np.random.default_rng(42)
pts = rng.random((n, 3))
kp = F.KernelParams(F.KernelType(kid), base_range=0.1, total_sill=0.1) if kid >= 3 else F.KernelParams(F.KernelType(kid))
w = torch.tensor(rng.random(n)).cuda().reshape(1, n)
out = {}
ys = {}
for name, flag in (("default", False), ("shared_basis", True)):
    t0 = time.time()
    tree = F.FmmTree(pts, order, kp, True, True, m2l_shared_basis=flag)
    t_create = time.time() - t0
    y = torch.zeros_like(w)
    tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
    tree.set_profiling(True); tree.phase_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(5):
        tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    ph = {k: round(v / 5, 3) for k, v in tree.phase_ms().items() if v / 5 > 0.05}
    ys[name] = y.cpu().numpy().ravel()
    st = tree.stats()
    out[name] = {"create_s": round(t_create, 2), "ms": round(ms, 2), "phases": ph, "m2l_flops_k1": float(st.m2l_flops_k1)}
    del tree
idx = rng.choice(n, 64, replace=False)
wn = w.cpu().numpy().ravel()
dense = np.zeros(64)
for k, i in enumerate(idx):
    r = np.sqrt(((pts - pts[i]) ** 2).sum(1))
    if kid == 0: phi = -r
    elif kid == 1: phi = np.where(r > 0, r * r * np.log(np.where(r > 0, r, 1.0)), 0.0)
    elif kid == 2: phi = r ** 3
    else: phi = None
    dense[k] = (phi * wn).sum() if phi is not None else np.nan
scale = np.abs(ys["default"]).max()
out["shared_vs_default_rel_max"] = float(np.abs(ys["shared_basis"] - ys["default"]).max() / scale)
if not np.isnan(dense).any():
    out["default_vs_dense"] = float(np.abs(ys["default"][idx] - dense).max() / np.abs(dense).max())
    out["shared_vs_dense"] = float(np.abs(ys["shared_basis"][idx] - dense).max() / np.abs(dense).max())
print(json.dumps(out))
