#!/usr/bin/env python3
"""Soak run on the GPU box: odd shapes at scale, each checked against sampled dense rows and for
linearity of the whole pipeline (device-resident matvec)."""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ferreus_rbf_rs_amd as F
from oracle import bbfmm_oracle as O

def clustered(rng, n, d):
    k = 12
    c = rng.random((k, d))
    s = 0.01 + 0.08 * rng.random(k)
    which = rng.integers(0, k, n)
    return np.clip(c[which] + rng.normal(size=(n, d)) * s[which, None], 0.0, 0.999)

cases = [
    ("clustered 3M 3-D p7 Linear", lambda r: clustered(r, 3_000_000, 3), 7, "LinearRbf", 1.0, 1.0, 1),
    ("uniform 7M 3-D p5 Cubic K=3", lambda r: r.random((7_000_000, 3)), 5, "CubicRbf", 1.0, 1.0, 3),
    ("uniform 2M 2-D p10 TPS", lambda r: r.random((2_000_000, 2)), 10, "ThinPlateSplineRbf", 1.0, 1.0, 1),
    ("clustered 1M 3-D p8 Spheroidal5 K=2", lambda r: clustered(r, 1_000_000, 3), 8, "Spheroidal5Rbf", 0.2, 0.1, 2),
    ("line 500k 1-D p12 Linear", lambda r: r.random((500_000, 1)), 12, "LinearRbf", 1.0, 1.0, 1),
    ("shell 4M 3-D p7 Laplacian", lambda r: (lambda v: 0.5 + 0.45 * v / np.linalg.norm(v, axis=1, keepdims=True))(r.normal(size=(4_000_000, 3))), 7, "Laplacian", 1.0, 1.0, 1),
]
for name, gen, order, kernel, br, sill, K in cases:
    rng = np.random.default_rng(zlib.crc32(name.encode()))       # fixed per case (hash() is randomised per process)
    pts = gen(rng)
    n, d = pts.shape
    kid = O.KERNEL_IDS[kernel]
    t0 = time.time()
    tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill), True, True,
                     m2l_shared_basis="--shared-basis" in sys.argv,   # the extensions of DESIGN.md section 5 on the same shapes
                     direct_small_w_leaves="--direct-w-leaves" in sys.argv)
    build = time.time() - t0
    w = torch.rand((K + 1, n), dtype=torch.float64, device="cuda")
    y = torch.zeros_like(w)
    tree.matvec_device(w.data_ptr(), n, K + 1, y.data_ptr(), n, True)
    coef = torch.rand(K + 1, dtype=torch.float64, device="cuda") - 0.5
    wc = (coef[:, None] * w).sum(0, keepdim=True).contiguous()
    yc = torch.zeros_like(wc)
    t0 = time.perf_counter(); tree.matvec_device(wc.data_ptr(), n, 1, yc.data_ptr(), n, True); ms = (time.perf_counter() - t0) * 1e3
    lin = float((yc[0] - (coef[:, None] * y).sum(0)).abs().max() / yc.abs().max())
    idx = rng.choice(n, 48, replace=False)
    ref = O.dense_sum(kid, br, sill, pts[idx], pts, wc.cpu().numpy().T)[:, 0]
    err = float(np.abs(yc[0, idx].cpu().numpy() - ref).max() / np.abs(ref).max())
    s = tree.stats()
    print(json.dumps({"case": name, "build_s": round(build, 2), "ms_k1": round(ms, 1), "depth": s.depth, "cells": s.n_cells,
                      "w_pairs": s.n_w, "x_pairs": s.n_x, "m2l_basis_len": s.m2l_basis_len, "linearity": lin, "rel_err_vs_dense": err}), flush=True)
    del tree
