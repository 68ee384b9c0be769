#!/usr/bin/env python3
"""Random small cases of the two extensions (shared basis, direct small W leaves; one or both per case) against the default path: orders, kernels, dimensions, tolerances,
right-hand sides, compression types -- every (coordinates per cell, column-group plan) combination the chunk plans of
stages 2 / 3 can meet.  Prints one JSON line per case and a summary; exit code 1 on a failure.  args: [cases] [seed]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ferreus_rbf_rs_amd as F

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
bad = 0
seen = set()
for it in range(cases):
    d = int(rng.choice([1, 2, 3, 3, 3]))
    order = int(rng.integers(3, 11 if d == 3 else 15))
    kid = int(rng.choice([0, 1, 2, 3, 8]))           # Linear, TPS, Cubic, Spheroidal3, 1/r^2
    eps = float(10.0 ** -rng.integers(3, min(order, 9) + 1))
    comp = int(rng.choice([1, 2]))
    nrhs = int(rng.choice([1, 1, 2, 5]))
    n = int(rng.integers(8000, 50000))
    pts = rng.random((n, d)) if rng.random() < 0.5 else np.clip(rng.normal(size=(n, d)) * 0.12 + 0.5, 0.0, 0.999)
    pts = np.unique(pts, axis=0)
    n = pts.shape[0]
    br, sill = (0.7, 0.5) if kid == 3 else (1.0, 1.0)
    kp = F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill)
    par = F.FmmParams(int(rng.choice([32, 64, 256])), F.M2LCompressionType(comp), eps, 1024)
    a = F.FmmTree(pts, order, kp, True, True, params=par)
    shared, direct = [(True, False), (False, True), (True, True)][int(rng.integers(0, 3))]
    b = F.FmmTree(pts, order, kp, True, True, params=par, m2l_shared_basis=shared, direct_small_w_leaves=direct)
    w = rng.standard_normal((n, nrhs))
    a.set_weights(w); b.set_weights(w)
    ya, yb = a.evaluate(w, pts), b.evaluate(w, pts)
    a.set_weights(np.abs(w))
    scale = max(np.abs(ya).max(), np.abs(a.evaluate(np.abs(w), pts)).max())
    diff = float(np.abs(yb - ya).max() / scale)
    sb = b.stats()
    ok = diff < 20 * eps and np.isfinite(yb).all()
    bad += not ok
    seen.add((sb.n_nodes, sb.m2l_basis_len))
    print(json.dumps({"shared_basis": shared, "direct_w_leaves": direct, "n_w": [int(a.stats().n_w), int(sb.n_w)], "d": d, "order": order, "kernel": kid, "eps": eps, "compression": comp, "nrhs": nrhs, "n": n,
                      "depth": sb.depth, "nodes": sb.n_nodes, "basis_rank": sb.m2l_basis_rank, "basis_len": sb.m2l_basis_len,
                      "diff_over_eps": round(diff / eps, 3), "ok": bool(ok)}), flush=True)
    del a, b
print(json.dumps({"cases": cases, "failed": bad, "distinct_(nodes, basis_len)": len(seen)}))
sys.exit(1 if bad else 0)
