#!/usr/bin/env python3
"""Pricing the 'source-sparse' partial products of the Schwarz sweep from their parts, at config 3's
sizes (10M uniform points, thin-plate spline): what a product K c costs when c lives on the 19.7k coarse points only,
 (a) as a direct sum over the coarse points -- timed here through the library's own near-field kernel: a tree over the
     coarse points alone whose eight level-1 cells are leaves (max_points_per_cell > their number) and each other's
     neighbours, so that every pair is near field, evaluated at the level's rows;
 (b) the K Q columns the polynomial projection of the fine-level corrections needs (schwarz.rs:117-126 makes every
     fine-level increment dense: s - Q Q^T s) -- 4 partial products onto the 1.25M level-1 rows, timed as one.
args: [points=10000000]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(42)
pts = rng.random((n, 3))
n1, n2, nc = n // 8, n // 64, n // 512
coarse = pts[rng.choice(n, nc, replace=False)]
out = {"points": n, "coarse_points": nc, "kernel": "ThinPlateSplineRbf"}
t = F.FmmTree(coarse, 4, F.KernelParams(F.KernelType(1)), True, True, extents=[0, 0, 0, 1, 1, 1],
              params=F.FmmParams(nc + 1, 2, 1e-4, 1024))
assert t.stats().n_leaves <= 8 and t.stats().n_v == 0     # the root's children are leaves and each other's neighbours: all near field
w = rng.standard_normal((nc, 1))
t.set_weights(w)
for label, m in (("direct_coarse_sources_to_level1_rows", n1), ("direct_coarse_sources_to_level2_rows", n2)):
    x = pts[rng.choice(n, m, replace=False)]
    t.evaluate(w, x)
    t.set_profiling(True); t.phase_ms(reset=True)
    t0 = time.perf_counter(); t.evaluate(w, x); wall = (time.perf_counter() - t0) * 1e3
    ph = t.phase_ms(); t.set_profiling(False)
    out[label] = {"targets": m, "pairs": m * nc, "p2p_ms": round(ph.get("P2P", 0.0), 2), "wall_ms_with_target_upload_and_grouping": round(wall, 1)}
del t
big = F.FmmTree(pts, 9, F.KernelParams(F.KernelType(1)), True, True)
rows = np.sort(rng.choice(n, n1, replace=False)).astype(np.int64)
wq = rng.standard_normal(n)
big.fast_matrix_vector_product(wq, target_indices=rows)
t0 = time.perf_counter(); big.fast_matrix_vector_product(wq, target_indices=rows); one = (time.perf_counter() - t0) * 1e3
out["one_partial_product_dense_sources_to_level1_rows_ms_host_buffers"] = round(one, 1)
out["k_q_columns_setup_ms"] = round(4 * one, 1)
print(json.dumps(out))
