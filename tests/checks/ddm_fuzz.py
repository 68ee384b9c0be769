#!/usr/bin/env python3
"""Randomised check of the host domain decomposition (csrc/ddm.cpp) against the restatement of
domain_decomposition.rs (oracle/ddm.py): dimensions 1-3, uniform / clustered / gridded (tied) points, random
DDMParams, and a random BBFMM_DDM_LARGE_DOMAIN so that both the serial and the threaded median split run.
No GPU.  usage: ddm_fuzz.py [cases] [seed]  -> one JSON line per case, summary at the end."""
import json, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from ferreus_rbf_rs_amd.ddm import DDMParams, DDMTree
from test_ddm_tree import _oracle_levels, _same_tree

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
bad = 0
t_start = time.time()
for c in range(cases):
    rng = np.random.default_rng(seed0 + c)
    d = int(rng.integers(1, 4))
    n = int(rng.integers(300, 7000))
    kind = ["uniform", "clustered", "grid", "duplicates"][int(rng.integers(0, 4))]
    if kind == "uniform":
        pts = rng.random((n, d))
    elif kind == "clustered":
        k = int(rng.integers(2, 8)); cen = rng.random((k, d))
        pts = np.clip(cen[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, d)), 0.0, 1.0)
    elif kind == "grid":
        pts = np.round(rng.random((n, d)), int(rng.integers(1, 3))) - 0.5
        pts[:: int(rng.integers(7, 90))] = 0.0
        if d > 0:
            pts[1:: int(rng.integers(7, 90)), 0] = -0.0
    else:
        base = rng.random((max(n // 3, 10), d)); pts = base[rng.integers(0, base.shape[0], n)]
    leaf = int(rng.integers(16, 200))
    prm = DDMParams(leaf, float(rng.choice([0.25, 0.5, 1.0])), float(rng.choice([0.1, 0.125, 0.25, 0.3])),
                    min(int(rng.integers(2 * leaf, 6 * leaf)), n - 1))      # (at least one fine level)
    thr = int(rng.choice([32, 64, 500, 10 ** 9]))
    os.environ["BBFMM_DDM_LARGE_DOMAIN"] = str(thr)
    rec = {"case": c, "d": d, "n": n, "kind": kind, "leaf_threshold": leaf, "overlap_quota": prm.overlap_quota,
           "coarse_ratio": prm.coarse_ratio, "coarse_threshold": prm.coarse_threshold, "large_domain": thr}
    try:
        tree = DDMTree(pts, prm)
        ref = _oracle_levels(pts, prm)
        _same_tree(tree, ref)
        rec.update(ok=True, levels=len(ref), leaves=[len(l.leaf_domains) for l in ref])
    except AssertionError as e:
        bad += 1
        rec.update(ok=False, error=str(e)[:200])
    except Exception as e:              # both sides may reject a degenerate case: they must do so together
        rec.update(ok=None, error=f"{type(e).__name__}: {e}"[:200])
    print(json.dumps(rec), flush=True)
print(json.dumps({"cases": cases, "failures": bad, "seconds": round(time.time() - t_start, 1)}))
sys.exit(1 if bad else 0)
