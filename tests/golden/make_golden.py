"""Generates tests/golden/*.json.

The Rust reference can be neither compiled nor imported in this environment (no cargo/rustc;
its dense algebra is the un-vendored faer crate), so no vector can be captured from it.  The
fixtures below are therefore
  (1) dense direct sums y = K(X_t, X_s) W evaluated here in plain numpy straight from the kernel
      formulas in ferreus_rbf_utils/src/rbf_kernels.rs:25-301, non_rbf_kernels.rs:20-156 and
      constants.rs:21-50 (independent of both oracle/passes.c and the HIP kernels), and
  (2) the tree / interaction lists of a small adaptive problem as produced by the oracle's
      restatement of linear_tree.rs (pinned separately by structural invariants in
      tests/test_oracle_fixtures.py).
Run:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

EPS = np.finfo(np.float64).eps
SPH = {3: (0.5000000000, 0.7500000000, 2.6798340586, 0.8734640537, 1),
       5: (0.4082482905, 1.0206207262, 1.5822795750, 0.8575980168, 2),
       7: (0.3535533906, 1.2374368671, 1.2008676644, 0.8494862533, 3),
       9: (0.3162277660, 1.4230249471, 1.0000000000, 0.8445585690, 4)}


def phi(kernel, r2, base_range=1.0, total_sill=1.0):
    r = np.sqrt(r2)
    with np.errstate(divide="ignore", invalid="ignore"):
        if kernel == "LinearRbf":
            return -r
        if kernel == "ThinPlateSplineRbf":
            return np.where(np.abs(r) < EPS, 0.0, r * r * np.log(np.where(r > 0, r, 1.0)))
        if kernel == "CubicRbf":
            return r * r * r
        if kernel.startswith("Spheroidal"):
            ip, slope, scaling, inv_y, pw = SPH[int(kernel[10])]
            s = scaling / base_range
            sr2 = s * s * r2
            t = 1.0 + sr2
            return np.where(sr2 <= ip * ip, total_sill - total_sill * slope * s * r,
                            total_sill * inv_y / (t ** pw * np.sqrt(t)))
        if kernel == "Laplacian":
            return np.where(np.abs(r) < EPS, 0.0, 1.0 / np.where(r > 0, r, 1.0))
        if kernel == "OneOverR2":
            return np.where(np.abs(r) < EPS, 0.0, 1.0 / np.where(r > 0, r * r, 1.0))
        if kernel == "OneOverR4":
            return np.where(np.abs(r) < EPS, 0.0, 1.0 / np.where(r > 0, (r * r) ** 2, 1.0))
        if kernel == "GaussianExt":
            return np.exp(-r2 / base_range ** 2)
        if kernel == "MultiquadricExt":
            return np.sqrt(1.0 + r2 / base_range ** 2)
    raise ValueError(kernel)


def dense(kernel, tgt, src, w, **kw):
    d2 = ((tgt[:, None, :] - src[None, :, :]) ** 2).sum(-1)
    return phi(kernel, d2, **kw) @ w


def main():
    from oracle import bbfmm_oracle as O
    rng = np.random.default_rng(20260101)
    out = {}
    # (1) dense sums for every kernel, 3-D, 2 rhs
    n, m = 300, 40
    src = rng.random((n, 3))
    tgt = np.vstack([src[:20], rng.random((m - 20, 3))])     # includes coincident points (r = 0)
    w = rng.random((n, 2))
    cases = []
    for name, kw in [("LinearRbf", {}), ("ThinPlateSplineRbf", {}), ("CubicRbf", {}),
                     ("Spheroidal3Rbf", dict(base_range=0.5, total_sill=0.4)),
                     ("Spheroidal5Rbf", dict(base_range=0.5, total_sill=0.4)),
                     ("Spheroidal7Rbf", dict(base_range=0.5, total_sill=0.4)),
                     ("Spheroidal9Rbf", dict(base_range=0.5, total_sill=0.4)),
                     ("Laplacian", {}), ("OneOverR2", {}), ("OneOverR4", {}),
                     ("GaussianExt", dict(base_range=0.7)), ("MultiquadricExt", dict(base_range=0.3, total_sill=0.3))]:
        y = dense(name, tgt, src, w, **kw)
        cases.append({"kernel": name, "params": kw, "y": y.tolist()})
    out["dense"] = {"sources": src.tolist(), "targets": tgt.tolist(), "weights": w.tolist(), "cases": cases}
    # (2) a small adaptive tree with mixed levels (W/X lists non-empty)
    pts = np.vstack([rng.random((260, 3)), rng.random((180, 3)) * 0.25 + 0.1])
    tree = O.FmmTree(pts, 4, 0, True, True, params=O.FmmParams(20, O.COMPRESSION_ACA, 1e-4, 1024))
    st = tree.structure()
    out["tree"] = {"points": pts.tolist(), "max_points_per_cell": 20,
                   "depth": st["depth"], "center": st["center"], "radius": st["radius"],
                   "tree": [str(k) for k in st["tree"]], "leaves": [str(k) for k in st["leaves"]],
                   "leaf_source_indices": {str(k): v for k, v in st["leaf_source_indices"].items()},
                   **{name: {str(k): [str(x) for x in v] for k, v in st[name].items()} for name in "uvwx"}}
    with open(os.path.join(HERE, "small_cases.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(HERE, "small_cases.json"),
          "nW", sum(len(v) for v in st["w"].values()), "depth", st["depth"])


if __name__ == "__main__":
    main()
