#!/usr/bin/env python3
"""The reference's own example data set as a binary fixture: tests/golden/albatite_SD_points.npz.

Run in the build container (needs /root/reference; the GPU box has only the .npz):
    python tests/golden/make_albatite_fixture.py

Source (values only -- a data file, no source text):
  * datasets/albatite_SD_points.csv: 35,801 rows `X,Y,Z,SignedDistance` (drill-hole samples, coordinates of
    3.3e5 / 7.7e6 / +-4e2), the input of the reference's examples ferreus_rbf/examples/isosurface_spheroidal.rs:71-81 and
    isosurface_linear.rs:71-81 (read by csv_to_point_arrays with a header row).

What the reference does to the rows before it builds its trees (RBFInterpolator::new, rbf.rs:318-355), restated here so
that the fixture holds exactly the rows the solver sees:
  * remove_duplicates (rbf.rs:1430-1467, Params.test_unique = true, config.rs:147): a cutoff distance r with
    |phi(r) - phi(0)| = eps * |phi(h) - phi(0)|, h = the longest side of the bounding box (duplicate_cutoff_distance,
    rbf.rs:1391-1415; root of the residual on [0, h]); rows are visited in order, a row not yet marked is kept and
    every row within r of it in the infinity norm is marked.  The reference finds the neighbours with its KD-tree
    (ferreus_rbf_utils, out of this repository's scope); scipy's cKDTree with p = inf answers the same query.
  * no global trend in the examples: the points stay as they are.
For both example kernels the cutoff is far below the smallest spacing of the data (printed and stored), so all 35,801
rows are kept and `keep` = 0 .. 35,800; the fixture stores the rows, `keep` for each kernel and the numbers that show it."""
import hashlib
import json
import os
import sys

import numpy as np
from scipy.optimize import brentq
from scipy.spatial import cKDTree

REF = os.environ.get("FERREUS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def duplicate_cutoff_distance(phi, h_ref):
    """rbf.rs:1391-1415 (the reference's inverse-quadratic root finder with rtol 1e-12; brentq to the same tolerance)"""
    eps = np.finfo(np.float64).eps
    phi0, phih = phi(0.0), phi(h_ref)
    target = 1.0 * eps * abs(phih - phi0)
    resid = lambda r: abs(phi(r) - phi0) - target
    if resid(h_ref) <= 0.0:
        return h_ref
    return brentq(resid, 0.0, h_ref, xtol=1e-300, rtol=1e-12)


def remove_duplicates(points, cutoff):
    """rbf.rs:1430-1467"""
    tree = cKDTree(points)
    visited = np.zeros(len(points), dtype=bool)
    keep = []
    for i in range(len(points)):
        if visited[i]:
            continue
        keep.append(i)
        visited[tree.query_ball_point(points[i], cutoff, p=np.inf)] = True
    return np.asarray(keep, dtype=np.int64)


def main():
    from oracle import bbfmm_oracle as O
    O.build_passes()
    path = os.path.join(REF, "datasets", "albatite_SD_points.csv")
    raw = open(path, "rb").read()
    rows = np.loadtxt(path, delimiter=",", skiprows=1)
    assert rows.shape == (35801, 4)
    pts = rows[:, :3]
    ext = np.concatenate([pts.min(0), pts.max(0)])                      # get_pointarray_extents
    h = float(np.abs(ext[3:] - ext[:3]).max())
    d_nn = cKDTree(pts).query(pts, k=2, p=np.inf)[0][:, 1]
    meta = {"source": "datasets/albatite_SD_points.csv (35,801 x 4 values; header X,Y,Z,SignedDistance)",
            "csv_sha256": hashlib.sha256(raw).hexdigest(), "rows": int(rows.shape[0]), "longest_side": h,
            "smallest_inf_norm_spacing": float(d_nn.min()), "kernels": {}}
    keeps = {}
    for name, kid, br, sill in (("Spheroidal3Rbf", 3, 50.0, 10.0), ("LinearRbf", 0, 1.0, 1.0)):
        cutoff = duplicate_cutoff_distance(lambda r: O.kernel_phi(kid, r, br, sill), h)
        keep = remove_duplicates(pts, cutoff)
        keeps[name] = keep
        meta["kernels"][name] = {"base_range": br, "total_sill": sill, "duplicate_cutoff": float(cutoff), "kept": int(len(keep))}
        print(name, "cutoff %.3e" % cutoff, "kept", len(keep), "of", len(pts), "(smallest spacing %.3e)" % d_nn.min())
        assert len(keep) == len(pts) and cutoff < 0.1 * d_nn.min()
    np.savez_compressed(os.path.join(HERE, "albatite_SD_points.npz"), rows=rows,
                        keep_spheroidal3=keeps["Spheroidal3Rbf"], keep_linear=keeps["LinearRbf"])
    with open(os.path.join(HERE, "albatite_SD_points.json"), "w") as f:
        json.dump(meta, f, indent=1)
        f.write("\n")
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
