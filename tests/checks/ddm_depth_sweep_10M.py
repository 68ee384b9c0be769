#!/usr/bin/env python3
"""The Schwarz hierarchy at the size where the reference's default DDMParams stagnate (10M points, config 3): device only
(the restatement cannot reach this size; device = restatement to 3e-10 on every hierarchy of tests/checks/ddm_depth_sweep.py).
Thin-plate spline, order 9, linear drift, uniform points, smooth values, FGMRES 4 x 5 to 1e-6 relative; one tree, the
hierarchies one after another: (leaf_threshold 1024, overlap 0.5, coarse_ratio r, coarse_threshold t).

  python tests/checks/ddm_depth_sweep_10M.py [points=10000000] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "ddm_depth_sweep_10M.json")
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd import solvers as S
    from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
    rng = np.random.default_rng(42)
    pts = rng.random((n, 3))
    vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.5 * pts[:, 2] ** 2
    tree = F.FmmTree(pts, 9, F.KernelParams(F.KernelType(1)), True, True)
    st = InterpolantSettings(1, 3, drift=1)
    rhs = np.concatenate([vals, np.zeros(st.basis_size)])
    fp = DDMParams.for_points(n)
    shapes = {
        "reference_defaults": (1024, 0.5, 0.125, 4096),
        "for_points": (fp.leaf_threshold, fp.overlap_quota, fp.coarse_ratio, fp.coarse_threshold),
        "ratio_1_16_coarse_2400": (1024, 0.5, 0.0625, 4096),       # four levels, the defaults' coarse size
        "ratio_0.2": (1024, 0.5, 0.2, 4096),                       # six levels, coarse ~3.2k
        "defaults_one_level_deeper": (1024, 0.5, 0.125, 2000),     # six levels, coarse ~300
        "ratio_0.35": (1024, 0.5, 0.35, 4096),                     # nine levels, coarse ~2.2k
    }
    only = os.environ.get("DEPTH_SWEEP_SHAPES")
    if only:
        shapes = {k: v for k, v in shapes.items() if k in only.split(",")}
    rec = {"points": n, "kernel": "ThinPlateSplineRbf", "order": 9, "drift": "linear", "fgmres": "4 x 5, relative 1e-6", "shapes": {}}
    if os.path.exists(out_path):                                 # a second call with DEPTH_SWEEP_SHAPES adds to the record
        old = json.load(open(out_path))
        if old.get("points") == n:
            rec["shapes"] = {k: v for k, v in old.get("shapes", {}).items() if "error" not in v}
    for label, prm in shapes.items():
        t0 = time.time()
        try:
            pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
        except Exception as e:  # noqa: BLE001
            rec["shapes"][label] = {"ddm_params": list(prm), "error": f"{type(e).__name__}: {e}"[:300]}
            print(label, rec["shapes"][label], flush=True)
            continue
        t_setup = time.time() - t0
        sizes = [len(pre.level_points(lv)) for lv in range(pre.num_levels)]
        op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, 0.0)
        t0 = time.time()
        x, hist = S.fgmres(op, rhs, pre, None, 4, 5, S.FittingAccuracy(1e-6))
        t_solve = time.time() - t0
        rd = [float(h[1]) for h in hist]
        idx = rng.choice(n, 2000, replace=False)
        e = {"ddm_params": [prm[0], prm[1], prm[2], prm[3]], "levels": pre.num_levels, "level_sizes": sizes, "coarse_points": sizes[-1],
             "iterations": len(rd), "final_residual": rd[-1], "converged": bool(rd[-1] <= 1e-6),
             "stagnated": bool(len(rd) >= 10 and rd[-1] > 0.5 * rd[-6]), "history": [float("%.4e" % r) for r in rd],
             "setup_s": t_setup, "solve_s": t_solve, "fit_max_on_sample": float(np.abs(op(x)[idx] - vals[idx]).max())}
        rec["shapes"][label] = e
        print(label, json.dumps(e), flush=True)
        pre.close()
        del pre, op
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(rec, f, indent=1)
            f.write("\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
