#!/usr/bin/env python3
"""What does DEPTH do to the Schwarz sweep?  One cloud, one kernel, the reference's leaf threshold;
only the number of levels varies, with the coarse domain held near 3,000 points (>= 2,400: "healthy"), plus one
comparator with a tiny coarse domain at the same leaf threshold.

  python tests/checks/ddm_depth_sweep.py [points=200000] [out.json]          (needs a GPU; tens of minutes of host time)

Thin-plate spline, order 9, linear drift, uniform points, smooth values, FGMRES 4 x 5 to 1e-6 relative -- config 3's
problem at a size the restatement reaches with its own FMM products.  DDMParams (config.rs:60-69) = (leaf_threshold 1024,
overlap_quota 0.5, coarse_ratio r, coarse_threshold 4096); a level is added while more than coarse_threshold points are
active (domain_decomposition.rs:82-100), each level keeps ceil(r x active) points (:165-168), so with
r = (3000 / n)^(1 / (L - 1)) the hierarchy has L levels and a coarse domain of ~3,000 points whatever L:
    L = 3: r = 0.1225 (the reference's 0.125 gives the same shape)    L = 4: r = 0.2466    L = 5: r = 0.35    L = 6: r = 0.4317
Comparator "tiny_coarse": r = 0.125, coarse_threshold 500 -> 4 levels, ~390 coarse points (leaf threshold still 1024).
Device (FGMRES + bbfmm_schwarz_apply) against oracle/solvers.py + oracle/ddm.py + the oracle's products on the product's
M2L factors: same level index sets, histories side by side, their largest relative difference."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "ddm_depth_sweep.json")
    oracle_too = os.environ.get("DEPTH_SWEEP_ORACLE", "1") == "1"
    from oracle import bbfmm_oracle as O
    from oracle import ddm as D
    from oracle import solvers as OS
    import ferreus_rbf_rs_amd as F
    from conftest import inject_product_operators
    from ferreus_rbf_rs_amd import solvers as S
    from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
    rng = np.random.default_rng(42)
    pts = rng.random((n, 3))
    vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.5 * pts[:, 2] ** 2
    kid, order = 1, 9
    ost = D.InterpolantSettings(kid, 3, drift=1)
    m = ost.basis_size
    rhs = np.concatenate([vals, np.zeros(m)])
    tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
    st = InterpolantSettings(kid, 3, drift=1)
    if oracle_too:
        otree = O.FmmTree(pts, order, kid, True, True)
        inject_product_operators(tree, otree)
        tr, sc = D.cheb_cube_scaling_factors(pts)
        mono, ortho = D.orthonormal_poly(pts, ost, tr, sc)
        o_matvec = lambda w: O.fast_matrix_vector_product(otree, w, m, None, mono, 0.0)
        o_partial = lambda w, idx: O.fast_matrix_vector_product(otree, w, m, idx, mono, 0.0)
    coarse_target = 3000.0
    shapes = {}
    for L in (3, 4, 5, 6):
        shapes["levels_%d" % L] = (1024, 0.5, (coarse_target / n) ** (1.0 / (L - 1)), 4096)
    shapes["tiny_coarse_4_levels"] = (1024, 0.5, 0.125, 500)
    only = os.environ.get("DEPTH_SWEEP_SHAPES")
    if only:
        shapes = {k: v for k, v in shapes.items() if k in only.split(",")}
    rec = {"points": n, "kernel": "ThinPlateSplineRbf", "order": order, "drift": "linear", "fgmres": "4 x 5, relative 1e-6",
           "leaf_threshold": 1024, "shapes": {}}
    for label, prm in shapes.items():
        pre = SchwarzPreconditioner(tree, pts, st, DDMParams(*prm))
        sizes = [len(pre.level_points(lv)) for lv in range(pre.num_levels)]
        op = S.RbfSystemOperator(tree, m, pre.monomial_matrix, 0.0)
        t0 = time.time()
        x, hist = S.fgmres(op, rhs, pre, None, 4, 5, S.FittingAccuracy(1e-6))
        rd = [float(h[1]) for h in hist]
        e = {"ddm_params": [prm[0], prm[1], float("%.4f" % prm[2]), prm[3]], "levels": pre.num_levels, "level_sizes": sizes,
             "coarse_points": sizes[-1], "device_iterations": len(rd), "device_final_residual": rd[-1],
             "device_converged": bool(rd[-1] <= 1e-6), "device_history": [float("%.4e" % r) for r in rd],
             "device_solve_s": time.time() - t0,
             "device_fit_max": float(np.abs(op(x)[:n] - vals).max())}
        if oracle_too:
            levels = D.build_ddm_tree(pts, ost, D.DDMParams(*prm))
            assert len(levels) == pre.num_levels
            for lv in range(len(levels)):
                assert np.array_equal(pre.level_points(lv), np.asarray(levels[lv].point_indices)), lv
            pre_o = lambda v: D.schwarz_preconditioner(v, levels, o_partial, ost, ortho)
            t0 = time.time()
            xo, histo = OS.fgmres(o_matvec, rhs, pre_o, None, 4, 5, OS.RELATIVE, 1e-6)
            ro = [float(h[1]) for h in histo]
            k = min(len(rd), len(ro))
            kk = k - 1 if (rd[-1] <= 1e-6 or ro[-1] <= 1e-6) else k      # the last step lands below the tolerance wherever rounding puts it
            e.update({"oracle_iterations": len(ro), "oracle_final_residual": ro[-1], "oracle_history": [float("%.4e" % r) for r in ro],
                      "oracle_solve_s": time.time() - t0,
                      "max_rel_diff_of_histories": float(max(abs(a - b) / b for a, b in zip(rd[:kk], ro[:kk]))) if kk else None})
        del pre, op
        rec["shapes"][label] = e
        print(label, json.dumps(e), flush=True)
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(rec, f, indent=1)
            f.write("\n")
    print("RESULT written", out_path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
