#!/usr/bin/env python3
"""Regression fixture of BASELINE.json config 3 at full size -> tests/golden/config3_10M_history.json.

  python tests/golden/make_config3_10M_history.py [out.json]          (needs an MI355X; about a minute)

NOT a reference-derived golden vector: the reference cannot be built or imported here, so the file
holds THIS repository's own run of the solve -- 10M uniform points (default_rng(42)), thin-plate spline, order 9, linear
drift, smooth values, FGMRES 20 x 5 right-preconditioned by the Schwarz sweep (iterative_solvers.rs:38-173,
preconditioning/schwarz.rs:32-79), once with DDMParams.for_points and once with the reference's default DDMParams
(config.rs:60-69) for two restart cycles.  tests/test_gpu_config3_full.py re-runs the same function
(bench.run_config3_solve) and holds the residual history to 5 % of this record: a change in the preconditioner, the
partial products or the solver that moves convergence shows up there.  What pins the arithmetic itself are the
8,000-point test against the dense restatement (tests/test_gpu_configs.py) and the mid-size device-vs-oracle
histories (tests/checks/config3_midsize_history.py -> profiles/)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    import ferreus_rbf_rs_amd as F
    rec = bench.run_config3_solve(F, 10_000_000, defaults_outer=2)
    keep = {"workload": rec["workload"], "generated_by": "tests/golden/make_config3_10M_history.py (this repository's own run: "
            "a regression fixture, not a reference output)", "source_hash": bench.source_hash()}
    for label in ("for_points", "reference_defaults"):
        e = rec[label]
        keep[label] = {k: e[k] for k in ("ddm_params", "levels", "iterations", "converged", "stagnated", "max_outer_iterations",
                                         "residual_history", "max_fit_error_on_sample")}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "config3_10M_history.json")
    with open(out, "w") as f:
        json.dump(keep, f, indent=1)
        f.write("\n")
    print(json.dumps({label: {"iterations": keep[label]["iterations"], "history": keep[label]["residual_history"],
                              "setup_s": rec[label]["setup_s"], "solve_s": rec[label]["solve_s"]}
                      for label in ("for_points", "reference_defaults")}))


if __name__ == "__main__":
    main()
