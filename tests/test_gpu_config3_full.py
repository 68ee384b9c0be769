"""BASELINE.json config 3 at FULL size in the driver-run suite: 10M uniform points, thin-plate
spline, order 9, linear drift, FGMRES 20 x 5 right-preconditioned by the multi-level Schwarz sweep
(ferreus_rbf/src/rbf.rs:523-574, iterative_solvers.rs:38-173, preconditioning/schwarz.rs:32-79) on one MI355X.

Leg 1, DDMParams.for_points (the labelled extension that keeps three fine levels): converges to 1e-6 relative in at most
six iterations, the residual history stays within 5 % of tests/golden/config3_10M_history.json (this repository's own
run, a REGRESSION fixture -- tests/golden/make_config3_10M_history.py), the fitted values reproduce the data on 2,000
sampled rows to 1e-5.  Leg 2, the reference's default DDMParams (config.rs:60-69: five levels, 2.4k coarse points for
16k level-0 domains): two restart cycles stagnate -- the flag the bench records -- at the fixture's residuals.
That the stagnation is the sweep's and not the device code's is shown against the restatement at the sizes the
restatement reaches: tests/test_gpu_configs.py (8,000 points, dense) and tests/checks/config3_midsize_history.py
(200k points, oracle FMM; profiles/r04_config3_midsize_history.json)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(ROOT, "tests", "golden", "config3_10M_history.json")


@pytest.mark.timeout(900)
def test_config3_ten_million_points_fgmres_schwarz_full_size():
    import ferreus_rbf_rs_amd as F
    sys.path.insert(0, ROOT)
    import bench
    with open(GOLDEN) as f:
        gold = json.load(f)
    rec = bench.run_config3_solve(F, 10_000_000, defaults_outer=2)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config3_10M_solve_in_suite.json"), "w") as f:
            json.dump(rec, f, indent=1)

    a, g = rec["for_points"], gold["for_points"]
    assert a["levels"] == g["levels"] == 4 and a["ddm_params"] == g["ddm_params"]
    assert a["converged"] and a["iterations"] <= 6, a
    assert a["residual_history"][-1] <= 1e-6
    assert a["iterations"] == g["iterations"]
    np.testing.assert_allclose(a["residual_history"], g["residual_history"], rtol=0.05)
    assert a["max_fit_error_on_sample"] < 1e-5, a["max_fit_error_on_sample"]

    b, h = rec["reference_defaults"], gold["reference_defaults"]
    assert b["ddm_params"] == {"leaf_threshold": 1024, "overlap_quota": 0.5, "coarse_ratio": 0.125, "coarse_threshold": 4096}
    assert b["levels"] == h["levels"] == 5
    assert b["iterations"] == 10 and not b["converged"] and b["stagnated"], b          # two cycles of five, no progress
    assert b["residual_history"][-1] > 1e-4
    np.testing.assert_allclose(b["residual_history"], h["residual_history"], rtol=0.05)
