"""bench.py's host-side pieces that need no GPU: the plain-torch dense rows it checks every configuration with
(restated kernel formulas) against the oracle's kernels, and the rule that committed counters are only quoted
for the sources they were taken on."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import bbfmm_oracle as O

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("name,kid,br,sill", [("LinearRbf", 0, 1.0, 1.0), ("ThinPlateSplineRbf", 1, 1.0, 1.0),
                                               ("Spheroidal3Rbf", 3, 0.1, 0.1), ("Spheroidal3Rbf", 3, 0.7, 0.4),
                                               ("MultiquadricExt", 101, 0.1, 0.1)])
def test_dense_rows_torch_equals_the_oracle_dense_sum(name, kid, br, sill):
    rng = np.random.default_rng(3)
    pts = rng.random((4000, 3))
    pts[7] = pts[3]                                   # a coincident pair: phi(0)
    w = rng.standard_normal((2, 4000))
    idx = np.array([3, 7, 100, 3999])
    got = bench.dense_rows_torch(torch, name, br, sill, torch.from_numpy(pts[idx]), torch.from_numpy(pts), torch.from_numpy(w))
    want = O.dense_sum(kid, br, sill, pts[idx], pts, w.T.copy())
    assert np.abs(got.numpy() - want).max() <= 1e-12 * np.abs(want).max()
    assert bench.dense_rows_torch(torch, "CubicRbf", 1.0, 1.0, torch.zeros(1, 3), torch.zeros(2, 3), torch.zeros(1, 2)) is None


def test_committed_counters_are_only_quoted_for_the_sources_they_were_taken_on(tmp_path, monkeypatch):
    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "ferreus_rbf_rs_amd" / "csrc")
    (tmp_path / "ferreus_rbf_rs_amd" / "csrc" / "a.hip").write_text("kernel v1")
    h1 = bench.source_hash()
    (prof / "r09_counters.json").write_text(json.dumps({"source_hash": h1, "workload": [10, "LinearRbf", 7, 1],
                                                        "per_launch_bytes": {"P2P": 5.0}}))
    assert bench.committed_counters((10, "LinearRbf", 7, 1))["per_launch_bytes"]["P2P"] == 5.0
    assert bench.committed_counters((11, "LinearRbf", 7, 1)) is None          # another workload
    (tmp_path / "ferreus_rbf_rs_amd" / "csrc" / "a.hip").write_text("kernel v2")
    assert bench.source_hash() != h1
    assert bench.committed_counters((10, "LinearRbf", 7, 1)) is None          # the kernels changed: never stale numbers


def test_extra_configs_name_baseline_json_configs():
    """The default line carries the reference's arithmetic only: configs 2 (x2), 3, 4 at N = 1, config 5 at N > 1;
    the extensions are opt-in and say what they are."""
    flagged = lambda c: bool(c.get("m2l_shared_basis") or c.get("direct_small_w_leaves"))
    names = [c["name"] for c in bench.EXTRA_CONFIGS + bench.EXTENSION_CONFIGS + [bench.CONFIG5]]
    assert len(names) == len(set(names)) == 9
    assert len(bench.EXTRA_CONFIGS) == 4 and not any(flagged(c) for c in bench.EXTRA_CONFIGS + [bench.CONFIG5])
    assert all(flagged(c) and c["name"].startswith("extension_") for c in bench.EXTENSION_CONFIGS)
    assert (bench.CONFIG5["points"], bench.CONFIG5["kernel"]) == (40_000_000, "Spheroidal3Rbf")
    for c in bench.EXTRA_CONFIGS + bench.EXTENSION_CONFIGS + [bench.CONFIG5]:
        assert c["total_sill"] <= c["base_range"]      # KernelParamsBuilder::build asserts this (kernel_helpers.rs:69-70)


def test_phase_roofline_accounts_every_phase_and_names_the_bound():
    class S:  # the fields of bbfmm_tree_stats the function reads (10M-point headline values)
        n_nodes, n_cells, d, n_leaves, n_w = 343, 298905, 3, 261542, 9216
        m2l_flops_k1, wx_tile_bytes_k1, wx_pairs = 1.7298e12, 5.0e8, 1.2e8
        p2p_tile_bytes_k1, p2p_pairs = 9.99e9, 1.008e10
    ms = {"gather": 0.05, "P2M": 1.06, "M2M": 0.43, "M2L_stage1": 17.3, "M2L_stage2": 15.8, "P2L": 0.2, "L2L": 0.3,
          "P2P": 5.5, "M2P": 0.0, "L2P": 0.95, "scatter": 0.05}
    pr = bench.phase_roofline(S, 10_000_000, 1, 7, "LinearRbf", ms, sym_pairs=True)
    assert set(pr) == set(ms) - {"M2P"}                                   # phases that took no time are left out
    assert pr["M2L_stage1"]["bound"] == "mfma" and abs(pr["M2L_stage1"]["frac_fp64"] - 0.636) < 0.01
    assert pr["P2P"]["bound"] == "fp64_valu" and abs(pr["P2P"]["frac_hbm"] - 0.227) < 0.01
    assert pr["gather"]["bound"] == "hbm" and pr["gather"]["flops"] == 0.0
    for e in pr.values():
        assert e["gbps"] > 0 and 0 <= e["frac_hbm"] < 1.5 and 0 <= e["frac_fp64"] < 1.0
