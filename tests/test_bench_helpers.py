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
    """The default line carries configs 2 (x2), 3, 4 (x2: the reference's kernel and the Gaussian extension kernel,
    SURVEY 8(d)) at N = 1, config 5 at N > 1, and -- VERDICT r04 next #7 -- ONE labelled extension (the shared-basis M2L on
    the headline workload, so that its figure is driver-measured; never the headline); the other extensions are opt-in.
    Every flagged configuration says what it is by its name."""
    flagged = lambda c: bool(c.get("m2l_shared_basis") or c.get("direct_small_w_leaves"))
    names = [c["name"] for c in bench.EXTRA_CONFIGS + bench.EXTENSION_CONFIGS + [bench.CONFIG5]]
    assert len(names) == len(set(names)) == 11
    assert len(bench.EXTRA_CONFIGS) == 7 and not flagged(bench.CONFIG5)
    assert [c["name"] for c in bench.EXTRA_CONFIGS if flagged(c)] == ["extension_shared_basis_linear_10M"]
    # round 6: ONE tuning entry -- the reference's own FmmParams.max_points_per_cell at 512 on the headline workload -- named as such
    assert [c["name"] for c in bench.EXTRA_CONFIGS if c.get("max_points_per_cell")] == ["tuning_max_points_per_cell_512_linear_10M"]
    assert all(flagged(c) and c["name"].startswith("extension_") for c in bench.EXTENSION_CONFIGS)
    assert (bench.CONFIG5["points"], bench.CONFIG5["kernel"]) == (40_000_000, "Spheroidal3Rbf")
    for c in bench.EXTRA_CONFIGS + bench.EXTENSION_CONFIGS + [bench.CONFIG5]:
        assert c["total_sill"] <= c["base_range"]      # KernelParamsBuilder::build asserts this (kernel_helpers.rs:69-70)


class _HeadlineStats:  # the fields of bbfmm_tree_stats bench.py reads (10M-point headline values)
    n_nodes, n_cells, d, n_leaves, n_w, depth, n_v = 343, 298905, 3, 261542, 9216, 6, 52151874
    m2l_flops_k1, wx_tile_bytes_k1, wx_pairs = 1.7298e12, 5.0e8, 1.2e8
    p2p_tile_bytes_k1, p2p_pairs = 9_990_000_000, 10_080_000_000
    m2l_basis_rank = m2l_basis_len = 0


_HEADLINE_MS = {"gather": 0.05, "P2M": 1.06, "M2M": 0.43, "M2L_stage1": 17.3, "M2L_stage2": 15.8, "P2L": 0.2, "L2L": 0.3,
                "P2P": 5.5, "M2P": 0.0, "L2P": 0.95, "scatter": 0.05}


def test_phase_roofline_accounts_every_phase_and_names_the_bound():
    S, ms = _HeadlineStats, _HEADLINE_MS
    pr = bench.phase_roofline(S, 10_000_000, 1, 7, "LinearRbf", ms, 3.1e13)
    assert set(pr) == set(ms) - {"M2P"}                                   # phases that took no time are left out
    assert pr["M2L_stage1"]["bound"] == "mfma" and abs(pr["M2L_stage1"]["frac_fp64"] - 0.636) < 0.01
    assert pr["P2P"]["bound"] == "fp64_valu" and abs(pr["P2P"]["frac_hbm"] - 0.227) < 0.01
    assert pr["gather"]["bound"] == "hbm" and pr["gather"]["flops"] == 0.0
    for e in pr.values():
        assert e["gbps"] > 0 and 0 <= e["frac_hbm"] < 1.5 and 0 <= e["frac_fp64"] < 1.0
    vi = pr["P2P"]["valu_issue"]                                          # every unordered pair once, 15 instructions (round 6: cubic sqrt step)
    assert vi["fp64_valu_instr_per_evaluation"] == 15 and abs(vi["kernel_evaluations"] - 5.045e9) < 1e7
    assert abs(vi["frac_of_measured_fma_rate"] - 5.045e9 * 15 / 5.5e-3 / 3.1e13) < 1e-3


def test_pair_phases_are_quoted_against_the_fp64_vector_roof():
    """a pair phase that dominates (config 2: fused M2P + P2L) reports bound fp64_valu with the
    instruction-issue fraction, and keeps the HBM tile-bytes figure beside it as frac_hbm."""
    S = _HeadlineStats
    ms = dict(_HEADLINE_MS, P2L=40.0)
    dom, r = bench.roofline_of(S, 1_000_000, 1, "Spheroidal3Rbf", ms, 1, None, 3.1e13)
    assert dom == "P2L" and r["bound"] == "fp64_valu" and r["unit"] == "Tinstr/s"
    # `frac` against the NOMINAL issue peak (a device that clocks lower must not look better); the rate the
    # device sustained in the same run stays beside it, labelled
    assert abs(r["frac"] - 1.2e8 * 24 / 40e-3 / (78.6e12 / 2)) < 1e-6 and abs(r["peak"] - 39.3) < 1e-9
    assert abs(r["frac_of_measured_fma_rate"] - 1.2e8 * 24 / 40e-3 / 3.1e13) < 1e-6 and r["peak_is"].startswith("nominal")
    assert r["instr_count_is"].startswith("ISA count")
    _, r8 = bench.roofline_of(S, 1_000_000, 8, "Spheroidal3Rbf", ms, 1, None, 3.1e13)
    assert r8["instr_count_is"].startswith("DERIVED for 8 rhs")
    assert abs(r["frac_hbm"] - 5.0e8 / 40e-3 / 8e12) < 1e-9
    dom, r = bench.roofline_of(S, 10_000_000, 1, "LinearRbf", _HEADLINE_MS, 1)
    assert dom == "M2L_stage1" and r["bound"] == "mfma" and abs(r["frac"] - 0.636) < 0.01 and "frac_hbm" not in r
    # K right-hand sides: one evaluation per unordered pair and pass of <= 4, a row and a column multiply-add per slot
    half = (S.p2p_pairs + 10_000_000) / 2.0
    assert bench.pair_issue(S, 10_000_000, 8, "LinearRbf", "P2P")[:2] == (half, 2 * (13 + 8))
    assert bench.pair_issue(S, 10_000_000, 3, "LinearRbf", "P2P")[:2] == (half, 13 + 8)        # three run the 4-slot instance
    assert bench.pair_issue(S, 10_000_000, 6, "LinearRbf", "P2P")[:2] == (half, 13 + 8 + 13 + 4)
    assert bench.sym_instances(3) == [4] and bench.sym_instances(11) == [4, 4, 4] and bench.sym_instances(1) == [1]
    assert bench.pair_probe_hash() == bench.committed_pair_instructions()["pair_probe_hash"]   # counts belong to kernels.hpp


def _synthetic_detail(world=1):
    """A full detail record as main() builds it, from synthetic statistics: headline + every extra configuration +
    config 3's solve + config 5, with the long fields real runs carry."""
    S, ms = _HeadlineStats, _HEADLINE_MS
    _, roof = bench.roofline_of(S, 10_000_000, 1, "LinearRbf", ms, world, {k: 1 for k in ms}, 3.1e13)
    roof.update(traffic=42384990805.333336, mfma_util_pct=71.48968633333332, counters_from="profiles/r04_final_counters.json")
    cfgs = {}
    for c in bench.EXTRA_CONFIGS + bench.EXTENSION_CONFIGS + [bench.CONFIG5]:
        _, r = bench.roofline_of(S, c["points"], c["nrhs"], c["kernel"], dict(ms, P2L=40.0), world, None, 3.1e13)
        cfgs[c["name"]] = {"workload": "x" * 80, "ms_per_step": 14.312587000313215, "roofline": r, "phase_ms_per_step": ms,
                           "dense_rows_rel_err": 9.955587767114334e-07,
                           "phase_roofline": bench.phase_roofline(S, c["points"], c["nrhs"], 7, c["kernel"], ms, 3.1e13)}
        if c.get("note"):
            cfgs[c["name"]]["note"] = c["note"]
    solve = {"ddm_params": {}, "levels": 4, "setup_s": 2.612345678, "solve_s": 3.1923456, "iterations": 5, "converged": True,
             "stagnated": False, "residual_history": [6.12e-5, 9.97e-6, 2.69e-6, 1.11e-6, 6.96e-7], "max_fit_error_on_sample": 4e-6}
    cfgs["config3_solve_tps_10M_fgmres_schwarz"] = {"workload": "y" * 100, "for_points": solve,
                                                     "reference_defaults": dict(solve, converged=False, stagnated=True)}
    cfgs["broken"] = {"error": "RuntimeError: " + "z" * 500}
    return {"metric": "BBFMM matvecs/s", "value": 24.108183244653507, "unit": "matvecs/s", "n_gpus": world, "steps": 20,
            "warmup": 5, "ms_per_step": 41.479691350104986, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "10000000 uniform 3D points, LinearRbf, order 7, 1 rhs, adaptive sparse tree, ACA eps=1e-7, "
                                   "set_weights + evaluate at the sources", "points": 10_000_000, "kernel": "LinearRbf",
                       "order": 7, "nrhs": 1, "parallelism": "single GPU" if world == 1 else "p" * 150},
            "roofline": roof, "phase_ms_per_step": ms, "tree": {"depth": 6}, "source_hash": bench.source_hash(),
            "phase_roofline": bench.phase_roofline(S, 10_000_000, 1, 7, "LinearRbf", ms, 3.1e13),
            "dense_rows_rel_err": 9.390407276931204e-09, "configs": cfgs,
            "cpu_baseline": {"value": 0.1374, "unit": "matvecs/s", "cores": 32, "kind": "port", "host_cpu_quota": 16.0,
                             "overstates_full_size_by": 1.2962, "measured_full_size_value": 0.106,
                             "sample": "overstates the full-size port 1.3x (0.106 matvecs/s measured once at 10M); 16-CPU cgroup quota, "
                                       "32 threads; " + "s" * 400},
            "cpu_baseline_detail": {"note": "n" * 600},
            "dropin_host_buffers": {"points": 10_000_000, "patched_caller_ms": 46.322698937729, "unchanged_caller_ms": 46.43676499836147,
                                    "unchanged_caller_general_path_ms": 57.23296804353595, "unchanged_caller_took_resident_path": True}}


@pytest.mark.parametrize("world", [1, 8])
def test_the_stdout_line_stays_under_4_kb_and_round_trips(world, tmp_path, monkeypatch):
    """round 3's 20.7 KB line came back from the driver unparsed.  The printed line is built by
    compact_line from the full record; whatever the record carries, the line is one line of < 4096 bytes that
    json.loads reads back, with the contract's keys, `roofline` and `cpu_baseline`."""
    detail = _synthetic_detail(world)
    assert len(json.dumps(detail)) > 20000                                # the record that broke the driver's parser
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r, wfd = os.pipe()
    bench.write_outputs(detail, wfd)
    os.close(wfd)
    text = os.read(r, 1 << 20).decode()
    os.close(r)
    assert text.endswith("\n") and text.count("\n") == 1 and len(text) < 4096
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "dense_rows_rel_err", "source_hash"):
        assert k in line, k
    assert line["n_gpus"] == world and line["config"]["workload"].startswith("10000000 uniform")
    assert set(line["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "mfma_util_pct"}
    assert abs(line["value"] - detail["value"]) < 1e-5 * detail["value"]
    cb = line["cpu_baseline"]
    assert set(cb) == {"value", "unit", "cores", "kind", "sample", "host_cpu_quota", "overstates_full_size_by", "measured_full_size_value"}
    assert len(cb["sample"]) <= 200
    # the two clauses the 200-character cut must never take -- in keys of their own AND at the front of `sample`
    assert cb["host_cpu_quota"] == 16.0 and abs(cb["overstates_full_size_by"] - 1.2962) < 1e-3 and cb["measured_full_size_value"] == 0.106
    assert cb["sample"].startswith("overstates the full-size port 1.3x") and "16-CPU cgroup quota" in cb["sample"]
    # and the unchanged caller's rate beside `value` (host buffers, PCIe inclusive)
    assert abs(line["value_host_buffers"] - 1e3 / 46.43676499836147) < 1e-3 and line["value_host_buffers"] < line["value"]
    assert line["p2p"]["frac_hbm"] < 0.5 and 0 < line["p2p"]["frac_fp64_valu"] < 1
    # round 5: config 4's Gaussian-extension instance and the shared-basis extension ride on the default N = 1 line; the
    # drop-in boundary's host-buffer figures (patched and unchanged caller) as one small entry
    assert {"config4_gaussian_ext_10M_8rhs", "extension_shared_basis_linear_10M"} <= {c["name"] for c in bench.EXTRA_CONFIGS}
    assert {"config4_gaussian_ext_10M_8rhs", "extension_shared_basis_linear_10M"} <= set(line["configs"])
    assert set(line["dropin_host_buffers_ms"]) == {"patched", "unchanged", "unchanged_general_path"}
    assert line["roofline"].get("peak_is") is None or isinstance(line["roofline"]["peak_is"], str)
    for name, c in line["configs"].items():
        if name == "broken":
            assert len(c["error"]) <= 120
        elif name.startswith("config3_solve"):
            assert c["for_points"]["iterations"] == 5 and c["reference_defaults"]["stagnated"] is True
        else:
            assert set(c) - {"note"} == {"ms_per_step", "dense_rows_rel_err", "roofline"} and set(c["roofline"]) == {"kernel", "bound", "frac"}
    # the narrow Gaussian's 1e-2 against its dense rows is the method's error -- said next to the number
    assert "METHOD" in line["configs"]["config4_gaussian_ext_10M_8rhs"]["note"]
    # the full record went to the side file, untouched
    with open(tmp_path / "bench_detail.json") as f:
        full = json.load(f)
    assert full["configs"]["config2_spheroidal3_1M"]["phase_roofline"] and full["cpu_baseline_detail"]["note"]


def test_full_size_cpu_figure_is_the_newest_committed_run(tmp_path, monkeypatch):
    """cpu_baseline quotes the port's one full-size run beside its scaled sample: the newest profiles/r*_cpu_port_full_10M.json."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r03_cpu_port_full_10M.json").write_text(json.dumps({"matvecs_per_s_full_size": 0.0204}))
    (prof / "r05_cpu_port_full_10M.json").write_text(json.dumps({"matvecs_per_s_full_size": 0.0905, "threads": 64}))
    (prof / "r05_cpu_port_full_10M.json.bak").write_text("not json")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.measured_cpu_full_size()
    assert got["matvecs_per_s_full_size"] == 0.0905 and got["file"].endswith("r05_cpu_port_full_10M.json")
    real = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_cpu_port_full_10M.json")))
    assert 0.05 < real["matvecs_per_s_full_size"] < 0.2 and real["threads"] in (32, 64, 128, 256)
