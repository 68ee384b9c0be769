"""The fallbacks DESIGN.md section 11 lists stay alive: every default-on switch flipped in a child process
(the switches are read once per process) against the default path on the same cloud, at 1e-12; and
BBFMM_FLAG_DETERMINISTIC gives bitwise equal results from run to run (the reference's per-target sums have a fixed
order; the default path's f64 atomics do not)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr

pytestmark = pytest.mark.gpu

SWITCHES = [
    ("default", {}, []),
    ("p2p_ordered", {"BBFMM_P2P_SYM": "0"}, []),
    ("wx_separate", {"BBFMM_WX_FUSED": "0"}, []),
    ("variants_off", {"BBFMM_M2L_VARIANTS": "0"}, []),
    ("variants_every_tile", {"BBFMM_M2L_VARIANTS": "1"}, []),
    ("tree_host", {"BBFMM_TREE_DEVICE": "0"}, []),
    ("s2_one_workgroup_per_tile", {"BBFMM_M2L_S2_ZSPLIT": "1"}, []),
    ("s2_no_ksplit", {"BBFMM_M2L_S2_KSPLIT": "1"}, []),
    ("operators_host_fill", {"BBFMM_M2L_ASSEMBLE_HOST": "1"}, []),
    ("evaluate_at_sources_general_path", {"BBFMM_EVAL_SOURCES_FAST": "0"}, []),
    ("near_field_chunk_jobs_for_one_rhs", {"BBFMM_P2P_SYM_LEAF": "0", "BBFMM_WX_SYM_LEAF": "0"}, []),   # round 6: no whole-leaf jobs
    ("near_field_jobs_in_morton_order", {"BBFMM_SYM_JOB_ORDER": "0"}, []),
    ("near_field_no_wave_jobs", {"BBFMM_P2P_SYM_WAVE": "0"}, []),                                       # every leaf in the workgroup kernels
    ("near_field_no_wave_jobs_chunks", {"BBFMM_P2P_SYM_WAVE": "0", "BBFMM_P2P_SYM_LEAF": "0"}, []),
    ("near_field_wave_jobs_in_a_small_tree", {"BBFMM_P2P_SYM_WAVE_MIN": "0"}, []),   # default: a tree this small has no wave jobs
    ("deterministic", {}, ["deterministic"]),
    ("deterministic_again", {}, ["deterministic"]),
]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    d = tmp_path_factory.mktemp("switches")
    res = {}
    for name, env_add, flags in SWITCHES:
        env = {k: v for k, v in os.environ.items() if not k.startswith("BBFMM_")}
        env.update(env_add)
        out = str(d / f"{name}.npz")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py"), out] + flags, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, f"{name}: " + p.stderr.decode()[-2000:]
        res[name] = dict(np.load(out))
    return res


@pytest.mark.timeout(3600)
@pytest.mark.parametrize("name", [s[0] for s in SWITCHES[1:]])
def test_switch_equals_default_path(runs, name):
    ref, got = runs["default"], runs[name]
    assert int(ref["n_w"]) > 0 and int(ref["on_device"]) == 1           # mixed levels, device-built tree by default
    assert relerr(got["y"], ref["y"]) < 1e-12, name
    assert relerr(got["z"], ref["z"]) < 1e-12, name
    assert relerr(got["u"], ref["u"]) < 1e-12, name
    assert int(ref["at_sources"]) == 1
    assert int(got["at_sources"]) == (0 if name == "evaluate_at_sources_general_path" else 1)
    if name == "tree_host":
        assert int(got["on_device"]) == 0
    if name == "variants_off":
        assert int(got["n_variants"]) == 0
    if name == "variants_every_tile":
        assert int(got["n_variants"]) > 0


@pytest.mark.timeout(3600)
def test_deterministic_flag_is_bitwise_reproducible(runs):
    a, b = runs["deterministic"], runs["deterministic_again"]
    assert np.array_equal(a["y"], a["y_again"])                          # same handle, twice
    assert np.array_equal(a["y"], b["y"]) and np.array_equal(a["z"], b["z"])   # another process
    assert np.array_equal(a["u"], b["u"])
