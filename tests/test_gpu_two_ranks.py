"""SURVEY.md 8(e) with two real processes: the partitioned device matvec + the owned-rows exchange, and
bench.py's own N > 1 launcher, on a one-GPU box (both ranks on cuda:0, exchange over gloo through pinned host
buffers; on an 8-GPU node the same code runs one rank per GPU over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(backend):
    n, k, world = 130000, 2, 2
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), str(n), str(k), backend],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=800)
        assert p.returncode == 0, err.decode()[-2000:]
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    assert sorted(r["rank"] for r in res) == [0, 1]
    assert all(r["cover"] and not r["nan_left"] for r in res), res
    assert sum(r["owned"] for r in res) == n and all(0 < r["owned"] < n for r in res), res
    assert all(r["n_w"] > 0 for r in res)                      # mixed-level tree: M2P / P2L run partitioned too
    assert all(r["coarse_count"] > 0 for r in res)             # the upward pass really was split and exchanged
    assert all(r["err"] < 1e-12 for r in res), res


@pytest.mark.timeout(900)
def test_two_processes_partitioned_matvec_and_exchange_equal_the_single_rank_product():
    _run_two_ranks("gloo")


@pytest.mark.timeout(900)
def test_two_ranks_over_rccl_one_gpu_each():
    """The same two-rank product with the collectives on RCCL (all-reduce of the coarse multipoles, all-gather of the
    owned rows), one GPU per rank: runs on the first box that has two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _run_two_ranks("nccl")


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two rank processes before
    touching the GPU and relays rank 0's line (here with the gloo exchange so that both fit one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "gloo",
                        "--points", "400000", "--steps", "3", "--warmup", "1", "--cpu-baseline", "off",
                        "--configs", "config5", "--config5-points", "300000"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0 and j["scaling"] == "strong"
    assert "target-subtree partition x2" in j["config"]["parallelism"]
    # the N > 1 line is a checked claim: dense rows of the exchanged result, and config 5's workload on the same ranks
    assert j["dense_rows_rel_err"] is not None and j["dense_rows_rel_err"] < 1e-6
    c5 = j["configs"]["config5_spheroidal3_40M"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["ms_per_step"] > 0 and c5["dense_rows_rel_err"] < 1e-5
    assert "Spheroidal3Rbf" in c5["workload"] and c5["workload"].startswith("300000 ")
