"""The Schwarz preconditioner with its factors sharded over the ranks of a job (SURVEY.md 8(f)-1 "across 8 x 288 GB";
bbfmm_schwarz_create_sharded): two and three real processes on the one GPU of the box, the level corrections summed over
gloo.  Each level's rows are written by exactly one rank (the internal points of the domains partition the level,
schwarz.rs:96-113), so the sharded apply must equal the unsharded one BIT FOR BIT, and the FGMRES solve with it too."""
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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,kid", [(2, 1), (3, 3)])        # thin-plate spline + linear drift; Spheroidal3, no polynomial
def test_sharded_factors_give_the_unsharded_preconditioner(world, kid):
    n = 60000
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "schwarz_shard_worker.py"), str(n), str(kid)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=800)
        assert p.returncode == 0, err.decode()[-2000:]
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    assert sorted(r["rank"] for r in res) == list(range(world))
    for r in res:
        assert r["apply_equal"], r                                  # one apply: bit for bit (twice)
        assert r["iterations"][0] == r["iterations"][1] and r["history_equal"] and r["solution_equal"], r
        assert r["final_residual"] < 1e-6
    levels = res[0]["levels"]
    assert levels >= 3
    for lv in range(levels - 1):                                     # fine levels: contiguous shares that cover the domains
        shares = sorted((r["owned"][lv][1], r["owned"][lv][0], r["owned"][lv][2]) for r in res)
        total = shares[0][2]
        assert shares[0][0] == 0 and sum(s[1] for s in shares) == total
        assert all(shares[i][0] + shares[i][1] == shares[i + 1][0] for i in range(world - 1))
    assert all(r["owned"][levels - 1][0] == 1 for r in res)         # the coarse domain is replicated
    for r in res:                                                    # and a rank holds about 1 / world of the factors
        assert r["factor_bytes_shard"] < r["factor_bytes_whole"] * (1.0 / world + 0.15)
