"""N > 1 path on CPU: two processes over the gloo backend run the partition bookkeeping and the
all-gather exchange that bench.py uses with RCCL (ferreus_rbf_rs_amd/distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, k, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ferreus_rbf_rs_amd as F
        from ferreus_rbf_rs_amd.distributed import OwnedRowsExchange
        pts = np.random.default_rng(5).random((n, 3))
        tree = F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
        tree.set_partition(rank, world)
        rows = tree.partition_rows()
        ex = OwnedRowsExchange(rows, n, k, torch.device("cpu"))
        ok_cover = ex.check_partition()
        # stand-in potentials: every rank fills only the rows it owns, the rest is garbage
        truth = torch.arange(n, dtype=torch.float64)[None, :] * torch.tensor([[1.0], [-2.5]], dtype=torch.float64)[:k]
        out = torch.full((k, n), float("nan"), dtype=torch.float64)
        out[:, torch.as_tensor(rows)] = truth[:, torch.as_tensor(rows)]
        ex.exchange(out)
        q.put((rank, ok_cover, bool(torch.equal(out, truth)), len(rows)))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, False, repr(e)))


@pytest.mark.timeout(300)
def test_two_rank_exchange_reassembles_the_matvec():
    world, n, k = 2, 30000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert all(r[2] for r in res), res
    assert sum(r[3] for r in res) == n
