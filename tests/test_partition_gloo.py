"""N > 1 path on CPU: two processes over the gloo backend run the partition bookkeeping, the all-gather of the owned
rows and the all-reduce of the split upward pass (on point counts in place of multipoles) that bench.py uses with RCCL
(ferreus_rbf_rs_amd/distributed.py)."""
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
        ok_exchange = bool(torch.equal(out, truth))
        # the split upward pass with the REAL collective: every rank walks its own plan with point counts in place of
        # multipoles (bbfmm_debug_partition_upward_counts), the coarse prefixes are summed by one all-reduce -- what
        # PartitionedMatvec does with the multipoles -- and must equal the whole upward pass on every rank
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_partition_upward import _true_counts
        counts, reads, info = tree.debug_partition_upward_counts()
        n_coarse = int(info[1])
        coarse = torch.from_numpy(counts[:n_coarse].astype(np.float64))
        dist.all_reduce(coarse)
        truth_counts, level = _true_counts(tree, 3)
        sel = level[:n_coarse] >= 1
        ok_upward = n_coarse > 0 and bool(np.array_equal(coarse.numpy()[sel], truth_counts[:n_coarse][sel].astype(np.float64)))
        fine = (level > int(info[0])) & (reads == 1)
        ok_upward = ok_upward and bool(np.array_equal(counts[fine], truth_counts[fine]))
        q.put((rank, ok_cover, ok_exchange and ok_upward, len(rows)))
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


def _mismatch_worker(rank, world, port, n, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ferreus_rbf_rs_amd as F
        from ferreus_rbf_rs_amd.distributed import PartitionedMatvec
        pts = np.random.default_rng(5).random((n, 3))
        tree = F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
        tree.set_partition(0, world)            # rank 1 holds part 0 as well: ITS handle does not match the group
        try:
            PartitionedMatvec(tree, n, 1, torch.device("cpu"))
            verdict = "accepted"
        except ValueError as e:
            verdict = "refused: " + str(e)
        dist.barrier()                          # both ranks are still in step: nobody hangs in a collective the other left
        q.put((rank, verdict))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error: " + repr(e)))


@pytest.mark.timeout(300)
def test_a_partition_mismatch_on_one_rank_is_refused_by_every_rank():
    """the check is per rank (`partition_rank() == rank`), the verdict must be collective -- a rank raising
    alone would leave its peers in the next all-reduce."""
    world, n = 2, 20000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res[0].startswith("refused"), res          # (rank 0 sees the doubled row counts; what matters: it did not go on alone)
    assert res[1].startswith("refused") and "this rank: MISMATCH" in res[1], res
