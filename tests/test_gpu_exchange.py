"""The exchange step of the partitioned matvec on real device tensors: a one-rank RCCL group runs
exactly the code path bench.py uses for N > 1 (partition, OwnedRowsExchange on the handle's stream)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_exchange_matches_unpartitioned():
    import torch
    import torch.distributed as dist
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd.distributed import OwnedRowsExchange

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, K = 200_000, 2
        pts = np.random.default_rng(3).random((n, 3))
        tree = F.FmmTree(pts, 6, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
        w = torch.rand((K, n), dtype=torch.float64, device=dev)
        ref = torch.zeros_like(w)
        tree.matvec_device(w.data_ptr(), n, K, ref.data_ptr(), n, True)
        tree.set_partition(0, 1)
        rows = tree.partition_rows()
        assert len(rows) == n
        x = OwnedRowsExchange(rows, n, K, dev)
        assert x.check_partition()
        out = torch.zeros_like(w)
        stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
        for _ in range(3):
            tree.matvec_device(w.data_ptr(), n, K, out.data_ptr(), n, sync=False)
            with torch.cuda.stream(stream):
                x.exchange(out)
        torch.cuda.synchronize()
        stream.synchronize()
        # (two rhs share one kernel evaluation per unordered near-field pair since round 4: f64 atomics, so the runs
        # agree to summation order, not bit for bit; BBFMM_FLAG_DETERMINISTIC is the bitwise path)
        assert (out - ref).abs().max() / ref.abs().max() < 1e-12
        # a 2-way split reassembled through the same exchange object type (ranks run in turn)
        parts = []
        for r in range(2):
            tree.set_partition(r, 2)
            o = torch.zeros_like(w)
            tree.matvec_device(w.data_ptr(), n, K, o.data_ptr(), n, True)
            parts.append((tree.partition_rows(), o))
        full = torch.zeros_like(w)
        for rws, o in parts:
            idx = torch.as_tensor(rws, device=dev)
            full[:, idx] = o[:, idx]
        assert (full - ref).abs().max() / ref.abs().max() < 1e-12
        # the split upward pass: every rank's partial coarse multipoles summed (what the all-reduce does), then
        # every rank's downward + leaf pass on the sum -- ranks run in turn on the one GPU, worlds 2 and 5
        for world in (2, 5):
            partial = []
            for r in range(world):
                tree.set_partition(r, world)
                cnt = tree.partition_coarse_count()
                assert cnt > 0
                c = torch.zeros((K, cnt), dtype=torch.float64, device=dev)
                tree.matvec_partition_upward(w.data_ptr(), n, K, c.data_ptr())
                torch.cuda.synchronize()
                stream.synchronize()
                partial.append(c)
            total = torch.stack(partial).sum(0).contiguous()
            dist.all_reduce(total)                                   # one-rank RCCL group: the collective itself runs
            torch.cuda.synchronize()
            full = torch.full_like(w, float("nan"))
            for r in range(world):
                tree.set_partition(r, world)
                scratch = torch.zeros_like(total)
                tree.matvec_partition_upward(w.data_ptr(), n, K, scratch.data_ptr())   # this rank's fine levels again
                o = torch.zeros_like(w)
                tree.matvec_partition_finish(total.data_ptr(), o.data_ptr(), n, True)
                idx = torch.as_tensor(tree.partition_rows(), device=dev)
                full[:, idx] = o[:, idx]
            assert not torch.isnan(full).any()
            assert (full - ref).abs().max() / ref.abs().max() < 1e-12, world
            # the same through the sorted-order blocks: every rank's owned potentials as one contiguous block, the
            # blocks side by side as the all-gather leaves them, one scatter over the tree's permutation
            tree.set_partition(0, world)
            bounds = tree.partition_bounds()
            assert tree.partition_world() == world and len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == n
            m_max = int(np.diff(bounds).max())
            recv = torch.full((world, K, m_max), float("nan"), dtype=torch.float64, device=dev)
            for r in range(world):
                tree.set_partition(r, world)
                assert tree.partition_rank() == r and len(tree.partition_rows()) == bounds[r + 1] - bounds[r]
                scratch = torch.zeros_like(total)
                tree.matvec_partition_upward(w.data_ptr(), n, K, scratch.data_ptr())
                tree.matvec_partition_finish_sorted(total.data_ptr(), recv[r].data_ptr(), m_max)
                stream.synchronize()
            full2 = torch.full_like(w, float("nan"))
            tree.partition_scatter(recv.data_ptr(), 0, world, m_max, K, full2.data_ptr(), n)
            stream.synchronize()
            assert not torch.isnan(full2).any()
            assert torch.equal(full2, full) or (full2 - full).abs().max() / ref.abs().max() < 1e-13, world
            with pytest.raises(ValueError):
                tree.partition_scatter(recv.data_ptr(), 1, world, m_max, K, full2.data_ptr(), n)   # parts beyond the world
        tree.set_partition(0, 1)
        # the whole N > 1 step on the one-rank RCCL group: all-reduce on the side stream between the library's events,
        # all-gather on the handle's stream, no host synchronisation in between, several steps back to back.  The
        # group has one rank, so the sum holds rank 0's share of the coarse multipoles only: the reference is the same
        # two calls made one after the other with host synchronisation (stream and event ordering is what is checked)
        from ferreus_rbf_rs_amd.distributed import PartitionedMatvec
        tree.set_partition(0, 2)
        cnt = tree.partition_coarse_count()
        c = torch.zeros((K, cnt), dtype=torch.float64, device=dev)
        tree.matvec_partition_upward(w.data_ptr(), n, K, c.data_ptr())
        torch.cuda.synchronize()
        stream.synchronize()
        o_sync = torch.zeros_like(w)
        tree.matvec_partition_finish(c.data_ptr(), o_sync.data_ptr(), n, True)
        idx = torch.as_tensor(tree.partition_rows(), device=dev)
        pm = PartitionedMatvec(tree, n, K, dev, always_exchange=True)
        assert pm.count == cnt > 0 and not pm.check_partition()      # (half of the rows: not a cover)
        outs = [torch.zeros_like(w) for _ in range(4)]
        for o in outs:
            pm.step(w, o)
        pm.synchronize()
        torch.cuda.synchronize()
        for o in outs:
            assert (o[:, idx] - o_sync[:, idx]).abs().max() / ref.abs().max() < 1e-13
        tree.set_partition(0, 1)
    finally:
        dist.destroy_process_group()


def test_partitioned_six_rhs_mixed_levels_against_the_deterministic_path():
    """with more than four right-hand sides the unordered-pair near field and the fused M2P + P2L kernel
    run in passes of four (second pass: rhs offset k0 = 4), and on a partition the fused kernel writes at an output
    offset (out_off != 0 for every rank but the first).  Six rhs, a mixed-level cloud (W / X lists live), three ranks
    run in turn -- against a BBFMM_FLAG_DETERMINISTIC handle (ordered pairs, separate P2L and M2P, fixed summation
    order) at 1e-12."""
    import torch
    import ferreus_rbf_rs_amd as F

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rng = np.random.default_rng(11)
    n, K, world = 150_000, 6, 3
    pts = np.vstack([rng.random((130_000, 3)), np.clip(rng.normal(size=(20_000, 3)) * 0.05 + 0.5, 0.0, 0.999)])
    kp = F.KernelParams(F.FmmKernelType.LinearRbf)
    det = F.FmmTree(pts, 6, kp, True, True, deterministic=True)
    assert det.stats().n_w > 0
    w = torch.from_numpy(rng.standard_normal((K, n))).to(dev)
    ref = torch.zeros_like(w)
    det.matvec_device(w.data_ptr(), n, K, ref.data_ptr(), n, True)
    ref2 = torch.zeros_like(w)
    det.matvec_device(w.data_ptr(), n, K, ref2.data_ptr(), n, True)
    assert torch.equal(ref, ref2)                                         # the reference path is bitwise reproducible
    del det
    tree = F.FmmTree(pts, 6, kp, True, True)
    one = torch.zeros_like(w)
    tree.matvec_device(w.data_ptr(), n, K, one.data_ptr(), n, True)       # unpartitioned, default kernels, two passes
    assert (one - ref).abs().max() / ref.abs().max() < 1e-12
    partial = []
    for r in range(world):
        tree.set_partition(r, world)
        c = torch.zeros((K, tree.partition_coarse_count()), dtype=torch.float64, device=dev)
        tree.matvec_partition_upward(w.data_ptr(), n, K, c.data_ptr())
        torch.cuda.synchronize()
        torch.cuda.ExternalStream(tree.stream(), device=dev).synchronize()
        partial.append(c)
    total = torch.stack(partial).sum(0).contiguous()
    full = torch.full_like(w, float("nan"))
    for r in range(world):
        tree.set_partition(r, world)
        scratch = torch.zeros_like(total)
        tree.matvec_partition_upward(w.data_ptr(), n, K, scratch.data_ptr())
        o = torch.zeros_like(w)
        tree.matvec_partition_finish(total.data_ptr(), o.data_ptr(), n, True)
        idx = torch.as_tensor(tree.partition_rows(), device=dev)
        full[:, idx] = o[:, idx]
    assert not torch.isnan(full).any()
    err = ((full - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max()   # per right-hand side
    assert err < 1e-12, float(err)
