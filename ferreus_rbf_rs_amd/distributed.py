"""Multi-GPU exchange steps of the partitioned matvec.

Every rank holds the whole tree and all weights, and owns a contiguous Morton range of leaves
(`FmmTree.set_partition`).  Two collectives per matvec, both on preallocated device buffers:

* upward pass: every rank anterpolates its own sources (plus the three-cell halo its V / W lists read at the fine
  levels); the partial multipoles of the coarse levels -- a contiguous 13 MB prefix of M at order 7 -- are summed
  with ONE all-reduce (`PartitionedMatvec`; M2M is linear, so the sum is the whole upward pass of bbfmm.rs:666-772);
* potentials: a rank owns ONE range of the tree's sorted points, so its potentials leave the library as one contiguous
  block (`matvec_partition_finish_sorted`), one all-gather of the blocks (padded to the largest share) brings every
  block to every rank, and the library writes them to their rows in a single pass over its permutation
  (`partition_scatter`) -- `PartitionedMatvec`.  `OwnedRowsExchange` is the same exchange by explicit row indices
  (one gather pass before and one scatter pass after the collective), for callers that hold per-rank row lists.

With backend "nccl" both are RCCL over xGMI.  With a "gloo" group the same bookkeeping runs on CPU tensors (tests), or -- for
device tensors -- stages the owned values through pinned host buffers, which lets two ranks share
one GPU (a functional check of the N > 1 path on a one-GPU box, never a scaling number).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class OwnedRowsExchange:
    """all-gather of per-rank owned rows into a full (K x N) result."""

    def __init__(self, rows, n_total: int, k: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.k = k
        self.n_total = n_total
        device = torch.device(device)
        # collectives run where the backend can: gloo moves host memory
        self.staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
        cdev = torch.device("cpu") if self.staged else device
        rows_t = torch.as_tensor(rows, dtype=torch.int64, device=cdev)
        counts = torch.zeros(self.world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(counts, torch.tensor([rows_t.numel()], dtype=torch.int64, device=cdev), group=group)
        self.counts = [int(c) for c in counts.tolist()]
        self.m_max = max(max(self.counts), 1)
        pad = torch.full((self.m_max,), -1, dtype=torch.int64, device=cdev)
        pad[: rows_t.numel()] = rows_t
        all_rows = torch.empty((self.world, self.m_max), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(all_rows.view(-1), pad, group=group)   # flat in, flat out: every backend takes it
        valid = all_rows >= 0                                           # world x m_max
        # flat element indices, once: recv is (world, k, m_max), out is (k, n_total)
        rk = torch.arange(self.world, device=cdev)[:, None, None]
        kk = torch.arange(k, device=cdev)[None, :, None]
        mm = torch.arange(self.m_max, device=cdev)[None, None, :]
        vmask = valid[:, None, :].expand(self.world, k, self.m_max)
        src = ((rk * k + kk) * self.m_max + mm)[vmask]
        dst = (kk * n_total + all_rows[:, None, :])[vmask]
        self.flat_rows = all_rows[valid]
        self.rows = rows_t.to(device)
        self.src_idx = src.to(device)
        self.dst_idx = dst.to(device)
        self.send = torch.zeros((k, self.m_max), dtype=torch.float64, device=device)
        self.recv = torch.empty((self.world, k, self.m_max), dtype=torch.float64, device=device)
        if self.staged:
            self.h_send = torch.zeros((k, self.m_max), dtype=torch.float64).pin_memory()
            self.h_recv = torch.empty((self.world, k, self.m_max), dtype=torch.float64).pin_memory()

    def check_partition(self) -> bool:
        """True when the owned rows of all ranks are a disjoint cover of 0..n_total-1."""
        r = torch.sort(self.flat_rows).values
        return r.numel() == self.n_total and bool(
            torch.equal(r, torch.arange(self.n_total, dtype=torch.int64, device=r.device)))

    def exchange(self, out: torch.Tensor) -> torch.Tensor:
        """out: K x N (contiguous) with this rank's owned columns filled in; on return every column is."""
        m = self.rows.numel()
        if m:
            self.send[:, :m].copy_(out.index_select(1, self.rows))
        if self.staged:
            self.h_send.copy_(self.send)                                 # device -> pinned host (synchronises)
            dist.all_gather_into_tensor(self.h_recv.view(-1), self.h_send.view(-1), group=self.group)
            self.recv.copy_(self.h_recv)
        else:
            dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1), group=self.group)
        out.view(-1).index_copy_(0, self.dst_idx, self.recv.view(-1).index_select(0, self.src_idx))
        return out


class PartitionedMatvec:
    """One matvec of a partitioned handle: upward (own share) -> all-reduce of the coarse multipoles, beside the near
    field -> downward and leaf pass of the owned targets -> all-gather of the owned potentials (contiguous blocks in the
    tree's sorted order) -> one scatter to the rows.  The kernels and the all-gather are queued on the handle's HIP
    stream (wrapped as a torch ExternalStream so that RCCL orders itself with the kernels), the all-reduce on a second
    stream that the library orders with events; nothing synchronises the host except the gloo staging path."""

    def __init__(self, tree, n_total: int, k: int, device, group=None, always_exchange: bool = False):
        """always_exchange: a one-rank group takes the N > 1 path too (both collectives run, on their streams) instead
        of the plain matvec -- the check of the stream and event ordering that a one-GPU box can make with RCCL; the
        handle's partition may then have more parts than the group has ranks (rank r of the group = part r)."""
        self.tree, self.group, self.k, self.n = tree, group, k, n_total
        device = torch.device(device)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.split = self.world > 1 or always_exchange
        self.staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
        self.count = tree.partition_coarse_count() if self.split else 0
        self.coarse = torch.zeros((k, max(self.count, 1)), dtype=torch.float64, device=device)
        if self.staged:
            self.h_coarse = torch.zeros((k, max(self.count, 1)), dtype=torch.float64).pin_memory()
        self.bounds = None
        if self.split:
            # every part's range of the sorted points; the group's ranks are parts 0 .. world - 1 of the handle's partition
            cdev = torch.device("cpu") if self.staged else device
            mine = torch.tensor([len(tree.partition_rows())], dtype=torch.int64, device=cdev)
            counts = torch.zeros(self.world, dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(counts, mine, group=group)
            self.counts = [int(c) for c in counts.tolist()]
            self.m_max = max(max(self.counts), 1)
            self.bounds = [int(x) for x in tree.partition_bounds()]
            # A verdict about the partition is taken by the whole group or not at all: a rank that raised alone would leave
            # its peers waiting in the next collective.  (min over the ranks of "my handle fits the group")
            enough = torch.tensor([1 if len(self.bounds) - 1 >= self.world else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(enough, op=dist.ReduceOp.MIN, group=group)
            if int(enough.item()) == 0:
                raise ValueError("the partition of a handle in the group has fewer parts than the group has ranks "
                                 "(this rank: %d parts, %d ranks)" % (len(self.bounds) - 1, self.world))
            self.send = torch.zeros((k, self.m_max), dtype=torch.float64, device=device)
            self.recv = torch.empty((self.world, k, self.m_max), dtype=torch.float64, device=device)
            if self.staged:
                self.h_send = torch.zeros((k, self.m_max), dtype=torch.float64).pin_memory()
                self.h_recv = torch.empty((self.world, k, self.m_max), dtype=torch.float64).pin_memory()
        self.stream = torch.cuda.ExternalStream(tree.stream(), device=device) if device.type == "cuda" else None
        # the all-reduce runs on its own stream: the library makes it wait for the packed multipoles only, so that the
        # collective overlaps the near field queued behind the pack, and makes its own stream wait for it in `finish`
        self.comm = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        if device.type == "cuda":
            # the buffers above were created (zero-filled) on torch's current stream; the handle's stream and the
            # communication stream write to them from now on: order both behind the fills explicitly
            cur = torch.cuda.current_stream(device)
            self.stream.wait_stream(cur)
            self.comm.wait_stream(cur)
        if self.split and not always_exchange:
            cdev = torch.device("cpu") if self.staged else device
            ok = torch.tensor([1 if self.check_partition() else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)      # every rank raises, or none does
            if int(ok.item()) == 0:
                raise ValueError("a handle's partition does not match the group (this rank: %s): rank %d of %d holds part "
                                 "%d, bounds %s, row counts %s" % ("ok" if self.check_partition() else "MISMATCH", self.rank,
                                                                   self.world, tree.partition_rank(), self.bounds, self.counts))

    def check_partition(self) -> bool:
        """True when the ranks' shares are the parts of the handle's partition, in order, and cover every row once."""
        if not self.split:
            return True
        b = self.bounds
        return (len(b) == self.world + 1 and b[0] == 0 and b[-1] == self.n
                and all(b[r + 1] - b[r] == self.counts[r] for r in range(self.world))
                and self.tree.partition_rank() == self.rank)

    def all_reduce_coarse(self):
        if self.count == 0 or not self.split:
            return
        if self.staged:
            self.h_coarse.copy_(self.coarse)                             # device -> pinned host (synchronises)
            dist.all_reduce(self.h_coarse, group=self.group)
            self.coarse.copy_(self.h_coarse)
        else:
            dist.all_reduce(self.coarse, group=self.group)

    def all_gather_blocks(self):
        if self.staged:
            self.h_send.copy_(self.send)                                 # device -> pinned host (synchronises)
            dist.all_gather_into_tensor(self.h_recv.view(-1), self.h_send.view(-1), group=self.group)
            self.recv.copy_(self.h_recv)
        else:
            dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1), group=self.group)

    def step(self, w: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """w: K x N weights (device, contiguous, the same on every rank); out: K x N, complete on return (with
        always_exchange on a smaller group: the rows of the group's parts)."""
        if not self.split:
            self.tree.matvec_device(w.data_ptr(), self.n, self.k, out.data_ptr(), self.n, sync=False)
            return out
        comm = self.comm.cuda_stream
        self.tree.matvec_partition_upward(w.data_ptr(), self.n, self.k, self.coarse.data_ptr(), comm)
        with torch.cuda.stream(self.comm):
            self.all_reduce_coarse()
        self.tree.matvec_partition_finish_sorted(self.coarse.data_ptr(), self.send.data_ptr(), self.m_max, comm)
        with torch.cuda.stream(self.stream):
            self.all_gather_blocks()
        self.tree.partition_scatter(self.recv.data_ptr(), 0, self.world, self.m_max, self.k, out.data_ptr(), self.n)
        return out

    def synchronize(self):
        if self.stream is not None:
            self.stream.synchronize()
