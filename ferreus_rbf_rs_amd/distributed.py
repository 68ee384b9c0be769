"""Multi-GPU exchange step of the partitioned matvec.

Every rank holds the whole tree and evaluates the potentials of the targets it owns (a
contiguous Morton range of leaves, `FmmTree.set_partition`).  Owned rows are disjoint by
construction, so one all-gather of the owned values (padded to the largest share) completes the
matvec on every rank; with backend "nccl" this is RCCL over xGMI, with "gloo" the same code runs
on CPU tensors (tests).
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
        rows_t = torch.as_tensor(rows, dtype=torch.int64, device=device)
        self.rows = rows_t
        counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world)]
        dist.all_gather(counts, torch.tensor([rows_t.numel()], dtype=torch.int64, device=device),
                        group=group)
        self.counts = [int(c.item()) for c in counts]
        self.m_max = max(max(self.counts), 1)
        pad = torch.full((self.m_max,), -1, dtype=torch.int64, device=device)
        pad[: rows_t.numel()] = rows_t
        gathered = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(gathered, pad, group=group)
        all_rows = torch.stack(gathered)                       # world x m_max
        self.valid = all_rows.reshape(-1) >= 0
        self.valid_idx = torch.nonzero(self.valid).reshape(-1)     # once: no mask (= no sync) per step
        self.flat_rows = all_rows.reshape(-1)[self.valid_idx]
        self.send = torch.zeros((k, self.m_max), dtype=torch.float64, device=device)
        self.recv = [torch.empty((k, self.m_max), dtype=torch.float64, device=device)
                     for _ in range(self.world)]

    def check_partition(self) -> bool:
        """True when the owned rows of all ranks are a disjoint cover of 0..n_total-1."""
        r = torch.sort(self.flat_rows).values
        return r.numel() == self.n_total and bool(
            torch.equal(r, torch.arange(self.n_total, dtype=torch.int64, device=r.device)))

    def exchange(self, out: torch.Tensor) -> torch.Tensor:
        """out: K x N with this rank's owned columns filled in; on return every column is."""
        self.send[:, : self.rows.numel()] = out.index_select(1, self.rows)
        dist.all_gather(self.recv, self.send, group=self.group)
        stacked = torch.stack(self.recv, dim=1).reshape(self.k, -1)   # K x (world*m_max)
        out.index_copy_(1, self.flat_rows, stacked.index_select(1, self.valid_idx))
        return out
