#!/usr/bin/env python3
"""Device-side cost of the potentials exchange around the all-gather, 10M points, K = 1, on one GPU (the collective itself
needs the node): row-index version (OwnedRowsExchange: index_select of the owned rows, index_select + index_copy_ of all
rows) against the sorted-block version (a contiguous copy in finish_sorted, one scatter over the tree's permutation)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd.distributed import OwnedRowsExchange

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, K, world = 10_000_000, 1, 8
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(0)), True, True)
stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
out = torch.zeros((K, n), dtype=torch.float64, device=dev)
res = {}
# row-index version at world 1 (all rows owned: the index passes touch all 10M rows, as they do on every rank of 8)
rows = []
for r in range(world):                      # the rows in the order the ranks own them (the tree's permutation)
    tree.set_partition(r, world); rows.append(tree.partition_rows())
tree.set_partition(0, 1)
x = OwnedRowsExchange(np.concatenate(rows), n, K, dev)
with torch.cuda.stream(stream):
    for _ in range(3): x.exchange(out)
    stream.synchronize(); t0 = time.perf_counter()
    for _ in range(20): x.exchange(out)
    stream.synchronize(); res["row_index_exchange_ms_incl_one_rank_all_gather"] = (time.perf_counter() - t0) / 20 * 1e3
    stream.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dist.all_gather_into_tensor(x.recv.view(-1), x.send.view(-1))
    stream.synchronize(); res["one_rank_all_gather_of_10M_ms"] = (time.perf_counter() - t0) / 20 * 1e3
# sorted blocks: the scatter of 8 gathered blocks
tree.set_partition(0, world)
b = tree.partition_bounds(); m_max = int(np.diff(b).max())
recv = torch.rand((world, K, m_max), dtype=torch.float64, device=dev)
for _ in range(3): tree.partition_scatter(recv.data_ptr(), 0, world, m_max, K, out.data_ptr(), n)
stream.synchronize(); t0 = time.perf_counter()
for _ in range(20): tree.partition_scatter(recv.data_ptr(), 0, world, m_max, K, out.data_ptr(), n)
stream.synchronize(); res["sorted_blocks_scatter_ms"] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(res))
dist.destroy_process_group()
