#!/usr/bin/env python3
"""Per-rank matvec time of an N-way target partition, ranks run one after another on one GPU
(predicts the multi-GPU step time = max over ranks + exchange)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 8]
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
w = torch.rand((1, n), dtype=torch.float64, device="cuda")
y = torch.zeros_like(w)
for world in worlds:
    res = []
    for rank in range(world):
        tree.set_partition(rank, world)
        rows = tree.partition_rows()
        tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
        tree.set_profiling(True); tree.phase_ms(reset=True)
        t0 = time.perf_counter()
        for _ in range(3):
            tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, False)
        tree.matvec_device(w.data_ptr(), n, 1, y.data_ptr(), n, True)
        ms = (time.perf_counter() - t0) / 4 * 1e3
        ph = tree.phase_ms(); tree.set_profiling(False)
        res.append({"rank": rank, "rows": int(len(rows)), "ms": round(ms, 2),
                    "phases": {k: round(v / 4, 2) for k, v in ph.items() if v / 4 > 0.3}})
    print(json.dumps({"world": world, "max_ms": max(r["ms"] for r in res), "ranks": res}))
