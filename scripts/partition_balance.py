#!/usr/bin/env python3
"""Per-rank matvec time of an N-way target partition, ranks run one after another on one GPU (predicts the multi-GPU
step time = max over ranks + the two collectives).  Each rank runs the real split path: its own share of the upward
pass (bbfmm_matvec_partition_upward), then -- on the coarse multipoles summed over all ranks, which stands in for the
all-reduce -- its downward and leaf passes (bbfmm_matvec_partition_finish)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 8]
kernel = sys.argv[3] if len(sys.argv) > 3 else "LinearRbf"
br = 0.1 if kernel.startswith("Spheroidal") else 1.0
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType[kernel], base_range=br, total_sill=br), True, True)
w = torch.rand((1, n), dtype=torch.float64, device="cuda")
y = torch.zeros_like(w)
ref = torch.zeros_like(w)
tree.matvec_device(w.data_ptr(), n, 1, ref.data_ptr(), n, True)
for world in worlds:
    total = None
    for rank in range(world):                                  # the all-reduce, emulated: sum of the partial multipoles
        tree.set_partition(rank, world)
        c = torch.zeros((1, tree.partition_coarse_count()), dtype=torch.float64, device="cuda")
        tree.matvec_partition_upward(w.data_ptr(), n, 1, c.data_ptr())
        torch.cuda.synchronize()
        total = c if total is None else total + c
    res = []
    full = torch.full_like(w, float("nan"))
    for rank in range(world):
        tree.set_partition(rank, world)
        rows = tree.partition_rows()
        scratch = torch.zeros_like(total)

        def step(sync):
            tree.matvec_partition_upward(w.data_ptr(), n, 1, scratch.data_ptr())
            tree.matvec_partition_finish(total.data_ptr(), y.data_ptr(), n, sync)

        step(True)
        tree.set_profiling(True); tree.phase_ms(reset=True)
        t0 = time.perf_counter()
        for _ in range(3):
            step(False)
        step(True)
        ms = (time.perf_counter() - t0) / 4 * 1e3
        ph = tree.phase_ms(); tree.set_profiling(False)
        idx = torch.as_tensor(rows, device="cuda")
        full[:, idx] = y[:, idx]
        res.append({"rank": rank, "rows": int(len(rows)), "ms": round(ms, 2),
                    "phases": {k: round(v / 4, 2) for k, v in ph.items() if v / 4 > 0.1}})
    err = float((full - ref).abs().max() / ref.abs().max())
    print(json.dumps({"world": world, "points": n, "kernel": kernel, "coarse_multipoles_MB": round(total.numel() * 8 / 1e6, 1),
                      "max_ms": max(r["ms"] for r in res), "reassembled_vs_one_rank": err, "ranks": res}), flush=True)
