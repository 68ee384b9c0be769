#!/usr/bin/env python3
"""The matvec against FmmParams.max_points_per_cell (bbfmm.rs:77-104: the reference's own tuning knob, default 256): a
shallower tree trades M2L work (FP64 matrix pipe, 0.64 of its peak) for near-field work (whole-leaf kernels, 0.84 of the
measured FMA rate).  args: [points=10000000] [kernel=LinearRbf] [leaf sizes=256,512,1024]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ferreus_rbf_rs_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kernel = sys.argv[2] if len(sys.argv) > 2 else "LinearRbf"
leaves = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "256,512,1024").split(",")]
br = 0.1 if kernel.startswith("Spher") else 1.0
dev = torch.device("cuda", 0)
pts = np.random.default_rng(42).random((n, 3))
w = torch.from_numpy(np.random.default_rng(43).random((1, n))).to(dev)
out = torch.zeros_like(w)
for leaf in leaves:
    t0 = time.time()
    tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType[kernel], base_range=br, total_sill=br), True, True,
                     params=F.FmmParams(leaf, F.M2LCompressionType.ACA, 1e-7, 1024))
    build = time.time() - t0
    st = tree.stats()
    for _ in range(2):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, True)
    tree.set_profiling(True); tree.phase_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(5):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, False)
    tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, True)
    ms = (time.perf_counter() - t0) / 6 * 1e3
    ph = tree.phase_ms(); tree.set_profiling(False)
    err = bench.dense_rows_err(torch, dev, kernel, br, br, pts, w, out)
    print(json.dumps({"points": n, "kernel": kernel, "max_points_per_cell": leaf, "depth": st.depth, "leaves": st.n_leaves,
                      "points_per_leaf": round(n / st.n_leaves, 1), "ms_per_matvec": round(ms, 2), "create_s": round(build, 2),
                      "dense_rows_rel_err": err, "phases": {k: round(v / 6, 2) for k, v in ph.items() if v / 6 > 0.05}}), flush=True)
    del tree
