#!/usr/bin/env python3
"""Achieved FP64 rate of the two M2L stages for one configuration (algorithmic flops = 2*n*r per V pair and
stage, TreeStats.m2l_flops_k1 / 2).  args: points kernel_id order [nrhs]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
import torch
n = int(sys.argv[1]); kid = int(sys.argv[2]); order = int(sys.argv[3]); K = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
st = tree.stats()
w = torch.randn(K, n, dtype=torch.float64, device="cuda"); out = torch.empty_like(w)
for _ in range(2): tree.matvec_device(w.data_ptr(), n, K, out.data_ptr(), n)
tree.set_profiling(True); tree.phase_ms(reset=True)
reps = 5
for _ in range(reps): tree.matvec_device(w.data_ptr(), n, K, out.data_ptr(), n)
ph = {k: v / reps for k, v in tree.phase_ms().items()}
fl = st.m2l_flops_k1 * K / 2.0
print(json.dumps({"points": n, "kernel": kid, "order": order, "nrhs": K, "v_pairs": st.n_v, "stage_flops": fl,
                  "stage1_ms": round(ph["M2L_stage1"], 3), "stage1_tflops": round(fl / ph["M2L_stage1"] * 1e-9, 2),
                  "stage2_ms": round(ph["M2L_stage2"], 3), "stage2_tflops": round(fl / ph["M2L_stage2"] * 1e-9, 2),
                  "matvec_ms": round(sum(ph.values()), 2), "phases": {k: round(v, 3) for k, v in ph.items() if v > 0.01}}))
