#!/usr/bin/env python3
"""Create / apply / destroy Schwarz preconditioners (and the trees under them) in a loop: device memory must come back.
args: [rounds, default 24] [points, default 60000]"""
import gc, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil
import torch
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000
rng = np.random.default_rng(3)
pts = rng.random((n, 3))
st = InterpolantSettings(0, 3, drift=1)
proc = psutil.Process()
torch.cuda.init()
free, rss = [], []
for r in range(rounds):
    tree = F.FmmTree(pts, 6, F.KernelParams(F.KernelType(0)), True, True)
    pre = SchwarzPreconditioner(tree, pts, st, DDMParams(128, 0.5, 0.125, 512))
    v = rng.standard_normal(n + pre.basis_size)
    v[n:] = 0.0
    z = pre(v)
    z = pre(v)
    del pre, tree
    gc.collect()
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info(0)[0])
    rss.append(proc.memory_info().rss)
half = rounds // 2
inc = sorted(rss[i + 1] - rss[i] for i in range(half, rounds - 1))
rec = {"rounds": rounds, "points": n, "device_free_MB_after_round": [round(f / 1e6, 1) for f in (free[0], free[half], free[-1])],
       "device_drift_MB_second_half": round((free[half] - free[-1]) / 1e6, 2),
       "host_median_growth_MB_per_round_second_half": round(inc[len(inc) // 2] / 1e6, 3)}
rec["ok"] = abs(rec["device_drift_MB_second_half"]) < 32 and rec["host_median_growth_MB_per_round_second_half"] < 1.0
print(json.dumps(rec))
sys.exit(0 if rec["ok"] else 1)
