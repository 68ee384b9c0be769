#!/usr/bin/env python3
"""Create / use / destroy device-group handles in a loop: device memory and host RSS must come back (streams, events, exchange
buffers, pinned staging, per-part trees): no drift of the device's free memory over the second half of the rounds, no steady growth
of the resident set.  args: [rounds, default 40] [points, default 150000] [parts, default 4]"""
import gc, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil
import torch
import ferreus_rbf_rs_amd as F

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 4
proc = psutil.Process()
rng = np.random.default_rng(1)
pts = rng.random((n, 3))
w = np.asfortranarray(rng.standard_normal((n, 2)))
x = rng.random((20000 * parts, 3))
kp = F.KernelParams(F.KernelType(0))
torch.cuda.init()
free, rss = [], []
for r in range(rounds):
    g = F.FmmTree(pts, 6, kp, True, True, devices=[0] * parts)
    g.set_weights(w)
    y = g.evaluate(w, pts)                      # partitioned at the sources
    z = g.evaluate(w, x)                        # sharded over the parts
    g.set_local_coefficients(w)
    u = g.evaluate_leaves(w, x)
    ug = g.evaluate_leaves_with_gradients(w, x[:30000])
    zg = g.evaluate_with_gradients(w, x[:30000])   # (few enough targets for the first device alone on a group)
    v = g.fast_matrix_vector_product(w[:, 0].copy(), target_indices=np.arange(0, n, 3))
    dw = torch.from_numpy(np.ascontiguousarray(w.T)).cuda()
    out = torch.zeros_like(dw)
    g.matvec_device(dw.data_ptr(), n, 2, out.data_ptr(), n, True)
    del g, dw, out
    gc.collect()
    torch.cuda.empty_cache()
    free.append(torch.cuda.mem_get_info(0)[0])
    rss.append(proc.memory_info().rss)
half = rounds // 2
rec = {"rounds": rounds, "points": n, "parts": parts,
       "device_free_MB_after_round": [round(f / 1e6, 1) for f in (free[0], free[half], free[-1])],
       "host_rss_MB_after_round": [round(m / 1e6, 1) for m in (rss[0], rss[half], rss[-1])],
       "device_drift_MB_second_half": round((free[half] - free[-1]) / 1e6, 2),
       "host_drift_MB_second_half": round((rss[-1] - rss[half]) / 1e6, 2)}
# the host's resident set moves in steps (the allocator maps and trims whole arenas: one step of ~200 MB shows up at a round that
# differs from run to run), so a leak is judged by the MEDIAN growth per round over the second half, not by the difference
inc = sorted(rss[i + 1] - rss[i] for i in range(half, rounds - 1))
rec["host_median_growth_MB_per_round_second_half"] = round(inc[len(inc) // 2] / 1e6, 3)
rec["ok"] = abs(rec["device_drift_MB_second_half"]) < 64 and rec["host_median_growth_MB_per_round_second_half"] < 1.0
print(json.dumps(rec))
sys.exit(0 if rec["ok"] else 1)
