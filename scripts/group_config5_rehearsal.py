#!/usr/bin/env python3
"""Config 5's size (40M Spheroidal3 points, BASELINE.json configs[4]) through ONE handle over eight LOGICAL parts on the one
GPU of a box: the rehearsal of the north-star's "partitioned across the 8 GPUs of one node" at its full size -- buffers,
index widths, the pinned staging area, the slot exchange and the owned blocks at 40M rows.  A functional check, never a
scaling number (eight parts share one device).

  1. plain handle at 40M: the device-resident product y1, its dense-row error, the host and device memory it takes
  2. as many logical parts as fit ONE device with room to spare (every part holds the whole tree; a box driven out of
     memory is lost), at most eight
  3. eight-part group at 40M: yG against y1 (the same arithmetic in another summation order: 1e-12), dense rows, the
     unchanged caller's sequence on host buffers (set_weights + evaluate at the source rows) against yG, step times

args: [points, default 40M] [parts, default 8]"""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psutil
import torch
import bench
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
KERNEL, RANGE, SILL, ORDER = "Spheroidal3Rbf", 0.1, 0.1, 7
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
proc = psutil.Process()


def host_limit():
    lim = psutil.virtual_memory().available
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            t = f.read().strip()
        if t != "max":
            with open("/sys/fs/cgroup/memory.current") as f:
                lim = min(lim, int(t) - int(f.read()))
    except OSError:
        pass
    return lim


def make(n, parts):
    pts = np.random.default_rng(42).random((n, 3))
    t0 = time.time()
    tree = F.FmmTree(pts, ORDER, F.KernelParams(F.KernelType[KERNEL], base_range=RANGE, total_sill=SILL), True, True,
                     devices=[0] * parts if parts > 1 else None)
    return pts, tree, time.time() - t0


def product(tree, n, w, steps=3):
    out = torch.zeros((1, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, sync=False)
    torch.cuda.synchronize(); stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tree.matvec_device(w.data_ptr(), n, 1, out.data_ptr(), n, sync=False)
    torch.cuda.synchronize(); stream.synchronize()
    return out, (time.perf_counter() - t0) / steps * 1e3


rec = {"points": N, "parts": G, "kernel": KERNEL, "source_hash": bench.source_hash()}
# 1 + 2. the plain handle at the full size, and what it takes: a part of a group holds the same tree and buffers (cell-indexed
# arrays are not cut down to a part's share), so G parts on ONE device take G times that -- on a node every device holds one
rss0, free0 = proc.memory_info().rss, torch.cuda.mem_get_info(0)[0]
pts, tree, t_build = make(N, 1)
w = torch.from_numpy(np.random.default_rng(43).random((1, N))).to(dev)
y1, ms1 = product(tree, N, w)
rss1, free1 = proc.memory_info().rss, torch.cuda.mem_get_info(0)[0]
rec["plain_handle"] = {"ms_per_matvec": round(ms1, 2), "create_s": round(t_build, 2),
                       "dense_rows_rel_err": bench.dense_rows_err(torch, dev, KERNEL, RANGE, SILL, pts, w, y1),
                       "host_GB": round((rss1 - rss0) / 1e9, 2), "device_GB": round((free0 - free1) / 1e9, 2)}
del tree
torch.cuda.empty_cache()
per_part_dev, per_part_host = 1.05 * (free0 - free1) + 1e9, 1.2 * max(rss1 - rss0, 1e9)
fit = int(min(0.85 * torch.cuda.mem_get_info(0)[0] / per_part_dev, 0.6 * host_limit() / per_part_host))
rec["fit"] = {"parts_that_fit_one_device": fit, "host_available_GB": round(host_limit() / 1e9, 1),
              "device_free_GB": round(torch.cuda.mem_get_info(0)[0] / 1e9, 1)}
if fit < 2:
    rec["skipped"] = "not even two parts of this size fit one device"
    print(json.dumps(rec))
    sys.exit(0)
if fit < G:
    G = fit
    rec["parts"] = G
    rec["note"] = "fewer logical parts than asked for: every part holds the whole tree, and they share ONE device here"

# 3. the group
rss0, free0 = proc.memory_info().rss, torch.cuda.mem_get_info(0)[0]
_, tree, t_build = make(N, G)
yG, msG = product(tree, N, w)
scale = float(y1.abs().max())
rec["group"] = {"ms_per_matvec": round(msG, 2), "create_s_all_parts": round(t_build, 2),
                "vs_plain_handle_rel_diff": float((yG - y1).abs().max()) / scale,
                "dense_rows_rel_err": bench.dense_rows_err(torch, dev, KERNEL, RANGE, SILL, pts, w, yG),
                "bounds": [int(b) for b in tree.group_bounds()]}
lib = L.load()
x = np.asfortranarray(pts)
wh = np.ascontiguousarray(w.cpu().numpy().ravel())
yh = np.zeros(N)
bad = ctypes.c_int64(-1)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    rc = lib.bbfmm_set_weights(tree._h, wh.ctypes.data, N, 1, N)
    rc = rc or lib.bbfmm_evaluate(tree._h, wh.ctypes.data, N, 1, N, x.ctypes.data, N, N, yh.ctypes.data, N, ctypes.byref(bad))
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0
rec["group"]["unchanged_caller_host_buffers_ms"] = round(sorted(ts[1:])[1], 2)
rec["group"]["unchanged_caller_path"] = int(tree.last_evaluate_path())
rec["group"]["host_buffers_vs_device_resident_rel_diff"] = float(np.abs(yh - yG.cpu().numpy().ravel()).max()) / scale
# a row subset through the group (matvec_partial): 1M rows drawn from all parts
idx = np.sort(np.random.default_rng(5).choice(N, 1_000_000, replace=False)).astype(np.int64)
res = tree.fast_matrix_vector_product(wh, target_indices=idx)
full = yG.cpu().numpy().ravel()
mask = np.ones(N, bool); mask[idx] = False
rec["group"]["row_subset_rel_diff"] = float(np.abs(res[idx] - full[idx]).max()) / scale
rec["group"]["row_subset_other_rows_zero"] = bool(np.all(res[:N][mask] == 0.0))
rss1, free1 = proc.memory_info().rss, torch.cuda.mem_get_info(0)[0]
rec["group"]["host_GB"] = round((rss1 - rss0) / 1e9, 2)
rec["group"]["device_GB"] = round((free0 - free1) / 1e9, 2)
ok = rec["group"]["vs_plain_handle_rel_diff"] < 1e-12 and rec["group"]["host_buffers_vs_device_resident_rel_diff"] < 1e-12 \
    and rec["group"]["row_subset_rel_diff"] < 1e-12 and rec["group"]["row_subset_other_rows_zero"]
rec["ok"] = bool(ok)
print(json.dumps(rec))
sys.exit(0 if ok else 1)
