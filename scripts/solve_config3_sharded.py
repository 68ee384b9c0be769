#!/usr/bin/env python3
"""Config 3 with the preconditioner's factors SHARDED over the ranks of a job (bbfmm_schwarz_create_sharded): thin-plate
spline, order 9, linear drift, DDMParams.for_points, FGMRES 20 x 5 to 1e-6 -- every rank runs the (replicated) solve on a
BBFMM_FLAG_DETERMINISTIC tree, holds 1 / world of every fine level's factors and sums the level corrections with the others.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 \
      scripts/solve_config3_sharded.py [points=10000000] [backend=gloo]

backend gloo: the ranks may share one GPU (exchange staged through the host) -- a functional rehearsal, NEVER a scaling
number; nccl: one GPU per rank.  Rank 0 prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
torch.cuda.set_device(local)
os.environ.setdefault("BBFMM_HOST_THREADS", str(max(4, (os.cpu_count() or 8) // world)))
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
else:
    dist.init_process_group("gloo")
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import solvers as S
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
rng = np.random.default_rng(42)
pts = rng.random((n, 3))
vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.5 * pts[:, 2] ** 2
t0 = time.time()
tree = F.FmmTree(pts, 9, F.KernelParams(F.KernelType(1)), True, True, deterministic=True)
t_tree = time.time() - t0
st = InterpolantSettings(1, 3)
t0 = time.time()
pre = SchwarzPreconditioner(tree, pts, st, DDMParams.for_points(n), shard_group=True)
t_setup = time.time() - t0
op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, 0.0)
rhs = np.concatenate([vals, np.zeros(st.basis_size)])
dist.barrier()
t0 = time.time()
x, hist = S.fgmres(op, rhs, pre, None, 20, 5, S.FittingAccuracy(1e-6))
t_solve = time.time() - t0
idx = rng.choice(n, 2000, replace=False)
fit = float(np.abs(op(x)[idx] - vals[idx]).max())
own = [pre.domains_owned(lv) for lv in range(pre.num_levels)]
free, total = torch.cuda.mem_get_info()
rec = {"rank": rank, "world": world, "backend": backend, "points": n, "levels": pre.num_levels,
       "domains_owned_first_total_per_level": own, "factor_gb_this_rank": round(pre.factor_bytes() / 1e9, 2),
       "tree_build_s": round(t_tree, 2), "setup_s": round(t_setup, 2), "solve_s": round(t_solve, 2), "iterations": len(hist),
       "residual_history": [float("%.4e" % r) for _, r in hist], "max_fit_error_on_sample": fit,
       "device_memory_in_use_gb_all_ranks_on_this_gpu": round((total - free) / 1e9, 1),
       "note": "deterministic trees; both ranks on one GPU over gloo is a functional rehearsal, never a scaling number"}
all_recs = [None] * world
dist.all_gather_object(all_recs, rec)
if rank == 0:
    print(json.dumps({"ranks": all_recs}))
dist.barrier()
dist.destroy_process_group()
