#!/usr/bin/env python3
"""bench.py -- BBFMM matvecs/s on MI355X (BASELINE.json: "BBFMM matvecs/s + achieved HBM GB/s,
10M 3D pts, 1/2/4/8 MI355X").

A "step" is one matvec = set_weights (upward pass) + evaluate at the sources (downward + leaf
pass), exactly what one FGMRES `matvec` closure call does in the reference
(ferreus_rbf/src/rbf.rs:1357-1364), on weights already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; every rank holds the whole tree and owns a contiguous Morton range of
target leaves (strong scaling: the total work is one matvec over all points); the owned potentials
are exchanged with one RCCL all-gather over xGMI per step.

Rank 0 prints ONE JSON line (see the prompt's contract) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Peaks.  HBM: /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec".  The guide
# lists no FP64 MFMA rate; 78.6 TFLOP/s is AMD's public MI355X FP64 matrix (= vector) figure, and
# the bench also reports the rate a bare v_mfma_f64_16x16x4 loop reaches on this device.
HBM_PEAK_GBPS = 8000.0
FP64_MFMA_PEAK_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--kernel", default="LinearRbf")
    ap.add_argument("--order", type=int, default=7)
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--base-range", type=float, default=1.0)
    ap.add_argument("--total-sill", type=float, default=1.0)
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU sample (0: points/64)")
    return ap.parse_args()


def cpu_baseline(args, kernel_id):
    """Times the CPU restatement of the reference algorithm (oracle/, kind "port": C + OpenMP
    passes over a Python-built tree) on a bounded sample: a cloud 64x smaller than the workload,
    which has the same leaf occupancy and list structure two levels shallower; the BBFMM matvec
    is O(N), so the rate is scaled by the point ratio."""
    from oracle import bbfmm_oracle as O
    n_cpu = args.cpu_points or max(20000, args.points // 64)
    rng = np.random.default_rng(42)
    pts = rng.random((n_cpu, 3))
    w = np.random.default_rng(43).random((n_cpu, args.nrhs))
    tree = O.FmmTree(pts, args.order, kernel_id, True, True, base_range=args.base_range,
                     total_sill=args.total_sill)
    tree.set_weights(w)                      # warm-up
    tree.evaluate(w, pts)
    times = []
    t_end = time.time() + 25.0
    while len(times) < 5 and (len(times) < 2 or time.time() < t_end):
        t0 = time.time()
        tree.set_weights(w)
        tree.evaluate(w, pts)
        times.append(time.time() - t0)
    t = float(np.median(times))
    scale = n_cpu / float(args.points)
    return {
        "value": (1.0 / t) * scale,
        "unit": "matvecs/s",
        "cores": int(O.lib().oracle_num_threads()),
        "kind": "port",
        "sample": (f"CPU restatement of the reference algorithm (not the Rust binary): median of "
                   f"{len(times)} matvecs on {n_cpu} uniform points ({t:.3f} s each, same kernel/"
                   f"order/nrhs, same leaf occupancy as the {args.points}-point workload), rate "
                   f"scaled by {n_cpu}/{args.points} (O(N) algorithm)"),
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import ferreus_rbf_rs_amd as F
    from oracle.bbfmm_oracle import KERNEL_IDS
    kernel_id = KERNEL_IDS[args.kernel]

    N, K = args.points, args.nrhs
    # synthetic inputs: i.i.d. uniform [0,1)^3 points, uniform [0,1) weights
    pts = np.random.default_rng(42).random((N, 3))
    t0 = time.time()
    tree = F.FmmTree(pts, args.order, F.KernelParams(F.KernelType(kernel_id), base_range=args.base_range,
                                                     total_sill=args.total_sill), True, True)
    t_build = time.time() - t0
    stats = tree.stats()
    tree.set_partition(rank, world)
    rows = tree.partition_rows()

    w = torch.from_numpy(np.random.default_rng(43).random((K, N))).to(dev)   # K x N, rhs-major
    out = torch.zeros((K, N), dtype=torch.float64, device=dev)

    if world > 1:
        from ferreus_rbf_rs_amd.distributed import OwnedRowsExchange
        xchg = OwnedRowsExchange(rows, N, K, dev)     # owned rows are a disjoint cover: all-gather
        assert xchg.check_partition(), "partition does not cover the targets exactly once"

    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)

    def step():
        # hot path: gather, P2M, M2M, M2L, P2L, L2L, P2P, M2P, L2P, scatter -- all on the handle's stream
        tree.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, sync=False)
        if world > 1:
            # exchange step: owned potentials only (disjoint by construction) -> all-gather
            with torch.cuda.stream(stream):
                xchg.exchange(out)

    def sync():
        torch.cuda.synchronize()
        stream.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    tree.set_profiling(True)
    tree.phase_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    phases, counts = tree.phase_ms(counts=True)
    tree.set_profiling(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed                     # whole-job matvecs/s (one matvec spans all ranks)

    if rank == 0:
        per_launch = {k: (phases[k] / counts[k] if counts[k] else 0.0) for k in phases}
        n = stats.n_nodes
        C = stats.n_cells
        # algorithmic work per launch (this rank; at N=1 the whole matvec) -- DESIGN.md section 5
        m2l_stage_flops = stats.m2l_flops_k1 * K / 2.0          # each stage does 2*n*r per pair
        p2p_tile_bytes = stats.p2p_tile_bytes_k1 + (K - 1) * 8 * (stats.p2p_tile_bytes_k1 // 32)
        kern = {
            "M2L_stage1": {"bound": "mfma", "work": m2l_stage_flops, "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "scale": 1e-12},
            "M2L_stage2": {"bound": "mfma", "work": m2l_stage_flops, "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "scale": 1e-12},
            "P2P": {"bound": "hbm", "work": float(p2p_tile_bytes), "unit": "GB/s", "peak": HBM_PEAK_GBPS, "scale": 1e-9},
        }
        dominant = max(kern, key=lambda k: per_launch[k])
        kd = kern[dominant]
        dur = per_launch[dominant] * 1e-3
        achieved = kd["work"] / dur * kd["scale"] if dur > 0 and world == 1 else None
        # HBM-side bytes of the dominant kernel: bench.py cannot run rocprofv3 on itself, so the figure
        # comes from the committed PMC passes of this same command (scripts/gpu_traffic.sh ->
        # profiles/r01_traffic.json: FETCH_SIZE x2 (gfx950 correction for 16-B/lane reads) + WRITE_SIZE,
        # KiB -> bytes, per launch); null when no profile matches the workload.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tj = json.load(f)
            if (tj.get("points"), tj.get("kernel"), tj.get("order"), tj.get("nrhs")) == (N, args.kernel, args.order, K) \
                    and world == 1:
                traffic = tj["per_launch_bytes"].get(dominant)
        except (OSError, ValueError, KeyError):
            pass
        # MFMA-pipe utilisation of the same kernel from the committed counter passes (scripts/gpu_mfma_util.sh ->
        # profiles/r01_mfma.json: derived metrics MfmaUtil, MfmaFlopsF64); null when no profile matches
        mfma_util = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_mfma.json")) as f:
                mj = json.load(f)
            if (mj.get("points"), mj.get("kernel"), mj.get("order"), mj.get("nrhs")) == (N, args.kernel, args.order, K) \
                    and world == 1:
                mfma_util = mj["per_kernel"].get(dominant, {}).get("mfma_util_pct")
        except (OSError, ValueError, KeyError):
            pass
        roofline = {
            "kernel": dominant, "bound": kd["bound"], "achieved": achieved, "peak": kd["peak"],
            "unit": kd["unit"], "frac": (achieved / kd["peak"]) if achieved else None,
            "traffic": traffic,
            "avg_launch_ms": per_launch[dominant],
            "algorithmic_work_per_launch": kd["work"],
            "mfma_util_pct": mfma_util,
        }
        compulsory_bytes = N * (16 * 3 + 16 * K) + 4 * C * n * 8 * K
        line = {
            "metric": "BBFMM matvecs/s", "value": value, "unit": "matvecs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{N} uniform 3D points, {args.kernel}, order {args.order}, "
                                   f"{K} rhs, adaptive sparse tree, ACA eps=1e-{args.order}, "
                                   "set_weights + evaluate at the sources",
                       "points": N, "kernel": args.kernel, "order": args.order, "nrhs": K,
                       "parallelism": f"target-subtree partition x{world}" if world > 1 else "single GPU"},
            "roofline": roofline,
            "achieved_hbm_gbps_compulsory": compulsory_bytes / (elapsed / args.steps) * 1e-9,
            "phase_ms_per_step": {k: phases[k] / args.steps for k in phases},
            "tree": {"depth": stats.depth, "cells": C, "leaves": stats.n_leaves, "v_pairs": stats.n_v,
                     "p2p_pairs": stats.p2p_pairs, "m2l_flops_k1": stats.m2l_flops_k1,
                     "build_s": t_build},
        }
        if world == 1:
            try:
                tf, errs = F.mfma_f64_selftest()
                line["fp64_mfma_microbench_tflops"] = tf
            except Exception as e:  # noqa: BLE001
                line["fp64_mfma_microbench_tflops"] = None
            if args.cpu_baseline != "off":
                line["cpu_baseline"] = cpu_baseline(args, kernel_id)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
