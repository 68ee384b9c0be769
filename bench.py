#!/usr/bin/env python3
"""bench.py -- BBFMM matvecs/s on MI355X (BASELINE.json: "BBFMM matvecs/s + achieved HBM GB/s,
10M 3D pts, 1/2/4/8 MI355X").

A "step" is one matvec = set_weights (upward pass) + evaluate at the sources (downward + leaf
pass), exactly what one FGMRES `matvec` closure call does in the reference
(ferreus_rbf/src/rbf.rs:1357-1364), on weights already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 without a launcher (WORLD_SIZE unset): this process starts N rank processes itself --
before anything touches the GPU -- and relays rank 0's JSON line.  One process per GPU; every
rank holds the whole tree and owns a contiguous Morton range of target leaves (strong scaling:
the total work is one matvec over all points); the owned potentials are exchanged with one RCCL
all-gather over xGMI per step.

Rank 0 prints ONE JSON line (see the prompt's contract) with `roofline` and `cpu_baseline`; at
N = 1 the line also carries `configs`: the other single-GPU configurations of BASELINE.json
(1M Spheroidal3 / multiquadric, the 10M thin-plate-spline p = 9 operator of config 3, 10M x 8 rhs)
with their own step time, dominant-kernel roofline and sampled dense-row error.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Peaks.  HBM: /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec".  The guide
# lists no FP64 MFMA rate; 78.6 TFLOP/s is AMD's public MI355X FP64 matrix (= vector) figure, and
# the bench also reports the rate a bare v_mfma_f64_4x4x4 loop reaches on this device.
HBM_PEAK_GBPS = 8000.0
FP64_MFMA_PEAK_TFLOPS = 78.6

# BASELINE.json configs that fit one GPU, beside the headline workload (SURVEY.md 8(d) instances)
EXTRA_CONFIGS = [
    {"name": "config2_spheroidal3_1M", "points": 1_000_000, "kernel": "Spheroidal3Rbf", "order": 7, "nrhs": 1,
     "base_range": 0.1, "total_sill": 0.1},
    {"name": "config2_multiquadric_ext_1M", "points": 1_000_000, "kernel": "MultiquadricExt", "order": 7, "nrhs": 1,
     "base_range": 0.1, "total_sill": 0.1},
    {"name": "config4_linear_10M_8rhs", "points": 10_000_000, "kernel": "LinearRbf", "order": 7, "nrhs": 8,
     "base_range": 1.0, "total_sill": 1.0},
    {"name": "config3_operator_tps_10M_order9", "points": 10_000_000, "kernel": "ThinPlateSplineRbf", "order": 9,
     "nrhs": 1, "base_range": 1.0, "total_sill": 1.0},
    # EXTENSION beyond the reference, never the headline: the headline workload with BBFMM_FLAG_M2L_SHARED_BASIS
    # (M2L stages in one basis per level, cut at the operators' epsilon)
    {"name": "extension_shared_basis_linear_10M", "points": 10_000_000, "kernel": "LinearRbf", "order": 7, "nrhs": 1,
     "base_range": 1.0, "total_sill": 1.0, "m2l_shared_basis": True},
    {"name": "extension_shared_basis_linear_10M_8rhs", "points": 10_000_000, "kernel": "LinearRbf", "order": 7, "nrhs": 8,
     "base_range": 1.0, "total_sill": 1.0, "m2l_shared_basis": True},
    # EXTENSION beyond the reference: config 2's workloads with BBFMM_FLAG_DIRECT_SMALL_W_LEAVES (W-list leaves with no
    # more points than nodes are summed directly instead of through M2P / P2L)
    {"name": "extension_direct_w_leaves_spheroidal3_1M", "points": 1_000_000, "kernel": "Spheroidal3Rbf", "order": 7, "nrhs": 1,
     "base_range": 0.1, "total_sill": 0.1, "direct_small_w_leaves": True},
    {"name": "extension_direct_w_leaves_multiquadric_ext_1M", "points": 1_000_000, "kernel": "MultiquadricExt", "order": 7,
     "nrhs": 1, "base_range": 0.1, "total_sill": 0.1, "direct_small_w_leaves": True},
]


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
    ap.add_argument("--configs", default="auto", choices=["auto", "off"],
                    help="auto: at N = 1 with the default workload also time the other single-GPU configs")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "gloo"],
                    help="gloo: CPU-staged exchange; ranks may then share a GPU (LOCAL_RANK modulo the device "
                         "count) -- for exercising the N > 1 path on a one-GPU box, never a scaling number")
    return ap.parse_args()


# --------------------------------------------------------------------------- N > 1 without a launcher
def launch_ranks(args) -> int:
    """Parent of a self-launched multi-rank run.  Touches neither torch nor the GPU: it only starts the
    rank processes (fresh interpreters), relays rank 0's stdout and returns the worst exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus),
                    "LOCAL_WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # relay rank 0's stdout; if any rank dies, stop the others (they would wait in the rendezvous for minutes)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            rc = max(abs(c) for c in codes if c not in (None, 0))
            for p in procs:
                if p.poll() is None:
                    p.terminate()          # our own children, by handle
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return rc


# --------------------------------------------------------------------------- helpers
def source_hash() -> str:
    """Hash of the kernel and orchestration sources: committed counter summaries are only quoted for the
    code that produced them."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ferreus_rbf_rs_amd", "csrc", "*"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def committed_counters(workload_key):
    """PMC summaries of this same command (scripts/gpu_round_profiles.sh -> profiles/r*_counters.json).  bench.py
    cannot run rocprofv3 on itself, so the figures come from the committed passes -- and only when they were
    taken on these exact sources and this workload; otherwise null (never stale numbers)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json"))):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        if j.get("source_hash") == source_hash() and tuple(j.get("workload", ())) == tuple(workload_key):
            best = j
            best["file"] = os.path.relpath(f, ROOT)
    return best


def dense_rows_torch(torch, kernel, br, sill, x, pts, w):
    """K(x, pts) w by direct summation in plain torch f64 (ferreus_rbf_utils/src/utils.rs:288-312): the quantity
    the BBFMM approximates, for the kernels this bench runs (rbf_kernels.rs:25-36, 69-84, 245-256)."""
    r2 = torch.zeros((x.shape[0], pts.shape[0]), dtype=torch.float64, device=x.device)
    for a in range(x.shape[1]):
        r2 += (x[:, a, None] - pts[None, :, a]) ** 2                           # distance_sq, utils.rs:230-237
    if kernel == "LinearRbf":
        phi = -torch.sqrt(r2)
    elif kernel == "ThinPlateSplineRbf":
        r = torch.sqrt(r2)
        phi = torch.where(r < 2.220446049250313e-16, torch.zeros_like(r), r2 * torch.log(torch.clamp(r, min=1e-300)))
    elif kernel == "Spheroidal3Rbf":
        ip, slope, scal, yint = 0.5, 0.75, 2.6798340586, 0.8734640537        # constants.rs:21-50
        s = scal / br
        sr2 = s * s * r2
        t = 1.0 + sr2
        phi = torch.where(sr2 <= ip * ip, sill - sill * slope * s * torch.sqrt(r2), sill * yint / (t * torch.sqrt(t)))
    elif kernel == "MultiquadricExt":
        phi = torch.sqrt(1.0 + r2 / (br * br))
    else:
        return None
    # rows x K (as row sums: rocBLAS picks a very slow kernel for a 32 x N x 1 product)
    return torch.stack([(phi * w[k]).sum(1) for k in range(w.shape[0])], 1)


def cpu_baseline(args, kernel_id):
    """Times the CPU restatement of the reference algorithm (oracle/, kind "port": C + OpenMP
    passes over a Python-built tree) on a bounded sample: a cloud 64x smaller than the workload,
    which has the same leaf occupancy and list structure two levels shallower; the BBFMM matvec
    is O(N), so the rate is scaled by the point ratio.  The sample is run at the host's full thread
    count and at a half and a quarter of it (small problems do not always like every hardware thread);
    the fastest is reported with its thread count.  Run once at the full 10M points the same code
    measured 46.9 s per matvec = 0.021 matvecs/s on 256 threads
    (scripts/cpu_port_full_size.py -> profiles/r02_cpu_port_full_10M.json): the scaled sample flatters the
    CPU by about 2x (caches), which only makes the reported baseline conservative for the GPU."""
    from oracle import bbfmm_oracle as O
    n_cpu = args.cpu_points or max(20000, args.points // 64)
    rng = np.random.default_rng(42)
    pts = rng.random((n_cpu, 3))
    w = np.random.default_rng(43).random((n_cpu, args.nrhs))
    tree = O.FmmTree(pts, args.order, kernel_id, True, True, base_range=args.base_range,
                     total_sill=args.total_sill)
    hw = int(O.lib().oracle_num_threads())
    best = None
    for threads in sorted({hw, max(hw // 2, 1), max(hw // 4, 1)}, reverse=True):
        O.lib().oracle_set_num_threads(threads)
        tree.set_weights(w)                      # warm-up
        tree.evaluate(w, pts)
        times = []
        t_end = time.time() + 8.0
        while len(times) < 5 and (len(times) < 2 or time.time() < t_end):
            t0 = time.time()
            tree.set_weights(w)
            tree.evaluate(w, pts)
            times.append(time.time() - t0)
        t = float(np.median(times))
        if best is None or t < best[0]:
            best = (t, threads, len(times))
    O.lib().oracle_set_num_threads(hw)
    t, threads, reps = best
    scale = n_cpu / float(args.points)
    return {
        "value": (1.0 / t) * scale,
        "unit": "matvecs/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"CPU restatement of the reference algorithm (not the Rust binary): median of "
                   f"{reps} matvecs on {n_cpu} uniform points ({t:.3f} s each on {threads} of {hw} threads, the "
                   f"fastest of full / half / quarter thread counts; same kernel/order/nrhs, same leaf occupancy as "
                   f"the {args.points}-point workload), rate scaled by {n_cpu}/{args.points} (O(N) algorithm); "
                   f"measured once at the full 10M points: 0.021 matvecs/s (profiles/r02_cpu_port_full_10M.json)"),
    }


def roofline_of(stats, K, per_launch, world):
    """Roofline entry of the dominant kernel (largest average launch among the M2L stages and P2P);
    algorithmic work per launch as DESIGN.md section 5 defines it."""
    m2l_stage_flops = stats.m2l_flops_k1 * K / 2.0          # each stage does 2*n*r per pair
    p2p_tile_bytes = stats.p2p_tile_bytes_k1 + (K - 1) * 8 * (stats.p2p_tile_bytes_k1 // 32)
    kern = {
        "M2L_stage1": {"bound": "mfma", "work": m2l_stage_flops, "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "scale": 1e-12},
        "M2L_stage2": {"bound": "mfma", "work": m2l_stage_flops, "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "scale": 1e-12},
        "P2P": {"bound": "hbm", "work": float(p2p_tile_bytes), "unit": "GB/s", "peak": HBM_PEAK_GBPS, "scale": 1e-9},
        # the adaptive-list kernels (M2P + P2L; fused into the P2L phase for one rhs): FP64-VALU bound like P2P,
        # quoted against HBM with their tile traffic (points of the leaf + nodes, multipoles and locals of the W cell)
        "P2L": {"bound": "hbm", "work": float(stats.wx_tile_bytes_k1 * K), "unit": "GB/s", "peak": HBM_PEAK_GBPS, "scale": 1e-9},
    }
    dominant = max(kern, key=lambda k: per_launch[k])
    kd = kern[dominant]
    dur = per_launch[dominant] * 1e-3
    achieved = kd["work"] / dur * kd["scale"] if dur > 0 and world == 1 else None
    return dominant, {
        "kernel": dominant, "bound": kd["bound"], "achieved": achieved, "peak": kd["peak"],
        "unit": kd["unit"], "frac": (achieved / kd["peak"]) if achieved else None,
        "traffic": None, "avg_launch_ms": per_launch[dominant], "algorithmic_work_per_launch": kd["work"],
    }


def time_matvecs(torch, dist, tree, w, out, steps, warmup, world, xchg, stream):
    """W untimed + exactly K timed steps between barrier + synchronize; returns (seconds, phases, counts)."""
    N, K = w.shape[1], w.shape[0]

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

    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    tree.set_profiling(True)
    tree.phase_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    phases, counts = tree.phase_ms(counts=True)
    tree.set_profiling(False)
    return elapsed, phases, counts


def run_extra_config(torch, F, dev, cfg, tree=None):
    """One of EXTRA_CONFIGS on one GPU: step time, phases, dominant-kernel roofline, sampled dense rows."""
    N, K = cfg["points"], cfg["nrhs"]
    pts = np.random.default_rng(42).random((N, 3))
    t0 = time.time()
    if tree is None:
        tree = F.FmmTree(pts, cfg["order"], F.KernelParams(F.KernelType[cfg["kernel"]], base_range=cfg["base_range"],
                                                           total_sill=cfg["total_sill"]), True, True,
                         m2l_shared_basis=bool(cfg.get("m2l_shared_basis")),
                         direct_small_w_leaves=bool(cfg.get("direct_small_w_leaves")))
    t_build = time.time() - t0
    stats = tree.stats()
    w = torch.from_numpy(np.random.default_rng(43).random((K, N))).to(dev)
    out = torch.zeros((K, N), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    steps = 5 if N * K <= 10_000_000 else 3
    elapsed, phases, counts = time_matvecs(torch, None, tree, w, out, steps, 1, 1, None, stream)
    # a phase interval brackets all launches of the phase (rhs chunks, column chunks): per interval = per pass
    per_step = {k: (phases[k] / counts[k] if counts[k] else 0.0) for k in phases}
    _, roof = roofline_of(stats, K, per_step, 1)
    idx = np.random.default_rng(2).choice(N, 32, replace=False)
    pts_d = torch.from_numpy(pts).to(dev)
    err = None
    yd = dense_rows_torch(torch, cfg["kernel"], cfg["base_range"], cfg["total_sill"], pts_d[idx], pts_d, w)
    if yd is not None:
        got = out[:, idx].T
        err = float((got - yd).abs().max() / yd.abs().max())
    del pts_d
    ext = {}
    if cfg.get("direct_small_w_leaves"):
        ext = {"extension": "BBFMM_FLAG_DIRECT_SMALL_W_LEAVES (W-list leaves with no more points than nodes summed directly: "
                            "exact where the reference's M2P / P2L approximate)"}
    if cfg.get("m2l_shared_basis"):
        ext = {"extension": "BBFMM_FLAG_M2L_SHARED_BASIS (not the reference's M2L arithmetic; results within a few epsilon "
                            "of the default path)", "m2l_basis_rank": stats.m2l_basis_rank, "m2l_basis_len": stats.m2l_basis_len}
    return {
        **ext,
        "workload": f"{N} uniform 3D points, {cfg['kernel']}, order {cfg['order']}, {K} rhs",
        "ms_per_step": elapsed / steps * 1e3, "matvecs_per_s": steps / elapsed, "steps": steps,
        "roofline": roof, "phase_ms_per_step": per_step,
        "dense_rows_rel_err": err, "dense_rows": 32,
        "tree": {"depth": stats.depth, "cells": stats.n_cells, "leaves": stats.n_leaves, "v_pairs": stats.n_v,
                 "n_w": stats.n_w, "p2p_pairs": stats.p2p_pairs, "build_s": t_build},
    }


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))        # nothing has touched the GPU in this process

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.exchange == "gloo":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import ferreus_rbf_rs_amd as F
    kernel_id = int(F.KernelType[args.kernel])

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

    xchg = None
    if world > 1:
        from ferreus_rbf_rs_amd.distributed import OwnedRowsExchange
        xchg = OwnedRowsExchange(rows, N, K, dev)     # owned rows are a disjoint cover: all-gather
        assert xchg.check_partition(), "partition does not cover the targets exactly once"

    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    elapsed, phases, counts = time_matvecs(torch, dist, tree, w, out, args.steps, args.warmup, world, xchg, stream)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.exchange == "rccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed                     # whole-job matvecs/s (one matvec spans all ranks)

    if rank == 0:
        per_launch = {k: (phases[k] / counts[k] if counts[k] else 0.0) for k in phases}
        n = stats.n_nodes
        C = stats.n_cells
        # algorithmic work per launch (this rank; at N=1 the whole matvec) -- DESIGN.md section 5
        dominant, roofline = roofline_of(stats, K, per_launch, world)
        # HBM-side bytes and MFMA-pipe utilisation of the dominant kernel from the committed PMC passes of this
        # same command on these same sources (FETCH_SIZE x2 (gfx950 correction for 16-B/lane reads) +
        # WRITE_SIZE, KiB -> bytes, per launch; derived metrics MfmaUtil, MfmaFlopsF64); null otherwise.
        roofline["mfma_util_pct"] = None
        roofline["counters_from"] = None
        if world == 1:
            cj = committed_counters((N, args.kernel, args.order, K))
            if cj:
                roofline["traffic"] = cj.get("per_launch_bytes", {}).get(dominant)
                roofline["mfma_util_pct"] = cj.get("per_kernel", {}).get(dominant, {}).get("mfma_util_pct")
                roofline["counters_from"] = cj["file"]
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
                       "parallelism": (f"target-subtree partition x{world}, owned potentials all-gathered over "
                                       f"{args.exchange}") if world > 1 else "single GPU"},
            "roofline": roofline,
            "achieved_hbm_gbps_compulsory": compulsory_bytes / (elapsed / args.steps) * 1e-9,
            "phase_ms_per_step": {k: phases[k] / args.steps for k in phases},
            "tree": {"depth": stats.depth, "cells": C, "leaves": stats.n_leaves, "v_pairs": stats.n_v,
                     "p2p_pairs": stats.p2p_pairs, "m2l_flops_k1": stats.m2l_flops_k1,
                     "build_s": t_build},
            "source_hash": source_hash(),
        }
        if world == 1:
            idx = np.random.default_rng(2).choice(N, 32, replace=False)
            pts_d = torch.from_numpy(pts).to(dev)
            yd = dense_rows_torch(torch, args.kernel, args.base_range, args.total_sill, pts_d[idx], pts_d, w)
            if yd is not None:
                line["dense_rows_rel_err"] = float((out[:, idx].T - yd).abs().max() / yd.abs().max())
            del pts_d
            try:
                tf, errs = F.mfma_f64_selftest()
                line["fp64_mfma_microbench_tflops"] = tf
            except Exception:  # noqa: BLE001
                line["fp64_mfma_microbench_tflops"] = None
            default_workload = (N, args.kernel, args.order, K) == (10_000_000, "LinearRbf", 7, 1)
            if args.configs == "auto" and default_workload:
                extra = {}
                # configs on the headline tree first (more rhs), then the tree is released for the others
                ordered = sorted(EXTRA_CONFIGS, key=lambda c: (c["points"], c["kernel"], c["order"]) != (N, args.kernel, args.order))
                for cfg in ordered:
                    reuse = tree if (cfg["points"], cfg["kernel"], cfg["order"]) == (N, args.kernel, args.order) and \
                        not cfg["name"].startswith("extension_") else None
                    if reuse is None and tree is not None:
                        del tree, w, out, stream
                        tree = w = out = stream = None
                        torch.cuda.empty_cache()
                    try:
                        extra[cfg["name"]] = run_extra_config(torch, F, dev, cfg, reuse)
                    except Exception as e:  # noqa: BLE001
                        extra[cfg["name"]] = {"error": f"{type(e).__name__}: {e}"}
                    torch.cuda.empty_cache()
                line["configs"] = extra
            if args.cpu_baseline != "off":
                line["cpu_baseline"] = cpu_baseline(args, kernel_id)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
