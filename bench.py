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

Rank 0 prints ONE JSON line of under 4 KB (`compact_line`): the contract's keys, `roofline` of the dominant kernel,
`cpu_baseline` (N = 1), `dense_rows_rel_err` (32 rows of the dense sum computed in plain torch on rank 0 after the
exchange), `source_hash`, and per extra configuration only its step time, dense-row error and the dominant kernel's
roofline kernel / bound / frac.  Everything else -- per-phase rooflines, instruction-issue figures, tree statistics,
microbenchmarks, the prose of the CPU sample -- goes to `bench_detail.json` beside this file and to stderr.
`configs`: at N = 1 the other single-GPU configurations of BASELINE.json (1M Spheroidal3 / multiquadric, the 10M
thin-plate-spline p = 9 operator of config 3, 10M x 8 rhs), at N > 1 config 5 (40M Spheroidal3, partitioned over the
same ranks).  `--configs` adds opt-in entries: `solve` = config 3 end to end (10M thin-plate spline + linear drift,
FGMRES 20 x 5 + Schwarz), `extensions` = the labelled extensions beyond the reference's arithmetic.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
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
    # config 4's second instance: the Gaussian EXTENSION kernel (not a kernel of the reference; its only
    # oracle is the dense sum: 32 rows per run)
    {"name": "config4_gaussian_ext_10M_8rhs", "points": 10_000_000, "kernel": "GaussianExt", "order": 7, "nrhs": 8,
     "base_range": 0.1, "total_sill": 0.1,
     "note": "dense-row error ~1e-2 is the METHOD's (order-7 BBFMM, kernel narrower than the upper cells); device = oracle to 1e-11"},
    # TUNING, never the headline: the headline workload with the reference's OWN knob FmmParams.max_points_per_cell
    # (ferreus_rbf/src/config.rs:216, default 256) at 512 -- a depth-5 tree with 305-point leaves instead of depth 6 with 38:
    # eight times less M2L, eight times more near field, which the whole-leaf kernels of round 6 run at 0.84 of the FMA rate.
    # The same arithmetic, the same accuracy (dense rows), a setting any ferreus_rbf user can make (Params.fmm_params).
    {"name": "tuning_max_points_per_cell_512_linear_10M", "points": 10_000_000, "kernel": "LinearRbf", "order": 7, "nrhs": 1,
     "base_range": 1.0, "total_sill": 1.0, "max_points_per_cell": 512},
    # EXTENSION, never the headline: the headline workload with BBFMM_FLAG_M2L_SHARED_BASIS, so that the figure the README
    # quotes for it is measured by the driver's own run
    {"name": "extension_shared_basis_linear_10M", "points": 10_000_000, "kernel": "LinearRbf", "order": 7, "nrhs": 1,
     "base_range": 1.0, "total_sill": 1.0, "m2l_shared_basis": True},
]

# EXTENSIONS beyond the reference (never the headline, never in the default line): `--configs extensions`
EXTENSION_CONFIGS = [
    # (extension_shared_basis_linear_10M -- the headline workload with BBFMM_FLAG_M2L_SHARED_BASIS -- runs with `auto`)
    {"name": "extension_shared_basis_linear_10M_8rhs", "points": 10_000_000, "kernel": "LinearRbf", "order": 7, "nrhs": 8,
     "base_range": 1.0, "total_sill": 1.0, "m2l_shared_basis": True},
    # config 2's workloads with BBFMM_FLAG_DIRECT_SMALL_W_LEAVES (small W-list leaves summed directly)
    {"name": "extension_direct_w_leaves_spheroidal3_1M", "points": 1_000_000, "kernel": "Spheroidal3Rbf", "order": 7, "nrhs": 1,
     "base_range": 0.1, "total_sill": 0.1, "direct_small_w_leaves": True},
    {"name": "extension_direct_w_leaves_multiquadric_ext_1M", "points": 1_000_000, "kernel": "MultiquadricExt", "order": 7,
     "nrhs": 1, "base_range": 0.1, "total_sill": 0.1, "direct_small_w_leaves": True},
]

# BASELINE.json config 5: rides on the N > 1 line, partitioned over the same ranks
CONFIG5 = {"name": "config5_spheroidal3_40M", "points": 40_000_000, "kernel": "Spheroidal3Rbf", "order": 7, "nrhs": 1,
           "base_range": 0.1, "total_sill": 0.1}

# FP64 vector issue peak: 256 CUs x 4 SIMDs, one wave64 FP64 instruction per 4 cycles, at the 2.4 GHz AMD's
# 78.6 TFLOP/s assumes (= 78.6e12 / 2 lane-FMAs per second)
FP64_VALU_LANE_INSTR_PEAK = 78.6e12 / 2.0

# the driver extracts the line from a bounded tail of stdout: round 3's 20.7 KB line came back unparsed
LINE_LIMIT = 4096


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
    ap.add_argument("--dropin", default="auto", choices=["auto", "off"],
                    help="auto: with the default workload at N = 1 also time the drop-in boundary on host buffers "
                         "(patched and unchanged caller; detail file + one small entry of the line)")
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU sample (0: points/8)")
    ap.add_argument("--configs", default="auto",
                    help="comma list of: auto (with the default workload: the other single-GPU configs at N = 1, "
                         "config 5 = 40M Spheroidal3 at N > 1), off, solve (config 3 end to end: 10M thin-plate spline "
                         "FGMRES + Schwarz, N = 1), extensions (the labelled extensions, N = 1), config5 (config 5 at "
                         "N > 1 whatever the headline workload)")
    ap.add_argument("--config5-points", type=int, default=CONFIG5["points"], help="points of config 5 on the N > 1 line")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "gloo"],
                    help="gloo: CPU-staged exchange; ranks may then share a GPU (LOCAL_RANK modulo the device "
                         "count) -- for exercising the N > 1 path on a one-GPU box, never a scaling number")
    ap.add_argument("--inprocess", action="store_true",
                    help="ONE process, one handle over --gpus devices (bbfmm_create_on_devices): the drop-in's own multi-GPU "
                         "path, exchange inside the library (peer copies).  A labelled second entry; the default N > 1 path "
                         "stays one process per GPU over RCCL so that the two can be compared")
    ap.add_argument("--devices", default="",
                    help="with --inprocess: explicit device list, e.g. 0,0,0,0 = four logical parts on device 0 (a functional "
                         "rehearsal on a one-GPU box, never a scaling number); default 0..gpus-1")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (the stdout line is the compact one)")
    return ap.parse_args()


# --------------------------------------------------------------------------- N > 1 without a launcher
def launch_ranks(args) -> int:
    """Parent of a self-launched multi-rank run.  Touches neither torch nor the GPU: it only starts the
    rank processes (fresh interpreters), relays rank 0's stdout and returns the worst exit code.  The ranks meet
    through a file store in a private temporary directory (no port to lose between picking and binding it); every
    rank's stderr is kept, and the tail of a failing rank's is printed with its id."""
    import tempfile
    import threading
    tmp = tempfile.mkdtemp(prefix="bbfmm_bench_")
    rdzv = os.path.join(tmp, "rendezvous")
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus),
                    "LOCAL_WORLD_SIZE": str(args.gpus), "BBFMM_RDZV_FILE": rdzv,
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        errs.append(open(os.path.join(tmp, f"rank{r}.stderr"), "w+b"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    # relay rank 0's stdout; if any rank dies, stop the others (they would wait in the rendezvous for minutes)
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, failed = 0, []
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            rc = max(abs(c) for _, c in failed)
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
    for r, f in enumerate(errs):
        f.seek(0)
        text = f.read().decode(errors="replace")
        f.close()
        if any(r == fr for fr, _ in failed):
            sys.stderr.write(f"[bench] rank {r} exited with {dict(failed)[r]}; last lines of its stderr:\n")
            sys.stderr.write("\n".join(text.splitlines()[-40:]) + "\n")
        elif r == 0 and text:
            sys.stderr.write(text)          # rank 0's warnings, as before
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return rc


# --------------------------------------------------------------------------- helpers
def source_hash() -> str:
    """Hash of the kernel and orchestration sources of the matvec: committed counter summaries are only quoted for the
    code that produced them.  (The preconditioner's and the solvers' files -- ddm*, schwarz*, solver* -- launch nothing in
    the timed steps and are left out.)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ferreus_rbf_rs_amd", "csrc", "*"))):
        if os.path.basename(f).startswith(("ddm", "schwarz", "solver")):
            continue
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
    elif kernel == "GaussianExt":                                             # extension (SURVEY finding 3): exp(-(r / base_range)^2)
        phi = torch.exp(-r2 / (br * br))
    else:
        return None
    # rows x K (as row sums: rocBLAS picks a very slow kernel for a 32 x N x 1 product)
    return torch.stack([(phi * w[k]).sum(1) for k in range(w.shape[0])], 1)


def measured_cpu_full_size():
    """The CPU port timed ONCE at the full 10M points on a GPU box's host (scripts/cpu_port_full_size.py; the oracle's
    Python tree build alone takes minutes there, so it is not part of a bench run): the newest committed
    profiles/r*_cpu_port_full_10M.json, or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_port_full_10M.json"))):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        j["file"] = os.path.relpath(f, ROOT)
        best = j
    return best


def host_cpu_quota():
    """CPUs of bandwidth the container may use (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`), or None when
    unlimited / unknown.  The GPU boxes show 256 hardware threads and grant 16 CPUs (`cpu.max` = 1600000 100000): every
    OpenMP thread beyond the quota only gets the job throttled, which is what made the port look as if it scaled negatively."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        if q > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(args, kernel_id):
    """Times the CPU restatement of the reference algorithm (oracle/, kind "port": C + OpenMP passes over a
    Python-built tree) in THIS run, on a bounded sample: a cloud 8x smaller than the workload (10M -> 1.25M points:
    the same leaf occupancy, one level shallower), rate scaled by the point ratio.  ONE measured number: `value`.
    The port runs in its GEMM-shaped mode (round 5): M2L per (target cell, reference vector) as gather, two
    register-blocked FMA GEMMs, permuted scatter -- what the reference's faer calls do (bbfmm.rs:910-982) -- and the
    near field on gathered copies in vectorised loops; the plain-loop passes the parity tests use stay its checker
    (tests/test_oracle_vs_dense.py, 1e-13).  Threads: all hardware threads, a half and a quarter of them are timed ON THE
    SAMPLE ITSELF (one tree build serves both; the 64x smaller pick of round 4 chose 32 of 128 because small problems dislike
    many threads -- and, found in round 5, because the boxes grant 16 CPUs of bandwidth behind 256 visible hardware threads:
    with a CPU quota the counts tried are one, two and four threads per granted CPU) and the fastest one is the value.  One warm-up, then matvecs until two are done and the budget is
    spent, median.  The oracle's Python tree build is timed separately (`tree_build_s`; not part of a matvec).
    The O(N) scaling of the sample OVERSTATES the port at the full size (caches: measured once at 10M, see
    `measured_full_size`); a baseline only, never the target.  Returns (entry for the line, detail)."""
    from oracle import bbfmm_oracle as O
    hw = int(O.lib().oracle_num_threads())
    n_cpu = args.cpu_points or max(20000, args.points // 8)
    pts = np.random.default_rng(42).random((n_cpu, 3))
    w = np.random.default_rng(43).random((n_cpu, args.nrhs))
    t0 = time.time()
    tree = O.FmmTree(pts, args.order, kernel_id, True, True, base_range=args.base_range, total_sill=args.total_sill)
    t_tree = time.time() - t0
    tree.gemm_shaped = True

    def timed(threads, budget_s, max_reps):
        O.lib().oracle_set_num_threads(threads)
        tree.set_weights(w)                      # warm-up
        tree.evaluate(w, pts)
        times = []
        t_end = time.time() + budget_s
        while len(times) < max_reps and (len(times) < 2 or time.time() < t_end):
            t0 = time.time()
            tree.set_weights(w)
            tree.evaluate(w, pts)
            times.append(time.time() - t0)
        return float(np.median(times)), threads, len(times)

    quota = host_cpu_quota()
    if quota and quota < hw:    # a CPU quota below the visible threads: one, two and four threads per granted CPU
        q = max(int(round(quota)), 1)
        cand = sorted({min(hw, q), min(hw, 2 * q), min(hw, 4 * q)}, reverse=True)
    else:
        cand = sorted({hw, max(hw // 2, 1), max(hw // 4, 1)}, reverse=True)
    runs = [timed(th, 5.0, 4) for th in cand]
    O.lib().oracle_set_num_threads(hw)
    t, threads, reps = min(runs)
    scale = n_cpu / float(args.points)
    full = measured_cpu_full_size() if (args.points, args.order, args.nrhs) == (10_000_000, 7, 1) and kernel_id == 0 else None
    over = None
    if full and full.get("matvecs_per_s_full_size"):
        over = (1.0 / t) * scale / full["matvecs_per_s_full_size"]
    # what a reader must not miss comes FIRST in `sample` and again under keys of its own (the line's 200-character
    # cut took exactly the clause that said the scaled sample overstates the port)
    entry = {
        "value": (1.0 / t) * scale, "unit": "matvecs/s", "cores": threads, "kind": "port",
        "host_cpu_quota": quota,                                       # CPUs the box's cgroup grants (`cores` = THREADS run on them)
        "overstates_full_size_by": over,                               # scaled sample / the one measured full-size run
        "measured_full_size_value": (full or {}).get("matvecs_per_s_full_size"),
        "sample": ((f"overstates the full-size port {over:.1f}x ({full['matvecs_per_s_full_size']:.3f} matvecs/s measured once at 10M); "
                    if over else "")
                   + (f"{quota:g}-CPU cgroup quota, {threads} threads; " if quota else f"{threads} of {hw} threads; ")
                   + f"median of {reps} matvecs on {n_cpu} uniform points ({t:.2f} s each), rate x {n_cpu}/{args.points}; "
                     "GEMM-shaped C/OpenMP port, not the Rust binary"),
    }
    detail = {
        **entry, "seconds_per_matvec_on_sample": t, "sample_points": n_cpu, "host_threads": hw, "host_cpu_quota": quota,
        "oracle_tree_build_s": t_tree,
        "thread_runs_on_the_sample": [{"threads": th, "seconds_per_matvec": tt, "reps": rp} for tt, th, rp in runs],
        "measured_full_size": full,
        "note": ("CPU restatement of the reference algorithm (oracle/passes.c, C + OpenMP over the oracle's Python-built "
                 "tree), same kernel / order / nrhs and the same leaf occupancy as the workload, GEMM-shaped M2L "
                 "(oracle_m2l_gemm) and gathered, vectorised near field; not the Rust binary: baseline only"),
    }
    return entry, detail


# Arithmetic of one kernel evaluation in the reference's pair loops (bbfmm.rs:1162-1251 with distance_sq
# utils.rs:230-237 and the kernels of rbf_kernels.rs): d subtractions, d multiplies, d - 1 additions, the square
# root, phi, and the multiply-add with the weight -- counted as flops, a transcendental as one.
PAIR_FLOPS = {"LinearRbf": 3 * 3 - 1 + 1 + 1 + 2, "ThinPlateSplineRbf": 3 * 3 - 1 + 1 + 3 + 2, "CubicRbf": 3 * 3 - 1 + 1 + 2 + 2,
              "Spheroidal3Rbf": 3 * 3 - 1 + 1 + 6 + 2, "MultiquadricExt": 3 * 3 - 1 + 1 + 2 + 2}

# right-hand sides the unordered-pair kernels (near field; fused M2P + P2L) take in one pass (device.hpp kSymMaxRhs);
# kernel instances exist for 1, 2 and 4 -- 3 runs the 4-slot instance with an idle slot
SYM_MAX_RHS = 4


def sym_instances(K):
    """Kernel instances (rhs slots) of the passes that serve K right-hand sides."""
    out = []
    for k0 in range(0, K, SYM_MAX_RHS):
        kb = min(SYM_MAX_RHS, K - k0)
        out.append(1 if kb == 1 else 2 if kb == 2 else 4)
    return out


def pair_probe_hash() -> str:
    """The inlined arithmetic of every pair kernel lives in csrc/kernels.hpp; scripts/pair_probe.hip wraps one
    evaluation of it.  The ISA counts are stamped with the hash of these two files only."""
    h = hashlib.sha256()
    for f in (os.path.join(ROOT, "ferreus_rbf_rs_amd", "csrc", "kernels.hpp"), os.path.join(ROOT, "scripts", "pair_probe.hip")):
        try:
            with open(f, "rb") as fh:
                h.update(os.path.basename(f).encode() + b"\0" + fh.read())
        except OSError:
            h.update(b"missing")
    return h.hexdigest()[:16]


def committed_pair_instructions():
    """FP64 VALU instructions per kernel evaluation of the pair kernels, counted from the ISA of these very sources
    (scripts/pair_instruction_counts.py -> profiles/r*_pair_instruction_counts.json, stamped with `pair_probe_hash`)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pair_instruction_counts.json"))):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        if j.get("pair_probe_hash") == pair_probe_hash():
            best = j
            best["file"] = os.path.relpath(f, ROOT)
    return best


def pair_issue(stats, N, K, kernel, phase):
    """(kernel evaluations executed, FP64 VALU instructions per evaluation, instruction-count file) of a pair phase
    of the matvec -- P2P, P2L (= M2P + P2L fused) -- or None.  The unordered kernels evaluate every pair once per
    pass of up to four right-hand sides and feed the row sum and the column sum of each rhs slot from it; the count
    of the probe is for one rhs, both sums (2)."""
    instr = committed_pair_instructions()
    if not instr or kernel not in instr.get("kernels", {}):
        return None
    sym = phase in ("P2P", "P2L")             # the matvec (targets = sources): unordered pairs, whatever K
    key = {"P2P": "p2p_sym", "P2L": "wx_sym", "M2P": "m2p"}[phase]
    ipp = instr["kernels"][kernel][key]["fp64_valu_per_pair"]
    if sym:      # per pass: one evaluation, a row and a column multiply-add per rhs slot of the kernel instance
        evals = (stats.p2p_pairs + N) / 2.0 if phase == "P2P" else float(stats.wx_pairs)
        per_eval = sum(ipp - 2 + 2 * kb for kb in sym_instances(K))
    else:
        evals = float(stats.wx_pairs)
        per_eval = ipp - 2 + K
    return evals, per_eval, instr["file"]


def roofline_of(stats, N, K, kernel, per_launch, world, launches=None, valu_lane_rate=None):
    """Roofline entry of the dominant kernel (largest time per matvec among the M2L stages and the pair phases);
    algorithmic work as DESIGN.md section 5 defines it.  `per_launch` holds the time of ALL launches of a phase in one
    matvec (one launch when the M2L intermediate is a single batch and all right-hand sides go in one pass -- the
    headline; otherwise `launches_per_step` of them, and work and time are both the sums over those launches).

    Rule: a phase is quoted against the roof that binds it.  The M2L stages against the FP64
    matrix peak (78.6 TFLOP/s) with their algorithmic flops.  The pair phases (P2P; P2L = M2P + P2L) are FP64
    vector-instruction bound: `achieved` = kernel evaluations executed x FP64 VALU instructions per evaluation (ISA
    count) per second, `peak` = the FP64 lane-instruction rate the same device sustained in this run's FMA
    microbenchmark (nominal 78.6e12 / 2 if it was not run); the HBM figure the north-star asks for -- tile bytes
    against 8 TB/s -- stays beside it as `frac_hbm`."""
    m2l_stage_flops = stats.m2l_flops_k1 * K / 2.0          # each stage does 2*n*r per pair
    p2p_tile_bytes = stats.p2p_tile_bytes_k1 + (K - 1) * 8 * (stats.p2p_tile_bytes_k1 // 32)
    kern = {
        "M2L_stage1": {"bound": "mfma", "work": m2l_stage_flops, "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "scale": 1e-12},
        "M2L_stage2": {"bound": "mfma", "work": m2l_stage_flops, "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "scale": 1e-12},
        "P2P": {"bound": "fp64_valu", "bytes": float(p2p_tile_bytes)},
        "P2L": {"bound": "fp64_valu", "bytes": float(stats.wx_tile_bytes_k1 * K)},
    }
    dominant = max(kern, key=lambda k: per_launch[k])
    kd = kern[dominant]
    dur = per_launch[dominant] * 1e-3
    extra = {}
    if kd["bound"] == "fp64_valu":
        # `frac` is against the NOMINAL issue peak (78.6e12 / 2 lane instructions per second at 2.4 GHz): a device that
        # clocks lower must not look better.  The rate the same device sustained in this run's bare FMA loop is quoted
        # beside it (`frac_of_measured_fma_rate`), never instead of it.
        pi = pair_issue(stats, N, K, kernel, dominant)
        if pi:
            work, unit, scale, peak = pi[0] * pi[1], "Tinstr/s", 1e-12, FP64_VALU_LANE_INSTR_PEAK * 1e-12
            extra = {"fp64_valu_instr_per_evaluation": pi[1], "counted_from": pi[2], "peak_is": "nominal 78.6e12 / 2 lane instr/s",
                     "instr_count_is": "ISA count of the one-rhs pair loop (scripts/pair_probe.hip)" if K == 1 else
                     "DERIVED for %d rhs: (one-rhs ISA count - 2 + 2 x slots) per pass of the 1/2/4-slot instances; idle slots "
                     "of a partly filled instance count as work" % K}
            if valu_lane_rate and dur > 0:
                extra["frac_of_measured_fma_rate"] = work / world / dur / valu_lane_rate
        else:   # no ISA count for these sources: the reference's flop count per evaluation against the FP64 flop peak
            work = (stats.p2p_pairs if dominant == "P2P" else stats.wx_pairs) * PAIR_FLOPS.get(kernel, 14) * K
            unit, scale, peak = "TFLOP/s", 1e-12, FP64_MFMA_PEAK_TFLOPS
            extra = {"peak_is": "nominal 78.6 TFLOP/s (FP64 vector = matrix); flops = ordered pairs x the reference's flops per pair"}
        extra["frac_hbm"] = (kd["bytes"] / world / dur * 1e-9 / HBM_PEAK_GBPS) if dur > 0 else None
        kd = {"bound": "fp64_valu", "work": work, "unit": unit, "peak": peak, "scale": scale}
    # N > 1: rank 0's launch against ITS share of the job's algorithmic work, taken as 1 / world (the partition balances
    # a work proxy; the halo a rank computes beyond its share is not algorithmic work and lowers the figure)
    work = kd["work"] / world
    achieved = work / dur * kd["scale"] if dur > 0 else None
    out = {
        "kernel": dominant, "bound": kd["bound"], "achieved": achieved, "peak": kd["peak"],
        "unit": kd["unit"], "frac": (achieved / kd["peak"]) if achieved else None,
        "traffic": None, "avg_launch_ms": per_launch[dominant], "algorithmic_work_per_launch": work,
        "launches_per_step": (launches or {}).get(dominant, 1), **extra,
    }
    if world > 1:
        out["work_share"] = "rank 0's launch; algorithmic work = the job's / %d" % world
    return dominant, out


def phase_roofline(stats, N, K, order, kernel, per_step_ms, valu_lane_rate=None):
    """per phase the algorithmic bytes and flops of one matvec, the achieved GB/s and TFLOP/s, the
    fractions of the HBM (8 TB/s) and FP64 (78.6 TFLOP/s) peaks, and which of the two binds the phase (the larger
    of bytes / peak-bandwidth and flops / peak-rate).  Bytes: what the phase has to move once (tile traffic for the
    pair phases, as the north-star defines it); flops: the reference's arithmetic (sum-factorised transfers for
    M2M / L2L, 4 n r per V pair for M2L).  The pair phases also get their FP64 instruction-issue figure."""
    n, C, d, p = stats.n_nodes, stats.n_cells, stats.d, order
    leaves = stats.n_leaves
    sum_r = stats.m2l_flops_k1 / (4.0 * n)                   # sum of the ranks over all V pairs
    xfer = 2.0 * d * p ** (d + 1) * max(C - 1, 0) * K
    pf = PAIR_FLOPS.get(kernel, 14)
    work = {
        "gather": (N * K * 16.0 + N * 4.0, 0.0),
        "P2M": (N * (8.0 * d + 8 * K) + leaves * n * 8.0 * K, (2.0 + d) * N * n * K),
        "M2M": (2.0 * C * n * 8 * K, xfer),
        "M2L_stage1": ((C * n * 8.0 + sum_r * 8.0) * K, stats.m2l_flops_k1 * K / 2.0),
        "M2L_stage2": ((sum_r * 8.0 + C * n * 8.0) * K, stats.m2l_flops_k1 * K / 2.0),
        "P2L": (float(stats.wx_tile_bytes_k1) * K, float(stats.wx_pairs) * pf * K),
        "L2L": (2.0 * C * n * 8 * K, xfer),
        "P2P": (float(stats.p2p_tile_bytes_k1 + (K - 1) * 8 * (stats.p2p_tile_bytes_k1 // 32)), float(stats.p2p_pairs) * pf * K),
        "M2P": (float(stats.wx_tile_bytes_k1) * K, float(stats.wx_pairs) * pf * K),
        "L2P": (N * (8.0 * d + 8 * K) + leaves * n * 8.0 * K, (2.0 + d) * N * n * K),
        "scatter": (N * K * 16.0 + N * 4.0, 0.0),
    }
    out = {}
    for ph, (nbytes, flops) in work.items():
        ms = per_step_ms.get(ph, 0.0)
        if ms <= 0.0:
            continue
        if ph in ("P2L", "M2P") and stats.n_w == 0:
            continue
        if ph == "M2P":
            continue                                     # fused with P2L into one kernel (timed under P2L)
        sec = ms * 1e-3
        t_hbm, t_fp = nbytes / (HBM_PEAK_GBPS * 1e9), flops / (FP64_MFMA_PEAK_TFLOPS * 1e12)
        e = {"ms": ms, "bytes": nbytes, "flops": flops, "gbps": nbytes / sec * 1e-9, "tflops": flops / sec * 1e-12,
             "frac_hbm": nbytes / sec * 1e-9 / HBM_PEAK_GBPS, "frac_fp64": flops / sec * 1e-12 / FP64_MFMA_PEAK_TFLOPS,
             "bound": "hbm" if t_hbm >= t_fp else ("mfma" if ph.startswith("M2L") else "fp64_valu")}
        # FP64 instruction issue of the pair kernels: kernel evaluations actually executed (every unordered pair once
        # in the symmetric kernels) x FP64 VALU instructions per evaluation (ISA count) against the issue peak
        pi = pair_issue(stats, N, K, kernel, ph) if ph in ("P2P", "P2L", "M2P") else None
        if pi:
            evals, per_eval, src = pi
            e["valu_issue"] = {"kernel_evaluations": evals, "fp64_valu_instr_per_evaluation": per_eval,
                               "achieved_lane_instr_per_s": evals * per_eval / sec, "peak_lane_instr_per_s": FP64_VALU_LANE_INSTR_PEAK,
                               "frac": evals * per_eval / sec / FP64_VALU_LANE_INSTR_PEAK, "counted_from": src}
            if valu_lane_rate:
                e["valu_issue"]["frac_of_measured_fma_rate"] = evals * per_eval / sec / valu_lane_rate
        out[ph] = e
    return out


def time_matvecs(torch, dist, tree, w, out, steps, warmup, world, pm, stream):
    """W untimed + exactly K timed steps between barrier + synchronize; returns (seconds, phases, counts)."""
    N, K = w.shape[1], w.shape[0]

    def step():
        if world > 1:
            # own share of the upward pass, all-reduce of the coarse multipoles, downward + leaf pass of the owned
            # targets, all-gather of the owned potentials -- all queued on the handle's stream (distributed.py)
            pm.step(w, out)
        else:
            # hot path: gather, P2M, M2M, M2L, P2L, L2L, P2P, M2P, L2P, scatter -- all on the handle's stream
            tree.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, sync=False)

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


def dense_rows_err(torch, dev, cfg_kernel, br, sill, pts, w, out, rows=32):
    idx = np.random.default_rng(2).choice(pts.shape[0], rows, replace=False)
    pts_d = torch.from_numpy(pts).to(dev)
    yd = dense_rows_torch(torch, cfg_kernel, br, sill, pts_d[idx], pts_d, w)
    del pts_d
    if yd is None:
        return None
    return float((out[:, idx].T - yd).abs().max() / yd.abs().max())


def run_config(torch, dist, F, dev, cfg, world, rank, exchange, tree=None, valu_lane_rate=None):
    """One configuration beside the headline: step time, phases, rooflines, sampled dense rows.  world > 1: the
    same partitioned step as the headline (every rank calls this; rank 0 reports)."""
    N, K = cfg["points"], cfg["nrhs"]
    pts = np.random.default_rng(42).random((N, 3))
    t0 = time.time()
    if tree is None:
        par = None
        if cfg.get("max_points_per_cell"):   # the reference's defaults (bbfmm.rs:96-103) with another leaf limit
            par = F.FmmParams(cfg["max_points_per_cell"], F.M2LCompressionType.ACA, 10.0 ** -cfg["order"], 1024)
        tree = F.FmmTree(pts, cfg["order"], F.KernelParams(F.KernelType[cfg["kernel"]], base_range=cfg["base_range"],
                                                           total_sill=cfg["total_sill"]), True, True, params=par,
                         m2l_shared_basis=bool(cfg.get("m2l_shared_basis")),
                         direct_small_w_leaves=bool(cfg.get("direct_small_w_leaves")))
    t_build = time.time() - t0
    stats = tree.stats()
    w = torch.from_numpy(np.random.default_rng(43).random((K, N))).to(dev)
    out = torch.zeros((K, N), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    pm = None
    if world > 1:
        from ferreus_rbf_rs_amd.distributed import PartitionedMatvec
        tree.set_partition(rank, world)
        pm = PartitionedMatvec(tree, N, K, dev)
        assert pm.check_partition(), "partition does not cover the targets exactly once"
    steps = 5 if N * K <= 10_000_000 else 3
    elapsed, phases, counts = time_matvecs(torch, dist, tree, w, out, steps, 1, world, pm, stream)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if exchange == "rccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank != 0:
        return None
    # a phase interval brackets all launches of the phase (rhs chunks, column chunks): per interval = per pass
    per_step_total = {k: phases[k] / steps for k in phases}
    _, roof = roofline_of(stats, N, K, cfg["kernel"], per_step_total, world, {k: counts[k] // max(steps, 1) for k in counts},
                          valu_lane_rate)
    err = dense_rows_err(torch, dev, cfg["kernel"], cfg["base_range"], cfg["total_sill"], pts, w, out)
    ext = {}
    if cfg.get("max_points_per_cell"):
        ext = {"tuning": "FmmParams.max_points_per_cell = %d (the reference's own parameter, config.rs:216; its default is 256): "
                         "never the headline" % cfg["max_points_per_cell"]}
    if cfg.get("direct_small_w_leaves"):
        ext = {"extension": "BBFMM_FLAG_DIRECT_SMALL_W_LEAVES (W-list leaves with no more points than nodes summed directly: "
                            "exact where the reference's M2P / P2L approximate)"}
    if cfg.get("m2l_shared_basis"):
        ext = {"extension": "BBFMM_FLAG_M2L_SHARED_BASIS (not the reference's M2L arithmetic; results within a few epsilon "
                            "of the default path)", "m2l_basis_rank": stats.m2l_basis_rank, "m2l_basis_len": stats.m2l_basis_len}
    res = {
        **ext,
        "workload": f"{N} uniform 3D points, {cfg['kernel']}, order {cfg['order']}, {K} rhs",
        "n_gpus": world,
        "ms_per_step": elapsed / steps * 1e3, "matvecs_per_s": steps / elapsed, "steps": steps,
        "roofline": roof, "phase_ms_per_step": per_step_total,
        "dense_rows_rel_err": err, "dense_rows": 32,
        "tree": {"depth": stats.depth, "cells": stats.n_cells, "leaves": stats.n_leaves, "v_pairs": stats.n_v,
                 "n_w": stats.n_w, "p2p_pairs": stats.p2p_pairs, "build_s": t_build},
    }
    if cfg.get("note"):
        res["note"] = cfg["note"]
    if world == 1 and not ext:
        res["phase_roofline"] = phase_roofline(stats, N, K, cfg["order"], cfg["kernel"], per_step_total, valu_lane_rate)
    return res


def dropin_host_buffer_times(tree, pts, reps=5):
    """The drop-in boundary on HOST buffers (PCIe inclusive; never `value`), both ways a ferreus_rbf caller can reach the
    matvec: (i) the patched caller, one bbfmm_fast_matrix_vector_product; (ii) the UNCHANGED caller of rbf.rs:1357-1364,
    bbfmm_set_weights(w) then bbfmm_evaluate(w, the source rows) -- which bbfmm_evaluate recognises (bit-for-bit
    comparison with the handle's sources beside the M2L) and serves from the resident target set; (iii) the same
    sequence with one bit of one target coordinate changed, i.e. the general path the sequence took before round 5
    (target upload + grouping, ordered pairs).  Median of `reps` calls after one warm-up each; preallocated outputs; the
    caller's own select_mat_rows copy (rbf.rs:1359-1360) is the caller's and not timed."""
    import ctypes
    from ferreus_rbf_rs_amd import _lib as L
    lib = L.load()
    n = pts.shape[0]
    x = np.asfortranarray(pts)                               # N x 3 column-major, as faer's Mat
    w = np.random.default_rng(43).random(n)
    y = np.zeros(n)
    bad = ctypes.c_int64(-1)

    def patched():
        return lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n, 0, None, 0, None, 0, 0.0, y.ctypes.data)

    def unchanged(xx):
        rc = lib.bbfmm_set_weights(tree._h, w.ctypes.data, n, 1, n)
        return rc or lib.bbfmm_evaluate(tree._h, w.ctypes.data, n, 1, n, xx.ctypes.data, n, n, y.ctypes.data, n, ctypes.byref(bad))

    def med(fn):
        assert fn() == 0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    res = {"points": n, "patched_caller_ms": med(patched)}
    y_patched = y.copy()
    res["unchanged_caller_ms"] = med(lambda: unchanged(x))
    res["unchanged_caller_took_resident_path"] = bool(tree.last_evaluate_at_sources())
    res["unchanged_vs_patched_rel_diff"] = float(np.abs(y - y_patched).max() / np.abs(y_patched).max())
    x2 = x.copy(order="F")
    x2[n // 2, 1] = np.nextafter(x2[n // 2, 1], 0.0)
    res["unchanged_caller_general_path_ms"] = med(lambda: unchanged(x2))
    res["general_path_took_resident_path"] = bool(tree.last_evaluate_at_sources())
    res["ratio_unchanged_over_patched"] = res["unchanged_caller_ms"] / res["patched_caller_ms"]
    # matvec_partial (rbf.rs:119-133) the same two ways, on the rows of a Schwarz coarse level (N / 512): patched =
    # target_indices, unchanged = set_weights + evaluate at select_mat_rows(source_points, idx) (rows of the sources are
    # recognised and served by the same cached plan)
    idx = np.sort(np.random.default_rng(44).choice(n, max(n // 512, 1), replace=False)).astype(np.int64)
    xs = np.asfortranarray(pts[idx])
    m = len(idx)
    z = np.zeros(m)

    def patched_partial():
        return lib.bbfmm_fast_matrix_vector_product(tree._h, w.ctypes.data, n, 0, idx.ctypes.data, m, None, 0, 0.0, y.ctypes.data)

    def unchanged_partial():
        rc = lib.bbfmm_set_weights(tree._h, w.ctypes.data, n, 1, n)
        return rc or lib.bbfmm_evaluate(tree._h, w.ctypes.data, n, 1, n, xs.ctypes.data, m, m, z.ctypes.data, m, ctypes.byref(bad))

    part = {"rows": m, "patched_ms": med(patched_partial)}
    part["unchanged_ms"] = med(unchanged_partial)
    part["unchanged_path"] = int(tree.last_evaluate_path())       # 2 = the cached plan of a row subset
    part["rel_diff"] = float(np.abs(y[idx] - z).max() / max(np.abs(z).max(), 1e-300))
    res["partial_product_rows_n_over_512"] = part
    return res


def _sig(x, digits=5):
    """Floats of the printed line at five significant digits (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def compact_line(detail: dict) -> dict:
    """The ONE stdout line: the driver has to parse it, so it stays under 4 KB whatever the run carried.  Keeps the
    contract's keys, the dominant kernel's roofline, the CPU baseline, the dense-row check and the source hash; of
    every extra configuration only its step time, dense-row error and roofline kernel / bound / frac (config 3's
    solve: iterations, seconds, final residual, converged).  The full record is `detail_file`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    line = {k: _sig(detail[k], 7) for k in keep if k in detail}
    for k in ("entry", "value_host_buffers", "ms_per_step_host_buffers", "part_sum_ms"):   # --inprocess; value_host_buffers on every N = 1 line
        if detail.get(k) is not None:
            line[k] = [_sig(v) for v in detail[k]] if isinstance(detail[k], list) else _sig(detail[k])
    cfg = detail.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "points", "kernel", "order", "nrhs", "parallelism") if k in cfg}
    roof = detail.get("roofline") or {}
    line["roofline"] = {k: _sig(roof.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                         "avg_launch_ms", "mfma_util_pct")}
    for k in ("frac_hbm", "frac_of_measured_fma_rate", "peak_is"):
        if roof.get(k) is not None:
            line["roofline"][k] = _sig(roof[k])
    p2p = (detail.get("phase_roofline") or {}).get("P2P")
    if p2p:          # the north-star's near-field figure, whatever the dominant kernel is
        vi = p2p.get("valu_issue", {})
        line["p2p"] = {"ms": _sig(p2p["ms"]), "frac_hbm": _sig(p2p["frac_hbm"]),
                       "frac_fp64_valu": _sig(vi.get("frac")),                       # against the nominal issue peak
                       "frac_of_measured_fma_rate": _sig(vi.get("frac_of_measured_fma_rate"))}
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _sig(cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample")}
        for k in ("host_cpu_quota", "overstates_full_size_by", "measured_full_size_value"):   # never cut: keys of their own
            if cb.get(k) is not None:
                line["cpu_baseline"][k] = _sig(cb[k])
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:200]
    line["dense_rows_rel_err"] = _sig(detail.get("dense_rows_rel_err"))
    dh = detail.get("dropin_host_buffers")
    if dh and "error" not in dh and dh.get("unchanged_caller_ms"):   # what the reference's own caller gets (rbf.rs:1357-1364)
        line["value_host_buffers"] = _sig(1e3 / dh["unchanged_caller_ms"])
    if dh and "error" not in dh:   # PCIe-inclusive, host buffers: the patched and the unchanged caller (never `value`)
        line["dropin_host_buffers_ms"] = {"patched": _sig(dh.get("patched_caller_ms")), "unchanged": _sig(dh.get("unchanged_caller_ms")),
                                          "unchanged_general_path": _sig(dh.get("unchanged_caller_general_path_ms"))}
    if detail.get("partition_covers_every_row_once") is not None:
        line["partition_covers_every_row_once"] = detail["partition_covers_every_row_once"]
    line["source_hash"] = detail.get("source_hash")
    cfgs = {}
    for name, c in (detail.get("configs") or {}).items():
        if not isinstance(c, dict):
            continue
        if "error" in c:
            cfgs[name] = {"error": str(c["error"])[:120]}
        elif "roofline" in c:
            r = c["roofline"]
            cfgs[name] = {"ms_per_step": _sig(c.get("ms_per_step")), "dense_rows_rel_err": _sig(c.get("dense_rows_rel_err")),
                          "roofline": {"kernel": r.get("kernel"), "bound": r.get("bound"), "frac": _sig(r.get("frac"))}}
            if c.get("note"):
                cfgs[name]["note"] = str(c["note"])[:140]
        else:                                            # config 3 end to end
            cfgs[name] = {}
            for label in ("for_points", "reference_defaults"):
                e = c.get(label)
                if e:
                    hist = e.get("residual_history") or [None]
                    cfgs[name][label] = {"iterations": e.get("iterations"), "solve_s": _sig(e.get("solve_s")),
                                         "setup_s": _sig(e.get("setup_s")), "converged": e.get("converged"),
                                         "stagnated": e.get("stagnated"), "final_residual": hist[-1]}
    if cfgs:
        line["configs"] = cfgs
    line["detail_file"] = detail.get("detail_file", "bench_detail.json")
    return line


def run_config3_solve(F, points=10_000_000, defaults_outer=4):
    """BASELINE.json config 3 end to end (opt-in, `--configs solve`): thin-plate spline, order 9, linear drift, smooth
    values, FGMRES 20 x 5 to 1e-6 relative, right-preconditioned by the multi-level Schwarz sweep -- with
    DDMParams.for_points (the extension that keeps three fine levels; it converges) and with the reference's default
    DDMParams (config.rs:60-69; it stagnates at this size, DESIGN.md section 9: the flag records that)."""
    from ferreus_rbf_rs_amd import solvers as S
    from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
    n = points
    rng = np.random.default_rng(42)
    pts = rng.random((n, 3))
    vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.5 * pts[:, 2] ** 2
    t0 = time.time()
    tree = F.FmmTree(pts, 9, F.KernelParams(F.KernelType["ThinPlateSplineRbf"]), True, True)
    t_tree = time.time() - t0
    st = InterpolantSettings(1, 3, nugget=0.0)
    rhs = np.concatenate([vals, np.zeros(st.basis_size)])
    out = {"workload": f"{n} uniform 3D points, ThinPlateSplineRbf, order 9, linear drift, FGMRES 20 x 5 + Schwarz, "
                       "tolerance 1e-6 relative", "fmm_tree_build_s": t_tree}
    for label, params, max_outer in (("for_points", DDMParams.for_points(n), 20), ("reference_defaults", DDMParams(), defaults_outer)):
        t0 = time.time()
        pre = SchwarzPreconditioner(tree, pts, st, params)
        t_ddm = time.time() - t0
        op = S.RbfSystemOperator(tree, st.basis_size, pre.monomial_matrix, 0.0)
        t0 = time.time()
        x, hist = S.fgmres(op, rhs, pre, None, max_outer, 5, S.FittingAccuracy(1e-6))
        t_solve = time.time() - t0
        res = [r for _, r in hist]
        converged = bool(res and res[-1] <= 1e-6)
        idx = rng.choice(n, 2000, replace=False)
        fit = float(np.abs(op(x)[idx] - vals[idx]).max())
        out[label] = {"ddm_params": {"leaf_threshold": params.leaf_threshold, "overlap_quota": params.overlap_quota,
                                     "coarse_ratio": params.coarse_ratio, "coarse_threshold": params.coarse_threshold},
                      "levels": pre.num_levels, "setup_s": t_ddm, "solve_s": t_solve, "iterations": len(hist),
                      "converged": converged,
                      "stagnated": bool(len(res) >= 10 and res[-1] > 0.5 * res[-6]),
                      "max_outer_iterations": max_outer,
                      "residual_history": [float("%.3e" % r) for r in res], "max_fit_error_on_sample": fit}
        del pre, op
    return out


def main_inprocess(args):
    """`--inprocess`: one process, ONE handle over the devices (bbfmm_create_on_devices).  `value` is the
    device-resident product (weights and result in the HBM of the first device, as the default line), `value_host_buffers`
    the unchanged caller's sequence on host buffers (bbfmm_set_weights + bbfmm_evaluate at the sources, PCIe inclusive) --
    the path a ferreus_rbf caller actually takes.  Phase times are per part (each on its own device's stream)."""
    import ctypes
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd import _lib as L
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    logical = len(set(devices)) < len(devices)
    torch.cuda.set_device(devices[0])
    dev = torch.device("cuda", devices[0])
    N, K = args.points, args.nrhs
    pts = np.random.default_rng(42).random((N, 3))
    t0 = time.time()
    tree = F.FmmTree(pts, args.order, F.KernelParams(F.KernelType[args.kernel], base_range=args.base_range,
                                                     total_sill=args.total_sill), True, True, devices=devices)
    t_build = time.time() - t0
    G = tree.device_count()
    stats = tree.stats()
    w = torch.from_numpy(np.random.default_rng(43).random((K, N))).to(dev)
    out = torch.zeros((K, N), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)

    def sync():
        torch.cuda.synchronize()
        stream.synchronize()

    for _ in range(args.warmup):
        tree.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, sync=False)
    sync()
    tree.set_profiling(True)
    tree.phase_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tree.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, sync=False)
    sync()
    elapsed = time.perf_counter() - t0
    parts = []
    for g in range(G):
        ms, cnt = tree.part_phase_ms(g)
        parts.append({k: ms[k] / args.steps for k in ms})
    tree.set_profiling(False)
    err = dense_rows_err(torch, dev, args.kernel, args.base_range, args.total_sill, pts, w, out)
    # the unchanged caller on host buffers through the same handle
    lib = L.load()
    x = np.asfortranarray(pts)
    wh = np.asfortranarray(w.cpu().numpy().T.copy())
    yh = np.zeros((N, K), order="F")
    bad = ctypes.c_int64(-1)

    def unchanged():
        rc = lib.bbfmm_set_weights(tree._h, wh.ctypes.data, N, K, N)
        return rc or lib.bbfmm_evaluate(tree._h, wh.ctypes.data, N, K, N, x.ctypes.data, N, N, yh.ctypes.data, N, ctypes.byref(bad))

    assert unchanged() == 0 and tree.last_evaluate_at_sources() == 1
    ts = []
    for _ in range(max(args.steps, 3)):
        t0 = time.perf_counter()
        unchanged()
        ts.append(time.perf_counter() - t0)
    host_ms = sorted(ts)[len(ts) // 2] * 1e3
    host_vs_device = float(np.abs(yh.T - out.cpu().numpy()).max() / np.abs(yh).max())
    sums = [sum(p.values()) for p in parts]
    slow = int(np.argmax(sums))
    per_launch = parts[slow]
    dominant, roofline = roofline_of(stats, N, K, args.kernel, per_launch, G, None, None)
    roofline["part"] = slow
    line = {
        "metric": "BBFMM matvecs/s", "value": args.steps / elapsed, "unit": "matvecs/s", "n_gpus": len(set(devices)),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "entry": "inprocess (one process, one handle over the devices; not the default one-process-per-GPU line)",
        "value_host_buffers": 1e3 / host_ms, "ms_per_step_host_buffers": host_ms,
        "host_vs_device_resident_rel_diff": host_vs_device,
        "config": {"workload": f"{N} uniform 3D points, {args.kernel}, order {args.order}, {K} rhs, adaptive sparse tree, "
                               f"ACA eps=1e-{args.order}, set_weights + evaluate at the sources",
                   "points": N, "kernel": args.kernel, "order": args.order, "nrhs": K,
                   "parallelism": f"ONE process, device group {devices}: slot exchange of the coarse multipoles by peer copies, "
                                  "owned blocks back to the first device (device-resident) / to the host (host buffers)"
                                  + ("; LOGICAL parts on one device: a functional rehearsal, never a scaling number" if logical else "")},
        "roofline": roofline, "dense_rows_rel_err": err, "dense_rows": 32,
        "phase_ms_per_step_per_part": parts, "part_sum_ms": sums,
        "tree": {"depth": stats.depth, "cells": stats.n_cells, "leaves": stats.n_leaves, "build_s_all_parts": t_build},
        "group_bounds": [int(b) for b in tree.group_bounds()],
        "source_hash": source_hash(),
    }
    write_outputs(line, real_stdout, args.detail_file)


def write_outputs(detail, stdout_fd, path=None):
    """The full record to `bench_detail.json` beside this file (best effort: a read-only tree only loses the file) and to
    stderr; the compact line -- one line, under 4 KB -- to the real stdout."""
    path = path or os.path.join(ROOT, "bench_detail.json")
    detail["detail_file"] = os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
    text = json.dumps(detail, indent=1)
    try:
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        detail["detail_file"] = f"stderr only ({e.__class__.__name__})"
    sys.stderr.write("[bench detail] " + json.dumps(detail) + "\n")
    sys.stderr.flush()
    line = compact_line(detail)
    out = json.dumps(line)
    for drop in ("configs", "p2p"):          # never reached with the configurations above; a line must stay parseable
        if len(out) >= LINE_LIMIT:
            line.pop(drop, None)
            out = json.dumps(line)
    sys.stdout.flush()
    os.write(stdout_fd, (out + "\n").encode())


def main():
    args = parse()
    if args.inprocess:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            sys.exit("--inprocess is one process: start it without a launcher")
        return main_inprocess(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))        # nothing has touched the GPU in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line: whatever a library prints there (gloo's connection notes, RCCL's banner) goes to
    # stderr instead -- file descriptor 1 is pointed at stderr for the whole run and the line is written to the saved one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1:  # every rank builds the whole tree: share the host cores instead of 8 x 64 setup threads
        os.environ.setdefault("BBFMM_HOST_THREADS", str(max(4, (os.cpu_count() or 8) // world)))

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # self-launched ranks meet through a file store; under torch.distributed.run the environment names the master
        rdzv = os.environ.get("BBFMM_RDZV_FILE")
        how = {"init_method": f"file://{rdzv}", "rank": rank, "world_size": world} if rdzv else {}
        if args.exchange == "gloo":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            dist.init_process_group("gloo", **how)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **how)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import ferreus_rbf_rs_amd as F
    kernel_id = int(F.KernelType[args.kernel])
    want = {c.strip() for c in args.configs.split(",") if c.strip()}

    N, K = args.points, args.nrhs
    # synthetic inputs: i.i.d. uniform [0,1)^3 points, uniform [0,1) weights
    pts = np.random.default_rng(42).random((N, 3))
    t0 = time.time()
    tree = F.FmmTree(pts, args.order, F.KernelParams(F.KernelType(kernel_id), base_range=args.base_range,
                                                     total_sill=args.total_sill), True, True)
    t_build = time.time() - t0
    stats = tree.stats()

    w = torch.from_numpy(np.random.default_rng(43).random((K, N))).to(dev)   # K x N, rhs-major
    out = torch.zeros((K, N), dtype=torch.float64, device=dev)

    pm = None
    if world > 1:
        from ferreus_rbf_rs_amd.distributed import PartitionedMatvec
        tree.set_partition(rank, world)
        pm = PartitionedMatvec(tree, N, K, dev)   # all-reduce of the coarse multipoles + all-gather of the owned rows
        assert pm.check_partition(), "partition does not cover the targets exactly once"

    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    elapsed, phases, counts = time_matvecs(torch, dist, tree, w, out, args.steps, args.warmup, world, pm, stream)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.exchange == "rccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed                     # whole-job matvecs/s (one matvec spans all ranks)
    default_workload = (N, args.kernel, args.order, K) == (10_000_000, "LinearRbf", 7, 1)

    line = None
    valu_lane_rate = None
    if rank == 0:
        per_launch = {k: phases[k] / args.steps for k in phases}
        n = stats.n_nodes
        C = stats.n_cells
        micro = {}
        try:  # what the vector pipe sustains on this device (the pair kernels' issue roofline) and at which clock
            vtf, vmhz = F.fp64_valu_selftest()
            valu_lane_rate = vtf * 1e12 / 2.0
            micro["fp64_valu_microbench"] = {"tflops": vtf, "clock_mhz": vmhz, "lane_instr_per_s": valu_lane_rate}
        except Exception:  # noqa: BLE001
            micro["fp64_valu_microbench"] = None
        if world == 1:
            try:
                tf, errs = F.mfma_f64_selftest()
                micro["fp64_mfma_microbench_tflops"] = tf
            except Exception:  # noqa: BLE001
                micro["fp64_mfma_microbench_tflops"] = None
        # algorithmic work per launch (this rank; at N=1 the whole matvec) -- DESIGN.md section 5
        dominant, roofline = roofline_of(stats, N, K, args.kernel, per_launch, world,
                                         {k: counts[k] // max(args.steps, 1) for k in counts}, valu_lane_rate)
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
                       "parallelism": (f"target-subtree partition x{world}: all-reduce of {pm.count * 8 * K / 1e6:.1f} MB "
                                       f"coarse multipoles + all-gather of owned potentials, {args.exchange}")
                       if world > 1 else "single GPU"},
            "roofline": roofline,
            "achieved_hbm_gbps_compulsory": compulsory_bytes / (elapsed / args.steps) * 1e-9,
            "phase_ms_per_step": {k: phases[k] / args.steps for k in phases},
            "tree": {"depth": stats.depth, "cells": C, "leaves": stats.n_leaves, "v_pairs": stats.n_v,
                     "p2p_pairs": stats.p2p_pairs, "m2l_flops_k1": stats.m2l_flops_k1,
                     "build_s": t_build},
            "source_hash": source_hash(),
            "partition_covers_every_row_once": True if world > 1 else None,     # asserted above on every rank
            **micro,
        }
        if world == 1:
            line["phase_roofline"] = phase_roofline(stats, N, K, args.order, args.kernel, line["phase_ms_per_step"],
                                                    valu_lane_rate)
        # the result every rank now holds, against 32 rows of the dense sum (plain torch on rank 0) -- at every world size
        line["dense_rows_rel_err"] = dense_rows_err(torch, dev, args.kernel, args.base_range, args.total_sill, pts, w, out)
        line["dense_rows"] = 32

    if rank == 0 and world == 1 and default_workload and K == 1 and args.dropin != "off":
        try:
            line["dropin_host_buffers"] = dropin_host_buffer_times(tree, pts)
        except Exception as e:  # noqa: BLE001
            line["dropin_host_buffers"] = {"error": f"{type(e).__name__}: {e}"}
    extra = {}
    if world == 1 and default_workload and want & {"auto", "extensions", "solve"}:
        todo = (EXTRA_CONFIGS if "auto" in want else []) + (EXTENSION_CONFIGS if "extensions" in want else [])
        # configs on the headline tree first (more rhs), then the tree is released for the others
        ordered = sorted(todo, key=lambda c: (c["points"], c["kernel"], c["order"]) != (N, args.kernel, args.order))
        for cfg in ordered:
            reuse = tree if (cfg["points"], cfg["kernel"], cfg["order"]) == (N, args.kernel, args.order) and \
                not cfg["name"].startswith(("extension_", "tuning_")) else None
            if reuse is None and tree is not None:
                del tree, w, out, stream
                tree = w = out = stream = None
                torch.cuda.empty_cache()
            try:
                extra[cfg["name"]] = run_config(torch, dist, F, dev, cfg, 1, 0, args.exchange, reuse, valu_lane_rate)
            except Exception as e:  # noqa: BLE001
                extra[cfg["name"]] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        if "solve" in want:
            if tree is not None:
                del tree, w, out, stream
                tree = w = out = stream = None
                torch.cuda.empty_cache()
            try:
                extra["config3_solve_tps_10M_fgmres_schwarz"] = run_config3_solve(F)
            except Exception as e:  # noqa: BLE001
                extra["config3_solve_tps_10M_fgmres_schwarz"] = {"error": f"{type(e).__name__}: {e}"}
    elif world > 1 and ((default_workload and "auto" in want) or "config5" in want):
        # config 5 (40M Spheroidal3) partitioned over the same ranks; every rank takes part
        del tree, w, out, stream, pm
        tree = w = out = stream = pm = None
        torch.cuda.empty_cache()
        cfg = dict(CONFIG5, points=args.config5_points)
        try:
            res = run_config(torch, dist, F, dev, cfg, world, rank, args.exchange, None, valu_lane_rate)
        except Exception as e:  # noqa: BLE001
            res = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            extra[cfg["name"]] = res
    if rank == 0:
        if extra:
            line["configs"] = extra
        if world == 1 and args.cpu_baseline != "off":
            try:
                line["cpu_baseline"], line["cpu_baseline_detail"] = cpu_baseline(args, kernel_id)
            except Exception as e:  # noqa: BLE001  (a missing C compiler on the box must not cost the GPU line)
                line["cpu_baseline"] = {"value": None, "unit": "matvecs/s", "cores": None, "kind": "port",
                                        "sample": f"not measured: {type(e).__name__}: {e}"}
        write_outputs(line, real_stdout, args.detail_file)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
