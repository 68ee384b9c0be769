#!/usr/bin/env python3
"""FP64 vector instructions per kernel evaluation of the pair loops (P2P, M2P, P2L), counted from the ISA.

  python scripts/pair_instruction_counts.py r03_x        -> profiles/r03_x_pair_instruction_counts.json

scripts/pair_probe.hip holds exactly one evaluation as the pair kernels inline it (distance from wave-uniform target
coordinates, kernel_value_r2<KID> of csrc/kernels.hpp, row and column accumulation) inside a loop that is not
unrolled; it is compiled for gfx950 with the library's flags, emitted as assembly (-S), and the instructions of
the loop body (the block that ends in the backward branch) are counted by class.  bench.py multiplies the count by the
kernel evaluations a phase executes and divides by the FP64 instruction-issue peak (`valu_issue`); the file is stamped
with the hash of csrc/kernels.hpp + scripts/pair_probe.hip (the only sources the count depends on) so that counts are
only quoted for the arithmetic they were taken on.  Runs without a GPU."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KIDS = {"LinearRbf": 0, "ThinPlateSplineRbf": 1, "CubicRbf": 2, "Spheroidal3Rbf": 3, "Spheroidal5Rbf": 4, "Spheroidal7Rbf": 5,
        "Spheroidal9Rbf": 6, "Laplacian": 7, "OneOverR2": 8, "OneOverR4": 9, "GaussianExt": 100, "MultiquadricExt": 101}
TRANS = re.compile(r"^v_(rsq|rcp|sqrt|log|exp|frexp_mant|ldexp)\w*_f64")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    tmp = tempfile.mkdtemp(prefix="pair_probe_")
    asm_path = os.path.join(tmp, "probe.s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           "-I", os.path.join(ROOT, "ferreus_rbf_rs_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "scripts", "pair_probe.hip"), "-o", asm_path])
    funcs = {}
    cur = None
    for line in open(asm_path).read().splitlines():
        m = re.match(r"^(_Z\w*pair_probe\w*):", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            funcs[cur].append(line)
    out = {}
    for name, kid in KIDS.items():
        body = next(v for k, v in funcs.items() if "pair_probeILi%dE" % kid in k)
        # the loop = the lines between a label and the LAST branch back to it
        label_at = {}
        loop = None
        for i, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                label_at[m.group(1)] = i
            m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in label_at:
                loop = (label_at[m.group(1)], i)
        assert loop, name
        cls = collections.Counter()
        for ln in body[loop[0]:loop[1] + 1]:
            m = re.match(r"^\s+(\w+)", ln)
            if not m:
                continue
            op = m.group(1)
            if re.match(r"^v_\w+_f64", op):
                cls["fp64_transcendental" if TRANS.match(op) else "fp64"] += 1
            elif op.startswith("v_cndmask") or op.startswith("v_cmp"):
                cls["select_compare"] += 1
            elif op.startswith("v_"):
                cls["other_valu"] += 1
        n64 = cls["fp64"] + cls["fp64_transcendental"]
        entry = {"fp64_valu_per_pair": n64, "of_which_transcendental": cls["fp64_transcendental"],
                 "select_compare_per_pair": cls["select_compare"], "other_valu_in_probe_loop": cls["other_valu"]}
        # the same inlined arithmetic in every pair kernel
        out[name] = {k: entry for k in ("p2p_sym", "p2p", "wx_sym", "p2l", "m2p")}
    res = {"pair_probe_hash": bench.pair_probe_hash(), "kernels": out,
           "note": "v_*_f64 instructions of one kernel evaluation incl. the distance and the two accumulations "
                   "(scripts/pair_probe.hip, loop body between the backward branch and its target); transcendental seeds "
                   "(v_rsq_f64 / v_rcp_f64, about two FMA issue slots each on MI355X, scripts/rsq_rate.hip) are counted once"}
    path = os.path.join(ROOT, "profiles", "%s_pair_instruction_counts.json" % tag)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(path)
    for k, v in out.items():
        print("%-20s %s" % (k, v["p2p_sym"]))


if __name__ == "__main__":
    main()
