#!/usr/bin/env python3
"""Raw PMC counters of the kernels whose name contains a substring, one counter per rocprofv3 pass (--kernel-trace --pmc
only: no other tracing domain), averaged per launch.  This driver never touches the GPU; the profiled program follows `--`
and is started by rocprofv3 itself (python3 directly, no shell in between).

  python3 scripts/kernel_counters.py <out.json> <kernel substring[,substring]> [-c COUNTER,COUNTER,...] [-e NAME=VALUE ...] -- python3 prog.py args

Counters default to the set that tells a VALU-bound pair kernel's story: wave / busy cycles, instruction mix, waits, LDS."""
import collections, csv, glob, json, os, shutil, subprocess, sys, tempfile

DEFAULT = ("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA "
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD "
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 "
           "SQ_LEVEL_WAVES GRBM_GUI_ACTIVE").split()


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    opts, prog = argv[:cut], argv[cut + 1:]
    out, subs = opts[0], opts[1].split(",")
    counters, env = list(DEFAULT), dict(os.environ)
    i = 2
    while i < len(opts):
        if opts[i] == "-c":
            counters = opts[i + 1].split(",")
        elif opts[i] == "-e":
            k, v = opts[i + 1].split("=", 1)
            env[k] = v
        i += 2
    env["TMPDIR"] = "/tmp"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for c in counters:
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--"] + prog, cwd="/tmp", env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if any(s in k for s in subs):
                    agg[k.split("(")[0].replace("void bbfmm::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/*/*_kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if any(s in k for s in subs):
                    dur[k.split("(")[0].replace("void bbfmm::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        shutil.rmtree(d, ignore_errors=True)
    res = {k: {"launches": max(len(v) for v in cs.values()), "avg_ms_under_counters": sum(dur[k]) / max(len(dur[k]), 1),
               **{c: sum(v) / len(v) for c, v in sorted(cs.items())}} for k, cs in agg.items()}
    json.dump({"program": prog, "env": {k: v for k, v in env.items() if k.startswith("BBFMM_")}, "kernels": res}, open(out, "w"), indent=1)
    for k, cs in res.items():
        wc = cs.get("SQ_WAVE_CYCLES")
        line = f"{k[:50]:50s} launches={cs['launches']} ms={cs['avg_ms_under_counters']:.3f}"
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if c in cs:
                    line += f" {c[3:]}={cs[c] / wc:.3f}"
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                  "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"):
            if c in cs:
                line += f" {c[3:]}={cs[c]:.4g}"
        print(line)


if __name__ == "__main__":
    main()
