#!/bin/bash
# Where M2L stage 1 (m2l_gemm_k4<11, 1, 1>, the bench's dominant kernel) loses against the FP64 matrix peak: the headline
# bench with the scatter stores as they are, switched off (BBFMM_M2L_S1_STORES=off: wrong results, timing only) and as
# non-temporal stores, each with one PMC pass per counter (no tracing domains besides kernel-trace).
#   bash scripts/gpu_stage1_experiments.sh r03_x   (through gpurun)  -> gpurun_out/<tag>_stage1_experiments.{txt,json}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-r03_x}
COUNTERS="MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VMEM_WR_TA_DATA_FIFO_FULL LdsUtil"
for v in default off nt; do
  if [ $v = default ]; then unset BBFMM_M2L_S1_STORES; else export BBFMM_M2L_S1_STORES=$v; fi
  python3 bench.py --steps 20 --warmup 3 --configs off --cpu-baseline off > gpurun_out/${TAG}_s1_$v.json 2>/dev/null
  [ $v = nt ] && continue
  cd /tmp && export TMPDIR=/tmp
  for c in $COUNTERS; do
    rm -rf $ROOT/gpurun_out/pmc_s1_${v}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_s1_${v}_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-baseline off --configs off > /dev/null 2>&1
  done
  cd $ROOT
done
unset BBFMM_M2L_S1_STORES
# the per-step synchronisation of both stages, timing only (wrong results): no wait for the LDS-DMA / no workgroup barrier
for v in nowait nobar; do
  BBFMM_M2L_DEBUG_SYNC=$v python3 bench.py --steps 20 --warmup 3 --configs off --cpu-baseline off > gpurun_out/${TAG}_s1_sync_$v.json 2>/dev/null
done
python3 - "$TAG" "$COUNTERS" <<'PY'
import csv, glob, json, os, sys
tag, counters = sys.argv[1], sys.argv[2].split()
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
res = {"source_hash": bench.source_hash(), "kernel": "m2l_gemm_k4<11, 1, 1> (M2L stage 1), 10M uniform points, LinearRbf, order 7", "variants": {}}
for v in ("default", "off", "nt"):
    j = json.load(open(root + '/%s_s1_%s.json' % (tag, v)))
    e = {"stage1_ms": j["phase_ms_per_step"]["M2L_stage1"], "stage2_ms": j["phase_ms_per_step"]["M2L_stage2"], "matvec_ms": j["ms_per_step"],
         "algorithmic_tflops": j["tree"]["m2l_flops_k1"] / 2 / (j["phase_ms_per_step"]["M2L_stage1"] * 1e-3) * 1e-12,
         "bare_mfma_loop_tflops": j.get("fp64_mfma_microbench_tflops"), "dense_rows_rel_err": j["dense_rows_rel_err"]}
    for c in counters:
        vals = []
        for f in glob.glob(root + '/pmc_s1_%s_%s/*/*_counter_collection.csv' % (v, c)):
            for r in csv.DictReader(open(f)):
                if r['Kernel_Name'].startswith('void bbfmm::m2l_gemm_k4<11, 1, 1>') and r['Counter_Name'] == c:
                    vals.append(float(r['Counter_Value']))
        if vals:
            e[c] = sum(vals) / len(vals)
    res["variants"][v] = e
res["sync"] = {}
for v in ("nowait", "nobar"):
    j = json.load(open(root + '/%s_s1_sync_%s.json' % (tag, v)))
    res["sync"][v] = {"stage1_ms": j["phase_ms_per_step"]["M2L_stage1"], "stage2_ms": j["phase_ms_per_step"]["M2L_stage2"]}
res["sync"]["note"] = ("BBFMM_M2L_DEBUG_SYNC=nowait: the per-step s_waitcnt vmcnt(0) for the LDS-DMA left out; nobar: the per-step "
                       "workgroup barrier left out -- both stages, timing only")
json.dump(res, open(root + '/%s_stage1_experiments.json' % tag, 'w'), indent=1)
lines = ["%-10s %s" % (v, "  ".join("%s=%s" % (k, ("%.4g" % x) if isinstance(x, float) else x) for k, x in e.items())) for v, e in res["variants"].items()]
lines += ["sync %-7s stage1_ms=%.4g stage2_ms=%.4g" % (v, e["stage1_ms"], e["stage2_ms"]) for v, e in res["sync"].items() if v != "note"]
open(root + '/%s_stage1_experiments.txt' % tag, 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
