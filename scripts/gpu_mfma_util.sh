#!/bin/bash
# MFMA utilisation and executed FP64 MFMA flops per kernel at the bench workload (north-star: "MFMA-utilisation
# counters against chip peak").  Derived rocprofv3 metrics, one per pass:
#   MfmaUtil     = 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM)
#   MfmaFlopsF64 = 512 * SQ_INSTS_VALU_MFMA_MOPS_F64
# Writes gpurun_out/mfma_util_summary.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in MfmaUtil MfmaFlopsF64; do
  rm -rf $ROOT/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_$c -- python3 $ROOT/bench.py --points 10000000 --steps 2 --warmup 1 --cpu-baseline off > $ROOT/gpurun_out/pmc_$c.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('MfmaUtil', 'MfmaFlopsF64'):
    for f in glob.glob(root + '/pmc_%s/*/*_counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void bbfmm::', '').replace('bbfmm::', '')[:40]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
            agg[name]['dur_' + c].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
with open(root + '/mfma_util_summary.txt', 'w') as o:
    for k, v in sorted(agg.items()):
        if not v.get('MfmaFlopsF64') or max(v['MfmaFlopsF64']) == 0: continue
        util = sum(v['MfmaUtil']) / max(len(v['MfmaUtil']), 1) if v.get('MfmaUtil') else float('nan')
        fl = sum(v['MfmaFlopsF64']) / len(v['MfmaFlopsF64'])
        dur = sum(v['dur_MfmaFlopsF64']) / len(v['dur_MfmaFlopsF64']) * 1e-9
        line = '%-42s launches=%d MfmaUtil=%.1f%% executed_FP64_MFMA_flops/launch=%.4e duration_under_counters=%.2f ms => %.1f TFLOP/s executed' % (
            k, len(v['MfmaFlopsF64']), util, fl, dur * 1e3, fl / dur * 1e-12)
        print(line); o.write(line + '\n')
PY
