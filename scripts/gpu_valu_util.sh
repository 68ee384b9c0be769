#!/bin/bash
# VALU-side counters of the direct-evaluation kernels at the bench workload (the near field is FP64-VALU bound):
# derived rocprofv3 metrics VALUBusy (% of time vector instructions are processed), VALUUtilization (% active
# lanes) and LDSBankConflict, one per --pmc pass.  Writes gpurun_out/valu_util_summary.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in VALUBusy VALUUtilization LDSBankConflict; do
  rm -rf $ROOT/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_$c -- python3 $ROOT/bench.py --points 10000000 --steps 2 --warmup 1 --cpu-baseline off > $ROOT/gpurun_out/pmc_$c.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('VALUBusy', 'VALUUtilization', 'LDSBankConflict'):
    for f in glob.glob(root + '/pmc_%s/*/*_counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void bbfmm::', '').replace('bbfmm::', '')[:40]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
with open(root + '/valu_util_summary.txt', 'w') as o:
    for k, v in sorted(agg.items()):
        if not any(s in k for s in ('p2p', 'm2p', 'p2l', 'l2p', 'p2m', 'm2l', 'm2m', 'l2l')): continue
        m = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else float('nan')
        line = '%-42s launches=%d VALUBusy=%.1f%% VALUUtilization=%.1f%% LDSBankConflict=%.2f%%' % (
            k, len(v.get('VALUBusy', [])), m('VALUBusy'), m('VALUUtilization'), m('LDSBankConflict'))
        print(line); o.write(line + '\n')
PY
