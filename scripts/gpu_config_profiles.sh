#!/bin/bash
# Kernel stats + PMC passes for the configurations beside the headline (BASELINE.json configs 2, 3-operator, 4, and the
# 40M-point workload of config 5 on one GPU), named by round:
#   bash scripts/gpu_config_profiles.sh r03_x      (through gpurun; outputs under gpurun_out/, copy to profiles/)
# Per config: rocprofv3 --kernel-trace --stats of tests/checks/check_config.py            -> <tag>_<cfg>_kernel_stats.csv
#             one PMC pass per counter / derived metric (FETCH_SIZE, WRITE_SIZE, VALUBusy, MfmaUtil), no other tracing
#             domains                                                                      -> <tag>_<cfg>_counters.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-r03_x}
run_cfg() {
  local name=$1; shift
  cd /tmp && export TMPDIR=/tmp
  rm -rf $ROOT/gpurun_out/prof_${TAG}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_$name -- python3 $ROOT/tests/checks/check_config.py "$@" > $ROOT/gpurun_out/${TAG}_${name}_under_rocprof.json 2>/dev/null
  cp $ROOT/gpurun_out/prof_${TAG}_$name/*/*_kernel_stats.csv $ROOT/gpurun_out/${TAG}_${name}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE VALUBusy MfmaUtil; do
    rm -rf $ROOT/gpurun_out/pmc_${name}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_${name}_$c -- python3 $ROOT/tests/checks/check_config.py --steps 1 "$@" > /dev/null 2>&1
  done
  cd $ROOT
  python3 - "$TAG" "$name" <<'PY'
import csv, glob, collections, os, sys
tag, name = sys.argv[1], sys.argv[2]
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'VALUBusy', 'MfmaUtil'):
    for f in glob.glob(root + '/pmc_%s_%s/*/*_counter_collection.csv' % (name, c)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void bbfmm::', '').replace('bbfmm::', '')[:44]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
mean = lambda v: sum(v) / len(v) if v else None
lines = []
for k, v in sorted(agg.items()):
    if not any(s in k for s in ('m2l_gemm', 'p2p', 'p2m', 'l2p', 'wx_sym', 'm2p', 'p2l')):
        continue
    line = '%-46s launches=%d' % (k, max(len(x) for x in v.values()))
    if v.get('FETCH_SIZE'): line += ' FETCH_SIZE x2 = %.3f GB' % (2 * mean(v['FETCH_SIZE']) * 1024 / 1e9)
    if v.get('WRITE_SIZE'): line += ' WRITE_SIZE = %.3f GB' % (mean(v['WRITE_SIZE']) * 1024 / 1e9)
    if v.get('VALUBusy'): line += ' VALUBusy=%.1f%%' % mean(v['VALUBusy'])
    if v.get('MfmaUtil') and 'm2l' in k: line += ' MfmaUtil=%.1f%%' % mean(v['MfmaUtil'])
    lines.append(line)
open(root + '/%s_%s_counters.txt' % (tag, name), 'w').write('\n'.join(lines) + '\n')
print(name); print('\n'.join(lines))
PY
  head -8 $ROOT/gpurun_out/${TAG}_${name}_kernel_stats.csv | cut -c1-140
}
run_cfg config2_spheroidal3_1M --points 1000000 --kernel Spheroidal3Rbf --base-range 0.1 --total-sill 0.1
run_cfg config3_operator_tps_10M --points 10000000 --kernel ThinPlateSplineRbf --order 9
run_cfg config4_linear_10M_8rhs --points 10000000 --kernel LinearRbf --nrhs 8
run_cfg config5_size_spheroidal3_40M --points 40000000 --kernel Spheroidal3Rbf --base-range 0.1 --total-sill 0.1 --samples 16
