#!/bin/bash
# usage: gpu_partial_prof.sh <points> <subset> <kernel_id> <order> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_$5 -o pp -- python3 $R/scripts/partial_matvec_profile.py $1 $2 $3 $4 > $R/gpurun_out/pp_$5.log 2>&1
grep "^{" $R/gpurun_out/pp_$5.log || tail -5 $R/gpurun_out/pp_$5.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/pp_$5/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
