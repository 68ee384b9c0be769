#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "11 22 2" "11 22 3" "11 22 1" "11 11 2"; do
  set -- $cfg
  echo "== S1=$1 S2=$2 ZSPLIT=$3"
  BBFMM_M2L_NG16_S1=$1 BBFMM_M2L_NG16_S2=$2 BBFMM_M2L_ZSPLIT=$3 python bench.py --steps 6 --warmup 2 --cpu-baseline off 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],2),'matvecs/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['phase_ms_per_step'].items() if v>0.5})"
done
