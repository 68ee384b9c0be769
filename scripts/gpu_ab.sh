#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "1 2" "8 2" "8 4"; do
  set -- $cfg
  echo "== STAGGER=$1 ZSPLIT=$2"
  BBFMM_M2L_STAGGER=$1 BBFMM_M2L_ZSPLIT=$2 python bench.py --steps 6 --warmup 2 --cpu-baseline off 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],2),'matvecs/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['phase_ms_per_step'].items() if v>0.5})"
done
