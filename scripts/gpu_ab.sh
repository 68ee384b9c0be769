#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --warmup 2 --cpu-baseline off 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],2),'matvecs/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['phase_ms_per_step'].items() if v>0.3})"
