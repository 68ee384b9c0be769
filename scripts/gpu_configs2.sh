#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "== $*"; timeout 900 python tests/checks/check_config.py "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(c['points'],c['kernel'],c['order'],c['nrhs'],'ms=%.1f'%d['ms_per_matvec'],'err=%.2e'%d['rel_err_vs_dense_sampled'],'m2l_TF=%.1f'%d['m2l_tflops_algorithmic'],{k:v for k,v in d['phase_ms'].items() if v>0.3})"; }
run --points 10000000 --kernel ThinPlateSplineRbf --order 9 --steps 2
run --points 10000000 --kernel LinearRbf --order 8 --steps 2
run --points 10000000 --kernel LinearRbf --order 6 --steps 2
