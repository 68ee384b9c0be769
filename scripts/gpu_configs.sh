#!/bin/bash
# BASELINE.json configs 2-5 on one GPU (single-GPU legs): timing + sampled dense check
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/configs.jsonl; : > $O
run() { echo "== $*"; timeout 900 python tests/checks/check_config.py "$@" 2>gpurun_out/cfg_err.txt | tee -a $O | cut -c1-600; tail -3 gpurun_out/cfg_err.txt; }
run --points 1000000 --kernel Spheroidal3Rbf --base-range 0.1 --total-sill 0.1
run --points 1000000 --kernel MultiquadricExt --base-range 0.1 --total-sill 0.1
run --points 10000000 --kernel LinearRbf --nrhs 8 --steps 2
run --points 10000000 --kernel ThinPlateSplineRbf --order 9 --steps 2
run --points 10000000 --kernel GaussianExt --base-range 0.1 --total-sill 0.1 --steps 2
run --points 40000000 --kernel Spheroidal3Rbf --base-range 0.1 --total-sill 0.1 --steps 2 --samples 32
