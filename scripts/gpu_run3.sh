#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tests/checks/gpu_first.py parity > gpurun_out/run3_parity.txt 2>&1
grep -E "^N=|EXC|Error" gpurun_out/run3_parity.txt | cut -c1-220
python tests/checks/gpu_first.py timings 2>&1 | grep -E "TIMING|phases|dense|Error|error" | cut -c1-400
