#!/bin/bash
# GPU session 2: remaining parity, timings, bench + rocprofv3 kernel trace
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tests/checks/gpu_first.py selftest parity timings > gpurun_out/gpu_first_stdout.txt 2>&1
tail -40 gpurun_out/gpu_first.log
python bench.py --points 1000000 --steps 5 --warmup 1 > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err
tail -3 gpurun_out/bench_1m.err; cat gpurun_out/bench_1m.json
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_10m.json 2> gpurun_out/bench_10m.err
tail -3 gpurun_out/bench_10m.err; cat gpurun_out/bench_10m.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline off > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_stdout.txt 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_r1_stdout.txt
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name "*stats*" | head
