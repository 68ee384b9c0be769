#!/bin/bash
# bench at the default workload + rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-x}
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -2 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline off > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_stdout.txt 2>&1
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_stdout.txt | cut -c1-300
head -14 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/*/*_kernel_stats.csv | cut -c1-160
cp $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.csv
