#!/bin/bash
# rocprofv3 kernel statistics of the config-3 solve (10M points, thin-plate spline, FGMRES + Schwarz)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_config3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_config3 -o c3 -- python3 $R/scripts/solve_config3.py --points ${1:-10000000} --coarse-threshold 0 > $R/gpurun_out/prof_config3_stdout.txt 2>&1
grep '^{' $R/gpurun_out/prof_config3_stdout.txt | cut -c1-300
cp $R/gpurun_out/prof_config3/c3_kernel_stats.csv $R/gpurun_out/config3_kernel_stats.csv 2>/dev/null || cp $(find $R/gpurun_out/prof_config3 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/config3_kernel_stats.csv
head -16 $R/gpurun_out/config3_kernel_stats.csv | cut -c1-120
rm -f $(find $R/gpurun_out/prof_config3 -name "*kernel_trace.csv")
