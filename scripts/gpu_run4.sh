#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/rocprof_counters.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/rocprof_counters.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --points 2000000 --steps 1 --warmup 0 --cpu-baseline off > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt | cut -c1-300
done
find $GRAFT_REPO_ROOT/gpurun_out -name "*counter_collection.csv" | head
