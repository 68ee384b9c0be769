#!/bin/bash
# What does the wave-per-leaf near field (p2p_sym2_kernel) wait for with 1 / 4 / 8 right-hand sides?  
#   bash scripts/gpu_nearfield_counters.sh <tag>      (through gpurun; outputs under gpurun_out/, copy to profiles/)
# 1.25M uniform points (the leaf occupancy of the 10M workload, one level shallower), LinearRbf order 7.  One raw SQ
# counter per rocprofv3 pass (--kernel-trace --pmc only: no other tracing domain), averaged over the launches of
# every p2p_sym2_kernel instance; SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles (guide: PMC slots).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-r05_nf}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $ROOT/gpurun_out/${TAG}_sq_counter_names.txt
COUNTERS=${NF_COUNTERS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SMEM SQ_WAVES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"}
for K in ${NF_RHS:-1 4 8}; do
  for c in $COUNTERS; do
    grep -qx "$c" $ROOT/gpurun_out/${TAG}_sq_counter_names.txt || continue
    rm -rf $ROOT/gpurun_out/pmc_${TAG}_k${K}_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_k${K}_$c -- python3 $ROOT/tests/checks/check_config.py --points 1250000 --nrhs $K --steps 1 > /dev/null 2>&1
  done
done
cd $ROOT
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, os, sys
tag = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
out = {}
for K in (1, 4, 8):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob(root + '/pmc_%s_k%d_*' % (tag, K)):
        for f in glob.glob(d + '/*/*_counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].split('(')[0].replace('void bbfmm::', '').replace('bbfmm::', '')
                if 'p2p_sym' in k or 'wx_sym' in k:
                    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    out['nrhs_%d' % K] = {k: {c: {'mean_per_launch': sum(v) / len(v), 'launches': len(v)} for c, v in sorted(cs.items())} for k, cs in agg.items()}
json.dump(out, open(root + '/%s_nearfield_counters.json' % tag, 'w'), indent=1)
for K, ks in out.items():
    for k, cs in ks.items():
        g = lambda c: cs.get(c, {}).get('mean_per_launch')
        wc = g('SQ_WAVE_CYCLES')
        line = '%s %-40s launches=%d' % (K, k[:40], max(v['launches'] for v in cs.values()))
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS'):
                if g(c) is not None: line += ' %s=%.3f' % (c.replace('SQ_', ''), g(c) / wc)
        for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_VMEM_RD', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAVES'):
            if g(c) is not None: line += ' %s=%.4g' % (c.replace('SQ_', ''), g(c))
        print(line)
PY
