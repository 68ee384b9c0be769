#!/bin/bash
# usage: gpu_pmc.sh <points> ; collects two PMC passes for all kernels of one matvec
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PTS=${1:-10000000}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc$i
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --points $PTS --steps 1 --warmup 0 --cpu-baseline off > $GRAFT_REPO_ROOT/gpurun_out/pmc$i.txt 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
agg=collections.defaultdict(lambda: collections.defaultdict(float))
dur=collections.defaultdict(float)
for f in glob.glob(root+'/pmc[12]/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void bbfmm::','').replace('bbfmm::','')[:34]
        agg[name][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in sorted(agg.items()):
    if not any(s in k for s in ('m2l','p2p','l2p','p2m','m2p')): continue
    wc=v.get('SQ_WAVE_CYCLES',1)
    print(k)
    print('   ', {c: (round(x/wc,3) if c.startswith('SQ_WAIT') or c.startswith('SQ_ACTIVE_INST_ANY') else int(x)) for c,x in sorted(v.items())})
PY
