#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc3
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py --points 10000000 --steps 1 --warmup 0 --cpu-baseline off > $GRAFT_REPO_ROOT/gpurun_out/pmc3.txt 2>&1
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root+'/pmc3/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void bbfmm::','').replace('bbfmm::','')[:34]
        agg[name][r['Counter_Name']]+=float(r['Counter_Value'])
        agg[name]['dur_ns']=max(agg[name]['dur_ns'], float(r['End_Timestamp'])-float(r['Start_Timestamp']))
for k,v in sorted(agg.items()):
    if not any(s in k for s in ('m2l','p2p','l2p','p2m','m2p','mfma_peak')): continue
    d=v['dur_ns']*1e-9
    print(k, 'dur_ms=%.2f'%(d*1e3), 'GUI_ACTIVE=%.3e'%v['GRBM_GUI_ACTIVE'], 'clock_GHz(if /8)=%.2f'%(v['GRBM_GUI_ACTIVE']/8/d/1e9 if d else 0), 'clock_GHz(raw)=%.2f'%(v['GRBM_GUI_ACTIVE']/d/1e9 if d else 0), {c:'%.3e'%x for c,x in v.items() if c not in('dur_ns','GRBM_GUI_ACTIVE')})
PY
