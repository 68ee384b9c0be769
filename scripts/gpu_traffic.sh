#!/bin/bash
# HBM-side traffic of the matvec kernels at the bench workload (MI355X_MICROARCH.md "HBM"): separate
# --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), KiB units, FETCH_SIZE x2 on
# gfx950 for 16-B-per-lane reads.  Writes gpurun_out/traffic_summary.txt and gpurun_out/r01_traffic.json.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_$c -- python3 $ROOT/bench.py --points 10000000 --steps 2 --warmup 1 --cpu-baseline off > $ROOT/gpurun_out/pmc_$c.txt 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os,json
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for c in ('FETCH_SIZE','WRITE_SIZE'):
    for f in glob.glob(root+'/pmc_%s/*/*_counter_collection.csv'%c):
        for r in csv.DictReader(open(f)):
            name=r['Kernel_Name'].split('(')[0].replace('void bbfmm::','').replace('bbfmm::','')[:40]
            agg[name][r['Counter_Name']]+=float(r['Counter_Value']); cnt[name][r['Counter_Name']]+=1
phase={'m2l_gemm_k4<11, 1, 1>':'M2L_stage1','m2l_gemm_k4<22, 2, 1>':'M2L_stage2','m2l_gemm_k4<11, 2, 1>':'M2L_stage2','p2p_kernel<0, false, 1>':'P2P'}
per={}
with open(root+'/traffic_summary.txt','w') as o:
    for k,v in sorted(agg.items()):
        nf=max(cnt[k]['FETCH_SIZE'],1); nw=max(cnt[k]['WRITE_SIZE'],1)
        fb=2*v['FETCH_SIZE']/nf*1024; wb=v['WRITE_SIZE']/nw*1024
        line='%-42s launches=%d FETCH_SIZE_raw_KiB/launch=%.0f (x2 gfx950 => %.3f GB) WRITE_SIZE_KiB/launch=%.0f (%.3f GB)'%(k,cnt[k]['FETCH_SIZE'],v['FETCH_SIZE']/nf,fb/1e9,v['WRITE_SIZE']/nw,wb/1e9)
        print(line); o.write(line+'\n')
        if k in phase: per[phase[k]]=fb+wb
json.dump({'points':10000000,'kernel':'LinearRbf','order':7,'nrhs':1,'per_launch_bytes':per,
           'note':'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = 2*FETCH_SIZE_KiB*1024 + WRITE_SIZE_KiB*1024 per launch (L2 memory-side requests; Infinity Cache hits are counted)'},
          open(root+'/r01_traffic.json','w'),indent=1)
PY
