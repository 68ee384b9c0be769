#!/bin/bash
# HBM traffic of the matvec kernels (guide: separate --pmc passes; FETCH_SIZE x2 on gfx950 for
# wide coalesced reads; KiB units)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_$c -- python3 $ROOT/bench.py --points 10000000 --steps 2 --warmup 1 --cpu-baseline off > $ROOT/gpurun_out/pmc_$c.txt 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for c in ('FETCH_SIZE','WRITE_SIZE'):
    for f in glob.glob(root+'/pmc_%s/*/*_counter_collection.csv'%c):
        for r in csv.DictReader(open(f)):
            name=r['Kernel_Name'].split('(')[0].replace('void bbfmm::','').replace('bbfmm::','')[:40]
            agg[name][r['Counter_Name']]+=float(r['Counter_Value']); cnt[name][r['Counter_Name']]+=1
with open(root+'/traffic_summary.txt','w') as o:
    for k,v in sorted(agg.items()):
        line='%-42s launches=%d FETCH_SIZE_raw_KiB/launch=%.0f (x2 gfx950 => %.3f GB) WRITE_SIZE_KiB/launch=%.0f (%.3f GB)'%(k,cnt[k]['FETCH_SIZE'],v['FETCH_SIZE']/max(cnt[k]['FETCH_SIZE'],1),2*v['FETCH_SIZE']/max(cnt[k]['FETCH_SIZE'],1)*1024/1e9,v['WRITE_SIZE']/max(cnt[k]['WRITE_SIZE'],1),v['WRITE_SIZE']/max(cnt[k]['WRITE_SIZE'],1)*1024/1e9)
        print(line); o.write(line+'\n')
PY
