#!/bin/bash
# One GPU job that produces every measurement the bench line quotes, named by round:
#   bash scripts/gpu_round_profiles.sh r02_b          (run through gpurun; outputs under gpurun_out/, copy to profiles/)
#  1. python bench.py (default: headline + the other single-GPU configs + cpu_baseline)      -> <tag>_bench.json (the <4 KB line)
#                                                                                              + <tag>_bench_detail.json (full record)
#  2. rocprofv3 --kernel-trace --stats of the same command (headline only)                    -> <tag>_kernel_stats.csv
#  3. PMC passes, one counter / derived metric per pass, no tracing domains besides kernel-trace:
#     FETCH_SIZE, WRITE_SIZE (HBM-side bytes; FETCH_SIZE x2 on gfx950 for 16-B/lane reads, KiB units),
#     MfmaUtil, MfmaFlopsF64 (matrix pipe), VALUBusy, VALUUtilization (near field)            -> <tag>_counters.json (+ .txt)
#     The JSON carries bench.py's source_hash: the bench only quotes counters taken on the sources it runs.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-r04_final}
python3 bench.py --steps 20 --warmup 5 --detail-file gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
wc -c gpurun_out/${TAG}_bench.json
grep -v "bench detail" gpurun_out/${TAG}_bench.err | tail -2; cut -c1-400 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline off --configs off --dropin off > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2>$ROOT/gpurun_out/prof_${TAG}.err
cp $ROOT/gpurun_out/prof_$TAG/*/*_kernel_stats.csv $ROOT/gpurun_out/${TAG}_kernel_stats.csv
head -12 $ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE MfmaUtil MfmaFlopsF64 VALUBusy VALUUtilization; do
  rm -rf $ROOT/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-baseline off --configs off --dropin off > $ROOT/gpurun_out/pmc_$c.txt 2>&1
done
cd $ROOT
python3 - "$TAG" <<'PY'
import csv, glob, collections, os, json, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
tag = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'MfmaUtil', 'MfmaFlopsF64', 'VALUBusy', 'VALUUtilization'):
    for f in glob.glob(root + '/pmc_%s/*/*_counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void bbfmm::', '').replace('bbfmm::', '')[:44]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
            if c == 'MfmaFlopsF64':
                agg[name]['dur_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
mean = lambda v: sum(v) / len(v) if v else None
phase = {'m2l_gemm_k4<11, 1, 1>': 'M2L_stage1', 'm2l_gemm_k4<22, 2, 1>': 'M2L_stage2', 'm2l_gemm_k4<11, 2, 1>': 'M2L_stage2', 'p2p_sym_kernel<0, 1>': 'P2P',
         'p2p_sym2_kernel<0, 1>': 'P2P', 'p2p_kernel<0, false, 1>': 'P2P', 'wx_sym_kernel<0, 1>': 'P2L',
         'p2p_sym3_kernel<0, 8>': 'P2P', 'p2p_sym3_kernel<0, 6>': 'P2P', 'wx_sym3_kernel<0, 6>': 'P2L', 'wx_sym3_kernel<0, 8>': 'P2L'}   # round 6: the whole-leaf kernels of a one-rhs pass
per_bytes, per_kernel, lines = {}, {}, []
for k, v in sorted(agg.items()):
    fb = 2 * mean(v.get('FETCH_SIZE')) * 1024 if v.get('FETCH_SIZE') else None
    wb = mean(v.get('WRITE_SIZE')) * 1024 if v.get('WRITE_SIZE') else None
    line = '%-46s launches=%d' % (k, len(v.get('FETCH_SIZE', [])))
    if fb is not None: line += ' FETCH_SIZE x2 = %.3f GB' % (fb / 1e9)
    if wb is not None: line += ' WRITE_SIZE = %.3f GB' % (wb / 1e9)
    if v.get('MfmaFlopsF64') and max(v['MfmaFlopsF64']) > 0:
        line += ' MfmaUtil=%.1f%% MFMA_flops/launch=%.4e (%.1f TFLOP/s executed under counters)' % (
            mean(v.get('MfmaUtil', [float("nan")])), mean(v['MfmaFlopsF64']), mean(v['MfmaFlopsF64']) / (mean(v['dur_ns']) * 1e-9) * 1e-12)
    if v.get('VALUBusy'): line += ' VALUBusy=%.1f%% VALUUtilization=%.1f%%' % (mean(v['VALUBusy']), mean(v.get('VALUUtilization', [float("nan")])))
    lines.append(line)
    if k in phase:
        p = phase[k]
        # (a phase that runs two kernels per step -- the near field: wave jobs + workgroup jobs -- adds their bytes and
        # keeps the counters of the one that moves more)
        if fb is not None and wb is not None:
            if p in per_bytes and per_bytes[p] > fb + wb:
                per_bytes[p] += fb + wb
                continue
            per_bytes[p] = per_bytes.get(p, 0.0) + fb + wb
        per_kernel[p] = {'mfma_util_pct': mean(v.get('MfmaUtil')) if v.get('MfmaFlopsF64') and max(v['MfmaFlopsF64']) > 0 else None,
                         'executed_fp64_mfma_flops_per_launch': mean(v.get('MfmaFlopsF64')),
                         'valu_busy_pct': mean(v.get('VALUBusy')), 'valu_utilization_pct': mean(v.get('VALUUtilization'))}
open(root + '/%s_counters.txt' % tag, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(l for l in lines if any(s in l for s in ('m2l_gemm', 'p2p', 'p2m', 'l2p', 'wx_sym'))))
json.dump({'source_hash': bench.source_hash(), 'workload': [10000000, 'LinearRbf', 7, 1], 'per_launch_bytes': per_bytes,
           'per_kernel': per_kernel,
           'note': 'rocprofv3 --kernel-trace --pmc <one counter or derived metric per pass>; bytes = 2*FETCH_SIZE_KiB*1024 + '
                   'WRITE_SIZE_KiB*1024 per launch (L2 memory-side requests; Infinity Cache hits are counted); produced by '
                   'scripts/gpu_round_profiles.sh'}, open(root + '/%s_counters.json' % tag, 'w'), indent=1)
PY
