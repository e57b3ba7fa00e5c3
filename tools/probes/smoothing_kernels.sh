#!/bin/bash
# Kernel times of the smoothing-spline search: bash tools/probes/smoothing_kernels.sh [planes [size [data [s_factor]]]]
OUT=$GRAFT_REPO_ROOT/gpurun_out/smoothing_kernels
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probes/smoothing_rate.py ${1:-1} ${2:-1024} ${3:-randn} ${4:-1.0} 3 1 > $OUT/run.log 2>&1 || echo failed
cd $GRAFT_REPO_ROOT
grep '^{' $OUT/run.log
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/smoothing_kernels/*/*kernel_trace.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:64]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = sum(sum(v) for v in acc.values())
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print(f'{k:64s} n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:8.2f} ms  {100*sum(v)/tot:5.1f} %')
PY
