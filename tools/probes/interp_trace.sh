#!/bin/bash
# Kernel times of the interp_rate probe: bash tools/probes/interp_trace.sh [planes] -> gpurun_out/interp_trace/kernel_stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/interp_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probes/interp_rate.py ${1:-65} 1024 > $OUT/run.log 2>&1 || echo failed
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/interp_trace/*/*kernel_trace.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:70], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:24]:
        print(f'{k[0]:70s} grid {k[1]:>9s} x {k[2]:>5s}  n {len(v):3d}  avg {sum(v)/len(v):9.1f} us')
PY
