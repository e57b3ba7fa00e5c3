#!/bin/bash
# Counters of the smoothing-spline kernels (counter passes only): bash tools/probes/smoothing_pmc.sh [planes]
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_smoothing
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=${1:-64}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/tools/probes/smoothing_rate.py $P 1024 randn 1.0 3 1 > $OUT/sq1.log 2>&1 || echo sq1 failed
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tools/probes/smoothing_rate.py $P 1024 randn 1.0 3 1 > $OUT/sq2.log 2>&1 || echo sq2 failed
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ('sq1', 'sq2'):
    for f in glob.glob(f'gpurun_out/pmc_smoothing/{d}/*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'k_smb' not in r['Kernel_Name']: continue
            key = r['Kernel_Name'][:44]
            acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        for key, c in acc.items():
            w = c.get('SQ_WAVES', 0)
            print(d, f'{key:44s}', {k: f'{v:.3g}' for k, v in c.items()})
PY
