#!/bin/bash
# Counters of k_reproject_smooth on the interp_rate probe (counter passes only): bash tools/probes/smooth_pmc.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_smooth
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/tools/probes/interp_rate.py 65 1024 > $OUT/sq1.log 2>&1 || echo sq1 failed
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tools/probes/interp_rate.py 65 1024 > $OUT/sq2.log 2>&1 || echo sq2 failed
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ('sq1', 'sq2'):
    for f in glob.glob(f'gpurun_out/pmc_smooth/{d}/*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'smooth' not in r['Kernel_Name']: continue
            key = (r['Kernel_Name'][:40], r['Grid_Size'])
            acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        for key, c in acc.items():
            print(d, key, {k: f'{v:.4g}' for k, v in c.items()})
PY
