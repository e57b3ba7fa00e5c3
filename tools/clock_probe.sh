#!/bin/bash
# Shader clock under the frame kernel: GRBM_GUI_ACTIVE (cycles summed over 8 XCDs) / kernel duration, for the
# shipped library and the instrumented builds of tools/exp (PM_EXPERIMENT). Usage: bash tools/clock_probe.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/clock_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ship exp1 exp2; do
  if [ $v = ship ]; then export PLANETMAPPER_HIP_LIB=$GRAFT_REPO_ROOT/planetmapper_amd/libplanetmapper_hip.so; else export PLANETMAPPER_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/lib$v.so; fi
  PREHEAT=300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$v -- python3 $GRAFT_REPO_ROOT/tools/kernel_sweep.py > $OUT/$v.log 2>&1 || echo "pass $v failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'clock_probe')
for v in ('ship', 'exp1', 'exp2'):
    cc = glob.glob(f'{out}/{v}/*/*counter_collection.csv')
    if not cc:
        print(v, 'no counters'); continue
    rows = list(csv.DictReader(open(cc[0])))
    # launches come in 4 groups of (PREHEAT + 100) per disc size: r0 = 1, 0.45, 0.9, 1e5
    ks = [r for r in rows if 'k_disc_sph' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
    n = len(ks) // 4
    for gi, name in enumerate(('store-only', '15%', 'headline', 'all-on-disc')):
        grp = ks[gi * n + n // 2:(gi + 1) * n]
        cyc = [float(r['Counter_Value']) / 8 for r in grp]
        dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in grp] if 'End_Timestamp' in grp[0] else None
        if dur:
            ghz = [c / d for c, d in zip(cyc, dur)]
            print(v, name, 'launches', len(grp), 'cycles', round(sum(cyc) / len(cyc)), 'us', round(sum(dur) / len(dur) / 1e3, 1), 'GHz', round(sum(ghz) / len(ghz), 3))
        else:
            print(v, name, 'cycles', round(sum(cyc) / len(cyc)), list(grp[0].keys()))
PY
