#!/bin/bash
# On the GPU box: the judged evidence of the workloads named on the command line (frame saturn all26 maps cube), each
#   1. bash tools/pmc_profile.sh <round>_<w> [--workload <w>]   (kernel trace + four counter passes of bench.py)
#   2. the stamped traffic file put where bench.py looks for it, then the default bench line un-traced
# Results under gpurun_out/; `python tools/collect_evidence.py <round> <workloads>` copies them into profiles/ afterwards.
# Usage (two calls fit the 20-minute limit): gpurun -- bash tools/refresh_evidence.sh r05 frame saturn all26
ROUND=$1; shift
for w in "$@"; do
  if [ "$w" = frame ]; then args=""; tf=profiles/traffic.json; else args="--workload $w"; tf=profiles/traffic_$w.json; fi
  bash tools/pmc_profile.sh ${ROUND}_$w $args > gpurun_out/${ROUND}_${w}_pmc.log 2>&1 || { echo "pmc_profile $w failed"; exit 1; }
  cp gpurun_out/pmc_${ROUND}_$w/traffic.json $tf
  python3 bench.py $args > gpurun_out/${ROUND}_${w}_bench.log 2>&1 || { echo "bench $w failed"; exit 1; }
  grep '^{' gpurun_out/${ROUND}_${w}_bench.log | tail -n 1 > gpurun_out/${ROUND}_${w}_bench.json
  python3 - <<PY
import json
d = json.load(open('gpurun_out/${ROUND}_${w}_bench.json'))
r = d.get('roofline', {})
print('$w', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'sha', str(r.get('library_sha256'))[:12])
PY
done
