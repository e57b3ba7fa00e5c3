#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per pm:: kernel, + traffic JSON."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:60]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
traffic = {}
for k, cs in sorted(acc.items()):
    if not k.startswith(('void pm::', 'pm::')):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}')
    if 'WRITE_SIZE' in cs and 'FETCH_SIZE' in cs:
        # rocprofv3 reports KiB. FETCH_SIZE under-reports wide coalesced streaming reads by 2x on
        # gfx950 (MI355X_MICROARCH.md, HBM); the image kernels read nothing but kernel arguments and
        # the gather kernel reads scattered 8-byte words (uncalibrated pattern), so the raw value is
        # kept and only named here.
        w = sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE']) * 1024
        r = sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE']) * 1024
        name = k.replace('void ', '')
        # ... except the map kernels: they stream their lon / lat grids in (8 B per lane from two arrays, coalesced), a byte
        # count known exactly - 16 B per cell - against which FETCH_SIZE reads one half here too (calibrated: 52.2 MB
        # reported for 103.7 MB at 1800 x 3600 cells): doubled, as the guide prescribes for streaming reads
        k_read = 2.0 if name.startswith(('pm::k_map_b0', 'pm::k_map_xy', 'pm::k_map<')) else 1.0
        traffic[name] = {'write_bytes': w, 'fetch_bytes_raw': r, 'hbm_bytes': w + k_read * r}
        if k_read != 1.0:
            traffic[name]['fetch_correction'] = k_read
    f64 = ['SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_TRANS_F64']
    if all(c in cs for c in f64):
        # wave-instructions per launch -> FP64 operations: 64 lanes, an FMA counts 2
        m = {c: sum(cs[c]) / len(cs[c]) for c in f64}
        name = k.replace('void ', '')
        traffic.setdefault(name, {})['fp64_flop'] = 64.0 * (2 * m[f64[0]] + m[f64[1]] + m[f64[2]] + m[f64[3]])
        traffic[name]['fp64_wave_insts'] = sum(m.values())
# the kernel-trace pass of the same command (pmc_profile.sh: $OUT/stats): the profiler's own average duration per kernel -
# bench.py prints it beside its HIP-event figure (roofline.kernel_ms_rocprof), so a line read on its own says both
for f in glob.glob(os.path.join(root, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Name'].split('(')[0][:60].replace('void ', '')  # (the key of the counter rows above)
        if name.startswith('pm::'):
            traffic.setdefault(name, {})['rocprof_avg_ns'] = float(row['AverageNs'])
            traffic[name]['rocprof_calls'] = int(row['Calls'])
# which build these counters belong to: bench.py reports them only while the library it has loaded is this one
import hashlib

lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'planetmapper_amd', 'libplanetmapper_hip.so')
if os.path.exists(lib):
    traffic['_library_sha256'] = hashlib.sha256(open(lib, 'rb').read()).hexdigest()
with open(os.path.join(root, 'traffic.json'), 'w') as f:
    json.dump(traffic, f, indent=1)
