"""Copy what tools/refresh_evidence.sh left under gpurun_out/ into profiles/: python tools/collect_evidence.py r05 frame saturn ..."""
import glob, os, shutil, sys, json

rnd, workloads = sys.argv[1], sys.argv[2:]
for w in workloads:
    src = f'gpurun_out/pmc_{rnd}_{w}'
    stats = sorted(glob.glob(f'{src}/stats/*/*kernel_stats.csv'), key=os.path.getmtime)  # (earlier passes stay in gpurun_out/)
    assert stats, f'no kernel stats for {w}'
    shutil.copy(stats[-1], f'profiles/{rnd}_{w}_kernel_stats.csv')
    shutil.copy(f'{src}/summary.txt', f'profiles/{rnd}_{w}_pmc_summary.txt')
    shutil.copy(f'{src}/traffic.json', f'profiles/{rnd}_{w}_traffic.json')
    shutil.copy(f'{src}/traffic.json', 'profiles/traffic.json' if w == 'frame' else f'profiles/traffic_{w}.json')
    lines = [l for l in open(f'{src}/bench_stats.log') if l.startswith('{')]
    open(f'profiles/{rnd}_{w}_bench_under_rocprof.json', 'w').write(lines[-1])
    shutil.copy(f'gpurun_out/{rnd}_{w}_bench.json', f'profiles/{rnd}_{w}_bench.json')
    d = json.load(open(f'profiles/{rnd}_{w}_bench.json'))
    t = json.load(open(f'{src}/traffic.json'))
    print(w, d['value'], d['unit'], 'frac', d['roofline'].get('frac'), 'traffic', d['roofline'].get('traffic'), 'sha', t.get('_library_sha256', '')[:12])
