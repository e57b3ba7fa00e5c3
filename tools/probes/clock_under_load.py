import sys, time, json, subprocess, threading
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario
g = load_scenario('jupiter_hst_2005'); sz = 1024; P = int(sys.argv[1])
e = Engine(0); e.set_geometry(g); x0 = (sz - 1) / 2; e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
cube = torch.randn((P, sz, sz), device='cuda', dtype=torch.float64)
lon = np.arange(0.5, 360, 1.0)[::-1]; lat = np.arange(-89.5, 90, 1.0)
lon_g, lat_g = np.meshgrid(lon, lat); n0, n1 = lon_g.shape
lon_d = torch.from_numpy(np.ascontiguousarray(lon_g)).cuda(); lat_d = torch.from_numpy(np.ascontiguousarray(lat_g)).cuda()
xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda'); ym = torch.empty_like(xm)
e.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
out = torch.empty((P, n0, n1), dtype=torch.float64, device='cuda')
def smi():
    time.sleep(1.0)
    for _ in range(2):
        r = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True)
        print('\n'.join(l for l in r.stdout.splitlines() if 'sclk' in l or 'mclk' in l or 'fclk' in l), flush=True)
        time.sleep(0.7)
t = threading.Thread(target=smi); t.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.5:
    for _ in range(50): e.map_cube_device(cube, np.float64, P, xm, ym, n0, n1, out, 'cubic', True)
    e.synchronize(); n += 50
print(json.dumps({'planes': P, 'ms_per_call': round((time.perf_counter() - t0) / n * 1e3, 3)}))
t.join(); e.close()
