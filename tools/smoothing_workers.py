#!/usr/bin/env python3
"""Smoothing-spline map_img of a cube vs the number of fit workers (PM_SM_WORKERS), GPU box."""
import json, os, subprocess, sys

for w in (1, 2, 4, 8):
    env = dict(os.environ, PM_DEBUG_ENV='1', PM_SM_WORKERS=str(w))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'smoothing_timing.py'), '1024', '16'],
                         env=env, capture_output=True, text=True).stdout
    rows = [json.loads(l) for l in out.splitlines() if 'spline_smoothing' in l]
    print(json.dumps({'workers': w, 'ms_per_plane': {r['interpolation']: r['host_call_ms_per_plane'] for r in rows}}))
