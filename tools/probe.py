#!/usr/bin/env python3
"""
One entry point for the measurement probes of this repo (GPU box): `python tools/probe.py` lists them, `python tools/probe.py
<name> [args ...]` runs tools/probes/<name>.py (or .sh) with the arguments it documents, `python tools/probe.py hip <name>`
builds and runs the stand-alone HIP microbenchmark tools/probes/hip/probe_<name>.hip. The probes are what the numbers of
profiles/ and profiles/EXPERIMENTS*.md came from; none of them is part of the product or of the test-suite.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PROBES = os.path.join(HERE, 'probes')


def first_doc_line(path: str) -> str:
    try:
        text = open(path, encoding='utf-8').read(2000)
    except OSError:
        return ''
    for quote in ('"""', "'''"):
        if quote in text:
            return text.split(quote)[1].strip().splitlines()[0][:110]
    for ln in text.splitlines():
        if ln.startswith(('#', '//')) and not ln.startswith('#!'):
            return ln.lstrip('#/ ').strip()[:110]
    return ''


def main() -> int:
    names = sorted(f for f in os.listdir(PROBES) if f.endswith(('.py', '.sh')))
    if len(sys.argv) < 2 or sys.argv[1] in ('-h', '--help', 'list'):
        print(__doc__)
        for f in names:
            print(f'  {os.path.splitext(f)[0]:<24} {first_doc_line(os.path.join(PROBES, f))}')
        print('  hip <name>               one of: ' + ', '.join(sorted(f[6:-4] for f in os.listdir(os.path.join(PROBES, 'hip')) if f.endswith('.hip'))))
        return 0
    name, args = sys.argv[1], sys.argv[2:]
    if name == 'hip':
        if not args:
            raise SystemExit('python tools/probe.py hip <name> [args]')
        src = os.path.join(PROBES, 'hip', f'probe_{args[0]}.hip')
        exe = os.path.join('/tmp', f'probe_{args[0]}')
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-o', exe, src, '-lpthread'], check=True)
        return subprocess.run([exe] + args[1:]).returncode
    for ext, runner in (('.py', [sys.executable]), ('.sh', ['bash'])):
        path = os.path.join(PROBES, name + ext)
        if os.path.exists(path):
            return subprocess.run(runner + [path] + args).returncode
    raise SystemExit(f'no probe named {name!r}: python tools/probe.py list')


if __name__ == '__main__':
    sys.exit(main())
