#!/usr/bin/env python3
"""
The fresh-seed legs of the four GPU fuzz tests (tests/test_gpu_parity.py) over many seeds in ONE process:
`python tests/soak_fuzz.py --seeds 60 [--first 1] [--only light_time_paths]`. Not collected by pytest. Prints one JSON line per failure
(test, seed, message) and a summary; a failing seed is replayed with `PM_FUZZ_SEED=<seed> pytest -m gpu -k fuzz`.
"""
import argparse
import json
import os
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=40)
    ap.add_argument('--first', type=int, default=int(time.time()) % 100000 * 1000)
    ap.add_argument('--only', default='', help='one of discs_fast, reprojection, geometries, light_time_paths, smoothing')
    args = ap.parse_args()
    import test_gpu_parity as T
    import test_gpu_splines_cube_scale as S
    from oracle import oracle
    from planetmapper_amd import _lib
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    jupiter, saturn = load_scenario('jupiter_hst_2005'), load_scenario('saturn_earth_2005')
    eng = Engine(0)
    failures = []
    knife_edges = []  # smoothing fits the library itself reported as decided by rounding noise in the reference, beyond the bar
    t0 = time.time()
    for k in range(args.seeds):
        seed = args.first + k
        os.environ['PM_FUZZ_SEED'] = str(seed)
        cases = [('discs_fast', lambda: T.test_random_discs_and_frames_fuzz.__wrapped__(eng, oracle, jupiter, saturn, 'fresh_seed')
                  if hasattr(T.test_random_discs_and_frames_fuzz, '__wrapped__') else T.test_random_discs_and_frames_fuzz(eng, oracle, jupiter, saturn, 'fresh_seed')),
                 ('reprojection', lambda: T.test_random_reprojection_fuzz(eng, oracle, jupiter, 'fresh_seed')),
                 ('geometries', lambda: T.test_random_geometries_fuzz(eng, oracle, 'fresh_seed')),
                 ('light_time_paths', lambda: T.test_random_epochs_body_sizes_and_spins_fuzz(eng, oracle, jupiter, saturn, 'fresh_seed')),
                 ('smoothing', lambda: knife_edges.extend(S.smoothing_fuzz(eng, oracle, jupiter, seed)))]  # fmt: skip
        if args.only:
            cases = [c for c in cases if c[0] == args.only]
        for general in (0, 1):
            eng.set_option(_lib.PM_OPT_GENERAL_KERNEL, general)
            # (general kernel forced: the frames sweep and the geometries sweep - near field, triaxial, fast approach)
            # (the epochs / sizes / spins sweep is about the library's OWN choice of kernel: not run forced)
            for name, fn in (cases if general == 0 else [c for c in cases if c[0] in ('discs_fast', 'geometries')]):
                try:
                    fn()
                except Exception as e:  # noqa: BLE001
                    failures.append({'test': name, 'general': general, 'seed': seed, 'error': str(e)[:400],
                                     'where': traceback.format_exc().strip().splitlines()[-3][:200]})
                    print(json.dumps(failures[-1]), flush=True)
        if k % 10 == 9:
            print(json.dumps({'done': k + 1, 'failures': len(failures), 'seconds': round(time.time() - t0, 1)}), flush=True)
    eng.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
    eng.close()
    for ke in knife_edges:
        print(json.dumps({'knife_edge': [str(v) for v in ke]}), flush=True)
    print(json.dumps({'seeds': args.seeds, 'first': args.first, 'failures': len(failures), 'reported_knife_edges_beyond_bar': len(knife_edges),
                      'smoothing_plane_fits': getattr(S.smoothing_fuzz, 'calls', 0), 'of_them_reported_as_knife_edges': getattr(S.smoothing_fuzz, 'flagged', 0),
                      'of_them_reported_as_ill_conditioned': getattr(S.smoothing_fuzz, 'ill', 0),
                      'seconds': round(time.time() - t0, 1)}), flush=True)
    return 1 if failures else 0


if __name__ == '__main__':
    sys.exit(main())
