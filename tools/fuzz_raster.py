"""developer tool: the selection kernel against the oracle's brute-force selection over random scenes (shapes, depths,
fields of view): every pixel whose faces differ must be a float64 near-tie (tests/test_raster_gpu.py's criterion)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests')]
from mhhip import synthetic
import test_raster_gpu as tr
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
rng = np.random.RandomState(int(os.environ.get('SEED', '7')))
n_cases = int(os.environ.get('CASES', '16'))
bad = 0
for c in range(n_cases):
    W, H = [(96, 54), (64, 96), (80, 80), (160, 90), (48, 135), (240, 135)][rng.randint(6)]
    T, N = int(rng.randint(1, 3)), int(rng.randint(1, 4))
    zlo = float(rng.choice([1.1, 1.6, 2.5, 4.0]))
    zhi = zlo + float(rng.choice([0.3, 1.0, 3.0]))
    fov = float(rng.choice([40.0, 60.0, 90.0]))
    seed = int(rng.randint(1 << 30))
    t0 = time.time()
    r = tr._run_case(struct, regs, T, N, W, H, seed, zlo=zlo, zhi=zhi, fov=fov)
    nd, live, not_ties = tr._selection_differences(r)
    print('case %2d  %3dx%-3d T%d N%d z %.1f-%.1f fov %2.0f : %5d pixels differ of %7d live (%.3f %%), not near-ties: %d   [%.0f s]'
          % (c, W, H, T, N, zlo, zhi, fov, nd, live, 100.0 * nd / max(live, 1), len(not_ties), time.time() - t0), flush=True)
    if not_ties:
        bad += 1
        for q in not_ties[:6]:
            print('    body %d pixel (%d,%d) face %d: pz %.7f inside %s d2 %.4e' % (q[0], q[2], q[1], q[3], q[4], q[5], q[6]))
print('cases with real differences:', bad)
sys.exit(1 if bad else 0)
