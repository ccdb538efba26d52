"""developer tool: gradients and loss values of the rasterised terms against the oracle (autograd through the oracle's
renderer fed with the HIP selection) over random scenes.  ORACLE=f64 (default): the oracle in float64 -- on the faces of a
fraction of a pixel the float32 oracle is itself up to 1e-3 (of the largest entry) off its float64 self; ORACLE=f32."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests')]
from mhhip import synthetic
import test_raster_gpu as tr
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
rng = np.random.RandomState(int(os.environ.get('SEED', '7')))
worst = 0.0
for c in range(int(os.environ.get('CASES', '16'))):
    W, H = [(96, 54), (64, 96), (80, 80), (160, 90), (48, 135), (240, 135)][rng.randint(6)]
    T, N = int(rng.randint(1, 3)), int(rng.randint(1, 4))
    zlo = float(rng.choice([1.1, 1.6, 2.5, 4.0]))
    zhi = zlo + float(rng.choice([0.3, 1.0, 3.0]))
    fov = float(rng.choice([40.0, 60.0, 90.0]))
    seed = int(rng.randint(1 << 30))
    mode = os.environ.get('ORACLE', 'f64')
    r = tr._run_case(struct, regs, T, N, W, H, seed, zlo=zlo, zhi=zhi, fov=fov, hip_selection=True,
                     oracle_dtype=torch.float32 if mode == 'f32' else torch.float64)
    g, w = r['gv'], r['want_gv']
    scale = max(np.abs(w).max(), 1e-12)
    err = np.abs(g - w) / scale
    if mode == 'both':        # an entry is right when it agrees with the oracle at either precision
        r32 = tr._run_case(struct, regs, T, N, W, H, seed, zlo=zlo, zhi=zhi, fov=fov, hip_selection=True, oracle_dtype=torch.float32)
        err = np.minimum(err, np.abs(g - r32['want_gv']) / scale)
    dv = np.abs(r['depth'] - r['want_depth']).max() / max(np.abs(r['want_depth']).max(), 1e-12)
    sv = np.abs(r['sil'] - r['want_sil']).max() / max(np.abs(r['want_sil']).max(), 1e-12)
    gz = max(np.abs(r['gzmin'] - r['want_gzmin']).max() / max(np.abs(r['want_gzmin']).max(), 1e-12),
             np.abs(r['gzmax'] - r['want_gzmax']).max() / max(np.abs(r['want_gzmax']).max(), 1e-12))
    worst = max(worst, err.max())
    print('case %2d %3dx%-3d T%d N%d z %.1f-%.1f fov %2.0f: dverts max %.2e (entries > 2e-4: %d of %d)  depth %.1e sil %.1e dz %.1e'
          % (c, W, H, T, N, zlo, zhi, fov, err.max(), int((err > 2e-4).sum()), err.size, dv, sv, gz), flush=True)
print('worst relative dverts error: %.2e' % worst)
