"""developer tool: selection (against the brute-force selection) and gradients (against the float64 oracle) of the rasterised
terms for extreme close-ups: bodies 0.35-1.2 m from the camera at 60-120 degrees field of view -- faces of tens of pixels,
windows clipped by the image on every side (30 scenes: no real selection difference, gradients within 2.4e-4)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests')]
from mhhip import synthetic
import test_raster_gpu as tr
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
rng = np.random.RandomState(77)
bad = 0; worst = 0
for c in range(30):
    W, H = [(96, 54), (64, 96), (240, 135), (160, 90)][rng.randint(4)]
    T, N = 1, int(rng.randint(2, 4))
    zlo = float(rng.choice([0.35, 0.5, 0.8])); zhi = zlo + float(rng.choice([0.1, 0.4]))
    fov = float(rng.choice([60.0, 90.0, 120.0]))
    seed = int(rng.randint(1 << 30))
    r = tr._run_case(struct, regs, T, N, W, H, seed, zlo=zlo, zhi=zhi, fov=fov)
    nd, live, nt = tr._selection_differences(r)
    r2 = tr._run_case(struct, regs, T, N, W, H, seed, zlo=zlo, zhi=zhi, fov=fov, hip_selection=True, oracle_dtype=torch.float64)
    g, w = r2['gv'].astype(np.float64), r2['want_gv']; sc = max(np.abs(w).max(), 1e-30); err = np.abs(g - w).max() / sc
    worst = max(worst, err if np.abs(w).max() > 0 else 0)
    print('case %2d %3dx%-3d N%d z %.2f-%.2f fov %3.0f: %4d of %6d live pixels differ, not ties %d; grads max %.1e finite %s' % (c, W, H, N, zlo, zhi, fov, nd, live, len(nt), err, bool(np.isfinite(g).all())), flush=True)
    bad += 1 if nt or not np.isfinite(g).all() else 0
print('bad', bad, 'worst grad', worst)
