import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import set_deterministic
import test_raster_gpu as trg
from oracle import raster_oracle as ro
st = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, st)
set_deterministic(True)
case = dict(T=2, N=3, W=120, H=68, seed=11, zlo=2.2, zhi=4.0)
r = trg._run_case(st, regs, hip_selection=True, **case)
g, w = r['gv'], r['want_gv']
scale = np.abs(w).max()
bad = np.argwhere(np.abs(g - w) > 2e-4 * scale)
print('scale', scale, 'violations', len(bad))
T, N, H, W = r['shape']
got = trg._hip_selection(r['sel'], T * N, H, W)
faces = r['faces']
ndc = ro.to_ndc(torch.tensor(r['verts']), r['K'], (W, H)).numpy().astype(np.float32)
xs, ys = ro.pixel_centres_ndc(H, W)
for b, v, c in bad:
    print('body', b, 'vertex', v, 'comp', c, 'hip', g[b, v, c], 'oracle', w[b, v, c])
    fs = np.nonzero((faces == v).any(axis=1))[0]
    hits = np.argwhere(np.isin(got[b], fs))
    for y, x, k in hits:
        f = got[b, y, x, k]
        pz, inside, d2 = trg._face_eval64(ndc, faces, b, f, float(xs[x]), float(ys[y]))
        print('    pixel', y, x, 'slot', k, 'face', f, faces[f], 'pz %.7f inside %s d2 %.4e' % (pz, inside, d2))
