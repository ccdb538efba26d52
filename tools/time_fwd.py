"""developer tool: time the LBS forward alone (split mode) on 800 bodies, with / without the vposed store"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
from mhhip import synthetic, engine, _lib
from mhhip._lib import ptr, check
st = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, st)
m = engine.BodyModel(st, regs)
L = _lib.lib()
rng = np.random.RandomState(0)
B, NB = int(os.environ.get('B', '800')), 4
betas = torch.tensor(rng.normal(0, 0.7, (NB, 10)).astype(np.float32)).cuda()
poses = torch.tensor(rng.normal(0, 0.3, (B, 72)).astype(np.float32)).cuda()
xs = torch.tensor(rng.normal(0, 1, (NB,)).astype(np.float32)).cuda()
tr = torch.tensor(rng.normal(0, 2, (B, 3)).astype(np.float32)).cuda()
verts = torch.empty(B, m.V, 3, device='cuda'); vposed = torch.empty_like(verts)
ws = m.workspace(B)
stp = _lib.stream_ptr(m.device)
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, vq in (('verts+vposed', vposed), ('verts only', None)):
    fn = lambda: check(L.mh_lbs_forward(m.handle, B, NB, ptr(betas), ptr(poses), ptr(xs), ptr(tr), ptr(verts), ptr(vq), None, ptr(ws), stp))
    print('abl', os.environ.get('MHHIP_ABL'), name, '%.1f us' % timeit(fn))
