"""developer tool: repeat the split-fp16 LBS forward and count deviations from the exact-fp32 kernel"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
from mhhip import synthetic, engine, _lib
st = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, st)
m = engine.BodyModel(st, regs)
L = _lib.lib()
rng = np.random.RandomState(0)
B, NB = 800, 4
betas = torch.tensor(rng.normal(0, 0.7, (NB, 10)).astype(np.float32)).cuda()
poses = torch.tensor(rng.normal(0, 0.3, (B, 72)).astype(np.float32)).cuda()
xs = torch.tensor(rng.normal(0, 1, (NB,)).astype(np.float32)).cuda()
tr = torch.tensor(rng.normal(0, 2, (B, 3)).astype(np.float32)).cuda()
L.mh_lbs_set_mode(0)
v0, q0, _, ws = m.lbs_forward(betas, poses, xs, tr)
torch.cuda.synchronize()
L.mh_lbs_set_mode(1)
bad = {}
N = int(os.environ.get('N', '200'))
for it in range(N):
    v, q, _, _ = m.lbs_forward(betas, poses, xs, tr, ws=ws)
    d = (v - v0).abs().amax(dim=2)
    idx = torch.nonzero(d > 1e-4).cpu().numpy()
    for b, vv in idx:
        key = (int(b) % 32, int(vv) % 32)
        bad[key] = bad.get(key, 0) + 1
    dq = float((q - q0).abs().max())
    if dq > 1e-5:
        print('vposed differs', dq)
print('dbg', os.environ.get('MHHIP_DBG'), 'iterations', N, 'bad (row, lane&31) -> count:', sorted(bad.items())[:40], 'total', sum(bad.values()))
