"""developer tool: run-to-run determinism of debug payloads written to vposed by the split-fp16 LBS forward"""
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
L.mh_lbs_set_mode(1)
ws = m.workspace(B)
runs = []
for it in range(60):
    v, q, _, _ = m.lbs_forward(betas, poses, xs, tr, ws=ws)
    runs.append(q)
torch.cuda.synchronize()
Q = torch.stack(runs)                      # (it, B, V, 3)
med = Q.median(dim=0).values
for c in range(3):
    d = (Q[..., c] - med[..., c]).abs() > 1e-6
    print('dbg', os.environ.get('MHHIP_DBG'), 'component', c, 'deviating entries', int(d.sum()),
          'rows', sorted(set((torch.nonzero(d)[:, 1] % 32).cpu().numpy().tolist()))[:10],
          'lanes', sorted(set((torch.nonzero(d)[:, 2] % 32).cpu().numpy().tolist()))[:40])
