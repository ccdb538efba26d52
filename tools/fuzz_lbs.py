"""developer tool: LBS forward / backward of the HIP model against the oracle in float64 over random batches with
EXTREME inputs: shape coefficients up to +-4, joint rotations up to pi about random axes (and exact zeros), scales 1.1^(+-6),
translations of +-20 m; batch sizes around the 32-body group boundaries."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic, engine
from oracle import lbs_oracle as lo
import test_lbs_gpu as tl
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
hip = engine.BodyModel(struct, regs)
m64 = lo.BodyModel(struct, regs, dtype=torch.float64)
rng = np.random.RandomState(int(os.environ.get('SEED', '3')))
wf = wb = 0.0
for c in range(int(os.environ.get('CASES', '12'))):
    B = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100]))
    NB = int(rng.choice([d for d in (1, 2, 3, 4, B) if B % d == 0]))
    amp_b, amp_p = float(rng.choice([0.7, 2.0, 4.0])), float(rng.choice([0.3, 1.0, np.pi]))
    betas = rng.uniform(-amp_b, amp_b, (NB, 10)).astype(np.float32)
    poses = rng.uniform(-1, 1, (B, 24, 3)).astype(np.float32)
    poses *= (amp_p * rng.uniform(0, 1, (B, 24, 1)) / np.maximum(np.linalg.norm(poses, axis=2, keepdims=True), 1e-6)).astype(np.float32)
    poses[rng.uniform(size=(B, 24)) < 0.2] = 0                                  # exact zero rotations
    poses = poses.reshape(B, 72); poses[:, 66:] = 0
    xs = rng.uniform(-6, 6, (NB,)).astype(np.float32)
    tr = rng.uniform(-20, 20, (B, 3)).astype(np.float32)
    wv = rng.normal(0, 1, (B, 6890, 3)).astype(np.float32)
    wj = rng.normal(0, 5, (B, 17, 3)).astype(np.float32)
    d = tl.dev
    verts, vposed, _, ws = hip.lbs_forward(d(betas), d(poses), d(xs), d(tr))
    bidx = np.arange(B) % NB
    ref = lo.smpl_forward(m64, torch.tensor(betas[bidx]).double(), torch.tensor(poses).double())
    s = torch.pow(torch.tensor(1.1, dtype=torch.float64), torch.tensor(xs[bidx]).double())[:, None, None]
    want = (s * ref['verts'] + torch.tensor(tr).double()[:, None]).numpy()
    ef = float(np.abs(verts.cpu().numpy() - want).max())
    size = float(np.abs(want - tr[:, None]).max())                               # extent of the scaled body
    got = hip.lbs_backward(d(betas), d(poses), d(xs), d(tr), vposed, d(wv), d(wj), ws)
    torch.cuda.synchronize()
    wantg = tl._oracle_grads(m64, betas, poses, xs, tr, wv, wj, NB, torch.float64)
    eb = {}
    for name, g, w in zip(['poses', 'transl', 'betas', 'xscale'], got, wantg):
        g = g.cpu().numpy().reshape(w.shape)
        eb[name] = float(np.abs(g - w).max() / max(np.abs(w).max(), 1e-30))
    wf, wb = max(wf, ef / max(size, 1e-9)), max(wb, max(eb.values()))
    print('case %2d B %3d NB %3d |betas| %.1f |rot| %.1f: verts %.2e m (body extent %.1f m: %.1e rel)  grads %s'
          % (c, B, NB, amp_b, amp_p, ef, size, ef / size, {k: '%.1e' % v for k, v in eb.items()}), flush=True)
print('worst forward error relative to the body extent: %.2e   worst gradient entry relative to its leaf\'s largest: %.2e' % (wf, wb))
