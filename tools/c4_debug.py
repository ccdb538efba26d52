"""developer tool: which run is right when a frame's depth-range gradient differs between the one-process 4 x 2000 run and a
shard of it?  Engine A holds all 2000 frames, engine B frames F0..F1 only; both run one eager cycle (deterministic scatter)
with the frame-local terms only; per-body loss values, windows and depth-range gradients of the shared frames are compared."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq
from mhhip.raster import RasterTerms, set_deterministic
T, F0, F1 = int(os.environ.get('T', 2000)), int(os.environ.get('F0', 1500)), int(os.environ.get('F1', 1750))
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
bench.COEFS.update(reg_velocity=0.0, reg_verts_filter=0.0, reg_contact=0.0, reg_foot_sliding=0.0)
def build(f0, f1, seq_all=None):
    opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), f1 - f0, 'cuda:0', K)
    seq = seq_all
    sub = {k: (v[f0:f1] if isinstance(v, np.ndarray) and v.shape[:1] == (T,) else v) for k, v in seq.items()}
    opt.init_optimized_variables(sub['pose2d'], sub['poses_smpl'], sub['betas_smpl'], sub['valid_smpl'], num_iter=0)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(sub), batch_size=bench.BATCH, shuffle=False)
    opt._stage_from_dataloader(dl)
    return opt
model = __import__('mhhip.engine', fromlist=['x']).BodyModel(struct, regs)
seq = synthetic_seq.make_sequence(model, bench.N_PEOPLE, T, bench.IMG, 1003, cam_K=K)
A, B = build(0, T, seq), build(F0, F1, seq)
rng = np.random.RandomState(0)
pT = np.stack([rng.uniform(-1.5, 1.5, (T, 4)), 0.95 + 0 * rng.rand(T, 4), rng.uniform(3, 7, (T, 4))], -1).astype(np.float32)
for opt, sl in ((A, slice(0, T)), (B, slice(F0, F1))):
    e = opt.engine
    e.leaf('poses_T').copy_(torch.tensor(pT[sl]).cuda())
    e.leaf('zmax_lin').fill_(8.0)
for k in ('betas', 'xscale'):                      # (the shape leaf is a mean over the optimiser's frames: the shard gets A's)
    B.engine.leaf(k).copy_(A.engine.leaf(k))
for k in ('poses_smpl', 'zmin_lin'):
    B.engine.leaf(k).copy_(A.engine.leaf(k)[F0:F1])
B.engine.betas_ref.copy_(A.engine.betas_ref)
old = set_deterministic(True)
out = {}
for name, opt in (('A', A), ('B', B)):
    e = opt.engine
    r = RasterTerms(e)
    e.cycle(0, raster=r)
    torch.cuda.synchronize()
    win, koff, keys = r.selection(e)
    out[name] = dict(gzmin=e.leaf('zmin_lin', e.grads).cpu().numpy(), gzmax=e.leaf('zmax_lin', e.grads).cpu().numpy(),
                     depth=e.depth_body.cpu().numpy(), sil=e.sil_body.cpu().numpy(), win=win, npx=np.diff(koff),
                     gpT=e.leaf('poses_T', e.grads).cpu().numpy())
set_deterministic(old)
a, b = out['A'], out['B']
N = bench.N_PEOPLE
for k in ('gzmin', 'gzmax'):
    d = np.abs(a[k][F0:F1] - b[k])
    print(k, 'max diff %.3e (scale %.3e) at local frame %d' % (d.max(), np.abs(b[k]).max(), d.argmax()))
for k in ('depth', 'sil'):
    d = np.abs(a[k][F0 * N:F1 * N] - b[k])
    print(k, 'max diff %.3e (scale %.3e) at local body %d' % (d.max(), np.abs(b[k]).max(), d.argmax()))
print('windows equal:', (a['win'][F0 * N:F1 * N] == b['win']).all(), ' window pixels equal:', (a['npx'][F0 * N:F1 * N] == b['npx']).all())
d = np.abs(a['gpT'][F0:F1] - b['gpT'])
print('gpT max diff %.3e (scale %.3e)' % (d.max(), np.abs(b['gpT']).max()))
