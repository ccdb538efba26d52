import os, sys, pathlib, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_deterministic
from oracle import lbs_oracle
import test_full_size_gpu as tf
st = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, st)
om = lbs_oracle.BodyModel(st, regs)
T, N, W, H, batch = 200, 4, 240, 135, 10
opt, dl, o, batches, seq = tf._setup(st, regs, om, pathlib.Path(tempfile.mkdtemp()), T, N, W, H, batch, 41, True)
opt._stage_from_dataloader(dl)
e = opt.engine
raster = RasterTerms(e)
hsel = tf._HipSelectionRasteriser(np.asarray(st.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N)
o.rasteriser = hsel
set_deterministic(True)
e.cycle(0, raster=raster)
hsel.take(raster, e)
want = o.cycle_grads(batches)
w = tf._oracle_grad(o, 'poses_T')
g = e.leaf('poses_T', e.grads).cpu().numpy().reshape(w.shape)
scale = np.abs(w).max()
bad = np.argwhere(np.abs(g - w) > 1e-3 * scale)
print('scale', scale, bad)
for t, n in sorted(set((int(b[0]), int(b[1])) for b in bad)):
    print('frame', t, 'person', n, 'hip', g[t, n], 'oracle', w[t, n])
    pT = e.leaf('poses_T').cpu().numpy()[t]
    print('  poses_T z of the frame', pT[:, 2], 'front', e.front.cpu().numpy().reshape(T, N)[t], 'sil_apply', e.sil_apply.cpu().numpy().reshape(T, N)[t],
          'p2d_valid', e.p2d_valid.cpu().numpy().reshape(T, N)[t], 'mask_valid', e.mask_valid.cpu().numpy().reshape(T, N)[t])
    print('  dy', e.dy.cpu().numpy().reshape(T, N)[t], 'low_idx', e.low_idx.cpu().numpy().reshape(T, N)[max(t-1,0):t+2])
    print('  depth_body', e.depth_body.cpu().numpy().reshape(T, N)[t], 'sil_body', e.sil_body.cpu().numpy().reshape(T, N)[t])
    # oracle single-batch terms of that frame
    b = t // batch
    total, terms, _ = o.batch_loss(batches[b])
    print('  oracle batch terms', {k: float(v) for k, v in terms.items()})
    gvo = None

from oracle import fit_oracle as fo, raster_oracle as ro
keys = ['proj2d', 'depth', 'silhouette', 'reg_contact', 'reg_foot_sliding', 'reg_velocity', 'reg_poses', 'reg_scales', 'reg_verts_filter']
for kk in keys:
    e.c[kk] = e.c[kk] if kk == 'depth' else 0.0
e._graphs = {}
e.cycle(0, raster=raster)
t = 105
gv = e.gverts.view(T, N, -1, 3)[t].cpu().numpy()
verts = e.verts.view(T, N, -1, 3)[t].cpu().clone().requires_grad_(True)
sel = hsel.sel[t]
K = synthetic.default_cam_K((W, H), 60.0)
zbuf, alpha = ro.render(verts, hsel.faces, K, (W, H), selection=(sel[..., :1], sel[..., 1:]))
seg = torch.tensor(seq['seg_mask'][t])
conf = (torch.tensor(seq['pose2d'][t][..., 2:3]) >= 0.5).float()
p2d_valid = (conf.sum(dim=(1, 2)) >= 2).float()
zmin, zmax = e.leaf('zmin_lin')[t].cpu(), e.leaf('zmax_lin')[t].cpu()
min_z = fo.softplus(zmin); max_z = min_z + 1.0 + fo.softplus(zmax)
tgt = torch.tensor(seq['depths'][t]) * (1.0 / min_z - 1.0 / max_z) + 1.0 / max_z
er = fo.erode3x3(fo.erode3x3(seg[None]))[0]
m = (zbuf > 0).float() * er * p2d_valid[:, None, None]
pred = 1.0 / torch.clamp(zbuf + 0.2, min=1e-3)
lp = m * torch.log(torch.clamp(pred, min=1e-3)); lt = m * torch.log(torch.clamp(tgt[None], min=1e-3))
cnt = m.sum(dim=(1, 2)) + 1
dep = (lp.sum(dim=(1, 2)) / cnt - lt.sum(dim=(1, 2)) / cnt) ** 2
print('oracle depth per person', dep.detach().numpy(), 'hip', e.depth_body.view(T, N)[t].cpu().numpy(), 'cnt', cnt.numpy())
(0.05 * dep.sum()).backward()
wv = verts.grad.numpy()
for n in range(N):
    d = np.abs(gv[n] - wv[n]).max(axis=1)
    print('person', n, 'sum grad hip', gv[n].sum(0), 'oracle', wv[n].sum(0), 'max vertex err', d.max(), 'at', d.argmax(), 'n>1e-5:', int((d > 1e-5).sum()), 'max |w|', np.abs(wv[n]).max())
    for v in np.argsort(-d)[:5]:
        print('    v', v, 'hip', gv[n][v], 'oracle', wv[n][v])
