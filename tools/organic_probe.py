"""developer tool: is the ORGANIC cycle self-consistent?  The C3 cycle with the device-side scene update running beside it
(its kernels share the CUs with the LBS kernels' matrix instructions) is replayed with a zero learning rate: the scene cloud
(count + checksum of the compacted points), the rasteriser's per-body key checksums and the deterministic gradients of every
repetition are held against the first one's."""
import ctypes, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq, _lib
from mhhip.raster import set_deterministic
T = 200
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False))
e = opt.engine
W, H = bench.IMG
opt.scene_depth = bench.ground_scene(K, W, H)
opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)
r = e.raster_terms()
e.update_filters()
e.scene_device_setup(seq['backmasks'])
set_deterministic(True)
B = e.B
off = (ctypes.c_size_t * 3)()
_lib.check(_lib.lib().mh_raster_workspace_offsets(*r.dims, off))
win = r.ws[off[0]:off[0] + B * 16].view(torch.int32).view(B, 4)
raw = r.ws[off[2]:off[2] + B * H * W * 40].view(torch.int64).view(B, H * W, 5)
idx = torch.arange(H * W, device=e.dev)[None, :]

def snap():
    torch.cuda.synchronize()
    npx = (win[:, 2].clamp(min=0) * win[:, 3].clamp(min=0)).to(torch.int64)
    ks = ((raw.sum(dim=2)) * (idx < npx[:, None])).sum(dim=1)
    d = e._scene_dev
    s = d['front'] if d.get('front') is not None else d['sets'][0]
    n = int(s['count'].item())
    P_ = s['pts'][:n].clone()
    pts = P_.contiguous().view(torch.int32).to(torch.int64).sum()
    g = e.grads.clone()
    return ks.clone(), (n, int(pts)), g, P_

for c in range(4):                       # graphs captured, both scene sets filled
    e.cycle_graphed(c, raster=r, scene_update=True); e.scene_device_swap(); e.step(0.0)
ref = snap()
bad = {'keys': 0, 'scene': 0, 'grads': 0}
worst, npts = 0.0, 0
for rep in range(reps):
    e.grads.zero_()
    e.cycle_graphed(4 + rep % 4, raster=r, scene_update=True); e.scene_device_swap(); e.step(0.0)
    cur = snap()
    if not torch.equal(cur[0], ref[0]):
        bad['keys'] += 1; print('repetition', rep, 'bodies with other keys', (cur[0] != ref[0]).nonzero().view(-1).tolist()[:8], flush=True)
    if cur[1] != ref[1]:
        bad['scene'] += 1
        if cur[3].shape == ref[3].shape:
            dd = (cur[3] - ref[3]).abs()
            worst = max(worst, float(dd.max())); npts = max(npts, int((dd > 0).any(dim=1).sum()))
        if bad['scene'] <= 3: print('repetition', rep, 'scene cloud', cur[1], 'first', ref[1], flush=True)
    if not torch.equal(cur[2], ref[2]):
        bad['grads'] += 1
        if bad['grads'] <= 3: print('repetition', rep, 'gradient entries that differ', int((cur[2] != ref[2]).sum()), 'largest', float((cur[2] - ref[2]).abs().max()), flush=True)
print('organic cycle,', reps, 'repetitions: repetitions with other keys %d, other scene cloud %d (at most %d points differ, by at most %.3g m), other gradients %d' % (bad['keys'], bad['scene'], npts, worst, bad['grads']))
