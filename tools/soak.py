"""developer tool: soak run -- 20 000 cycles of the C3 bench loop, then 12 complete fits (250 cycles each, organic scene,
filters) on one optimiser and on fresh ones: no NaN, no hang, device memory back to where it started"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
T = 200
def make():
    opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
    opt.scene_update = 'device'
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    dl = torch.utils.data.DataLoader(synthetic_seq.ShardDataset(seq), batch_size=10, shuffle=True)
    return opt, dl
opt, dl = make()
t0 = time.time()
log = opt.fit(dl, num_iter=int(os.environ.get('LONG', '5000')))
torch.cuda.synchronize()
print('one fit of %d cycles: %.1f s, all log entries finite: %s, last loss_pose24j %.4g' % (len(log), time.time() - t0,
      all(np.isfinite(float(v)) for l in log for v in l.values()), float(log[-1]['loss_pose24j'])))
assert torch.isfinite(opt.engine.params).all()
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
for i in range(6):
    log = opt.fit(dl, num_iter=250)
torch.cuda.synchronize(); m1 = torch.cuda.memory_allocated()
print('6 more fits on the same optimiser: device memory %+d bytes, finite %s' % (m1 - m0, bool(torch.isfinite(opt.engine.params).all())))
del opt, dl, log
import gc; gc.collect(); torch.cuda.empty_cache()
base = torch.cuda.memory_allocated()
for i in range(6):
    o2, d2 = make(); o2.fit(d2, num_iter=250); torch.cuda.synchronize()
    del o2, d2; gc.collect()
torch.cuda.empty_cache()
print('6 fresh optimisers created, fitted and dropped: device memory %+d bytes' % (torch.cuda.memory_allocated() - base))
