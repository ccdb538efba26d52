"""developer tool: the device-side scene update of one cycle (optimizer.py:578-584) alone on the GPU: time per update"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq
T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.ShardDataset(seq), batch_size=10, shuffle=False))
e, sh = opt.engine, opt.sh
W, H = bench.IMG
opt.scene_depth = bench.ground_scene(K, W, H)
opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)
if e._scene_dev is None:
    sh.scene_setup(torch.as_tensor(seq['backmasks'])) if hasattr(sh, 'scene_setup') else e.scene_device_setup(seq['backmasks'])
def upd():
    e.scene_device_update(); e.scene_device_swap()
for _ in range(5): upd()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n): upd()
torch.cuda.synchronize()
print('scene update alone: %.1f us per update' % ((time.perf_counter() - t0) / n * 1e6))
