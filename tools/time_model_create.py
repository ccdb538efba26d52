"""developer tool: wall time of mh_model_create (host-side table building + uploads) and of init_optimized_variables"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq, engine
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
torch.zeros(1, device='cuda:0')
for _ in range(3):
    t0 = time.perf_counter()
    m = engine.BodyModel(struct, regs)
    torch.cuda.synchronize()
    print('BodyModel(...) %.1f ms' % ((time.perf_counter() - t0) * 1e3))
K = synthetic.default_cam_K(bench.IMG, 60.0)
tmp = tempfile.mkdtemp()
opt = bench.build_optimizer(struct, regs, tmp, 200, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, 200, bench.IMG, 1003, cam_K=K)
for _ in range(2):
    opt = bench.build_optimizer(struct, regs, tmp, 200, 'cuda:0', K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    torch.cuda.synchronize()
    print('init_optimized_variables(num_iter=100) incl. model construction %.1f ms' % ((time.perf_counter() - t0) * 1e3))
