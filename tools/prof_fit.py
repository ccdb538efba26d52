"""developer tool: cProfile of opt.fit(dataloader, 250) on C3 (host-side view: captures, filter updates, scene set-up)"""
import os, sys, tempfile, time, cProfile, pstats
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
tmp = tempfile.mkdtemp()
opt = bench.build_optimizer(struct, regs, tmp, 200, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, 200, bench.IMG, 1003, cam_K=K)
for rep in range(2):
    opt = bench.build_optimizer(struct, regs, tmp, 200, 'cuda:0', K)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
    opt._stage_from_dataloader(dl)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    opt.fit(dl, num_iter=250)
    torch.cuda.synchronize()
    pr.disable()
    print('fit wall %.1f ms' % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
