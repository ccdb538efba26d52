"""developer tool: the selection kernel's pair counters on the C3 bench sequence (with experiment builds: whatever the
build counts in their place)"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq, _lib
from mhhip.raster import RasterTerms
T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False))
e = opt.engine
L = _lib.lib()
r = RasterTerms(e)
e.cycle(0, raster=r)
torch.cuda.synchronize()
L.mh_profile_enable(2)
gv = torch.zeros_like(e.verts); log = torch.zeros(16, device=e.dev)
a0 = r.pair_counters(e)
for _ in range(4): r(e, gv, log, phases=1)
torch.cuda.synchronize()
a1 = r.pair_counters(e)
n = a1[0] - a0[0]
print('launches', n, 'counter A per launch %.0f' % ((a1[1] - a0[1]) / n), 'counter B per launch %.0f' % ((a1[2] - a0[2]) / n))
