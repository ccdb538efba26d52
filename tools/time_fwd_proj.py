"""developer tool: the LBS forward of the C3 sequence (800 bodies) alone -- plain (verts + v_posed) against the
"LBS + projection" form (+ NDC vertices, screen boxes, motion flags, lowest vertices) -- mean time per call
(k_pose_fwd + k_skin_fwd16), after a few full cycles so that the report filters hold the previous launch's extremes.
MHHIP_LIB=... selects a variant build (tools/ab_fwd_proj.sh)."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), bench.T_LOCAL, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, bench.N_PEOPLE, bench.T_LOCAL, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
dl = torch.utils.data.DataLoader(synthetic_seq.ShardDataset(seq), batch_size=bench.BATCH, shuffle=False)
opt._stage_from_dataloader(dl)
e = opt.engine
raster = e.raster_terms()
for c in range(5):
    e.cycle(c, raster=raster); e.step(0.001)
torch.cuda.synchronize()
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tp = timeit(lambda: e.forward(regress=False))
tq = timeit(lambda: e.forward(regress=False, raster=raster))
print('%-40s plain %.1f us   projection form %.1f us' % (os.path.basename(os.environ.get('MHHIP_LIB', 'libmhmocap_hip.so')), tp, tq))
