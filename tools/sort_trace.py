"""developer tool: face sorts per cycle of a fit(250) at C3 -- on the chain (k_raster_prepare) and deferred (beside the gradient kernel)"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq
T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
opt.scene_update = 'device'
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(dl)
e = opt.engine
r = e.raster_terms()
rows = []
orig = e.cycle_graphed
def wrapped(*a, **k):
    out = orig(*a, **k)
    rows.append(r.sort_counters3(e))
    return out
e.cycle_graphed = wrapped
opt.fit(dl, num_iter=250)
rows = np.array(rows, dtype=np.int64)
d = np.diff(np.vstack([[0, 0, 0], rows]), axis=0)
chain = d[:, 1] - d[:, 2]
print('cycle: chain sorts / deferred sorts (of %d bodies); note: the counters of cycle c are written by the backward of cycle c' % e.B)
for c0 in range(0, 250, 10):
    print('%3d-%3d  chain %s   deferred %s' % (c0, c0 + 9, ' '.join('%3d' % x for x in chain[c0:c0 + 10]), ' '.join('%3d' % x for x in d[c0:c0 + 10, 2])))
print('cycles with at least one sort on the chain: %d of 250; with none: %d' % (int((chain > 0).sum()), int((chain == 0).sum())))
