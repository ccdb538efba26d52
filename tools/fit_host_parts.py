"""developer tool: host wall time of the parts of fit() that are not cycles (each bracketed by synchronize)"""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq
T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
acc = {}
def timed(obj, name, label=None):
    orig = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); acc.setdefault(label or name, []).append((time.perf_counter() - t) * 1e3)
        return r
    setattr(obj, name, w)
for rep in range(2):
    acc.clear()
    opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
    opt.scene_update = 'device'
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    opt._stage_from_dataloader(dl)
    e, sh = opt.engine, opt.sh
    timed(sh, 'scene_setup'); timed(e, 'enable_filter_gate'); timed(e, 'update_filters'); timed(e, 'scene_device_result')
    timed(sh, 'scene_image'); timed(opt, '_global_leaves'); timed(sh, 'read_log'); timed(opt, '_finish_scene')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.fit(dl, num_iter=250)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('fit %d: wall %.1f ms (with the synchronisations of this tool)' % (rep, (t1 - t0) * 1e3))
    for k, v in acc.items():
        print('   %-22s n=%d  total %.2f ms  (%s)' % (k, len(v), sum(v), ', '.join('%.2f' % x for x in v[:10])))
