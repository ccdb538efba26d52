"""developer tool (GPU box): stand-alone time of the "LBS + projection" forward at C3 (k_pose_fwd + the skinning kernel, eager
launches with events around them) for each form of the kernel: mh_lbs_set_forward_pipeline 0 (tile after tile) / 1 (software
pipeline over a wave's tiles) / 2 (producer and consumer waves).  python tools/fwd_probe.py [modes [dbg [tiles per workgroup]]]
(the dbg column drove the timing-only variants of the producer / consumer kernel while they existed: csrc/mh_lbs.hip has
their numbers)"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq, _lib

T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(dl)
e = opt.engine
raster = e.raster_terms()
for c in range(3):
    e.cycle(0, raster=raster); e.step(0.01)
L = _lib.lib()


def timed(n=40):
    for _ in range(5):
        e.forward(regress=False, raster=raster)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); e.forward(regress=False, raster=raster); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,1,2').split(',')]
dbgs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '0').split(',')]
for m in modes:
    _lib.check(L.mh_lbs_set_forward_pipeline(m))
    for d in (dbgs if m == 2 else [0]):
        os.environ['MHHIP_FWDPC_DBG'] = str(d)
        for tpb in ([None] if len(sys.argv) <= 3 else [int(x) for x in sys.argv[3].split(',')]):
            if tpb is None:
                os.environ.pop('MHHIP_FWD_TPB', None)
            else:
                os.environ['MHHIP_FWD_TPB'] = str(tpb)
            med, mn = timed()
            print('pipe %d dbg %d tpb %s: forward (pose + skinning) median %.1f us, min %.1f us' % (m, d, tpb, med, mn), flush=True)
