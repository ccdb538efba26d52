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
if os.environ.get('R_TIMING') in ('6', '7'):
    from mhhip.raster import set_sort_margin
    e.leaf('poses_T')[::7, :, 1] += 0.05      # every seventh frame moves: its bodies sort, the others keep their lists
    e.forward(regress=False, raster=r)
if os.environ.get('IN_CYCLE') == '1':
    # the last launch measured is one inside the replayed cycle (side branch beside it, work lists a cycle old)
    e.scene_pts = None
    for c in range(30):
        e.cycle_graphed(c % 8, raster=r)
        e.step(0.0)
for _ in range(0 if os.environ.get('IN_CYCLE') == '1' else 4): r(e, gv, log, phases=3 if os.environ.get('R_TIMING') in ('4', '5') else 1)
torch.cuda.synchronize()
a1 = r.pair_counters(e)
n = a1[0] - a0[0]
print('launches', n, 'counter A per launch %.0f' % ((a1[1] - a0[1]) / n), 'counter B per launch %.0f' % ((a1[2] - a0[2]) / n))
if os.environ.get('R_TIMING') == '3':
    a = r.pair_counters(e)
    print('timing build 3: last launch spans %.1f us, workgroup life spans sum to %.1f us = %.1f %% of 512 slots' % (a[1] / 100., a[2] / 100., 100. * a[2] / max(1, a[1]) / 512))
elif os.environ.get('R_TIMING') in ('6', '7'):
    A = a1[1] - a0[1]; B = a1[2] - a0[2]
    print('timing build %s (k_raster_prepare, sorting workgroups, cycles summed over %d launches): %s' % (os.environ['R_TIMING'], n, [A & 0xffffffff, A >> 32, B & 0xffffffff, B >> 32]))
elif os.environ.get('R_TIMING'):
    # timing builds: two 32-bit halves per counter, units of 1024 wave-cycles (csrc/mh_raster.hip, R_TMARK)
    A = a1[1] - a0[1]; B = a1[2] - a0[2]
    print('timing build %s: wave-cycles per launch by phase %s' % (os.environ['R_TIMING'], [x * 1024 // n for x in (A & 0xffffffff, A >> 32, B & 0xffffffff, B >> 32)]))
