"""developer tool: phase timings of the two skinning kernels from a timing build (tools/mkvariant.sh lt mh_lbs.hip -DMH_EXPERIMENT -DLBS_TIMING;
MHHIP_LIB=variants/lib_lt.so python tools/time_lbs_phases.py): wave-elapsed cycles by phase, inside the replayed C3 cycle"""
import ctypes, os, sys, tempfile
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
for i in range(5):
    e.cycle_graphed(0, raster=r); e.step(1e-4)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
L.mh_lbs_debug_timing(out)
n = 20
for i in range(n):
    e.cycle_graphed(0, raster=r); e.step(1e-4)
L.mh_lbs_debug_timing(out)
o = [int(x) for x in out]
fw, bw = max(1, o[4]), max(1, o[15])
print('forward  per wave (cycles): staging %d, constants + matrix %d, epilogue %d, everything %d; waves per launch %d' % (
    o[0] // fw, o[1] // fw, o[2] // fw, o[3] // fw, fw // n))
print('backward per wave (cycles): staging %d, stage %d, wait A %d, blend %d, wait B %d, matrix %d, everything %d; waves per launch %d' % (
    o[8] // bw, o[9] // bw, o[10] // bw, o[11] // bw, o[12] // bw, o[13] // bw, o[14] // bw, bw // n))
