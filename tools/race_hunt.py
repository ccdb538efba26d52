"""developer tool: the rasteriser's selection launched over and over on a fixed state while other processes share the GPU
(start several copies): every launch's keys / per-body depth sums against the first launch's -- which bodies, which pixels,
which tiles differ when a launch is disturbed?   ITER=200 T=250 python tools/race_hunt.py"""
import hashlib, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq
from mhhip.raster import RasterTerms
T, ITER = int(os.environ.get('T', 250)), int(os.environ.get('ITER', 100))
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, bench.N_PEOPLE, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=20)
dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=bench.BATCH, shuffle=False)
opt._stage_from_dataloader(dl)
e = opt.engine
raster = RasterTerms(e)
e.cycle(0, raster=raster)          # state: verts, sil stats
torch.cuda.synchronize()
gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
import time
if os.environ.get('START_AT'):
    time.sleep(max(0.0, float(os.environ['START_AT']) - time.time()))      # every copy starts its launches at the same moment
phase = int(os.environ.get('PHASES', 1))
ref = None
bad = 0
churn = os.environ.get('CHURN') == '1'
for it in range(ITER):
    if churn:                      # allocator / copy traffic beside the launches (what another process' set-up does)
        x = torch.empty(64 << 20, dtype=torch.uint8, device='cuda'); x.copy_(torch.empty(64 << 20, dtype=torch.uint8).pin_memory(), non_blocking=True); del x; torch.cuda.empty_cache()
    if os.environ.get('FRESH') == '1':
        raster.init_workspace()       # sort every launch
    e._projected_into = None
    raster(e, gv, log, phases=phase)
    torch.cuda.synchronize()
    win, koff, keys = raster.selection(e)
    depth = e.depth_body.cpu().numpy().copy()
    if ref is None:
        ref = (win.copy(), koff.copy(), keys.copy(), depth)
        continue
    dk = (keys != ref[2]).any(axis=1) if keys.shape == ref[2].shape else None
    dd = depth != ref[3]
    if dk is None or dk.any() or dd.any() or (win != ref[0]).any():
        bad += 1
        if bad <= 4:
            print('launch %d: windows differ %s; %s key pixels differ; depth sums differ for bodies %s' % (
                it, (win != ref[0]).any(), 'shape' if dk is None else int(dk.sum()), np.nonzero(dd)[0][:10]))
            if dk is not None and dk.any():
                px = np.nonzero(dk)[0]
                bodies = np.searchsorted(koff, px, side='right') - 1
                for b in np.unique(bodies)[:4]:
                    loc = px[bodies == b] - koff[b]
                    ww = win[b, 2]
                    rows, cols = loc // ww, loc % ww
                    print('   body %d window %s: %d pixels, rows %d..%d cols %d..%d; first: ref %s now %s' % (
                        b, win[b], len(loc), rows.min(), rows.max(), cols.min(), cols.max(), ref[2][px[bodies == b][0]], keys[px[bodies == b][0]]))
print('%d of %d launches differ from the first' % (bad, ITER - 1))
