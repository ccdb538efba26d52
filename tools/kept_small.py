"""developer tool: kept face lists against a fresh sort, cycle by cycle, on a SMALL image (96x54, 2 humans x 6 frames)"""
import os, sys, tempfile, pathlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_sort_margin
from oracle import lbs_oracle as lo
import test_fit_full_gpu as tf
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct); om = lo.BodyModel(struct, regs)
W, H = [int(x) for x in os.environ.get('IMG', '96x54').split('x')]
T, N, batch = 6, 2, 3
opt, dl, o, batches, seq = tf._setup(struct, regs, om, pathlib.Path(tempfile.mkdtemp()), T, N, W, H, batch, 43, True)
opt._stage_from_dataloader(dl)
e = opt.engine
kept, fresh = RasterTerms(e), RasterTerms(e)
gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
set_sort_margin(1)
lr = 0.01
from oracle import raster_oracle as ro
faces = np.asarray(struct.f).astype(np.int64)
K = synthetic.default_cam_K((W, H), 60.0)
ndc_hist = []
for c in range(3):
    e.cycle(c, raster=kept); torch.cuda.synchronize()
    ndc_hist.append(ro.to_ndc(e.verts.cpu(), K, (W, H)).numpy())
    win1, koff1, k1 = kept.selection(e)
    set_sort_margin(0); fresh(e, gv, log, phases=1); torch.cuda.synchronize(); set_sort_margin(1)
    win0, koff0, k0 = fresh.selection(e)
    same = k1.shape == k0.shape and bool((k1 == k0).all())
    print('cycle %2d: kept == fresh: %s  (rebuilt so far %s)' % (c, same, kept.sort_counters(e)), flush=True)
    if not same:
        bad = np.nonzero((k1 != k0).any(axis=1))[0]
        print('   %d window pixels differ; windows equal: %s' % (len(bad), bool((win1 == win0).all())))
        rk = H / 2.0; ra = H - 0.5 - 0.5 * H            # row = ra - y * rk  (r_row_affine, W >= H)
        bd = float(np.sqrt(1e-4))
        for px in bad[:4]:
            b = int(np.searchsorted(koff1, px, side='right') - 1); loc = px - koff1[b]; ww = win1[b, 2]
            x, y = win1[b, 0] + loc % ww, win1[b, 1] + loc // ww
            print('   body %d pixel (%d,%d) window %s kept %s fresh %s' % (b, x, y, win1[b], k1[px] & 0xffffffff, k0[px] & 0xffffffff))
            missing = set(int(v) for v in (k0[px] & 0xffffffff)) - set(int(v) for v in (k1[px] & 0xffffffff))
            for f in sorted(missing)[:3]:
                for ci, nd in enumerate(ndc_hist):
                    yy = nd[b][faces[f]][:, 1]
                    lo = np.ceil(ra - (yy.max() + bd) * rk - 1e-3); hi = np.floor(ra - (yy.min() - bd) * rk + 1e-3)
                    print('       face %d at cycle %d: rows %d..%d (continuous %.3f..%.3f)' % (f, ci, lo, hi, ra - (yy.max() + bd) * rk, ra - (yy.min() - bd) * rk))
    e.step(lr); lr *= 0.99
