"""developer tool: distribution of face heights (pixel rows) and candidate entries per tile with one / two height classes"""
import os, sys, tempfile, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq, _lib
from mhhip.raster import RasterTerms
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), 200, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, 200, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False))
e = opt.engine
r = RasterTerms(e)
e.cycle(0, raster=r)
torch.cuda.synchronize()
L = _lib.lib()
off = (ctypes.c_size_t * 6)()
L.mh_raster_debug_offsets.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
L.mh_raster_debug_offsets(*r.dims, off)
B, V, F, H, W = e.B, e.V, r.dims[3], e.H, e.W
frows = r.ws[off[1]:off[1] + B * F * 4].view(torch.int32).cpu().numpy().reshape(B, F).view(np.uint32)
maxh = r.ws[off[4]:off[4] + B * 4].view(torch.int32).cpu().numpy()
win, koff, keys = r.selection(e)
lo = (frows & 0x7fff).astype(np.int64); hi = (frows >> 16).astype(np.int64)
ok = lo <= hi
hgt = np.where(ok, hi - lo, -1)
print('faces per body in lists: %.0f of %d' % (ok.sum() / B, F))
hh = hgt[ok]
for k in range(0, 12):
    print('  height %2d rows: %.4f' % (k, (hh == k).mean()))
print('maxh: mean %.1f  median %.0f  max %d' % (maxh.mean(), np.median(maxh), maxh.max()))
# candidate entries per tile: tiles = row strips of th = 640 // ww rows
for HS in (None, 1, 2, 3, 4):
    tot = 0; need = 0
    for b in range(0, B, 7):
        x0, y0, ww, wh = win[b]
        if ww <= 0: continue
        th = max(1, 640 // ww) if ww <= 640 else 10
        l, h = lo[b][ok[b]], hi[b][ok[b]]
        ht = h - l
        for r0 in range(y0, y0 + wh, th):
            r1 = min(r0 + th, y0 + wh) - 1
            need += int(((l <= r1) & (h >= r0)).sum())
            if HS is None:
                tot += int(((l >= r0 - maxh[b] - 1) & (l <= r1 + 1)).sum())
            else:
                short = ht <= HS
                tot += int((short & (l >= r0 - HS - 1) & (l <= r1 + 1)).sum()) + int((~short & (l >= r0 - maxh[b] - 1) & (l <= r1 + 1)).sum())
    print('height split', HS, ': entries scanned %.2f x the faces that really overlap a tile' % (tot / need), tot)
