import os, sys, pathlib, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_sort_margin
from oracle import lbs_oracle
import test_full_size_gpu as tf
st = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, st)
om = lbs_oracle.BodyModel(st, regs)
T, N, W, H, batch = 100, 4, 240, 135, 10
opt, dl, o, batches, seq = tf._setup(st, regs, om, pathlib.Path(tempfile.mkdtemp()), T, N, W, H, batch, 47, True)
opt._stage_from_dataloader(dl)
e = opt.engine
kept, fresh, fresh2 = RasterTerms(e), RasterTerms(e), RasterTerms(e)
kept.ws.copy_(torch.randint(0, 256, kept.ws.shape, dtype=torch.uint8, device=kept.ws.device)); kept.init_workspace()
gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
set_sort_margin(1)
lr = 0.01
for c in range(40):
    e.cycle(c, raster=kept)
    torch.cuda.synchronize()
    win, koff, k1 = kept.selection(e)
    kept(e, gv, log, phases=1); torch.cuda.synchronize()
    _, _, k1b = kept.selection(e)
    set_sort_margin(0)
    fresh(e, gv, log, phases=1); torch.cuda.synchronize()
    _, _, k0 = fresh.selection(e)
    fresh2(e, gv, log, phases=1); torch.cuda.synchronize()
    _, _, k0b = fresh2.selection(e)
    set_sort_margin(1)
    d_kk = int((k1 != k1b).any(axis=1).sum()); d_ff = int((k0 != k0b).any(axis=1).sum()); d_kf = int((k1 != k0).any(axis=1).sum())
    if d_kk or d_ff or d_kf:
        print('cycle', c, 'kept vs kept-again', d_kk, '| fresh vs fresh-again', d_ff, '| kept vs fresh', d_kf)
        import ctypes
        from mhhip import _lib
        L = _lib.lib()
        off = (ctypes.c_size_t * 6)()
        L.mh_raster_debug_offsets.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
        L.mh_raster_debug_offsets(*kept.dims, off)
        B_, V_, F_ = e.B, e.V, kept.dims[3]
        faces = kept.faces.cpu().numpy().reshape(-1, 3)
        def arr(r, i, n, dt):
            return r.ws[off[i]:off[i] + n].view(dt).cpu().numpy()
        for px in np.nonzero((k1 != k0).any(axis=1))[0][:2]:
            body = int(np.searchsorted(koff, px, side='right') - 1)
            fk = set(int(x & 0xffffffff) for x in k1[px] if x != 0xffffffffffffffff); ff = set(int(x & 0xffffffff) for x in k0[px] if x != 0xffffffffffffffff)
            loc = px - koff[body]; ww = win[body, 2]; xi = win[body, 0] + loc % ww; yi = win[body, 1] + loc // ww
            print('   pixel', xi, yi, 'body', body, 'win', win[body], 'only kept', fk - ff, 'only fresh', ff - fk)
            for r, nm in ((kept, 'kept'), (fresh, 'fresh')):
                ndc = arr(r, 0, B_ * V_ * 12, torch.float32).reshape(B_, V_, 3)[body]
                frows = arr(r, 1, B_ * F_ * 4, torch.int32).reshape(B_, F_)[body].view(np.uint32)
                maxh = arr(r, 4, B_ * 4, torch.int32)[body]
                rowb = arr(r, 5, B_ * V_ * 4, torch.float32).reshape(B_, V_)[body]
                fs = arr(r, 2, B_ * F_ * 4, torch.int32).reshape(B_, F_)[body].view(np.uint32)
                rs = arr(r, 3, B_ * (3 * (H + 1) + 1) * 4, torch.int32).reshape(B_, -1)[body]
                for f in (fk ^ ff):
                    v = faces[f]
                    y = ndc[v, 1]; rows_now = (H - 0.5 - 0.5 * H) - y * (H / 2.0)
                    fr = int(frows[f]); inlist = np.nonzero((fs[:rs[-1]] & 0xfffff) == f)[0]
                    print('      ', nm, 'face', f, 'verts', v, 'rows now', rows_now, 'rowb', rowb[v], 'frows lo', fr & 0x7fff, 'hi', fr >> 16, 'maxh', maxh, 'in fsort at', inlist, 'entry hi', [int(fs[i] >> 20) for i in inlist], 'nlist', rs[-1])
        for a, b, nm in ((k1, k0, 'kept/fresh'), (k0, k0b, 'fresh/fresh')):
            for px in np.nonzero((a != b).any(axis=1))[0][:3]:
                body = int(np.searchsorted(koff, px, side='right') - 1)
                print('   ', nm, 'body', body, 'slots', [(int(a[px, k] & 0xffffffff), int(b[px, k] & 0xffffffff), hex(int(a[px, k] >> 32)), hex(int(b[px, k] >> 32))) for k in range(5) if a[px, k] != b[px, k]])
    e.step(lr); lr *= 0.99
    if c == 19:
        e.leaf('poses_T')[::4, :, 1] += 0.08
print('done', kept.sort_counters(e))
