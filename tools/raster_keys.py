"""developer tool: selection keys (40 B per window pixel) + loss sums of the rasteriser on the C3 bench sequence, as a hash
and timing -- run with two builds (MHHIP_LIB=...) to check that a kernel change leaves the selection bit-identical."""
import hashlib, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq, _lib
from mhhip._lib import ptr, check
from mhhip.raster import RasterTerms

T = int(os.environ.get('T', '200'))
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False))
e = opt.engine
L = _lib.lib()
r = RasterTerms(e)
r.ws.zero_()
r.init_workspace()
e.cycle(0, raster=r)               # mask statistics + forward + raster
torch.cuda.synchronize()
W, H = bench.IMG
B = e.B
win, koff_, keys5 = r.selection(e)
npx = int(koff_[-1])
keys = keys5.view(np.uint8).reshape(-1)
print('window pixels', npx, 'keys sha1', hashlib.sha1(keys.tobytes()).hexdigest(), 'log', [float(x) for x in e.log[0][:3].cpu()])
gv = torch.zeros_like(e.verts); log = torch.zeros(16, device=e.dev)
def run():
    r(e, gv, log, phases=1)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): run()
e1.record(); torch.cuda.synchronize()
print('selection phase (windows + sort + strip + sums) %.1f us' % (e0.elapsed_time(e1) / 30 * 1e3))
ref = os.environ.get('KEYS_REF')
if ref and os.path.exists(ref):
    old = np.load(ref)
    a = keys.view(np.uint64).reshape(-1, 5); b = old.view(np.uint64).reshape(-1, 5)
    diff = np.nonzero((a != b).any(axis=1))[0]
    print('pixels with different keys:', len(diff))
    koff = np.concatenate([[0], np.cumsum(np.maximum(win[:, 2], 0).astype(np.int64) * np.maximum(win[:, 3], 0))])
    for px in diff[:8]:
        body = int(np.searchsorted(koff, px, side='right') - 1)
        loc = px - koff[body]
        ww = win[body, 2]
        print(' body', body, 'win', win[body], 'pixel (x,y)', win[body, 0] + loc % ww, win[body, 1] + loc // ww)
        for k in range(5):
            fa, fb = int(a[px, k] & 0xffffffff), int(b[px, k] & 0xffffffff)
            za, zb = np.array([a[px, k] >> 32], np.uint32).view(np.float32)[0], np.array([b[px, k] >> 32], np.uint32).view(np.float32)[0]
            print('   slot', k, 'new', fa if a[px, k] != 0xffffffffffffffff else None, za, '| old', fb if b[px, k] != 0xffffffffffffffff else None, zb)
if os.environ.get('KEYS_SAVE'):
    np.save(os.environ['KEYS_SAVE'], keys)
if hasattr(L, 'mh_raster_debug_counters'):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    L.mh_raster_debug_counters(buf)
    n = 36.0      # launches so far (1 cycle + 5 + 30)
    print('per launch: entries %.0f  row-hit %.0f  survivors %.0f  input rounds %.0f  bounded px %.0f of %.0f' % (buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[4] / n, buf[5] / n))
a = keys.view(np.uint64).reshape(-1, 5)
koff = np.concatenate([[0], np.cumsum(np.maximum(win[:, 2], 0).astype(np.int64) * np.maximum(win[:, 3], 0))])
tot = 0
for bdy in range(0, B, 50):
    blk = a[koff[bdy]:koff[bdy + 1]]
    f = (blk[blk != 0xffffffffffffffff] & 0xffffffff)
    tot += len(np.unique(f))
print('distinct selected faces per body (sample of %d bodies): %.0f ; live pixels per body %.0f' % (len(range(0, B, 50)), tot / len(range(0, B, 50)),
      float((a[:, 0] != 0xffffffffffffffff).sum()) / B))
