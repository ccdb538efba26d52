"""developer tool: one random sequence of tools/fuzz_cycle.py (SEED, ONLY) -- the same cycle launched several times from the
same leaves: do the selection keys, the gradient leaves and the agreement with the oracle depend on whether the face lists
were sorted in that launch (first) or kept (later ones)?"""
import os, sys, tempfile, pathlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_deterministic, set_sort_margin
from oracle import lbs_oracle as lo
import test_fit_full_gpu as tf
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
omodel = lo.BodyModel(struct, regs)
rng = np.random.RandomState(int(os.environ.get('SEED', '5')))
set_deterministic(True)
if os.environ.get('MARGIN'):
    set_sort_margin(int(os.environ['MARGIN']))
for c in range(int(os.environ.get('ONLY', '0')) + 1):
    sizes = [(96, 54), (64, 96), (80, 80), (120, 68)] + ([(32, 24), (40, 72), (200, 40)] if os.environ.get('EDGE') == '1' else [])
    W, H = sizes[rng.randint(len(sizes))]
    T, N = int(rng.randint(1 if os.environ.get('EDGE') == '1' else 3, 14)), int(rng.randint(1, 7 if os.environ.get('EDGE') == '1' else 4))
    batch = int(rng.choice([2, 3, 5, 7]))
    scene = bool(rng.randint(2))
    seed = int(rng.randint(1 << 30))
opt, dl, o, batches, seq = tf._setup(struct, regs, omodel, pathlib.Path(tempfile.mkdtemp()), T, N, W, H, batch, seed, scene)
opt._stage_from_dataloader(dl)
e = opt.engine
raster = RasterTerms(e)
hsel = tf._HipSelectionRasteriser(np.asarray(struct.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N, wide=True)
o.rasteriser = hsel
print('%dx%d T%d N%d batch %d scene %d' % (W, H, T, N, batch, scene))
grads, keys = [], []
for run in range(3):
    e.cycle(run, raster=raster)
    torch.cuda.synchronize()
    grads.append(e.grads.clone())
    win, koff, k = raster.selection(e)
    keys.append(k.copy())
    hsel.take(raster, e, oracle=o)
    o.cycle_grads(batches)
    worst = 0.0
    for name, ename in tf.LEAF_MAP:
        w = tf._oracle_grad(o, name)
        g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
        worst = max(worst, float(np.abs(g - w).max() / max(np.abs(w).max(), 1e-8)))
    print('launch %d: worst leaf entry against the oracle %.1e; keys equal to launch 0: %s (%d of %d pixels differ); grads equal to launch 0: %s (max %.2e); lists sorted so far %s' % (
        run, worst, np.array_equal(keys[0], k), int((keys[0] != k).any(1).sum()), k.shape[0], bool(torch.equal(grads[0], grads[run])),
        float((grads[0] - grads[run]).abs().max()), raster.sort_counters(e)))
