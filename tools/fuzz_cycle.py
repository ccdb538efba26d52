"""developer tool: one full optimisation cycle (nine terms; then the one-euro filters and a second cycle with the
filtered-vertex term) of the drop-in against the CPU oracle on random small sequences -- frame counts that are not batch
multiples, 1-3 humans, portrait / landscape / square images, with and without a scene cloud.  Prints the worst entry of
every leaf gradient (relative to the leaf's largest) and of the loss log.  ONLY=<case> runs one case of the sequence;
F64=1 lets the oracle render in float64 (on faces of a fraction of a pixel its float32 autograd is itself ~1e-3 off);
F64=both evaluates it at both precisions and counts an entry as right when it agrees with either (a pixel centre within
rounding of a face's edge is decided by float32 arithmetic in the reference and in the kernel, differently in float64).
COEF_OFF=depth,silhouette takes terms out on both sides."""
import os, sys, tempfile, pathlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_deterministic
from oracle import lbs_oracle as lo
import test_fit_full_gpu as tf
import test_full_size_gpu as tfs
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
omodel = lo.BodyModel(struct, regs)
import golden_inputs as _gi
for _k in [k for k in os.environ.get('COEF_OFF', '').split(',') if k]:      # COEF_OFF=depth,silhouette: terms taken out on both sides
    _gi.COEFS[_k] = 0.0
rng = np.random.RandomState(int(os.environ.get('SEED', '5')))
worst = {}
set_deterministic(True)
for c in range(int(os.environ.get('CASES', '10'))):
    sizes = [(96, 54), (64, 96), (80, 80), (120, 68)] + ([(32, 24), (40, 72), (200, 40)] if os.environ.get('EDGE') == '1' else [])
    W, H = sizes[rng.randint(len(sizes))]
    T, N = int(rng.randint(1 if os.environ.get('EDGE') == '1' else 3, 14)), int(rng.randint(1, 7 if os.environ.get('EDGE') == '1' else 4))
    if os.environ.get('NSET'):
        N = int(rng.choice([int(x) for x in os.environ['NSET'].split(',')])); T = min(T, 3)
    batch = int(rng.choice([2, 3, 5, 7]))
    scene = bool(rng.randint(2))
    seed = int(rng.randint(1 << 30))
    if os.environ.get('ONLY') and c != int(os.environ['ONLY']):
        continue
    tmp = pathlib.Path(tempfile.mkdtemp())
    opt, dl, o, batches, seq = tf._setup(struct, regs, omodel, tmp, T, N, W, H, batch, seed, scene)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    raster = RasterTerms(e)
    hsel = tf._HipSelectionRasteriser(np.asarray(struct.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N,
                                       wide=os.environ.get('F64') == '1')
    o.rasteriser = hsel
    line = 'case %2d %3dx%-3d T%-2d N%d batch %d scene %d:' % (c, W, H, T, N, batch, scene)
    for cyc in range(2):
        if cyc == 1:
            e.update_filters(); o.update_filters()
        e.cycle(cyc, raster=raster)
        hsel.take(raster, e, oracle=o)
        log = e.read_log(cyc + 1)[cyc]
        if os.environ.get('F64') == 'both':
            want, both = tf._oracle_grads_both(o, hsel, batches)
        else:
            want, both = o.cycle_grads(batches), None
        lw = 0.0
        for k in tfs.LOG_KEYS + (['reg_filter_verts'] if cyc else []):
            lw = max(lw, abs(log[k] - want[k]) / max(abs(want[k]), 1e-6))
        gw = 0.0
        for name, ename in tf.LEAF_MAP:
            w = tf._oracle_grad(o, name) if both is None else both[name][1]
            g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
            err = np.abs(g - w) if both is None else np.minimum(np.abs(g - both[name][0]), np.abs(g - both[name][1]))
            r = float(err.max() / max(np.abs(w).max(), 1e-8))
            worst[name] = max(worst.get(name, 0.0), r)
            if r > 1e-4:
                i = int(np.argmax(np.abs(g - w)))
                print('    cycle %d leaf %s: entry %d hip %+.6e oracle %+.6e (largest of the leaf %.3e)' % (cyc, name, i, g.reshape(-1)[i], w.reshape(-1)[i], np.abs(w).max()))
            gw = max(gw, r)
        line += '  cycle %d: log %.1e grads %.1e' % (cyc, lw, gw)
    print(line, flush=True)
print('worst per leaf:', {k: '%.1e' % v for k, v in worst.items()})
