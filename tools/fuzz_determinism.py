"""developer tool: run-to-run identity on random scenes -- the selection keys of repeated launches (the selection must not
depend on which wave reaches a pixel first), and in the deterministic mode every bit of every gradient leaf of repeated
cycles"""
import os, sys, tempfile, pathlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_deterministic, set_sort_margin
from oracle import lbs_oracle as lo
import test_fit_full_gpu as tf
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct); om = lo.BodyModel(struct, regs)
rng = np.random.RandomState(int(os.environ.get('SEED', '3')))
bad = 0
for case in range(int(os.environ.get('CASES', '12'))):
    W, H = [(48, 80), (96, 54), (80, 80), (160, 90), (240, 135)][rng.randint(5)]
    T, N = int(rng.randint(2, 12)), int(rng.randint(1, 5))
    scene = bool(rng.randint(2))
    opt, dl, o, batches, seq = tf._setup(struct, regs, om, pathlib.Path(tempfile.mkdtemp()), T, N, W, H, max(1, T // 2), int(rng.randint(1 << 30)), scene)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    r = RasterTerms(e)
    keys, grads = [], []
    old = set_deterministic(True)
    try:
        for rep in range(5):
            e.cycle(rep, raster=r); torch.cuda.synchronize()
            keys.append(r.selection(e)[2].copy()); grads.append(e.grads.clone())
    finally:
        set_deterministic(old)
    same_k = all(k.shape == keys[0].shape and (k == keys[0]).all() for k in keys[1:])
    same_g = all(torch.equal(g, grads[0]) for g in grads[1:])
    # production scatter: how far apart are two runs?
    e.cycle(0, raster=r); torch.cuda.synchronize(); ga = e.grads.clone()
    e.cycle(1, raster=r); torch.cuda.synchronize(); gb = e.grads.clone()
    spread = float((ga - gb).abs().max() / max(float(ga.abs().max()), 1e-30))
    bad += 0 if (same_k and same_g) else 1
    print('case %2d %3dx%-3d T%-2d N%d scene %d: keys identical over 5 launches %s, deterministic gradients bit-identical %s; atomics-mode spread %.1e'
          % (case, W, H, T, N, scene, same_k, same_g, spread), flush=True)
print('cases with run-to-run differences:', bad)
