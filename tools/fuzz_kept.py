"""developer tool: kept face lists against a fresh sort under random motion -- image sizes from 32x24 to 240x135 (landscape,
portrait), 1-4 humans, per-launch random perturbations of the translations / poses / shape from a hundredth of a pixel to
several pixels, sort margins 1-3: every 40-byte selection key of every launch must be identical."""
import os, sys, tempfile, pathlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic
from mhhip.raster import RasterTerms, set_sort_margin
from oracle import lbs_oracle as lo
import test_fit_full_gpu as tf
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct); om = lo.BodyModel(struct, regs)
rng = np.random.RandomState(int(os.environ.get('SEED', '3')))
bad = 0
for case in range(int(os.environ.get('CASES', '10'))):
    W, H = [(32, 24), (48, 80), (64, 36), (96, 54), (80, 80), (160, 90), (240, 135), (54, 96)][rng.randint(8)]
    T, N = int(rng.randint(2, 8)), int(rng.randint(1, 5))
    margin = int(rng.choice([1, 1, 2, 3]))
    amp = float(rng.choice([1e-4, 1e-3, 5e-3, 2e-2]))           # metres per launch (0.01 m ~ 0.2 px at 96x54 and 3 m)
    opt, dl, o, batches, seq = tf._setup(struct, regs, om, pathlib.Path(tempfile.mkdtemp()), T, N, W, H, max(1, T // 2), int(rng.randint(1 << 30)), False)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    kept, fresh = RasterTerms(e), RasterTerms(e)
    gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
    old = set_sort_margin(margin)
    ndiff = 0
    try:
        for c in range(16):
            e.leaf('poses_T').add_(torch.tensor(rng.normal(0, amp, (T, N, 3)).astype(np.float32), device=e.dev))
            e.leaf('poses_smpl').add_(torch.tensor(rng.normal(0, amp, (T, N, 72)).astype(np.float32), device=e.dev))
            if c == 9:
                e.leaf('poses_T')[::2, :, :2] += 0.05                                  # a jump
            e.cycle(c, raster=kept); torch.cuda.synchronize()
            _, _, k1 = kept.selection(e)
            set_sort_margin(0); fresh(e, gv, log, phases=1); torch.cuda.synchronize(); set_sort_margin(margin)
            _, _, k0 = fresh.selection(e)
            if k1.shape != k0.shape or not (k1 == k0).all():
                ndiff += 1
        seen, rebuilt = kept.sort_counters(e)
    finally:
        set_sort_margin(old)
    bad += 1 if ndiff else 0
    print('case %2d %3dx%-3d T%d N%d margin %d step %.0e m: launches with different keys %d of 16; lists rebuilt %d of %d'
          % (case, W, H, T, N, margin, amp, ndiff, rebuilt, seen), flush=True)
print('cases with differences:', bad)
sys.exit(1 if bad else 0)
