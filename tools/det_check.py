"""developer tool: deterministic vs production gradient scatter on the raster fixture and on C3 (errors, bit equality, time)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import pathlib, tempfile
import golden_inputs as gi
from mhhip import synthetic, _lib
import test_optimizer_raster_gpu as tor
from mhhip.raster import RasterTerms
L = _lib.lib()
st = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, st)
gr = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_raster_cpu.npz'))
for scene in (False, True):
    res = {}
    for mode in (0, 1):
        L.mh_raster_set_deterministic(mode)
        runs = []
        for rep in range(3):
            tmp = pathlib.Path(tempfile.mkdtemp())
            fin, opt, dl = tor._start(st, regs, tmp, gr, scene)
            opt._stage_from_dataloader(dl)
            e = opt.engine
            e.cycle(0, raster=RasterTerms(e))
            torch.cuda.synchronize()
            runs.append(e.grads.clone())
        same = all(torch.equal(runs[0], r) for r in runs[1:])
        print('scene', scene, 'det', mode, 'bit-equal over 3 runs:', same, 'max|diff|', max(float((runs[0] - r).abs().max()) for r in runs[1:]))
        pre = 'scene_k1_grad_' if scene else 'k1_grad_'
        for n in tor.LEAVES:
            g = gr[pre + n]
            got = tor._leaf(opt, n, e.grads).reshape(g.shape)
            scale = max(np.abs(g).max(), 1e-8)
            err = np.abs(got - g) / scale
            print('   %-14s max %.2e  p99 %.2e  median %.2e  frac>2e-4 %.4f  n=%d' % (n, err.max(), np.percentile(err, 99), np.median(err), (err > 2e-4).mean(), err.size))
        res[mode] = runs[0]
    print('   det vs prod max rel diff', float((res[0] - res[1]).abs().max() / res[1].abs().max()))
