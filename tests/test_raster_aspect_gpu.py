"""The drop-in with the rasteriser live on a portrait image (54 x 90) and a square one (64 x 64, three humans) against the
REFERENCE's own loop (tests/golden/reference_raster_aspect_cpu.npz; the oracle is pinned to the same fixture in
tests/test_oracle_golden_raster_aspect.py): NDC scale and offsets of the kernels for H > W and H = W (transforms.py:222-255),
the composition of the depth and silhouette terms, cycle-1 gradients and the leaves after the step."""
import numpy as np
import pytest
import torch

from aspect_inputs import VARIANTS, load
from test_optimizer_raster_gpu import LEAVES, _leaf, _start

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag', VARIANTS)
def test_first_cycle_other_aspects(smpl_struct, smpl_regs, tmp_path, tag):
    from mhhip.raster import RasterTerms
    gr = load(tag)
    fin, opt, dl = _start(smpl_struct, smpl_regs, tmp_path, gr, True)
    assert (fin['H'] > fin['W']) == (tag == 'por')
    opt._stage_from_dataloader(dl)
    e = opt.engine
    e.cycle(0, raster=RasterTerms(e))
    torch.cuda.synchronize()
    log = e.read_log(1, nbatches_total=None)[0]
    np.testing.assert_allclose(log['loss_depth'], gr['k1_loss_depth_per_batch'].mean(), rtol=2e-3)
    np.testing.assert_allclose(log['loss_silhouette'], gr['k1_loss_sil_calls'].sum() / len(gr['k1_loss_depth_per_batch']), rtol=2e-3)
    for n in LEAVES:
        g = gr['k1_grad_' + n]
        got = _leaf(opt, n, e.grads).reshape(g.shape)
        scale = max(np.abs(g).max(), 1e-8)
        err = np.abs(got - g)
        np.testing.assert_allclose(got, g, atol=1.5e-3 * scale, rtol=0, err_msg=n)          # every entry (blur-band flips)
        assert np.median(err) < 3e-4 * scale and np.percentile(err, 99) < 6e-4 * scale, (n, float(np.median(err)), float(err.max()))


@pytest.mark.parametrize('tag', VARIANTS)
def test_leaves_after_one_cycle_other_aspects(smpl_struct, smpl_regs, tmp_path, tag):
    gr = load(tag)
    fin, opt, dl = _start(smpl_struct, smpl_regs, tmp_path, gr, True)
    opt.fit(dl, num_iter=1)
    for n in LEAVES:
        want = gr['k1_' + n]
        err = np.abs(_leaf(opt, n).reshape(want.shape) - want)
        assert (err > 5e-5).mean() <= 0.01 and err.max() <= 2.5e-2, '%s: %.4f of entries off, max %.2e' % (n, float((err > 5e-5).mean()), float(err.max()))
