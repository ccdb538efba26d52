"""The rasterised terms at other aspect ratios -- a portrait image (54 x 90) and a square one (64 x 64, three humans) --
against the reference's own ``fit`` (tests/golden/make_golden_raster_aspect.py): the reference converts its intrinsics to the
NDC convention with one branch per aspect (transforms.py:222-255), and its stub rasteriser takes the NDC coordinates from the
reference's own camera objects -- so the oracle's (and the kernels') NDC scale / offsets for H > W and H = W are pinned here."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from aspect_inputs import VARIANTS, load
from test_oracle_golden_raster import LEAVES, _batches, _oracle


@pytest.mark.parametrize('tag', VARIANTS)
def test_first_cycle_with_live_raster_other_aspects(oracle_model, tag):
    gr = load(tag)
    fin, o = _oracle(oracle_model, gr, True)
    assert (fin['H'] > fin['W']) == (tag == 'por') and (fin['H'] == fin['W']) == (tag == 'sq')
    dep, sil = [], []
    for data in _batches(fin):
        _, terms, _ = o.batch_loss(data)
        dep.append(float(terms['loss_depth']))
        sil.append(float(terms['loss_silhouette']))
    np.testing.assert_allclose(dep, gr['k1_loss_depth_per_batch'], rtol=2e-4)
    np.testing.assert_allclose(np.sum(sil), gr['k1_loss_sil_calls'].sum(), rtol=2e-4)
    fin, o = _oracle(oracle_model, gr, True)
    o.cycle_grads(_batches(fin))
    for n, p in zip(LEAVES, o.leaves()):
        g = gr['k1_grad_' + n]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(g)
        np.testing.assert_allclose(got, g, atol=3e-4 * max(np.abs(g).max(), 1e-6), err_msg=n)
    fin, o = _oracle(oracle_model, gr, True)
    o.fit(_batches(fin), 1)
    for n, p in zip(LEAVES, o.leaves()):
        err = np.abs(p.detach().numpy() - gr['k1_' + n])
        assert (err > 5e-5).mean() <= 0.01 and err.max() <= 2.5e-2, (n, float((err > 5e-5).mean()), float(err.max()))
