"""Pins the oracle's composition of the depth and silhouette terms (oracle/fit_oracle.SequenceOracle.batch_loss,
restating reference optimizer.py:425-477) to the REFERENCE loop: the fixtures come from the reference's own ``fit``
executing around a non-empty z-buffer / non-zero silhouette (tests/golden/make_golden_raster.py plugs
oracle/raster_oracle into the reference's pytorch3d call sites).  Both sides share the rasteriser restatement, so
what is pinned here is everything the reference does with the raster outputs; PyTorch3D's inside stays unpinned."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import fit_oracle as fo
from oracle import raster_oracle as ro

LEAVES = ['poses_T', 'poses_smpl', 'betas_smpl', 'zmin_lin', 'zmax_lin', 'xscale_factor']


def _batches(fin, b=5):
    out = []
    for s in range(0, fin['T'], b):
        sl = slice(s, s + b)
        out.append(dict(idxs=torch.arange(s, min(s + b, fin['T'])), pose2d=torch.tensor(fin['pose2d'][sl]),
                        seg_mask=torch.tensor(fin['seg_mask'][sl]), depths=torch.tensor(fin['depths'][sl]),
                        poses_smpl=torch.tensor(fin['poses_smpl'][sl])))
    return out


def _oracle(oracle_model, gr, scene):
    fin = gi.fit_raster_inputs(gr)
    rast = ro.make_rasteriser(oracle_model.faces, fin['cam_K'], (fin['W'], fin['H']))
    o = fo.SequenceOracle(oracle_model, (fin['W'], fin['H']), fin['T'], fin['cam_K'], coefs=gi.COEFS, rasteriser=rast)
    o.xscale = torch.zeros(1, fin['N'], 1, 1)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=gr['init_poses_T'])
    if scene:
        o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    return fin, o


def test_warmup_100_iterations(golden_raster, oracle_model):
    gr = golden_raster
    fin = gi.fit_raster_inputs(gr)
    o = fo.SequenceOracle(oracle_model, (fin['W'], fin['H']), fin['T'], fin['cam_K'], coefs=gi.COEFS)
    o.xscale = torch.zeros(1, fin['N'], 1, 1)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=100)
    got = dict(zip(LEAVES, [p.detach().numpy() for p in o.leaves()]))
    # 100 Adam steps end in the sign-like regime of m/sqrt(v): rounding differences are amplified to ~1-2 mm (the
    # 5-iteration fixture of test_oracle_golden.py pins the recurrence itself to 2e-5)
    for n in LEAVES:
        np.testing.assert_allclose(got[n], gr['init_' + n], atol=4e-3, err_msg=n)


@pytest.mark.parametrize('scene', [False, True])
def test_first_cycle_gradients_with_live_raster(golden_raster, oracle_model, scene):
    gr = golden_raster
    fin, o = _oracle(oracle_model, gr, scene)
    o.cycle_grads(_batches(fin))
    pre = 'scene_k1_grad_' if scene else 'k1_grad_'
    for n, p in zip(LEAVES, o.leaves()):
        g = gr[pre + n]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(g)
        np.testing.assert_allclose(got, g, atol=3e-4 * max(np.abs(g).max(), 1e-6), err_msg=n)
    # the depth-range leaves only receive gradient through target_disp of the depth term (:425)
    assert np.abs(gr[pre + 'zmin_lin']).max() > 0 and np.abs(gr[pre + 'zmax_lin']).max() > 0


def test_first_cycle_term_values(golden_raster, oracle_model):
    gr = golden_raster
    fin, o = _oracle(oracle_model, gr, False)
    dep, sil = [], []
    for data in _batches(fin):
        _, terms, _ = o.batch_loss(data)
        dep.append(float(terms['loss_depth']))
        sil.append(float(terms['loss_silhouette']))
    np.testing.assert_allclose(dep, gr['k1_loss_depth_per_batch'], rtol=2e-4)
    # the reference's masked-MSE builder is called once per gated (frame, rank): their sum is the term
    np.testing.assert_allclose(np.sum(sil), gr['k1_loss_sil_calls'].sum(), rtol=2e-4)
    assert int(gr['k1_rank_gate_differs']) > 0          # the inputs exercise the rank-indexed gate quirk (:472)


@pytest.mark.parametrize('k,scene', [(1, False), (5, False), (5, True)])
def test_leaves_after_k_cycles(golden_raster, oracle_model, k, scene):
    gr = golden_raster
    fin, o = _oracle(oracle_model, gr, scene)
    log = o.fit(_batches(fin), k)
    got = dict(zip(LEAVES, [p.detach().numpy() for p in o.leaves()]))
    pre = ('scene_k%d_' if scene else 'k%d_') % k
    for n in LEAVES:
        err = np.abs(got[n] - gr[pre + n])
        if k == 1:
            assert err.max() <= 2e-5, (n, err.max())
        else:
            # RMSprop's g/sqrt(v) turns a last-ulp difference of a near-zero gradient (sign() of the L1 priors, a
            # face-selection flip) into a full lr-sized step: measured 1 of 120 / 2 of 2880 entries at 4e-4 / 1.2e-3
            frac = float((err > 3e-4).mean())
            assert frac <= 0.01 and err.max() <= 5e-3, '%s: %.4f of entries above 3e-4, max %.2e' % (n, frac, err.max())
    ref = gr[pre + 'loss_depth_per_batch'].reshape(k, -1).mean(1)
    np.testing.assert_allclose([l['loss_depth'] for l in log], ref, rtol=5e-3)
