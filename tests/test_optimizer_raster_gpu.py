"""GPU parity of the drop-in optimiser WITH the rasteriser live against the reference's own loop
(tests/golden/reference_raster_cpu.npz: the reference ``fit`` executed around the oracle rasteriser through the
pytorch3d stubs, tests/golden/make_golden_raster.py).  Pins the HIP composition of the depth and silhouette terms
(reference optimizer.py:425-477: target disparity, supervision mask, clamp, mean-log-disparity, near->far order,
rank-indexed gate, accumulated occlusion mask) -- per-leaf gradients of cycle 1, term values, leaves after 1 / 5
cycles, with and without an injected scene."""
import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu

LEAVES = ['poses_T', 'poses_smpl', 'betas_smpl', 'zmin_lin', 'zmax_lin', 'xscale_factor']
ENGINE_NAME = {'poses_T': 'poses_T', 'poses_smpl': 'poses_smpl', 'betas_smpl': 'betas', 'zmin_lin': 'zmin_lin',
               'zmax_lin': 'zmax_lin', 'xscale_factor': 'xscale'}


class _DS(torch.utils.data.Dataset):
    def __init__(self, fin):
        self.f = fin

    def __len__(self):
        return self.f['T']

    def __getitem__(self, i):
        f = self.f
        return dict(images=f['images'][i], depths=f['depths'][i], seg_mask=f['seg_mask'][i], backmasks=f['backmasks'][i],
                    pose2d=f['pose2d'][i], poses_smpl=f['poses_smpl'][i], betas_smpl=f['betas_smpl'][i],
                    valid_smpl=f['valid_smpl'][i], idxs=i)


def _start(smpl_struct, smpl_regs, tmp_path, gr, scene):
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    fin = gi.fit_raster_inputs(gr)
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    c = gi.COEFS
    opt = SMPLDepthSequenceOptimizer(
        image_size=(fin['W'], fin['H']), num_frames=fin['T'], cam_K=fin['cam_K'], device='cuda:0',
        smpl_model_parameters_path=str(tmp_path), smpl_data_struct=smpl_struct, use_rasteriser=True, scene_update='none',
        proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
        reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
        reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])
    opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=0)
    opt.engine.leaf('poses_T').copy_(torch.tensor(gr['init_poses_T']).view(fin['T'], fin['N'], 3))
    opt.engine.leaf('zmax_lin').copy_(torch.tensor(gr['init_zmax_lin']).view(-1))
    if scene:
        opt.scene_depth = fin['scene_depth']
        opt.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    dl = torch.utils.data.DataLoader(_DS(fin), batch_size=5, shuffle=False)
    return fin, opt, dl


def _leaf(opt, n, buf=None):
    return opt.engine.leaf(ENGINE_NAME[n], buf).cpu().numpy()


def test_warmup_lands_on_the_reference_start(golden_raster, smpl_struct, smpl_regs, tmp_path):
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    gr = golden_raster
    fin, opt, _ = _start(smpl_struct, smpl_regs, tmp_path, gr, False)
    opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=100)
    # 100 Adam steps: rounding differences are amplified to ~1-2 mm in the sign-like end regime (CPU oracle: 1.7 mm)
    np.testing.assert_allclose(_leaf(opt, 'poses_T').reshape(gr['init_poses_T'].shape), gr['init_poses_T'], atol=4e-3)
    np.testing.assert_allclose(_leaf(opt, 'zmax_lin').reshape(gr['init_zmax_lin'].shape), gr['init_zmax_lin'], atol=8e-3)


@pytest.mark.parametrize('scene', [False, True])
@pytest.mark.parametrize('det', [False, True])
def test_first_cycle_gradients_with_live_raster(golden_raster, smpl_struct, smpl_regs, tmp_path, scene, det):
    """Every entry of every leaf gradient of cycle 1 against the reference's own loop.  Measured (both modes alike):
    worst entry 7.8e-4 of the leaf's largest (poses_T), 99 % of poses_smpl within 2.2e-5, medians 1e-7..2e-6 -- the
    worst entries are blur-band flips of single pixels (tests/test_raster_gpu.py enumerates them), not summation order:
    the deterministic scatter differs from the atomics one by 1.2e-7."""
    from mhhip.raster import RasterTerms, set_deterministic
    gr = golden_raster
    old = set_deterministic(det)
    try:
        runs = []
        for rep in range(2 if det else 1):
            fin, opt, dl = _start(smpl_struct, smpl_regs, tmp_path, gr, scene)
            opt._stage_from_dataloader(dl)
            e = opt.engine
            e.cycle(0, raster=RasterTerms(e))
            torch.cuda.synchronize()
            runs.append(e.grads.clone())
    finally:
        set_deterministic(old)
    if det:
        assert torch.equal(runs[0], runs[1]), 'deterministic mode: the six gradient leaves must be bit-identical between runs'
    log = e.read_log(1, nbatches_total=None)[0]
    pre = 'scene_k1_grad_' if scene else 'k1_grad_'
    for n in LEAVES:
        g = gr[pre + n]
        got = _leaf(opt, n, e.grads).reshape(g.shape)
        scale = max(np.abs(g).max(), 1e-8)
        err = np.abs(got - g)
        np.testing.assert_allclose(got, g, atol=1.5e-3 * scale, rtol=0, err_msg=n)          # EVERY entry
        assert np.median(err) < 3e-4 * scale and np.percentile(err, 99) < 6e-4 * scale, \
            '%s: median %.2e, p99 %.2e, max %.2e (scale %.2e)' % (n, np.median(err), np.percentile(err, 99), err.max(), scale)
    if not scene:
        # the log holds the per-batch mean like the reference's optim_log (:588-590)
        np.testing.assert_allclose(log['loss_depth'], gr['k1_loss_depth_per_batch'].mean(), rtol=2e-3)
        np.testing.assert_allclose(log['loss_silhouette'], gr['k1_loss_sil_calls'].sum() / 4, rtol=2e-3)


@pytest.mark.parametrize('k,scene', [(1, False), (5, False), (5, True)])
def test_leaves_after_k_cycles(golden_raster, smpl_struct, smpl_regs, tmp_path, k, scene):
    gr = golden_raster
    fin, opt, dl = _start(smpl_struct, smpl_regs, tmp_path, gr, scene)
    log = opt.fit(dl, num_iter=k)
    pre = ('scene_k%d_' if scene else 'k%d_') % k
    for n in LEAVES:
        want = gr[pre + n]
        err = np.abs(_leaf(opt, n).reshape(want.shape) - want)
        # A FREE-RUNNING comparison with the reference's trajectory is statistical by construction: one RMSprop step moves
        # every entry by ~lr*sign(g)/sqrt(1-alpha) = 1.4e-2, with momentum up to 10x that over a few cycles, so an entry whose
        # gradient is below the rounding noise (or a pixel whose K-nearest selection is a near-tie, tests/test_raster_gpu.py)
        # takes the other branch and the two trajectories part there.  Measured at k=5: median 5e-5, 90th percentile up to
        # 9e-4 on the 120 poses_T entries, max 1e-2.  The strict statement -- every entry, every cycle, from identical
        # state -- is tests/test_fit_full_gpu.py::test_eight_cycles_step_by_step.
        if k == 1:
            frac = float((err > 5e-5).mean())
            assert frac <= 0.01 and err.max() <= 2.5e-2, '%s: %.4f of entries off, max %.2e' % (n, frac, err.max())
        else:
            p50, p90 = np.percentile(err, 50), np.percentile(err, 90)
            if err.size < 50:           # betas (20 entries), xscale (2): every entry within 1e-3 (measured 3e-4)
                assert err.max() <= 1e-3, '%s: max %.2e' % (n, err.max())
            else:
                assert p50 <= 1e-4 and p90 <= 2e-3 and err.max() <= 0.1, '%s: median %.2e, p90 %.2e, max %.2e' % (n, p50, p90, err.max())
    ref = gr[pre + 'loss_depth_per_batch'].reshape(k, -1).mean(1)
    np.testing.assert_allclose([l['loss_depth'] for l in log], ref, rtol=1e-2)
