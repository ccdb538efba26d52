"""The per-person sums of the shape / scale gradients inside the update's launch (round 6): a captured
cycle of ``fit`` leaves the per-body rows of the LBS backward where they are (``mh_lbs_backward*`` with gbetas = gxscale =
NULL), ``mh_rmsprop_step_person`` sums them over a person's frames in extra workgroups of the update -- in the order
``k_person_reduce`` uses -- and updates those leaves; one launch and one gap less on the cycle's chain.  The fit must be the
same fit bit for bit (reference: optimizer.py:343-356 the optimiser and its leaves, :586-587 backward + step)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fit(smpl_struct, smpl_regs, tmp_path, defer, num_iter, optim_scale=True, T=6, N=2, W=120, H=68, batch=2, seed=43):
    from mhhip import synthetic, synthetic_seq
    from mhhip.raster import set_deterministic
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    import golden_inputs as gi
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    c = gi.COEFS
    K = synthetic.default_cam_K((W, H), 60.0)
    old_env = os.environ.get('MHHIP_DEFER_PERSON')
    os.environ['MHHIP_DEFER_PERSON'] = '1' if defer else '0'
    old_det = set_deterministic(True)
    try:
        opt = SMPLDepthSequenceOptimizer(
            image_size=(W, H), num_frames=T, fov=60, device='cuda:0', smpl_model_parameters_path=str(tmp_path),
            smpl_data_struct=smpl_struct, scene_update='device', cam_K=K,
            proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
            reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
            reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])
        seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, N, T, (W, H), seed, cam_K=K, z_range=(2.6, 3.6))
        opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=30,
                                     scale_factor=None if optim_scale else np.full((N,), 1.04, np.float32))
        dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=batch, shuffle=False)
        log = opt.fit(dl, num_iter=num_iter, update_filters_every=7)
        torch.cuda.synchronize()
        e = opt.engine
        return dict(params=e.params.cpu().numpy().copy(), grads=e.grads.cpu().numpy().copy(), sq=e.sq.cpu().numpy().copy(),
                    log=log, e=e, keys=list(getattr(e, '_graphs', {}).keys()))
    finally:
        set_deterministic(old_det)
        if old_env is None:
            os.environ.pop('MHHIP_DEFER_PERSON', None)
        else:
            os.environ['MHHIP_DEFER_PERSON'] = old_env


@pytest.mark.parametrize('optim_scale', [True, False])
def test_fit_is_the_same_fit_with_the_sums_in_the_update(smpl_struct, smpl_regs, tmp_path, optim_scale):
    n = 38
    a = _fit(smpl_struct, smpl_regs, tmp_path, False, n, optim_scale)
    b = _fit(smpl_struct, smpl_regs, tmp_path, True, n, optim_scale)
    assert any(k[-1] is True for k in b['keys']) and not any(k[-1] is True for k in a['keys']), 'the deferring form must be the one that ran'
    np.testing.assert_array_equal(b['params'], a['params'])
    np.testing.assert_array_equal(b['sq'], a['sq'])
    e = b['e']
    lo, hi = int(e.offs[4]), int(e.offs[6])
    if optim_scale:
        # the sums are added to the gradient buffer too (by the update's launch): the buffer reads as it always did
        np.testing.assert_array_equal(b['grads'], a['grads'])
    else:
        np.testing.assert_array_equal(b['grads'][:int(e.offs[5])], a['grads'][:int(e.offs[5])])
        assert np.all(b['params'][int(e.offs[5]):hi] == a['params'][int(e.offs[5]):hi])
    assert np.any(a['grads'][lo:int(e.offs[5])] != 0)
    for c in range(n):
        for k in a['log'][c]:
            if k == 'reg_contact':
                np.testing.assert_allclose(b['log'][c][k], a['log'][c][k], rtol=1e-6, err_msg='cycle %d' % c)
            else:
                assert b['log'][c][k] == a['log'][c][k], 'log entry %s of cycle %d' % (k, c)


def test_person_update_entry_matches_reduce_then_update():
    """the C entry alone: mh_rmsprop_step_person on per-body rows == k_person_reduce's sums added, then mh_rmsprop_step_log"""
    from mhhip import _lib, engine
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(5)
    B, NB, nbeta = 203 * 3, 3, 10           # (not a multiple of anything)
    n = 1000 + NB * nbeta + NB
    ob, ox = 1000, 1000 + NB * nbeta
    gb = torch.randn(B, nbeta, generator=g).to(dev)
    gx = torch.randn(B, generator=g).to(dev)
    p0 = torch.randn(n, generator=g).to(dev)
    g0 = torch.randn(n, generator=g).to(dev)
    sq0 = torch.rand(n, generator=g).to(dev)
    bf0 = torch.randn(n, generator=g).to(dev)
    src = torch.arange(16, dtype=torch.float32, device=dev)
    for freeze in (False, True):
        # reference: sums in k_person_reduce's order = thread t takes bodies n + NB (t + 256 i), then the 256-leaf tree
        def tree(x):                                        # x: (B_n,) values of one (person, component) in frame order
            s = np.zeros(256, np.float32)
            for t in range(min(256, len(x))):
                a = np.float32(0)
                for v in x[t::256]:
                    a = np.float32(a + v)
                s[t] = a
            o = 128
            while o > 0:
                s[:o] = s[:o] + s[o:2 * o]
                o >>= 1
            return s[0]
        gbh, gxh = gb.cpu().numpy(), gx.cpu().numpy()
        g1 = g0.clone()
        add = np.zeros(n, np.float32)
        for pn in range(NB):
            for q in range(nbeta):
                add[ob + pn * nbeta + q] = tree(gbh[pn::NB, q])
            if not freeze:
                add[ox + pn] = tree(gxh[pn::NB])
        g1 += torch.from_numpy(add).to(dev)
        p1, sq1, bf1 = p0.clone(), sq0.clone(), bf0.clone()
        d1 = torch.zeros(16, device=dev)
        engine.rmsprop_step_log(p1, g1, sq1, bf1, 0.01, src, d1)
        p2, g2, sq2, bf2 = p0.clone(), g0.clone(), sq0.clone(), bf0.clone()
        d2 = torch.zeros(16, device=dev)
        poke = torch.zeros(2, dtype=torch.int32, device=dev)
        ps = _lib.PersonSums(gb.data_ptr(), gx.data_ptr(), B, NB, nbeta, ob, -1 if freeze else ox)
        engine.rmsprop_step_log(p2, g2, sq2, bf2, 0.01, src, d2, poke_dst=poke, poke=(1, 7), person=ps)
        torch.cuda.synchronize()
        for x, y in ((p1, p2), (g1, g2), (sq1, sq2), (bf1, bf2), (d1, d2)):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
        assert poke.tolist() == [1, 7]
    # argument checks
    bad = _lib.PersonSums(gb.data_ptr(), gx.data_ptr(), B, NB, nbeta, n - 3, ox)
    with pytest.raises(_lib.MhError):
        engine.rmsprop_step_log(p2, g2, sq2, bf2, 0.01, None, None, person=bad)
