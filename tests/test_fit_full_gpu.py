"""End-to-end parity of one full optimisation cycle (ALL nine terms, rasteriser included) of the
drop-in optimiser on the GPU against the CPU oracle (oracle/fit_oracle.SequenceOracle driving
oracle/raster_oracle): per-leaf gradients of a cycle, loss log, and leaves after a few steps."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import fit_oracle as fo
from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu

LEAF_MAP = [('poses_T', 'poses_T'), ('poses_smpl', 'poses_smpl'), ('betas', 'betas'), ('zmin_lin', 'zmin_lin'),
            ('zmax_lin', 'zmax_lin'), ('xscale', 'xscale')]


def _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, seed, scene):
    from mhhip import engine, synthetic_seq
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    c = gi.COEFS
    from mhhip import synthetic
    K = synthetic.default_cam_K((W, H), 60.0)
    opt = SMPLDepthSequenceOptimizer(
        image_size=(W, H), num_frames=T, fov=60, device='cuda:0', smpl_model_parameters_path=str(tmp_path),
        smpl_data_struct=smpl_struct, scene_update='none', cam_K=K,
        proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
        reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
        reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, N, T, (W, H), seed, cam_K=K, z_range=(2.6, 3.6))
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=30)
    pT0 = opt.poses_T.cpu().numpy().copy()
    scene_depth = scene_mask = None
    if scene:
        ys = (np.arange(H, dtype=np.float32) + 0.5 - K[1, 2]) / K[1, 1]
        scene_depth = np.minimum(np.where(ys[:, None] > 1e-3, 1.15 / np.maximum(ys[:, None], 1e-3), 10.0), 10.0)
        scene_depth = np.tile(scene_depth, (1, W)).astype(np.float32)
        scene_mask = seq['backmasks'].min(axis=0) > 0
        opt.scene_depth = scene_depth
        opt.update_scene_pointcloud(scene_depth, scene_mask)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=batch, shuffle=False)
    # ---- oracle with the same start ----
    faces = np.asarray(smpl_struct.f).astype(np.int64)
    o = fo.SequenceOracle(oracle_model, (W, H), T, K, coefs=c, rasteriser=ro.make_rasteriser(faces, K, (W, H)))
    o.xscale = torch.zeros(1, N, 1, 1)
    o.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], poses_T=pT0)
    if scene:
        o.update_scene_pointcloud(scene_depth, scene_mask)
    batches = []
    for s in range(0, T, batch):
        sl = slice(s, s + batch)
        batches.append(dict(idxs=torch.arange(s, min(s + batch, T)), pose2d=torch.tensor(seq['pose2d'][sl]),
                            seg_mask=torch.tensor(seq['seg_mask'][sl]), depths=torch.tensor(seq['depths'][sl]),
                            poses_smpl=torch.tensor(seq['poses_smpl'][sl])))
    return opt, dl, o, batches, seq


class _HipSelectionRasteriser(object):
    """the oracle's differentiable rendering on the faces the HIP selection pass picked (per frame): separates "which
    faces" (tests/test_raster_gpu.py enumerates the differences and verifies them as near-ties) from "what comes out
    of them", so that the gradients can be compared on EVERY entry"""
    wants_frames = True

    def __init__(self, faces, K, image_size, N, wide=False):
        """wide: render (and differentiate) in float64 -- on faces of a fraction of a pixel the float32 autograd of the
        oracle's renderer is itself up to ~1e-3 off (tools/fuzz_cycle.py F64=1, tools/grad_debug.py)"""
        self.faces, self.K, self.size, self.N, self.sel, self.wide = faces, K, image_size, N, None, wide

    def take(self, raster, e, oracle=None):
        """the selection of the cycle the engine just ran; oracle: also evaluate the oracle's terms AT the engine's vertices
        (oracle/fit_oracle.py verts_value_override: sliver faces make the rasterised gradients ill-conditioned in the
        vertices, and the LBS forward has its own parity tests at 1e-5 m)"""
        from test_raster_gpu import _hip_selection
        self.sel = _hip_selection(raster.selection(e), e.B, e.H, e.W).reshape(e.T, self.N, e.H, e.W, 5)
        if oracle is not None:
            oracle.verts_value_override = e.verts.view(e.T, self.N, -1, 3).cpu().clone()

    def __call__(self, verts, frames):
        s = self.sel[np.asarray(frames)].reshape(-1, *self.sel.shape[2:])
        if self.wide:
            z, a = ro.render(verts.double(), self.faces, self.K, self.size, selection=(s[..., :1], s[..., 1:]))
            return z.to(verts.dtype), a.to(verts.dtype)
        return ro.render(verts, self.faces, self.K, self.size, selection=(s[..., :1], s[..., 1:]))


def _oracle_grads_both(o, hsel, batches):
    """the oracle's cycle with its renderer in float32 and in float64: (log, {leaf: (float32 grads, float64 grads)}).
    Neither precision is the truth for every entry: on a face of a fraction of a pixel the float32 AUTOGRAD of the renderer
    is up to 3e-3 off (float64 then agrees with the kernel to 1e-6), while a pixel centre within rounding of a face's edge
    is clipped / kept by the float32 rasteriser the reference runs -- and by the kernel -- but not in float64 (then float32
    agrees to 1e-5 and float64 is 7e-4 off).  tools/fuzz_cycle.py has found both kinds; an entry is right when it
    agrees with either."""
    out, log = {}, None
    for wide in (False, True):
        hsel.wide = wide
        lg = o.cycle_grads(batches)
        log = log or lg
        for name, _ in LEAF_MAP:
            out.setdefault(name, []).append(_oracle_grad(o, name).copy())
    return log, {k: tuple(v) for k, v in out.items()}


def _oracle_grad(o, name):
    p = dict(zip(['poses_T', 'poses_smpl', 'betas', 'zmin_lin', 'zmax_lin', 'xscale'], o.leaves()))[name]
    return p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)


@pytest.mark.parametrize('scene', [False, True])
def test_full_cycle_gradients_and_log(smpl_struct, smpl_regs, oracle_model, tmp_path, scene):
    T, N, W, H, batch = 4, 2, 120, 68, 2
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 31, scene)
    assert seq['seg_mask'].sum() > 50
    opt._stage_from_dataloader(dl)
    from mhhip.raster import RasterTerms
    e = opt.engine
    e.cycle(0, raster=RasterTerms(e))
    log = e.read_log(1)[0]
    want_log = o.cycle_grads(batches)
    for k in ['loss_pose24j', 'loss_depth', 'loss_silhouette', 'reg_ref_poses', 'reg_scale', 'reg_contact',
              'reg_foot_sliding', 'reg_vel']:
        np.testing.assert_allclose(log[k], want_log[k], rtol=3e-3, atol=1e-6, err_msg=k)
    assert want_log['loss_depth'] > 0 and want_log['loss_silhouette'] > 0
    if scene:
        assert want_log['reg_contact'] > 0
    for name, ename in LEAF_MAP:
        w = _oracle_grad(o, name)
        g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
        scale = max(np.abs(w).max(), 1e-8)
        err = np.abs(g - w)
        # rasteriser terms: float atomics and last-ulp selection flips -> tight on (almost) all entries
        frac = float((err > 5e-3 * scale).mean())
        assert frac < 0.01, '%s: %.4f of the entries above 5e-3*max (max err %.2e, scale %.2e)' % (name, frac, err.max(), scale)
        assert np.median(err) < 1e-3 * scale, name


def test_fit_three_cycles_all_terms(smpl_struct, smpl_regs, oracle_model, tmp_path):
    T, N, W, H, batch = 4, 2, 120, 68, 2
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 32, True)
    log = opt.fit(dl, num_iter=3)
    want = o.fit(batches, 3)
    for k in ['loss_pose24j', 'loss_depth', 'loss_silhouette', 'reg_contact']:
        np.testing.assert_allclose([l[k] for l in log], [l[k] for l in want], rtol=2e-2, atol=1e-6, err_msg=k)
    ov = opt.get_optimized_variables()
    wv = o.optimized_variables()
    # MPJPE-style check on the optimised translations / poses after 3 RMSprop steps
    np.testing.assert_allclose(ov['poses_T'], wv['poses_T'], atol=2e-3)
    np.testing.assert_allclose(ov['poses_smpl'], wv['poses_smpl'], atol=2e-3)
    np.testing.assert_allclose(ov['betas_smpl'], wv['betas_smpl'], atol=2e-3)
    np.testing.assert_allclose(ov['min_z'], wv['min_z'], atol=2e-3)
    np.testing.assert_allclose(ov['max_z'], wv['max_z'], atol=2e-3)


def test_graph_replay_matches_eager_across_filter_updates(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """captured cycle graphs read the filter buffers by address: a second filter update must reach the replays"""
    T, N, W, H, batch = 4, 2, 120, 68, 2
    from mhhip.raster import RasterTerms
    runs = []
    for graphs in (False, True):
        opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 31, True)
        opt._stage_from_dataloader(dl)
        e = opt.engine
        raster = RasterTerms(e)
        for c in range(6):
            if c in (1, 3):
                e.update_filters()
            if graphs:
                e.cycle_graphed(c, raster=raster)
                e.step_dev()
            else:
                e.cycle(c, raster=raster)
                e.step(0.01 * 0.99 ** c)
        torch.cuda.synchronize()
        runs.append((e.params.cpu().numpy().copy(), e.grads.cpu().numpy().copy(), e.read_log(6)))
    (p0, g0, l0), (p1, g1, l1) = runs
    np.testing.assert_allclose(p1, p0, atol=2e-4 * np.abs(p0).max())
    np.testing.assert_allclose(g1, g0, atol=2e-3 * np.abs(g0).max())
    for c in range(6):
        for k in l0[c]:
            np.testing.assert_allclose(l1[c][k], l0[c][k], rtol=2e-3, atol=1e-6, err_msg='%s cycle %d' % (k, c))
    assert l0[4]['reg_filter_verts'] > 0


def test_second_fit_on_the_same_optimizer_replays_valid_graphs(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """``fit`` twice on one optimiser (round-1 advisor finding): the rasteriser workspace the captured graphs point
    at belongs to the engine, so the second call replays against live memory -- same result as launching eagerly."""
    T, N, W, H, batch = 4, 2, 120, 68, 2
    runs = []
    for graphs in (False, True):
        opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 33, True)
        opt.use_graphs = graphs
        opt.fit(dl, num_iter=3)
        junk = [torch.empty(1 << 22, device='cuda:0').normal_() for _ in range(8)]      # churn the caching allocator
        del junk
        log = opt.fit(dl, num_iter=3)
        torch.cuda.synchronize()
        runs.append((opt.engine.params.cpu().numpy().copy(), log))
    (p0, l0), (p1, l1) = runs
    np.testing.assert_allclose(p1, p0, atol=2e-4 * np.abs(p0).max())
    for c in range(3):
        for k in ['loss_depth', 'loss_silhouette', 'loss_pose24j']:
            np.testing.assert_allclose(l1[c][k], l0[c][k], rtol=2e-3, atol=1e-6, err_msg='%s cycle %d' % (k, c))


def test_every_fit_starts_a_new_rmsprop(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """The reference builds a new torch.optim.RMSprop (and ExponentialLR) inside every ``fit`` (optimizer.py:355-356), so the
    first step of a SECOND fit is again lr * g / (sqrt((1 - alpha) g^2) + eps) = 0.01 * sign(g) / sqrt(0.5) for every entry
    with a gradient -- with the moments of the first fit kept it would be anything else.  (Found while listing where the
    staged loop could diverge silently from the reference; round 3.)"""
    T, N, W, H, batch = 4, 2, 96, 54, 2
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 37, False)
    opt.fit(dl, num_iter=4)
    e = opt.engine
    before = e.params.clone()
    opt.fit(dl, num_iter=1)
    torch.cuda.synchronize()
    step = (e.params - before).abs().cpu().numpy()
    g = e.grads.cpu().numpy()
    big = np.abs(g) > 1e-4 * np.abs(g).max()                      # entries whose sign is not rounding noise
    assert big.sum() > 100
    np.testing.assert_allclose(step[big], 0.01 / np.sqrt(0.5), rtol=2e-3)
    assert (step[g == 0] == 0).all()


def test_fit_with_another_dataloader_reads_the_new_inputs(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """The reference reads whatever dataloader it is handed, every cycle.  The staged loop reads its inputs once -- so a
    DIFFERENT dataloader object in a later ``fit`` must be staged afresh (new device tensors, captured cycles and the device
    scene state dropped), and the result must be that of an optimiser that had those inputs from the start."""
    from mhhip import synthetic_seq
    T, N, W, H, batch = 6, 2, 96, 54, 3
    opt, dl_a, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 43, True)
    opt.fit(dl_a, num_iter=3)
    seq_b = dict(seq)
    seq_b['depths'] = (1.0 - seq['depths']).astype(np.float32)
    seq_b['pose2d'] = seq['pose2d'].copy()
    seq_b['pose2d'][..., :2] += 3.0
    dl_b = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq_b), batch_size=batch, shuffle=False)
    leaves = opt.engine.params.clone()
    log1 = opt.fit(dl_b, num_iter=2)
    np.testing.assert_array_equal(opt.engine.depths.cpu().numpy().reshape(T, H, W), seq_b['depths'])
    np.testing.assert_array_equal(opt.engine.pose2d.cpu().numpy().reshape(T, N, 17, 3), seq_b['pose2d'])
    # the same leaves on an optimiser that only ever saw the second loader
    sub = tmp_path / 'b'
    sub.mkdir()
    opt2, _, _, _, _ = _setup(smpl_struct, smpl_regs, oracle_model, sub, T, N, W, H, batch, 43, True)
    opt2._stage_from_dataloader(dl_b)
    opt2.engine.params.copy_(leaves)
    log2 = opt2.fit(dl_b, num_iter=2)
    for c in range(2):
        for k in ['loss_pose24j', 'loss_depth', 'loss_silhouette', 'reg_contact']:
            np.testing.assert_allclose(log1[c][k], log2[c][k], rtol=1e-4, atol=1e-7, err_msg='%s cycle %d' % (k, c))
    p1, p2 = opt.engine.params.cpu().numpy(), opt2.engine.params.cpu().numpy()
    assert float((np.abs(p1 - p2) > 1e-4).mean()) < 0.01          # float atomics + sign-like first steps on noise-level entries
    # and the same loader object again is NOT staged again
    d_ptr = opt.engine.depths.data_ptr()
    opt.fit(dl_b, num_iter=1)
    assert opt.engine.depths.data_ptr() == d_ptr


def test_eight_cycles_step_by_step(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """Eight cycles of the whole loop, compared cycle by cycle instead of only at the end: a free-running comparison of two
    fp32 implementations of this optimiser diverges by construction (RMSprop's first steps are lr * sign(g) / sqrt(1 - alpha):
    an entry whose gradient is below the rounding noise takes the other sign and is off by 0.028 after ONE step), so the
    end-state tests can only be statistical.  Here every cycle starts from the SAME state on both sides (the engine's leaves
    and RMSprop moments are copied to the oracle), the oracle renders the faces the kernel selected at the vertices the
    kernel produced, and then: every entry of every leaf gradient within 2e-4 of the largest, and every entry of every
    leaf after the step within 1e-5 wherever the gradient is at least 1e-3 of the leaf's largest (above the noise) --
    deterministic scatter, no percentiles."""
    from mhhip import synthetic
    from mhhip.raster import RasterTerms, set_deterministic
    T, N, W, H, batch = 6, 2, 120, 68, 3
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 35, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    raster = RasterTerms(e)
    hsel = _HipSelectionRasteriser(np.asarray(smpl_struct.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N)
    o.rasteriser = hsel
    names = [ename for _, ename in LEAF_MAP]
    leaves = o.leaves()
    sq = [torch.zeros_like(p) for p in leaves]
    buf = [torch.zeros_like(p) for p in leaves]
    old = set_deterministic(True)
    try:
        lr = 0.01
        for c in range(8):
            with torch.no_grad():            # same state on both sides
                for p_, s_, b_, ename in zip(leaves, sq, buf, names):
                    p_.copy_(e.leaf(ename).cpu().view(p_.shape))
                    s_.copy_(e.leaf(ename, e.sq).cpu().view(p_.shape))
                    b_.copy_(e.leaf(ename, e.buf).cpu().view(p_.shape))
            e.cycle(c, raster=raster)
            hsel.take(raster, e, oracle=o)
            o.cycle_grads(batches)
            grads = []
            for (name, ename), p_ in zip(LEAF_MAP, leaves):
                w = _oracle_grad(o, name)
                g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
                scale = max(np.abs(w).max(), 1e-8)
                np.testing.assert_allclose(g, w, atol=2e-4 * scale, rtol=0, err_msg='cycle %d grad %s' % (c, name))
                grads.append((w, scale))
            e.step(lr)
            with torch.no_grad():
                for p_, s_, b_ in zip(leaves, sq, buf):
                    fo.rmsprop_step(p_, p_.grad if p_.grad is not None else torch.zeros_like(p_), s_, b_, lr)
            lr *= 0.99
            for (name, ename), p_, (w, scale) in zip(LEAF_MAP, leaves, grads):
                got = e.leaf(ename).cpu().numpy().reshape(w.shape)
                want = p_.detach().numpy()
                sure = np.abs(w) >= 1e-3 * scale
                assert sure.mean() > 0.3 or name in ('poses_smpl',), (name, float(sure.mean()))
                np.testing.assert_allclose(got[sure], want[sure], atol=1e-5, rtol=0, err_msg='cycle %d leaf %s' % (c, name))
                assert np.abs(got - want).max() <= 0.03, name           # nowhere more than one sign-like step (2 lr / sqrt(.5))
    finally:
        set_deterministic(old)
