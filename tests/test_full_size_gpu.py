"""Parity at BASELINE.json's full sizes.

* C3 (the configuration the metric is quoted on): 4 humans x 200 frames at 240x135, full loss stack with a
  scene cloud -- one complete cycle of the drop-in optimiser against the CPU oracle (loss log and the gradient
  of every leaf), then the one-euro filters of the whole sequence and a second cycle with the filtered-vertex
  term switched on.  The oracle needs ~10 s per cycle at this size.
* C5 (contact term dominant): a 200 000-point scene cloud -- the grid search against the brute-force scan on
  every lowest-vertex query of an 8 humans x 500 frames batch (4000 queries), and against a float64 numpy
  search on a sample of them; then a cycle at the C5 shape whose per-frame gradients of the first batch must
  equal the oracle's on that batch alone (the per-frame terms do not couple frames once the temporal
  coefficients are zero), a size-independent restriction property."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import raster_oracle as ro
from test_fit_full_gpu import LEAF_MAP, _HipSelectionRasteriser, _oracle_grad, _setup
from test_scene_knn_gpu import _dy_reference, _run

pytestmark = pytest.mark.gpu

LOG_KEYS = ['loss_pose24j', 'loss_depth', 'loss_silhouette', 'reg_ref_poses', 'reg_scale', 'reg_contact',
            'reg_foot_sliding', 'reg_vel']


def _compare_grads(e, o, frac=0.01, tight=5e-3, med=1e-3):
    for name, ename in LEAF_MAP:
        w = _oracle_grad(o, name)
        g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
        assert np.isfinite(g).all(), name
        scale = max(np.abs(w).max(), 1e-8)
        err = np.abs(g - w)
        off = float((err > tight * scale).mean())
        assert off < frac, '%s: %.4f of the entries above %g*max (max err %.2e, scale %.2e)' % (name, off, tight, err.max(), scale)
        assert np.median(err) < med * scale, name


def test_c3_full_size_cycle_matches_oracle(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """BASELINE C3 exactly as the bench runs it: 4 humans x 200 frames at 240x135 in batches of TEN (the batch size fixes
    the in-batch foot-sliding pairs and the per-batch regularisers), deterministic gradient scatter, the oracle
    rendering the faces the kernel selected at the vertices the kernel produced: loss log and EVERY entry of EVERY
    leaf gradient (no percentiles, no outlier allowance)."""
    from mhhip import synthetic
    from mhhip.raster import RasterTerms, set_deterministic
    T, N, W, H, batch = 200, 4, 240, 135, 10
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 41, True)
    assert len(batches) == 20
    opt._stage_from_dataloader(dl)
    e = opt.engine
    raster = RasterTerms(e)
    hsel = _HipSelectionRasteriser(np.asarray(smpl_struct.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N)
    o.rasteriser = hsel
    old = set_deterministic(True)
    try:
        e.cycle(0, raster=raster)
        hsel.take(raster, e, oracle=o)
        log = e.read_log(1)[0]
        want = o.cycle_grads(batches)
        for k in LOG_KEYS:
            np.testing.assert_allclose(log[k], want[k], rtol=1e-3, atol=1e-6, err_msg=k)
        for k in ['loss_pose24j', 'loss_depth', 'loss_silhouette', 'reg_contact', 'reg_foot_sliding', 'reg_vel']:
            assert want[k] > 0, k
        _compare_grads_everywhere(e, o)
        # filters over the 200-frame sequence (optimizer.py:664-675), then the filtered-vertex term (:571-573)
        e.update_filters()
        o.update_filters()
        np.testing.assert_allclose(e.verts_filt.cpu().numpy().reshape(T, -1), o.v_filt.numpy().reshape(T, -1), atol=2e-5)
        np.testing.assert_allclose(e.pT_filt.cpu().numpy().reshape(T, -1), o.pT_filt.numpy().reshape(T, -1), atol=2e-6)
        e.cycle(1, raster=raster)
        hsel.take(raster, e, oracle=o)
        log = e.read_log(2)[1]
        want = o.cycle_grads(batches)
        for k in LOG_KEYS + ['reg_filter_verts']:
            np.testing.assert_allclose(log[k], want[k], rtol=1e-3, atol=1e-6, err_msg=k)
        assert want['reg_filter_verts'] >= 0
        _compare_grads_everywhere(e, o)
    finally:
        set_deterministic(old)


def test_c3_full_size_cycle_against_the_oracles_own_selection(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """END TO END at C3: the same cycle, but the oracle selects its faces ITSELF (fp32 brute force over
    every face and pixel, ``oracle/raster_select.c``) instead of rendering the kernel's selection -- nothing of the HIP path
    enters the expected values.  The two selections differ on ~0.9 % of the live (pixel, pass) entries, every one a float64
    near-tie (the test below), and float atomics order the sums: the gate is the one of the small end-to-end test
    (``test_fit_full_gpu.py::test_full_cycle_gradients_and_log``) -- under 1 % of a leaf's entries further than 5e-3 of its
    largest entry from the oracle's, median under 1e-3 -- and the measured figures are printed."""
    from mhhip.raster import RasterTerms
    T, N, W, H, batch = 200, 4, 240, 135, 10
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 41, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    e.cycle(0, raster=RasterTerms(e))
    log = e.read_log(1)[0]
    want = o.cycle_grads(batches)
    for k in LOG_KEYS:
        np.testing.assert_allclose(log[k], want[k], rtol=3e-3, atol=1e-6, err_msg=k)
    for name, ename in LEAF_MAP:
        w = _oracle_grad(o, name)
        g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
        scale = max(np.abs(w).max(), 1e-8)
        err = np.abs(g - w)
        print('%-10s own selection: max %.2e  p99 %.2e  median %.2e (x largest entry), above 5e-3: %.5f of the entries'
              % (name, err.max() / scale, np.percentile(err, 99) / scale, np.median(err) / scale, float((err > 5e-3 * scale).mean())))
    _compare_grads(e, o)


def test_c3_selection_against_brute_force_on_every_body(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """The C3 cycle above feeds the kernel's own face selection into the oracle; the selection itself is held here against the
    oracle's brute-force selection on ALL 800 bodies of the C3 launch (round 4 sampled 40; the C selection of 800 bodies is
    ~2 s of host time), both passes, every window pixel -- a pixel whose faces differ must be a float64 near-tie
    (tests/test_raster_gpu.py::_selection_differences: depth within 1e-5 of the cut of the list, or distance within 1e-4 of
    the blur radius)."""
    import torch
    from mhhip import synthetic
    from mhhip.raster import RasterTerms
    from test_raster_gpu import _selection_differences
    T, N, W, H, batch = 200, 4, 240, 135, 10
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 41, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    raster = RasterTerms(e)
    e.cycle(0, raster=raster)
    torch.cuda.synchronize()
    win, koff, keys = raster.selection(e)
    faces = np.asarray(smpl_struct.f).astype(np.int64)
    K = synthetic.default_cam_K((W, H), 60.0)
    ndiff = live = 0
    not_ties = []
    for b0 in range(0, e.B, 100):              # (the dense (B, H, W, K) arrays of the brute force: 100 bodies at a time)
        idx = np.arange(b0, min(b0 + 100, e.B))
        sub_koff = np.concatenate([[0], np.cumsum(koff[idx + 1] - koff[idx])])
        sub_keys = keys[koff[idx[0]]:koff[idx[-1] + 1]]
        r = dict(shape=(len(idx), 1, H, W), faces=faces, K=K, verts=e.verts[idx[0]:idx[-1] + 1].cpu().numpy(), sel=(win[idx], sub_koff, sub_keys))
        nd, lv, nt = _selection_differences(r)
        ndiff += nd; live += lv
        not_ties += [(b0 + t[0],) + tuple(t[1:]) for t in nt]
    print('C3, all %d bodies: selection differs on %d of %d live (pixel, pass) entries; not explained as near-ties: %d'
          % (e.B, ndiff, live, len(not_ties)))
    # measured (round 6, this launch): 15 481 of 1 668 098 = 0.93 %; the gate is 1.5x that -- a regression from 1 % to 3 % used to pass
    assert live > 400000 and ndiff <= 1.4e-2 * live, (ndiff, live)
    assert not not_ties, not_ties[:5]


def _compare_grads_everywhere(e, o, tol=2e-4):       # measured worst entry at C3: 3.1e-5 of the leaf's largest
    for name, ename in LEAF_MAP:
        w = _oracle_grad(o, name)
        g = e.leaf(ename, e.grads).cpu().numpy().reshape(w.shape)
        scale = max(np.abs(w).max(), 1e-8)
        err = np.abs(g - w)
        print('%-10s max %.2e  p99 %.2e  median %.2e (x largest entry)' % (name, err.max() / scale, np.percentile(err, 99) / scale, np.median(err) / scale))
        np.testing.assert_allclose(g, w, atol=tol * scale, rtol=0, err_msg=name)


def test_c5_cloud_of_200k_points():
    rng = np.random.RandomState(11)
    M = 200000
    # ground sheet with relief + furniture-like boxes, 8 humans x 500 frames of lowest-vertex queries above it
    pts = np.stack([rng.uniform(-6, 6, M), 1.2 + 0.03 * rng.randn(M) + 0.1 * np.sin(rng.uniform(0, 6, M)),
                    rng.uniform(2, 14, M)], 1).astype(np.float32)
    box = rng.rand(M) < 0.15
    pts[box, 1] -= rng.uniform(0.2, 0.9, box.sum()).astype(np.float32)
    B = 8 * 500
    q = np.stack([rng.uniform(-5, 5, B), 1.2 + rng.uniform(-0.3, 0.1, B), rng.uniform(2.5, 13, B)], 1).astype(np.float32)
    brute, grid = _run(pts, q)
    np.testing.assert_allclose(grid, brute, atol=1e-6, rtol=0)      # same 32 points (summation order differs)
    sel = rng.choice(B, 96, replace=False)
    np.testing.assert_allclose(grid[sel], _dy_reference(pts, q[sel], 32), atol=2e-5, rtol=1e-5)


def test_c5_shape_cycle_restricts_to_the_first_batch(smpl_struct, smpl_regs, oracle_model, tmp_path, monkeypatch):
    """8 humans x 500 frames with the 200 000-point cloud INSIDE the cycle (set_scene_points -> grid -> contact term)"""
    T, N, W, H, batch = 500, 8, 240, 135, 25
    coefs = dict(gi.COEFS)
    coefs.update(reg_velocity=0.0, reg_verts_filter=0.0, reg_foot_sliding=0.0)
    monkeypatch.setattr(gi, 'COEFS', coefs)
    opt, dl, o_full, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 43, True)
    opt._stage_from_dataloader(dl)
    from mhhip.raster import RasterTerms
    e = opt.engine
    # BASELINE C5's scene: 200 000 points injected directly (the reference ties the cloud to the image, optimizer.py:609-613;
    # 600x338 px is the equivalent image) -- the same relief + boxes as the stand-alone neighbour-search test above
    rng = np.random.RandomState(11)
    M = 200000
    pts = np.stack([rng.uniform(-6, 6, M), 1.2 + 0.03 * rng.randn(M) + 0.1 * np.sin(rng.uniform(0, 6, M)),
                    rng.uniform(2, 14, M)], 1).astype(np.float32)
    box = rng.rand(M) < 0.15
    pts[box, 1] -= rng.uniform(0.2, 0.9, box.sum()).astype(np.float32)
    e.set_scene_points(torch.tensor(pts))
    o_full.scene_pcd = torch.tensor(pts).view(1, 1, M, 3)
    from mhhip import synthetic
    from mhhip.raster import set_deterministic
    raster = RasterTerms(e)
    hsel = _HipSelectionRasteriser(np.asarray(smpl_struct.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N)
    o_full.rasteriser = hsel
    old = set_deterministic(True)
    try:
        e.cycle(0, raster=raster)
    finally:
        set_deterministic(old)
    hsel.take(raster, e, oracle=o_full)
    log = e.read_log(1)[0]
    g_all = {ename: e.leaf(ename, e.grads).cpu().numpy() for _, ename in LEAF_MAP}
    for v in g_all.values():
        assert np.isfinite(v).all()
    for k in LOG_KEYS:
        assert np.isfinite(log[k]), k
    assert log['reg_contact'] > 0 and log['loss_depth'] > 0 and log['loss_silhouette'] > 0
    # the oracle on the first batch alone (same leaves, same scene): gradients of the per-frame leaves of these
    # frames, and the first batch's share of the shared-shape gradient
    want = o_full.cycle_grads(batches[:1])
    for name, ename in [('poses_T', 'poses_T'), ('poses_smpl', 'poses_smpl'), ('zmin_lin', 'zmin_lin'),
                        ('zmax_lin', 'zmax_lin')]:
        w = _oracle_grad(o_full, name)
        g = g_all[ename].reshape(w.shape)
        w, g = w[:batch], g[:batch]
        scale = max(np.abs(w).max(), 1e-8)
        err = np.abs(g - w)
        print('%-10s max %.2e  median %.2e (x largest entry)' % (name, err.max() / scale, np.median(err) / scale))
        np.testing.assert_allclose(g, w, atol=2e-4 * scale, rtol=0, err_msg=name)        # EVERY entry of the first batch (measured 6.6e-6)
        assert np.abs(w).max() > 0, name
    # the contact term of the first batch against the oracle's full argsort over the 200 000 points
    np.testing.assert_allclose(float(e.batch_contact[0]), want['reg_contact'], rtol=1e-4)


def test_selection_does_not_depend_on_what_the_workspace_held(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """The tile schedule (longest first) is estimated beside the face sort from whatever the previous cycle left in the
    workspace (kept face lists, cost classes); a fresh workspace holds anything beside its cleared control words.  ~7000
    tiles at C3: the per-body depth sums, which come out of the selection in fixed order, must be bit-identical whether
    the workspace held zeros, random bits or the previous cycle's tables."""
    T, N, W, H, batch = 200, 4, 240, 135, 50
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 41, True)
    opt._stage_from_dataloader(dl)
    from mhhip.raster import RasterTerms
    e = opt.engine
    raster = RasterTerms(e)
    outs = []
    for fill in ('zeros', 'random', 'previous', 'random'):
        if fill == 'zeros':
            raster.ws.zero_()
        elif fill == 'random':
            raster.ws.copy_(torch.randint(0, 256, raster.ws.shape, dtype=torch.uint8, device=raster.ws.device))
        if fill != 'previous':
            raster.init_workspace()        # the contract: control words cleared once per (overwritten) workspace
        e.cycle(0, raster=raster)
        torch.cuda.synchronize()
        outs.append((e.depth_body.clone(), e.sil_body.clone(), e.leaf('poses_T', e.grads).clone()))
    for d, s_, g in outs[1:]:
        assert torch.equal(d, outs[0][0])
        torch.testing.assert_close(s_, outs[0][1], rtol=1e-5, atol=1e-7)
        scale = float(outs[0][2].abs().max())
        assert float((g - outs[0][2]).abs().max()) <= 2e-3 * scale        # float atomics in the gradient scatter


@pytest.mark.parametrize('T,N,W,H,batch', [(100, 4, 240, 135, 10), (12, 3, 96, 54, 3), (10, 2, 48, 80, 5), (8, 2, 64, 36, 4)])
def test_kept_face_lists_give_the_same_selection_as_a_fresh_sort(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch):
    """Temporal coherence of the rasteriser's preparation (mh_raster_set_sort_margin, default 1 row): over 40 optimisation
    cycles at the C3 shape -- RMSprop's first, largest steps included -- the selection keys (40 B per window pixel: z and
    face of the nearest face and of the K=4 list) of the launch that KEEPS its face lists must equal, bit for bit, those
    of a launch that sorts afresh on a second workspace; the lists must actually be kept (most bodies, most cycles) and
    must be rebuilt when a body moves by more than the margin."""
    from mhhip.raster import RasterTerms, set_sort_margin
    # (images under 100 rows: the blur band is narrower than a pixel there, and a face whose band holds no pixel-centre row
    # at sort time can hold one a cycle later -- the kept lists of round 3's first version had lost those faces)
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 47, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    kept, fresh = RasterTerms(e), RasterTerms(e)
    kept.ws.copy_(torch.randint(0, 256, kept.ws.shape, dtype=torch.uint8, device=kept.ws.device))     # a workspace holds anything
    kept.init_workspace()
    gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
    old = set_sort_margin(1)
    try:
        lr, seen0 = 0.01, None
        for c in range(40):
            e.cycle(c, raster=kept)                              # full cycle on the kept lists (selection + gradients)
            torch.cuda.synchronize()
            _, _, k1 = kept.selection(e)
            set_sort_margin(0)
            fresh(e, gv, log, phases=1)                          # same vertices, lists sorted now
            torch.cuda.synchronize()
            _, _, k0 = fresh.selection(e)
            set_sort_margin(1)
            assert k1.shape == k0.shape and (k1 == k0).all(), 'cycle %d: %d window pixels differ' % (c, int((k1 != k0).any(axis=1).sum()))
            e.step(lr)
            lr *= 0.99
            if c == 19:                                          # a jump of a few pixels for every fourth frame
                e.leaf('poses_T')[::4, :, 1] += 0.08
        seen, rebuilt, deferred = kept.sort_counters3(e)
    finally:
        set_sort_margin(old)
    assert seen == 40 * e.B
    print('face lists rebuilt for %d of %d (body, cycle) pairs, %d of them beside the gradient kernel' % (rebuilt, seen, deferred))
    # everything once, the jump, the early large steps -- not every cycle (round 6: the sorts that ride beside the gradient kernel,
    # tests/test_deferred_sort_gpu.py, are not the chain's)
    assert e.B <= rebuilt - deferred < (0.6 if H >= 100 else 0.9) * seen
