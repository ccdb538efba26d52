"""Scene aggregation on the device (row a24/f1) against the numpy restatement of the reference's host pipeline
(oracle/scene_oracle.py; its median is pinned to the reference through tests/golden): masked median over time,
bilateral + Sobel edge mask + erode + iterative median fill, compaction/un-projection."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from mhhip import _lib
    return _lib, _lib.lib()


def _ranges(zmin, zmax):
    min_z = np.log(1.0 + np.exp(zmin.astype(np.float32))).astype(np.float32)
    max_z = (min_z + np.float32(1.0) + np.log(1.0 + np.exp(zmax.astype(np.float32))).astype(np.float32)).astype(np.float32)
    return (np.float32(1.0) / min_z).astype(np.float32), (np.float32(1.0) / max_z).astype(np.float32)


@pytest.mark.parametrize('T,H,W', [(23, 12, 20), (200, 9, 70), (8, 16, 16)])
def test_masked_median_over_time(T, H, W):
    from oracle import scene_oracle as scene_host
    _l, L = _lib()
    rng = np.random.RandomState(T)
    dn = rng.uniform(0, 1, (T, H, W)).astype(np.float32)
    dn[:, :2] = np.round(dn[:, :2] * 4) / 4                          # ties
    back = (rng.uniform(0, 1, (T, H, W)) > 0.4).astype(np.uint8)
    back[:, 0, 0] = 0                                                # a pixel no frame sees
    back[1:, 0, 1] = 0                                               # a pixel one frame sees
    back[2:, 0, 2] = 0                                               # two frames
    zmin = rng.uniform(0.5, 1.5, T).astype(np.float32)
    zmax = rng.uniform(4, 9, T).astype(np.float32)
    inv_min, inv_max = _ranges(zmin, zmax)
    disp = (dn * (inv_min - inv_max)[:, None, None] + inv_max[:, None, None]).astype(np.float32)
    depths = (np.float32(1.0) / disp).astype(np.float32)
    _, want, want_mask = scene_host.aggregate_scene_median(depths, None, back)
    dev = torch.device('cuda:0')
    t = lambda a: torch.tensor(a, device=dev)
    ws = torch.empty(L.mh_scene_workspace_bytes(T, H, W), dtype=torch.uint8, device=dev)
    md, mm = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    tdn, tback, tzmin, tzmax = t(dn), t(back), t(zmin), t(zmax)        # keep the buffers alive across the call
    _l.check(L.mh_scene_median(T, H, W, _l.ptr(tdn), _l.ptr(tback), _l.ptr(tzmin), _l.ptr(tzmax), _l.ptr(md), _l.ptr(mm),
                               _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(mm.cpu().numpy() > 0.5, want_mask)
    got = md.cpu().numpy()
    np.testing.assert_allclose(got[want_mask], want[want_mask], rtol=3e-6)
    assert (got[~want_mask] == 0).all()
    # pixel-major form (one wave per pixel, ballots)
    md2, mm2 = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    tdn_t, tback_t = tdn.view(T, H * W).t().contiguous(), tback.view(T, H * W).t().contiguous()
    _l.check(L.mh_scene_median_t(T, H, W, _l.ptr(tdn_t), _l.ptr(tback_t), _l.ptr(tzmin), _l.ptr(tzmax), _l.ptr(md2), _l.ptr(mm2),
                                 _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(mm2.cpu().numpy(), mm.cpu().numpy())
    np.testing.assert_array_equal(md2.cpu().numpy(), got)


def _scene(H, W, seed):
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = 3.0 + 0.03 * ys + 0.5 * np.sin(xs / 9.0) + 0.01 * rng.randn(H, W)
    depth[H // 3:H // 2, W // 4:W // 3] -= 1.2                        # a box in front: depth edges
    mask = np.ones((H, W), np.float32)
    mask[H // 2:H // 2 + 14, W // 2:W // 2 + 9] = 0                   # a person-sized hole
    mask[2:5, 3:6] = 0
    depth = depth.astype(np.float32)
    depth[mask == 0] = 0.0
    return depth, mask


@pytest.mark.parametrize('H,W,bilateral', [(68, 120, 1), (45, 37, 1), (68, 120, 0)])
def test_postprocess_depthmap(H, W, bilateral):
    from oracle import scene_oracle as scene_host
    _l, L = _lib()
    depth, mask = _scene(H, W, H + W)
    want = scene_host.postprocess_depthmap(depth, mask, use_bilateral_filter=bool(bilateral))
    dev = torch.device('cuda:0')
    ws = torch.empty(L.mh_scene_workspace_bytes(1, H, W), dtype=torch.uint8, device=dev)
    out = torch.empty(H, W, device=dev)
    tdepth, tmask = torch.tensor(depth, device=dev), torch.tensor(mask, device=dev)
    _l.check(L.mh_scene_postprocess(H, W, _l.ptr(tdepth), _l.ptr(tmask), bilateral, 7, _l.ptr(out), _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all() and (got > 0).all()
    bad = np.abs(got - want) > 1e-4 * np.maximum(1.0, np.abs(want))
    # the edge threshold (3 x mean of float32 statistics) may flip a borderline pixel; everything else must agree
    assert bad.mean() < 0.005, '%.4f of the pixels differ (%d)' % (float(bad.mean()), int(bad.sum()))


def test_points_are_compacted_in_row_major_order():
    from mhhip import synthetic
    _l, L = _lib()
    H, W = 27, 41
    rng = np.random.RandomState(0)
    depth = rng.uniform(2, 9, (H, W)).astype(np.float32)
    mask = (rng.uniform(0, 1, (H, W)) > 0.3).astype(np.float32)
    K = synthetic.default_cam_K((W, H), 60.0)
    dev = torch.device('cuda:0')
    pts = torch.zeros(H * W, 3, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    Kc = np.ascontiguousarray(K.reshape(9))
    tdepth, tmask = torch.tensor(depth, device=dev), torch.tensor(mask, device=dev)
    _l.check(L.mh_scene_points(H, W, Kc.ctypes.data_as(_l.c_float_p), _l.ptr(tdepth), _l.ptr(tmask), _l.ptr(pts), _l.ptr(cnt),
                               _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    n = int(cnt.item())
    assert n == int(mask.sum())
    u = (np.arange(W, dtype=np.float32) + 0.5 - K[0, 2]) / K[0, 0]
    v = (np.arange(H, dtype=np.float32) + 0.5 - K[1, 2]) / K[1, 1]
    want = np.stack([depth * u[None, :], depth * v[:, None], depth], -1)[mask > 0.5]
    np.testing.assert_allclose(pts[:n].cpu().numpy(), want, rtol=2e-6, atol=1e-6)


def _fit(smpl_struct, smpl_regs, tmp_path, mode, num_iter):
    """the organic path of `fit`: scene built from the sequence itself once cycle 30 is reached"""
    from mhhip import synthetic, synthetic_seq
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    import golden_inputs as gi
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    T, N, W, H, batch = 8, 2, 120, 68, 4
    c = gi.COEFS
    K = synthetic.default_cam_K((W, H), 60.0)
    opt = SMPLDepthSequenceOptimizer(
        image_size=(W, H), num_frames=T, fov=60, device='cuda:0', smpl_model_parameters_path=str(tmp_path),
        smpl_data_struct=smpl_struct, scene_update=mode, cam_K=K,
        proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
        reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
        reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, N, T, (W, H), 77, cam_K=K, z_range=(2.6, 3.6))
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=20)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=batch, shuffle=False)
    log = opt.fit(dl, num_iter=num_iter)
    return opt, log


def test_fit_builds_the_scene_on_the_device(smpl_struct, smpl_regs, tmp_path):
    """same optimiser state -> the device pipeline must reproduce the host one (numpy, as the reference does it); then
    the organic path of `fit` (first scene update at cycle 30) runs with the contact term live"""
    from oracle import scene_oracle as scene_host
    opt, _ = _fit(smpl_struct, smpl_regs, tmp_path, 'none', 3)
    e = opt.engine
    depths = scene_host.target_depths(e)
    _, ma_depth, ma_mask = scene_host.aggregate_scene_median(depths, None, opt._backmasks)
    want = scene_host.postprocess_depthmap(ma_depth, ma_mask, use_bilateral_filter=True)
    e.scene_device_setup(opt._backmasks)
    e.scene_device_update()
    e.scene_device_swap()
    got, got_mask, pts = e.scene_device_result()
    np.testing.assert_array_equal(got_mask, ma_mask)
    bad = np.abs(got - want) > 1e-5 * np.maximum(1.0, np.abs(want))
    assert bad.mean() < 0.005, '%.4f of the pixels differ (%d)' % (float(bad.mean()), int(bad.sum()))
    assert pts.shape[0] == int(ma_mask.sum())

    opt, log = _fit(smpl_struct, smpl_regs, tmp_path, 'device', 36)
    assert all(np.isfinite(float(v)) for row in log for v in row.values())
    assert log[29]['reg_contact'] == 0 and log[33]['reg_contact'] > 0
    assert opt.scene_depth.shape == (68, 120) and opt.scene_pcd.shape[2] > 1000 and opt.scene_img is not None


def test_scene_image_median_and_fill_on_the_device(smpl_struct, smpl_regs, tmp_path):
    """optimizer.py:595-600: colour median over time (background pixels only) + looped 11x11 median fill of the holes"""
    from oracle import scene_oracle as scene_host
    opt, _ = _fit(smpl_struct, smpl_regs, tmp_path, 'none', 1)
    e = opt.engine
    T, H, W = e.T, e.H, e.W
    rng = np.random.RandomState(4)
    images = rng.randint(0, 256, (T, H, W, 3)).astype(np.uint8)
    back = (rng.uniform(0, 1, (T, H, W)) > 0.5).astype(np.int64)
    back[:, 20:44, 30:52] = 0                                         # a region no frame sees: filled in
    back[:, 0:3, 0:3] = 0
    want_img, _, _ = scene_host.aggregate_scene_median(None, images, back, images_only=True)
    want_mask = (back.max(axis=0) > 0).astype(np.float32)
    while want_mask.min() == 0:
        want_img, want_mask = scene_host.fillin_values(want_img, want_mask, filter_size=11)
    e.scene_device_setup(back)
    got_img, got_mask = e.scene_device_image(images)
    assert got_img.dtype == np.uint8 and got_img.shape == (H, W, 3)
    np.testing.assert_array_equal(got_mask, np.ones((H, W), np.float32))
    np.testing.assert_array_equal(got_img, want_img)


# ---- known answers (tests/test_scene_oracle.py pins the numpy checker with the same cases) -----------------------------------
def _known_cases():
    import test_scene_oracle as tso
    return tso.scene_known_answer_cases()


@pytest.mark.parametrize('case', _known_cases(), ids=lambda c: c[0])
def test_postprocess_known_answers_on_the_device(case):
    """constant image with a hole / clean depth step / step on a ramp: outputs that follow from the definitions of the
    bilateral filter, Sobel, two 3x3 erosions and the looped 7x7 median fill (utils.py:174-209, 91-135) -- no checker"""
    _l, L = _lib()
    _, depth, mask, bil, want, rtol = case
    H, W = depth.shape
    dev = torch.device('cuda:0')
    ws = torch.empty(L.mh_scene_workspace_bytes(1, H, W), dtype=torch.uint8, device=dev)
    out = torch.empty(H, W, device=dev)
    tdepth, tmask = torch.tensor(depth, device=dev), torch.tensor(mask, device=dev)
    _l.check(L.mh_scene_postprocess(H, W, _l.ptr(tdepth), _l.ptr(tmask), int(bil), 7, _l.ptr(out), _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=max(rtol, 2e-6))


def test_fill_sweeps_known_answer_on_the_device():
    """value = column index with nine masked columns: medians of the valid window values, the second sweep reading the
    first sweep's results (and never a value written in the same sweep)"""
    import test_scene_oracle as tso
    _l, L = _lib()
    x, mask, filled = tso.column_hole_case()
    H, W = x.shape
    dev = torch.device('cuda:0')
    ws = torch.empty(L.mh_scene_workspace_bytes(1, H, W), dtype=torch.uint8, device=dev)
    tv, tm = torch.tensor(x, device=dev), torch.tensor(mask, device=dev)
    _l.check(L.mh_scene_fill(H, W, 7, 0, _l.ptr(tv), _l.ptr(tm), _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    got = tv.cpu().numpy()
    np.testing.assert_array_equal(tm.cpu().numpy(), np.ones((H, W), np.float32))
    np.testing.assert_allclose(got[:, 10:19], filled[None, :].repeat(H, 0), rtol=1e-6)
    np.testing.assert_array_equal(got[:, :10], x[:, :10])
    np.testing.assert_array_equal(got[:, 19:], x[:, 19:])


def test_raw_value_median_frame_major_equals_pixel_major():
    """colour planes of sequences longer than the register form's 2048 frames go through mh_scene_median with
    zmin = zmax = NULL (raw values); same result as the pixel-major form and as numpy"""
    _l, L = _lib()
    T, H, W = 37, 10, 23
    rng = np.random.RandomState(3)
    vals = rng.randint(0, 256, (T, H, W)).astype(np.float32)
    back = (rng.uniform(0, 1, (T, H, W)) > 0.5).astype(np.uint8)
    back[:, 0, 0] = 0
    dev = torch.device('cuda:0')
    ws = torch.empty(L.mh_scene_workspace_bytes(T, H, W), dtype=torch.uint8, device=dev)
    tv, tb = torch.tensor(vals, device=dev), torch.tensor(back, device=dev)
    a, am = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    b, bm = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    _l.check(L.mh_scene_median(T, H, W, _l.ptr(tv), _l.ptr(tb), None, None, _l.ptr(a), _l.ptr(am), _l.ptr(ws), _l.stream_ptr(dev)))
    tvt, tbt = tv.view(T, H * W).t().contiguous(), tb.view(T, H * W).t().contiguous()
    _l.check(L.mh_scene_median_t(T, H, W, _l.ptr(tvt), _l.ptr(tbt), None, None, _l.ptr(b), _l.ptr(bm), _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_array_equal(am.cpu().numpy(), bm.cpu().numpy())
    want = np.ma.median(np.ma.array(vals, mask=back == 0), axis=0)
    seen = back.max(axis=0) > 0
    np.testing.assert_allclose(a.cpu().numpy()[seen], want.data[seen], rtol=1e-6)
    assert not seen[0, 0] and float(am[0, 0]) == 0.0
