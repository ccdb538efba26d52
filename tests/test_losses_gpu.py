"""GPU parity of the drop-in mhmocap.losses / mhmocap.morphology against the reference-generated
fixtures (make_golden.py (iv), (v)) -- values and gradients."""
import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.tensor(np.asarray(a, np.float32), device='cuda:0')


def test_avg_depth_loss_matches_reference(golden):
    from mhmocap.losses import build_avg_depth_loss_fn
    pred, true, mask = gi.image_loss_inputs()
    tp, tt = dev(pred).requires_grad_(True), dev(true).requires_grad_(True)
    l = build_avg_depth_loss_fn()(tp, tt, dev(mask))
    l.backward()
    np.testing.assert_allclose(float(l), golden['depth_loss'], rtol=2e-5)
    np.testing.assert_allclose(tp.grad.cpu().numpy(), golden['depth_loss_gpred'], atol=1e-7, rtol=2e-4)
    np.testing.assert_allclose(tt.grad.cpu().numpy(), golden['depth_loss_gtrue'], atol=1e-7, rtol=2e-4)


def test_masked_mse_matches_reference(golden):
    from mhmocap.losses import build_masked_mse_loss_fn
    pred, true, mask = gi.image_loss_inputs()
    a = dev(pred[:, 0]).requires_grad_(True)
    l = build_masked_mse_loss_fn()(a, dev(true[:, 0]), dev(mask[:, 0]))
    l.backward()
    np.testing.assert_allclose(float(l), golden['mse_loss'], rtol=2e-5)
    np.testing.assert_allclose(a.grad.cpu().numpy(), golden['mse_loss_grad'], atol=1e-8, rtol=2e-5)


def test_erode_twice_matches_reference(golden):
    from mhmocap.morphology import Erode2D, Dilate2D
    x = dev(gi.erode_inputs())
    er = torch.nn.Sequential(Erode2D(kernel_size=3), Erode2D(kernel_size=3))
    np.testing.assert_array_equal(er(x).cpu().numpy(), golden['erode2'])
    # dilation is the dual of erosion on the complement (zero padding makes borders differ: compare the interior)
    d = Dilate2D(3)(x).cpu().numpy()
    e = 1 - Erode2D(3)(1 - x).cpu().numpy()
    np.testing.assert_array_equal(d[..., 1:-1, 1:-1], e[..., 1:-1, 1:-1])


def test_packed_bit_erosion_equals_float_erosion():
    """the staging path (bit planes, mh_erode_bits) and the module path (float maps) agree"""
    from mhhip import _lib
    from mhhip._lib import ptr, check
    from mhmocap.morphology import Erode2D
    rng = np.random.RandomState(3)
    T, N, H, W = 3, 5, 21, 30
    seg = (rng.rand(T, N, H, W) > 0.3).astype(np.float32)
    L = _lib.lib()
    st = _lib.stream_ptr(torch.device('cuda:0'))
    bits = torch.zeros(T, H, W, dtype=torch.int32, device='cuda:0')
    area = torch.zeros(T * N, device='cuda:0')
    check(L.mh_pack_masks(ptr(dev(seg)), T, N, H, W, ptr(bits), ptr(area), st))
    np.testing.assert_allclose(area.cpu().numpy().reshape(T, N), seg.sum((2, 3)))
    t1, t2 = torch.zeros_like(bits), torch.zeros_like(bits)
    check(L.mh_erode_bits(ptr(bits), ptr(t1), T, H, W, st))
    check(L.mh_erode_bits(ptr(t1), ptr(t2), T, H, W, st))
    want = Erode2D(3)(Erode2D(3)(dev(seg).view(T * N, 1, H, W))).view(T, N, H, W).cpu().numpy()
    got = ((t2.cpu().numpy()[:, None] >> np.arange(N)[None, :, None, None]) & 1).astype(np.float32)
    np.testing.assert_array_equal(got, want)


def test_camera_helpers_match_reference(golden):
    from mhmocap import transforms as tf
    pts, K, Kd = gi.projection_inputs()
    uv = tf.camera_projection_torch(dev(pts), dev(K))
    np.testing.assert_allclose(uv.cpu().numpy(), golden['proj_plain'], atol=2e-4)
    np.testing.assert_allclose(tf.camera_projection_torch(dev(pts), dev(K), Kd=Kd).cpu().numpy(), golden['proj_dist'], atol=2e-4)
    uvd = tf.camera_projection_torch(dev(pts), dev(K), return_depth=True)
    np.testing.assert_allclose(tf.camera_inverse_projection_torch(uvd, dev(K)).cpu().numpy(), golden['unproj'], atol=2e-5)
    np.testing.assert_allclose(tf.compute_calibration_matrix(1.0, 100.0, K[0], (240, 135)), golden['calib_land'], atol=1e-6)
    np.testing.assert_allclose(tf.compute_calibration_matrix(1.0, 100.0, K[0], (135, 240)), golden['calib_port'], atol=1e-6)
    np.testing.assert_allclose(tf.compute_calibration_matrix(1.0, 100.0, K[0], (256, 256)), golden['calib_sq'], atol=1e-6)


def test_cached_mask_statistics_follow_the_depth_ordering():
    """mh_sil_mask_stats_cached (round 4) keeps a frame's pixel counts while the near-to-far ordering of its people is
    unchanged: against the uncached kernel over a sequence of translations whose ordering changes in some frames at some
    steps, stays in others, with ties -- every output equal, every step."""
    import numpy as np
    from mhhip import _lib
    from mhhip._lib import check, ptr
    L = _lib.lib()
    dev = torch.device('cuda:0')
    st = _lib.stream_ptr(dev)
    rng = np.random.RandomState(5)
    for T, N, H, W in [(7, 4, 27, 40), (3, 1, 16, 16), (5, 9, 20, 33)]:
        bits = torch.tensor(rng.randint(0, 2 ** N, size=(T, H, W)).astype(np.int32), device=dev)
        p2d = torch.tensor(rng.rand(T * N).astype(np.float32) > 0.2, device=dev).float()
        mv = torch.tensor(rng.rand(T * N).astype(np.float32) > 0.2, device=dev).float()
        pT = torch.tensor(rng.randn(T, N, 3).astype(np.float32), device=dev)
        z = lambda dt: torch.zeros(T * N, dtype=dt, device=dev)
        f0, a0, D0, S0 = z(torch.int32), z(torch.float32), z(torch.float32), z(torch.float32)
        f1, a1, D1, S1 = z(torch.int32), z(torch.float32), z(torch.float32), z(torch.float32)
        tag = torch.zeros(T, dtype=torch.int32, device=dev)
        D1.fill_(-7.0)          # whatever the arrays held
        recount = 0
        for step in range(12):
            if step % 3 == 1:
                pT[:, :, 2] += torch.tensor(rng.randn(T, N).astype(np.float32) * 0.4, device=dev) * (torch.arange(T, device=dev) % 2 == 0).float()[:, None]
            if step == 7 and N > 1:
                pT[0, 1, 2] = pT[0, 0, 2]                    # a tie: the lower index is in front
            before = f1.clone()
            check(L.mh_sil_mask_stats(ptr(bits), T, N, H, W, ptr(pT), ptr(p2d), ptr(mv), ptr(f0), ptr(a0), ptr(D0), ptr(S0), st))
            check(L.mh_sil_mask_stats_cached(ptr(bits), T, N, H, W, ptr(pT), ptr(p2d), ptr(mv), ptr(f1), ptr(a1), ptr(D1), ptr(S1), ptr(tag), st))
            torch.cuda.synchronize()
            recount += int((before != f1).view(T, N).any(dim=1).sum())
            assert torch.equal(f0, f1) and torch.equal(a0, a1) and torch.equal(D0, D1) and torch.equal(S0, S1), (T, N, step)
        assert int(tag.sum()) == T
        if N > 1:
            assert 0 < recount < 12 * T
