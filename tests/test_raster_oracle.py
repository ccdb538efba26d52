"""Known-answer pins of the rasteriser restatement (oracle/raster_oracle.py, raster_select.c).
PyTorch3D itself is unavailable (parity unpinned), so the semantics are pinned analytically:
fronto-parallel plane depth, barycentric depth of a tilted triangle, sigmoid alpha at a known
edge distance, nearest-wins / product rule for two overlapping triangles, non-square aspect,
pixel-centre convention, and finite-difference checks of the autograd gradients."""
import numpy as np
import torch

from oracle import raster_oracle as ro
from mhhip import synthetic


def _cam(W, H):
    return synthetic.default_cam_K((W, H), 60.0)


def _project_px(pt, K):
    return K[0, 0] * pt[0] / pt[2] + K[0, 2], K[1, 1] * pt[1] / pt[2] + K[1, 2]


def test_pixel_centre_convention_and_aspect():
    # a camera-space point projecting to (u, v) pixels must land on NDC of pixel centre (u-.5, v-.5)
    for (W, H) in [(48, 32), (32, 48), (40, 40)]:
        K = _cam(W, H)
        xs, ys = ro.pixel_centres_ndc(H, W)
        for (xi, yi) in [(0, 0), (W - 1, H - 1), (7, 11)]:
            z = 3.0
            X = (xi + 0.5 - K[0, 2]) * z / K[0, 0]
            Y = (yi + 0.5 - K[1, 2]) * z / K[1, 1]
            ndc = ro.to_ndc(torch.tensor([[[X, Y, z]]], dtype=torch.float64), K, (W, H))[0, 0]
            assert abs(float(ndc[0]) - xs[xi]) < 1e-5 and abs(float(ndc[1]) - ys[yi]) < 1e-5
        assert abs(max(abs(xs.max()), abs(ys.max())) - max(W, H) / min(W, H) * (1 - 1 / max(W, H))) < 1e-5


def test_fronto_parallel_triangle_depth_and_empty():
    W, H = 48, 32
    K = _cam(W, H)
    verts = torch.tensor([[[-1.0, -0.8, 4.0], [1.2, -0.7, 4.0], [0.1, 0.9, 4.0]]])
    faces = np.array([[0, 1, 2]])
    z, a = ro.render(verts, faces, K, (W, H))
    inside = z[0] > 0
    assert inside.sum() > 60
    np.testing.assert_allclose(z[0][inside].numpy(), 4.0, atol=1e-5)
    assert float(z[0, 0, 0]) == -1.0 and float(a[0, 0, 0]) == 0.0
    # alpha deep inside: sigmoid(d^2/1e-4) saturates to 1
    u, v = _project_px([0.1, -0.2, 4.0], K)
    assert float(a[0, int(v), int(u)]) > 0.999


def test_tilted_triangle_barycentric_depth():
    W, H = 64, 64
    K = _cam(W, H)
    V = np.array([[-1.0, -1.0, 3.0], [1.0, -1.0, 5.0], [0.0, 1.0, 4.0]], np.float32)
    z, _ = ro.render(torch.tensor(V[None]), np.array([[0, 1, 2]]), K, (W, H))
    # screen-space (not perspective-correct) interpolation of z
    uv = np.array([_project_px(p, K) for p in V])
    for (xi, yi) in [(30, 30), (28, 25), (34, 28)]:
        p = np.array([xi + 0.5, yi + 0.5])
        T = np.array([[uv[0, 0] - uv[2, 0], uv[1, 0] - uv[2, 0]], [uv[0, 1] - uv[2, 1], uv[1, 1] - uv[2, 1]]])
        l = np.linalg.solve(T, p - uv[2])
        w = np.array([l[0], l[1], 1 - l.sum()])
        assert (w > 0).all()
        np.testing.assert_allclose(float(z[0, yi, xi]), (w * V[:, 2]).sum(), atol=2e-4)


def test_alpha_at_known_edge_distance_and_blur_band():
    W, H = 64, 64
    K = _cam(W, H)
    z0 = 4.0
    # a big triangle whose left edge is the vertical line u = 20.25 px
    X = (20.25 - K[0, 2]) * z0 / K[0, 0]
    verts = torch.tensor([[[X, -3.0, z0], [X, 3.0, z0], [3.0, 0.0, z0]]])
    zb, a = ro.render(verts, np.array([[0, 1, 2]]), K, (W, H))
    px = 2.0 / 64                                       # NDC per pixel (short side 64)
    d_out = (20.25 - 19.5) * px                         # pixel 19 centre is 0.75 px outside
    d_in = (20.5 - 20.25) * px                          # pixel 20 centre is 0.25 px inside
    # outside: only within sqrt(2e-5) = 0.143 px -> pixel 19 is NOT covered by the silhouette pass
    assert float(a[0, 32, 19]) == 0.0
    np.testing.assert_allclose(float(a[0, 32, 20]), 1 / (1 + np.exp(-(d_in ** 2) / 1e-4)), rtol=1e-4)
    # depth pass: blur 1e-4 -> 0.32 px band: pixel 19 (0.75 px away) empty, pixel 20 filled
    assert float(zb[0, 32, 19]) == -1.0 and abs(float(zb[0, 32, 20]) - z0) < 1e-5
    # move the edge so that pixel 19 is 0.1 px outside: inside both blur bands
    X2 = (19.6 - K[0, 2]) * z0 / K[0, 0]
    verts2 = torch.tensor([[[X2, -3.0, z0], [X2, 3.0, z0], [3.0, 0.0, z0]]])
    zb2, a2 = ro.render(verts2, np.array([[0, 1, 2]]), K, (W, H))
    d = 0.1 * px
    np.testing.assert_allclose(float(a2[0, 32, 19]), 1 / (1 + np.exp((d ** 2) / 1e-4)), rtol=1e-3)
    assert abs(float(zb2[0, 32, 19]) - z0) < 1e-5       # clipped barycentrics give the edge depth


def test_two_overlapping_triangles_nearest_wins_and_product_rule():
    W, H = 64, 64
    K = _cam(W, H)
    verts = torch.tensor([[[-1.0, -1.0, 5.0], [1.0, -1.0, 5.0], [0.0, 1.0, 5.0],
                           [-0.5, -0.5, 3.0], [0.5, -0.5, 3.0], [0.0, 0.5, 3.0]]])
    faces = np.array([[0, 1, 2], [3, 4, 5]])
    z, a = ro.render(verts, faces, K, (W, H))
    assert abs(float(z[0, 32, 32]) - 3.0) < 1e-5        # nearest face wins the z-buffer
    u, v = _project_px([0.9, -0.95, 5.0], K)
    assert abs(float(z[0, int(v), int(u)]) - 5.0) < 1e-5
    ndc = ro.to_ndc(verts, K, (W, H))
    f4, _ = ro.select_faces(ndc.numpy(), faces, H, W, 2e-5, 4)
    assert list(f4[0, 32, 32][:2]) == [1, 0] and f4[0, 32, 32][2] == -1     # sorted near -> far
    _, sd, valid = ro.fragments(ndc, faces, f4, H, W)
    p = torch.sigmoid(-sd[0, 32, 32] / 1e-4) * valid[0, 32, 32]
    np.testing.assert_allclose(float(a[0, 32, 32]), float(1 - (1 - p[0]) * (1 - p[1])), rtol=1e-6)


def test_gradients_finite_differences():
    W, H = 24, 16
    K = _cam(W, H)
    rng = np.random.RandomState(0)
    V = np.array([[-0.9, -0.7, 3.0], [0.8, -0.6, 3.6], [0.1, 0.8, 3.3], [-0.4, 0.5, 2.8], [0.6, 0.4, 4.0]], np.float64)
    faces = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 2]])
    wz = torch.tensor(rng.normal(0, 1, (1, H, W)))
    wa = torch.tensor(rng.normal(0, 1, (1, H, W)))

    def loss(v):
        z, a = ro.render(v, faces, K, (W, H))
        return ((z > 0).to(z.dtype) * z * wz).sum() + (a * wa).sum()

    v = torch.tensor(V[None], requires_grad=True)
    loss(v).backward()
    g = v.grad.numpy()[0]
    eps = 1e-6
    for i in range(V.shape[0]):
        for c in range(3):
            vp, vm = V.copy(), V.copy()
            vp[i, c] += eps
            vm[i, c] -= eps
            fd = (float(loss(torch.tensor(vp[None]))) - float(loss(torch.tensor(vm[None])))) / (2 * eps)
            assert abs(fd - g[i, c]) < 2e-4 * max(1.0, abs(g).max()), (i, c, fd, g[i, c])


def test_capsule_body_render_is_sane():
    st = synthetic.make_smpl_struct(1)
    W, H = 60, 34
    K = _cam(W, H)
    v = torch.tensor(np.asarray(st.v_template, np.float32))[None] * torch.tensor([1.0, -1.0, 1.0]) + torch.tensor([0.2, 0.1, 4.0])
    z, a = ro.render(v, np.asarray(st.f).astype(np.int64), K, (W, H))
    cover = (z[0] > 0)
    assert 20 < int(cover.sum()) < 400
    assert float(z[0][cover].min()) > 3.7 and float(z[0][cover].max()) < 4.3
    assert float(a.max()) > 0.9 and float(a[0, 0, 0]) == 0.0
    # the depth pass has the wider blur band (1e-4 vs 2e-5): a few rim pixels have depth but no alpha
    assert float((a[0][cover] > 0.5).float().mean()) > 0.8


def test_face_straddling_the_camera_plane_under_both_backend_rules():
    """one triangle with its apex BEHIND the camera (z = -1), base in front (z = 2), given directly in NDC.
    Whole-face rule (the CUDA kernels' zmin < kEpsilon; default, what the HIP kernel implements): nothing is rasterised.
    Per-pixel rule (the plain naive CPU loop): the pixels whose interpolated depth is >= 0 are kept -- analytically, with the
    apex's barycentric weight w2 = (y + 0.5) / 1.1, pz = 2 (1 - w2) - w2 >= 0  <=>  y <= 0.2333."""
    H = W = 41
    v = np.array([[[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.0, 0.6, -1.0]]], np.float32)
    f = np.array([[0, 1, 2]])
    xs, ys = ro.pixel_centres_ndc(H, W)
    old = ro.set_behind_camera_rule(1)
    try:
        face, z = ro.select_faces(v, f, H, W, 0.0, 1)
        assert (face == -1).all()
        ro.set_behind_camera_rule(0)
        face, z = ro.select_faces(v, f, H, W, 0.0, 1)
    finally:
        ro.set_behind_camera_rule(old)
    got = face[0, :, :, 0] >= 0
    assert got.sum() > 50
    for yi in range(H):
        for xi in range(W):
            x, y = float(xs[xi]), float(ys[yi])
            # inside the NDC triangle (strictly), away from the edges and from the pz = 0 line by a pixel fraction
            w2 = (y + 0.5) / 1.1
            half = 0.5 * (1.0 - w2)
            inside = -0.5 < y < 0.6 and abs(x) < half
            margin = min(abs(abs(x) - half), abs(y + 0.5), abs(y - 0.2333)) > 0.03
            if inside and margin:
                assert bool(got[yi, xi]) == (y < 0.2333), (xi, yi, x, y)
                if got[yi, xi]:
                    assert abs(float(z[0, yi, xi, 0]) - (2.0 * (1.0 - w2) - w2)) < 1e-4
