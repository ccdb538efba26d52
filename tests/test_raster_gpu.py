"""GPU parity of the fused rasteriser + depth / silhouette residual kernel (mh_raster_terms)
against the oracle (oracle/raster_oracle.py selection + torch autograd): per-body loss values,
dL/dverts, dL/dzmin_lin, dL/dzmax_lin.  The oracle itself is pinned analytically
(tests/test_raster_oracle.py); PyTorch3D is unavailable, so this term is "parity unpinned"
with respect to the reference."""
import numpy as np
import pytest
import torch

from mhhip import synthetic
from oracle import fit_oracle as fo
from oracle import lbs_oracle as lo
from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu


def _scene(T, N, W, H, seed, zlo, zhi):
    rng = np.random.RandomState(seed)
    sp = synthetic.make_sequence_params(N, T, seed)
    pT = sp['trans_gt'].copy()
    pT[..., 2] = rng.uniform(zlo, zhi, (T, N))
    pT[..., 0] = rng.uniform(-0.25, 0.25, (T, N)) * pT[..., 2]
    pT[..., 1] = rng.uniform(-0.05, 0.1, (T, N))
    return sp, pT.astype(np.float32), rng


def _run_case(smpl_struct, smpl_regs, T, N, W, H, seed, zlo=3.0, zhi=6.0, fov=60.0, hip_selection=False, oracle_dtype=torch.float32, coefs=None):
    from mhhip import engine
    from mhhip.sequence import SequenceEngine
    from mhhip.raster import RasterTerms
    K = synthetic.default_cam_K((W, H), fov)
    sp, pT, rng = _scene(T, N, W, H, seed, zlo, zhi)
    model = engine.BodyModel(smpl_struct, smpl_regs)
    omodel = lo.BodyModel(smpl_struct, smpl_regs)
    faces = np.asarray(smpl_struct.f).astype(np.int64)
    betas = sp['betas_gt']
    coefs = dict(coefs or dict(depth=0.05, silhouette=0.1))
    e = SequenceEngine(model, (W, H), T, N, K, None, coefs, batch_size=2)
    zmin = rng.uniform(0.5, 1.5, T).astype(np.float32)
    zmax = rng.uniform(4.0, 9.0, T).astype(np.float32)
    e.set_leaves(pT, sp['poses_gt'], betas, zmin, zmax, rng.normal(0, 0.5, N).astype(np.float32))
    # masks: silhouette of a slightly different pose, so that alpha - seg is non-trivial
    with torch.no_grad():
        be = torch.tensor(betas)[None].expand(T, N, 10).reshape(-1, 10)
        out = lo.smpl_forward(omodel, be, torch.tensor(sp['poses_init']).view(-1, 72))
        s = torch.pow(torch.tensor(1.1), e.leaf('xscale').cpu())[None, :, None, None]
        v0 = s * out['verts'].view(T, N, -1, 3) + torch.tensor(pT).view(T, N, 1, 3) + torch.tensor([0.03, -0.02, 0.0])
        _, a0 = ro.render(v0.view(T * N, -1, 3), faces, K, (W, H))
    seg = (a0.view(T, N, H, W) > 0.5).float().numpy()
    seg[0, 0] = 0                                                  # an empty mask
    depths = rng.uniform(0, 1, (T, H, W)).astype(np.float32)
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    pose2d[..., 2] = 0.9
    if T > 1:
        pose2d[1, N - 1, :, 2] = 0.1                               # a body without a valid 2D pose
    e.stage(pose2d, sp['poses_init'], sp['valid'], betas, seg, depths)
    e.forward()
    L = __import__('mhhip._lib', fromlist=['lib']).lib()
    from mhhip._lib import ptr, check, stream_ptr
    check(L.mh_sil_mask_stats(ptr(e.bits), T, N, H, W, ptr(e.leaf('poses_T')), ptr(e.p2d_valid), ptr(e.mask_valid),
                              ptr(e.front), ptr(e.sil_apply), ptr(e.sil_D), ptr(e.sil_S), stream_ptr(e.dev)))
    gv = torch.zeros_like(e.verts)
    e.grads.zero_()
    log = torch.zeros(16, device=e.dev)
    rt = RasterTerms(e)
    rt(e, gv, log)
    torch.cuda.synchronize()
    sel = rt.selection(e)

    # ---- oracle -----------------------------------------------------------------------------------
    dt = oracle_dtype                                  # float64: what the float32 oracle itself is worth on an entry
    verts = e.verts.cpu().clone().to(dt).requires_grad_(True)
    tzmin = torch.tensor(zmin, dtype=dt, requires_grad=True)
    tzmax = torch.tensor(zmax, dtype=dt, requires_grad=True)
    got_sel = _hip_selection(sel, T * N, H, W)
    zbuf, alpha = ro.render(verts, faces, K, (W, H), selection=(got_sel[..., :1], got_sel[..., 1:]) if hip_selection else None)
    zbuf, alpha = zbuf.view(T, N, H, W), alpha.view(T, N, H, W)
    tseg = torch.tensor(seg, dtype=dt)
    conf = (torch.tensor(pose2d[..., 2:3]) >= 0.5).to(dt)
    p2d_valid = (conf.sum(dim=(2, 3)) >= 2).to(dt)
    mask_valid = (tseg.sum(dim=(2, 3)) >= 0.005 * H * W).to(dt)
    min_z = fo.softplus(tzmin).view(T, 1, 1)
    max_z = min_z.detach() + 1.0 + fo.softplus(tzmax).view(T, 1, 1)
    tgt = torch.tensor(depths, dtype=dt) * (1.0 / min_z - 1.0 / max_z) + 1.0 / max_z
    m = (zbuf > 0).to(dt) * fo.erode3x3(fo.erode3x3(tseg)) * p2d_valid[..., None, None]
    pred = 1.0 / torch.clamp(zbuf + 0.2, min=1e-3)
    lp = m * torch.log(torch.clamp(pred, min=1e-3))
    lt = m * torch.log(torch.clamp(tgt.unsqueeze(1), min=1e-3))
    cnt = m.sum(dim=(2, 3)) + 1
    depth_tn = (lp.sum(dim=(2, 3)) / cnt - lt.sum(dim=(2, 3)) / cnt) ** 2
    order = torch.argsort(torch.tensor(pT)[..., 2], dim=1)
    sil_tn = torch.zeros(T, N, dtype=dt)
    sil_list = []
    for t in range(T):
        acc = torch.zeros(H, W, dtype=dt)
        for r in range(N):
            n = int(order[t, r])
            if float(mask_valid[t, r] * p2d_valid[t, r]) > 0:
                sil_list.append((t, n, fo.masked_mse_loss(alpha[t, n], tseg[t, n], 1 - acc)))
            acc = ((acc + tseg[t, n]) > 0).to(dt)
    total = coefs['depth'] * depth_tn.sum() + coefs['silhouette'] * sum(x[2] for x in sil_list)
    total.backward()
    want_sil = np.zeros((T, N), np.float32)
    for t, n, v in sil_list:
        want_sil[t, n] = float(v)
    return dict(e=e, gv=gv.cpu().numpy(), want_gv=verts.grad.numpy(), depth=e.depth_body.cpu().numpy().reshape(T, N),
                want_depth=depth_tn.detach().numpy(), sil=e.sil_body.cpu().numpy().reshape(T, N), want_sil=want_sil,
                gzmin=e.leaf('zmin_lin', e.grads).cpu().numpy(), gzmax=e.leaf('zmax_lin', e.grads).cpu().numpy(),
                want_gzmin=tzmin.grad.numpy(), want_gzmax=tzmax.grad.numpy(), log=log.cpu().numpy(),
                zbuf=zbuf.detach().numpy(), sel=sel, faces=faces, K=K, verts=e.verts.cpu().numpy(), shape=(T, N, H, W))


def _hip_selection(sel, B, H, W):
    """(B,H,W,5) face ids of the HIP selection pass (-1 = empty): slot 0 = nearest face of the blur-1e-4 pass, 1..4 = the
    K=4 list of the blur-2e-5 pass"""
    win, koff, keys = sel
    EMPTY = np.uint64(0xffffffffffffffff)
    got = np.full((B, H, W, 5), -1, np.int64)
    for b in range(B):
        x0, y0, ww, wh = [int(v) for v in win[b]]
        if ww > 0 and wh > 0:
            k = keys[koff[b]:koff[b + 1]].reshape(wh, ww, 5)
            got[b, y0:y0 + wh, x0:x0 + ww] = np.where(k == EMPTY, -1, (k & np.uint64(0xffffffff)).astype(np.int64))
    return got


def _check(r):
    assert (r['zbuf'] > 0).sum() > 50, 'test scene does not cover any pixels'
    np.testing.assert_allclose(r['depth'], r['want_depth'], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(r['sil'], r['want_sil'], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(r['log'][1], r['want_depth'].sum(), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(r['log'][2], r['want_sil'].sum(), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(r['gzmin'], r['want_gzmin'], atol=2e-3 * max(np.abs(r['want_gzmin']).max(), 1e-8))
    np.testing.assert_allclose(r['gzmax'], r['want_gzmax'], atol=2e-3 * max(np.abs(r['want_gzmax']).max(), 1e-8))
    g, w = r['gv'], r['want_gv']
    scale = np.abs(w).max()
    assert scale > 0
    err = np.abs(g - w)
    # float atomics + the rare pixel whose blur-band test flips in the last ulp: demand a tight match on
    # all but a handful of vertices, and a tight match in aggregate
    assert (err > 2e-3 * scale).sum() <= 6, (err > 2e-3 * scale).sum()
    assert np.abs(g.sum(axis=1) - w.sum(axis=1)).max() < 5e-3 * np.abs(w).sum(axis=1).max()


def test_raster_terms_small(smpl_struct, smpl_regs):
    _check(_run_case(smpl_struct, smpl_regs, T=3, N=2, W=60, H=34, seed=5))


def test_raster_terms_portrait_and_square(smpl_struct, smpl_regs):
    _check(_run_case(smpl_struct, smpl_regs, T=2, N=2, W=34, H=60, seed=6))
    _check(_run_case(smpl_struct, smpl_regs, T=2, N=2, W=40, H=40, seed=7))


def test_raster_terms_strips(smpl_struct, smpl_regs):
    """bodies close to the camera: the screen window exceeds the LDS capacity -> row strips, two sweeps"""
    r = _run_case(smpl_struct, smpl_regs, T=1, N=2, W=160, H=96, seed=8, zlo=1.6, zhi=2.0)
    assert (r['zbuf'] > 0).sum() > 1920
    _check(r)


def test_raster_terms_wide_window_column_tiles(smpl_struct, smpl_regs):
    """a body whose screen window is wider than one LDS strip (640 px): the window is cut into column tiles"""
    r = _run_case(smpl_struct, smpl_regs, T=1, N=2, W=1000, H=40, seed=9, zlo=0.85, zhi=1.0, fov=1.6)
    span = [int(np.ptp(np.nonzero(c)[0])) if c.any() else 0 for c in (r['zbuf'][0] > 0).any(axis=1)]
    assert max(span) > 640, span
    _check(r)


def test_bodies_behind_or_off_camera_contribute_nothing(smpl_struct, smpl_regs):
    """edge cases: a body behind the camera, one far off screen, one straddling the image border"""
    from mhhip import engine
    from mhhip.sequence import SequenceEngine
    from mhhip.raster import RasterTerms
    T, N, W, H = 1, 3, 64, 48
    K = synthetic.default_cam_K((W, H), 60.0)
    sp = synthetic.make_sequence_params(N, T, 3)
    pT = sp['trans_gt'].copy()
    pT[0, 0] = [0.0, 0.0, -3.0]          # behind the camera
    pT[0, 1] = [40.0, 0.0, 4.0]          # far off to the side
    pT[0, 2] = [1.9, 0.1, 3.5]           # partly inside
    model = engine.BodyModel(smpl_struct, smpl_regs)
    e = SequenceEngine(model, (W, H), T, N, K, None, dict(depth=0.05, silhouette=0.1), batch_size=1)
    e.set_leaves(pT.astype(np.float32), sp['poses_gt'], sp['betas_gt'], np.ones(T, np.float32), 5 * np.ones(T, np.float32),
                 np.zeros(N, np.float32))
    seg = np.zeros((T, N, H, W), np.float32)
    seg[:, :, 10:40, 40:64] = 1
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    pose2d[..., 2] = 0.9
    e.stage(pose2d, sp['poses_init'], sp['valid'], sp['betas_gt'], seg, np.full((T, H, W), 0.5, np.float32))
    e.forward()
    L = __import__('mhhip._lib', fromlist=['lib']).lib()
    from mhhip._lib import ptr, check, stream_ptr
    check(L.mh_sil_mask_stats(ptr(e.bits), T, N, H, W, ptr(e.leaf('poses_T')), ptr(e.p2d_valid), ptr(e.mask_valid),
                              ptr(e.front), ptr(e.sil_apply), ptr(e.sil_D), ptr(e.sil_S), stream_ptr(e.dev)))
    gv = torch.zeros_like(e.verts)
    e.grads.zero_()
    log = torch.zeros(16, device=e.dev)
    zbuf = torch.empty(T * N, H, W, device=e.dev)
    RasterTerms(e)(e, gv, log, zbuf_out=zbuf)
    torch.cuda.synchronize()
    g = gv.cpu().numpy().reshape(N, -1, 3)
    assert np.isfinite(g).all() and np.isfinite(log.cpu().numpy()).all()
    assert (zbuf[0] < 0).all() and (zbuf[1] < 0).all()          # nothing rasterised for the first two bodies
    assert np.abs(g[0]).max() == 0 and np.abs(g[1]).max() == 0
    assert (zbuf[2] > 0).sum() > 20 and np.abs(g[2]).max() > 0


def _face_eval64(ndc, faces, b, f, xf, yf):
    """float64 (pz, inside, d2) of face f of body b at the pixel centre (xf, yf) -- CheckPixelInsideFace"""
    v = ndc[b][faces[f]].astype(np.float64)
    (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = v
    edge = lambda px, py, ax, ay, bx, by: (px - ax) * (by - ay) - (py - ay) * (bx - ax)
    area = edge(x2, y2, x0, y0, x1, y1) + 1e-8
    w = np.array([edge(xf, yf, x1, y1, x2, y2), edge(xf, yf, x2, y2, x0, y0), edge(xf, yf, x0, y0, x1, y1)]) / area
    c = np.maximum(w, 0)
    pz = float((c / max(c.sum(), 1e-5)) @ np.array([z0, z1, z2]))

    def seg(ax, ay, bx, by):
        bax, bay = bx - ax, by - ay
        l2 = bax * bax + bay * bay
        if l2 <= 1e-8:
            return (xf - bx) ** 2 + (yf - by) ** 2
        t = min(max((bax * (xf - ax) + bay * (yf - ay)) / l2, 0.0), 1.0)
        return (ax + t * bax - xf) ** 2 + (ay + t * bay - yf) ** 2
    return pz, bool((w > 0).all()), min(seg(x0, y0, x1, y1), seg(x0, y0, x2, y2), seg(x1, y1, x2, y2))


def _selection_differences(r):
    """Pixels where the HIP selection and the oracle's brute-force selection (same vertices) pick different faces, each
    checked to be a NEAR-TIE in float64: the HIP list is a valid K-nearest list (depths within 1e-5 of the cut count as equal)
    of the faces in the blur band for some resolution of the faces that reach the pixel centre to within 1e-4 of the blur
    radius.  (A tie face that drops out of the band is replaced by the next-nearest face, which is itself no tie:
    tools/fuzz_raster.py found two such pixels in 180 random scenes.)"""
    T, N, H, W = r['shape']
    B = T * N
    faces = r['faces']
    ndc = ro.to_ndc(torch.tensor(r['verts']), r['K'], (W, H)).numpy().astype(np.float32)
    got = _hip_selection(r['sel'], B, H, W)
    f8, _ = ro.select_faces(ndc, faces, H, W, 1e-4, 8)
    f4, _ = ro.select_faces(ndc, faces, H, W, 2e-5, 4)
    want = np.concatenate([f8[..., :1], f4], axis=-1)
    xs, ys = ro.pixel_centres_ndc(H, W)
    live = int((want[..., 0] >= 0).sum() + (want[..., 1] >= 0).sum())
    ndiff, not_ties = 0, []
    d0 = got[..., 0] != want[..., 0]
    d4 = (np.sort(got[..., 1:], -1) != np.sort(want[..., 1:], -1)).any(-1)    # alpha is a product: the K=4 list is a set
    import itertools
    for b, y, x in zip(*np.nonzero(d0 | d4)):
        ndiff += 1
        xf, yf = float(xs[x]), float(ys[y])
        for lo, hi, blur in ((0, 1, 1e-4), (1, 5, 2e-5)):
            A, Bset = set(got[b, y, x, lo:hi]) - {-1}, set(want[b, y, x, lo:hi]) - {-1}
            if A == Bset:
                continue
            K = hi - lo
            U = A | Bset
            ev = {f: _face_eval64(ndc, faces, b, f, xf, yf) for f in U}
            # faces whose band membership is a tie in float64; every other face of U is in the band (it was selected by one side)
            ties = [f for f in U if (not ev[f][1]) and abs(ev[f][2] - blur) <= 1e-4 * blur]
            sure = [f for f in U if f not in ties]
            ok = False
            # the HIP list must be a K-nearest list (depth ties at the cut allowed) of the faces in the band for SOME resolution
            # of the band ties -- a tie face that drops out is replaced by the next face, which is then no tie itself
            for r_ in range(len(ties) + 1):
                for S in itertools.combinations(ties, r_):
                    members = set(sure) | set(S)
                    if not A <= members or len(A) != min(K, len(members)):
                        continue
                    zmax = max(ev[f][0] for f in A)
                    if all(ev[f][0] >= zmax - 1e-5 * abs(zmax) for f in members - A):
                        ok = True
            if not ok:
                for f in A ^ Bset:
                    not_ties.append((int(b), int(y), int(x), int(f)) + ev[f])
    return ndiff, live, not_ties


@pytest.mark.parametrize('case', [dict(T=3, N=2, W=60, H=34, seed=5), dict(T=2, N=3, W=120, H=68, seed=11, zlo=2.2, zhi=4.0)])
def test_deterministic_scatter_and_where_the_gradient_errors_come_from(smpl_struct, smpl_regs, case):
    """MHHIP_DETERMINISTIC / mh_raster_set_deterministic(1):
    (a) dL/dverts is bit-identical from run to run (the production scatter is not: fp32 atomics) and differs from the
        production result by rounding only;
    (b) SELECTION: the K nearest faces of a pixel are a discontinuous function of fp32 depths and distances -- a pixel
        centre sees ~50 sub-pixel faces inside the blur radius, the kernel interpolates with v_rcp_f32 where the oracle
        divides.  The two selections differ on ~1 % of the live pixels, and EVERY differing face is verified in float64 to
        be a near-tie (depth within 1e-5 of the cut of the list, or distance within 1e-4 of the blur radius);
    (c) GRADIENTS: with the oracle evaluating the faces the HIP kernel selected, EVERY entry of dL/dverts agrees to 5e-4
        of the largest entry (measured worst 2.3e-4; the depth-range gradients to 2e-4) -- no percentile, no outlier
        allowance.  What is left is the fp32 noise floor of the barycentric Jacobian itself: its terms cancel to
        translation invariance and are multiplied by 1/area ~ 1e6 of a sub-pixel face, in the kernel and in the oracle's
        fp32 autograd alike.  (This test found a real defect on the way: the normalised clipped weights went through
        v_rcp_f32, whose 1-ulp residue survived that amplification -- 6.6e-4; they are IEEE divisions now.)"""
    from mhhip.raster import set_deterministic
    old = set_deterministic(True)
    try:
        r = _run_case(smpl_struct, smpl_regs, hip_selection=True, **case)
        r2 = _run_case(smpl_struct, smpl_regs, hip_selection=True, **case)
        np.testing.assert_array_equal(r['gv'], r2['gv'])
        np.testing.assert_array_equal(r['gzmin'], r2['gzmin'])
        np.testing.assert_array_equal(r['gzmax'], r2['gzmax'])
        set_deterministic(False)
        rp = _run_case(smpl_struct, smpl_regs, hip_selection=True, **case)
    finally:
        set_deterministic(old)
    g, w = r['gv'], r['want_gv']
    scale = np.abs(w).max()
    assert scale > 0 and np.abs(rp['gv'] - g).max() <= 2e-6 * scale              # atomics noise of the production scatter
    # (c) same faces -> same numbers, everywhere
    np.testing.assert_allclose(g, w, atol=5e-4 * scale, rtol=0)
    np.testing.assert_allclose(rp['gv'], w, atol=5e-4 * scale, rtol=0)
    assert (np.abs(g - w) > 2e-4 * scale).sum() <= 3
    np.testing.assert_allclose(r['depth'], r['want_depth'], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(r['sil'], r['want_sil'], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(r['gzmin'], r['want_gzmin'], atol=2e-4 * max(np.abs(r['want_gzmin']).max(), 1e-8), rtol=0)
    np.testing.assert_allclose(r['gzmax'], r['want_gzmax'], atol=2e-4 * max(np.abs(r['want_gzmax']).max(), 1e-8), rtol=0)
    # (b) the selections
    ndiff, live, not_ties = _selection_differences(r)
    print('selection differs on %d of %d live (pixel, pass) entries; not explained as near-ties: %d' % (ndiff, live, len(not_ties)))
    # measured (round 6): 5 of 344 and 45 of 3111 = 1.45 % at these close-up sizes (0.93 % at C3, tests/test_full_size_gpu.py);
    # the gate is 1.5x that
    assert live > 300 and ndiff <= 2.2e-2 * live + 3, (ndiff, live)
    assert not not_ties, not_ties[:5]


def test_face_straddling_the_camera_plane_is_skipped_whole():
    """Known answer for the behind-the-camera rule (oracle/raster_select.c says which PyTorch3D backend each rule restates): the
    HIP kernel implements the WHOLE-FACE rule (zmin < kEpsilon: the CUDA kernels') -- a triangle with its apex behind the
    camera leaves the z-buffer and the silhouette empty, while the same triangle pushed in front of the camera is drawn; the
    oracle agrees under its default rule."""
    import types
    from mhhip.raster import render
    W, H = 64, 48
    K = synthetic.default_cam_K((W, H), 60.0)
    faces = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    model = types.SimpleNamespace(faces=faces)
    tri = np.array([[-0.6, -0.5, 2.0], [0.6, -0.5, 2.0], [0.0, 0.5, -1.0]], np.float32)          # apex behind the camera
    far = np.array([[5.0, 5.0, 50.0], [5.1, 5.0, 50.0], [5.0, 5.1, 50.0]], np.float32)           # (off screen: the table needs a 2nd face)
    front = tri + np.array([0, 0, 1.5], np.float32)
    verts = torch.tensor(np.stack([np.concatenate([tri, far]), np.concatenate([front, far])]).astype(np.float32), device='cuda:0')
    zbuf, alpha = render(model, verts, K, (W, H))
    zbuf, alpha = zbuf.cpu().numpy(), alpha.cpu().numpy()
    assert (zbuf[0] == -1).all() and (alpha[0] == 0).all()                   # straddling: skipped entirely
    assert (zbuf[1] > 0).sum() > 30 and alpha[1].max() > 0.99                # the same triangle, wholly in front: drawn
    ndc = ro.to_ndc(verts.cpu(), K, (W, H)).numpy().astype(np.float32)
    f8, _ = ro.select_faces(ndc, faces.astype(np.int64), H, W, 1e-4, 8)
    assert (f8[0] == -1).all() and ((f8[1, ..., 0] >= 0) == (zbuf[1] > 0)).mean() > 0.995
