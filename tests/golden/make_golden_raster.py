"""Pin the depth / silhouette COMPOSITION of ``fit`` (reference optimizer.py:425-477) to the reference's own loop.

Runs only in the build container (``/root/reference`` present).  PyTorch3D is absent here, so the
``pytorch3d`` stubs of ``make_golden.py`` are replaced by stubs that hand the reference's
``MeshRasterizer`` / ``MeshRenderer`` calls to the build's CPU rasteriser restatement
(``oracle/raster_oracle.py``, differentiable: the reference's autograd runs through it).  With that,
the REFERENCE code executes, around a non-empty z-buffer and a non-zero silhouette:

* ``target_disp`` from the normalised disparity and the depth-range leaves (:425),
* the supervision mask ``(zbuf>0) * erode^2(seg) * pose2d_valid`` (:432-438),
* ``1/clamp(zbuf+0.2, eps)`` and the mean-log-disparity loss (:440-442, losses.py:19-30),
* near->far ordering, the rank-indexed gate (:472) and the accumulated occlusion mask (:450-477),
* the camera hand-over: the stubs use the ``R``, ``T``, 4x4 ``K`` and the two ``RasterizationSettings`` the
  reference's constructor builds (:204-232), not the build's copies of them.

What stays "parity unpinned" is ONLY the inside of PyTorch3D (face selection, clipped barycentrics, edge
distances, sigmoid blend): that is ``oracle/raster_oracle.py`` on both sides of this fixture.

    python tests/golden/make_golden_raster.py      ->  tests/golden/reference_raster_cpu.npz
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import golden_inputs as gi  # noqa: E402
import make_golden as mg  # noqa: E402
from mhhip import synthetic  # noqa: E402
from oracle import lbs_oracle, raster_oracle  # noqa: E402

CALLS = {'raster': [], 'render': []}       # settings every stub call was made with (checked at the end)


def _install_raster_stubs(faces):
    rend = types.ModuleType('pytorch3d.renderer')
    stru = types.ModuleType('pytorch3d.structures')

    class _Bag(object):
        def __init__(self, *a, **kw):
            self.__dict__.update(kw)

    class Meshes(object):
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces

    def _ndc(cam, verts):
        """MeshRasterizer.transform: world -> view (X.R + T), clip = K.[x,y,z,1], NDC xy = clip_xy / clip_w,
        z = view depth (PyTorch3D keeps the view-space z for the z-buffer)."""
        R, T, K = cam.R[0], cam.T[0], cam.K[0]
        view = verts @ R + T
        hom = torch.cat([view, torch.ones_like(view[..., :1])], dim=-1)
        clip = hom @ K.t()
        return torch.stack([clip[..., 0] / clip[..., 3], clip[..., 1] / clip[..., 3], view[..., 2]], dim=-1)

    def _frags(rast, meshes):
        s = rast.raster_settings
        H, W = s.image_size
        assert s.perspective_correct is False
        f_np = meshes.faces[0].cpu().numpy().astype(np.int64)
        assert np.array_equal(f_np, faces)
        ndc = _ndc(rast.cameras, meshes.verts)
        p2f, _ = raster_oracle.select_faces(ndc.detach().numpy().astype(np.float32), f_np, H, W, s.blur_radius,
                                            s.faces_per_pixel)
        z, d, valid = raster_oracle.fragments(ndc, f_np, p2f, H, W)
        return z, d, valid, s

    class MeshRasterizer(_Bag):
        def __call__(self, meshes):
            z, d, valid, s = _frags(self, meshes)
            CALLS['raster'].append((tuple(s.image_size), s.blur_radius, s.faces_per_pixel))
            return _Bag(zbuf=z, dists=d)

    class MeshRenderer(_Bag):
        def __call__(self, meshes):
            z, d, valid, s = _frags(self.rasterizer, meshes)
            CALLS['render'].append((tuple(s.image_size), s.blur_radius, s.faces_per_pixel))
            # SoftSilhouetteShader -> sigmoid_alpha_blend with BlendParams().sigma = 1e-4
            prob = torch.sigmoid(-d / 1e-4) * valid.to(d.dtype)
            alpha = 1.0 - torch.prod(1.0 - prob, dim=-1)
            return torch.stack([torch.ones_like(alpha)] * 3 + [alpha], dim=-1)

    for name in ['FoVPerspectiveCameras', 'RasterizationSettings', 'SoftSilhouetteShader']:
        setattr(rend, name, type(name, (_Bag,), {}))
    rend.MeshRasterizer = MeshRasterizer
    rend.MeshRenderer = MeshRenderer
    stru.Meshes = Meshes
    sys.modules.update({'pytorch3d': types.ModuleType('pytorch3d'), 'pytorch3d.renderer': rend,
                        'pytorch3d.structures': stru, 'cv2': types.ModuleType('cv2')})


def make_inputs(model, T=20, N=2, W=96, H=60, seed=41):
    """Two bodies that cross in depth and overlap on screen; ground-truth silhouettes / disparity from the
    oracle renderer.  Stored in the fixture file (inputs are data), so the tests do not re-render them."""
    rng = np.random.RandomState(seed)
    sp = synthetic.make_sequence_params(N, T, seed, z_range=(2.3, 3.5))
    tr = np.zeros((T, N, 3), np.float32)
    t = np.arange(T, dtype=np.float32)
    tr[:, 0] = np.stack([-0.40 + 0.03 * t, 0.2 + 0 * t, 2.4 + 0.055 * t], -1)       # walks away, left -> right
    tr[:, 1] = np.stack([0.45 - 0.035 * t, 0.2 + 0 * t, 3.4 - 0.055 * t], -1)         # approaches, right -> left
    for n in range(2, N):                                                              # (timing tool: more tracks)
        tr[:, n] = np.stack([-1.2 + 0.8 * (n - 2) + 0.01 * t, 0.2 + 0 * t, 4.0 + 0.5 * (n - 2) + 0 * t], -1)
    cam_K = synthetic.default_cam_K((W, H), 60.0)
    with torch.no_grad():
        out = lbs_oracle.smpl_forward(model, torch.tensor(np.tile(sp['betas_gt'][None], (T, 1, 1))).view(-1, 10),
                                      torch.tensor(sp['poses_gt']).view(-1, 72))
        verts = out['verts'].view(T, N, -1, 3) + torch.tensor(tr)[:, :, None]
        j17 = out['joints_alphapose'].view(T, N, 17, 3) + torch.tensor(tr)[:, :, None]
        zb, al = raster_oracle.render(verts.view(T * N, -1, 3), model.faces, cam_K, (W, H))
    zb = zb.view(T, N, H, W).numpy()
    al = al.view(T, N, H, W).numpy()
    zfar = np.where((zb > 0) & (al > 0.5), zb, 1e9)
    nearest = zfar.argmin(1)
    covered = zfar.min(1) < 1e8
    seg = np.zeros((T, N, H, W), np.float32)
    for n in range(N):
        seg[:, n] = (covered & (nearest == n)).astype(np.float32)
    ys = (np.arange(H, dtype=np.float32) + 0.5 - cam_K[1, 2]) / cam_K[1, 1]
    bg = np.minimum(np.where(ys[:, None] > 1e-3, 1.15 / np.maximum(ys[:, None], 1e-3), 10.0), 10.0)
    bg = np.tile(bg, (1, W)).astype(np.float32)
    depth = np.minimum(np.where(covered, zfar.min(1), 1e9), bg[None])
    disp = 1.0 / depth
    lo, hi = disp.min(axis=(1, 2), keepdims=True), disp.max(axis=(1, 2), keepdims=True)
    depths = ((disp - lo) / np.maximum(hi - lo, 1e-6)).astype(np.float32)
    uv = (j17[..., :2] / j17[..., 2:]).numpy() * np.array([cam_K[0, 0], cam_K[1, 1]], np.float32) + cam_K[:2, 2]
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    pose2d[..., :2] = uv + rng.normal(0, 0.3, uv.shape)
    conf = rng.uniform(0.6, 1.0, (T, N, 17)).astype(np.float32)
    conf[rng.rand(T, N, 17) < 0.1] = 0.1
    pose2d[..., 2] = conf
    # gates: bodies without a valid 2D pose / without a mask, before AND after the two swap depth order, so the
    # rank-indexed gate of optimizer.py:472 differs from a person-indexed one
    pose2d[3, 1, :, 2] = 0.1
    pose2d[14, 1, :, 2] = 0.1
    seg[5, 0] = 0
    seg[16, 0] = 0
    images = rng.randint(0, 255, (T, H, W, 3)).astype(np.uint8)
    backmasks = (seg.sum(1) == 0).astype(np.int64)
    scene_mask = backmasks.min(axis=0) > 0
    return dict(T=T, N=N, H=H, W=W, cam_K=cam_K, pose2d=pose2d, seg_mask=seg, depths=depths, images=images,
                backmasks=backmasks, poses_smpl=sp['poses_init'], betas_smpl=sp['betas_init'], valid_smpl=sp['valid'],
                trans_gt=tr, scene_depth=bg, scene_mask=scene_mask)


def main():
    assert os.path.isdir(mg.REF), 'reference not present: fixtures can only be regenerated in the build container'
    sys.argv = ['x']
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    faces = np.asarray(struct.f).astype(np.int64)
    _install_raster_stubs(faces)
    mg._ref_package()
    smpl = importlib.import_module('refmh.smpl')
    losses = importlib.import_module('refmh.losses')
    optim = importlib.import_module('refmh.optimizer')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp()
    paths = {}
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        paths[k] = os.path.join(tmp, fn)
        np.save(paths[k], regs[k])

    def build_smpl(model_path=None, **kw):
        kw.setdefault('J_reg_extra9_path', paths['extra9'])
        kw.setdefault('J_reg_h36m17_path', paths['h36m'])
        kw.setdefault('J_reg_alphapose_path', paths['alphapose'])
        return smpl.SMPL(model_path, data_struct=smpl.Struct(**struct.__dict__), **kw)

    optim.SMPL = lambda path, **kw: build_smpl(**kw)
    fin = make_inputs(lbs_oracle.BodyModel(struct, regs))
    out = {}
    for k in ['cam_K', 'pose2d', 'depths', 'poses_smpl', 'betas_smpl', 'valid_smpl', 'trans_gt', 'scene_depth']:
        out['in_' + k] = np.asarray(fin[k], np.float32)
    out['in_seg_mask'] = fin['seg_mask'].astype(np.uint8)
    out['in_images'] = fin['images']
    out['in_backmasks'] = fin['backmasks'].astype(np.uint8)
    out['in_scene_mask'] = fin['scene_mask'].astype(np.uint8)
    out['in_dims'] = np.array([fin['T'], fin['N'], fin['H'], fin['W']], np.int64)

    # record what the reference's own loss builders return, call by call (optimizer.py:442, 474)
    rec = {'depth': [], 'sil': []}
    real_depth, real_mse = losses.build_avg_depth_loss_fn, losses.build_masked_mse_loss_fn

    def rec_depth(*a, **kw):
        fn = real_depth(*a, **kw)

        def wrapped(pred, true, mask):
            v = fn(pred, true, mask)
            rec['depth'].append(float(v.detach()))
            rec.setdefault('mask_px', []).append(float(mask.sum()))
            return v
        return wrapped

    def rec_mse(*a, **kw):
        fn = real_mse(*a, **kw)

        def wrapped(a_, b_, m_):
            v = fn(a_, b_, m_)
            rec['sil'].append(float(v.detach()))
            return v
        return wrapped

    optim.build_avg_depth_loss_fn = rec_depth
    optim.build_masked_mse_loss_fn = rec_mse
    c = gi.COEFS
    coef_kw = dict(proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
                   reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'],
                   reg_poses_coef=c['reg_poses'], reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'],
                   reg_foot_sliding_coef=c['reg_foot_sliding'])

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return fin['T']

        def __getitem__(self, i):
            return dict(images=fin['images'][i], depths=fin['depths'][i], seg_mask=fin['seg_mask'][i],
                        backmasks=fin['backmasks'][i], pose2d=fin['pose2d'][i], poses_smpl=fin['poses_smpl'][i],
                        betas_smpl=fin['betas_smpl'][i], valid_smpl=fin['valid_smpl'][i], idxs=i)

    names = ['poses_T', 'poses_smpl', 'betas_smpl', 'zmin_lin', 'zmax_lin', 'xscale_factor']

    def run_fit(k, scene):
        rec['depth'], rec['sil'], rec['mask_px'] = [], [], []
        opt = optim.SMPLDepthSequenceOptimizer(image_size=(fin['W'], fin['H']), num_frames=fin['T'], cam_K=fin['cam_K'],
                                               device='cpu', smpl_model_parameters_path=tmp, **coef_kw)
        opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=100)
        init = {n: getattr(opt, n).detach().numpy().copy() for n in names}
        if scene:
            opt.scene_depth = fin['scene_depth']
            opt.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
        dl = torch.utils.data.DataLoader(DS(), batch_size=5, shuffle=False)
        try:
            opt.fit(dl, num_iter=k)
        except UnboundLocalError:
            pass                                       # optimizer.py:595 quirk for num_iter <= 30
        leaves = {n: getattr(opt, n).detach().numpy().copy() for n in names}
        grads = {n: (getattr(opt, n).grad.numpy().copy() if getattr(opt, n).grad is not None else None) for n in names}
        return init, leaves, grads, dict(depth=np.array(rec['depth'], np.float32), sil=list(rec['sil']),
                                         mask_px=np.array(rec['mask_px'], np.float32))

    init, leaves, grads, r = run_fit(1, scene=False)
    for n, v in init.items():
        out['init_' + n] = v
    for n, v in grads.items():
        if v is not None:
            out['k1_grad_' + n] = v
    for n, v in leaves.items():
        out['k1_' + n] = v
    out['k1_loss_depth_per_batch'] = r['depth']                       # 4 batches of cycle 0
    out['k1_loss_sil_calls'] = np.array(r['sil'], np.float32)         # one entry per gated (frame, rank)
    out['k1_depth_mask_px'] = r['mask_px']
    print('cycle 0: loss_depth per batch', r['depth'], 'supervised px', r['mask_px'], 'sil calls', len(r['sil']),
          'sum', float(np.sum(r['sil'])))
    assert (r['mask_px'] > 20).all() and (r['depth'] > 0).all(), 'bodies must cover supervised pixels'

    # the rank-indexed gate must differ from a person-indexed gate somewhere in these inputs (:450, :472)
    conf = (fin['pose2d'][..., 2] >= 0.5).sum(-1) >= 2
    mval = fin['seg_mask'].sum(axis=(2, 3)) >= 0.005 * fin['H'] * fin['W']
    gate = (conf & mval)
    order = np.argsort(init['poses_T'][:, :, 0, 2], axis=1)
    differs = sum(int((gate[t] != gate[t][order[t]]).any()) for t in range(fin['T']))
    print('frames where the rank gate differs from the person gate:', differs)
    assert differs > 0
    out['k1_rank_gate_differs'] = np.int64(differs)

    _, leaves, _, r = run_fit(5, scene=False)
    for n, v in leaves.items():
        out['k5_' + n] = v
    out['k5_loss_depth_per_batch'] = r['depth']                       # 5 cycles x 4 batches
    out['k5_loss_sil_calls'] = np.array(r['sil'], np.float32)
    _, leaves, grads, r = run_fit(1, scene=True)
    for n, v in grads.items():
        if v is not None:
            out['scene_k1_grad_' + n] = v
    _, leaves, _, r = run_fit(5, scene=True)
    for n, v in leaves.items():
        out['scene_k5_' + n] = v
    out['scene_k5_loss_depth_per_batch'] = r['depth']

    # masked median over time with never-seen pixels, pixels seen by one / two frames and ties (fhsog.py:180-202)
    fhsog = importlib.import_module('refmh.fhsog')
    dn, back, imgs = gi.median_inputs()
    img, dep, msk = fhsog.aggegrate_scene_geometry_median((1.0 / (dn + 0.5)).astype(np.float32), imgs, back)
    out['median2_img'], out['median2_depth'], out['median2_mask'] = img, dep, msk

    # the numpy projection the evaluator / visualiser call with the MuPoTs distortion vector (transforms.py:19-54)
    transforms = importlib.import_module('refmh.transforms')
    pts, K, Kd = gi.projection_inputs()
    out['proj_np_plain'] = transforms.camera_projection(pts.reshape(-1, 3).copy(), K[0], return_depth=True)
    out['proj_np_dist'] = transforms.camera_projection(pts.reshape(-1, 3).copy(), K[0], Kd=Kd)
    out['unproj_np'] = transforms.camera_inverse_projection(out['proj_np_plain'].copy(), K[0])

    # every stub call was made with the settings the reference's constructor built (:211-225)
    assert set(CALLS['raster']) == {((fin['H'], fin['W']), 1e-4, 8)}, set(CALLS['raster'])
    assert set(CALLS['render']) == {((fin['H'], fin['W']), 2e-5, 4)}, set(CALLS['render'])
    path = os.path.join(HERE, 'reference_raster_cpu.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
