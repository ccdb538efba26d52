"""Synthetic MuPoTs-shaped per-frame inputs for tests, ``smoke()`` and ``bench.py``:
the real pre-processed MuPoTs inputs are not available offline.  Ground truth is produced by the
HIP path itself (LBS + render); the noise / mask / depth composition below is data synthesis, not
part of the optimisation path."""
import numpy as np
import torch

from . import engine, raster, synthetic


def make_sequence(model, num_people, num_frames, image_size, seed, cam_K=None, chunk=256, z_range=(3.0, 8.0)):
    """Returns a dict with the arrays the reference's dataset yields (datautils.py:531-542) for the
    whole sequence, as numpy: pose2d (T,N,17,3), poses_smpl (T,N,72), betas_smpl (T,N,10),
    valid_smpl (T,N,1), seg_mask (T,N,H,W), depths (T,H,W), backmasks (T,H,W), images (T,H,W,3),
    plus cam_K and the ground-truth parameters."""
    N, T = num_people, num_frames
    W, H = image_size
    dev = model.device
    K = synthetic.default_cam_K(image_size, 60.0) if cam_K is None else np.asarray(cam_K, np.float32)
    rng = np.random.RandomState(seed + 7)
    sp = synthetic.make_sequence_params(N, T, seed, z_range)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    betas = t(sp['betas_gt'])
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    seg = np.zeros((T, N, H, W), np.float32)
    depths = np.zeros((T, H, W), np.float32)
    # background: ground plane y = 1.15 m (y points down) and a back wall at z = 10 m
    ys = (np.arange(H, dtype=np.float32) + 0.5 - K[1, 2]) / K[1, 1]
    ray_y = np.tile(ys[:, None], (1, W))
    bg = np.where(ray_y > 1e-3, 1.15 / np.maximum(ray_y, 1e-3), 10.0).astype(np.float32)
    bg = np.minimum(bg, 10.0)
    bg_t = t(bg)
    fchunk = max(1, chunk // N)
    for f0 in range(0, T, fchunk):
        f1 = min(T, f0 + fchunk)
        nb = (f1 - f0) * N
        poses = t(sp['poses_gt'][f0:f1]).view(nb, 72)
        tr = t(sp['trans_gt'][f0:f1]).view(nb, 3)
        verts, _, _, _ = model.lbs_forward(betas, poses, None, tr, want_vposed=False)
        kp = model.joints_regress(engine.REG_ALPHAPOSE, verts, corr=tr)
        uv, _, _ = engine.project_joints_loss(kp, K, None, torch.zeros(nb, 17, 3, device=dev), 0.5, 0, W, H)
        zbuf, alpha = raster.render(model, verts, K, image_size)
        zb = zbuf.view(f1 - f0, N, H, W)
        al = alpha.view(f1 - f0, N, H, W)
        zfar = torch.where((zb > 0) & (al > 0.5), zb, torch.full_like(zb, 1e9))
        nearest = torch.argmin(zfar, dim=1, keepdim=True)
        covered = zfar.gather(1, nearest) < 1e8
        sg = torch.zeros_like(zb).scatter_(1, nearest, covered.float())
        body_z = torch.where(covered[:, 0], zfar.gather(1, nearest)[:, 0], torch.full_like(bg_t, 1e9)[None].expand(f1 - f0, H, W))
        depth = torch.minimum(body_z, bg_t[None])
        disp = 1.0 / depth
        lo, hi = disp.amin(dim=(1, 2), keepdim=True), disp.amax(dim=(1, 2), keepdim=True)
        depths[f0:f1] = ((disp - lo) / torch.clamp(hi - lo, min=1e-6)).cpu().numpy()
        seg[f0:f1] = sg.cpu().numpy()
        pose2d[f0:f1, :, :, :2] = uv.view(f1 - f0, N, 17, 2).cpu().numpy()
    pose2d[..., :2] += rng.normal(0, 1.0, (T, N, 17, 2)).astype(np.float32)
    conf = rng.uniform(0.6, 1.0, (T, N, 17)).astype(np.float32)
    conf[rng.rand(T, N, 17) < 0.1] = 0.1
    pose2d[..., 2] = conf
    backmasks = (seg.sum(1) == 0).astype(np.int64)
    images = rng.randint(0, 255, (T, H, W, 3)).astype(np.uint8)
    return dict(pose2d=pose2d, poses_smpl=sp['poses_init'], betas_smpl=sp['betas_init'], valid_smpl=sp['valid'],
                seg_mask=seg, depths=depths, backmasks=backmasks, images=images, cam_K=K, gt=sp)


class SequenceDataset(torch.utils.data.Dataset):
    """Yields the per-frame dict of the reference's dataset (datautils.py:531-542)."""

    def __init__(self, seq):
        self.s = seq

    def __len__(self):
        return self.s['pose2d'].shape[0]

    def __getitem__(self, i):
        s = self.s
        return dict(images=s['images'][i], depths=s['depths'][i], seg_mask=s['seg_mask'][i], backmasks=s['backmasks'][i],
                    pose2d=s['pose2d'][i], poses_smpl=s['poses_smpl'][i], betas_smpl=s['betas_smpl'][i],
                    valid_smpl=s['valid_smpl'][i], idxs=i)
