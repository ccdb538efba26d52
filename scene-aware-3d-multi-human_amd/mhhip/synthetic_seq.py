"""Synthetic MuPoTs-shaped per-frame inputs for tests, ``smoke()`` and ``bench.py``:
the real pre-processed MuPoTs inputs are not available offline.  Ground truth is produced by the
HIP path itself (LBS + render); the noise / mask / depth composition below is data synthesis, not
part of the optimisation path."""
import numpy as np
import torch

from . import engine, raster, synthetic


def make_sequence(model, num_people, num_frames, image_size, seed, cam_K=None, chunk=256, z_range=(3.0, 8.0),
                  render_frames=None):
    """Returns a dict with the arrays the reference's dataset yields (datautils.py:531-542) for the
    whole sequence, as numpy: pose2d (T,N,17,3), poses_smpl (T,N,72), betas_smpl (T,N,10),
    valid_smpl (T,N,1), seg_mask (T,N,H,W), depths (T,H,W), backmasks (T,H,W), images (T,H,W,3),
    plus cam_K and the ground-truth parameters.  ``render_frames=(f0, f1)``: the frame-sharded run -- the parameter
    tracks and key-points of the WHOLE sequence (they are what every rank is handed), images rendered only for the
    frames this rank owns (``seg_mask`` / ``depths`` / ``images`` / ``backmasks`` then have f1-f0 frames, see
    ``ShardDataset``)."""
    N, T = num_people, num_frames
    W, H = image_size
    dev = model.device
    K = synthetic.default_cam_K(image_size, 60.0) if cam_K is None else np.asarray(cam_K, np.float32)
    rng = np.random.RandomState(seed + 7)
    sp = synthetic.make_sequence_params(N, T, seed, z_range)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    betas = t(sp['betas_gt'])
    r0, r1 = (0, T) if render_frames is None else (int(render_frames[0]), int(render_frames[1]))
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    seg = np.zeros((r1 - r0, N, H, W), np.float32)
    depths = np.zeros((r1 - r0, H, W), np.float32)
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
        pose2d[f0:f1, :, :, :2] = uv.view(f1 - f0, N, 17, 2).cpu().numpy()
        if f1 <= r0 or f0 >= r1:
            continue
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
        a, b = max(f0, r0), min(f1, r1)
        depths[a - r0:b - r0] = ((disp - lo) / torch.clamp(hi - lo, min=1e-6)).cpu().numpy()[a - f0:b - f0]
        seg[a - r0:b - r0] = sg.cpu().numpy()[a - f0:b - f0]
    pose2d[..., :2] += rng.normal(0, 1.0, (T, N, 17, 2)).astype(np.float32)
    conf = rng.uniform(0.6, 1.0, (T, N, 17)).astype(np.float32)
    conf[rng.rand(T, N, 17) < 0.1] = 0.1
    pose2d[..., 2] = conf
    backmasks = (seg.sum(1) == 0).astype(np.int64)
    images = rng.randint(0, 255, (r1 - r0, H, W, 3)).astype(np.uint8)
    return dict(pose2d=pose2d, poses_smpl=sp['poses_init'], betas_smpl=sp['betas_init'], valid_smpl=sp['valid'],
                seg_mask=seg, depths=depths, backmasks=backmasks, images=images, cam_K=K, gt=sp, render_frames=(r0, r1))


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


class ShardDataset(torch.utils.data.Dataset):
    """Whole-sequence dataset of a frame-sharded run built with ``make_sequence(render_frames=(f0, f1))``: every rank
    iterates all frames (that is what ``predict.py`` does under torchrun), the image tensors of frames this rank does
    not own are a shared block of zeros (the optimiser only stages its own frames)."""

    def __init__(self, seq):
        self.s = seq
        self.r0, self.r1 = seq['render_frames']
        self._z = {k: np.zeros_like(seq[k][0]) for k in ('images', 'depths', 'seg_mask', 'backmasks')}

    def __len__(self):
        return self.s['pose2d'].shape[0]

    def __getitem__(self, i):
        s = self.s
        own = self.r0 <= i < self.r1
        g = lambda k: s[k][i - self.r0] if own else self._z[k]
        return dict(images=g('images'), depths=g('depths'), seg_mask=g('seg_mask'), backmasks=g('backmasks'), pose2d=s['pose2d'][i],
                    poses_smpl=s['poses_smpl'][i], betas_smpl=s['betas_smpl'][i], valid_smpl=s['valid_smpl'][i], idxs=i)
