"""Seeded inputs shared by ``make_golden.py`` (which feeds them to the reference, in the build
container only) and by the tests (which feed them to the oracle and to the HIP path).
Only numbers derived from these inputs are committed as fixtures."""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, 'scene-aware-3d-multi-human_amd'))

from mhhip import synthetic  # noqa: E402


def lbs_inputs(num=6, seed=7):
    rng = np.random.RandomState(seed)
    betas = rng.normal(0, 0.8, (num, 10)).astype(np.float32)
    poses = rng.normal(0, 0.35, (num, 72)).astype(np.float32)
    poses[0] = 0.0                       # exercises the eps-shifted norm at the zero vector
    poses[1, 3:6] = [np.pi, 0, 0]        # |r| = pi
    poses[2, 6:9] = [1e-6, -2e-6, 3e-7]  # tiny angle
    return betas, poses


def rodrigues_inputs():
    rng = np.random.RandomState(11)
    r = rng.normal(0, 1.0, (32, 3)).astype(np.float32)
    r[0] = 0
    r[1] = [np.pi, 0, 0]
    r[2] = [0, 1e-7, 0]
    r[3] = [2.0, -2.0, 2.0]
    return r


def projection_inputs():
    rng = np.random.RandomState(12)
    pts = rng.normal(0, 1.0, (5, 17, 3)).astype(np.float32)
    pts[..., 2] = np.abs(pts[..., 2]) + 2.0
    K = np.tile(np.array([[[117.0, 0.3, 120.0], [0.1, 118.5, 67.5], [0, 0, 1]]], np.float32), (5, 1, 1))
    Kd = np.array([-0.12, 0.05, 1e-3, -2e-3, 0.01], np.float32)
    return pts, K, Kd


def image_loss_inputs():
    rng = np.random.RandomState(13)
    b, N, H, W = 3, 2, 20, 28
    pred = rng.uniform(0.05, 1.0, (b, N, H, W)).astype(np.float32)
    pred[0, 0, :2] = 1e-4                      # below the clamp
    true = rng.uniform(0.05, 1.0, (b, 1, H, W)).astype(np.float32)
    mask = (rng.rand(b, N, H, W) > 0.5).astype(np.float32)
    mask[2, 1] = 0
    return pred, true, mask


def erode_inputs():
    rng = np.random.RandomState(14)
    x = (rng.rand(3, 1, 24, 31) > 0.25).astype(np.float32)
    x[0, 0, 4:18, 5:25] = 1
    x[1, 0, :, :3] = 1                          # blob touching the border
    return x


def one_euro_inputs():
    rng = np.random.RandomState(15)
    T = 40
    x = np.cumsum(rng.normal(0, 0.02, (T, 3, 5)), axis=0).astype(np.float32) + rng.normal(0, 0.005, (T, 3, 5)).astype(np.float32)
    return x


def fit_inputs(T=20, N=2, H=32, W=48, seed=21):
    """A small sequence for the stubbed-raster ``fit`` fixtures."""
    rng = np.random.RandomState(seed)
    sp = synthetic.make_sequence_params(N, T, seed)
    cam_K = synthetic.default_cam_K((W, H), 60.0)
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    pose2d[..., 0] = rng.uniform(0, W, (T, N, 17))
    pose2d[..., 1] = rng.uniform(0, H, (T, N, 17))
    pose2d[..., 2] = rng.uniform(0.6, 1.0, (T, N, 17))
    pose2d[..., 2][rng.rand(T, N, 17) < 0.1] = 0.1
    pose2d[3, 1, :, 2] = 0.1                   # a body without valid 2D pose
    seg = np.zeros((T, N, H, W), np.float32)
    for t in range(T):
        for n in range(N):
            x0 = int(rng.uniform(2, W - 14)); y0 = int(rng.uniform(2, H - 16))
            seg[t, n, y0:y0 + 12 + n, x0:x0 + 8 + 2 * n] = 1
    seg[5, 0] = 0                               # an empty mask (mask_valid = 0)
    depths = rng.uniform(0, 1, (T, H, W)).astype(np.float32)
    images = rng.randint(0, 255, (T, H, W, 3)).astype(np.uint8)
    backmasks = (seg.sum(1) == 0).astype(np.int64)
    scene_depth = (0.9 + 0.01 * np.arange(W)[None] + 0.02 * np.arange(H)[:, None]).astype(np.float32)
    scene_mask = (rng.rand(H, W) > 0.1)
    return dict(T=T, N=N, H=H, W=W, cam_K=cam_K, pose2d=pose2d, seg_mask=seg, depths=depths,
                images=images, backmasks=backmasks, poses_smpl=sp['poses_init'],
                betas_smpl=sp['betas_init'], valid_smpl=sp['valid'], trans_gt=sp['trans_gt'],
                scene_depth=scene_depth, scene_mask=scene_mask)


def median_inputs(T=23, H=12, W=20, seed=51):
    """Normalised disparity, background masks and colour frames for the scene-median fixture: ties, a pixel no frame
    sees, pixels one / two / an even number of frames see."""
    rng = np.random.RandomState(seed)
    dn = rng.uniform(0, 1, (T, H, W)).astype(np.float32)
    dn[:, :2] = np.round(dn[:, :2] * 4) / 4
    back = (rng.uniform(0, 1, (T, H, W)) > 0.4).astype(np.int64)
    back[:, 0, 0] = 0
    back[:, 5:7, 8:11] = 0
    back[1:, 0, 1] = 0; back[0, 0, 1] = 1
    back[2:, 0, 2] = 0; back[:2, 0, 2] = 1
    back[4:, 0, 3] = 0; back[:4, 0, 3] = 1
    imgs = rng.randint(0, 255, (T, H, W, 3)).astype(np.uint8)
    return dn, back, imgs


COEFS = dict(proj2d=1.0, depth=0.05, silhouette=0.1, reg_poses=0.002, reg_scales=1e-4,
             reg_velocity=0.05, reg_verts_filter=0.002, reg_contact=0.001, reg_foot_sliding=0.01)


def fit_raster_inputs(gr):
    """Inputs of the raster-pinned ``fit`` fixtures (tests/golden/make_golden_raster.py).  They were rendered
    once in the build container and are stored in the fixture file itself (``in_*`` arrays of ``gr``)."""
    T, N, H, W = [int(v) for v in gr['in_dims']]
    return dict(T=T, N=N, H=H, W=W, cam_K=gr['in_cam_K'], pose2d=gr['in_pose2d'],
                seg_mask=gr['in_seg_mask'].astype(np.float32), depths=gr['in_depths'], images=gr['in_images'],
                backmasks=gr['in_backmasks'].astype(np.int64), poses_smpl=gr['in_poses_smpl'],
                betas_smpl=gr['in_betas_smpl'], valid_smpl=gr['in_valid_smpl'], trans_gt=gr['in_trans_gt'],
                scene_depth=gr['in_scene_depth'], scene_mask=gr['in_scene_mask'] > 0)
