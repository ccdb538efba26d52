"""Camera helpers with the call surface of reference ``mhmocap/transforms.py`` (the functions the
optimisation path and its direct callers use: :19-54, :57-95, :98-130, :222-255, :258-265, :296-306).
Host-side scalars are numpy; the per-point device work of the hot path is fused into the HIP kernels
(``mh_project_joints_loss``, ``mh_scene_unproject``, ``mh_raster_*``).

This module SHADOWS the reference's ``mhmocap.transforms`` in the namespace overlay: every
public name of the reference module that is not defined here (``batch_orthographic_projection``,
``transform_3dpoints``, ``recover_camera_intrinsics*``, ``disp_from_depth``, ``bounded_splus_exp*`` ... used by
``datautils.py:19``, ``evaluate.py:5``) is re-exported from it by ``mhhip._overlay.inherit`` at the bottom."""
import math

import numpy as np


def _hip():
    from mhhip import _lib
    return _lib


def get_focal(w, theta):
    return 0.5 * w / math.tan(math.pi * theta / 360.0)


def get_fov(w, f):
    return 360.0 * math.atan(0.5 * w / f) / math.pi


def softplus_np(x):
    return np.log(1.0 + np.exp(x))


def inverse_softplus_np(s):
    return np.log(np.exp(s) - 1.0)


def compute_calibration_matrix(znear, zfar, cam_K, image_size):
    """4x4 NDC projection for a (W, H) image: short side spans [-1, 1] (reference :222-255)."""
    W, H = image_size
    fx, fy, cx, cy = cam_K[0, 0], cam_K[1, 1], cam_K[0, 2], cam_K[1, 2]
    if W > H:
        s, u = 2 * fy / H, W / H
        w1, h1 = u * (W - 2 * cx) / W, (H - 2 * cy) / H
    elif H > W:
        s, u = 2 * fx / W, H / W
        w1, h1 = (W - 2 * cx) / W, u * (H - 2 * cy) / H
    else:
        s = 2 * (fx + fy) / (W + H)
        w1, h1 = (W - 2 * cx) / W, (H - 2 * cy) / H
    f1 = zfar / (zfar - znear)
    f2 = -(zfar * znear) / (zfar - znear)
    return np.array([[s, 0, w1, 0], [0, s, h1, 0], [0, 0, f1, f2], [0, 0, 1, 0]], np.float32)


def camera_projection(pts3d, K, return_depth=False, Kd=None):
    """numpy pinhole projection, pts3d (M,3), K (3,3), optional distortion Kd = [k1,k2,p1,p2,k3] (reference
    :19-54, called by evaluate.py:234,256 and predict.py:105-106).  Same tangential form as the torch version
    (:78-90), including its second term ``2*p2*y^2`` where OpenCV has ``2*p2*x*y``."""
    xy = pts3d[:, :2] / pts3d[:, 2:3]
    if Kd is not None:
        x, y = xy[:, 0].copy(), xy[:, 1].copy()
        r = x * x + y * y
        radial = 1 + Kd[0] * r + Kd[1] * r * r + Kd[4] * r * r * r
        xd = x * radial + 2 * Kd[2] * x * y + Kd[3] * (r + 2 * x * x)
        yd = y * radial + 2 * Kd[3] * y * y + Kd[2] * (r + 2 * y * y)      # (sic)
        xy = np.stack([xd, yd], axis=-1)
    uv = xy @ K[:2, :2].T + K[0:2, 2:3].T
    return np.concatenate([uv, pts3d[:, 2:]], -1) if return_depth else uv


def camera_inverse_projection(ptsuvd, K):
    xy = ptsuvd[:, 2:3] * ((ptsuvd[:, :2] - K[0:2, 2:3].T) @ np.linalg.inv(K[:2, :2].T))
    return np.concatenate([xy, ptsuvd[:, 2:3]], axis=-1)


def _on_device(t):
    """Host tensors are uploaded to the current HIP device, pushed through the same kernel and downloaded again: the
    overlay has ONE arithmetic path (the HIP library) and never executes the module it shadows."""
    import torch
    if t.is_cuda:
        return t, None
    if not torch.cuda.is_available():
        raise RuntimeError('mhmocap.transforms (MI355X build) needs a HIP device: there is no CPU path')
    return t.to('cuda'), t.device


def camera_projection_torch(pts3d, K, return_depth=False, Kd=None):
    """pts3d (N,M,3), K (N,3,3) device tensors -> (N,M,2) pixels (reference :57-95); forward only (the optimiser's
    differentiable projection is fused into ``mh_project_joints_loss``).  Host tensors make the round trip over the
    device (same kernel, result returned on the host)."""
    import torch
    pts3d, home = _on_device(pts3d)
    K = K.to(pts3d.device)
    L = _hip()
    p = pts3d.contiguous().float()
    Kc = K.contiguous().float()
    N, M = p.shape[0], p.shape[1]
    out = torch.empty(N, M, 3 if return_depth else 2, dtype=torch.float32, device=p.device)
    kd = None if Kd is None else np.ascontiguousarray(np.asarray(Kd, np.float32).reshape(5))
    L.check(L.lib().mh_project_points(N, M, L.ptr(p), L.ptr(Kc), None if kd is None else kd.ctypes.data_as(L.c_float_p),
                                      1 if return_depth else 0, L.ptr(out), L.stream_ptr(p.device)))
    return out if home is None else out.to(home)


def camera_inverse_projection_torch(ptsuvd, K):
    """ptsuvd (N,M,3) pixels + depth, K (N,3,3) -> camera-space points (reference :114-130)."""
    import torch
    ptsuvd, home = _on_device(ptsuvd)
    K = K.to(ptsuvd.device)
    L = _hip()
    p = ptsuvd.contiguous().float()
    Kc = K.contiguous().float()
    out = torch.empty_like(p)
    L.check(L.lib().mh_unproject_points(p.shape[0], p.shape[1], L.ptr(p), L.ptr(Kc), L.ptr(out), L.stream_ptr(p.device)))
    return out if home is None else out.to(home)


def softplus(x):
    """reference :296-297 (naive form); a one-liner on tensors, kept for call compatibility"""
    import torch
    return torch.log(1.0 + torch.exp(x))


def inverse_softplus(s):
    import torch
    return torch.log(torch.exp(s) - 1.0)


from mhhip._overlay import inherit as _inherit  # noqa: E402

_inherit(globals())
