"""ORACLE (test infrastructure only) -- CPU restatement of the two PyTorch3D renders used by
``fit`` (reference optimizer.py:211-232, 428-431, 447-448): the nearest-face z-buffer
(``fragments.zbuf[..., 0]`` of a K=8, blur 1e-4 rasterisation) and the soft silhouette
(K=4, blur 2e-5, ``SoftSilhouetteShader`` with BlendParams.sigma = 1e-4).

PARITY UNPINNED: PyTorch3D is an un-vendored, un-pinned third-party dependency of the reference
(environment.yml:13) that is not installable offline, so no output of it exists to pin against.
This file follows its published semantics (see raster_select.c) and is pinned by analytic
known-answer tests (tests/test_raster_oracle.py).

Discrete face selection is done by ``raster_select.c`` (built by oracle/Makefile); every
differentiable quantity is recomputed here in torch from the selected faces, so autograd yields
the gradients (the same split PyTorch3D makes between its forward and backward kernels).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

from . import fit_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
K_EPS = 1e-8


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, '_build', 'libraster_select.so')
        if not os.path.exists(so):
            subprocess.run(['make', '-C', _HERE], check=True, stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
        _LIB.raster_select.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def pixel_centres_ndc(H, W):
    """NDC coordinates of the pixel centres (PixToNonSquareNdc; +x left, +y up)."""
    def axis(S1, S2):
        r = 2.0 * S1 / S2 if S1 > S2 else 2.0
        i = np.arange(S1 - 1, -1, -1, dtype=np.float32)          # flipped index
        return (-np.float32(r / 2) + (np.float32(r) * i + np.float32(r / 2)) / np.float32(S1)).astype(np.float32)
    return axis(W, H), axis(H, W)


def to_ndc(verts, cam_K, image_size, znear=1.0, zfar=100.0):
    """Camera-space vertices -> PyTorch3D NDC xy + view z (MeshRasterizer.transform with
    R = diag(-1,-1,1), T = 0, K = compute_calibration_matrix(...); optimizer.py:204-210)."""
    Kndc = fit_oracle.calibration_matrix_ndc(znear, zfar, cam_K, image_size)
    s, w1, h1 = float(Kndc[0, 0]), float(Kndc[0, 2]), float(Kndc[1, 2])
    x, y, z = verts[..., 0], verts[..., 1], verts[..., 2]
    return torch.stack([s * (-x) / z + w1, s * (-y) / z + h1, z], dim=-1)


def set_behind_camera_rule(whole_face):
    """raster_select.c: 1 (default) = a face with a vertex behind the camera (zmin < 1e-8) is skipped entirely (the CUDA
    kernels' rule, what the HIP kernel implements), 0 = only its pixels with interpolated pz < 0 are (the plain naive CPU
    loop).  Returns the previous setting."""
    L = _lib()
    old = L.raster_select_get_behind_camera_rule()
    L.raster_select_set_behind_camera_rule(1 if whole_face else 0)
    return old


def select_faces(verts_ndc, faces, H, W, blur_radius, K):
    """(B,V,3) float32 NDC verts -> pix_to_face (B,H,W,K) int64 (-1 empty)."""
    v = np.ascontiguousarray(verts_ndc, dtype=np.float32)
    f = np.ascontiguousarray(faces, dtype=np.int32)
    out_f = np.empty((v.shape[0], H, W, K), np.int32)
    out_z = np.empty((v.shape[0], H, W, K), np.float32)
    L = _lib()

    def one(b):
        L.raster_select(v[b].ctypes.data, f.ctypes.data, f.shape[0], H, W, ctypes.c_float(blur_radius), K,
                        out_f[b].ctypes.data, out_z[b].ctypes.data)
    # the C selection is single-threaded per body; bodies are independent and ctypes releases the GIL, so the bodies of
    # a batch are spread over ORACLE_THREADS host threads (default: torch's thread count) -- same results, any order
    nthr = int(os.environ.get('ORACLE_THREADS', '0')) or torch.get_num_threads()
    if nthr > 1 and v.shape[0] > 1:
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(nthr, v.shape[0])) as ex:
            list(ex.map(one, range(v.shape[0])))
    else:
        for b in range(v.shape[0]):
            one(b)
    return out_f.astype(np.int64), out_z


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _seg_d2(px, py, ax, ay, bx, by):
    bax, bay = bx - ax, by - ay
    l2 = bax * bax + bay * bay
    t = (bax * (px - ax) + bay * (py - ay)) / torch.clamp(l2, min=1e-30)
    t = torch.clamp(t, 0.0, 1.0).detach()            # PointLineDistanceBackward treats the clamped t as constant
    qx, qy = ax + t * bax - px, ay + t * bay - py
    d = qx * qx + qy * qy
    deg = (px - bx) ** 2 + (py - by) ** 2
    return torch.where(l2 <= K_EPS, deg, d)


def fragments(verts_ndc, faces, pix_to_face, H, W):
    """Differentiable per-(pixel, k) z (clipped barycentric interpolation) and signed squared
    distance for the selected faces.  verts_ndc (B,V,3) torch, pix_to_face (B,H,W,K).
    Only the occupied (pixel, k) entries are evaluated (a body covers a few percent of the image)."""
    xs, ys = pixel_centres_ndc(H, W)
    p2f = torch.as_tensor(pix_to_face)
    valid = p2f >= 0
    bi, yi, xi, ki = torch.nonzero(valid, as_tuple=True)
    dt = verts_ndc.dtype
    px = torch.tensor(xs, dtype=dt)[xi]
    py = torch.tensor(ys, dtype=dt)[yi]
    ft = torch.as_tensor(np.asarray(faces, np.int64))
    tri = ft[p2f[bi, yi, xi, ki]]                                # (M,3)
    v = verts_ndc[bi[:, None], tri]                              # (M,3,3)
    x0, y0, z0 = v[:, 0, 0], v[:, 0, 1], v[:, 0, 2]
    x1, y1, z1 = v[:, 1, 0], v[:, 1, 1], v[:, 1, 2]
    x2, y2, z2 = v[:, 2, 0], v[:, 2, 1], v[:, 2, 2]
    area = _edge(x2, y2, x0, y0, x1, y1) + K_EPS
    w0 = _edge(px, py, x1, y1, x2, y2) / area
    w1 = _edge(px, py, x2, y2, x0, y0) / area
    w2 = _edge(px, py, x0, y0, x1, y1) / area
    inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
    c0, c1, c2 = torch.clamp(w0, min=0), torch.clamp(w1, min=0), torch.clamp(w2, min=0)
    cs = torch.clamp(c0 + c1 + c2, min=1e-5)
    pz = (c0 / cs) * z0 + (c1 / cs) * z1 + (c2 / cs) * z2
    d01, d02, d12 = _seg_d2(px, py, x0, y0, x1, y1), _seg_d2(px, py, x0, y0, x2, y2), _seg_d2(px, py, x1, y1, x2, y2)
    # PointTriangleDistanceBackward's edge choice: e01 first, then e02, then e12
    pick01 = (d01 <= d02) & (d01 <= d12)
    pick02 = (~pick01) & (d02 <= d01) & (d02 <= d12)
    dist = torch.where(pick01, d01, torch.where(pick02, d02, d12))
    sdist = torch.where(inside, -dist, dist)
    shape = tuple(p2f.shape)
    zout = torch.full(shape, -1.0, dtype=dt).index_put((bi, yi, xi, ki), pz)
    dout = torch.full(shape, -1.0, dtype=dt).index_put((bi, yi, xi, ki), sdist)
    return zout, dout, valid


def render(verts, faces, cam_K, image_size, znear=1.0, zfar=100.0, sigma=1e-4, selection=None):
    """verts (B,V,3) camera space -> zbuf0 (B,H,W) [-1 where empty], alpha (B,H,W).  selection: (f1 (B,H,W,1), f4
    (B,H,W,4)) pix_to_face arrays to use INSTEAD of the brute-force selection (tests hand in the selection of the HIP
    kernel to separate "which faces" from "what comes out of them")."""
    W, H = image_size
    ndc = to_ndc(verts, cam_K, image_size, znear, zfar)
    ndc_np = ndc.detach().numpy().astype(np.float32)
    if selection is not None:
        f8, f4 = np.asarray(selection[0], np.int64), np.asarray(selection[1], np.int64)
    else:
        f8, _ = select_faces(ndc_np, faces, H, W, 1e-4, 8)                      # optimizer.py:211-215
    z8, _, _ = fragments(ndc, faces, f8[..., :1], H, W)
    zbuf0 = z8[..., 0]                                                          # optimizer.py:430
    if selection is None:
        f4, _ = select_faces(ndc_np, faces, H, W, 2e-5, 4)                      # optimizer.py:221-225
    _, sd, valid = fragments(ndc, faces, f4, H, W)
    prob = torch.sigmoid(-sd / sigma) * valid.to(sd.dtype)                      # SoftSilhouetteShader / sigmoid_alpha_blend
    alpha = 1.0 - torch.prod(1.0 - prob, dim=-1)
    return zbuf0, alpha


def make_rasteriser(faces, cam_K, image_size, znear=1.0, zfar=100.0):
    return lambda verts: render(verts, faces, cam_K, image_size, znear, zfar)
