"""Binding of the fused rasteriser + depth / silhouette residual kernel (``mh_raster_terms``)."""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


@_lib.on_own_device
class RasterTerms(object):
    """Callable handed to ``SequenceEngine.cycle``: adds the depth and silhouette terms of all
    local frames (reference optimizer.py:425-477) to ``gverts`` / the depth-range gradients and
    writes the two loss sums into the cycle's log row."""

    def __init__(self, engine, znear=1.0, zfar=100.0):
        e = engine
        self.dev = e.dev
        self.faces = torch.as_tensor(np.ascontiguousarray(np.asarray(e.m.faces).astype(np.int32))).to(e.dev)
        self.ws = torch.empty(_lib.lib().mh_raster_workspace_bytes(e.T, e.N, e.V, self.faces.shape[0], e.H, e.W), dtype=torch.uint8, device=e.dev)
        self.K = np.ascontiguousarray(e.K.reshape(9))

    def __call__(self, e, gverts, log, with_grads=True, zbuf_out=None, alpha_out=None, phases=3):
        """phases: 1 = selection + values (does not touch gverts), 2 = gradients + log entries, 3 = both"""
        L = _lib.lib()
        st = _lib.stream_ptr(e.dev)
        g = e.grads
        check(L.mh_raster_terms_phase(e.T, e.N, e.V, self.faces.shape[0], e.H, e.W, self.K.ctypes.data_as(_lib.c_float_p),
                                      ptr(e.verts), ptr(self.faces), ptr(e.bits), ptr(e.ebits), ptr(e.depths),
                                      ptr(e.leaf('zmin_lin')), ptr(e.leaf('zmax_lin')), ptr(e.p2d_valid), ptr(e.front),
                                      ptr(e.sil_apply), ptr(e.sil_D), ptr(e.sil_S), float(e.c['depth']),
                                      float(e.c['silhouette']), float(e.eps), ptr(gverts) if with_grads else None,
                                      ptr(e.leaf('zmin_lin', g)) if with_grads else None,
                                      ptr(e.leaf('zmax_lin', g)) if with_grads else None, ptr(e.depth_body), ptr(e.sil_body),
                                      ptr(self.ws), ptr(zbuf_out), ptr(alpha_out), int(phases), st))
        if phases & 2:
            check(L.mh_reduce_sum2(ptr(e.depth_body), ptr(e.sil_body), e.B, 1.0, ptr(log[1:2]), ptr(log[2:3]), st))


def render(model, verts, cam_K, image_size):
    """Nearest-face depth (-1 = empty) and soft-silhouette images of B bodies: (B,H,W) each.
    Inspection / synthetic-data helper on top of ``mh_raster_terms`` (losses disabled)."""
    W, H = int(image_size[0]), int(image_size[1])
    B, V = verts.shape[0], verts.shape[1]
    dev = verts.device
    faces = torch.as_tensor(np.ascontiguousarray(np.asarray(model.faces).astype(np.int32))).to(dev)
    zi = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
    zf = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    bits, depths, tz, ones = zi(B, H, W), zf(B, H, W), zf(B), torch.ones(B, device=dev)
    zbuf, alpha = torch.empty(B, H, W, device=dev), torch.empty(B, H, W, device=dev)
    K = np.ascontiguousarray(np.asarray(cam_K, np.float32).reshape(9))
    ws = torch.empty(_lib.lib().mh_raster_workspace_bytes(B, 1, V, faces.shape[0], H, W), dtype=torch.uint8, device=dev)
    check(_lib.lib().mh_raster_terms(B, 1, V, faces.shape[0], H, W, K.ctypes.data_as(_lib.c_float_p), ptr(verts.contiguous()),
                                     ptr(faces), ptr(bits), ptr(bits), ptr(depths), ptr(tz), ptr(tz), ptr(ones), ptr(zi(B)),
                                     ptr(tz), ptr(ones), ptr(tz), 0.0, 0.0, 1e-3, None, None, None, ptr(zf(B)), ptr(zf(B)),
                                     ptr(ws), ptr(zbuf), ptr(alpha), _lib.stream_ptr(dev)))
    return zbuf, alpha
