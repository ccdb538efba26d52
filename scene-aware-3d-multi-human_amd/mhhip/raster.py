"""Binding of the fused rasteriser + depth / silhouette residual kernel (``mh_raster_terms``)."""
import ctypes
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
        self.dims = (e.T, e.N, e.V, int(self.faces.shape[0]), e.H, e.W)
        self.init_workspace()

    def init_workspace(self):
        """mh_raster_workspace_init: once per workspace (again after its bytes were overwritten)"""
        check(_lib.lib().mh_raster_workspace_init(*self.dims, ptr(self.ws), _lib.stream_ptr(self.dev)))

    def forward_targets(self):
        """mh_fwd_proj of this workspace: what ``SequenceEngine.forward`` hands to mh_lbs_forward_proj so that the skinning
        epilogue projects the vertices for this rasteriser (the motion threshold follows the current sort margin)."""
        import ctypes
        t = _lib.FwdProj()
        check(_lib.lib().mh_raster_forward_targets(*self.dims, self.K.ctypes.data_as(_lib.c_float_p), ptr(self.ws), ctypes.byref(t)))
        return t

    def __call__(self, e, gverts, log, with_grads=True, zbuf_out=None, alpha_out=None, phases=3, defer=False):
        """phases: 1 = selection + values (does not touch gverts), 2 = gradients + log entries, 3 = both.
        defer (with phases & 2): the closing kernel is NOT launched; its job is returned (a ``_lib.RasterFin``) for the
        caller's LBS backward to carry out (mh_lbs_backward_kp_fin)."""
        L = _lib.lib()
        st = _lib.stream_ptr(e.dev)
        g = e.grads
        # the engine's last forward has projected the vertices into this workspace: no pass over them here.  A ONE-SHOT token
        # (ADVICE r04): the launch that runs the preparation consumes it, so that a caller who writes e.verts afterwards -- or
        # launches the rasteriser a second time -- gets the projection from the vertices as they are, not the forward's
        projected = 1 if getattr(e, '_projected_into', None) is self else 0
        if int(phases) & (1 | 4):
            e._projected_into = None
            self.last_projected = projected
        args = (e.T, e.N, e.V, self.faces.shape[0], e.H, e.W, self.K.ctypes.data_as(_lib.c_float_p),
                ptr(e.verts), ptr(self.faces), ptr(e.bits), ptr(e.ebits), ptr(e.depths),
                ptr(e.leaf('zmin_lin')), ptr(e.leaf('zmax_lin')), ptr(e.p2d_valid), ptr(e.front),
                ptr(e.sil_apply), ptr(e.sil_D), ptr(e.sil_S), float(e.c['depth']),
                float(e.c['silhouette']), float(e.eps), ptr(gverts) if with_grads else None,
                ptr(e.leaf('zmin_lin', g)) if with_grads else None,
                ptr(e.leaf('zmax_lin', g)) if with_grads else None, ptr(e.depth_body), ptr(e.sil_body),
                ptr(self.ws), ptr(zbuf_out), ptr(alpha_out), int(phases), ptr(log[1:2]), ptr(log[2:3]), projected)
        if defer:
            fin = _lib.RasterFin()
            check(L.mh_raster_terms_deferred(*args, ctypes.byref(fin), st))
            return fin
        check(L.mh_raster_terms_projected(*args, st))
        return None

    def sort_counters(self, e):
        """(bodies seen, bodies whose face lists were re-sorted), cumulative over the launches on this workspace"""
        import ctypes
        out = (ctypes.c_ulonglong * 2)()
        check(_lib.lib().mh_raster_sort_counters(*self.dims, ptr(self.ws), out, _lib.stream_ptr(e.dev)))
        return int(out[0]), int(out[1])

    def sort_counters3(self, e):
        """(bodies seen, bodies re-sorted, of which beside the gradient kernel: off the chain), cumulative"""
        import ctypes
        out = (ctypes.c_ulonglong * 3)()
        check(_lib.lib().mh_raster_sort_counters3(*self.dims, ptr(self.ws), out, _lib.stream_ptr(e.dev)))
        return int(out[0]), int(out[1]), int(out[2])

    def pair_counters(self, e):
        """(launches, candidate pairs, evaluated pairs) of the selection kernel, counted while mh_profile_enable(1)"""
        import ctypes
        out = (ctypes.c_ulonglong * 3)()
        check(_lib.lib().mh_raster_pair_counters(*self.dims, ptr(self.ws), out, _lib.stream_ptr(e.dev)))
        return int(out[0]), int(out[1]), int(out[2])

    def selection(self, e):
        """Inspection aid: what the last selection pass left in the workspace -- (win (B,4) int32: x0, y0, width, height of
        every body's screen window; koff (B+1,): first window pixel of every body; keys (window pixels, 5) uint64: per
        pixel, row-major inside the window, slot 0 = nearest face of the blur-1e-4 pass, slots 1-4 = the K=4 list of the
        blur-2e-5 pass, ascending; key = float bits of z << 32 | face, all ones = empty)."""
        import ctypes
        B = e.B
        off = (ctypes.c_size_t * 3)()
        check(_lib.lib().mh_raster_workspace_offsets(*self.dims, off))
        win = self.ws[off[0]:off[0] + B * 16].view(torch.int32).view(B, 4).cpu().numpy()
        first = self.ws[off[1]:off[1] + B * 8].view(torch.int64).cpu().numpy()
        npix = np.maximum(win[:, 2], 0).astype(np.int64) * np.maximum(win[:, 3], 0)
        total = int(npix.sum())
        # every body's keys lie in a region of their own (first[b]): gathered into body order on the device
        koff = np.concatenate([[0], np.cumsum(npix)])
        src = np.concatenate([first[b] + np.arange(npix[b], dtype=np.int64) for b in range(B)]) if total else np.zeros(0, np.int64)
        raw = self.ws[off[2]:off[2] + B * e.H * e.W * 40].view(torch.int64).view(-1, 5)
        keys = raw[torch.as_tensor(src, device=self.ws.device)].cpu().numpy().view(np.uint64).reshape(-1, 5)
        return win, koff, keys


def set_sort_margin(rows):
    """mh_raster_set_sort_margin (0 = sort the face lists every launch); returns the previous setting"""
    L = _lib.lib()
    old = L.mh_raster_get_sort_margin()
    check(L.mh_raster_set_sort_margin(int(rows)))
    return int(old)


def set_sort_defer(fraction):
    """mh_raster_set_sort_defer (fraction of the margin from which a body is re-sorted beside the gradient kernel; 0 = never);
    returns the previous setting"""
    L = _lib.lib()
    old = float(L.mh_raster_get_sort_defer())
    check(L.mh_raster_set_sort_defer(float(fraction)))
    return old


def set_winners(on):
    """mh_raster_set_winners: winners' list of the face sort on / off (same keys either way); returns the previous setting"""
    L = _lib.lib()
    old = L.mh_raster_get_winners()
    check(L.mh_raster_set_winners(1 if on else 0))
    return bool(old)


def set_deterministic(on):
    """mh_raster_set_deterministic: bit-reproducible gradient scatter (64-bit fixed-point accumulation, one workgroup per
    body) instead of fp32 atomics; returns the previous setting"""
    L = _lib.lib()
    old = L.mh_raster_get_deterministic()
    check(L.mh_raster_set_deterministic(1 if on else 0))
    return bool(old)


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
    check(_lib.lib().mh_raster_workspace_init(B, 1, V, int(faces.shape[0]), H, W, ptr(ws), _lib.stream_ptr(dev)))
    check(_lib.lib().mh_raster_terms(B, 1, V, faces.shape[0], H, W, K.ctypes.data_as(_lib.c_float_p), ptr(verts.contiguous()),
                                     ptr(faces), ptr(bits), ptr(bits), ptr(depths), ptr(tz), ptr(tz), ptr(ones), ptr(zi(B)),
                                     ptr(tz), ptr(ones), ptr(tz), 0.0, 0.0, 1e-3, None, None, None, ptr(zf(B)), ptr(zf(B)),
                                     ptr(ws), ptr(zbuf), ptr(alpha), _lib.stream_ptr(dev)))
    return zbuf, alpha
