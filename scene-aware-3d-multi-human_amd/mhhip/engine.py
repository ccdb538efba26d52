"""Torch-facing wrappers over the C ABI: device memory, streams and shapes only -- every
floating-point operation of the path happens in the HIP library."""
import ctypes
import pickle

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

H36M_ROW_ORDER = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]   # reference smpl.py:242
REG_ALPHAPOSE, REG_H36M17, REG_MUPOTS, REG_EXTRA9 = 0, 1, 2, 3
NUM_REG_JOINTS = {0: 17, 1: 17, 2: 17, 3: 9}


class _ChStub(object):
    """Stand-in for chumpy.Ch objects inside the official SMPL pickle (chumpy is not needed)."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'x': state})

    def __array__(self, dtype=None, copy=None):
        x = self.__dict__.get('x', None)
        if x is None:
            raise ValueError('unsupported chumpy object in the SMPL pickle')
        return np.array(x, dtype=dtype) if copy else np.asarray(x, dtype=dtype)


class _SmplUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('chumpy'):
            return _ChStub
        return super().find_class(module, name)


def load_smpl_pickle(path):
    """The fields the reference reads from SMPL_{GENDER}.pkl (smpl.py:187-188) without chumpy."""
    with open(path, 'rb') as f:
        d = _SmplUnpickler(f, encoding='latin1').load()
    return d


def _dense(a, dtype=np.float32):
    if 'scipy.sparse' in str(type(a)):
        a = a.todense()
    return np.ascontiguousarray(np.array(a, dtype=dtype))


@_lib.on_own_device
class BodyModel(object):
    """Device-resident SMPL constants (an ``mh_model``)."""

    def __init__(self, struct, regs=None, device='cuda:0'):
        if not torch.cuda.is_available():
            raise _lib.MhError('no HIP device: the MI355X path cannot run (there is no CPU fallback)')
        self.device = torch.device(device)
        get = (lambda k: struct[k]) if isinstance(struct, dict) else (lambda k: getattr(struct, k))
        self.v_template = _dense(get('v_template'))
        V = self.v_template.shape[0]
        shapedirs = _dense(get('shapedirs'))[:, :, :10].copy()
        posedirs = _dense(get('posedirs'))
        assert posedirs.shape == (V, 3, 207), posedirs.shape
        J_regressor = _dense(get('J_regressor'))
        weights = _dense(get('weights'))
        parents = np.asarray(get('kintree_table'))[0].astype(np.int64)
        parents[0] = -1
        parents = parents.astype(np.int32)
        self.faces = np.asarray(get('f')).astype(np.int64)
        faces32 = np.ascontiguousarray(self.faces.astype(np.int32))
        regs = regs or {}
        self.has_reg = {}
        keep = [self.v_template, shapedirs, posedirs, J_regressor, weights, parents, faces32]
        h = _lib.ModelHost()
        h.num_verts, h.num_faces = V, faces32.shape[0]
        fp = lambda a: a.ctypes.data_as(_lib.c_float_p)
        h.v_template, h.shapedirs, h.posedirs = fp(self.v_template), fp(shapedirs), fp(posedirs)
        h.J_regressor, h.lbs_weights = fp(J_regressor), fp(weights)
        h.parents = parents.ctypes.data_as(_lib.c_int_p)
        h.faces = faces32.ctypes.data_as(_lib.c_int_p)
        for key, field, which in [('alphapose', 'reg_alphapose', REG_ALPHAPOSE), ('h36m', 'reg_h36m17', REG_H36M17),
                                  ('mupots', 'reg_mupots', REG_MUPOTS), ('extra9', 'reg_extra9', REG_EXTRA9)]:
            a = regs.get(key)
            self.has_reg[which] = a is not None
            if a is None:
                continue
            a = _dense(a)
            if key in ('alphapose', 'mupots'):
                a = np.ascontiguousarray(a.T)            # files are (V,17): smpl.py:250,257
            if key == 'h36m':
                a = np.ascontiguousarray(a[H36M_ROW_ORDER])   # smpl.py:243
            assert a.shape == (NUM_REG_JOINTS[which], V), (key, a.shape)
            keep.append(a)
            setattr(h, field, fp(a))
        self.V, self.F = V, faces32.shape[0]
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().mh_model_create(ctypes.byref(handle), ctypes.byref(h)))
        self.handle = handle
        del keep
        self._self_check()

    _CHECKED = set()
    _FAILED = set()
    _SHARED = {}

    @staticmethod
    def _fingerprint(struct, regs, device):
        """content hash of everything the constructor reads (19 MB of model arrays: ~2 ms with xxhash, ~8 ms with crc32)"""
        try:
            import xxhash
            h = xxhash.xxh64()
            upd, done = h.update, h.hexdigest
        except ImportError:
            import zlib
            crc = [0]

            def upd(b):
                crc[0] = zlib.crc32(b, crc[0])
            done = lambda: '%08x' % crc[0]
        for name in sorted(vars(struct)):
            a = getattr(struct, name)
            a = a.toarray() if hasattr(a, 'toarray') else a
            try:
                a = np.ascontiguousarray(np.asarray(a))
            except Exception:
                continue
            if a.dtype == object:
                continue
            upd(('%s %s %s|' % (name, a.dtype, a.shape)).encode())
            upd(a.tobytes())
        for key in sorted(regs or {}):
            a = np.ascontiguousarray(np.asarray(regs[key]))
            upd(('reg %s %s %s|' % (key, a.dtype, a.shape)).encode())
            upd(a.tobytes())
        return '%s@%s' % (done(), torch.device(device))

    @classmethod
    def shared(cls, struct, regs=None, device='cuda:0'):
        """One BodyModel per (model contents, device) and process: the constants are immutable device tables, and building
        them costs ~45 ms of host time -- predict_mupots.py constructs a new optimiser (and with it a new SMPL) for every
        sequence of the test set."""
        key = cls._fingerprint(struct, regs, device)
        m = cls._SHARED.get(key)
        if m is None:
            m = cls._SHARED[key] = cls(struct, regs, device=device)
        return m

    def _self_check(self):
        """First use of the split 16-bit LBS forward on a device: 40 back-to-back launches must be bit-identical among
        themselves and within 2e-5 m of the exact-fp32 kernels.  A toolchain that packs the fp32 epilogue arithmetic beside
        the MFMAs (the hazard mhhip/build.py describes: wrong values on some lanes, more often under back-to-back launches)
        is caught here -- the process then stays on the exact kernels and says so -- instead of in somebody's results.
        ~2 ms once per device; MHHIP_LBS_SELFCHECK=0 skips it."""
        import os
        import warnings
        L = _lib.lib()
        key = str(self.device)
        if os.environ.get('MHHIP_LBS_SELFCHECK', '1') == '0' or key in BodyModel._CHECKED or L.mh_lbs_get_mode() == 0:
            return
        if torch.cuda.is_current_stream_capturing():      # 40 launches and a mode flip do not belong into somebody's graph:
            return                                        # the next model built outside a capture runs the check
        BodyModel._CHECKED.add(key)
        g = torch.Generator().manual_seed(7)
        B = 96
        betas = (0.7 * torch.randn(B, 10, generator=g)).to(self.device)
        poses = (0.3 * torch.randn(B, 72, generator=g)).to(self.device)
        ws = self.workspace(B)
        first, stable = None, True
        for _ in range(40):
            v = self.lbs_forward(betas, poses, ws=ws, want_vposed=False)[0]
            if first is None:
                first = v
            else:
                stable = stable and bool(torch.equal(v, first))
        prev_mode = int(L.mh_lbs_get_mode())             # (non-zero here; restored, not assumed)
        check(L.mh_lbs_set_mode(0))
        try:
            exact = self.lbs_forward(betas, poses, ws=ws, want_vposed=False)[0]
        finally:
            check(L.mh_lbs_set_mode(prev_mode))
        err = float((first - exact).abs().max())
        if not (stable and err <= 2e-5):
            check(L.mh_lbs_set_mode(0))                  # the arithmetic mode is process-wide in the C library: a failed check
            BodyModel._FAILED.add(key)                   # on ANY device downgrades the process (said below), recorded per device
            warnings.warn('split 16-bit LBS kernels failed their first-use check on %s (bit-stable over 40 launches: %s, max '
                          'deviation from the exact fp32 kernels %.2e m): staying on the exact fp32 kernels for this process '
                          '(3-4x slower LBS).  See mhhip/build.py for the compiler hazard this guards against.' % (key, stable, err))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.lib().mh_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- raw calls ------------------------------------------------------------------------------
    def workspace(self, B):
        n = _lib.lib().mh_lbs_workspace_bytes(B)
        return torch.empty(n, dtype=torch.uint8, device=self.device)

    def backward_workspace(self, B):
        n = _lib.lib().mh_lbs_backward_workspace_bytes(B)
        return torch.empty(n, dtype=torch.uint8, device=self.device)

    def lbs_forward(self, betas, poses, xscale=None, transl=None, ws=None, want_vposed=True, want_posed=False):
        B, NB = poses.shape[0], betas.shape[0]
        f = lambda t: None if t is None else t.contiguous().float()
        betas, poses, xscale, transl = f(betas), f(poses), f(xscale), f(transl)
        verts = torch.empty(B, self.V, 3, dtype=torch.float32, device=self.device)
        vposed = torch.empty_like(verts) if want_vposed else None
        posed = torch.empty(B, 24, 3, dtype=torch.float32, device=self.device) if want_posed else None
        ws = ws if ws is not None else self.workspace(B)
        check(_lib.lib().mh_lbs_forward(self.handle, B, NB, ptr(betas), ptr(poses), ptr(xscale), ptr(transl),
                                        ptr(verts), ptr(vposed), ptr(posed), ptr(ws), _lib.stream_ptr(self.device)))
        return verts, vposed, posed, ws

    def lbs_forward_rotmats(self, betas, rotmats, xscale=None, transl=None, want_posed=True, want_vposed=False):
        """``lbs(pose2rot=False)``: rotmats (B,24,3,3).  want_vposed: also (v_posed, workspace) for ``lbs_backward_ex``
        -> (verts, vposed, posed, ws); else (verts, posed)."""
        B, NB = rotmats.shape[0], betas.shape[0]
        f = lambda t: None if t is None else t.contiguous().float()
        betas, rotmats, xscale, transl = f(betas), f(rotmats), f(xscale), f(transl)
        verts = torch.empty(B, self.V, 3, dtype=torch.float32, device=self.device)
        vposed = torch.empty_like(verts) if want_vposed else None
        posed = torch.empty(B, 24, 3, dtype=torch.float32, device=self.device) if want_posed else None
        ws = self.workspace(B)
        check(_lib.lib().mh_lbs_forward_ex(self.handle, B, NB, ptr(betas), None, ptr(rotmats), ptr(xscale), ptr(transl),
                                           ptr(verts), ptr(vposed), ptr(posed), ptr(ws), _lib.stream_ptr(self.device)))
        return (verts, vposed, posed, ws) if want_vposed else (verts, posed)

    def joints_regress(self, which, verts, corr=None, root=-1):
        B = verts.shape[0]
        out = torch.empty(B, NUM_REG_JOINTS[which], 3, dtype=torch.float32, device=self.device)
        check(_lib.lib().mh_joints_regress(self.handle, which, B, ptr(verts), ptr(corr), root, ptr(out),
                                           _lib.stream_ptr(self.device)))
        return out

    def lbs_backward(self, betas, poses, xscale, transl, vposed, gverts, gjoints, ws, gposes=None, gtransl=None,
                     gbetas=None, gxscale=None, ws2=None):
        B, NB = poses.shape[0], betas.shape[0]
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        gposes = gposes if gposes is not None else z(B, 72)
        gtransl = gtransl if gtransl is not None else z(B, 3)
        gbetas = gbetas if gbetas is not None else z(NB, 10)
        gxscale = gxscale if gxscale is not None else z(NB)
        ws2 = ws2 if ws2 is not None else self.backward_workspace(B)
        check(_lib.lib().mh_lbs_backward(self.handle, B, NB, ptr(betas), ptr(poses), ptr(xscale), ptr(transl),
                                         ptr(vposed), ptr(gverts), ptr(gjoints), ptr(gposes), ptr(gtransl),
                                         ptr(gbetas), ptr(gxscale), ptr(ws), ptr(ws2), _lib.stream_ptr(self.device)))
        return gposes, gtransl, gbetas, gxscale


def _bm_lbs_backward_ex(self, betas, vposed, gverts, ws, poses=None, rotmats=None, gjoints=None, gposed=None, ws2=None):
    """mh_lbs_backward_ex: backward of the forward on axis-angle ``poses`` (B,72) or on ``rotmats`` (B,24,3,3), with the
    adjoints of the vertices, the 17 AlphaPose key-points and / or the 24 posed joints.  Returns (gposes | grotmats, gbetas)."""
    B, NB = gverts.shape[0], betas.shape[0]
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
    gposes = z(B, 72) if poses is not None else None
    grot = z(B, 24, 3, 3) if rotmats is not None else None
    gbetas = z(NB, 10)
    ws2 = ws2 if ws2 is not None else self.backward_workspace(B)
    check(_lib.lib().mh_lbs_backward_ex(self.handle, B, NB, ptr(betas), ptr(poses), ptr(rotmats), ptr(vposed), ptr(gverts),
                                        ptr(gjoints), ptr(gposed), ptr(gposes), ptr(grot), None, ptr(gbetas), None,
                                        ptr(ws), ptr(ws2), _lib.stream_ptr(self.device)))
    return (gposes if poses is not None else grot), gbetas


def _bm_joints_regress_backward(self, which, gjoints, root=-1, gverts=None, gcorr=None):
    """gverts (B,V,3) += reg^T gjoints (allocated as zeros when not given); gcorr (B,3) += (1 - row sums) gjoints"""
    B = gjoints.shape[0]
    if gverts is None:
        gverts = torch.zeros(B, self.V, 3, dtype=torch.float32, device=self.device)
    check(_lib.lib().mh_joints_regress_backward(self.handle, which, B, ptr(gjoints.contiguous()), int(root), ptr(gverts), ptr(gcorr),
                                                _lib.stream_ptr(self.device)))
    return gverts


BodyModel.lbs_backward_ex = _bm_lbs_backward_ex
BodyModel.joints_regress_backward = _bm_joints_regress_backward


def project_joints_loss(joints, K, Kd, pose2d, thr, mode, img_w, img_h, coef=1.0, want_uv=True):
    B = joints.shape[0]
    dev = joints.device
    _, Kp = _lib.host_f32(np.asarray(K, np.float32).reshape(3, 3))
    Kk = np.ascontiguousarray(np.asarray(K, np.float32).reshape(9))
    Kdd = None if Kd is None else np.ascontiguousarray(np.asarray(Kd, np.float32).reshape(5))
    uv = torch.empty(B, 17, 2, dtype=torch.float32, device=dev) if want_uv else None
    gj = torch.empty(B, 17, 3, dtype=torch.float32, device=dev)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    check(_lib.lib().mh_project_joints_loss(
        B, ptr(joints), Kk.ctypes.data_as(_lib.c_float_p),
        None if Kdd is None else Kdd.ctypes.data_as(_lib.c_float_p), ptr(pose2d), float(thr), int(mode),
        float(img_w), float(img_h), float(coef), ptr(uv), ptr(gj), ptr(loss), _lib.stream_ptr(dev)))
    return uv, gj, loss


def rmsprop_step(params, grads, sq, buf, lr, alpha=0.5, momentum=0.9, eps=1e-8):
    check(_lib.lib().mh_rmsprop_step(ptr(params), ptr(grads), ptr(sq), ptr(buf), params.numel(), lr, alpha, momentum,
                                     eps, _lib.stream_ptr(params.device)))


def rmsprop_step_log(params, grads, sq, buf, lr, log_src, log_dst, alpha=0.5, momentum=0.9, eps=1e-8, poke_dst=None, poke=None,
                     person=None):
    """mh_rmsprop_step_log[_poke]: the update and, in the same launch, the cycle's log entries from their staging row into
    log_dst (log_src None: no copy) and up to two int32 words ``poke`` into ``poke_dst``.  person (a ``_lib.PersonSums``):
    mh_rmsprop_step_person -- the launch also sums the per-body shape / scale gradients of a backward that left them per body"""
    nlog = 0 if log_src is None else log_src.numel()
    if person is not None:
        pk = [int(v) for v in (poke or [])] + [0, 0]
        check(_lib.lib().mh_rmsprop_step_person(ptr(params), ptr(grads), ptr(sq), ptr(buf), params.numel(), lr, alpha, momentum,
                                                eps, ptr(log_src), ptr(log_dst), nlog, poke_dst.data_ptr() if poke is not None else None,
                                                0 if poke is None else min(len(poke), 2), pk[0], pk[1], ctypes.byref(person),
                                                _lib.stream_ptr(params.device)))
        return
    if poke is None:
        check(_lib.lib().mh_rmsprop_step_log(ptr(params), ptr(grads), ptr(sq), ptr(buf), params.numel(), lr, alpha, momentum,
                                             eps, ptr(log_src), ptr(log_dst), nlog, _lib.stream_ptr(params.device)))
        return
    poke = [int(v) for v in poke] + [0]
    check(_lib.lib().mh_rmsprop_step_log_poke(ptr(params), ptr(grads), ptr(sq), ptr(buf), params.numel(), lr, alpha, momentum,
                                              eps, ptr(log_src), ptr(log_dst), nlog, poke_dst.data_ptr(), min(len(poke) - 1, 2),
                                              poke[0], poke[1], _lib.stream_ptr(params.device)))


def adam_step(params, grads, m, v, step, lr, b1=0.5, b2=0.5, eps=1e-6):
    check(_lib.lib().mh_adam_step(ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), int(step), lr, b1, b2, eps,
                                  _lib.stream_ptr(params.device)))


def one_euro_scan(x, min_cutoff, beta, frame_rate=25.0, out=None):
    """out: a contiguous float32 tensor of x's size to write the filtered sequence into (returned, viewed like x)"""
    x = x.contiguous()
    if out is None:
        y = torch.empty_like(x)
    else:
        assert out.is_contiguous() and out.dtype == x.dtype and out.numel() == x.numel() and out.device == x.device
        y = out.view(x.shape)
    T = x.shape[0]
    check(_lib.lib().mh_one_euro_scan(ptr(x), ptr(y), T, x.numel() // T, float(min_cutoff), float(beta),
                                      float(frame_rate), _lib.stream_ptr(x.device)))
    return y


def one_euro_time_before(first_frame, frame_rate=25.0):
    """float32 running time stamp after frame ``first_frame - 1`` (reference optimizer.py:671)."""
    t = np.float32(0)
    for i in range(1, int(first_frame)):
        t = np.float32(t + np.float32(i / frame_rate))
    return float(t)


def one_euro_scan_shard(x, min_cutoff, beta, first_frame, state_in=None, frame_rate=25.0):
    """x (T_local, ...) -> y, state_out=(y[-1], dxhat[-1]) for the next rank."""
    x = x.contiguous()
    y = torch.empty_like(x)
    T = x.shape[0]
    E = x.numel() // T
    dx_out = torch.empty(E, dtype=torch.float32, device=x.device)
    xp, dxp = (None, None) if state_in is None else state_in
    check(_lib.lib().mh_one_euro_scan_shard(ptr(x), ptr(y), T, E, float(min_cutoff), float(beta), float(frame_rate),
                                            int(first_frame), one_euro_time_before(first_frame, frame_rate),
                                            ptr(xp), ptr(dxp), ptr(dx_out), _lib.stream_ptr(x.device)))
    return y, (y[-1].reshape(-1).contiguous(), dx_out)


def velocity_term(pT, coef, gpT, prev_halo=None, next_halo=None):
    T, N = pT.shape[0], pT.shape[1]
    loss = torch.empty(1, dtype=torch.float32, device=pT.device)
    check(_lib.lib().mh_velocity_term(T, N, ptr(pT), ptr(prev_halo), ptr(next_halo), float(coef), ptr(gpT), ptr(loss),
                                      _lib.stream_ptr(pT.device)))
    return loss


def filtered_verts_term(verts, verts_filt, coef, gverts, prev=None, nxt=None):
    T = verts.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=verts.device)
    pv, pvf = prev if prev is not None else (None, None)
    nv, nvf = nxt if nxt is not None else (None, None)
    E = verts.numel() // T
    ws = torch.empty(_lib.lib().mh_filtered_verts_workspace_bytes(T, E), dtype=torch.uint8, device=verts.device)
    check(_lib.lib().mh_filtered_verts_term(T, E, ptr(verts), ptr(verts_filt), ptr(pv), ptr(pvf),
                                            ptr(nv), ptr(nvf), float(coef), ptr(gverts), ptr(loss), ptr(ws),
                                            _lib.stream_ptr(verts.device)))
    return loss
