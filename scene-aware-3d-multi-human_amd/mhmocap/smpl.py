"""Drop-in for the reference's ``mhmocap.smpl`` on MI355X.

Same call surface as reference ``mhmocap/smpl.py`` (``SMPL(model_path, J_reg_*_path=...,
data_struct=...)``, ``.to(device)``, ``.faces``, ``__call__(batch_size=512, betas=, poses=)`` with
the same output dict, ``lbs(...)``, ``Struct``), but every floating point operation runs in the
hand-written HIP kernels behind ``include/mhmocap_hip.h`` (``mh_lbs_forward`` /
``mh_lbs_backward`` / ``mh_joints_regress``).  There is no CPU path: constructing the model
without a HIP device raises.

Differentiability: every entry of the output dict carries its graph to ``betas`` / ``poses`` like the
reference's (smpl.py:362-397) -- ``verts``, ``joints_alphapose`` and ``joints_smpl24`` through the
hand-written LBS backward (``mh_lbs_backward_ex``), the other joint sets through the regressors'
adjoint (``mh_joints_regress_backward``); ``lbs(pose2rot=False)`` is differentiable in the rotation
matrices.
"""
import os
import os.path as osp

import numpy as np
import torch
import torch.nn as nn

from mhhip import engine

# vertex-picked joints appended to the 24 regressed ones (reference smpl.py:67-106, 402-425)
VERTEX_IDS = {
    'smplh': {'nose': 332, 'reye': 6260, 'leye': 2800, 'rear': 4071, 'lear': 583, 'rthumb': 6191, 'rindex': 5782,
              'rmiddle': 5905, 'rring': 6016, 'rpinky': 6133, 'lthumb': 2746, 'lindex': 2319, 'lmiddle': 2445,
              'lring': 2556, 'lpinky': 2673, 'LBigToe': 3216, 'LSmallToe': 3226, 'LHeel': 3387, 'RBigToe': 6617,
              'RSmallToe': 6624, 'RHeel': 6787},
}
_EXTRA_ORDER = ['nose', 'reye', 'leye', 'rear', 'lear', 'LBigToe', 'LSmallToe', 'LHeel', 'RBigToe', 'RSmallToe',
                'RHeel', 'lthumb', 'lindex', 'lmiddle', 'lring', 'lpinky', 'rthumb', 'rindex', 'rmiddle', 'rring',
                'rpinky']


class Struct(object):
    def __init__(self, **kwargs):
        for key, val in kwargs.items():
            setattr(self, key, val)


class _LbsFn(torch.autograd.Function):
    """verts, joints_alphapose, joints_smpl24 = f(betas, pose) with the hand-written HIP backward; ``pose`` is (B,72)
    axis-angle or, with ``rot``, (B,24,3,3) rotation matrices (``lbs(pose2rot=False)``, smpl.py:541-558).  All three
    outputs are differentiable, like the reference's autograd graph (smpl.py:362-397)."""

    @staticmethod
    def forward(ctx, betas, pose, model, want_kp, rot):
        if rot:
            verts, vposed, posed, ws = model.lbs_forward_rotmats(betas, pose, want_posed=True, want_vposed=True)
        else:
            verts, vposed, posed, ws = model.lbs_forward(betas, pose, want_posed=True)
        kp = model.joints_regress(engine.REG_ALPHAPOSE, verts) if want_kp else verts.new_zeros(pose.shape[0], 17, 3)
        ctx.model, ctx.ws, ctx.want_kp, ctx.rot = model, ws, want_kp, rot
        ctx.save_for_backward(betas, pose, vposed)
        return verts, kp, posed

    @staticmethod
    def backward(ctx, gverts, gkp, gposed):
        betas, pose, vposed = ctx.saved_tensors
        f = lambda g: None if g is None else g.contiguous().float()
        gv, gk, gp = f(gverts), f(gkp) if ctx.want_kp else None, f(gposed)
        if gv is None and gk is None and gp is None:
            return None, None, None, None, None
        if gv is None:
            gv = torch.zeros(pose.shape[0], ctx.model.V, 3, device=pose.device)
        kw = dict(rotmats=pose) if ctx.rot else dict(poses=pose)
        gpose, gbetas = ctx.model.lbs_backward_ex(betas, vposed, gv, ctx.ws, gjoints=gk, gposed=gp, **kw)
        return gbetas, gpose.view_as(pose), None, None, None


class _RegressFn(torch.autograd.Function):
    """joints = regressor . verts (optionally relative to a root joint) with its adjoint in HIP: the other joint sets
    of SMPL.forward (joints_h36m17 / joints_mupots / the extra nine of j3d) keep their graph (smpl.py:367-386)"""

    @staticmethod
    def forward(ctx, verts, model, which, root):
        ctx.model, ctx.which, ctx.root = model, which, root
        return model.joints_regress(which, verts.contiguous(), root=root)

    @staticmethod
    def backward(ctx, gj):
        return ctx.model.joints_regress_backward(ctx.which, gj.contiguous().float(), root=ctx.root), None, None, None


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23
    NUM_BETAS = 10

    def __init__(self, model_path, J_reg_extra9_path=None, J_reg_h36m17_path=None, J_reg_alphapose_path=None,
                 J_reg_mupots_path=None, data_struct=None, betas=None, global_orient=None, body_pose=None,
                 transl=None, dtype=torch.float32, batch_size=1, joint_mapper=None, gender='neutral',
                 vertex_ids=None, device=None, **kwargs):
        super(SMPL, self).__init__()
        self.gender = gender
        if data_struct is None:                                    # reference smpl.py:179-188
            if osp.isdir(model_path):
                smpl_path = os.path.join(model_path, 'SMPL_{}.{ext}'.format(gender.upper(), ext='pkl'))
            else:
                smpl_path = model_path
            assert osp.exists(smpl_path), 'Path {} does not exist!'.format(smpl_path)
            data_struct = Struct(**engine.load_smpl_pickle(smpl_path))
        assert dtype == torch.float32, 'the MI355X path computes in fp32 like the reference (smpl.py:134)'
        self.dtype = dtype
        self.batch_size = batch_size
        self._struct = data_struct
        self._regs = {}
        for key, path in [('extra9', J_reg_extra9_path), ('h36m', J_reg_h36m17_path),
                          ('alphapose', J_reg_alphapose_path), ('mupots', J_reg_mupots_path)]:
            if path is not None:
                self._regs[key] = np.load(path)
        self.faces = data_struct.f                                  # read by optimizer.py:73-74
        ids = VERTEX_IDS['smplh'] if vertex_ids is None else vertex_ids
        self._extra_idx = [ids[k] for k in _EXTRA_ORDER]
        self._model = None
        self._device = torch.device(device) if device is not None else None
        default_betas = torch.zeros([batch_size, self.NUM_BETAS]) if betas is None else torch.as_tensor(betas).float()
        self.register_parameter('betas', nn.Parameter(default_betas, requires_grad=True))

    # nn.Module.to() moves the parameter; the constants follow on first use
    def to(self, device=None, *args, **kwargs):
        if device is not None:
            dev = torch.device(device)
            if self._device != dev:
                self._device, self._model = dev, None
        return super(SMPL, self).to(device, *args, **kwargs)

    @property
    def body_model(self):
        if self._model is None:
            dev = self._device or torch.device('cuda:0')
            if dev.type != 'cuda':
                raise RuntimeError('mhmocap.smpl.SMPL (MI355X build) needs a HIP device; got %s' % dev)
            self._model = engine.BodyModel.shared(self._struct, self._regs, device=dev)
            self._device = dev
        return self._model

    def get_num_verts(self):
        return int(np.asarray(self._struct.v_template).shape[0])

    def get_num_faces(self):
        return int(np.asarray(self.faces).shape[0])

    def extra_repr(self):
        return 'Number of betas: {}'.format(self.NUM_BETAS)

    def forward(self, batch_size=512, **kwargs):
        # the reference chunks by 512 bodies and concatenates (smpl.py:297-310); one launch covers
        # any number of bodies here, the argument is accepted for call compatibility
        return self.single_forward(**kwargs)

    def single_forward(self, betas=None, poses=None, transl=None, return_verts=True, return_full_pose=False, **kwargs):
        m = self.body_model
        betas = betas if betas is not None else self.betas
        if isinstance(betas, np.ndarray):
            betas = torch.from_numpy(betas).float()
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses).float()
        if isinstance(transl, np.ndarray):
            transl = torch.from_numpy(transl).float()
        betas = betas.to(m.device).float().contiguous()
        poses = poses.to(m.device).float().contiguous()
        if betas.shape[0] != poses.shape[0]:
            betas = betas.expand(poses.shape[0], -1).contiguous()
        want_kp = m.has_reg[engine.REG_ALPHAPOSE]
        verts, kp, j24 = _LbsFn.apply(betas, poses, m, want_kp, False)
        j3d = torch.cat([j24, verts[:, self._extra_idx]], dim=1)    # smpl.py:362-365 (the 21 picked vertices: an index, not arithmetic)
        out = {'verts': verts, 'j3d': j3d, 'joints_smpl24': j24}
        if m.has_reg[engine.REG_H36M17]:
            out['joints_h36m17'] = _RegressFn.apply(verts, m, engine.REG_H36M17, 14)   # smpl.py:367-373
        if want_kp:
            out['joints_alphapose'] = kp
        if m.has_reg[engine.REG_MUPOTS]:
            out['joints_mupots'] = _RegressFn.apply(verts, m, engine.REG_MUPOTS, -1)
        if m.has_reg[engine.REG_EXTRA9]:
            out['j3d'] = torch.cat([j3d, _RegressFn.apply(verts, m, engine.REG_EXTRA9, -1)], dim=1)   # smpl.py:383-386
        if transl is not None:
            t = transl.to(m.device).unsqueeze(1)
            out = {k: v + t for k, v in out.items()}                # smpl.py:396-397
        return out


def create(model_path, model_type='smpl', **kwargs):
    if model_type.lower() == 'smpl':
        return SMPL(model_path, **kwargs)
    raise ValueError('Unknown model type {}, exiting!'.format(model_type))


_LBS_CACHE = {}


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot=True,
        dtype=torch.float32):
    """Call-compatible with reference smpl.py:490-576: builds (and caches) a device model from the given tensors
    and runs the HIP forward (+ backward under autograd).  ``pose2rot=False``: pose = (B,24,3,3) rotation matrices, :553-558."""
    key = (v_template.data_ptr(), posedirs.data_ptr(), lbs_weights.data_ptr())
    if key not in _LBS_CACHE:
        V = v_template.shape[0]
        par = parents.detach().cpu().numpy().astype(np.int64).copy()
        kin = np.zeros((2, 24), np.int64)
        kin[0] = par
        st = Struct(v_template=v_template.detach().cpu().numpy(), shapedirs=shapedirs.detach().cpu().numpy(),
                    posedirs=posedirs.detach().cpu().numpy().T.reshape(V, 3, -1),
                    J_regressor=J_regressor.detach().cpu().numpy(), kintree_table=kin,
                    weights=lbs_weights.detach().cpu().numpy(), f=np.zeros((1, 3), np.int64))
        dev = v_template.device if v_template.device.type == 'cuda' else torch.device('cuda:0')
        _LBS_CACHE[key] = engine.BodyModel(st, {}, device=dev)
    m = _LBS_CACHE[key]
    b = betas.to(m.device).float()
    p = pose.to(m.device).float()
    if b.shape[0] != p.shape[0]:
        b = b.expand(p.shape[0], -1)
    if not pose2rot:
        verts, _, posed = _LbsFn.apply(b.contiguous(), p.reshape(p.shape[0], 24, 3, 3).contiguous(), m, False, True)
        return verts, posed
    verts, _, posed = _LbsFn.apply(b.contiguous(), p.contiguous(), m, False, False)
    return verts, posed


# helper names of the shadowed reference module (VertexJointSelector, batch_rodrigues, vertices2joints ...; pure
# torch utilities outside the accelerated path) stay importable from here (mhhip/_overlay.py)
from mhhip._overlay import inherit as _inherit  # noqa: E402

_inherit(globals())
