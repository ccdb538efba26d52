"""ORACLE (test infrastructure only) -- CPU restatement of the SMPL / LBS forward.

This file restates, in the build's own words, the arithmetic of the reference's
``mhmocap/smpl.py`` so that the HIP path can be checked on any machine (the
reference itself never travels to the GPU box).  It is imported ONLY by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg;
the product path (``scene-aware-3d-multi-human_amd/``) never imports it.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
fixtures captured from the reference's own CPU code in the build container
(``tests/golden/make_golden.py``).

Everything is written on torch CPU tensors (the reference's own arithmetic
engine is ATen-CPU) so that gradients of the restatement come from autograd and
can pin the hand-written HIP backward.  dtype follows the inputs (fp32 for
parity with the reference, fp64 for tight derivative checks).
"""
import numpy as np
import torch

H36M_ROW_ORDER = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]  # smpl.py:242
# vertex-picked extra joints (nose, eyes, ears, feet, finger tips): smpl.py:402-425, order of :67-106
EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583,
                    3216, 3226, 3387, 6617, 6624, 6787,
                    2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]


class BodyModel(object):
    """Constants of an SMPL-shaped model as torch tensors (smpl.py:201-275)."""

    def __init__(self, struct, regs=None, dtype=torch.float32):
        f = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), dtype=dtype)
        self.dtype = dtype
        self.v_template = f(struct.v_template)                       # (V,3)      smpl.py:211
        self.shapedirs = f(np.asarray(struct.shapedirs)[:, :, :10])  # (V,3,10)   smpl.py:227
        self.J_regressor = f(struct.J_regressor)                     # (24,V)     smpl.py:231
        pd = np.asarray(struct.posedirs, dtype=np.float32)
        self.posedirs = f(pd.reshape(-1, pd.shape[-1]).T)            # (207,V*3)  smpl.py:264-267
        par = np.asarray(struct.kintree_table[0]).astype(np.int64)
        par[0] = -1                                                  # smpl.py:270-271
        self.parents = par
        self.weights = f(struct.weights)                             # (V,24)     smpl.py:274
        self.faces = np.asarray(struct.f).astype(np.int64)           # smpl.py:205
        regs = regs or {}
        self.reg_extra9 = f(regs['extra9']) if 'extra9' in regs else None               # smpl.py:235
        self.reg_h36m17 = f(np.asarray(regs['h36m'])[H36M_ROW_ORDER]) if 'h36m' in regs else None  # :243
        self.reg_alphapose = f(np.asarray(regs['alphapose']).T) if 'alphapose' in regs else None   # :250
        self.reg_mupots = f(np.asarray(regs['mupots']).T) if 'mupots' in regs else None            # :257


def rodrigues(rvec):
    """Axis-angle (M,3) -> rotation (M,3,3); smpl.py:647-678.

    Quirk kept: the angle is the norm of ``rvec + 1e-8`` (eps added to every
    component before the norm) while the axis divides the un-shifted vector.
    """
    ang = torch.sqrt(((rvec + 1e-8) ** 2).sum(dim=1, keepdim=True))     # :662
    ax = rvec / ang                                                     # :663
    x, y, z = ax[:, 0], ax[:, 1], ax[:, 2]
    o = torch.zeros_like(x)
    K = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=1).view(-1, 3, 3)   # :673
    s = torch.sin(ang).unsqueeze(-1)
    c = torch.cos(ang).unsqueeze(-1)
    eye = torch.eye(3, dtype=rvec.dtype).unsqueeze(0)
    return eye + s * K + (1 - c) * torch.bmm(K, K)                      # :677


def rigid_chain(R, J, parents):
    """World transforms along the kinematic tree; smpl.py:692-746.

    R (B,24,3,3), J (B,24,3) -> posed joints (B,24,3), A (B,24,4,4) with the
    rest pose removed (``A = G - pad(G @ [J;0])``, :743-744).
    """
    B, NJ = J.shape[:2]
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]                           # :718-719
    L = torch.zeros(B, NJ, 4, 4, dtype=R.dtype)
    L[:, :, :3, :3] = R
    L[:, :, :3, 3] = rel
    L[:, :, 3, 3] = 1                                                   # :680-690
    G = [L[:, 0]]
    for j in range(1, NJ):
        G.append(torch.matmul(G[int(parents[j])], L[:, j]))             # :726-731
    G = torch.stack(G, dim=1)
    posed = G[:, :, :3, 3]                                              # :736
    Jh = torch.cat([J, torch.zeros(B, NJ, 1, dtype=R.dtype)], dim=2).unsqueeze(-1)
    corr = torch.matmul(G, Jh)                                          # (B,24,4,1)
    A = G.clone()
    A[:, :, :, 3:4] = A[:, :, :, 3:4] - corr                            # :743-744
    return posed, A


def lbs(model, betas, pose):
    """betas (B,10), pose (B,72) -> verts (B,V,3), posed joints (B,24,3); smpl.py:490-576."""
    B = pose.shape[0]
    dt = pose.dtype
    v_shaped = model.v_template[None] + torch.einsum('bl,vcl->bvc', betas, model.shapedirs)   # :532
    J = torch.einsum('jv,bvc->bjc', model.J_regressor, v_shaped)                              # :535
    R22 = rodrigues(pose[:, :66].reshape(-1, 3)).view(B, 22, 3, 3)                            # :544 (pose[:, :-6])
    eye = torch.eye(3, dtype=dt)
    R = torch.cat([R22, eye.view(1, 1, 3, 3).expand(B, 2, 3, 3)], dim=1)                      # :542,546 hands = identity
    feat = (R[:, 1:] - eye).reshape(B, 207)                                                   # :547
    v_posed = v_shaped + torch.matmul(feat, model.posedirs).view(B, -1, 3)                    # :549,558
    posed_joints, A = rigid_chain(R, J, model.parents)                                        # :560
    T = torch.matmul(model.weights[None].expand(B, -1, -1), A.view(B, 24, 16)).view(B, -1, 4, 4)   # :567
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], dim=2)
    verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]                                    # :573-574
    return verts, posed_joints


def smpl_forward(model, betas, poses):
    """The dict of ``SMPL.single_forward`` (smpl.py:357-386)."""
    verts, j24 = lbs(model, betas, poses)
    out = {'verts': verts, 'joints_smpl24': j24}
    j3d = torch.cat([j24, verts[:, EXTRA_VERTEX_IDS]], dim=1)                                  # :363, :111-113
    if model.reg_h36m17 is not None:
        h = torch.einsum('jv,bvc->bjc', model.reg_h36m17, verts)
        out['joints_h36m17'] = h - h[:, 14:15]                                                 # :369-372
    if model.reg_alphapose is not None:
        out['joints_alphapose'] = torch.einsum('jv,bvc->bjc', model.reg_alphapose, verts)      # :376
    if model.reg_mupots is not None:
        out['joints_mupots'] = torch.einsum('jv,bvc->bjc', model.reg_mupots, verts)            # :380
    if model.reg_extra9 is not None:
        j3d = torch.cat([j3d, torch.einsum('jv,bvc->bjc', model.reg_extra9, verts)], dim=1)    # :385
    out['j3d'] = j3d
    return out
