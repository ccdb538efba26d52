"""The 2D key-point term evaluated from the pose features and joint transforms (mh_keypoint_terms, round 4) against the pass
over the vertices it replaces: J_regressor_alphapose . verts (smpl.py:374-376) through mh_joints_regress, the projection /
residual through mh_project_joints_loss_w, and the adjoint through the vertex scatter inside the LBS backward.  Both are
fp32 evaluations of the same linear map in different orders; the oracle fixtures pin either to the reference."""
import numpy as np
import pytest
import torch

from test_fit_full_gpu import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('T,N,W,H,batch,dist', [(12, 3, 96, 54, 3, False), (9, 5, 64, 80, 4, True), (40, 4, 240, 135, 10, False)])
def test_keypoint_terms_equal_the_regression_from_the_vertices(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, dist):
    from mhhip import _lib
    from mhhip._lib import check, ptr
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 61, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    if dist:
        e.Kd = np.array([0.05, -0.02, 0.001, -0.002, 0.003], np.float32)
    e.joint_w = (np.linspace(0.5, 1.5, 17) / np.linspace(0.5, 1.5, 17).mean()).astype(np.float32)
    L = _lib.lib()
    st = _lib.stream_ptr(e.dev)
    B = e.B
    e.leaf('xscale').copy_(torch.linspace(-1.5, 2.0, N, device=e.dev))
    e.forward(regress=True)                      # verts + key-points regressed from them (self.kp)
    Kp = e.K.ctypes.data_as(_lib.c_float_p)
    Kdp = None if e.Kd is None else e.Kd.ctypes.data_as(_lib.c_float_p)
    jwp = e.joint_w.ctypes.data_as(_lib.c_float_p)
    uv0, gj0, l0 = torch.zeros(B, 17, 2, device=e.dev), torch.zeros(B, 17, 3, device=e.dev), torch.zeros(B, device=e.dev)
    check(L.mh_project_joints_loss_w(B, ptr(e.kp), Kp, Kdp, jwp, ptr(e.pose2d), e.thr, 0, float(W), float(H), 0.7, ptr(uv0), ptr(gj0), ptr(l0), st))
    kp0 = e.kp.clone()
    kp1, uv1, gj1, l1 = torch.zeros_like(kp0), torch.zeros_like(uv0), torch.zeros_like(gj0), torch.zeros_like(l0)
    check(L.mh_keypoint_terms(e.m.handle, B, ptr(e.leaf('poses_T')), Kp, Kdp, jwp, ptr(e.pose2d), e.thr, float(W), float(H), 0.7,
                              ptr(kp1), ptr(uv1), ptr(gj1), ptr(l1), ptr(e.ws), ptr(e.ws2), ptr(e.kp_ws), st))
    torch.cuda.synchronize()
    assert float((kp1 - kp0).abs().max()) < 3e-6                                   # metres (bodies 3-8 m from the camera)
    assert float((uv1 - uv0).abs().max()) < 2e-3                                   # pixels
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=2e-4, atol=1e-9)
    assert float((gj1 - gj0).abs().max()) <= 2e-4 * float(gj0.abs().max())
    # the adjoint: leaf gradients of the 2D term alone through both backward forms, from the SAME dL/dkp
    g = e.grads
    outs = []
    for form in ('vertices', 'features'):
        g.zero_()
        args = (ptr(e.leaf('poses_smpl', g)), ptr(e.leaf('poses_T', g)), ptr(e.leaf('betas', g)), ptr(e.leaf('xscale', g)), ptr(e.ws), ptr(e.ws2), st)
        if form == 'vertices':
            check(L.mh_lbs_backward(e.m.handle, B, N, ptr(e.leaf('betas')), ptr(e.leaf('poses_smpl')), ptr(e.leaf('xscale')),
                                    ptr(e.leaf('poses_T')), ptr(e.vposed), None, ptr(gj1), *args))
        else:
            check(L.mh_lbs_backward_kp(e.m.handle, B, N, ptr(e.leaf('betas')), ptr(e.leaf('poses_smpl')), ptr(e.vposed), None, *args))
        torch.cuda.synchronize()
        outs.append({k: e.leaf(k, g).clone() for k in ('poses_smpl', 'poses_T', 'betas', 'xscale')})
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        scale = float(a.abs().max())
        assert scale > 0, k
        err = float((a - b).abs().max())
        print('%-10s %.2e of the largest entry' % (k, err / scale))
        assert err <= 1e-5 * scale, (k, err, scale)          # measured 1.2e-6
