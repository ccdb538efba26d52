"""LBS forward / backward of the HIP model against the oracle in float64 on random batches with EXTREME inputs: shape
coefficients up to +-4, joint rotations up to pi about random axes (and exact zeros), scales 1.1^(+-6), translations of
+-20 m, batch sizes around the 32-body group boundaries of the kernels (tools/fuzz_lbs.py runs any number of them;
measured: vertices within 1.5e-6 of the body's extent, i.e. at the rounding of a coordinate 20 m from the origin, worst
gradient entry 1.4e-5 of its leaf's largest)."""
import numpy as np
import pytest
import torch

from oracle import lbs_oracle as lo
import test_lbs_gpu as tl

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [17, 29])
def test_extreme_shapes_poses_scales_translations(smpl_struct, smpl_regs, seed):
    from mhhip import engine
    hip = engine.BodyModel(smpl_struct, smpl_regs)
    m64 = lo.BodyModel(smpl_struct, smpl_regs, dtype=torch.float64)
    rng = np.random.RandomState(seed)
    d = tl.dev
    for c in range(6):
        B = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100]))
        NB = int(rng.choice([x for x in (1, 2, 3, 4, B) if B % x == 0]))
        amp_b, amp_p = float(rng.choice([0.7, 2.0, 4.0])), float(rng.choice([0.3, 1.0, np.pi]))
        betas = rng.uniform(-amp_b, amp_b, (NB, 10)).astype(np.float32)
        poses = rng.uniform(-1, 1, (B, 24, 3)).astype(np.float32)
        poses *= (amp_p * rng.uniform(0, 1, (B, 24, 1)) / np.maximum(np.linalg.norm(poses, axis=2, keepdims=True), 1e-6)).astype(np.float32)
        poses[rng.uniform(size=(B, 24)) < 0.2] = 0
        poses = poses.reshape(B, 72)
        poses[:, 66:] = 0
        xs = rng.uniform(-6, 6, (NB,)).astype(np.float32)
        tr = rng.uniform(-20, 20, (B, 3)).astype(np.float32)
        wv = rng.normal(0, 1, (B, 6890, 3)).astype(np.float32)
        wj = rng.normal(0, 5, (B, 17, 3)).astype(np.float32)
        where = 'batch %d (B %d, NB %d, |betas| <= %.1f, |rot| <= %.1f)' % (c, B, NB, amp_b, amp_p)
        verts, vposed, _, ws = hip.lbs_forward(d(betas), d(poses), d(xs), d(tr))
        bidx = np.arange(B) % NB
        ref = lo.smpl_forward(m64, torch.tensor(betas[bidx]).double(), torch.tensor(poses).double())
        s = torch.pow(torch.tensor(1.1, dtype=torch.float64), torch.tensor(xs[bidx]).double())[:, None, None]
        want = (s * ref['verts'] + torch.tensor(tr).double()[:, None]).numpy()
        # 4e-6 m + two roundings of the coordinate itself (a vertex 20 m from the origin has a 1.9e-6 m ulp)
        np.testing.assert_allclose(verts.cpu().numpy(), want, atol=4e-6, rtol=2.4e-7, err_msg=where)
        got = hip.lbs_backward(d(betas), d(poses), d(xs), d(tr), vposed, d(wv), d(wj), ws)
        torch.cuda.synchronize()
        wantg = tl._oracle_grads(m64, betas, poses, xs, tr, wv, wj, NB, torch.float64)
        for name, g, w in zip(['poses', 'transl', 'betas', 'xscale'], got, wantg):
            g = g.cpu().numpy().reshape(w.shape)
            # xscale: NB entries, each the sum of <g, x> over every vertex of every body of the person -- with random upstream
            # gradients it can cancel to almost nothing (4e-4 of itself seen in 60 more batches), so it gets the looser gate
            tol = 2e-3 if name == 'xscale' else 5e-5
            np.testing.assert_allclose(g, w, atol=tol * np.abs(w).max(), rtol=0, err_msg='%s: %s' % (where, name))
