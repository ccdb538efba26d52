"""Micro-timings of the LBS forward / backward on the C3 workload (developer tool, not part of the product)."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq, _lib
from mhhip._lib import ptr, check
from time_kernels import timeit


def main():
    T = int(os.environ.get('T', '200'))
    struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(bench.IMG, 60.0)
    opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=5)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=50, shuffle=False)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    L = _lib.lib(); st = _lib.stream_ptr(e.dev)
    fwd = lambda: check(L.mh_lbs_forward(e.m.handle, e.B, e.N, ptr(e.leaf('betas')), ptr(e.leaf('poses_smpl')), ptr(e.leaf('xscale')),
                                         ptr(e.leaf('poses_T')), ptr(e.verts), ptr(e.vposed), None, ptr(e.ws), st))
    for mode in (0, 1):
        L.mh_lbs_set_mode(mode)
        e.verts.zero_(); e.vposed.zero_()
        t = timeit(fwd, 50)
        torch.cuda.synchronize()
        if mode == 0:
            v0, q0 = e.verts.clone(), e.vposed.clone()
        print('lbs fwd mode %d ms %.4f' % (mode, t))
    print('split16 vs fp32: max |dverts| %.3e  max |dvposed| %.3e  (max |verts| %.2f)' % (
        float((e.verts - v0).abs().max()), float((e.vposed - q0).abs().max()), float(v0.abs().max())))
    gv = torch.randn_like(e.verts) * 1e-3
    g = e.grads
    bwd = lambda: check(L.mh_lbs_backward(e.m.handle, e.B, e.N, ptr(e.leaf('betas')), ptr(e.leaf('poses_smpl')), ptr(e.leaf('xscale')),
                                          ptr(e.leaf('poses_T')), ptr(e.vposed), ptr(gv), ptr(e.gj), ptr(e.leaf('poses_smpl', g)),
                                          ptr(e.leaf('poses_T', g)), ptr(e.leaf('betas', g)), ptr(e.leaf('xscale', g)), ptr(e.ws),
                                          ptr(e.ws2), st))
    res = {}
    for mode in (0, 1):
        L.mh_lbs_set_mode(mode)
        fwd()
        for n in ('poses_smpl', 'poses_T', 'betas', 'xscale'):
            e.leaf(n, g).zero_()
        bwd()
        torch.cuda.synchronize()
        res[mode] = {n: e.leaf(n, g).clone() for n in ('poses_smpl', 'poses_T', 'betas', 'xscale')}
        print('lbs bwd mode %d ms %.4f' % (mode, timeit(bwd, 50)))
    for n in res[0]:
        a, b = res[0][n], res[1][n]
        print('  grad %-10s max|fp32| %.3e  max|split16 - fp32| %.3e  rel %.2e' % (n, float(a.abs().max()), float((a - b).abs().max()),
                                                                                 float((a - b).abs().max() / a.abs().max())))


if __name__ == '__main__':
    main()
