"""Micro-timings of the hot kernels on the C3 workload (developer tool, not part of the product)."""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq
from mhhip.raster import RasterTerms


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    T = int(os.environ.get('T', '200'))
    struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(bench.IMG, 60.0)
    tmp = tempfile.mkdtemp()
    opt = bench.build_optimizer(struct, regs, tmp, T, 'cuda:0', K)
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=20)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    opt.scene_depth = bench.ground_scene(K, *bench.IMG)
    opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)
    r = RasterTerms(e)
    e.cycle(0, raster=r)
    gv = torch.zeros_like(e.verts); log = torch.zeros(16, device='cuda:0')
    v = e.verts
    u = K[0, 0] * v[..., 0] / v[..., 2] + K[0, 2]; w = K[1, 1] * v[..., 1] / v[..., 2] + K[1, 2]
    ww = (u.amax(1).clamp(0, 239) - u.amin(1).clamp(0, 239) + 5); wh = (w.amax(1).clamp(0, 134) - w.amin(1).clamp(0, 134) + 5)
    area = (ww * wh)
    print('window px: mean %.0f max %.0f; >1920: %d of %d; z range %.2f..%.2f' % (area.mean(), area.max(), (area > 1920).sum(), area.numel(), v[..., 2].min(), v[..., 2].max()))
    print('seg coverage px/body', seq['seg_mask'].sum() / (T * 4))
    print('raster fwd+bwd  ms', timeit(lambda: r(e, gv, log)))
    print('raster fwd only ms', timeit(lambda: r(e, gv, log, with_grads=False)))
    import ctypes as _ct
    from mhhip import _lib as _l3
    if hasattr(_l3.lib(), 'mh_debug_ts'):
        r(e, gv, log); torch.cuda.synchronize()
        buf = (_ct.c_ulonglong * 16)(); _l3.lib().mh_debug_ts(buf); ts = list(buf)
        print('grads block 0 phases (100 MHz ticks -> us):', [round((ts[i + 1] - ts[i]) / 100.0, 2) for i in range(5)])
        buf2 = (_ct.c_ulonglong * 8192)(); _l3.lib().mh_debug_tb(buf2); tb = np.array(list(buf2), dtype=np.float64).reshape(4096, 2); tb = tb[tb[:, 1] > 0]
        t0 = tb[:, 0].min(); dur = (tb[:, 1] - tb[:, 0]) / 100.0; st = (tb[:, 0] - t0) / 100.0
        print('blocks %d: dur us mean %.1f p50 %.1f p90 %.1f max %.1f; sum/256 = %.1f us; last end %.1f us' % (len(dur), dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / 256, ((tb[:, 1] - t0) / 100.0).max()))
        print('start times us (every 64th block):', [round(x, 1) for x in st[::64]])
    import ctypes
    from mhhip import _lib as _ll
    if hasattr(_ll.lib(), 'mh_debug_counters'):
        buf = (ctypes.c_ulonglong * 8)()
        _ll.lib().mh_debug_counters(buf); c0 = list(buf)
        r(e, gv, log, with_grads=False); torch.cuda.synchronize()
        _ll.lib().mh_debug_counters(buf); c1 = list(buf)
        print('counters per call: rounds %d pairs %d runpath-rounds %d faces-with-cands %d kept-after-cull %d' % tuple(c1[i] - c0[i] for i in range(5)))
    print('lbs fwd ms', timeit(lambda: e.forward()))
    from mhhip import _lib as _l
    from mhhip._lib import ptr as _p, check as _c
    def bwd():
        g = e.grads
        _c(_l.lib().mh_lbs_backward(e.m.handle, e.B, e.N, _p(e.leaf('betas')), _p(e.leaf('poses_smpl')), _p(e.leaf('xscale')), _p(e.leaf('poses_T')), _p(e.vposed), _p(gv), _p(e.gj), _p(e.leaf('poses_smpl', g)), _p(e.leaf('poses_T', g)), _p(e.leaf('betas', g)), _p(e.leaf('xscale', g)), _p(e.ws), _p(e.ws2), _l.stream_ptr(e.dev)))
    print('lbs bwd ms', timeit(bwd))
    from mhhip import _lib
    from mhhip._lib import ptr, check
    L = _lib.lib(); st = _lib.stream_ptr(e.dev)
    print('lowest ms', timeit(lambda: check(L.mh_lowest_vertex(ptr(e.verts), e.B, e.V, ptr(e.low_idx), ptr(e.low_xyz), st))))
    print('knn ms (M=%d)' % e.scene_pts.shape[0], timeit(lambda: check(L.mh_contact_knn(ptr(e.scene_pts), e.scene_pts.shape[0], ptr(e.low_xyz), e.B, 32, ptr(e.dy), st))))
    print('knn grid ms', timeit(lambda: check(L.mh_contact_knn_grid(ptr(e.scene_grid), e.scene_pts.shape[0], ptr(e.low_xyz), e.B, 32, ptr(e.dy), st))))
    print('grid build ms', timeit(lambda: e._build_scene_grid()))
    if hasattr(L, 'mh_debug_tq'):
        import ctypes as _c2
        check(L.mh_contact_knn_grid(ptr(e.scene_grid), e.scene_M, ptr(e.low_xyz), e.B, 32, ptr(e.dy), st)); torch.cuda.synchronize()
        b1 = (_c2.c_ulonglong * 8192)(); b2 = (_c2.c_int * 4096)(); L.mh_debug_tq(b1, b2)
        tq = np.array(list(b1), dtype=np.float64).reshape(4096, 2)[:e.B]; fnd = np.array(list(b2))[:e.B]
        dur = (tq[:, 1] - tq[:, 0]) / 100.0; t0 = tq[:, 0].min()
        print('knn queries: dur us mean %.1f p50 %.1f p90 %.1f max %.1f; start spread %.1f us; last end %.1f; points scanned mean %.0f max %d' % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), (tq[:, 0].max() - t0) / 100.0, (tq[:, 1].max() - t0) / 100.0, fnd.mean(), fnd.max()))
    e.scene_device_setup(seq['backmasks'])
    def scene_upd():
        e.scene_device_update(); e._scene_dev['stream'].synchronize()
    print('device scene update ms (median+postprocess+points+grid, own stream, synced)', timeit(scene_upd))
    from mhhip._lib import check as _ck
    d = e._scene_dev
    Ls = _l.lib() if False else __import__('mhhip._lib', fromlist=['lib']).lib()
    stp = __import__('mhhip._lib', fromlist=['lib']).stream_ptr(e.dev)
    print('  median ms', timeit(lambda: _ck(Ls.mh_scene_median(e.T, e.H, e.W, ptr(e.depths), ptr(d['back']), ptr(d['sets'][0]['zsnap'][:e.T]), ptr(d['sets'][0]['zsnap'][e.T:]), ptr(d['ma_depth']), ptr(d['ma_mask']), ptr(d['ws']), stp))))
    print('  postprocess ms', timeit(lambda: _ck(Ls.mh_scene_postprocess(e.H, e.W, ptr(d['ma_depth']), ptr(d['ma_mask']), 1, 7, ptr(d['depth']), ptr(d['ws']), stp))))
    hdr = e.scene_grid[:32].cpu().numpy()
    print('grid mn', hdr[:12].view(np.float32), 'cell', hdr[12:16].view(np.float32), 'dim', hdr[16:28].view(np.int32), 'ncells', hdr[28:32].view(np.int32))
    d = torch.cdist(e.low_xyz.view(-1, 3), e.scene_pts)
    kd = d.topk(32, largest=False).values
    print('query->nearest: mean %.3f max %.3f; 32nd: mean %.3f max %.3f' % (kd[:, 0].mean(), kd[:, 0].max(), kd[:, 31].mean(), kd[:, 31].max()))
    print('points within 32nd radius+cell', float((d < (kd[:, 31:32] + float(hdr[12:16].view(np.float32)[0]))).sum(1).float().mean()))


if __name__ == '__main__':
    main()


def window_stats():
    pass
