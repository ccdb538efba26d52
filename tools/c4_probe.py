"""developer aid for the eight-process dry run ("a multi-process artefact"): called by bench.py in a rank's turn when
MHHIP_C4_PROBE is set -- the parity cycle is repeated from the same (unstepped) leaves and every repetition is held against
the first one: per-body checksum of the selection keys (exact), per-body depth / silhouette sums and the depth-range
gradients (deterministic mode: exact).  A transient shows as repetitions that disagree with each other."""
import ctypes, os
import numpy as np, torch
from mhhip import _lib


def snapshot(e, raster):
    B, H, W = e.B, e.H, e.W
    off = (ctypes.c_size_t * 3)()
    _lib.check(_lib.lib().mh_raster_workspace_offsets(*raster.dims, off))
    win = raster.ws[off[0]:off[0] + B * 16].view(torch.int32).view(B, 4).clone()
    raw = raster.ws[off[2]:off[2] + B * H * W * 40].view(torch.int64).view(B, H * W, 5)
    # only the window's pixels are this launch's (the rest of a body's region is older)
    npx = (win[:, 2].clamp(min=0) * win[:, 3].clamp(min=0)).to(torch.int64)
    idx = torch.arange(H * W, device=raw.device)[None, :]
    live = (idx < npx[:, None])
    kpx = raw.sum(dim=2) * live                     # per window pixel
    ksum = kpx.sum(dim=1)
    g = e.grads
    t = raster.forward_targets()
    o = t.ndc - raster.ws.data_ptr()
    ndc = raster.ws[o:o + B * e.V * 12].view(torch.int32).to(torch.int64).view(B, -1).sum(dim=1)
    hv = getattr(e, '_halo_verts', None)
    extra = {} if hv is None else {'halo_verts': hv.view(torch.int32).to(torch.int64).view(hv.shape[0], -1).sum(dim=1)}
    return {**extra, 'win': win, 'keys': ksum.clone(), 'ndc': ndc, '_kpx': kpx, '_raw': raw.clone() if os.environ.get('MHHIP_C4_RAW') else raw[:0], 'depth_body': e.depth_body.view(-1).clone(), 'sil_body': e.sil_body.view(-1).clone(),
            'gzmin': e.leaf('zmin_lin', g).view(-1).clone(), 'gzmax': e.leaf('zmax_lin', g).view(-1).clone(),
            'gposes_T': e.leaf('poses_T', g).reshape(-1).clone(), 'verts': e.verts.view(torch.int32).to(torch.int64).view(B, -1).sum(dim=1)}


def repeat_and_compare(rank, e, sh, raster, reps=None):
    reps = int(os.environ.get('MHHIP_C4_PROBE', '3')) if reps is None else reps
    if os.environ.get('MHHIP_C4_SERIAL') == '1':      # the side branch on the chain's own stream: nothing runs beside the rasteriser
        e._side_stream = lambda: torch.cuda.current_stream(e.dev)
        e.grads.zero_()
        sh.cycle(0, raster=raster, graphs=False)
        torch.cuda.synchronize()
    if os.environ.get('MHHIP_C4_SERIAL') == '2':      # the neighbours' boundary vertices are not skinned again (kept from the first cycle)
        e._halo_forward = lambda h, st: None
    fake = int(os.environ.get('MHHIP_C4_FAKEHALO', '0'))
    if fake:      # one process: an LBS forward of `fake` bodies in the side branch, as the frame-sharded run has for its neighbours' frames
        e.halo = dict(poses=torch.zeros(fake, 72, device=e.dev), transl=torch.tensor([[0., 1., 5.]] * fake, device=e.dev),
                      has_prev=False, has_next=False)
        sh = type('S', (), {'cycle': staticmethod(lambda row, raster=None, graphs=False: e.cycle(row, raster=raster))})
        kind = os.environ.get('MHHIP_C4_FAKEKIND', 'lbs')
        if kind == 'matmul':        # some other kernel beside the rasteriser (rocBLAS: LDS + MFMA)
            ma = torch.randn(2048, 2048, device=e.dev); mb = torch.randn(2048, 2048, device=e.dev); mc = torch.empty(2048, 2048, device=e.dev)
            e._halo_forward = lambda h, st: torch.mm(ma, mb, out=mc)
        elif kind == 'copy':        # a streaming kernel without LDS
            ca = torch.randn(64 << 20, device=e.dev); cb = torch.empty_like(ca)
            e._halo_forward = lambda h, st: cb.copy_(ca)
        elif kind in ('scratch', 'noscratch'):     # tools/ubench/scratch_corunner.hip: arithmetic on a private array (in scratch / in registers)
            co = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ubench', 'scratch_corunner.so'))
            co.corunner_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            cidx = torch.randint(0, 64, (1024,), dtype=torch.int32, device=e.dev); cout = torch.empty(2048 * 256, device=e.dev)
            e._halo_forward = lambda h, st: co.corunner_launch(1 if kind == 'scratch' else 0, cidx.data_ptr(), cout.data_ptr(), 2048,
                                                               int(os.environ.get('MHHIP_C4_ITERS', '2000')), st)
        elif kind.startswith('lds'):               # lds67072 / lds61440 ...: a workgroup that only fills and sums that much LDS
            co = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ubench', 'scratch_corunner.so'))
            co.corunner_launch_lds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            cout = torch.empty(4096 * 512, device=e.dev)
            nbytes = int(kind[3:])
            e._halo_forward = lambda h, st: co.corunner_launch_lds(cout.data_ptr(), 1024, nbytes, int(os.environ.get('MHHIP_C4_ITERS', '20')), st)
        if os.environ.get('MHHIP_C4_PATH'):
            _lib.lib().mh_raster_set_path(int(os.environ['MHHIP_C4_PATH']))
        e.grads.zero_()
        sh.cycle(0, raster=raster, graphs=False)
        torch.cuda.synchronize()
    first = snapshot(e, raster)
    N = e.N
    for r in range(reps):
        e.grads.zero_()
        sh.cycle(0, raster=raster, graphs=False)
        torch.cuda.synchronize()
        cur = snapshot(e, raster)
        L_ = _lib.lib()
        if hasattr(L_, 'mh_raster_debug_verify'):        # -DR_VERIFY build of the selection kernel
            buf = (ctypes.c_uint * 32)()
            L_.mh_raster_debug_verify(buf)
            if any(buf[:3]):
                print('[c4_probe] rank %d repetition %d: verify counters gather %d staging %d 1/area %d | first records %s' %
                      (rank, r + 1, buf[0], buf[1], buf[2], [hex(x) if i in (14, 15) else int(x) for i, x in enumerate(buf[3:16], 3)]), flush=True)
        for k in first:
            a, b = first[k], cur[k]
            if k == '_raw':
                continue
            if k == '_kpx':
                for bd in (a != b).any(dim=1).nonzero().view(-1).tolist():
                    px = (a[bd] != b[bd]).nonzero().view(-1)
                    ww = int(first['win'][bd, 2])
                    xs, ys = (px % ww), (px // ww)
                    print('[c4_probe] rank %d repetition %d: body %d window %s: %d pixels differ, x %d..%d y %d..%d' %
                          (rank, r + 1, bd, first['win'][bd].tolist(), px.numel(), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())), flush=True)
                    if first['_raw'].numel():
                        for q in px[:8].tolist():
                            ka, kb = first['_raw'][bd, q].tolist(), cur['_raw'][bd, q].tolist()
                            fmt = lambda k: 'empty' if k == -1 else 'f%d z%.6f' % (k & 0xffffffff, np.array([(k >> 32) & 0xffffffff], np.uint32).view(np.float32)[0])
                            print('[c4_probe]    pixel (%d,%d): first [%s] | now [%s]' % (q % ww, q // ww, ', '.join(fmt(k) for k in ka), ', '.join(fmt(k) for k in kb)), flush=True)
                continue
            if a.dtype.is_floating_point:
                bad = ((a - b).abs() > 0).nonzero().view(-1)
            else:
                bad = (a != b).view(a.shape[0], -1).any(dim=1).nonzero().view(-1)
            if bad.numel():
                per = 1 if k in ('win', 'halo_verts') else a.numel() // (e.T if k in ('gzmin', 'gzmax') else e.B)
                rows = sorted(set((bad // max(per, 1)).tolist()))
                rel = float(((a - b).abs().max() / a.abs().max())) if a.dtype.is_floating_point else -1.0
                print('[c4_probe] rank %d repetition %d: %s differs from the first cycle in %d rows %s (max rel %.1e)' %
                      (rank, r + 1, k, len(rows), rows[:12], rel), flush=True)
    print('[c4_probe] rank %d done (%d repetitions)' % (rank, reps), flush=True)
