"""developer tool: does any result of a cycle depend on what uninitialised device memory held?  ``torch.empty`` /
``empty_like`` are wrapped to fill every new CUDA allocation with a poison pattern (POISON=0: zeros, 1: 0xFF bytes = NaN,
2: random bytes); one C3-shaped cycle (T frames) runs eagerly with the deterministic scatter and prints hashes of the
gradients, per-body loss values and selection keys.  Equal hashes for all POISON values = no such dependence."""
import hashlib, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
mode = int(os.environ.get('POISON', '0'))
_empty, _empty_like = torch.empty, torch.empty_like
def _poison(t):
    if t.is_cuda and t.numel():
        b = t.view(-1).view(torch.uint8) if t.is_contiguous() else None
        if b is not None:
            if mode == 0: b.zero_()
            elif mode == 1: b.fill_(255)
            else: b.copy_(torch.randint(0, 256, b.shape, dtype=torch.uint8, device=t.device))
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
import bench
from mhhip import synthetic, synthetic_seq
from mhhip.raster import set_deterministic
T = int(os.environ.get('T', 60))
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, bench.N_PEOPLE, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=20)
dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=bench.BATCH, shuffle=False)
opt._stage_from_dataloader(dl)
opt.scene_depth = bench.ground_scene(K, *bench.IMG)
opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)
e = opt.engine
raster = e.raster_terms()
e.update_filters()
set_deterministic(True)
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
for c in range(int(os.environ.get('CYCLES', 3))):
    e.cycle(c, raster=raster)
    torch.cuda.synchronize()
    win, koff, keys = raster.selection(e)
    import ctypes
    from mhhip import _lib
    off = (ctypes.c_size_t * 6)()
    _lib.lib().mh_raster_debug_offsets.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
    _lib.check(_lib.lib().mh_raster_debug_offsets(*raster.dims, off))
    B, V, F, H = e.B, e.V, int(raster.faces.shape[0]), e.H
    seg = lambda o, n: raster.ws[off[o]:off[o] + n]
    fsort = seg(2, B * F * 4).view(torch.int32).view(B, F)
    rs = seg(3, B * (4 * (H + 1) + 1) * 4).view(torch.int32).view(B, -1)
    # the face lists as SETS per body (their order inside a row is whatever the sort's atomics gave)
    nf = rs[:, -1].clamp(0, F)
    fs_sorted = torch.where(torch.arange(F, device=e.dev)[None] < nf[:, None], fsort, torch.full_like(fsort, 2 ** 31 - 1)).sort(dim=1).values
    print('   verts', h(e.verts), 'ndc', h(seg(0, B * V * 12)), 'frows', h(seg(1, B * F * 4)), 'row_start', h(rs), 'fsort(set)', h(fs_sorted),
          'maxh', h(seg(4, B * 4)), 'win', hashlib.sha1(win.tobytes()).hexdigest()[:12], 'npx', int(koff[-1]))
    print('POISON', mode, 'cycle', c, 'grads', h(e.grads), 'depth', h(e.depth_body), 'sil', h(e.sil_body), 'keys', hashlib.sha1(keys.tobytes()).hexdigest()[:12],
          'gverts', h(e.gverts), 'loss2d', h(e.loss2d))
    e.step(0.01)
