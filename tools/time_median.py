"""developer tool: stand-alone time of the temporal-median kernel (mh_scene_median_t) at the C3 size, on depths that differ in
every frame and on depths that repeat (the descent ends early on the first, runs all varying bits on ties)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'scene-aware-3d-multi-human_amd'))
from mhhip import _lib as _l
L = _l.lib()
T, H, W = 200, 135, 240
dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
for name in ('distinct', 'quantised', 'bench-like'):
    dn = rng.uniform(0, 1, (H * W, T)).astype(np.float32)
    if name == 'quantised':
        dn = np.round(dn * 16) / 16
    if name == 'bench-like':
        base = rng.uniform(0.2, 0.8, (H * W, 1)).astype(np.float32)
        dn = (base + 0.01 * rng.standard_normal((H * W, T))).astype(np.float32)
    back = (rng.uniform(0, 1, (H * W, T)) > 0.2).astype(np.uint8)
    zmin = rng.uniform(0.5, 1.5, T).astype(np.float32)
    zmax = rng.uniform(4, 9, T).astype(np.float32)
    t = lambda a: torch.tensor(a, device=dev)
    ws = torch.empty(L.mh_scene_workspace_bytes(T, H, W), dtype=torch.uint8, device=dev)
    md, mm = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    a, b, c, d = t(dn), t(back), t(zmin), t(zmax)
    st = _l.stream_ptr(dev)
    call = lambda: _l.check(L.mh_scene_median_t(T, H, W, _l.ptr(a), _l.ptr(b), _l.ptr(c), _l.ptr(d), _l.ptr(md), _l.ptr(mm), _l.ptr(ws), st))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(name, 'us per launch (incl. the ranges kernel): %.1f' % (e0.elapsed_time(e1) * 1e3 / 20), 'checksum %.6f' % float(md.double().sum()))
