"""developer tool: is the selection bit-stable in ONE process while other processes merely hold contexts on the same GPU?
usage: python tools/multiproc_probe.py <idle processes> [iterations]
The parent runs the C3 selection (preparation + selection kernel + sums) over and over and compares a per-body checksum of
the keys and the depth sums with the first launch; the children create a context and a few streams, launch one small
kernel on each and then sleep (what the ranks of the eight-process dry run do while another rank computes)."""
import os, subprocess, sys, tempfile, time
if len(sys.argv) > 1 and sys.argv[1] == 'idle':
    import torch
    ss = [torch.cuda.Stream() for _ in range(4)]
    x = torch.zeros(1024, device='cuda:0')
    for s in ss:
        with torch.cuda.stream(s):
            x.add_(1.0)
    torch.cuda.synchronize()
    print('ready', flush=True)
    busy = os.environ.get('PROBE_BUSY') == '1'
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        if busy:
            x.add_(1.0); torch.cuda.synchronize()
        time.sleep(0.002 if busy else 0.2)
    sys.exit(0)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
nidle = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'idle', '600'], stdout=subprocess.PIPE, text=True) for _ in range(nidle)]
for k in kids:
    k.stdout.readline()
import bench
from mhhip import synthetic, synthetic_seq, _lib
from mhhip.raster import RasterTerms
import ctypes
T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False))
e = opt.engine
r = RasterTerms(e)
e.cycle(0, raster=r)
torch.cuda.synchronize()
B, H, W = e.B, e.H, e.W
off = (ctypes.c_size_t * 3)()
_lib.check(_lib.lib().mh_raster_workspace_offsets(*r.dims, off))
raw = r.ws[off[2]:off[2] + B * H * W * 40].view(torch.int64).view(B, -1)
gv = torch.zeros_like(e.verts); log = torch.zeros(16, device=e.dev)
def launch():
    gv.zero_()
    for k in ('zmin_lin', 'zmax_lin'):
        e.leaf(k, e.grads).zero_()
    r(e, gv, log, phases=3)
    return raw.sum(dim=1).clone(), \
        torch.cat([e.leaf('zmin_lin', e.grads).view(-1), e.leaf('zmax_lin', e.grads).view(-1), e.depth_body.view(-1), e.sil_body.view(-1)]).clone()
ref, gref = launch()
bad, gbad = 0, 0.0
t0 = time.time()
for i in range(iters):
    cur, g = launch()
    d = (cur != ref).nonzero().view(-1)
    if d.numel():
        bad += 1
        print('launch', i, 'bodies whose keys differ:', d.tolist()[:16], flush=True)
    gbad = max(gbad, float(((g - gref).abs() / gref.abs().max()).max()))
print('idle processes', nidle, 'launches', iters, 'launches with different keys:', bad, 'largest relative difference of the depth-range gradients and per-body sums %.2e' % gbad,
      '(%.1f s)' % (time.time() - t0))
for k in kids:
    k.kill()
