"""developer tool: where the wall time of one fit(250) goes -- GPU time between consecutive cycle launches (events), the host
time of the calls that are not cycles (graph captures, filter updates, scene bookkeeping)"""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq

T = 200
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
def shares(a, b, spin=int(1.0e6)):
    torch.cuda.synchronize()
    e0, ea1, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record(a); torch.cuda._sleep(spin); ea1.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(1000); eb1.record(b)
    torch.cuda.synchronize()
    return int(e0.elapsed_time(eb1) >= 0.8 * e0.elapsed_time(ea1))


def one(n):
    opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
    opt.scene_update = 'device'
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    evs, hosts = [], []
    orig = e.cycle_graphed
    def wrapped(*a, **k):
        ev = torch.cuda.Event(enable_timing=True); ev.record(); evs.append(ev)
        t = time.perf_counter(); r = orig(*a, **k); hosts.append(time.perf_counter() - t)
        return r
    e.cycle_graphed = wrapped
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.fit(dl, num_iter=n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    end = torch.cuda.Event(enable_timing=True); end.record(); torch.cuda.synchronize()
    d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)] + [evs[-1].elapsed_time(end)])
    h = np.array(hosts) * 1e3
    print('fit(%d): wall %.1f ms; %d cycle launches; GPU time between launches: median %.3f ms, sum %.1f ms' % (n, (t1 - t0) * 1e3, len(d), np.median(d), d.sum()))
    print('  cycles  0-29: mean %.3f ms; 30-59: %.3f; 60-249: %.3f' % (d[:30].mean(), d[30:60].mean(), d[60:].mean()))
    big = np.argsort(-d)[:12]
    print('  longest gaps (cycle: ms [host ms of the launch call]):', ', '.join('%d: %.2f [%.2f]' % (i, d[i], h[i]) for i in sorted(big)))
    print('  host time inside cycle_graphed: sum %.1f ms, median %.3f ms' % (h.sum(), np.median(h)))
    from mhhip import sequence, queues
    main = torch.cuda.current_stream()
    st = dict(sequence._SHARED_STREAMS)
    scene = e._scene_dev['stream']
    print('  hardware-queue sharing: scene~main %d, ' % shares(main, scene) + ', '.join('%s~main %d' % (k[1], shares(main, v)) for k, v in st.items())
          + ', side~scene %d' % shares(st[(0, 'side')], scene) + '; plan: %s' % queues.plan('cuda:0').stats
          + ' lane test ms: %s' % [[round(x, 2) for x in lt.ms] for lt in getattr(e, '_lane_tests', {}).values() if getattr(lt, 'ms', None)])
one(250)
one(250)
