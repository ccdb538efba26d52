"""developer experiment: do two half-size sequences replayed on two streams finish sooner than one after the other?
(how much of the cycle's serial chain -- LBS forward -> face lists -> selection -> gradients -> LBS backward -- could
hide under another half's selection kernel)"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq

def make(T, seed):
    struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(bench.IMG, 60.0)
    opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, seed, cam_K=K)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    opt._stage_from_dataloader(torch.utils.data.DataLoader(synthetic_seq.ShardDataset(seq) if hasattr(synthetic_seq, 'ShardDataset') else synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False))
    W, H = bench.IMG
    opt.scene_depth = bench.ground_scene(K, W, H)
    opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)
    e, sh = opt.engine, opt.sh
    r = e.raster_terms()
    sh.update_filters()
    return opt, e, sh, r

def run(items, streams, n):
    for i in range(n):
        for (opt, e, sh, r), st in zip(items, streams):
            with torch.cuda.stream(st):
                sh.cycle(1 + i % 20, raster=r, graphs=True)
                sh.step(0.005)

def timeit(items, streams, n=200):
    run(items, streams, 30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(items, streams, n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

full = make(200, 1003)
s0 = torch.cuda.Stream()
print('one sequence of 200 frames        : %.4f ms per cycle' % timeit([full], [s0]))
a, b = make(100, 1003), make(100, 1004)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print('two of 100 frames, same stream    : %.4f ms per pair of cycles' % timeit([a, b], [sa, sa]))
print('two of 100 frames, two streams    : %.4f ms per pair of cycles' % timeit([a, b], [sa, sb]))
for off_us in (100, 200, 300):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        torch.cuda._sleep(int(off_us * 2400))          # ~2.4 GHz: start the second sequence off_us late
    print('two streams, second one %3d us late: %.4f ms per pair of cycles' % (off_us, timeit([a, b], [sa, sb])))
print('one of 100 frames alone           : %.4f ms per cycle' % timeit([a], [sa]))
