"""End-to-end wall time of the drop-in optimiser on the C3 workload (developer tool): warm-up (100 Adam iterations on the
global translations) + fit(250 cycles) with the organic scene path and the one-euro filters, as predict.py would call it."""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import bench
from mhhip import synthetic, synthetic_seq


def main():
    T = int(os.environ.get('T', '200'))
    struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(bench.IMG, 60.0)
    tmp = tempfile.mkdtemp()
    opt = bench.build_optimizer(struct, regs, tmp, T, 'cuda:0', K)
    opt.scene_update = os.environ.get('SCENE', 'device')
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    n = int(os.environ.get('CYCLES', '250'))
    sect = {}
    def timed(name, fn):
        def w(*a, **k):
            torch.cuda.synchronize(); ta = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); sect[name] = sect.get(name, 0.0) + time.perf_counter() - ta
            return r
        return w
    opt._stage_from_dataloader = timed('staging', opt._stage_from_dataloader)
    opt._finish_scene = timed('final scene image (host)', opt._finish_scene)
    opt.engine.update_filters = timed('filter updates', opt.engine.update_filters)
    log = opt.fit(dl, num_iter=n)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = opt.get_optimized_variables()
    t3 = time.perf_counter()
    print('warm-up 100 it: %.3f s; fit %d cycles: %.3f s (%.2f ms/cycle incl. staging, graph capture, scene, filters); outputs %.3f s'
          % (t1 - t0, n, t2 - t1, 1e3 * (t2 - t1) / n, t3 - t2))
    print('  of which:', {k: round(v, 3) for k, v in sect.items()})
    print('first/last loss_pose24j %.5f -> %.5f ; reg_contact last %.4f ; scene points %d' % (
        log[0]['loss_pose24j'], log[-1]['loss_pose24j'], log[-1]['reg_contact'], opt.scene_pcd.shape[2] if opt.scene_pcd is not None else 0))


if __name__ == '__main__':
    main()
