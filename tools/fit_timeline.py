"""developer tool: where is the GPU idle during a fit(250)?
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o t -- python tools/fit_timeline.py run
  python tools/fit_timeline.py show <dir>
'run' does two fits in one process (the second is what a process that has built optimisers before sees) and prints wall times;
'show' lists every hole of more than 0.3 ms in the union of all kernels and copies, with what ran before and after it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]


def run():
    import tempfile
    import numpy as np, torch
    import bench
    from mhhip import synthetic, synthetic_seq
    T = 200
    struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(bench.IMG, 60.0)
    for rep in range(2):
        opt = bench.build_optimizer(struct, regs, tempfile.mkdtemp(), T, 'cuda:0', K)
        opt.scene_update = 'device'
        seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, 4, T, bench.IMG, 1003, cam_K=K)
        dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
        opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
        opt._stage_from_dataloader(dl)
        torch.cuda.synchronize()
        # a marker the trace can find: 7 launches of the spin kernel in a row
        from mhhip import queues
        for _ in range(7):
            queues.shares(None, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.fit(dl, num_iter=250)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print('fit %d: wall %.1f ms' % (rep, (t1 - t0) * 1e3))


def show(d):
    import csv, glob
    rows = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
    for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy', r.get('Direction', '') + ' ' + r.get('Name', '')[:40]))
    rows.sort()
    t0 = rows[0][0]
    # fits start behind the markers (k_spin launches); rmsprop launches count the cycles
    end = rows[0][1]
    ncyc = 0
    for i, (s, e, q, n) in enumerate(rows):
        if 'k_rmsprop' in n:
            ncyc += 1
        if s - end > 300e3:
            prev = [r for r in rows[max(0, i - 3):i]]
            print('hole of %7.2f ms at %9.2f ms (after %d updates): before %s | after %s %s' % (
                (s - end) / 1e6, (end - t0) / 1e6, ncyc, '; '.join('%s %s' % (r[2], r[3][:28]) for r in prev[-2:]), q, n[:40]))
        end = max(end, e)
    print('%d kernels/copies, %d updates, span %.1f ms' % (len(rows), ncyc, (end - t0) / 1e6))


if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else show(sys.argv[2])
