#!/usr/bin/env python
"""Headline benchmark: optimizer iterations/sec of the scene-constrained SMPL optimisation loop
(one iteration = one ``fit`` cycle: all frames' residuals with the full nine-term loss stack,
hand-written backward, one RMSprop step; reference optimizer.py:375-587) on the MuPoTs-shaped
configuration BASELINE.json quotes the target on: 4 humans x 200 frames at 240x135, batch 10.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: ONE contiguous sequence, frames sharded over the ranks through the drop-in's own distributed path
(mhmocap/optimizer.py -> mhhip/sharded.py): one RCCL all-reduce on the shared shape/scale gradients per iteration
plus the one-frame halos.  Default = weak scaling, every rank owns 200 frames of a 4 x 200*N sequence (N=1 is C3);
``value`` counts iterations/sec in units of the 4x200 configuration (iterations/sec x frames/200), i.e. at N=1 it
is plain iterations/sec.  ``--config c4``: 250 frames per GPU (8 GPUs = BASELINE C4, 4 x 2000).  ``--strong``:
BASELINE C5 -- a fixed 8 humans x 500 frames job with a 200 000-point scene cloud, split over the ranks.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'))
sys.path.insert(0, ROOT)

COEFS = dict(proj2d=1.0, depth=0.05, silhouette=0.1, reg_poses=0.002, reg_scales=1e-4, reg_velocity=0.05,
             reg_verts_filter=0.002, reg_contact=0.001, reg_foot_sliding=0.01)   # configs/predict_mupots.yml:17-25
N_PEOPLE, T_LOCAL, IMG, BATCH = 4, 200, (240, 135), 10
# algorithmic bytes / flops of the LBS+projection kernels per human.frame.iteration (fwd+bwd)
PEAK_HBM_GBS = 8000.0
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_VALU_GIPS = 614.4     # 256 CUs x 4 SIMDs x 2.4 GHz, one wave64 vector instruction per 4 cycles


def build_optimizer(struct, regs, tmp, num_frames, device, K):
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(os.path.join(tmp, fn), regs[k])
    c = COEFS
    return SMPLDepthSequenceOptimizer(
        image_size=IMG, num_frames=num_frames, cam_K=K, device=device, smpl_model_parameters_path=tmp,
        smpl_data_struct=struct, scene_update='none', shard_frames=True, proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'],
        silhouette_loss_coef=c['silhouette'], reg_velocity_coef=c['reg_velocity'],
        reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'], reg_scales_coef=c['reg_scales'],
        reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])


def ground_scene(K, W, H):
    ys = (np.arange(H, dtype=np.float32) + 0.5 - K[1, 2]) / K[1, 1]
    d = np.minimum(np.where(ys[:, None] > 1e-3, 1.15 / np.maximum(ys[:, None], 1e-3), 10.0), 10.0)
    return np.tile(d, (1, W)).astype(np.float32)


def load_pmc_traffic():
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE and
    WRITE_SIZE need separate passes, profiles/README.md); None when the file is absent."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def file_sha(rel):
    import hashlib
    with open(os.path.join(ROOT, rel), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def profile_stamp():
    """profiles/stamp.json (written by tools/make_profiles.py): git hash and sha256 of every kernel source the committed
    counter passes were taken on.  Counter-derived numbers are only printed for sources that still have that hash."""
    path = os.path.join(ROOT, 'profiles', 'stamp.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def counters_fresh(stamp, src):
    rel = 'scene-aware-3d-multi-human_amd/csrc/' + src
    return bool(stamp) and stamp.get('sha256', {}).get(src) == file_sha(rel)


def load_profile_kernel_us(kernel):
    """average duration (us) of `kernel` inside REPLAYED cycles from the newest committed rocprofv3 --kernel-trace --stats
    summary (profiles/rNN_bench_c3_kernel_stats.csv) -- HIP events cannot be read back from inside a replayed graph"""
    import csv, glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench_c3_kernel_stats.csv')))
    if not found:
        return None
    with open(found[-1]) as f:
        for row in csv.DictReader(f):
            if row.get('Name', '').startswith(kernel) or (kernel + '(') in row.get('Name', '') or (kernel + '<') in row.get('Name', ''):
                try:
                    return float(row['AverageNs']) / 1e3
                except (KeyError, ValueError):
                    return None
    return None


def load_pmc_valu(kernel):
    """Vector instructions per launch of `kernel` from the committed SQ counter pass (profiles/README.md)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_sq_counters.txt')))
    if not found:
        return None
    path = found[-1]                      # the newest round's pass
    cur = None
    with open(path) as f:
        for line in f:
            w = line.split()
            if line[:1] not in ' \t' and w:
                cur = w[0]
            elif cur == kernel and w and w[0] == 'SQ_INSTS_VALU':
                return float(w[1])
    return None


def cpu_baseline(struct, regs, K, seq, pT0, frames, cycles):
    """The oracle (CPU restatement, torch-CPU + C face selection) on a bounded sample of the same
    workload: `frames` frames x 4 humans at 240x135, full loss stack, `cycles` cycles."""
    from oracle import fit_oracle as fo, lbs_oracle as lo, raster_oracle as ro
    W, H = IMG
    model = lo.BodyModel(struct, regs)
    faces = np.asarray(struct.f).astype(np.int64)
    o = fo.SequenceOracle(model, IMG, frames, K, coefs=COEFS, rasteriser=ro.make_rasteriser(faces, K, IMG))
    o.xscale = torch.zeros(1, N_PEOPLE, 1, 1)
    sl = slice(0, frames)
    o.init_optimized_variables(seq['pose2d'][sl], seq['poses_smpl'][sl], seq['betas_smpl'][sl], seq['valid_smpl'][sl],
                               poses_T=pT0[sl])
    o.update_scene_pointcloud(ground_scene(K, W, H), seq['backmasks'][sl].min(axis=0) > 0)
    batches = []
    for s in range(0, frames, BATCH):
        b = slice(s, min(frames, s + BATCH))
        batches.append(dict(idxs=torch.arange(b.start, b.stop), pose2d=torch.tensor(seq['pose2d'][b]),
                            seg_mask=torch.tensor(seq['seg_mask'][b]), depths=torch.tensor(seq['depths'][b]),
                            poses_smpl=torch.tensor(seq['poses_smpl'][b])))
    o.cycle_grads(batches)                      # untimed first touch
    t0 = time.perf_counter()
    o.fit(batches, cycles)
    dt = time.perf_counter() - t0
    return o, dt / cycles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)      # a fit and a bit: 0.25 s timed at 1200 it/s, twelve filter updates inside
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fit', action='store_true', help='skip the fit_250 block (wall time of the drop-in fit call)')
    ap.add_argument('--eager', action='store_true', help='launch every kernel from the host instead of replaying captured graphs')
    ap.add_argument('--cpu-frames', type=int, default=0, help='frames of the CPU-oracle sample (0 = all of them: no extrapolation)')
    ap.add_argument('--cpu-cycles', type=int, default=2)
    ap.add_argument('--config', choices=['c3', 'c4'], default='c3', help='c3: 200 frames per GPU (N=1 is BASELINE C3); '
                    'c4: 250 frames per GPU (8 GPUs = BASELINE C4)')
    ap.add_argument('--strong', action='store_true', help='BASELINE C5: fixed 8 humans x 500 frames + 200k-point scene, '
                    'split over the ranks (strong scaling)')
    # developer switches (parity-test shapes, not bench lines): the JSON names the workload it actually ran
    ap.add_argument('--humans', type=int, default=None)
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--image', type=str, default=None)
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl', help='nccl = RCCL over xGMI (the measured '
                    'configuration).  gloo: DRY RUN of the multi-rank path -- every collective staged through host memory '
                    '(tests/hostdist.py); with --one-device all ranks share cuda:0, so the N-rank orchestration (halos, '
                    'all-reduce, filter hand-off) can be executed on a 1-GPU box.  Its numbers are not scaling results.')
    ap.add_argument('--one-device', action='store_true', help='every rank uses cuda:0 (only with --backend gloo)')
    ap.add_argument('--presteps', type=int, default=300, help='untimed steady-state cycles before the warm-up')
    ap.add_argument('--dump-leaves', type=str, default=None, help='(parity tests) save the whole-sequence leaves after the timed '
                    'region to this .npz (a collective in multi-rank runs; rank 0 writes)')
    args = ap.parse_args()
    global N_PEOPLE, T_LOCAL, IMG
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks (one process per GPU) under torch.distributed.run,
        # the launch line of the module docstring, on a free local port -- never a silent one-process run that prints n_gpus 1
        import socket
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks (run `python bench.py --gpus N`, or '
                         'torch.distributed.run --nproc-per-node N bench.py --gpus N)' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.config == 'c4':
        T_LOCAL = 250
    if args.strong:
        N_PEOPLE = 8
    if args.humans:
        N_PEOPLE = args.humans
    if args.frames:
        T_LOCAL = args.frames
    if args.image:
        IMG = tuple(int(x) for x in args.image.split('x'))
    T_TOTAL = (args.frames or 500) if args.strong else T_LOCAL * world

    import torch.distributed as dist
    assert args.backend == 'gloo' or not args.one_device, '--one-device needs --backend gloo (RCCL refuses two ranks per device)'
    dev_index = 0 if args.one_device else local_rank
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(dev_index)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group('gloo')
            sys.path.insert(0, os.path.join(ROOT, 'tests'))
            import hostdist                    # test infrastructure (tests/hostdist.py), only reachable through this dry-run switch
            dist = hostdist.install()          # host-staged collectives for the driver, the drop-in and this file
    device = 'cuda:%d' % dev_index
    torch.cuda.set_device(dev_index)
    if not args.one_device:
        assert torch.cuda.device_count() >= (world if world > 1 else 1), \
            '%d ranks on a node with %d visible GPUs (one process per GPU; --backend gloo --one-device is the dry run)' % (world, torch.cuda.device_count())
    # proof of what the collective layer saw: every rank's device, gathered through the process group itself
    rccl_info = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev_index)
        mine = [rank, dev_index, '%s %s' % (props.name, getattr(props, 'uuid', getattr(props, 'pci_bus_id', '')))]
        got = [None] * world
        import torch.distributed as tdist
        tdist.all_gather_object(got, mine)
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        assert int(probe.item()) == world, 'all-reduce over the process group summed %s ranks, not %d' % (probe.item(), world)
        rccl_info = {'ranks': int(tdist.get_world_size()), 'backend': str(tdist.get_backend()), 'devices': [g[1] for g in sorted(got)],
                     'device_ids': [g[2] for g in sorted(got)], 'allreduce_of_ones': int(probe.item()),
                     'one_device_dry_run': bool(args.one_device)}

    import tempfile
    from mhhip import build as mhbuild, synthetic, synthetic_seq, sharded
    if rank == 0:
        mhbuild.build()
    if world > 1:
        dist.barrier()
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(IMG, 60.0)
    tmp = tempfile.mkdtemp()
    W, H = IMG
    # the drop-in itself shards: every rank is handed the whole sequence's tracks / key-points (what predict.py does
    # under torchrun) and keeps its block of frames; replicas of betas|xscale are broadcast from rank 0
    opt = build_optimizer(struct, regs, tmp, T_TOTAL, device, K)
    model = opt.SMPLPY.body_model
    f0, f1 = sharded.shard_bounds(T_TOTAL, world, BATCH)[rank]
    seq = synthetic_seq.make_sequence(model, N_PEOPLE, T_TOTAL, IMG, 1003, cam_K=K, render_frames=(f0, f1))
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    pT0 = opt.get_optimized_variables()['poses_T']
    dl = torch.utils.data.DataLoader(synthetic_seq.ShardDataset(seq), batch_size=BATCH, shuffle=False)
    opt._stage_from_dataloader(dl)
    assert (opt.first_frame, opt.last_frame) == (f0, f1)
    e, sh = opt.engine, opt.sh
    scene_mask = torch.tensor(seq['backmasks'].min(axis=0), device=device, dtype=torch.int32)
    if world > 1:
        dist.all_reduce(scene_mask, op=dist.ReduceOp.MIN)          # background in EVERY frame of the sequence
    scene_mask = scene_mask.cpu().numpy() > 0
    opt.scene_depth = ground_scene(K, W, H)
    if args.strong:
        # C5: the reference ties the cloud size to the image (optimizer.py:609-613); 200 000 points are injected directly
        rng = np.random.RandomState(11)
        M = 200000
        pts = np.stack([rng.uniform(-6, 6, M), 1.15 + 0.02 * rng.randn(M), rng.uniform(2, 14, M)], 1).astype(np.float32)
        e.set_scene_points(torch.tensor(pts))
    else:
        opt.update_scene_pointcloud(opt.scene_depth, scene_mask)                       # contact term live
    raster = e.raster_terms()
    sh.update_filters()                                                                # filtered-vertex term live
    sh.refresh_halo()                                                                  # (collective, every rank: the cycles themselves issue none)
    nstep = [0]

    # one-euro filter updates every 25 cycles as in fit (optimizer.py:383-392); the phase is chosen so that the first one
    # falls inside the timed region whatever --steps is (the driver's 20 steps: at step 10)
    upd_at = args.warmup + min(10, max(args.steps // 2, 1))

    def one_cycle(c, graphs, scene=False):
        if c >= upd_at and (c - upd_at) % 25 == 0:
            sh.update_filters()
        # scene: the organic path of fit (cycle >= 30): scene rebuilt from the sequence every cycle, on its own stream, from
        # the leaves as they are before this cycle's step
        sh.cycle(c % e.log.shape[0], raster=raster, graphs=graphs, scene_update=scene)
        if scene:
            e.scene_device_swap()     # read by the next cycle's contact term (which waits on the update's event)
        sh.step(0.01 * 0.99 ** min(nstep[0], 250))      # RMSprop, ExponentialLR(0.99) on the host as in the reference (:355-356)
        nstep[0] += 1

    use_graphs = not args.eager
    if world == 1 and use_graphs:
        e.defer_person = True         # as fit does: the per-person gradient sums ride in the update's launch
    grad0 = None
    if args.dump_leaves:
        # (parity tests) the gradient of one UNSTEPPED cycle at the initial variables, whole sequence: identical inputs in a
        # one-process and an N-rank run (later cycles are not: RMSprop's sign-like first steps amplify rounding noise)
        from mhhip.raster import set_deterministic
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        old_det = set_deterministic(True)
        if world > 1 and args.one_device:
            # the dry run's ranks take turns for the cycle the parity test holds against the one-process run (what used to
            # look like the processes disturbing each other was packed fp32 arithmetic beside matrix instructions: DESIGN 7)
            for r_turn in range(world):
                if r_turn == rank:
                    sh.cycle(0, raster=raster, graphs=False)
                    torch.cuda.synchronize()
                    if os.environ.get('MHHIP_C4_PROBE'):      # developer aid (tools/c4_probe.sh): is the cycle self-consistent?
                        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
                        import c4_probe
                        c4_probe.repeat_and_compare(rank, e, sh, raster)
                dist.barrier()
        else:
            sh.cycle(0, raster=raster, graphs=False)
            if os.environ.get('MHHIP_C4_PROBE'):
                torch.cuda.synchronize()
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
                import c4_probe
                c4_probe.repeat_and_compare(rank, e, sh, raster)
        set_deterministic(old_det)
        tail = e.grads[e.shared_lo:].clone()
        if world > 1:
            dist.all_reduce(tail, op=dist.ReduceOp.SUM)
        grad0 = {'grad0_' + k: opt._gather_frames(e.leaf(k, e.grads)) for k in ('poses_T', 'poses_smpl', 'zmin_lin', 'zmax_lin')}
        grad0['grad0_tail'] = tail.cpu().numpy()
        e.grads.zero_()
    # bring the device to its steady state before the W warm-up steps: graph capture, lazy allocations, and enough
    # back-to-back work for the clocks to ramp (a fresh box that idled through the CPU-side set-up was once measured
    # at 0.57x for the first tens of milliseconds)
    for c_pre in range(args.presteps):  # a fixed count: every rank must issue the same collectives
        one_cycle(1 + c_pre % 20, use_graphs)
        if c_pre % 10 == 9:
            torch.cuda.synchronize()
    for c in range(args.warmup):
        one_cycle(c, use_graphs)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(args.steps):
        one_cycle(args.warmup + c, use_graphs)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        opt.check_replicas()
    if args.dump_leaves:
        opt._global_cache = None
        g_all = opt._global_leaves()              # collective
        if rank == 0:
            np.savez(args.dump_leaves, **{k: np.asarray(v) for k, v in g_all.items()}, **grad0)
    nsteps_org = min(args.steps, 50)
    # the same cycles with the device-side scene aggregation of optimizer.py:578-584 running every cycle (reported
    # beside the headline, which uses the injected static scene BASELINE.json's C3 names)
    organic = None
    if world == 1 and not args.strong:
        e.scene_device_setup(seq['backmasks'])
        for c in range(40):       # two captures (the injected scene, then the device-built sets) + lane test and lane picker of mhhip/queues.py
            one_cycle(args.warmup + args.steps + c, use_graphs, scene=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for c in range(nsteps_org):
            one_cycle(args.warmup + args.steps + 40 + c, use_graphs, scene=True)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        organic = {'value': round(nsteps_org / dt1, 3), 'ms_per_step': round(1e3 * dt1 / nsteps_org, 4), 'steps': nsteps_org,
                   'lane_test': [{'replay_ms': [round(x, 3) for x in getattr(lt, 'ms', [])], 'busy_lanes': len(getattr(lt, 'busy', [])),
                                  'candidates': len(lt.cand)} for lt in getattr(e, '_lane_tests', {}).values()],
                   'lane_pick': [{'cycle_ms': [round(x, 4) for x in (pk.ms or [])], 'best': pk.best} for pk in getattr(e, '_pickers', {}).values()],
                   'what': 'per-cycle masked median over the 200 frames + bilateral/Sobel/erode/median-fill + un-projection '
                           '+ grid rebuild on a second stream, overlapped with the next cycle'}
        gp = getattr(e, '_gate_probe', None)
        if gp:      # MHHIP_GATE_PROBE=1: what the main stream waited for the previous cycle's scene update, per cycle of the timed region
            organic['gate_wait_ms'] = round(sum(a.elapsed_time(b) for a, b in gp[-nsteps_org:]) / min(len(gp), nsteps_org), 4)
        opt.scene_depth = ground_scene(K, W, H)
        opt.update_scene_pointcloud(opt.scene_depth, scene_mask)   # back to the static scene
    # per-kernel durations: HIP events cannot be read back from inside a replayed graph, so the same
    # launch sequence runs once more eagerly with events around the kernel groups (same kernels, same
    # stream, same data; only the launch mechanism differs)
    e.enable_timing(True)
    import ctypes
    from mhhip import _lib
    L = _lib.lib()
    L.mh_profile_enable(1)
    prof_names = ['k_raster_strip', 'k_raster_grads', 'lbs_skin_forward', 'lbs_skin_backward', 'k_contact_knn_grid', 'k_raster_sums',
                  'k_pose_fwd', 'keypoint_terms', 'pose_bwd_reduce', 'raster_prepare_lists']
    prof = {k: [] for k in prof_names}
    for c in range(min(args.steps, 40)):
        one_cycle(args.warmup + args.steps + c, False)
        torch.cuda.synchronize()
        for i, k in enumerate(prof_names):      # duration of this cycle's launch (HIP events on the launch stream)
            ms1 = ctypes.c_float(0)
            if L.mh_profile_read(i, ctypes.byref(ms1)) == 0:
                prof[k].append(float(ms1.value))
    # the kernels of the LBS + projection unit once more ALONE on the device (inside the cycle the small ones run beside --
    # and are stretched by -- the selection kernel): forward pair, key-point launches, backward chain, 20 calls each
    alone = {k: [] for k in ('k_pose_fwd', 'lbs_skin_forward', 'keypoint_terms', 'lbs_skin_backward', 'pose_bwd_reduce')}
    if world == 1 and e.kp_fused:
        def read(keys):
            torch.cuda.synchronize()
            for k in keys:
                ms1 = ctypes.c_float(0)
                if L.mh_profile_read(prof_names.index(k), ctypes.byref(ms1)) == 0:
                    alone[k].append(float(ms1.value))
        st0 = _lib.stream_ptr(e.dev)
        for rep in range(22):
            e.forward(regress=False, raster=raster)
            if rep >= 2:
                read(['k_pose_fwd', 'lbs_skin_forward'])
            e.keypoint_terms(st0)
            if rep >= 2:
                read(['keypoint_terms'])
            e._finish_b(None, raster=raster)          # LBS backward chain on the gradients the last cycle left (timing only)
            if rep >= 2:
                read(['lbs_skin_backward', 'pose_bwd_reduce'])
        e.grads.zero_()
    L.mh_profile_enable(2)                # a few more cycles with the kernel's own work counters on (they cost it ~4 %: not timed)
    for c in range(3):
        one_cycle(args.warmup + args.steps + 40 + c, False)
    L.mh_profile_enable(0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pc = raster.pair_counters(e)          # counted by k_raster_strip at profile level 2
    sort_seen, sort_rebuilt = raster.sort_counters(e)
    kern = e.timing_summary()
    e.enable_timing(False)
    kernel_us = {k: 1e3 * float(np.mean(v)) for k, v in prof.items() if v}
    win = raster.ws[:e.B * 16].view(torch.int32).view(e.B, 4).cpu().numpy()
    window_px = int((np.maximum(win[:, 2], 0).astype(np.int64) * np.maximum(win[:, 3], 0)).sum())
    log = sh.read_log(1)

    if rank == 0:
        ms = 1e3 * dt / args.steps
        its = args.steps / dt
        frames_here = f1 - f0
        bodies = N_PEOPLE * frames_here
        V, F = int(e.V), int(raster.faces.shape[0])
        dom = max(kernel_us, key=kernel_us.get) if kernel_us else None
        traffic = load_pmc_traffic()
        pairs = None
        if pc[0] > 0 and 'k_raster_strip' in kernel_us:
            t_s = kernel_us['k_raster_strip'] * 1e-6
            pairs = {'candidate_pairs_per_launch': round(pc[1] / pc[0]), 'pairs_evaluated_per_launch': round(pc[2] / pc[0]),
                     'candidate_pairs_per_s': round(pc[1] / pc[0] / t_s, 1), 'pairs_evaluated_per_s': round(pc[2] / pc[0] / t_s, 1),
                     'what': '(face, pixel-centre) pairs inside the blurred bounding boxes / pairs that survive the depth cull and '
                             'are evaluated (barycentrics, distances, K-nearest insertion)'}
        # roofline of the dominant kernel (k_raster_strip) as the contract defines it: ALGORITHMIC bytes per launch (DESIGN.md
        # section 4: every body's projected vertices 12 B x V and row-sorted face list 4 B x F in, the 40-byte key record of
        # every window pixel out, the face table once) / the launch duration measured in THIS run (HIP events on the launch
        # stream) / 8 TB/s.  The kernel is a face-parallel z-buffer selection in LDS, bound by vector-instruction issue:
        # that utilisation is kept beside it ("valu": SQ_INSTS_VALU of the committed counter pass / this launch time
        # against one wave64 VALU instruction per 4 cycles per SIMD), and the pair-test rates SURVEY 8(d)(iv) asks for in
        # "pairs" (counted live by the kernel while the instrumented region runs).  Counter-derived entries are null when the
        # kernel source no longer has the hash the committed passes were taken on (profiles/stamp.json).
        stamp = profile_stamp()
        roof = None
        if 'k_raster_strip' in kernel_us:
            # launch duration INSIDE the cycle: (i) HIP events around the launch in eager cycles of this run -- same kernels,
            # same two queues, the side branch beside it; (ii) the rocprofv3 average over REPLAYED cycles of the committed
            # profile (profiles/, only while the kernel source still has the profiled hash).  frac uses the larger one.
            us_live = kernel_us['k_raster_strip']
            fresh = counters_fresh(stamp, 'mh_raster.hip')
            us_prof = load_profile_kernel_us('k_raster_strip') if (fresh and not args.strong and frames_here == 200 and N_PEOPLE == 4) else None
            us = max(us_live, us_prof or 0.0)
            algo = bodies * (12.0 * V + 4.0 * F) + 40.0 * window_px + 12.0 * F
            gbs = algo / (us * 1e-6) / 1e9
            roof = {'kernel': 'k_raster_strip', 'bound': 'valu-issue', 'contract_bound': 'hbm',
                    'bound_note': 'what limits the kernel is vector-instruction issue (face-parallel z-buffer selection in LDS: see "valu"); '
                                  'achieved / peak / frac are the contract\'s figures all the same: algorithmic bytes per launch over the '
                                  'launch duration against the HBM peak', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / PEAK_HBM_GBS, 4), 'traffic': None, 'traffic_detail': None,
                    'launch_us': round(us, 1), 'launch_us_events_in_eager_cycles': round(us_live, 1),
                    'launch_us_rocprof_replayed_cycles': None if us_prof is None else round(us_prof, 1),
                    'algorithmic_bytes': algo, 'window_pixels': window_px, 'dominant_by_events': dom,
                    'counters_taken_on': stamp.get('git') if fresh else None, 'valu': None, 'pairs': pairs}
            td = traffic.get('k_raster_strip') if fresh else None
            if td and not args.strong and frames_here == 200 and N_PEOPLE == 4:
                # HBM bytes per launch from the committed FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, KiB x 1024).  The
                # guide's gfx950 correction (x2) applies to wide 16 B/lane streaming reads only; this kernel reads 12-byte
                # gathers and 4-byte list entries, which are uncalibrated widths: the raw sum is printed
                roof['traffic'] = float(td['fetch_bytes'] + td['write_bytes'])
                roof['traffic_detail'] = td
            nv = load_pmc_valu('k_raster_strip') if fresh else None
            if nv and not args.strong and frames_here == 200 and N_PEOPLE == 4:
                gi = nv / (us * 1e-6) / 1e9
                roof['valu'] = {'what': 'vector-instruction issue utilisation (not a roofline fraction of useful work); the peak prices every '
                                        'wave64 instruction at 4 cycles per SIMD -- measured on gfx950 (tools/ubench/valu_rate.hip, '
                                        'profiles/r04_ubench_valu_lds.txt) fp32 fma / mul / add, integer add / logic, moves and cndmask '
                                        'issue every 2.3 cycles from two or more waves, everything else every 4.1-4.6: with the '
                                        "kernel's mix the issue bound is ~1.3x this peak and the kernel sits at ~60 % of it "
                                        '(DESIGN.md 4.1: the rest is the dependent chain of a 64-pair batch)',
                                'achieved': round(gi, 1), 'peak': PEAK_VALU_GIPS, 'unit': 'G wave-instr/s',
                                'utilisation': round(gi / PEAK_VALU_GIPS, 3), 'wave_instructions_per_launch': nv}
        # SURVEY 8(d) unit of the LBS + projection pair: 167 028 B per human.frame.iteration + the 19.35 MB of constants
        # once per launch pair, over forward + backward of the skinning kernels (HIP events around them)
        lbs = None
        if 'lbs_skin_forward' in kernel_us and 'lbs_skin_backward' in kernel_us:
            t_pair = (kernel_us['lbs_skin_forward'] + kernel_us['lbs_skin_backward']) * 1e-6
            by = bodies * 167028.0 + 19.35e6
            fl16 = 3.0 * 2.0 * bodies * V * (2 * 3 * 224.0 + 12 * 32.0)       # issued on the 16-bit matrix pipe (3 products)
            unit_keys = ['k_pose_fwd', 'lbs_skin_forward', 'keypoint_terms', 'lbs_skin_backward', 'pose_bwd_reduce']
            in_cycle = {k: kernel_us[k] for k in unit_keys if k in kernel_us}
            standalone = {k: 1e3 * float(np.mean(v)) for k, v in alone.items() if v}
            full = None
            if len(in_cycle) == len(unit_keys):
                t_in = sum(in_cycle.values()) * 1e-6
                full = {'what': 'every kernel that makes up SURVEY 8(d)\'s unit (pose features and joint transforms; skinning + '
                                'NDC projection + screen box + lowest vertex; the 17 key-points with projection, residual and '
                                'adjoint; the skinning adjoint; the pose adjoint + per-person reduction), summed',
                        'in_cycle_us': {k: round(v, 1) for k, v in in_cycle.items()},
                        'in_cycle_note': 'the three key-point launches run in the side branch UNDER the selection kernel, which holds '
                                         'every vector register of its CUs: their in-cycle time is waiting, not work',
                        'frac_in_cycle': round(by / t_in / 1e9 / PEAK_HBM_GBS, 4)}
                if len(standalone) == len(unit_keys):
                    t_al = sum(standalone.values()) * 1e-6
                    full['standalone_us'] = {k: round(v, 1) for k, v in standalone.items()}
                    full['frac_standalone'] = round(by / t_al / 1e9 / PEAK_HBM_GBS, 4)
            lbs = {'kernels': 'k_skin_fwd16p (projection epilogue, software-pipelined over a wave\'s tiles) | k_skinbwd16 (split-fp16 / split-bf16 contractions)',
                   'forward_us': round(kernel_us['lbs_skin_forward'], 1), 'backward_us': round(kernel_us['lbs_skin_backward'], 1),
                   'bound': 'hbm', 'algorithmic_bytes': by, 'achieved': round(by / t_pair / 1e9, 1), 'peak': PEAK_HBM_GBS,
                   'unit': 'GB/s', 'frac': round(by / t_pair / 1e9 / PEAK_HBM_GBS, 4),
                   'frac_note': 'skinning pair only (forward + backward of the two dense kernels); full_unit is SURVEY 8(d)\'s unit',
                   'full_unit': full,
                   'mfma_16bit': {'issued_tflops': round(fl16 / t_pair / 1e12, 1), 'peak': 2500.0, 'frac': round(fl16 / t_pair / 2.5e15, 4)},
                   'tolerance': 'vertices within 2.4e-7 m and gradients within 6e-6 (relative to the largest entry) of the '
                                'exact-fp32 MFMA kernels of round 1 (tools/time_lbs.py); fixtures: 1e-5 m / 2e-4',
                   'traffic': ({k: traffic.get(k) for k in ('k_skin_fwd16p', 'k_skin_fwd16', 'k_skinbwd16') if traffic.get(k)}
                               if counters_fresh(stamp, 'mh_lbs.hip') else None)}
        unit_frames = T_TOTAL if args.strong else T_LOCAL
        out = {
            'metric': 'optimizer iterations/sec (N humans x T frames)',
            'value': round(its if args.strong else its * world, 3),
            'unit': 'iterations/s (%d humans x %d frames per iteration unit)' % (N_PEOPLE, unit_frames), 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32 (the two dense LBS contractions as split-16-bit MFMA operands -- three products of (hi, lo) fp16 / bf16 terms, fp32 accumulate; everything else plain fp32)', 'data': 'synthetic', 'launch': 'eager' if args.eager else 'hipGraph replay',
            'rccl': rccl_info,
            'backend': ('RCCL (nccl)' if args.backend == 'nccl' else 'DRY RUN: gloo, collectives staged through the host%s -- not a '
                        'scaling measurement' % (', all ranks on one device' if args.one_device else '')) if world > 1 else None,
            'config': {'workload': ('BASELINE C5: %d humans x %d frames + 200 000-point scene cloud, %dx%d, batch 10, full nine-term '
                                    'loss stack + RMSprop, one-euro filters live' % (N_PEOPLE, T_TOTAL, IMG[0], IMG[1])) if args.strong else
                                   ('MuPoTs TS13-shape %d humans x %d frames, %dx%d, batch 10, full nine-term loss stack '
                                    '(2D joints, raster depth, soft silhouette, contact, foot sliding, priors, velocity, '
                                    'filtered vertices) + RMSprop; scene injected (ground plane), one-euro filters live '
                                    '(updated every 25 cycles inside the timed region)' % (N_PEOPLE, T_TOTAL, IMG[0], IMG[1])),
                       'humans': N_PEOPLE, 'frames': T_TOTAL, 'frames_per_gpu': frames_here, 'image': list(IMG),
                       'presteps': args.presteps,
                       'state': 'steady state: %d untimed optimisation cycles (+ %d warm-up) precede the timed ones -- bodies nearly at rest, '
                                'few face lists re-sorted; fit_250_ms_per_cycle / early_fit_ms_per_cycle are the drop-in call from the '
                                'initial variables' % (args.presteps, args.warmup),
                       'parallelism': 'one contiguous sequence, frames sharded x%d by the drop-in (mhmocap.optimizer under '
                                      'torch.distributed), one RCCL all-reduce on the betas/scale gradients + one-frame halos per cycle' % world},
            'timed_region_s': round(dt, 4),
            'filter_updates_in_timed_region': sum(1 for c in range(args.warmup, args.warmup + args.steps)
                                                  if c >= upd_at and (c - upd_at) % 25 == 0),
            'organic_scene': organic, 'roofline': roof, 'roofline_lbs_projection': lbs,
            'face_lists_rebuilt': {'bodies_seen': sort_seen, 'bodies_resorted': sort_rebuilt,
                                   'what': 'rasteriser preparation since the start of the run: a body\'s face lists are re-sorted only when '
                                           'one of its vertices has moved a pixel row since its last sort'},
            'kernel_us': {k: round(v, 1) for k, v in kernel_us.items()},
            'kernel_group_ms': {k: round(v, 4) for k, v in kern.items()},
            'loss_first_cycle': {k: float(v) for k, v in log[0].items()},
        }
        if not args.no_fit and world == 1 and not args.strong:
            out['fit_250'] = fit_block(struct, regs, tmp, device, K, seq)
            # what a caller of the drop-in sees, beside the steady-state headline: the whole fit(250) call of
            # predict.py:343 per cycle (staging, captures, scene updates, filter updates included) and its first 41 cycles
            out['fit_250_ms_per_cycle'] = round(1e3 * out['fit_250']['cycles_s'] / 250.0, 4)
            out['early_fit_ms_per_cycle'] = out['fit_250']['early_fit']['ms_per_cycle']
        if not args.no_cpu_baseline and world == 1 and not args.strong:
            ncores = os.cpu_count() or 1
            ncores = min(ncores, 16)
            torch.set_num_threads(ncores)
            os.environ['ORACLE_THREADS'] = str(ncores)
            if not args.cpu_frames:
                args.cpu_frames = T_LOCAL
            o, cpu_s = cpu_baseline(struct, regs, K, seq, pT0, args.cpu_frames, args.cpu_cycles)
            cpu_its = 1.0 / (cpu_s * (T_LOCAL / float(args.cpu_frames)))
            whole = args.cpu_frames == T_LOCAL
            out['cpu_baseline'] = {
                'value': round(cpu_its, 6), 'unit': 'iterations/s (4 humans x 200 frames per iteration unit)',
                'cores': ncores, 'kind': 'port',
                'sample': ('%d whole cycles of the CPU oracle (torch-CPU LBS/losses on %d threads + C face selection, one body per '
                           'thread) on ALL %d frames (x4 humans, 240x135, same nine-term stack): no extrapolation'
                           % (args.cpu_cycles, ncores, args.cpu_frames)) if whole else
                          ('%d cycles of the CPU oracle (torch-CPU LBS/losses on %d threads + C face selection, one body per '
                           'thread) on the first %d of the %d frames (x4 humans, 240x135, same nine-term stack); per-cycle time '
                           'scaled by %d/%d' % (args.cpu_cycles, ncores, args.cpu_frames, T_LOCAL, T_LOCAL, args.cpu_frames)),
                'sec_per_cycle_sample': round(cpu_s, 3),
                'reference_loop_in_build_container': 'the reference\'s own fit loop (PyTorch3D replaced by the oracle rasteriser through '
                                                     'the stubs) took 1.67 s per cycle on a 20-frame sample on 8 Xeon threads = '
                                                     '0.0598 it/s at 200 frames (BASELINE.md; it cannot run on the GPU box)'}
            out['speedup_vs_cpu_port'] = round(its / cpu_its, 1)
            out.update(mpjpe_block(struct, regs, tmp, device, K, seq, pT0, o, args))
            # SURVEY 8(d) states its tolerance after THIRTY cycles (<= 1e-3 m): the same comparison on a 20-frame sample, 30 cycles
            # on either side (outside every timed region; ~10 s of host time)
            import copy
            a30 = copy.copy(args)
            a30.cpu_frames, a30.cpu_cycles, a30.gpu_twice = 20, 30, True
            o30, _ = cpu_baseline(struct, regs, K, seq, pT0, a30.cpu_frames, a30.cpu_cycles)
            m30 = mpjpe_block(struct, regs, tmp, device, K, seq, pT0, o30, a30)
            out['mpjpe_mm_after_30_cycles'] = m30['mpjpe_mm_vs_cpu_oracle']
            out['mpjpe15_mupots_mm_after_30_cycles'] = m30['mpjpe15_mupots_mm_vs_cpu_oracle']
            out['mpjpe_after_30_cycles_detail'] = m30['mpjpe_detail']
            out['mpjpe_after_30_cycles_note'] = m30['mpjpe_note'] + (
                '; a FREE-RUNNING comparison: two runs of the same GPU code part by gpu_vs_gpu_* over the same cycles (float atomics in the '
                'gradient scatter, RMSprop\'s sign-like first steps); the strict statement -- every entry of every leaf, every cycle, from '
                'identical state -- is tests/test_fit_full_gpu.py::test_eight_cycles_step_by_step')
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def fit_block(struct, regs, tmp, device, K, seq):
    """Wall time of the drop-in call ``predict.py:343`` makes: ``opt.fit(dataloader, num_iter=250)`` on a fresh optimiser --
    staging of the dataloader's tensors, graph captures, 250 cycles, nine filter updates, 220 device-side scene updates
    (cycles 30..249) and the scene image, until the log is back on the host."""
    from mhhip import synthetic_seq
    W, H = IMG
    c = COEFS
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    opt = SMPLDepthSequenceOptimizer(
        image_size=IMG, num_frames=T_LOCAL, cam_K=K, device=device, smpl_model_parameters_path=tmp, smpl_data_struct=struct,
        proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
        reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
        reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=BATCH, shuffle=False)
    t0 = time.perf_counter()
    opt._stage_from_dataloader(dl)
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t0
    params0 = opt.engine.params.clone()
    # where the fit's time goes: an event in front of every cycle's launch (GPU time between consecutive cycle launches)
    e = opt.engine
    evs, orig = [], e.cycle_graphed

    host_t = []

    def traced(*a, **k):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        evs.append(ev)
        host_t.append(time.perf_counter())
        return orig(*a, **k)
    e.cycle_graphed = traced
    t0 = time.perf_counter()
    log = opt.fit(dl, num_iter=250)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    e.cycle_graphed = orig
    gaps = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)])
    phases = {'cycle_0_ms': round(float(gaps[0]), 3), 'cycles_1_29_ms': round(float(gaps[1:30].mean()), 4),
              'cycles_30_59_ms': round(float(gaps[30:60].mean()), 4), 'cycles_60_248_ms': round(float(gaps[60:].mean()), 4),
              'host_ms_before_first_cycle': round(1e3 * (host_t[0] - t0), 2), 'host_ms_first_to_last_launch': round(1e3 * (host_t[-1] - host_t[0]), 2),
              'host_ms_after_last_launch': round(1e3 * (t0 + t_fit - host_t[-1]), 2),
              'graphs_captured': len(getattr(e, '_graphs', {})),
              'lane_test': [{'replay_ms': [round(x, 3) for x in getattr(lt, 'ms', [])], 'busy_lanes': len(getattr(lt, 'busy', [])),
                             'candidates': len(lt.cand), 'clean': bool(getattr(lt, 'clean', False)), 'done': bool(lt.done)} for lt in getattr(e, '_lane_tests', {}).values()],
              'lane_pick': [{'cycle_ms': [round(x, 4) for x in (pk.ms or [])], 'best': pk.best} for pk in getattr(e, '_pickers', {}).values()],
              'what': 'GPU time between consecutive cycle launches of this fit (events): cycle 0 holds the eager run + the one capture'}
    ov = opt.get_optimized_variables()
    early = early_fit_block(opt, dl, params0)
    return {'early_fit': early, 'phases': phases, 'wall_s': round(t_stage + t_fit, 4), 'staging_s': round(t_stage, 4), 'cycles_s': round(t_fit, 4),
            'cycles_per_s_incl_everything': round(250.0 / (t_stage + t_fit), 1), 'init_optimized_variables_s': round(t_init, 4),
            'init_note': 'init_optimized_variables(num_iter=100) of a fresh optimiser in a process that has built this body model '
                         'before (what predict_mupots.py does per sequence): content hash of the model arrays (~3-8 ms; the '
                         'device tables are shared per process since round 3 -- building them is ~45 ms, once) + one LBS '
                         'forward + 100 Adam iterations on the (T,N,3) problem (~4 ms)',
            'what': 'opt.fit(dataloader, num_iter=250) as predict.py:343 calls it on C3: staging, graph captures, 250 cycles, '
                    '9 one-euro filter updates, 220 device scene updates, scene image; log read back',
            'final_loss_pose24j': float(log[-1]['loss_pose24j']), 'scene_points': int(opt.scene_pcd.shape[2]),
            'scene_img_shape': list(np.asarray(ov['scene_img']).shape)}


def early_fit_block(opt, dl, params0, cycles=41):
    """Cycles 0..40 of a fit from the initial variables, on an optimiser whose graphs exist (captured by the fit_250 call
    above): RMSprop's first, largest steps move the bodies by more than a pixel row per cycle, so most face lists are
    re-sorted and the projection epilogue's report filters miss more often than in the steady state the headline is
    timed in.  The variables are put back to where init_optimized_variables left them, the face lists are invalidated."""
    e = opt.engine
    e.params.copy_(params0)                              # (filters stay as the 250-cycle fit left them: same term set, same graphs)
    raster = e.raster_terms()
    raster.init_workspace()                              # sort tags cleared: every body sorts in cycle 0
    keep = opt.scene_update
    opt.scene_update = 'none'                            # cycles 0..40 never reach the scene update (cycle >= 30 with the device path
    torch.cuda.synchronize()                             # would only add its set-up to the wall time)
    seen0, reb0 = raster.sort_counters(e)
    t0 = time.perf_counter()
    opt.fit(dl, num_iter=cycles)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    seen1, reb1 = raster.sort_counters(e)
    opt.scene_update = keep
    return {'cycles': cycles, 'ms_per_cycle': round(1e3 * dt / cycles, 4), 'iterations_per_s': round(cycles / dt, 1),
            'bodies_resorted_share': round((reb1 - reb0) / max(1, seen1 - seen0), 4),
            'what': 'opt.fit(dataloader, num_iter=%d) from the initial variables on staged inputs and existing graphs: wall time / '
                    'cycles (includes the log read-back), share of (body, cycle) pairs whose face lists were re-sorted' % cycles}


def mpjpe_block(struct, regs, tmp, device, K, seq, pT0, o, args):
    """MPJPE (mm) between the GPU path and the CPU oracle after the same cycles on the CPU sample: the 17 AlphaPose
    key-points the loop optimises and the 15 MuPoTs joints the evaluator scores (evaluate.py:231-232, 254)."""
    from mhhip import synthetic, synthetic_seq
    from oracle import lbs_oracle as lo
    W, H = IMG
    nf = args.cpu_frames
    regs_m = dict(regs)
    opt2 = build_optimizer(struct, regs, tmp, nf, device, K)
    sl = slice(0, nf)
    opt2.init_optimized_variables(seq['pose2d'][sl], seq['poses_smpl'][sl], seq['betas_smpl'][sl], seq['valid_smpl'][sl], num_iter=0)
    opt2.engine.leaf('poses_T').copy_(torch.tensor(pT0[sl]).view(nf, N_PEOPLE, 3))
    mz = np.clip(np.max(pT0[sl][..., 0, 2], axis=1), 2, None)
    opt2.engine.leaf('zmax_lin').copy_(torch.tensor(2.0 * mz))
    sub = {k: v[sl] for k, v in seq.items() if isinstance(v, np.ndarray) and v.shape[:1] == (seq['pose2d'].shape[0],)}
    dl2 = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(sub), batch_size=BATCH, shuffle=False)
    opt2.scene_depth = ground_scene(K, W, H)
    opt2.update_scene_pointcloud(opt2.scene_depth, seq['backmasks'][sl].min(axis=0) > 0)
    params0 = opt2.engine.params.clone()
    opt2.fit(dl2, num_iter=args.cpu_cycles)
    ov, wv = opt2.get_optimized_variables(), o.optimized_variables()
    ov_again = None
    if getattr(args, 'gpu_twice', False):
        # the same GPU fit once more from the same start: what two runs of the SAME code part by (the gradient scatter sums with
        # float atomics, RMSprop's first steps are lr * sign(g) / sqrt(1 - alpha) whatever |g| is) -- the yardstick for the
        # distance to the CPU oracle's trajectory
        opt2.engine.params.copy_(params0)
        opt2.fit(dl2, num_iter=args.cpu_cycles)
        ov_again = opt2.get_optimized_variables()
    om = lo.BodyModel(struct, regs_m)

    def joints(v, key):
        B = nf * N_PEOPLE
        be = torch.tensor(v['betas_smpl']).expand(nf, N_PEOPLE, 10).reshape(B, 10)
        j = lo.smpl_forward(om, be, torch.tensor(v['poses_smpl']).view(B, 72))[key]
        s = torch.tensor(v['scale_factor']).view(1, N_PEOPLE, 1, 1)
        return s * j.view(nf, N_PEOPLE, -1, 3) + torch.tensor(v['poses_T'])
    extra = {}
    with torch.no_grad():
        e17 = (joints(ov, 'joints_alphapose') - joints(wv, 'joints_alphapose')).norm(dim=-1)
        d17 = e17.mean()
        d15 = (joints(ov, 'joints_mupots')[:, :, :15] - joints(wv, 'joints_mupots')[:, :, :15]).norm(dim=-1).mean()
        extra['median_mm'] = round(float(e17.median()) * 1000.0, 4)
        extra['p90_mm'] = round(float(e17.flatten().kthvalue(max(1, int(0.9 * e17.numel())))[0]) * 1000.0, 4)
        if ov_again is not None:
            g17 = (joints(ov, 'joints_alphapose') - joints(ov_again, 'joints_alphapose')).norm(dim=-1)
            extra['gpu_vs_gpu_mean_mm'] = round(float(g17.mean()) * 1000.0, 4)
            extra['gpu_vs_gpu_median_mm'] = round(float(g17.median()) * 1000.0, 4)
    return {'mpjpe_detail': extra,
            'mpjpe_mm_vs_cpu_oracle': round(float(d17) * 1000.0, 4), 'mpjpe15_mupots_mm_vs_cpu_oracle': round(float(d15) * 1000.0, 4),
            'mpjpe_note': '17 AlphaPose key-points / first 15 MuPoTs joints (evaluate.py:231-232), %d frames x 4 humans, after %d '
                          'identical cycles from identical inputs' % (nf, args.cpu_cycles)}


if __name__ == '__main__':
    main()
