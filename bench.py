#!/usr/bin/env python
"""Headline benchmark: optimizer iterations/sec of the scene-constrained SMPL optimisation loop
(one iteration = one ``fit`` cycle: all frames' residuals with the full nine-term loss stack,
hand-written backward, one RMSprop step; reference optimizer.py:375-587) on the MuPoTs-shaped
configuration BASELINE.json quotes the target on: 4 humans x 200 frames at 240x135, batch 10.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: frames are sharded over the ranks (weak scaling: every rank owns 200 frames of a
4 x 200*N sequence); one RCCL all-reduce on the shared shape/scale gradients per iteration plus
the one-frame halos (mhhip/sharded.py).  ``value`` counts iterations/sec in units of the 4x200
configuration (iterations/sec x frames/200), i.e. at N=1 it is plain iterations/sec.

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
        smpl_data_struct=struct, scene_update='none', proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'],
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


def load_pmc_valu(kernel):
    """Vector instructions per launch of `kernel` from the committed SQ counter pass (profiles/README.md)"""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_sq_counters.txt')
    if not os.path.exists(path):
        return None
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
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='launch every kernel from the host instead of replaying captured graphs')
    ap.add_argument('--cpu-frames', type=int, default=20)
    ap.add_argument('--cpu-cycles', type=int, default=3)
    # developer switches for the other BASELINE configs (parity-test cases, not bench lines): e.g. C5 =
    # --humans 8 --frames 500 --image 600x338 ; the JSON then names the workload it actually ran
    ap.add_argument('--humans', type=int, default=None)
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--image', type=str, default=None)
    args = ap.parse_args()
    global N_PEOPLE, T_LOCAL, IMG
    if args.humans:
        N_PEOPLE = args.humans
    if args.frames:
        T_LOCAL = args.frames
    if args.image:
        IMG = tuple(int(x) for x in args.image.split('x'))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    device = 'cuda:%d' % local_rank
    torch.cuda.set_device(local_rank)

    import tempfile
    from mhhip import build as mhbuild, synthetic, synthetic_seq, sharded
    from mhhip.raster import RasterTerms
    if rank == 0:
        mhbuild.build()
    if world > 1:
        dist.barrier()
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    K = synthetic.default_cam_K(IMG, 60.0)
    tmp = tempfile.mkdtemp()
    opt = build_optimizer(struct, regs, tmp, T_LOCAL, device, K)
    model = opt.SMPLPY.body_model
    seq = synthetic_seq.make_sequence(model, N_PEOPLE, T_LOCAL, IMG, 1003 + rank, cam_K=K)
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    pT0 = opt.poses_T.cpu().numpy().copy()
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=BATCH, shuffle=False)
    opt._stage_from_dataloader(dl)
    W, H = IMG
    opt.scene_depth = ground_scene(K, W, H)
    opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)     # contact term live
    e = opt.engine
    sh = sharded.ShardedSequence(e, rank * T_LOCAL, world * T_LOCAL)
    raster = RasterTerms(e)
    sh.update_filters()                                                                # filtered-vertex term live
    nstep = [0]

    def one_cycle(c, graphs, scene=False):
        if c % 25 == 0 and c > 0:
            sh.update_filters()
        # scene: the organic path of fit (cycle >= 30): scene rebuilt from the sequence every cycle, on its own stream, from
        # the leaves as they are before this cycle's step
        sh.cycle(c % e.log.shape[0], raster=raster, graphs=graphs, scene_update=scene)
        if scene:
            e.scene_device_swap()     # read by the next cycle's contact term (which waits on the update's event)
        sh.step(0.01 * 0.99 ** nstep[0])      # RMSprop, ExponentialLR(0.99) on the host as in the reference (optimizer.py:355-356)
        nstep[0] += 1

    use_graphs = not args.eager
    # bring the device to its steady state before the W warm-up steps: graph capture, lazy allocations, and enough
    # back-to-back work for the clocks to ramp (a fresh box that idled through the CPU-side set-up was once measured
    # at 0.57x for the first tens of milliseconds)
    for c_pre in range(300):            # a fixed count: every rank must issue the same collectives
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
    # the same cycles with the device-side scene aggregation of optimizer.py:578-584 running every cycle (reported
    # beside the headline, which uses the injected static scene BASELINE.json's C3 names)
    organic = None
    if world == 1:
        e.scene_device_setup(seq['backmasks'])
        for c in range(3):
            one_cycle(args.warmup + args.steps + c, use_graphs, scene=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for c in range(args.steps):
            one_cycle(args.warmup + args.steps + 3 + c, use_graphs, scene=True)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        organic = {'value': round(args.steps / dt1, 3), 'ms_per_step': round(1e3 * dt1 / args.steps, 4),
                   'what': 'per-cycle masked median over the 200 frames + bilateral/Sobel/erode/median-fill + un-projection '
                           '+ grid rebuild on a second stream, overlapped with the next cycle'}
        opt.scene_depth = ground_scene(K, W, H)
        opt.update_scene_pointcloud(opt.scene_depth, seq['backmasks'].min(axis=0) > 0)   # back to the static scene
    # per-kernel durations: HIP events cannot be read back from inside a replayed graph, so the same
    # launch sequence runs once more eagerly with events around the kernel groups (same kernels, same
    # stream, same data; only the launch mechanism differs)
    e.enable_timing(True)
    import ctypes
    from mhhip import _lib
    L = _lib.lib()
    L.mh_profile_enable(1)
    prof_names = ['k_raster_strip', 'k_raster_grads', 'k_skin_fwd', 'k_skin_bwd', 'k_contact_knn_grid', 'k_raster_sums']
    prof = {k: [] for k in prof_names}
    for c in range(args.steps):
        one_cycle(args.warmup + args.steps + c, False)
        torch.cuda.synchronize()
        for i, k in enumerate(prof_names):      # duration of this cycle's launch (HIP events on the launch stream)
            ms1 = ctypes.c_float(0)
            if L.mh_profile_read(i, ctypes.byref(ms1)) == 0:
                prof[k].append(float(ms1.value))
    L.mh_profile_enable(0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kern = e.timing_summary()
    e.enable_timing(False)
    kernel_us = {k: 1e3 * float(np.mean(v)) for k, v in prof.items() if v}
    win = raster.ws[:e.B * 16].view(torch.int32).view(e.B, 4).cpu().numpy()
    window_px = int((np.maximum(win[:, 2], 0).astype(np.int64) * np.maximum(win[:, 3], 0)).sum())
    log = sh.read_log(1)

    if rank == 0:
        ms = 1e3 * dt / args.steps
        its = args.steps / dt
        bodies = N_PEOPLE * T_LOCAL
        V, F = int(e.V), int(raster.faces.shape[0])
        dom = max(kernel_us, key=kernel_us.get) if kernel_us else None
        traffic = load_pmc_traffic()
        # dominant kernel: k_raster_strip.  Algorithmic bytes per launch: every body's projected
        # vertices (12 B x V) and row-sorted face list (4 B x F) in, the 40-byte key record of every window pixel out,
        # plus the face table once.  The kernel is VALU-issue bound, not HBM bound (profiles/: SQ_ACTIVE_INST_VALU is
        # ~80 % of the wave-resident cycles); the HBM fraction is reported because the contract asks for one.
        roof = None
        if 'k_raster_strip' in kernel_us:
            us = kernel_us['k_raster_strip']
            algo = bodies * (12.0 * V + 4.0 * F) + 40.0 * window_px + 12.0 * F
            gbs = algo / (us * 1e-6) / 1e9
            roof = {'kernel': 'k_raster_strip', 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS,
                    'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4), 'traffic': traffic.get('k_raster_strip'),
                    'launch_us': round(us, 1), 'algorithmic_bytes': algo, 'window_pixels': window_px,
                    'note': 'dominant kernel by time; integer/LDS-atomic z-buffer selection, VALU-issue bound '
                            '(see DESIGN.md section 4 and profiles/)', 'dominant_by_events': dom}
            nv = load_pmc_valu('k_raster_strip')
            if nv:   # the bound that does apply: wave64 vector instructions against 1024 SIMDs x 2.4 GHz / 4 cycles
                gi = nv / (us * 1e-6) / 1e9
                roof['valu_issue'] = {'wave_instructions': nv, 'achieved': round(gi, 1), 'peak': PEAK_VALU_GIPS,
                                      'unit': 'G wave-instr/s', 'frac': round(gi / PEAK_VALU_GIPS, 3)}
        # the GEMM-shaped kernels against the dense f32 MFMA peak: flops = 2 x 3 x 217 x V per body forward, plus the
        # 12 x 24 bone-transform adjoint per vertex backward
        roof_mfma = {}
        for k, fl in (('k_skin_fwd', bodies * V * 217.0 * 6.0), ('k_skin_bwd', bodies * V * (217.0 * 6.0 + 12.0 * 24.0 * 2.0))):
            if k in kernel_us:
                tf = fl / (kernel_us[k] * 1e-6) / 1e12
                # the same launch against HBM: per body two vertex arrays of 12 V bytes (forward: skinned + posed
                # vertices out; backward: vertex adjoints + posed vertices in) and the basis once (12 x 217 x V bytes)
                by = bodies * 2.0 * 12.0 * V + 12.0 * 217.0 * V
                gb = by / (kernel_us[k] * 1e-6) / 1e9
                roof_mfma[k] = {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                'frac': round(tf / PEAK_F32_MFMA_TFLOPS, 4), 'launch_us': round(kernel_us[k], 1),
                                'algorithmic_flops': fl, 'traffic': traffic.get(k),
                                'hbm': {'algorithmic_bytes': by, 'achieved': round(gb, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                        'frac': round(gb / PEAK_HBM_GBS, 4)}}
        out = {
            'metric': 'optimizer iterations/sec (N humans x T frames)', 'value': round(its * world, 3),
            'unit': 'iterations/s (%d humans x %d frames per iteration unit)' % (N_PEOPLE, T_LOCAL), 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'launch': 'eager' if args.eager else 'hipGraph replay',
            'config': {'workload': 'MuPoTs TS13-shape %d humans x %d frames, %dx%d, batch 10, full nine-term loss stack '
                                   '(2D joints, raster depth, soft silhouette, contact, foot sliding, priors, velocity, '
                                   'filtered vertices) + RMSprop; scene injected (ground plane), one-euro filters live'
                                   % (N_PEOPLE, T_LOCAL * world, IMG[0], IMG[1]),
                       'humans': N_PEOPLE, 'frames': T_LOCAL * world, 'frames_per_gpu': T_LOCAL, 'image': list(IMG),
                       'parallelism': 'frames sharded x%d, RCCL all-reduce on betas/scale grads' % world},
            'organic_scene': organic, 'roofline': roof, 'roofline_mfma': roof_mfma, 'kernel_us': {k: round(v, 1) for k, v in kernel_us.items()},
            'kernel_group_ms': {k: round(v, 4) for k, v in kern.items()},
            'loss_first_cycle': {k: float(v) for k, v in log[0].items()},
        }
        if not args.no_cpu_baseline and world == 1:
            ncores = os.cpu_count() or 1
            ncores = min(ncores, 16)
            torch.set_num_threads(ncores)
            o, cpu_s = cpu_baseline(struct, regs, K, seq, pT0, args.cpu_frames, args.cpu_cycles)
            cpu_its = 1.0 / (cpu_s * (T_LOCAL / float(args.cpu_frames)))
            out['cpu_baseline'] = {
                'value': round(cpu_its, 6), 'unit': 'iterations/s (4 humans x 200 frames per iteration unit)',
                'cores': ncores, 'kind': 'port',
                'sample': '%d cycles of the CPU oracle (torch-CPU LBS/losses + C face selection) on the first %d of the '
                          '200 frames (x4 humans, 240x135, same nine-term stack); per-cycle time scaled by 200/%d'
                          % (args.cpu_cycles, args.cpu_frames, args.cpu_frames),
                'sec_per_cycle_sample': round(cpu_s, 3)}
            out['speedup_vs_cpu_port'] = round(its / cpu_its, 1)
            # MPJPE (mm) between the GPU path and the CPU oracle after the same cycles on the sample
            from oracle import lbs_oracle as lo
            opt2 = build_optimizer(struct, regs, tmp, args.cpu_frames, device, K)
            sl = slice(0, args.cpu_frames)
            opt2.init_optimized_variables(seq['pose2d'][sl], seq['poses_smpl'][sl], seq['betas_smpl'][sl],
                                          seq['valid_smpl'][sl], num_iter=0)
            opt2.engine.leaf('poses_T').copy_(torch.tensor(pT0[sl]).view(args.cpu_frames, N_PEOPLE, 3))
            mz = np.clip(np.max(pT0[sl][..., 0, 2], axis=1), 2, None)
            opt2.engine.leaf('zmax_lin').copy_(torch.tensor(2.0 * mz))
            sub = {k: v[sl] for k, v in seq.items() if isinstance(v, np.ndarray) and v.shape[:1] == (T_LOCAL,)}
            dl2 = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(sub), batch_size=BATCH, shuffle=False)
            opt2.scene_depth = ground_scene(K, W, H)
            opt2.update_scene_pointcloud(opt2.scene_depth, seq['backmasks'][sl].min(axis=0) > 0)
            opt2.fit(dl2, num_iter=args.cpu_cycles)
            ov, wv = opt2.get_optimized_variables(), o.optimized_variables()
            om = lo.BodyModel(struct, regs)

            def joints(v):
                B = args.cpu_frames * N_PEOPLE
                be = torch.tensor(v['betas_smpl']).expand(args.cpu_frames, N_PEOPLE, 10).reshape(B, 10)
                j = lo.smpl_forward(om, be, torch.tensor(v['poses_smpl']).view(B, 72))['joints_alphapose']
                s = torch.tensor(v['scale_factor']).view(1, N_PEOPLE, 1, 1)
                return s * j.view(args.cpu_frames, N_PEOPLE, 17, 3) + torch.tensor(v['poses_T'])
            with torch.no_grad():
                d = (joints(ov) - joints(wv)).norm(dim=-1).mean()
            out['mpjpe_mm_vs_cpu_oracle'] = round(float(d) * 1000.0, 4)
            out['mpjpe_note'] = '17 key-points, %d frames x 4 humans, after %d identical cycles from identical inputs' % (
                args.cpu_frames, args.cpu_cycles)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
