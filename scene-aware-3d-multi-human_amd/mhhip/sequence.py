"""Device-resident state of one sequence (or one frame shard of it) and the per-cycle launch
sequence.  All arithmetic is in the HIP library; this file owns buffers, pointers and order.

Leaves live in ONE flat fp32 buffer so that a single RMSprop launch updates everything and the
shared tail (betas, xscale) is one contiguous slice for the per-cycle RCCL all-reduce:

    [ poses_T (T,N,3) | poses_smpl (T,N,72) | zmin_lin (T) | zmax_lin (T) | betas (N,10) | xscale (N) ]

Constant per-frame inputs are staged to HBM once (the reference re-uploads every batch every
cycle, optimizer.py:396-400): pose2d, reference poses, validity, depth maps, and the N float
instance masks of a frame packed into one 32-bit word per pixel (raw and twice-eroded).
"""
import ctypes

import numpy as np
import os
import torch

from . import _lib, engine, queues
from ._lib import check, ptr

LOG_KEYS = ['loss_pose24j', 'loss_depth', 'loss_silhouette', 'reg_ref_poses', 'reg_scale', 'reg_contact',
            'reg_foot_sliding', 'reg_vel', 'reg_filter_verts']
COEF_KEYS = ['proj2d', 'depth', 'silhouette', 'reg_velocity', 'reg_verts_filter', 'reg_poses', 'reg_scales',
             'reg_contact', 'reg_foot_sliding']


# The two extra streams of an engine (side branch of the cycle, scene update) are shared by all engines of a device:
# how streams land on the hardware queues decides whether the scene update really runs beside the cycle or in its
# queue, and a process that builds one optimiser per sequence would otherwise hand every new engine two
# fresh streams on whatever queues are next (measured: the same fit 0.25 s or 0.34 s depending on how many engines the
# process had built before).
_SHARED_STREAMS = {}


def _shared_stream(device, role):
    key = (torch.device(device).index or 0, role)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SHARED_STREAMS[key] = st
    return st


def _dev(a, device, dtype=torch.float32):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device).contiguous()


@_lib.on_own_device
class SequenceEngine(object):
    def __init__(self, model, image_size, num_frames, num_people, cam_K, cam_dist_coef=None, coefs=None,
                 joint_confidence_thr=0.5, eps=1e-3, batch_size=10, max_cycles=1024, joint_weights=None, joints_reg=None):
        self.m = model
        # the 17 key-points of the 2D term: (regressor, root joint or -1) -- reference smpl_sparse_joints_key
        # (optimizer.py:41, 75, 695-696).  The AlphaPose regressor's adjoint is fused into the LBS backward; any other
        # regressor goes through mh_joints_regress_backward into the vertex-gradient buffer.
        self.joints_reg = (engine.REG_ALPHAPOSE, -1) if joints_reg is None else (int(joints_reg[0]), int(joints_reg[1]))
        assert engine.NUM_REG_JOINTS[self.joints_reg[0]] == 17, 'the 2D term compares with 17 detected key-points'
        self.kp_fused = self.joints_reg[0] == engine.REG_ALPHAPOSE
        # per-key-point weights of the 2D term (reference optimizer.py:75-130, 259), mean 1; None = uniform
        self.joint_w = None if joint_weights is None else np.ascontiguousarray(np.asarray(joint_weights, np.float32).reshape(17))
        self.dev = model.device
        self.W, self.H = int(image_size[0]), int(image_size[1])
        self.T, self.N = int(num_frames), int(num_people)
        self.B = self.T * self.N
        self.V = model.V
        self.batch = int(batch_size)
        self.nbatches = (self.T + self.batch - 1) // self.batch
        self.K = np.ascontiguousarray(np.asarray(cam_K, np.float32).reshape(3, 3))
        self.Kd = None if cam_dist_coef is None else np.ascontiguousarray(np.asarray(cam_dist_coef, np.float32))
        c = {k: 1.0 for k in COEF_KEYS}
        c.update(coefs or {})
        self.c = c
        self.thr, self.eps = float(joint_confidence_thr), float(eps)
        T, N, B = self.T, self.N, self.B
        self.sizes = [B * 3, B * 72, T, T, N * 10, N]
        self.offs = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        n = int(self.offs[-1])
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
        self.params, self.sq, self.buf = z(n), z(n), z(n)
        self._grads_log = z(n + 16)      # gradients | staging row of the loss log: cleared by one fill per cycle
        self.grads = self._grads_log[:n]
        self.shared_lo = int(self.offs[4])          # betas | xscale: the all-reduced tail
        # round 6: a captured cycle whose caller steps right behind it (``fit``) leaves the per-person sums of the shape /
        # scale gradients to the update's launch (mh_rmsprop_step_person): one launch and one gap less on the chain
        self.defer_person = False       # set by the caller (mhmocap.optimizer.fit, bench.py); MHHIP_DEFER_PERSON=0 overrides
        self.freeze_xscale = False      # the deferred scale sums are dropped (optim_scale_factor off: the reference zeroes that gradient)
        self._person_now = False
        self._person_pending = False
        self.ws = model.workspace(B)
        self.ws2 = model.backward_workspace(B)
        self.kp_ws = torch.empty(max(1, _lib.lib().mh_keypoint_workspace_bytes(model.handle, B)), dtype=torch.uint8, device=self.dev)
        self.fv_ws = torch.empty(_lib.lib().mh_filtered_verts_workspace_bytes(T, N * self.V * 3), dtype=torch.uint8, device=self.dev)
        self.verts = z(B, self.V, 3)
        self.vposed = z(B, self.V, 3)
        self.gverts = None
        self.kp = z(B, 17, 3)
        self.gj = z(B, 17, 3)
        self.uv = z(B, 17, 2)
        self.loss2d = z(B)
        self.prior_body = z(B)
        self.loss3 = z(3)
        self.vel_loss = z(1)
        self.filt_loss = z(1)
        self.log = z(max_cycles, 16)
        self.tmp_log = self._grads_log[n:]
        self._log_pending = None         # row of self.log the staging row still has to go to (cycle_graphed -> step)
        self.scene_pts = None
        self.scene_grid = None
        self.scene_M = 0                 # capacity the grid workspace was sized for
        self._scene_dev = None           # device-side scene aggregation state (scene_device_setup)
        self._scene_pending = False
        self.verts_filt = None
        self.pT_filt = None
        self.has_images = False
        self.halo = None              # filled by the frame-sharded driver
        # Device-resident switches of the captured cycle: [0] the one-euro filters have run (optimizer.py:383-392: first at
        # cycle 50), [1] a device-built scene is live (:578-584: from cycle 31 on), [2] which of the two scene sets the contact
        # term reads.  The launches that depend on them are gated ON THE DEVICE, so ONE captured graph serves every phase of a
        # fit (round 5 captured five more graphs of 2.2-2.5 ms at cycles 30 / 31 / 32 / 50 / 51).
        self.phase = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self._phase_const = [torch.tensor([1, k], dtype=torch.int32, device=self.dev) for k in (0, 1)]
        self._phase_host = [0, 0, 0]
        self._phase_poke = None       # (live, set) the next step() writes in its own launch (scene_device_swap)
        self._filt_gate = False
        self.batch_frames = None      # batch table of the current cycle (set_batch_table); None = contiguous batches
        self.timing = None            # {name: [(start_event, end_event), ...]} when enabled by bench.py

    def enable_timing(self, on=True):
        self.timing = {} if on else None

    def _tic(self, name):
        if self.timing is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.timing.setdefault(name, []).append((a, b))
        a.record(torch.cuda.current_stream(self.dev))
        return b

    def _toc(self, ev):
        if ev is not None:
            ev.record(torch.cuda.current_stream(self.dev))

    def timing_summary(self):
        """mean milliseconds per launch group (call after a synchronize)."""
        out = {}
        for k, evs in (self.timing or {}).items():
            out[k] = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        return out

    def set_batch_size(self, batch_size):
        """Frames per batch of the reference's dataloader: fixes the in-batch foot-sliding pairs and
        the number of per-batch regulariser additions (optimizer.py:512-518, 531-539)."""
        self.batch = int(batch_size)
        self.nbatches = (self.T + self.batch - 1) // self.batch
        self.batch_frames = None

    def set_batch_table(self, table):
        """The batch composition of THIS cycle: table (nbatches, batch) int32, table[b, k] = frame at position k of
        batch b, -1 = empty (what a shuffling dataloader delivered, configs/predict_mupots.yml:14; the foot-sliding
        term pairs position k with position k-1, optimizer.py:512-518).  ``table`` may be a device tensor (a row of
        the per-fit table uploaded once) or a host array; None = contiguous batches.  The device buffer keeps its
        address: captured cycle graphs read it."""
        if table is None:
            self.batch_frames = None
            return
        n = self.nbatches * self.batch
        if getattr(self, 'batch_frames', None) is None or self.batch_frames.numel() != n:
            self.batch_frames = torch.full((n,), -1, dtype=torch.int32, device=self.dev)
        if isinstance(table, torch.Tensor):
            self.batch_frames.copy_(table.reshape(-1), non_blocking=True)
        else:
            self.batch_frames.copy_(torch.as_tensor(np.ascontiguousarray(table, dtype=np.int32).reshape(-1)))

    # -- leaves as views -------------------------------------------------------------------------
    def _view(self, buf, i, shape):
        return buf[int(self.offs[i]):int(self.offs[i + 1])].view(*shape)

    def leaf(self, name, buf=None):
        buf = self.params if buf is None else buf
        T, N = self.T, self.N
        i, shape = {'poses_T': (0, (T, N, 3)), 'poses_smpl': (1, (T, N, 72)), 'zmin_lin': (2, (T,)),
                    'zmax_lin': (3, (T,)), 'betas': (4, (N, 10)), 'xscale': (5, (N,))}[name]
        return self._view(buf, i, shape)

    def set_leaves(self, poses_T, poses_smpl, betas, zmin_lin, zmax_lin, xscale=None):
        self.leaf('poses_T').copy_(_dev(poses_T, self.dev).view(self.T, self.N, 3))
        self.leaf('poses_smpl').copy_(_dev(poses_smpl, self.dev).view(self.T, self.N, 72))
        self.leaf('betas').copy_(_dev(betas, self.dev).view(self.N, 10))
        self.leaf('zmin_lin').copy_(_dev(zmin_lin, self.dev).view(self.T))
        self.leaf('zmax_lin').copy_(_dev(zmax_lin, self.dev).view(self.T))
        if xscale is not None:
            self.leaf('xscale').copy_(_dev(xscale, self.dev).view(self.N))
        else:
            self.leaf('xscale').zero_()
        self.sq.zero_()
        self.buf.zero_()

    # -- staging ---------------------------------------------------------------------------------
    def stage(self, pose2d, poses_ref, valid, betas_ref, seg_mask=None, depths=None):
        T, N, H, W = self.T, self.N, self.H, self.W
        L = _lib.lib()
        st = _lib.stream_ptr(self.dev)
        # staging again (another dataloader handed to a later fit): captured cycles bake the addresses of the tensors
        # allocated below, and the device scene update holds transposed copies of the old depths / background masks
        if getattr(self, '_graphs', None):
            torch.cuda.current_stream(self.dev).synchronize()
            self._graphs = {}
            self._lane_tests = {}
            self._pickers = {}
        self._scene_dev = None
        self.pose2d = _dev(pose2d, self.dev).view(self.B, 17, 3)
        self.poses_ref = _dev(poses_ref, self.dev).view(self.B, 72)
        self.valid = _dev(valid, self.dev).view(self.B)
        self.betas_ref = _dev(betas_ref, self.dev).view(N, 10)
        self.area = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
        self.p2d_valid = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
        self.mask_valid = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
        if seg_mask is not None:
            self.bits = torch.zeros(T, H, W, dtype=torch.int32, device=self.dev)
            self.ebits = torch.zeros_like(self.bits)
            tmp = torch.zeros_like(self.bits)
            # upload in slabs: the float masks are 4*N bytes/pixel, the packed form 4 bytes/pixel
            slab = max(1, (256 << 20) // (N * H * W * 4))
            for s in range(0, T, slab):
                e = min(T, s + slab)
                seg = _dev(seg_mask[s:e], self.dev)
                check(L.mh_pack_masks(ptr(seg), e - s, N, H, W, ptr(self.bits[s:e]), ptr(self.area[s * N:e * N]), st))
                torch.cuda.current_stream(self.dev).synchronize()
            check(L.mh_erode_bits(ptr(self.bits), ptr(tmp), T, H, W, st))       # Erode2D(3) twice, optimizer.py:306-309
            check(L.mh_erode_bits(ptr(tmp), ptr(self.ebits), T, H, W, st))
            self.depths = _dev(depths, self.dev).view(T, H, W)
            self.front = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
            self.sil_apply = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
            self.sil_D = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
            self.sil_S = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
            self.sil_tag = torch.zeros(T, dtype=torch.int32, device=self.dev)    # 0: the frame's pixel counts do not exist yet
            self.sil_body = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
            self.depth_body = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
            self.has_images = True
        check(L.mh_stage_gates(ptr(self.pose2d), ptr(self.area), self.B, self.thr, 0.005 * H * W, ptr(self.p2d_valid),
                               ptr(self.mask_valid), st))
        self.low_idx = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.low_xyz = torch.zeros(self.B, 3, dtype=torch.float32, device=self.dev)
        self.dy = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
        self.batch_contact = torch.zeros(self.nbatches, dtype=torch.float32, device=self.dev)
        self.batch_foot = torch.zeros(self.nbatches, dtype=torch.float32, device=self.dev)

    def set_scene_points(self, pts):
        """pts (M,3) camera-space scene points or None (optimizer.py:605-616)."""
        self.scene_pts = None if pts is None else _dev(pts, self.dev).view(-1, 3)
        self._build_scene_grid()

    def _build_scene_grid(self):
        self.scene_grid = None
        if self.scene_pts is None or self.scene_pts.shape[0] == 0:
            return
        L = _lib.lib()
        M = self.scene_pts.shape[0]
        self.scene_grid = torch.empty(L.mh_scene_grid_bytes(M), dtype=torch.uint8, device=self.dev)
        self.scene_M = M
        self._scene_pending = False
        check(L.mh_scene_grid_build(ptr(self.scene_pts), M, ptr(self.scene_grid), _lib.stream_ptr(self.dev)))

    # -- scene aggregation on the device (optimizer.py:578-584 + fhsog.py:180-202 + utils.py:174-209) ---------
    def scene_device_setup(self, backmasks):
        """backmasks (T,H,W), non-zero = background.  Allocates the buffers of the per-cycle scene update.  The update
        of cycle c only depends on the depth-range leaves as they are at the START of cycle c and its result is first
        read by the contact term of cycle c+1, so it is launched on its own stream at the start of the cycle into the
        back set of (points, count, grid) and swapped in after the cycle: more than a full cycle of slack, nothing
        waits for the host."""
        L = _lib.lib()
        T, H, W = self.T, self.H, self.W
        P = H * W
        d = {}
        bm = np.asarray(backmasks)
        if bm.dtype != np.uint8 and bm.dtype != np.bool_:
            bm = bm != 0
        bm = np.ascontiguousarray(bm).view(np.uint8)          # (the staging pass of the optimiser already hands over bytes)
        d['back'] = (torch.as_tensor(bm).to(self.dev) != 0).to(torch.uint8)
        d['ws'] = torch.empty(L.mh_scene_workspace_bytes(T, H, W), dtype=torch.uint8, device=self.dev)
        if T <= 512:      # pixel-major copies of the constant inputs for the register form of the median
            d['depths_t'] = self.depths.view(T, P).t().contiguous()
            d['back_t'] = d['back'].view(T, P).t().contiguous()
        d['ma_depth'] = torch.zeros(H, W, device=self.dev)
        d['ma_mask'] = torch.zeros(H, W, device=self.dev)
        d['depth'] = torch.zeros(H, W, device=self.dev)
        # the update's own stream: one that does NOT drain through the launch stream's hardware queue (mhhip/queues.py)
        if queues.enabled():
            d['stream'] = queues.plan(self.dev).scene_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        else:
            d['stream'] = _shared_stream(self.dev, 'scene')
        d['ev_main'] = torch.cuda.Event()
        d['ev_snap'] = torch.cuda.Event()
        with torch.cuda.stream(d['stream']):      # first submission now: the stream gets its hardware queue before any
            d['ma_depth'].zero_()                 # cycle graph exists
        d['stream'].synchronize()
        d['sets'] = [dict(pts=torch.zeros(P, 3, device=self.dev), count=torch.zeros(1, dtype=torch.int32, device=self.dev),
                          grid=torch.empty(L.mh_scene_grid_bytes(P), dtype=torch.uint8, device=self.dev),
                          zsnap=torch.zeros(2 * T, device=self.dev), ev=torch.cuda.Event()) for _ in range(2)]
        d['next'] = 0                 # set the next update writes
        d['ready'] = None             # set written by the last update, not yet swapped in
        d['front'] = None
        self._scene_dev = d
        self._phase_poke = None       # new sets: none of them is live yet
        if self._phase_host[1]:
            self.phase[1:3].zero_()
            self._phase_host[1] = self._phase_host[2] = 0

    def _scene_move(self, stream):
        """the scene update continues on another stream (the updates share one workspace: the new stream first waits for
        whatever the old one still has to do)"""
        d = self._scene_dev
        if d is None or d['stream'] is stream:
            return
        stream.wait_stream(d['stream'])
        d['stream'] = stream

    def _scene_pick(self, key):
        """which stream the scene update runs on is settled by trying (mhhip/queues.py LanePicker): the candidates are the lanes
        the lane test of this graph found idle (all lanes off the launch stream's queue when the test ran beside a live scene
        update, or not at all) and the pooled stream of rounds 2-5"""
        if not hasattr(self, '_pickers'):
            self._pickers = {}
        pk = self._pickers.get(key)
        main = torch.cuda.current_stream(self.dev)
        if pk is None:
            if key not in getattr(self, '_graphs', {}) and ('full+scene',) + key not in getattr(self, '_graphs', {}):
                return                                   # (the capturing cycle: not a replay yet)
            pl = queues.plan(self.dev)
            lt = getattr(self, '_lane_tests', {}).get(('full+scene',) + key)
            if lt is not None and not lt.done and getattr(lt, 'recorded', False):
                # the host runs tens of cycles ahead of the device: the test's last replay has been launched long ago but
                # may not have run yet.  Waiting for it costs nothing -- the device has the cycles in between to work on
                lt.evs[-1][1].synchronize()
                lt.poll()
            clean = lt is not None and lt.done and getattr(lt, 'clean', False)
            raws = [l for l in lt.cand if l not in lt.busy] if clean else pl.free_lanes(main.cuda_stream)
            cands = [pl.view(r) for r in raws] + [_shared_stream(self.dev, 'scene')]
            pk = self._pickers[key] = queues.LanePicker(cands)
        st = pk.tick(main)
        if st is not None:
            self._scene_move(st)

    def scene_device_update(self):
        """Launch one scene update from the current depth-range leaves into the back set (own stream)."""
        self.scene_device_mark()
        self.scene_device_launch()

    def scene_device_mark(self):
        """The point of the main stream whose depth-range leaves the next ``scene_device_launch`` reads (an event, no
        work): ``cycle_graphed`` marks before it enqueues the cycle and launches the update after the first replay, so
        the first kernels of the cycle are not queued behind the update's dozen launches."""
        d = self._scene_dev
        d['ev_main'].record(torch.cuda.current_stream(self.dev))
        d['marked'] = True

    def scene_device_launch(self):
        d, L = self._scene_dev, _lib.lib()
        T, H, W = self.T, self.H, self.W
        if not d.get('marked'):
            self.scene_device_mark()
        d['marked'] = False
        k = d['next']
        s = d['sets'][k]
        side = d['stream']
        side.wait_event(d['ev_main'])
        st = side.cuda_stream
        with torch.cuda.stream(side):        # snapshot of the leaves on the update's own stream; step() waits for it
            zl, zh = self.leaf('zmin_lin').view(-1), self.leaf('zmax_lin').view(-1)
            if zh.data_ptr() == zl.data_ptr() + 4 * T and zl.numel() == T:
                # the two leaves are neighbours in the flat buffer: ONE copy (a launch less on this stream -- each one costs the
                # chain ~1.3 us)
                off = (zl.data_ptr() - self.params.data_ptr()) // 4
                s['zsnap'].copy_(self.params[off:off + 2 * T])
            else:
                s['zsnap'][:T].copy_(zl)
                s['zsnap'][T:].copy_(zh)
        d['ev_snap'].record(side)
        d['snap_pending'] = True
        if os.environ.get('MHHIP_SCENE_DELAY_US'):      # (probe: the update's wide kernels later in the cycle, DESIGN App. A)
            check(L.mh_stream_spin(st, float(os.environ['MHHIP_SCENE_DELAY_US'])))
        if 'depths_t' in d:
            check(L.mh_scene_median_t(T, H, W, ptr(d['depths_t']), ptr(d['back_t']), ptr(s['zsnap'][:T]), ptr(s['zsnap'][T:]),
                                      ptr(d['ma_depth']), ptr(d['ma_mask']), ptr(d['ws']), st))
        else:
            check(L.mh_scene_median(T, H, W, ptr(self.depths), ptr(d['back']), ptr(s['zsnap'][:T]), ptr(s['zsnap'][T:]),
                                    ptr(d['ma_depth']), ptr(d['ma_mask']), ptr(d['ws']), st))
        self._scene_finish(s, st)
        for _ in range(int(os.environ.get('MHHIP_SCENE_DUMMIES', '0') or 0)):      # (probe: what do kernel boundaries on this stream cost the chain?)
            check(L.mh_stream_spin(st, 0.02))
        s['ev'].record(side)
        d['ready'] = k
        d['next'] = 1 - k

    def _wait_scene_snapshot(self):
        """the leaves must not change before a pending scene update has copied them"""
        d = self._scene_dev
        if d is not None and d.get('snap_pending'):
            torch.cuda.current_stream(self.dev).wait_event(d['ev_snap'])
            d['snap_pending'] = False

    def _scene_finish(self, s, st):
        """median (d['ma_depth'], d['ma_mask']) -> post-processed depth map -> compacted cloud -> grid, into set s"""
        d, L = self._scene_dev, _lib.lib()
        H, W = self.H, self.W
        check(L.mh_scene_postprocess(H, W, ptr(d['ma_depth']), ptr(d['ma_mask']), 1, 7, ptr(d['depth']), ptr(d['ws']), st))
        check(L.mh_scene_points(H, W, self.K.ctypes.data_as(_lib.c_float_p), ptr(d['depth']), ptr(d['ma_mask']), ptr(s['pts']),
                                ptr(s['count']), st))
        check(L.mh_scene_grid_build_dev(ptr(s['pts']), ptr(s['count']), H * W, ptr(s['grid']), st))

    def scene_device_swap(self):
        """Make the last update the scene the contact term reads from now on (pointer swap; its consumer waits on the
        update's event)."""
        d = self._scene_dev
        if d['ready'] is None:
            return
        k = d['ready']
        s = d['sets'][k]
        d['front'], d['ready'] = s, None
        self._phase_poke = (1, k)                 # the device-side words follow in the next step()'s launch (or _flush_phase)
        self.scene_pts = s['pts']                 # capacity buffer; the live count is s['count'] (device)
        self.scene_grid = s['grid']
        self.scene_M = self.H * self.W
        self._scene_event = s['ev']
        self._scene_pending = True

    def _scene_sel(self):
        """The two device-built scene sets when the captured cycle can read the scene through the device-resident selector
        (no scene yet, or the live one IS one of the sets); None: a scene handed in from outside (``set_scene_points``,
        ``update_scene_pointcloud``) is read through its own pointers, as before."""
        d = self._scene_dev
        if d is None or os.environ.get('MHHIP_UNIFORM') == '0':
            return None
        if self.scene_pts is None or any(self.scene_pts is x['pts'] for x in d['sets']):
            return d['sets']
        return None

    def _flush_phase(self):
        """scene words scheduled by ``scene_device_swap`` that no ``step`` has taken along: one 8-byte copy on the stream"""
        if self._phase_poke is not None:
            (live, k), self._phase_poke = self._phase_poke, None
            self.phase[1:3].copy_(self._phase_const[k])
            self._phase_host[1], self._phase_host[2] = live, k

    def enable_filter_gate(self):
        """Allocate the filtered trajectories NOW (zeros) and run the filtered-vertex term behind the device-resident switch
        ``phase[0]`` from the first cycle on: the same captured launches before and after the first filter update
        (optimizer.py:383-392, 571-573).  No-op when filters already exist (a second fit keeps them, like the reference's
        ``self.verts_filtered``)."""
        if self.verts_filt is not None or self.pT_filt is not None:
            return self._filt_gate
        self.pT_filt = torch.zeros(self.T, self.N, 3, dtype=torch.float32, device=self.dev)
        self.verts_filt = torch.zeros(self.T, self.N, self.V, 3, dtype=torch.float32, device=self.dev)
        self._filt_gate = True
        return True

    def _filters_live(self):
        if self._filt_gate and not self._phase_host[0]:
            self.phase[0:1].fill_(1)
            self._phase_host[0] = 1

    # -- hooks of the frame-sharded driver (mhhip/sharded.py; the CPU stand-in of the tests implements the same) -----
    def main_stream(self):
        return torch.cuda.current_stream(self.dev)

    def stream_ctx(self, stream):
        return torch.cuda.stream(stream)

    def scene_median_rows(self, depths_t, back_t, zmin, zmax, med, msk, stream):
        """masked median over the frame axis of pixel-major rows (rows, frames); zmin = zmax = None: raw values"""
        rows, frames = depths_t.shape
        check(_lib.lib().mh_scene_median_t(frames, 1, rows, ptr(depths_t), ptr(back_t), ptr(zmin), ptr(zmax), ptr(med), ptr(msk),
                                           ptr(self._scene_dev['ws']), stream.cuda_stream))

    def scene_fill_plane(self, val, mask, ksize, stream):
        """looped ksize x ksize median fill of an integer-valued (H,W) plane, in place (optimizer.py:595-600)"""
        check(_lib.lib().mh_scene_fill(self.H, self.W, int(ksize), 1, ptr(val), ptr(mask), ptr(self._scene_dev['ws']), stream.cuda_stream))

    def scene_device_image(self, images):
        """images (T,H,W,3) uint8 -> (H,W,3) uint8 masked median over time of the background colour, holes filled with
        the 11x11 median like optimizer.py:595-600 (the colour median does not depend on the optimised variables: once
        per fit).  Returns (scene_img, scene_mask).  Up to 2048 frames: pixel-major register form of the median; longer
        sequences: the frame-major form (any T)."""
        d, L = self._scene_dev, _lib.lib()
        T, H, W = self.T, self.H, self.W
        P = H * W
        st = _lib.stream_ptr(self.dev)
        img = torch.as_tensor(np.ascontiguousarray(images)).to(self.dev)
        if T <= 2048:
            back_t = d['back_t'] if 'back_t' in d else d['back'].view(T, P).t().contiguous()
        out = torch.empty(H, W, 3, device=self.dev)
        mask = torch.empty(H, W, device=self.dev)
        for ch in range(3):
            val = torch.empty(H, W, device=self.dev)
            if T <= 2048:
                plane_t = img[..., ch].reshape(T, P).t().contiguous().float()
                check(L.mh_scene_median_t(T, H, W, ptr(plane_t), ptr(back_t), None, None, ptr(val), ptr(mask), ptr(d['ws']), st))
            else:
                plane = img[..., ch].float().contiguous()
                check(L.mh_scene_median(T, H, W, ptr(plane), ptr(d['back']), None, None, ptr(val), ptr(mask), ptr(d['ws']), st))
            val.floor_()                                        # .astype(np.uint8) of the reference
            m = mask.clone()
            check(L.mh_scene_fill(H, W, 11, 1, ptr(val), ptr(m), ptr(d['ws']), st))
            out[..., ch] = val
        return out.clamp_(0, 255).to(torch.uint8).cpu().numpy(), m.cpu().numpy()

    def scene_device_result(self):
        """(scene_depth (H,W), ma_mask (H,W) bool, points (M,3)) of the last device update, on the host."""
        d = self._scene_dev
        d['stream'].synchronize()
        s = d['front'] if d['ready'] is None else d['sets'][d['ready']]
        n = int(s['count'].item())
        return d['depth'].cpu().numpy(), d['ma_mask'].cpu().numpy() > 0.5, s['pts'][:n].clone()

    def scene_from_depth(self, depth, mask):
        d = _dev(depth, self.dev).view(self.H, self.W)
        pts = torch.empty(self.H * self.W, 3, dtype=torch.float32, device=self.dev)
        check(_lib.lib().mh_scene_unproject(ptr(d), self.H, self.W, self.K.ctypes.data_as(_lib.c_float_p), ptr(pts),
                                            _lib.stream_ptr(self.dev)))
        keep = _dev(np.asarray(mask, np.float32) if not isinstance(mask, torch.Tensor) else mask.float(), self.dev).view(-1) > 0.5
        self.scene_pts = pts[keep].contiguous()      # compaction only: plumbing
        self._build_scene_grid()
        return self.scene_pts

    # -- forward of all local frames ---------------------------------------------------------------
    def forward(self, regress=True, raster=None, clear=None):
        """LBS forward of all local frames.  raster (a ``RasterTerms`` of this engine): the "LBS + projection" form --
        the skinning epilogue also writes the NDC vertices, screen boxes, motion flags and lowest vertices that
        ``raster`` / the contact term would otherwise each take a pass over the vertices for (mh_lbs_forward_proj)."""
        m = self.m
        ev = self._tic('lbs_forward')
        self._projected_into = None       # one-shot tokens of what this forward's epilogue left behind: consumed by the
        self._lowkey_fresh = False        # rasteriser's preparation (mhhip/raster.py) and by the contact term (_scene_terms)
        if raster is not None and os.environ.get('MHHIP_NO_PROJ') != '1' and _lib.lib().mh_lbs_get_mode() != 0:
            t = raster.forward_targets()
            if clear is not None:
                t.clear, t.clear_n = clear.data_ptr(), clear.numel()
            check(_lib.lib().mh_lbs_forward_proj(m.handle, self.B, self.N, ptr(self.leaf('betas')), ptr(self.leaf('poses_smpl')),
                                                 ptr(self.leaf('xscale')), ptr(self.leaf('poses_T')), ptr(self.verts),
                                                 ptr(self.vposed), ctypes.byref(t), ptr(self.ws), _lib.stream_ptr(self.dev)))
            self._projected_into = raster
            self._lowkey_fresh = True
            self._lowkey = t.lowkey
        else:
            check(_lib.lib().mh_lbs_forward(m.handle, self.B, self.N, ptr(self.leaf('betas')), ptr(self.leaf('poses_smpl')),
                                            ptr(self.leaf('xscale')), ptr(self.leaf('poses_T')), ptr(self.verts),
                                            ptr(self.vposed), None, ptr(self.ws), _lib.stream_ptr(self.dev)))
        self._toc(ev)
        if regress:
            self._regress(_lib.stream_ptr(self.dev))

    def _regress(self, st):
        check(_lib.lib().mh_joints_regress(self.m.handle, self.joints_reg[0], self.B, ptr(self.verts),
                                           ptr(self.leaf('poses_T')), self.joints_reg[1], ptr(self.kp), st))

    def _halo_forward(self, h, st):
        """Frame-sharded run: the vertices of the neighbours' boundary frames (filtered-vertex term, optimizer.py:571-573)
        are skinned HERE from those frames' leaves -- 75 N floats per frame that arrive with the cycle's one all-reduce
        (mhhip/sharded.py) -- instead of being exchanged as 83 KB x N of vertices in the middle of the cycle: one small
        launch pair in the side branch, no communication between the forward and the term that reads its result, the whole
        cycle one graph.  Same kernel, same inputs as on the owning rank: the same bits.
        h['poses'] (Bh,72) / h['transl'] (Bh,3): [previous rank's last frame (N) | next rank's first frame (N)], a missing
        side left out."""
        Bh = int(h['poses'].shape[0])
        if getattr(self, '_halo_verts', None) is None or self._halo_verts.shape[0] != Bh:
            self._halo_verts = torch.empty(Bh, self.V, 3, dtype=torch.float32, device=self.dev)
            self._halo_ws = self.m.workspace(Bh)
        check(_lib.lib().mh_lbs_forward(self.m.handle, Bh, self.N, ptr(self.leaf('betas')), ptr(h['poses']), ptr(self.leaf('xscale')),
                                        ptr(h['transl']), ptr(self._halo_verts), None, None, ptr(self._halo_ws), st))
        k = 0
        h['v_prev'] = h['v_next'] = None
        if h.get('has_prev'):
            h['v_prev'] = self._halo_verts[:self.N]
            k = self.N
        if h.get('has_next'):
            h['v_next'] = self._halo_verts[k:k + self.N]

    def step_local(self, lr, alpha=0.5, momentum=0.9, eps=1e-8):
        """RMSprop on the per-frame leaves only (frame-sharded run: they need no other rank's gradients; the shared tail
        follows the all-reduce, ``step_shared``)"""
        self._wait_scene_snapshot()
        self._flush_person()
        self._flush_log()
        self._flush_phase()
        lo = self.shared_lo
        engine.rmsprop_step(self.params[:lo], self.grads[:lo], self.sq[:lo], self.buf[:lo], float(lr), alpha, momentum, eps)

    def step_shared(self, lr, alpha=0.5, momentum=0.9, eps=1e-8):
        lo = self.shared_lo
        engine.rmsprop_step(self.params[lo:], self.grads[lo:], self.sq[lo:], self.buf[lo:], float(lr), alpha, momentum, eps)

    def keypoint_terms(self, st):
        """The 2D term of the AlphaPose key-points from the pose features and joint transforms the forward left in its
        workspace: value, projection, residual AND the term's adjoint (one more chunk of the LBS backward's partial sums) in
        three small launches -- no pass over the vertices in either direction (csrc/mh_keypoints.hip)."""
        jwp = None if self.joint_w is None else self.joint_w.ctypes.data_as(_lib.c_float_p)
        Kp = self.K.ctypes.data_as(_lib.c_float_p)
        Kdp = None if self.Kd is None else self.Kd.ctypes.data_as(_lib.c_float_p)
        check(_lib.lib().mh_keypoint_terms(self.m.handle, self.B, ptr(self.leaf('poses_T')), Kp, Kdp, jwp, ptr(self.pose2d), self.thr,
                                           float(self.W), float(self.H), float(self.c['proj2d']), ptr(self.kp), ptr(self.uv),
                                           ptr(self.gj), ptr(self.loss2d), ptr(self.ws), ptr(self.ws2), ptr(self.kp_ws), st))
        self._kp_chunk = True

    def _side_stream(self):
        if not hasattr(self, '_side'):
            self._side = _shared_stream(self.dev, 'side')
        return self._side

    # -- one optimisation cycle (optimizer.py:375-575), gradients accumulated into self.grads -----
    def cycle(self, row, use_images=True, raster=None):
        # the same launch order as the captured form (cycle_graphed): the sums of a cycle are then added in the same order
        # either way, and eager and replayed fits stay bit-identical (deterministic mode) until something else differs
        self._flush_person()
        nj = raster is not None and use_images and self.has_images
        self.cycle_begin(join=not nj, raster=raster if (use_images and self.has_images) else None)
        self.cycle_finish(row, use_images, raster)

    def cycle_begin(self, join=True, raster=None):
        """zero the gradient buffer and run the LBS forward of all local frames (the frame-sharded
        driver exchanges boundary vertices between this and ``cycle_finish``).  join=False: the leaf-only terms of the side
        branch are NOT waited for here -- the caller's next join covers them (``cycle_graphed``: the one in front of the
        rasteriser's gradient half; its selection half reads nothing they write)."""
        L = _lib.lib()
        c = self.c
        T, N = self.T, self.N
        g = self.grads
        log = self.tmp_log
        self._flush_log()                # (never inside a capture: cycle_graphed has flushed before it replays)
        if not torch.cuda.is_current_stream_capturing():
            self._flush_phase()
        # the gradient buffer (+ the staging row of the log) is cleared by the forward's first kernel when the projection form
        # runs (mh_fwd_proj.clear): the captured cycle starts with k_pose_fwd instead of a fill and a 5-us gap
        # (only in the single-join form: with join=True the leaf-only terms start beside the forward and add into the buffer)
        clear_in_fwd = (not join) and raster is not None and os.environ.get('MHHIP_NO_PROJ') != '1' and os.environ.get('MHHIP_CLEAR_FILL') != '1' \
            and _lib.lib().mh_lbs_get_mode() != 0
        if not clear_in_fwd:
            self._grads_log.zero_()
        # the terms that only read the leaves (silhouette mask statistics, priors, velocity) run on the second stream
        # beside the MFMA-bound forward; their scalars land directly in the log row
        main = torch.cuda.current_stream(self.dev)
        side = self._side_stream()
        side.wait_stream(main)
        s2 = side.cuda_stream
        pT = self.leaf('poses_T')
        h = self.halo or {}
        # the chain's kernels are captured BEFORE the side branch's: a replayed graph keeps the branch whose nodes come first
        # on the queue it was launched on and moves the other one to a second queue, and every hop between queues costs
        # 10-14 us of idle time (rocprofv3 trace: the forward used to start 19 us into the cycle, now 9)
        self.forward(regress=False, raster=raster, clear=self._grads_log if clear_in_fwd else None)   # (the per-body pose-prior values are summed with the other log entries, _finish_a)
        def leaf_terms(part=3):
            # part: 1 = the silhouette mask statistics (read by the rasteriser's gradient half), 2 = priors + velocity, 3 = both
            if self.has_images and (part & 1):
                check(L.mh_sil_mask_stats_cached(ptr(self.bits), T, N, self.H, self.W, ptr(pT), ptr(self.p2d_valid),
                                                 ptr(self.mask_valid), ptr(self.front), ptr(self.sil_apply), ptr(self.sil_D),
                                                 ptr(self.sil_S), ptr(self.sil_tag), s2))
            if not (part & 2):
                return
            check(L.mh_prior_terms(T, N, self.nbatches, ptr(self.leaf('poses_smpl')), ptr(self.poses_ref), ptr(self.valid),
                                   ptr(self.leaf('betas')), ptr(self.betas_ref), ptr(self.leaf('xscale')),
                                   float(c['reg_poses']), float(c['reg_scales']), ptr(self.leaf('poses_smpl', g)),
                                   ptr(self.leaf('betas', g)), ptr(self.leaf('xscale', g)), ptr(self.prior_body), ptr(log[9:12]), s2))
            check(L.mh_velocity_term(T, N, ptr(pT), ptr(h.get('pT_prev')), ptr(h.get('pT_next')), float(c['reg_velocity']),
                                     ptr(self.leaf('poses_T', g)), ptr(log[7:8]), s2))

        # without a join here nothing needs these terms before the side branch ends: they are launched at its END then
        # (_finish_a), so that the vertex-dependent kernels start right behind the forward instead of 50 us later
        self._leaf_terms_later = None
        if not join:
            self._leaf_terms_later = leaf_terms
        else:
            leaf_terms()
        if join:
            main.wait_stream(side)

    def cycle_finish(self, row, use_images=True, raster=None, scene_ready=False):
        self._finish_a(use_images, raster, scene_ready=scene_ready)
        self._finish_b(row, use_images, raster)

    def _finish_a(self, use_images=True, raster=None, scene_ready=False):
        """Everything between the LBS forward and the scene-dependent part.  scene_ready: the caller has already made
        the stream wait for the device-side scene update whose cloud this cycle reads, so the contact chain runs in the
        side branch as with a static scene.  The vertex-gradient buffer starts as the
        filtered-vertex term (or zero); then the rasterised terms run on the main stream while the small terms that
        need the vertices (key-point regression + 2D joints and -- with a static scene -- contact / foot sliding) run
        on the second stream: in the captured graph they are a parallel branch that fills the tails of the raster
        kernels instead of a dozen serial launches of a few microseconds each."""
        L = _lib.lib()
        main = torch.cuda.current_stream(self.dev)
        st = main.cuda_stream
        c = self.c
        T, N, B = self.T, self.N, self.B
        Kp = self.K.ctypes.data_as(_lib.c_float_p)
        Kdp = None if self.Kd is None else self.Kd.ctypes.data_as(_lib.c_float_p)
        h = self.halo or {}
        # sel: the device-built scene sets, read through the device-resident selector (gated launches: they run -- and do
        # nothing -- while there is no scene yet, so that the captured sequence is the same before and after cycle 30)
        sel = self._scene_sel() if scene_ready else None
        scene = self.scene_pts is not None or sel is not None
        filt = self.verts_filt is not None and self.pT_filt is not None
        images = use_images and self.has_images
        need_gv = scene or filt or (images and raster is not None) or not self.kp_fused
        log = self.tmp_log
        gv = None
        if need_gv:
            if self.gverts is None:
                self.gverts = torch.empty_like(self.verts)
            gv = self.gverts
        self._gv_cur = gv
        # ---- side branch: vertex-gradient buffer initialised by the filtered-vertex term (or cleared), then the small
        # terms that need the vertices.  The main branch meanwhile runs the selection half of the rasteriser, which
        # does not touch the buffer, and waits for the initialisation before its gradient half.
        # Order of the side branch: the key-point regression (one pass over the vertices) runs beside the rasteriser's
        # projection kernel, the filtered-vertex initialisation (three streams of that size) after it, beside the face
        # sort, which hardly touches HBM -- the other way round the projection kernel took 70 us instead of 40.
        side = self._side_stream()
        side.wait_stream(main)
        s2 = side.cuda_stream

        def regress_project():
            if self.kp_fused and os.environ.get('MHHIP_NO_KPALG') != '1':
                self.keypoint_terms(s2)
                return
            jwp = None if self.joint_w is None else self.joint_w.ctypes.data_as(_lib.c_float_p)
            self._kp_chunk = False
            self._regress(s2)
            check(L.mh_project_joints_loss_w(B, ptr(self.kp), Kp, Kdp, jwp, ptr(self.pose2d), self.thr, 0, float(self.W),
                                             float(self.H), float(c['proj2d']), ptr(self.uv), ptr(self.gj), ptr(self.loss2d), s2))

        def side_branch():
            # order (same-box A/B with the chain on the launch queue): the vertex-gradient initialisation first -- 200 MB beside
            # the rasteriser's preparation, which is bound by latency -- then the key-point terms, the contact chain and, when
            # the caller left them to this branch, the leaf-only terms: everything behind the initialisation lands under
            # the selection kernel.  (With the side branch on the launch queue, round 2, the regression had to come first.)
            # round 4: the key-point launches (operands in L2) and the leaf-only terms go FIRST -- small kernels beside the
            # rasteriser's preparation -- and the 200 MB of the vertex-gradient initialisation behind them: beside it the
            # one-workgroup list kernel of the preparation took 38 us instead of 11 (0.719 -> 0.711 ms same-box)
            order_old = os.environ.get('MHHIP_SIDE_ORDER') == '0'
            fv_first = self.kp_fused and order_old
            later = getattr(self, '_leaf_terms_later', None)
            self._leaf_terms_later = None
            # (a second side branch under the gradient half and the skinning adjoint -- key-point launches, priors, velocity --
            # was tried in round 5, MHHIP_SIDE_SPLIT: +45 us; DESIGN App. A)
            if not fv_first:
                regress_project()
                if later is not None and not order_old:
                    later(3)
                    later = None
            with torch.cuda.stream(side):
                if need_gv and filt and h.get('poses') is not None:
                    self._halo_forward(h, s2)
                if need_gv:
                    if filt:
                        E = N * self.V * 3
                        ev = self._tic('filtered_verts')
                        if self._filt_gate:
                            check(L.mh_filtered_verts_term_init_gated(T, E, ptr(self.verts), ptr(self.verts_filt), ptr(h.get('v_prev')),
                                                                      ptr(h.get('vf_prev')), ptr(h.get('v_next')), ptr(h.get('vf_next')),
                                                                      float(c['reg_verts_filter']), ptr(gv), ptr(log[8:9]),
                                                                      self.phase.data_ptr(), ptr(self.fv_ws), s2))
                        else:
                            check(L.mh_filtered_verts_term_init(T, E, ptr(self.verts), ptr(self.verts_filt), ptr(h.get('v_prev')),
                                                                ptr(h.get('vf_prev')), ptr(h.get('v_next')), ptr(h.get('vf_next')),
                                                                float(c['reg_verts_filter']), ptr(gv), ptr(log[8:9]), ptr(self.fv_ws), s2))
                        self._toc(ev)
                    else:
                        gv.zero_()
                if not self.kp_fused:
                    # key-points from another regressor: their adjoint goes into the (just initialised) vertex gradients and
                    # the translation gradient here; the LBS backward then runs without key-point adjoints
                    check(L.mh_joints_regress_backward(self.m.handle, self.joints_reg[0], B, ptr(self.gj), self.joints_reg[1], ptr(gv),
                                                       ptr(self.leaf('poses_T', self.grads)), s2))
                if not hasattr(self, '_ev_gv'):
                    self._ev_gv = torch.cuda.Event()
                self._ev_gv.record(side)
            if fv_first:
                regress_project()
            self._scene_done = False
            sums = [(self.loss2d, log[0:1]), (self.prior_body, log[3:4])]
            if scene and (self._scene_dev is None or scene_ready):     # static scene, or its event already waited for
                self._scene_terms(s2, reduce=False, sel=sel)
                sums += [(self.batch_contact, log[5:6]), (self.batch_foot, log[6:7])]
                self._scene_done = True
            if later is not None:
                later()
            _lib.reduce_sum_multi(sums, s2)                            # the small log sums of the side branch: one launch
        # ---- main branch: rasterised depth / silhouette terms ----------------------------------------------------------
        joined = False
        # capture order: the chain's kernels BEFORE the side branch's -- the replayed graph keeps the branch whose nodes
        # come first on the queue it was launched on and puts the other one on a second queue; a hop between queues costs
        # the chain 10-14 us of idle time each way (rocprofv3 trace), the side branch has the slack for it
        main_first = images and raster is not None
        if not main_first:
            side_branch()
        if images:
            if raster is not None:
                ev = self._tic('raster_terms')
                # the rasteriser's work lists (a schedule) are rebuilt beside the LBS backward, for the NEXT cycle, when the
                # backward is the fused form that carries the closing job (one launch and 16 us less between the forward and
                # the selection kernel); the kernels find every tile with lists that are a cycle old
                ldef = 128 if (self.kp_fused and os.environ.get('MHHIP_NO_KPALG') != '1' and os.environ.get('MHHIP_NO_DEFER') != '1'
                               and os.environ.get('MHHIP_LISTS_ONCHAIN') != '1') else 0
                side_late = os.environ.get('MHHIP_SIDE_LATE', '1' if ldef else '0') == '1'
                if main_first and side_late:
                    # The side branch opens BEHIND the rasteriser's preparation, with the selection kernel already launched.
                    # With the work lists on the chain (two small launches between the forward and the selection) this order
                    # loses (r04: 0.750 ms against 0.735: the side branch starts 20 us later and ends under the selection
                    # kernel's tail).  With the lists deferred it is the one that wins: the selection kernel must get its
                    # first 512 workgroups -- the longest tiles -- onto the CUs before the side branch's kernels arrive; forked
                    # behind the forward they arrive together with it, take their slots at full speed (the key-point kernels
                    # ran in 9 us instead of 65) and the selection kernel pays 34 us for it (0.700 ms against 0.679 with the lists
                    # on the chain, 0.676 this way: same box).
                    raster(self, gv, log, phases=4 | ldef)
                    side.wait_stream(main)
                    raster(self, gv, log, phases=8)
                    side_branch()
                else:
                    raster(self, gv, log, phases=(4 | 8 | ldef) if ldef else 1)
                    if main_first:
                        side_branch()
                # the whole side branch ends long before the selection does: ONE join here instead of a wait for the
                # buffer initialisation here and a second join in front of the backward (every cross-stream edge of
                # the replayed graph costs several us of idle time on the chain, even when its event has long been
                # signalled: +0.5 % same-box)
                main.wait_stream(side)
                joined = True
                # the closing kernel's job rides in the LBS backward's pose kernel when that is the fused form
                # (_finish_b): one launch and one dependent kernel less on the chain
                defer = bool(getattr(self, '_kp_chunk', False)) and os.environ.get('MHHIP_NO_DEFER') != '1'
                self._raster_fin = raster(self, gv, log, phases=2 | (ldef if defer else 0), defer=defer)
                self._toc(ev)
            else:
                # no rasteriser: alpha = 0, zbuf empty -> the mask-only silhouette term (tests only)
                self.sil_body.copy_(self.sil_apply * self.sil_S / (self.sil_D + 1.0))
                check(L.mh_reduce_sum(ptr(self.sil_body), B, 1.0, ptr(log[2:3]), st))
        if not joined:
            main.wait_stream(side)

    def _scene_terms(self, st, reduce=True, sel=None):
        """contact + in-batch foot sliding (optimizer.py:485-518) on stream st; gradients by atomics / disjoint writes.
        sel: the two device-built scene sets (``_scene_sel``) -- the launches read the live one through ``phase[1:3]`` and do
        nothing while there is none."""
        L = _lib.lib()
        c = self.c
        T, N, B = self.T, self.N, self.B
        gpT = self.leaf('poses_T', self.grads)
        gv, log = self._gv_cur, self.tmp_log
        fresh, self._lowkey_fresh = getattr(self, '_lowkey_fresh', False), False
        if sel is not None:
            words = self.phase.data_ptr() + 4                       # [scene live, which set]
            if not fresh:
                check(L.mh_lowest_vertex(ptr(self.verts), B, self.V, ptr(self.low_idx), ptr(self.low_xyz), st))
            check(L.mh_contact_knn_grid_sel(ptr(sel[0]['grid']), ptr(sel[1]['grid']), self.H * self.W, words,
                                            ptr(self.verts) if fresh else None, self.V, self._lowkey if fresh else None, B, 32,
                                            ptr(self.low_idx), ptr(self.low_xyz), ptr(self.dy), st))
            check(L.mh_contact_foot_terms_gated(T, N, self.V, self.batch, self.nbatches, ptr(getattr(self, 'batch_frames', None)),
                                                ptr(self.verts), ptr(self.low_idx), ptr(self.low_xyz), ptr(self.dy),
                                                float(c['reg_contact']), float(c['reg_foot_sliding']), ptr(gpT), ptr(gv),
                                                ptr(self.batch_contact), ptr(self.batch_foot), words, st))
            if reduce:
                _lib.reduce_sum_multi([(self.batch_contact, log[5:6]), (self.batch_foot, log[6:7])], st)
            return
        if fresh:                                                   # the forward's epilogue has reported the lowest vertices
            check(L.mh_contact_knn_grid_key(ptr(self.scene_grid), self.scene_M, ptr(self.verts), self.V, self._lowkey, B, 32,
                                            ptr(self.low_idx), ptr(self.low_xyz), ptr(self.dy), st))
        else:
            check(L.mh_lowest_vertex(ptr(self.verts), B, self.V, ptr(self.low_idx), ptr(self.low_xyz), st))
            check(L.mh_contact_knn_grid(ptr(self.scene_grid), self.scene_M, ptr(self.low_xyz), B, 32, ptr(self.dy), st))
        if getattr(self, 'batch_frames', None) is not None:
            check(L.mh_contact_foot_terms_idx(T, N, self.V, self.batch, self.nbatches, ptr(self.batch_frames), ptr(self.verts),
                                              ptr(self.low_idx), ptr(self.low_xyz), ptr(self.dy), float(c['reg_contact']),
                                              float(c['reg_foot_sliding']), ptr(gpT), ptr(gv), ptr(self.batch_contact),
                                              ptr(self.batch_foot), st))
        else:
            check(L.mh_contact_foot_terms(T, N, self.V, self.batch, ptr(self.verts), ptr(self.low_idx), ptr(self.low_xyz),
                                          ptr(self.dy), float(c['reg_contact']), float(c['reg_foot_sliding']), ptr(gpT),
                                          ptr(gv), ptr(self.batch_contact), ptr(self.batch_foot), st))
        if reduce:
            _lib.reduce_sum_multi([(self.batch_contact, log[5:6]), (self.batch_foot, log[6:7])], st)

    def _finish_b(self, row, use_images=True, raster=None):
        """scene terms when they could not run in the side branch (eager launches with the cloud rebuilt on the device
        every cycle: they wait for its event here), LBS backward, log row"""
        L = _lib.lib()
        st = _lib.stream_ptr(self.dev)
        T, N, B = self.T, self.N, self.B
        g = self.grads
        gpT, gposes = self.leaf('poses_T', g), self.leaf('poses_smpl', g)
        gbetas, gxs = self.leaf('betas', g), self.leaf('xscale', g)
        pT = self.leaf('poses_T')
        filt = self.verts_filt is not None and self.pT_filt is not None
        gv, log = self._gv_cur, self.tmp_log
        if self.scene_pts is not None and not self._scene_done:
            ev = self._tic('scene_terms')
            if self._scene_pending:              # the scene of the previous cycle is built on its own stream
                torch.cuda.current_stream(self.dev).wait_event(self._scene_event)
                self._scene_pending = False
            self._scene_terms(st)
            self._toc(ev)
        ev = self._tic('lbs_backward')
        fin, self._raster_fin = getattr(self, '_raster_fin', None), None
        if getattr(self, '_kp_chunk', False):
            pb, px = (None, None) if self._person_now else (ptr(gbetas), ptr(gxs))      # (deferred: summed by the update's launch)
            args = (self.m.handle, B, N, ptr(self.leaf('betas')), ptr(self.leaf('poses_smpl')), ptr(self.vposed),
                    ptr(gv), ptr(gposes), ptr(gpT), pb, px, ptr(self.ws), ptr(self.ws2),
                    ctypes.byref(fin) if fin is not None else None, st)
            check(L.mh_lbs_backward_kp_fin(*args))
        else:
            check(L.mh_lbs_backward(self.m.handle, B, N, ptr(self.leaf('betas')), ptr(self.leaf('poses_smpl')),
                                    ptr(self.leaf('xscale')), ptr(pT), ptr(self.vposed), ptr(gv), ptr(self.gj) if self.kp_fused else None, ptr(gposes),
                                    ptr(gpT), None if self._person_now else ptr(gbetas), None if self._person_now else ptr(gxs),
                                    ptr(self.ws), ptr(self.ws2), st))
        self._toc(ev)
        if row is not None:
            self.log[row].copy_(log)

    def step(self, lr, alpha=0.5, momentum=0.9, eps=1e-8):
        self._wait_scene_snapshot()
        poke, self._phase_poke = self._phase_poke, None
        person = None
        if self._person_pending:
            self._person_pending = False
            person = self._person_sums()
        if self._log_pending is not None or poke is not None or person is not None:
            # the log entries of the graph that was just replayed travel to their row in the update's launch -- and so do
            # the scene words of the next cycle (which of the two device-built scene sets is live)
            row, self._log_pending = self._log_pending, None
            engine.rmsprop_step_log(self.params, self.grads, self.sq, self.buf, float(lr), self.tmp_log if row is not None else None,
                                    self.log[row] if row is not None else None, alpha, momentum, eps,
                                    poke_dst=self.phase[1:3] if poke is not None else None, poke=poke, person=person)
            if poke is not None:
                self._phase_host[1], self._phase_host[2] = poke
        else:
            engine.rmsprop_step(self.params, self.grads, self.sq, self.buf, float(lr), alpha, momentum, eps)

    def _person_sums(self):
        """where the last backward left the per-body shape / scale gradients (``_lib.PersonSums``)"""
        gb, gx = ctypes.c_void_p(), ctypes.c_void_p()
        check(_lib.lib().mh_lbs_backward_person_partials(self.m.handle, self.B, ptr(self.ws2), ctypes.byref(gb), ctypes.byref(gx)))
        return _lib.PersonSums(gb.value, gx.value, self.B, self.N, 10, int(self.offs[4]), -1 if self.freeze_xscale else int(self.offs[5]))

    def _flush_person(self):
        """the per-person sums a deferring cycle left undone, when no ``step`` has taken them along (somebody else is about to
        read the gradients): the launch the cycle skipped, now"""
        if self._person_pending:
            self._person_pending = False
            g = self.grads
            check(_lib.lib().mh_lbs_person_reduce(self.m.handle, self.B, self.N, ptr(self.ws2), ptr(self.leaf('betas', g)),
                                                  ptr(self.leaf('xscale', g)), _lib.stream_ptr(self.dev)))

    def _flush_log(self):
        """the staging row of the last replayed cycle into its row of the log, when no ``step`` has taken it along"""
        if self._log_pending is not None:
            row, self._log_pending = self._log_pending, None
            self.log[row].copy_(self.tmp_log)

    # -- hipGraph replay of a cycle -------------------------------------------------------------------
    # A cycle is ~45 launches of 5-600 us kernels; replaying it as a captured graph removes the host
    # launch gaps.  Everything a captured launch reads must keep its address and no host scalar may
    # change between replays: the log row goes through a staging row, the learning rate lives on the
    # device.  Capture is per configuration (raster / scene / filters on or off).
    def _graph_key(self, raster):
        # device addresses and the by-value sizes a capture bakes in (an address recycled by the allocator with the same
        # sizes replays correctly: the kernels read whatever is there now)
        scene = None
        sel = self._scene_sel()
        if sel is not None:               # the scene is read through the device-resident selector: one key for the whole fit
            scene = ('sel', sel[0]['grid'].data_ptr(), sel[1]['grid'].data_ptr())
        elif self.scene_pts is not None:
            scene = (self.scene_pts.data_ptr(), self.scene_grid.data_ptr(), self.scene_M)
        rast = None if raster is None else (raster.ws.data_ptr(), raster.faces.data_ptr())
        bt = None if getattr(self, 'batch_frames', None) is None else self.batch_frames.data_ptr()
        # process-wide switches a capture bakes in: the gradient-scatter kernel (mh_raster_set_deterministic picks it at capture
        # time), the sort margin (a kernel argument by value) and the LBS arithmetic mode
        L = _lib.lib()
        glob = (L.mh_raster_get_deterministic(), L.mh_raster_get_sort_margin(), L.mh_lbs_get_mode(), L.mh_raster_get_path(),
                L.mh_raster_get_winners(), float(L.mh_raster_get_sort_defer()), L.mh_lbs_get_forward_pipeline()) if raster is not None else None
        hk = None if self.halo is None else (bool(self.halo.get('has_prev')), bool(self.halo.get('has_next')), self.halo.get('poses') is not None)
        return (rast, scene, self.verts_filt is not None and self.pT_filt is not None, self._filt_gate, hk, bt, glob)

    def raster_terms(self, znear=1.0, zfar=100.0):
        """The engine's rasteriser binding (workspace + face table), created once and kept alive with the engine:
        captured cycle graphs hold its device addresses, so it must outlive every ``fit`` call."""
        if getattr(self, '_raster', None) is None:
            from . import raster as _raster
            self._raster = _raster.RasterTerms(self, znear, zfar)
        return self._raster

    def replay(self, key, fn, wait_scene=True):
        """Run ``fn`` (a fixed launch sequence on static buffers) through a captured graph; the first
        call runs it eagerly (lazy allocations, one-time attribute calls) and captures it."""
        if wait_scene and self._scene_pending:   # cross-stream dependency stays outside the captured sequence
            main = torch.cuda.current_stream(self.dev)
            if os.environ.get('MHHIP_GATE_PROBE') == '1':      # (probe: how long does the cycle wait for the previous scene update?)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                main.wait_event(self._scene_event)
                e1.record(main)
                self.__dict__.setdefault('_gate_probe', []).append((e0, e1))
            else:
                main.wait_event(self._scene_event)
            self._scene_pending = False
        if not hasattr(self, '_graphs'):
            self._graphs = {}
        if not hasattr(self, '_lane_tests'):
            self._lane_tests = {}
        g = self._graphs.get(key)
        if g is None:
            fn()
            torch.cuda.current_stream(self.dev).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            self._graphs[key] = g
        else:
            # a new graph that runs beside the device-side scene update: its first replays also find out which hardware queues
            # it keeps busy, and the scene update moves to one it does not (mhhip/queues.py)
            want = self._scene_dev is not None and queues.enabled() and str(key[0]).startswith('full') and os.environ.get('MHHIP_LANE_TEST') != '0'      # (the cycle's graphs only)
            lt = self._lane_tests.get(key) if want else None
            if lt is None and want and key not in self._lane_tests:
                lt = self._lane_tests[key] = queues.LaneTest(queues.plan(self.dev), torch.cuda.current_stream(self.dev).cuda_stream)
                lt.clean = True
            if lt is not None and not lt.done and not getattr(lt, 'recorded', False):
                if self._scene_dev.get('snap_pending') or self._scene_pending:
                    lt.clean = False              # a scene update in flight: a spun lane may stall IT, not the graph
                main = torch.cuda.current_stream(self.dev)
                lt.before(main)
                g.replay()
                lt.after(main)
            else:
                if lt is not None and not lt.done and lt.poll() and lt.choice is not None:
                    self._scene_move(queues.plan(self.dev).view(lt.choice))
                g.replay()

    def cycle_graphed(self, row, raster=None, scene_update=False):
        """``cycle`` through a captured graph (single-process form; the sharded driver replays
        ``cycle_begin`` / ``cycle_finish`` separately around its exchanges).  scene_update: also launch this cycle's
        device-side scene update (``scene_device_update``), after the first replay has been enqueued."""
        self._flush_person()
        self._person_now = bool(self.defer_person) and os.environ.get('MHHIP_DEFER_PERSON') != '0'
        key = self._graph_key(raster) + (self._person_now,)
        self._flush_log()
        self._flush_phase()
        if scene_update and queues.enabled() and os.environ.get('MHHIP_LANE_PICK') != '0':
            self._scene_pick(key)
        if scene_update:
            self.scene_device_mark()
        if self._scene_dev is not None:
            # The scene cloud is rebuilt every cycle on its own stream; the update this cycle's contact term reads was
            # launched a whole cycle ago (and takes less than half of one), so the wait for its event is hoisted to the
            # start of the cycle (replay() issues it in front of the graph: it never stalls in practice) and the cycle
            # stays ONE graph with the contact chain in the side branch, hidden under the selection kernel as with a
            # static scene.  (Until late in round 2 the cycle was split in two graphs at the contact chain: a graph
            # boundary of ~30 us and the chain on the critical path.)
            def body_org():
                nj = raster is not None and self.has_images
                self.cycle_begin(join=not nj, raster=raster if self.has_images else None)
                self.cycle_finish(None, raster=raster, scene_ready=True)
            self.replay(('full+scene',) + key, body_org)
            if scene_update:
                self.scene_device_launch()
        else:
            def body():
                # nothing of the rasteriser's selection half reads what the leaf-only terms of the side branch write: with
                # gradients asked for, the one join in front of the gradient half covers them
                nj = raster is not None and self.has_images
                self.cycle_begin(join=not nj, raster=raster if self.has_images else None)
                self.cycle_finish(None, raster=raster)
            self.replay(('full',) + key, body)
            if scene_update:
                self.scene_device_launch()
        self._log_pending = row            # copied by the next step() (same launch) or by whoever reads the log first
        self._person_pending, self._person_now = self._person_now, False

    def step_dev(self, alpha=0.5, momentum=0.9, eps=1e-8, gamma=0.99, lr0=0.01):
        if not hasattr(self, 'lr_dev'):
            self.lr_dev = torch.full((1,), lr0, dtype=torch.float32, device=self.dev)
        self._flush_person()
        self._flush_log()
        self._flush_phase()
        self._wait_scene_snapshot()
        check(_lib.lib().mh_rmsprop_step_dev(ptr(self.params), ptr(self.grads), ptr(self.sq), ptr(self.buf),
                                             self.params.numel(), ptr(self.lr_dev), gamma, alpha, momentum, eps,
                                             _lib.stream_ptr(self.dev)))

    # -- filters (optimizer.py:383-392) -------------------------------------------------------------
    def set_filters(self, pT_filt, verts_filt):
        """Install new filtered trajectories.  The buffers keep their addresses from the first call on: captured
        cycle graphs read them."""
        verts_filt = verts_filt.view(self.T, self.N, self.V, 3)
        if self.pT_filt is None or self.pT_filt.shape != pT_filt.shape:
            self.pT_filt = pT_filt.clone()
        else:
            self.pT_filt.copy_(pT_filt)
        if self.verts_filt is None or self.verts_filt.shape != verts_filt.shape:
            self.verts_filt = verts_filt.clone()
        else:
            self.verts_filt.copy_(verts_filt)
        self._filters_live()

    def update_filters(self, c1=0.01, b1=0.02, c2=0.001, b2=0.5):
        """optimizer.py:383-392: the translations and the vertices of the current leaves through the one-euro filters.  From
        the second call on the scans write straight into the buffers the captured cycle graphs read (the first call creates
        them): no second pass over the 66 MB of vertices; the forward runs without the key-point regression."""
        pT = self.leaf('poses_T')
        if self.pT_filt is not None and self.verts_filt is not None and self.pT_filt.numel() == pT.numel():
            engine.one_euro_scan(pT, c1, b1, out=self.pT_filt)
            self.forward(regress=False)
            engine.one_euro_scan(self.verts.view(self.T, -1), c2, b2, out=self.verts_filt)
            self._filters_live()
            return
        pf = engine.one_euro_scan(pT, c1, b1)
        self.forward(regress=False)
        self.set_filters(pf, engine.one_euro_scan(self.verts.view(self.T, -1), c2, b2))

    def one_euro_shard(self, x, min_cutoff, beta, first_frame, state_in=None):
        return engine.one_euro_scan_shard(x, min_cutoff, beta, first_frame, state_in)

    # -- logs back on the host (one D2H per fit) ---------------------------------------------------
    def read_log(self, rows, nbatches_total=None):
        self._flush_log()
        raw = self.log[:rows].cpu().numpy().astype(np.float64)
        nb = float(nbatches_total or self.nbatches)
        out = []
        for r in raw:
            d = {'loss_pose24j': r[0] / nb, 'loss_depth': r[1] / nb, 'loss_silhouette': r[2] / nb,
                 'reg_ref_poses': (r[3] + r[9]) / nb, 'reg_scale': r[10] + r[11], 'reg_contact': r[5] / nb,
                 'reg_foot_sliding': r[6] / nb, 'reg_vel': r[7], 'reg_filter_verts': r[8]}
            out.append({k: np.float32(v) for k, v in d.items()})
        return out
