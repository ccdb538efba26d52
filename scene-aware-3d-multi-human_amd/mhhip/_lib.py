"""ctypes binding of include/mhmocap_hip.h.  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in the product path."""
import ctypes
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MHHIP_LIB') or os.path.join(HERE, 'libmhmocap_hip.so')   # override: kernel experiments
HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'mhmocap_hip.h')

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int32)
vp = ctypes.c_void_p


class MhError(RuntimeError):
    pass


class ModelHost(ctypes.Structure):
    _fields_ = [('num_verts', ctypes.c_int32), ('num_faces', ctypes.c_int32),
                ('v_template', c_float_p), ('shapedirs', c_float_p), ('posedirs', c_float_p),
                ('J_regressor', c_float_p), ('lbs_weights', c_float_p), ('parents', c_int_p),
                ('faces', c_int_p), ('reg_alphapose', c_float_p), ('reg_h36m17', c_float_p),
                ('reg_mupots', c_float_p), ('reg_extra9', c_float_p)]


class FwdProj(ctypes.Structure):
    """mh_fwd_proj of include/mhmocap_hip.h: where the LBS forward's projection epilogue writes for one raster workspace"""
    _fields_ = [('s', ctypes.c_float), ('w1', ctypes.c_float), ('h1', ctypes.c_float),
                ('ra', ctypes.c_float), ('rk', ctypes.c_float), ('thr', ctypes.c_float),
                ('slack_ndc', ctypes.c_float), ('slack_y', ctypes.c_float), ('thr_soft', ctypes.c_float),
                ('ndc', vp), ('rowb', vp), ('bbox', vp), ('bbox_prev', vp), ('lowkey', vp), ('lowkey_prev', vp),
                ('moved', vp), ('clear', vp), ('clear_n', ctypes.c_ulonglong)]


class RasterFin(ctypes.Structure):
    """mh_raster_fin of include/mhmocap_hip.h: the rasterised terms' closing job, left by mh_raster_terms_deferred for the
    LBS backward's pose kernel (mh_lbs_backward_kp_fin)"""
    _fields_ = [('T', ctypes.c_int), ('N', ctypes.c_int), ('B', ctypes.c_int), ('from_partials', ctypes.c_int),
                ('coef_depth', ctypes.c_float)] + [(k, vp) for k in (
                    'body_first', 'body_ns', 'partial', 'dinv', 'sil_apply', 'sil_D', 'sil_S', 'sil_corr', 'depth_body',
                    'sil_body', 'zmin_lin', 'zmax_lin', 'gzmin', 'gzmax', 'log_depth', 'log_sil')] + [
                    ('has_lists', ctypes.c_int), ('lists', ctypes.c_ulonglong * 80)]


class PersonSums(ctypes.Structure):
    """mh_person_sums of include/mhmocap_hip.h: the per-body shape / scale gradients a backward with gbetas = gxscale = NULL
    left in its workspace, for the update that sums them (mh_rmsprop_step_person)"""
    _fields_ = [('gbeta_b', vp), ('gxs_b', vp), ('B', ctypes.c_int), ('NB', ctypes.c_int), ('nbeta', ctypes.c_int),
                ('off_betas', ctypes.c_longlong), ('off_xscale', ctypes.c_longlong)]


_lib = None


def declared_symbols():
    """Every function the public header declares (used by the load/export test)."""
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mh_[a-z0-9_]+)\s*\(', txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MhError('%s is missing: build it with `python -m mhhip.build` (hipcc, gfx950). '
                          'There is no CPU fallback.' % LIB_PATH)
        # torch first: it ships its own HIP/HSA runtime libraries with the same SONAMEs as /opt/rocm; two
        # different runtimes in one process see no devices, so ours must bind to the ones torch loaded
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.mh_last_error.restype = ctypes.c_char_p
        L.mh_model_faces.restype = vp
        L.mh_lbs_workspace_bytes.restype = ctypes.c_size_t
        L.mh_lbs_backward_workspace_bytes.restype = ctypes.c_size_t
        L.mh_lbs_workspace_bytes.argtypes = [ctypes.c_int]
        L.mh_lbs_backward_workspace_bytes.argtypes = [ctypes.c_int]
        L.mh_model_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ModelHost)]
        L.mh_model_destroy.argtypes = [vp]
        L.mh_model_faces.argtypes = [vp]
        L.mh_lbs_forward.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 9
        L.mh_lbs_forward_rotmats.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 8
        L.mh_lbs_forward_ex.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 10
        L.mh_joints_regress.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp]
        L.mh_lbs_backward.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 14
        L.mh_lbs_backward_ex.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 15
        L.mh_joints_regress_backward.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp]
        L.mh_project_joints_loss.argtypes = [ctypes.c_int, vp, c_float_p, c_float_p, vp, ctypes.c_float, ctypes.c_int,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        L.mh_project_joints_loss_w.argtypes = [ctypes.c_int, vp, c_float_p, c_float_p, c_float_p, vp, ctypes.c_float,
                                               ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        L.mh_warmup_project_w.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, c_float_p, c_float_p, c_float_p, vp,
                                          ctypes.c_float, ctypes.c_float, vp, vp, vp]
        L.mh_lbs_set_mode.argtypes = [ctypes.c_int]
        L.mh_lbs_set_forward_pipeline.argtypes = [ctypes.c_int]
        L.mh_rmsprop_step.argtypes = [vp, vp, vp, vp, ctypes.c_size_t] + [ctypes.c_float] * 4 + [vp]
        L.mh_rmsprop_step_log.argtypes = [vp, vp, vp, vp, ctypes.c_size_t] + [ctypes.c_float] * 4 + [vp, vp, ctypes.c_int, vp]
        L.mh_rmsprop_step_log_poke.argtypes = [vp, vp, vp, vp, ctypes.c_size_t] + [ctypes.c_float] * 4 + [vp, vp, ctypes.c_int, vp,
                                                                                                   ctypes.c_int, ctypes.c_int32, ctypes.c_int32, vp]
        L.mh_rmsprop_step_person.argtypes = [vp, vp, vp, vp, ctypes.c_size_t] + [ctypes.c_float] * 4 + [vp, vp, ctypes.c_int, vp,
                                                                                                 ctypes.c_int, ctypes.c_int32, ctypes.c_int32,
                                                                                                 ctypes.POINTER(PersonSums), vp]
        L.mh_lbs_backward_person_partials.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
        L.mh_lbs_person_reduce.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.mh_rmsprop_step_dev.argtypes = [vp, vp, vp, vp, ctypes.c_size_t, vp] + [ctypes.c_float] * 4 + [vp]
        L.mh_adam_step.argtypes = [vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_int] + [ctypes.c_float] * 4 + [vp]
        L.mh_one_euro_scan.argtypes = [vp, vp, ctypes.c_int, ctypes.c_size_t] + [ctypes.c_float] * 3 + [vp]
        L.mh_one_euro_scan_shard.argtypes = [vp, vp, ctypes.c_int, ctypes.c_size_t] + [ctypes.c_float] * 3 + [ctypes.c_int, ctypes.c_float, vp, vp, vp, vp]
        L.mh_velocity_term.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_float, vp, vp, vp]
        L.mh_filtered_verts_term.argtypes = [ctypes.c_int, ctypes.c_size_t] + [vp] * 6 + [ctypes.c_float, vp, vp, vp, vp]
        L.mh_filtered_verts_term_init.argtypes = [ctypes.c_int, ctypes.c_size_t] + [vp] * 6 + [ctypes.c_float, vp, vp, vp, vp]
        L.mh_filtered_verts_term_init_gated.argtypes = [ctypes.c_int, ctypes.c_size_t] + [vp] * 6 + [ctypes.c_float, vp, vp, vp, vp, vp]
        L.mh_filtered_verts_workspace_bytes.restype = ctypes.c_size_t
        L.mh_filtered_verts_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_size_t]
        u32p = vp
        L.mh_warmup_project.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, c_float_p, c_float_p, vp,
                                        ctypes.c_float, ctypes.c_float, vp, vp, vp]
        L.mh_pack_masks.argtypes = [vp] + [ctypes.c_int] * 4 + [u32p, vp, vp]
        L.mh_erode_bits.argtypes = [u32p, u32p] + [ctypes.c_int] * 3 + [vp]
        L.mh_stage_gates.argtypes = [vp, vp, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        L.mh_sil_mask_stats.argtypes = [u32p] + [ctypes.c_int] * 4 + [vp] * 8
        L.mh_sil_mask_stats_cached.argtypes = [u32p] + [ctypes.c_int] * 4 + [vp] * 9
        L.mh_prior_terms.argtypes = [ctypes.c_int] * 3 + [vp] * 6 + [ctypes.c_float] * 2 + [vp] * 6
        L.mh_reduce_sum.argtypes = [vp, ctypes.c_size_t, ctypes.c_float, vp, vp]
        L.mh_reduce_sum_multi.argtypes = [ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(vp), vp]
        L.mh_reduce_sum2.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_float, vp, vp, vp]
        L.mh_lowest_vertex.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]
        L.mh_contact_knn.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]
        L.mh_scene_workspace_bytes.restype = ctypes.c_size_t
        L.mh_scene_workspace_bytes.argtypes = [ctypes.c_int] * 3
        L.mh_scene_median.argtypes = [ctypes.c_int] * 3 + [vp] * 8
        L.mh_scene_median_t.argtypes = [ctypes.c_int] * 3 + [vp] * 8
        L.mh_scene_postprocess.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]
        L.mh_scene_fill.argtypes = [ctypes.c_int] * 4 + [vp] * 4
        L.mh_scene_points.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]
        L.mh_stream_create.argtypes = [ctypes.POINTER(vp)]
        L.mh_stream_destroy.argtypes = [vp]
        L.mh_stream_shares_any.argtypes = [ctypes.POINTER(vp), ctypes.c_int, vp, ctypes.c_float, ctypes.POINTER(ctypes.c_int)]
        L.mh_streams_classify.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp), ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int)]
        L.mh_stream_spin.argtypes = [vp, ctypes.c_float]
        L.mh_streams_share_queue.argtypes = [vp, vp, ctypes.c_float, ctypes.POINTER(ctypes.c_int)]
        L.mh_profile_enable.argtypes = [ctypes.c_int]
        L.mh_profile_read.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.mh_scene_grid_bytes.restype = ctypes.c_size_t
        L.mh_scene_grid_bytes.argtypes = [ctypes.c_int]
        L.mh_scene_grid_build.argtypes = [vp, ctypes.c_int, vp, vp]
        L.mh_scene_grid_build_dev.argtypes = [vp, vp, ctypes.c_int, vp, vp]
        L.mh_contact_knn_grid.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]
        L.mh_contact_knn_grid_key.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.mh_contact_knn_grid_sel.argtypes = [vp, vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.mh_contact_foot_terms_gated.argtypes = [ctypes.c_int] * 5 + [vp] * 5 + [ctypes.c_float] * 2 + [vp] * 6
        L.mh_contact_foot_terms.argtypes = [ctypes.c_int] * 4 + [vp] * 4 + [ctypes.c_float] * 2 + [vp] * 5
        L.mh_contact_foot_terms_idx.argtypes = [ctypes.c_int] * 5 + [vp] * 5 + [ctypes.c_float] * 2 + [vp] * 5
        L.mh_scene_unproject.argtypes = [vp, ctypes.c_int, ctypes.c_int, c_float_p, vp, vp]
        L.mh_raster_workspace_bytes.restype = ctypes.c_size_t
        L.mh_raster_workspace_bytes.argtypes = [ctypes.c_int] * 6
        L.mh_raster_terms.argtypes = [ctypes.c_int] * 6 + [c_float_p] + [vp] * 12 + [ctypes.c_float] * 3 + [vp] * 9
        L.mh_raster_terms_phase.argtypes = [ctypes.c_int] * 6 + [c_float_p] + [vp] * 12 + [ctypes.c_float] * 3 + [vp] * 8 + [ctypes.c_int, vp]
        L.mh_raster_set_deterministic.argtypes = [ctypes.c_int]
        L.mh_raster_set_sort_margin.argtypes = [ctypes.c_int]
        L.mh_raster_set_sort_defer.argtypes = [ctypes.c_float]
        L.mh_raster_get_sort_defer.restype = ctypes.c_float
        L.mh_raster_set_path.argtypes = [ctypes.c_int]
        L.mh_raster_set_winners.argtypes = [ctypes.c_int]
        L.mh_raster_pair_counters.argtypes = [ctypes.c_int] * 6 + [vp, ctypes.POINTER(ctypes.c_ulonglong), vp]
        L.mh_raster_sort_counters.argtypes = [ctypes.c_int] * 6 + [vp, ctypes.POINTER(ctypes.c_ulonglong), vp]
        L.mh_raster_sort_counters3.argtypes = [ctypes.c_int] * 6 + [vp, ctypes.POINTER(ctypes.c_ulonglong), vp]
        L.mh_raster_terms_phase_log.argtypes = [ctypes.c_int] * 6 + [c_float_p] + [vp] * 12 + [ctypes.c_float] * 3 + [vp] * 8 + [ctypes.c_int, vp, vp, vp]
        L.mh_raster_terms_projected.argtypes = [ctypes.c_int] * 6 + [c_float_p] + [vp] * 12 + [ctypes.c_float] * 3 + [vp] * 8 + [ctypes.c_int, vp, vp, ctypes.c_int, vp]
        L.mh_raster_forward_targets.argtypes = [ctypes.c_int] * 6 + [c_float_p, vp, ctypes.POINTER(FwdProj)]
        L.mh_lbs_forward_proj.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 6 + [ctypes.POINTER(FwdProj), vp, vp]
        L.mh_lowest_resolve.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.mh_keypoint_workspace_bytes.restype = ctypes.c_size_t
        L.mh_keypoint_workspace_bytes.argtypes = [vp, ctypes.c_int]
        L.mh_keypoint_terms.argtypes = [vp, ctypes.c_int, vp, c_float_p, c_float_p, c_float_p, vp] + [ctypes.c_float] * 4 + [vp] * 8
        L.mh_lbs_backward_kp.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 11
        L.mh_lbs_backward_kp_fin.argtypes = [vp, ctypes.c_int, ctypes.c_int] + [vp] * 10 + [ctypes.POINTER(RasterFin), vp]
        L.mh_raster_terms_deferred.argtypes = [ctypes.c_int] * 6 + [c_float_p] + [vp] * 12 + [ctypes.c_float] * 3 + [vp] * 8 + [
            ctypes.c_int, vp, vp, ctypes.c_int, ctypes.POINTER(RasterFin), vp]
        L.mh_raster_workspace_init.argtypes = [ctypes.c_int] * 6 + [vp, vp]
        L.mh_raster_workspace_offsets.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
        L.mh_avg_depth_loss.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_float, vp, vp, vp]
        L.mh_avg_depth_loss_backward.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_float, vp, ctypes.c_float, vp, vp, vp]
        L.mh_masked_mse.argtypes = [vp, vp, vp, ctypes.c_size_t, vp, vp]
        L.mh_masked_mse_backward.argtypes = [vp, vp, vp, ctypes.c_size_t, vp, ctypes.c_float, vp, vp]
        L.mh_morph_f32.argtypes = [vp, vp] + [ctypes.c_int] * 5 + [vp]
        L.mh_project_points.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, c_float_p, ctypes.c_int, vp, vp]
        L.mh_unproject_points.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        _lib = L
    return _lib


def reduce_sum_multi(items, stream):
    """items: [(x tensor, out tensor of one float)] -> every out = sum(x), one launch"""
    n = len(items)
    xs = (vp * n)(*[t.data_ptr() for t, _ in items])
    ls = (ctypes.c_size_t * n)(*[t.numel() for t, _ in items])
    os_ = (vp * n)(*[o.data_ptr() for _, o in items])
    check(lib().mh_reduce_sum_multi(n, xs, ls, os_, stream))


def check(rc):
    if rc != 0:
        raise MhError('mhmocap_hip error %d: %s' % (rc, lib().mh_last_error().decode()))


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return t.data_ptr()


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def host_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_float_p)


def on_own_device(cls):
    """Class decorator: every public method runs with the object's device current (``self.dev`` / ``self.device``).  The C
    ABI launches on the stream it is handed and allocates / sets attributes on the CURRENT device, so an engine built for
    "cuda:1" must not depend on the caller having selected that device (predict.py passes an arbitrary device_name)."""
    import functools
    import torch

    def wrap(fn):
        @functools.wraps(fn)
        def inner(self, *a, **kw):
            dev = getattr(self, 'dev', None) or getattr(self, 'device', None)
            if dev is None or getattr(dev, 'type', 'cpu') != 'cuda' or torch.cuda.current_device() == (dev.index or 0):
                return fn(self, *a, **kw)
            with torch.cuda.device(dev):
                return fn(self, *a, **kw)
        return inner
    for name, attr in list(vars(cls).items()):
        if callable(attr) and not name.startswith('__') and not isinstance(attr, (staticmethod, classmethod, property)):
            setattr(cls, name, wrap(attr))
    return cls
