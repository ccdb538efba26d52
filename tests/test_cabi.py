"""CPU-side checks of the drop-in boundary: the in-tree C-ABI library builds, loads without a GPU and
exports every symbol declared in include/mhmocap_hip.h; the product path refuses to run without a
HIP device (no CPU fallback); host-side helpers behave."""
import ctypes
import inspect
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from mhhip import build, _lib
    so = build.build()
    assert os.path.exists(so) and so.startswith(os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'))
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.mh_version() >= 1
    assert L.mh_device_count() >= 0
    assert L.mh_lbs_workspace_bytes(800) > 800 * 24 * 12 * 4
    assert L.mh_raster_workspace_bytes(200, 4, 6890, 13776, 135, 240) > 0


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the behaviour of a GPU-less host')
def test_no_cpu_fallback(smpl_struct, smpl_regs):
    from mhhip import engine, _lib
    with pytest.raises(_lib.MhError):
        engine.BodyModel(smpl_struct, smpl_regs)
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    with pytest.raises(RuntimeError):
        SMPLDepthSequenceOptimizer(image_size=(48, 32), num_frames=4, smpl_data_struct=smpl_struct)


def test_invalid_arguments_are_reported_not_crashing():
    from mhhip import _lib
    L = _lib.lib()
    rc = L.mh_rmsprop_step(None, None, None, None, 10, 0.01, 0.5, 0.9, 1e-8, None)
    assert rc == -1 and b'null' in L.mh_last_error()
    rc = L.mh_lbs_forward(None, 0, 0, None, None, None, None, None, None, None, None, None)
    assert rc == -1


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src and 'oracle.' not in src, os.path.join(dirpath, f)


def test_overlay_never_calls_into_the_module_it_shadows():
    """An overlay module may RE-EXPORT names of the reference module it shadows (mhhip/_overlay.inherit) but no function
    body of the product may execute one: the only place that touches ``__shadowed__`` is mhhip/_overlay.py."""
    import re
    pkg = os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if not f.endswith('.py') or path.endswith(os.path.join('mhhip', '_overlay.py')):
                continue
            code = re.sub(r'#.*', '', open(path).read())
            assert '__shadowed__' not in code and not re.search(r'\b_shadowed\s*\(', code), path


def test_overlay_signatures_match_the_reference_call_sites():
    """predict.py:290-306 / 332-344 call these with these keywords."""
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    from mhmocap.smpl import SMPL
    p = inspect.signature(SMPLDepthSequenceOptimizer.__init__).parameters
    for k in ['image_size', 'num_frames', 'fov', 'focal_length', 'znear', 'zfar', 'cam_K', 'cam_dist_coef',
              'proj2d_loss_coef', 'depth_loss_coef', 'silhouette_loss_coef', 'reg_velocity_coef', 'reg_verts_filter_coef',
              'reg_poses_coef', 'reg_scales_coef', 'reg_contact_coef', 'reg_foot_sliding_coef', 'joint_confidence_thr', 'eps']:
        assert k in p, k
    f = inspect.signature(SMPLDepthSequenceOptimizer.fit).parameters
    assert list(f)[1:] == ['dataloader', 'num_iter', 'min_cutoff1', 'min_cutoff2', 'beta1', 'beta2', 'update_filters_every', 'verbose']
    i = inspect.signature(SMPLDepthSequenceOptimizer.init_optimized_variables).parameters
    assert list(i)[1:] == ['pose2d', 'poses_smpl', 'betas_smpl', 'valid_smpl', 'scale_factor', 'num_iter']
    s = inspect.signature(SMPL.__init__).parameters
    for k in ['model_path', 'J_reg_extra9_path', 'J_reg_h36m17_path', 'J_reg_alphapose_path', 'J_reg_mupots_path', 'data_struct']:
        assert k in s, k


def test_shard_bounds_respect_batches():
    from mhhip.sharded import shard_bounds
    b = shard_bounds(2000, 8, 10)
    assert b[0][0] == 0 and b[-1][1] == 2000 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all((s % 10 == 0) for s, _ in b) and all(e - s == 250 for s, e in b)
    b = shard_bounds(205, 4, 10)
    assert b[-1][1] == 205 and all(s % 10 == 0 for s, _ in b)


def test_one_euro_time_base_matches_numpy_accumulation():
    from mhhip.engine import one_euro_time_before
    t = np.zeros((), np.float32)
    for i in range(1, 37):
        t = t + (i / 25)
    assert one_euro_time_before(37) == float(np.float32(t))
    assert one_euro_time_before(0) == 0.0 and one_euro_time_before(1) == 0.0


def test_argument_checks_return_status_and_message_without_a_device():
    """bad arguments are rejected before any HIP call: status < 0 and a readable mh_last_error()"""
    import ctypes
    from mhhip import _lib
    L = _lib.lib()
    L.mh_last_error.restype = ctypes.c_char_p
    rc = L.mh_raster_terms(1, 1, 10, 10, 8, 8, *([None] * 13), 0.0, 0.0, 1e-3, *([None] * 9))
    assert rc < 0 and b'null' in L.mh_last_error()
    rc = L.mh_contact_knn_grid(None, 10, None, 1, 32, None, None)
    assert rc < 0 and L.mh_last_error()
    assert L.mh_scene_median(0, 8, 8, *([None] * 8)) < 0
    assert L.mh_scene_workspace_bytes(4, 8, 8) > 0 and L.mh_raster_workspace_bytes(1, 1, 10, 10, 8, 8) > 0


def test_bench_refuses_a_rank_count_that_is_not_what_gpus_says():
    """``--gpus N`` is binding -- a launcher that started another number of ranks is an error, not a run that
    prints the wrong ``n_gpus`` (without a launcher bench.py starts the N ranks itself: tests/test_bench_multirank_gpu.py)"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in (p.stderr + p.stdout)
