"""The evaluator the optimiser's output goes to, with the call surface of reference ``mhmocap/evaluate.py`` (what
``eval_mupots.py:13-31`` imports: ``compute_smpl_pred_error_3dproj`` :180-296, ``masked_average_error`` :401-416,
``masked_average_pck`` :419-435, and the two joint-layout maps :93-98).

The reference walks the frames in Python (projection, a K x N x J tile, a loop over the K*N pairs and one over the matched
pairs, per frame): at T = 2000 that loop outlasts the 250 optimisation cycles it evaluates.  Here the body model runs
where ``SMPLPY`` lives (the drop-in ``mhmocap.smpl.SMPL``: the HIP forward, all T*N bodies in one call) and everything
behind it is whole-sequence array arithmetic on the host -- host logic in the reference as well; only the K x N
assignment of a frame stays a per-frame call (``scipy.optimize.linear_sum_assignment``, what ``utils.py:309`` calls).
Outputs, dtypes (float32) and the row convention (row k = k-th MATCHED pair of the frame, not the k-th reference person)
are the reference's; pinned by ``tests/golden/make_golden_eval.py``.

This module SHADOWS ``mhmocap.evaluate`` in the namespace overlay; the reference's other evaluators
(``compute_smpl_pred_error_ortho``, ``..._3dproj_matched``, ``match_pred_to_pref``: no shipped caller) are re-exported
by ``mhhip._overlay.inherit`` when a reference tree is on the path."""
import numpy as np

from .transforms import camera_projection

# MuPoTs-15 joint j as a mean of source joints (reference :28-64, there as weight / index lists)
_CMU19_OF_MUPOTS15 = ((1,), (0,), (9,), (10,), (11,), (3,), (4,), (5,), (12,), (13,), (14,), (6,), (7,), (8,), (2,))
_ALPHAPOSE17_OF_MUPOTS15 = ((0,), (5, 6), (6,), (8,), (10,), (5,), (7,), (9,), (12,), (14,), (16,), (11,), (13,), (15,),
                            (11, 12))


def _mean_of(pose, sources):
    assert pose.ndim == 3, 'Error: invalid input pose with shape %s' % (pose.shape,)
    out = np.zeros((pose.shape[0], len(sources), pose.shape[2]), np.float32)
    for j, src in enumerate(sources):
        w = np.float32(1.0 / len(src))
        out[:, j] = (w * pose[:, list(src)]).sum(axis=1)
    return out


def map_cmu_panoptic_to_mupots15j(pose):
    """(M,19,D) CMU-Panoptic joints -> (M,15,D) float32 in MuPoTs order (reference :93-94)"""
    return _mean_of(pose, _CMU19_OF_MUPOTS15)


def map_alphapose_to_mupots15j(pose):
    """(M,17,D) AlphaPose key-points -> (M,15,D) float32 in MuPoTs order (reference :97-98)"""
    return _mean_of(pose, _ALPHAPOSE17_OF_MUPOTS15)


def _match_costs(ref2d, pred2d, thr=0.5):
    """(T,K,J,3), (T,N,J,3) [x, y, visibility] -> (T,K,N) float32: mean distance over the joints visible in both, 1e6 where
    there is none (reference utils.py:295-307; the distance runs over all three channels, visibility included, as there)"""
    d = np.sqrt(np.sum(np.square(ref2d[:, :, None] - pred2d[:, None]), axis=-1))
    v = (ref2d[:, :, None, :, 2] > thr) & (pred2d[:, None, :, :, 2] > thr)
    n = v.sum(-1)
    s = np.where(v, d, 0).sum(-1)
    return np.where(n > 0, s / np.maximum(n, 1), 1e6).astype(np.float32)


def compute_smpl_pred_error_3dproj(output_data, ref_poses3d, visibility, SMPLPY, cam_K, Kd=None):
    """Per-joint errors of the optimised bodies against ground-truth poses, predictions matched to reference persons per
    frame by their projected 2D joints (reference :180-296).

    output_data: dict of ``get_optimized_variables()``: 'poses_T' (T,N,1,3), 'poses_smpl' (T,N,72), 'betas_smpl' (T,N,10),
    'scale_factor' (T or 1,N,1,1).  ref_poses3d (T,K,17 or 19,3), visibility (T,K,J,1).  Returns the reference's dict:
    abs_dist, rel_dist, valid_joints, abs_jitter (T,K,14), abs_root_pos_err, valid_root (T,K), all float32."""
    from scipy.optimize import linear_sum_assignment
    poses_T = np.asarray(output_data['poses_T'])
    scale = np.asarray(output_data['scale_factor'])
    T, N = poses_T.shape[0:2]
    if scale.shape[0] == 1:
        scale = np.tile(scale, (T, 1, 1, 1))
    K, J = ref_poses3d.shape[1:3]
    assert (J == 17) or (J == 19), (
        f'Invalid number of joints ({J}), only 17 (MuPoTs) or 19 (Panoptic) joints are supported, {J} joints given!')
    if J == 19:
        ref_poses3d = map_cmu_panoptic_to_mupots15j(ref_poses3d.reshape((T * K, -1, 3))).reshape((T, K, -1, 3))
        visibility = map_cmu_panoptic_to_mupots15j(visibility.reshape((T * K, -1, 1))).reshape((T, K, -1, 1))
    else:
        ref_poses3d = ref_poses3d[:, :, 0:15]
        visibility = visibility[:, :, 0:15]

    results = SMPLPY(betas=np.asarray(output_data['betas_smpl']).reshape((-1, 10)),
                     poses=np.asarray(output_data['poses_smpl']).reshape((-1, 72)))
    if J == 19:
        j15 = map_alphapose_to_mupots15j(results['joints_alphapose'].cpu().numpy().reshape((T * N, -1, 3))).reshape((T, N, -1, 3))
    else:
        j15 = results['joints_mupots'].cpu().numpy().reshape((T, N, 17, 3))[:, :, 0:15, :]
    pred3d = scale * j15 + poses_T                                                     # (T,N,15,3)

    ref2d = camera_projection(ref_poses3d.reshape((-1, 3)), cam_K, Kd=Kd).reshape((T, K, -1, 2))
    ref2d = np.concatenate([ref2d, visibility], axis=-1)
    pred2d = camera_projection(pred3d.reshape((-1, 3)), cam_K, Kd=Kd).reshape((T, N, -1, 2))
    pred2d = np.concatenate([pred2d, np.ones_like(pred2d[..., 0:1])], axis=-1)
    cost = _match_costs(ref2d, pred2d)

    # the assignment of every frame, as index tables: pair m of frame t = (reference gi[t,m], prediction pi[t,m])
    M = min(K, N)
    gi = np.zeros((T, M), np.int64)
    pi = np.zeros((T, M), np.int64)
    for t in range(T):
        gi[t], pi[t] = linear_sum_assignment(cost[t])
    tt = np.arange(T)[:, None]
    gt = ref_poses3d[tt, gi]                                                           # (T,M,15,3)
    pr = pred3d[tt, pi]
    vis = visibility[tt, gi][..., 0]                                                   # (T,M,15)

    z = lambda *s: np.zeros(s, np.float32)
    abs_dist, rel_dist, valid_joints = z(T, K, 14), z(T, K, 14), z(T, K, 14)
    abs_root_pos_err, valid_root = z(T, K), z(T, K)
    matched_ref, matched_pred = z(T, K, 14, 3), z(T, K, 14, 3)
    root_seen = vis[..., 14] > 0
    valid_root[:, :M] = root_seen
    abs_root_pos_err[:, :M] = np.where(root_seen, np.sqrt(np.sum(np.square(gt[:, :, 14] - pr[:, :, 14]), axis=-1)), 0)
    matched_ref[:, :M] = gt[:, :, :14]
    matched_pred[:, :M] = pr[:, :, :14]
    abs_dist[:, :M] = np.sqrt(np.sum(np.square(gt[:, :, :14] - pr[:, :, :14]), axis=-1))
    rel_dist[:, :M] = np.sqrt(np.sum(np.square((gt[:, :, :14] - gt[:, :, 14:15]) - (pr[:, :, :14] - pr[:, :, 14:15])), axis=-1))
    valid_joints[:, :M] = vis[..., :14] > 0.49

    step = lambda x: np.sqrt(np.sum(np.square(x[1:] - x[:-1]), axis=-1))
    abs_jitter = np.abs(step(matched_ref) - step(matched_pred))
    abs_jitter = np.concatenate([abs_jitter[0:1], abs_jitter], axis=0)
    return {
        'abs_dist': abs_dist,
        'rel_dist': rel_dist,
        'valid_joints': valid_joints,
        'abs_root_pos_err': abs_root_pos_err,
        'valid_root': valid_root,
        'abs_jitter': abs_jitter,
    }


def _masked(dist, vis):
    assert dist.shape == vis.shape, (f'Invalid input shapes ({dist.shape}), ({vis.shape})')
    return dist.reshape((-1,)).astype(np.float32), (vis > 0.5).reshape((-1,)).astype(np.float32)


def masked_average_error(dist, vis):
    """mean of ``dist`` over the entries with ``vis > 0.5`` (reference :401-416)"""
    dist, vis = _masked(dist, vis)
    return np.sum(vis * dist) / np.clip(np.sum(vis), 1, None)


def masked_average_pck(dist, vis, thr):
    """fraction of the entries with ``vis > 0.5`` whose ``dist <= thr`` (reference :419-435)"""
    dist, vis = _masked(dist, vis)
    return np.sum(vis * (dist <= thr)) / np.clip(np.sum(vis), 1, None)


from mhhip._overlay import inherit as _inherit  # noqa: E402

_inherit(globals())
