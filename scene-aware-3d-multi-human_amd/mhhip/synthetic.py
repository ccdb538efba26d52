"""Seeded synthetic SMPL-shaped body model and MuPoTs-shaped sequences.

The licensed ``SMPL_NEUTRAL.pkl`` is not distributable (reference
``.gitignore:16``, ``README.md:52``) and the MuPoTs pre-processed inputs are
not available offline, so tests, ``bench.py`` and ``smoke()`` run on a body
model with exactly SMPL's array shapes (V=6890, F=13776, 24 joints, 10 shape
and 207 pose blend-shapes) and on sequences generated here.  This is the
build's own generator; nothing in it comes from the reference.

Field names follow the pickle the reference unpacks in ``mhmocap/smpl.py:201-275``
(``v_template, shapedirs, posedirs, J_regressor, kintree_table, weights, f``) so
the same ``Struct`` can be handed to the reference's ``SMPL(data_struct=...)``
when golden vectors are captured.
"""
import math

import numpy as np

NUM_VERTS = 6890
NUM_FACES = 13776
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207

# standard SMPL kinematic tree (reference reads it from kintree_table[0], smpl.py:270-272)
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int64)

# rough rest-pose joint locations (metres, y up, pelvis near origin)
_REST_JOINTS = np.array([
    [0.00, -0.05, 0.00],   # 0 pelvis
    [0.07, -0.14, 0.00],   # 1 l hip
    [-0.07, -0.14, 0.00],  # 2 r hip
    [0.00, 0.06, -0.02],   # 3 spine1
    [0.09, -0.52, 0.00],   # 4 l knee
    [-0.09, -0.52, 0.00],  # 5 r knee
    [0.00, 0.20, -0.01],   # 6 spine2
    [0.08, -0.78, -0.02],  # 7 l ankle
    [-0.08, -0.78, -0.02],  # 8 r ankle
    [0.00, 0.27, 0.00],    # 9 spine3
    [0.09, -0.80, 0.06],   # 10 l foot
    [-0.09, -0.80, 0.06],  # 11 r foot
    [0.00, 0.46, -0.02],   # 12 neck
    [0.06, 0.38, -0.01],   # 13 l collar
    [-0.06, 0.38, -0.01],  # 14 r collar
    [0.00, 0.58, 0.00],    # 15 head
    [0.10, 0.42, -0.02],   # 16 l shoulder
    [-0.10, 0.42, -0.02],  # 17 r shoulder
    [0.11, 0.22, -0.03],   # 18 l elbow
    [-0.11, 0.22, -0.03],  # 19 r elbow
    [0.12, 0.02, -0.01],   # 20 l wrist
    [-0.12, 0.02, -0.01],  # 21 r wrist
    [0.12, -0.05, 0.00],   # 22 l hand
    [-0.12, -0.05, 0.00],  # 23 r hand
], dtype=np.float64)


class Struct(object):
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _capsule_mesh(rings=82, segs=84):
    """Closed genus-0 capsule: rings*segs + 2 verts = 6890, 2*segs*rings faces = 13776."""
    assert rings * segs + 2 == NUM_VERTS and 2 * segs * rings == NUM_FACES
    verts = np.zeros((NUM_VERTS, 3), np.float64)
    y_top, y_bot = 0.78, -0.90
    # smooth radius profile: feet .. hips .. chest .. head
    verts[0] = [0.0, y_top, 0.0]
    for r in range(rings):
        u = (r + 1) / (rings + 1)            # 0 (top) -> 1 (bottom)
        y = y_top + (y_bot - y_top) * u
        rad = 0.035 + 0.14 * math.sin(math.pi * u) ** 0.6 + 0.03 * math.sin(3 * math.pi * u)
        for s in range(segs):
            a = 2 * math.pi * s / segs
            verts[1 + r * segs + s] = [rad * 1.25 * math.cos(a), y, rad * 0.8 * math.sin(a)]
    verts[-1] = [0.0, y_bot, 0.0]
    faces = []
    for s in range(segs):
        faces.append([0, 1 + s, 1 + (s + 1) % segs])
    for r in range(rings - 1):
        a0 = 1 + r * segs
        b0 = a0 + segs
        for s in range(segs):
            s1 = (s + 1) % segs
            faces.append([a0 + s, b0 + s, b0 + s1])
            faces.append([a0 + s, b0 + s1, a0 + s1])
    last = 1 + (rings - 1) * segs
    for s in range(segs):
        faces.append([NUM_VERTS - 1, last + (s + 1) % segs, last + s])
    faces = np.asarray(faces, np.int64)
    assert faces.shape == (NUM_FACES, 3)
    return verts, faces


def _sparse_regressor(rng, verts, targets, nnz_per_row, spread=0.06):
    """Rows of positive weights summing to 1 on the vertices closest to each target."""
    R = np.zeros((len(targets), len(verts)), np.float64)
    for j, (tgt, k) in enumerate(zip(targets, nnz_per_row)):
        d = np.linalg.norm(verts - tgt[None], axis=1) + rng.uniform(0, spread, len(verts))
        idx = np.argsort(d)[:k]
        w = rng.uniform(0.2, 1.0, k)
        R[j, idx] = w / w.sum()
    return R


def make_smpl_struct(seed=1):
    """SMPL-shaped model arrays (float64/uint32 like the unpickled original)."""
    rng = np.random.RandomState(seed)
    v_template, faces = _capsule_mesh()
    # joint regressor: sparse, rows sum to 1, on vertices near each rest joint
    J_regressor = _sparse_regressor(rng, v_template, _REST_JOINTS,
                                    rng.randint(8, 40, NUM_JOINTS), spread=0.10)
    joints = J_regressor @ v_template
    # skinning weights: <= 4 bones per vertex (as in the real model)
    d2 = ((v_template[:, None, :] - joints[None]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1)[:, :4]
    weights = np.zeros((NUM_VERTS, NUM_JOINTS), np.float64)
    for k in range(4):
        weights[np.arange(NUM_VERTS), order[:, k]] = np.exp(
            -d2[np.arange(NUM_VERTS), order[:, k]] / (2 * 0.08 ** 2)) + 1e-6
    # a third of the vertices are rigidly bound to one bone
    rigid = rng.rand(NUM_VERTS) < 0.33
    weights[rigid] = 0
    weights[rigid, order[rigid, 0]] = 1.0
    weights /= weights.sum(1, keepdims=True)
    # shape blend-shapes: smooth global modes + a little noise
    shapedirs = np.zeros((NUM_VERTS, 3, NUM_BETAS), np.float64)
    for l in range(NUM_BETAS):
        axis_scale = rng.uniform(-0.04, 0.04, 3)
        phase = rng.uniform(0, 2 * math.pi)
        mod = np.cos(2.5 * (l % 4 + 1) * v_template[:, 1] + phase)[:, None]
        shapedirs[:, :, l] = v_template * axis_scale[None] * mod + rng.normal(0, 0.002, (NUM_VERTS, 3))
    posedirs = rng.normal(0, 0.003, (NUM_VERTS, 3, NUM_POSE_BASIS))
    kintree = np.zeros((2, NUM_JOINTS), np.uint32)
    kintree[0] = SMPL_PARENTS.astype(np.uint32)   # parents[0] == 2**32-1 (== -1 as uint32)
    kintree[1] = np.arange(NUM_JOINTS)
    return Struct(v_template=v_template, f=faces.astype(np.uint32), shapedirs=shapedirs,
                  posedirs=posedirs, J_regressor=J_regressor, kintree_table=kintree,
                  weights=weights)


def make_extra_regressors(seed=1, struct=None):
    """Synthetic stand-ins for the four regressor .npy files of
    ``model_data/parameters`` with their on-disk shapes/dtypes and sparsity
    (reference smpl.py:234-261): extra9 (9,V) f64, h36m (17,V) f64,
    alphapose (V,17) f32 (~670 nnz), mupots (V,17) f32 (~1160 nnz)."""
    struct = struct or make_smpl_struct(seed)
    rng = np.random.RandomState(seed + 100)
    v = struct.v_template
    j24 = struct.J_regressor @ v
    # 17 keypoints in a COCO-like order placed near body joints
    kp_src = [15, 15, 15, 15, 15, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]
    tg17 = j24[kp_src] + rng.normal(0, 0.02, (17, 3))
    alphapose = _sparse_regressor(rng, v, tg17, [31, 15, 14, 8, 17, 22, 34, 11, 10, 63, 48, 83, 50, 33, 54, 92, 88])
    mupots = _sparse_regressor(rng, v, tg17[::-1] + 0.01, [4, 89, 51, 9, 79, 87, 21, 15, 163, 15, 27, 78, 12, 29, 344, 40, 97])
    h36m = _sparse_regressor(rng, v, j24[[0, 2, 5, 8, 1, 4, 7, 3, 12, 15, 15, 16, 18, 20, 17, 19, 21]],
                             [8, 6, 6, 5, 5, 7, 4, 6, 8, 8, 6, 8, 5, 6, 9, 6, 4])
    extra9 = _sparse_regressor(rng, v, j24[[8, 5, 2, 1, 4, 7, 12, 15, 0]] + 0.01,
                               [5, 5, 9, 1, 8, 12, 6, 8, 8])
    return {
        'extra9': extra9.astype(np.float64),
        'h36m': h36m.astype(np.float64),
        'alphapose': alphapose.T.astype(np.float32).copy(),
        'mupots': mupots.T.astype(np.float32).copy(),
    }


# ---------------------------------------------------------------------------------------------
# sequences
# ---------------------------------------------------------------------------------------------

def _rodrigues_np(r):
    a = np.linalg.norm(r)
    if a < 1e-12:
        return np.eye(3)
    k = r / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(a) * K + (1 - math.cos(a)) * (K @ K)


def make_sequence_params(num_people, num_frames, seed, z_range=(3.0, 8.0)):
    """Ground-truth and ROMP-like initial parameters for an N x T sequence."""
    rng = np.random.RandomState(seed)
    N, T = num_people, num_frames
    betas_gt = rng.normal(0, 0.5, (N, NUM_BETAS))
    theta = np.zeros((T, N, 72))
    theta0 = rng.normal(0, 0.2, (N, 72))
    # camera looks down +z with y down: flip the y-up template by pi about x
    theta0[:, 0:3] = np.array([math.pi, 0.0, 0.0]) + rng.normal(0, 0.15, (N, 3))
    theta[0] = theta0
    for t in range(1, T):
        theta[t] = theta[t - 1] + rng.normal(0, 0.02, (N, 72))
    theta[..., 66:] = 0.0
    pos = np.zeros((T, N, 3))
    pos[0, :, 0] = rng.uniform(-2, 2, N)
    pos[0, :, 1] = rng.uniform(0.15, 0.25, N)
    pos[0, :, 2] = rng.uniform(z_range[0], z_range[1], N)
    vel = rng.uniform(-0.03, 0.03, (N, 3)) * np.array([1, 0.0, 1])
    for t in range(1, T):
        vel = np.clip(vel + rng.normal(0, 0.004, (N, 3)) * np.array([1, 0.0, 1]), -0.05, 0.05)
        pos[t] = pos[t - 1] + vel
        pos[t, :, 0] = np.clip(pos[t, :, 0], -2.5, 2.5)
        pos[t, :, 2] = np.clip(pos[t, :, 2], z_range[0] - 0.2, z_range[1] + 0.5)
    poses_init = theta + rng.normal(0, 0.05, theta.shape)
    poses_init[..., 66:] = 0.0
    betas_init = betas_gt[None] + rng.normal(0, 0.3, (T, N, NUM_BETAS))
    valid = (rng.rand(T, N, 1) > 0.05).astype(np.float32)
    return dict(betas_gt=betas_gt.astype(np.float32), poses_gt=theta.astype(np.float32),
                trans_gt=pos.astype(np.float32), poses_init=poses_init.astype(np.float32),
                betas_init=betas_init.astype(np.float32), valid=valid)


def default_cam_K(image_size, fov=60.0):
    W, H = image_size
    f = 0.5 * min(W, H) / math.tan(math.pi * fov / 360.0)
    return np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
