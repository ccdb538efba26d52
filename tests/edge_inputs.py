"""sub-sequences of the standard fixture inputs used by the smallest-problem fixture (tests/golden/make_golden_edge.py)"""
import numpy as np
import torch

import golden_inputs as gi

VARIANTS = {'t1': (1, [0, 1], 5), 'n1': (7, [0], 3), 't2': (2, [1, 0], 5)}          # frames, humans kept, batch size


def sub_inputs(tag):
    T, people, batch = VARIANTS[tag]
    fin = gi.fit_inputs()
    f = dict(fin)
    for k in ['pose2d', 'poses_smpl', 'betas_smpl', 'valid_smpl', 'seg_mask']:
        f[k] = np.ascontiguousarray(fin[k][:T][:, people])
    for k in ['images', 'depths', 'backmasks']:
        if k in fin:
            f[k] = fin[k][:T]
    f['T'], f['N'] = T, len(people)
    return f, batch


def batches(f, b):
    out = []
    for s in range(0, f['T'], b):
        sl = slice(s, s + b)
        out.append(dict(idxs=torch.arange(s, min(s + b, f['T'])), pose2d=torch.tensor(f['pose2d'][sl]),
                        seg_mask=torch.tensor(f['seg_mask'][sl]), depths=torch.tensor(f['depths'][sl]),
                        poses_smpl=torch.tensor(f['poses_smpl'][sl])))
    return out
