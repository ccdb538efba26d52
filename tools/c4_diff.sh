#!/bin/bash
# developer tool: where the eight-rank dry run's first-cycle gradients differ from the one-process run (frame indices)
cd /root/repo
C="--steps 2 --warmup 1 --presteps 1 --no-cpu-baseline --no-fit"
python bench.py --gpus 1 --frames 2000 --dump-leaves /tmp/one.npz $C > /dev/null 2>&1
for a in 0 1 2; do
HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29871+a)) bench.py --gpus 8 --backend gloo --one-device --config c4 --dump-leaves /tmp/eight$a.npz $C > /dev/null 2>&1
python - <<PY
import numpy as np
A=np.load('/tmp/one.npz'); B=np.load('/tmp/eight$a.npz')
for k in ('grad0_zmin_lin','grad0_zmax_lin','grad0_poses_T'):
    d=np.abs(A[k]-B[k])/np.abs(A[k]).max()
    d=d.reshape(d.shape[0],-1).max(axis=1) if d.ndim>1 else d
    idx=np.nonzero(d>2e-5)[0]
    print('attempt $a',k,'frames/rows over 2e-5:',len(idx),idx[:24].tolist(), ['%.1e'%x for x in d[idx[:8]]])
PY
done
