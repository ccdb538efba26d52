#!/bin/bash
# developer tool: the eight-process dry run with the parity cycle repeated inside every rank (tools/c4_probe.py)
cd /root/repo
C="--steps 1 --warmup 1 --presteps 1 --no-cpu-baseline --no-fit"
for a in ${RUNS:-0 1}; do
MHHIP_C4_SERIAL=${SERIAL:-0} MHHIP_C4_PROBE=${REPS:-3} HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NP:-8} --master-addr 127.0.0.1 --master-port $((29771+a)) bench.py --gpus ${NP:-8} --backend gloo --one-device --config c4 --dump-leaves /tmp/probe$a.npz $C 2>&1 | grep "c4_probe" | sort
done
