#!/bin/bash
# developer tool (GPU box): SQ counters of one kernel of the bench cycle:  bash tools/pmc_kernel.sh <kernel-substring>
K=${1:-k_raster_grads}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
W=/tmp/pmck; rm -rf $W; mkdir -p $W
CMD="python $R/bench.py --no-cpu-baseline --no-fit --steps 6 --warmup 2"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $W/s$i -o q -- $CMD > $W/s$i.log 2>&1
done
python $R/tools/pmc_summary.py $W $K
