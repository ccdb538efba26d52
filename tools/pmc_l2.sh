#!/bin/bash
# developer tool (GPU box): L2 (TCC) request / hit / miss counters of the LBS kernels in the bench cycle
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
W=/tmp/pmcl2; rm -rf $W; mkdir -p $W
CMD="python $R/bench.py --no-cpu-baseline --no-fit --steps 6 --warmup 2 --presteps 10"
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $W/s$i -o q -- $CMD > $W/s$i.log 2>&1
  tail -2 $W/s$i.log
done
for k in k_skinbwd16 k_skin_fwd16 k_raster_prepare k_filtered_verts; do python $R/tools/pmc_summary.py $W $k; done
