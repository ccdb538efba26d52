#!/bin/bash
# developer tool: the small kernels at the end of the cycle's chain under rocprofv3, for every library under variants/
cd /tmp && export TMPDIR=/tmp
for f in /root/repo/variants/lib*.so; do
  v=$(basename $f .so)
  MHHIP_LIB=$f timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$v -o x -- python /root/repo/bench.py --steps 100 --warmup 10 --no-fit --no-cpu-baseline > /dev/null 2>&1 < /dev/null
  s=$(find /tmp/pb_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  [ -n "$s" ] && grep "k_pose_bwd\|k_pose_fwd\|k_person_reduce\|k_rmsprop\|k_raster_finish\|k_raster_lists" "$s" | awk -F, '{n=$1; sub(/\(.*/,"",n); print n, "calls", $(NF-6), "avg_ns", $(NF-4)}'
done
