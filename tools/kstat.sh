#!/bin/bash
# developer tool (GPU box): average duration of the named kernels inside the bench's replayed cycle for the tree's library
# and each library given (rocprofv3 --kernel-trace --stats, csv):  KERNELS="k_raster_grads|k_raster_strip" tools/kstat.sh variants/lib_a.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${KERNELS:-k_raster_grads|k_raster_strip|k_skin}
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  if [ "$v" = default ]; then unset MHHIP_LIB; else export MHHIP_LIB=$(realpath $R/$v 2>/dev/null || echo $v); fi
  d=/tmp/kstat_$$_$(basename $v .so); rm -rf $d
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o s -- python $R/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-fit > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "$K" $f | awk -F'","|",|,"' '{printf "%-60s calls %s avg_ns %s\n", substr($1,2,58), $2, $4}'
done
