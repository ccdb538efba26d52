#!/bin/bash
# developer tool (GPU box, through gpurun): kernel timeline of one replayed C3 cycle, static scene and organic scene --
# both queues, start / duration / gap of every kernel, the join -- into gpurun_out/timeline_<tag>/ (copy into profiles/).
#   bash tools/timeline.sh r04
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
W=$R/gpurun_out/tl_$TAG
O=$R/gpurun_out/timeline_$TAG
rm -rf $W; mkdir -p $W $O
rocprofv3 --kernel-trace --output-format csv -d $W/t -o t -- python $R/bench.py --no-cpu-baseline --no-fit --steps 20 --warmup 5 > $W/t.log 2>&1
cd $R
{
  echo "# one replayed C3 cycle (static scene), bench.py --no-cpu-baseline --no-fit --steps 20 --warmup 5 under rocprofv3 --kernel-trace"
  python tools/trace_cycle.py $W/t 250
  echo
  echo "# one replayed C3 cycle of the organic-scene path (scene rebuilt on the device every cycle, own stream)"
  python tools/trace_cycle_with.py $W/t k_scene_median 50
} > $O/${TAG}_cycle_timeline.txt 2>&1
tail -3 $W/t.log > $O/${TAG}_bench_under_trace.txt
rm -rf $W
