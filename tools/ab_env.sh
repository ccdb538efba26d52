#!/bin/bash
# developer tool: short bench of the tree's library under environment switches, interleaved twice (same box)
# usage: tools/ab_env.sh "VAR=1" "OTHER=1" ...   (the plain run is always included)
cd /root/repo
for rep in 1 2; do
  for e in "" "$@"; do
    echo "== ${e:-default}"
    env $e python bench.py --steps 300 --warmup 10 --no-fit --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_us'))"
  done
done
