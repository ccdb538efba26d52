#!/bin/bash
# developer tool: like ab_env.sh, prints the static and the organic-scene cycle of the bench
cd /root/repo
for rep in 1 2; do
  for e in "" "$@"; do
    echo "== ${e:-default}"
    env $e python bench.py --steps 300 --warmup 10 --no-fit --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'organic', d['organic_scene']['ms_per_step'])"
  done
done
