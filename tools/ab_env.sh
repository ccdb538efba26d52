#!/bin/bash
# developer tool: same-box A/B of an environment switch of the engine -- ab_env.sh NAME [pairs]
cd /root/repo
for rep in $(seq 1 ${2:-4}); do
for v in 0 1; do
  echo "== $1=$v"
  env $1=$v python bench.py --steps 300 --warmup 10 --no-fit --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['organic_scene']['value'])"
done
done
