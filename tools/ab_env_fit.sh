#!/bin/bash
# developer tool: the bench's steady cycle and fit figures under environment switches (same box, interleaved)
cd /root/repo
for rep in 1 2; do
  for e in "" "$@"; do
    echo "== ${e:-default}"
    env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fit_250']; print(d['ms_per_step'], 'organic', d['organic_scene']['ms_per_step'], 'fit250', f['wall_s'], 'cycles', f['cycles_s'], 'early', f['early_fit']['ms_per_cycle'], f['early_fit']['bodies_resorted_share'])"
  done
done
