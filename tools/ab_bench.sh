#!/bin/bash
# developer tool: short bench of every library under variants/, twice (same box)
cd /root/repo
for rep in 1 2 3; do
for f in variants/lib*.so; do
  echo "== $f"
  MHHIP_LIB=/root/repo/$f python bench.py --steps 300 --warmup 10 --no-fit --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_us'))"
done
done
