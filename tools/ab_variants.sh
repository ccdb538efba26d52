#!/bin/bash
# developer tool: key hash, selection-phase time, pair counters and a short bench of every library under variants/
cd /root/repo
export PYTHONPATH=/root/repo/scene-aware-3d-multi-human_amd
for rep in 1 2; do
for f in variants/lib*.so; do
  echo "== $f"
  export MHHIP_LIB=/root/repo/$f
  python tools/raster_keys.py 2>&1 | grep "selection phase\|sha1" | cut -c1-90
  if [ $rep = 1 ]; then python tools/pair_stats.py 2>&1 | grep launches; fi
  python bench.py --steps 200 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_us'))"
done
done
