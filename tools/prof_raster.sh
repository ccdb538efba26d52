#!/bin/bash
# developer tool: per-kernel durations of tools/raster_keys.py (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_raster
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/tools/raster_keys.py > $OUT/log.txt 2>&1
grep "keys sha1\|different" $OUT/log.txt
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'raster' in r['Name']:
            print('%-50s calls %5s avg %9.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
