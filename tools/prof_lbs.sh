#!/bin/bash
# developer tool: per-kernel durations of tools/time_lbs.py (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_lbs
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/tools/time_lbs.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('skin', 'pose', 'skinbwd', 'person')):
            print('%-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
