"""Like trace_cycle.py, but prints the first replayed cycle (between two RMSprop launches) that contains a kernel whose
name has the given substring (e.g. k_scene_median for a cycle of the organic-scene path):
  python tools/trace_cycle_with.py <trace-dir> <substring> [skip]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
pat = sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rmsprop' in r['Kernel_Name']]
hits = [k for k in range(len(idx) - 1) if any(pat in r['Kernel_Name'] for r in rows[idx[k]:idx[k + 1]])]
print('cycles with', pat, ':', len(hits))
k = hits[min(skip, len(hits) - 1)]
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]['End_Timestamp']); prev = t0
print('start_us  dur_us  gap_us  queue  kernel')
for r in rows[a + 1:b + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f %7.1f %6.1f  q%s  %s' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:50]))
    prev = max(prev, en)
