"""Timeline of one replayed cycle from a rocprofv3 kernel trace (developer tool):
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --no-cpu-baseline
  python tools/trace_cycle.py <dir> [cycle-index]
Prints start, duration, gap to the end of everything before it, queue and name of every kernel between two RMSprop
launches: serial micro-kernels, cross-queue hops and kernels stretched by what runs beside them show up directly."""
import csv, glob, sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 250
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rmsprop' in r['Kernel_Name']]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]['End_Timestamp'])
prev_end = t0
print('start_us  dur_us  gap_us  queue  kernel')
for r in rows[a + 1:b + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f %7.1f %6.1f  q%s  %s' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:48]))
    prev_end = max(prev_end, en)
# the graph boundary: from the end of the update to the first kernel of the next cycle, and the period over the next cycles
if b + 1 < len(rows):
    nxt = int(rows[b + 1]['Start_Timestamp'])
    print('update ends at %.1f us; first kernel of the next cycle starts %.1f us later (%s)' % (
        (int(rows[b]['End_Timestamp']) - t0) / 1e3, (nxt - int(rows[b]['End_Timestamp'])) / 1e3, rows[b + 1]['Kernel_Name'][:40]))
ends = [int(rows[i]['End_Timestamp']) for i in idx[which:which + 11]]
if len(ends) > 1:
    print('period over the next %d cycles: %.1f us' % (len(ends) - 1, (ends[-1] - ends[0]) / 1e3 / (len(ends) - 1)))
