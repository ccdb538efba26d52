"""developer tool: per-kernel averages over a window of update launches of a rocprofv3 kernel trace of tools/fit_timeline.py run
  python tools/fit_kernels.py <trace-dir> lo hi [lo hi ...]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rmsprop' in r['Kernel_Name']]
print('%d update launches' % len(idx))
w = [int(x) for x in sys.argv[2:]]
for lo, hi in zip(w[::2], w[1::2]):
    a, b = idx[lo], idx[hi]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a:b]:
        n = r['Kernel_Name'][:44]; acc[n][0] += 1; acc[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    span = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3 / (hi - lo)
    print('updates %d..%d: %.1f us per cycle' % (lo, hi, span))
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
        print('   %-46s %5.2f per cycle  avg %7.1f us  sum/cycle %7.1f' % (n, c / (hi - lo), t / c, t / (hi - lo)))
