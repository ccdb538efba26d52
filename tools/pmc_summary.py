"""Sum rocprofv3 --pmc counters per kernel (developer tool): python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        if pat in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    print(k, 'dispatches', len(n[k]))
    for c, v in sorted(acc[k].items()):
        print('   %-32s %.4g per dispatch' % (c, v / max(1, len(n[k]))))
