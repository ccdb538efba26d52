"""Condense rocprofv3 outputs under gpurun_out/ into the committed profiles/ (developer tool).
  python tools/make_profiles.py <round-tag> <stats-dir> <fetch-dir> <write-dir> [out-dir]"""
import csv, glob, json, os, shutil, sys, collections

tag, dstats, dfetch, dwrite = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, 'profiles')
os.makedirs(out, exist_ok=True)
st = glob.glob(os.path.join(dstats, '**', '*kernel_stats.csv'), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, '%s_bench_c3_kernel_stats.csv' % tag))


def per_kernel(d, counter):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
                acc[k] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
    return {k: acc[k] / len(n[k]) for k in acc}


# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 B?  The derived metric is
# (TCC_EA0_RDREQ_32B*32 + (RDREQ - RDREQ_32B)*64)/1024 -> kilobytes; store bytes.
fetch, write = per_kernel(dfetch, 'FETCH_SIZE'), per_kernel(dwrite, 'WRITE_SIZE')
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith('k_'):
        traffic[k] = {'fetch_bytes': round(fetch.get(k, 0.0) * 1024.0), 'write_bytes': round(write.get(k, 0.0) * 1024.0),
                      'note': 'rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB) x 1024, per launch, separate passes; gfx950: '
                              'wide (16 B/lane) streaming reads are under-counted by 2x, other widths uncalibrated '
                              '(MI355X_MICROARCH.md, HBM)'}
json.dump(traffic, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in traffic.items() if k in ('k_raster_strip', 'k_skin_fwd16p', 'k_skin_fwd16', 'k_skinbwd16', 'k_raster_grads')}, indent=1))

# stamp: what these passes were taken on (bench.py prints counter-derived numbers only for sources that still match)
import hashlib, subprocess
csrc = os.path.join(ROOT, 'scene-aware-3d-multi-human_amd', 'csrc')
sha = {f: hashlib.sha256(open(os.path.join(csrc, f), 'rb').read()).hexdigest() for f in sorted(os.listdir(csrc)) if os.path.isfile(os.path.join(csrc, f))}
try:
    git = subprocess.run(['git', '-C', ROOT, 'rev-parse', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
except Exception:
    git = None
json.dump({'round': tag, 'git': git or os.environ.get('GRAFT_GIT_HEAD'), 'sha256': sha}, open(os.path.join(out, 'stamp.json'), 'w'), indent=1, sort_keys=True)
