"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.csv, and refresh profiles/traffic.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(root, 'profiles')
line = [l for l in open(os.path.join(src, 'bench.json')).read().splitlines() if l.startswith('{')][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)
B = bench['config']['frames_per_step_per_gpu']
stats = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)
shutil.copy(stats[0], os.path.join(dst, tag + '_kernel_stats.csv'))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join(src, 'pmc_' + c, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if int(r.get('Grid_Size_Y', r.get('Grid_Size_y', 0)) or 0) >= 0:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(os.path.join(dst, tag + '_pmc_summary.csv'), 'w') as f:
    f.write('kernel,counter,dispatches,avg_value_KB_per_dispatch\n')
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for k in sorted(agg):
            v = agg[k].get(c)
            if v:
                v = v[len(v) // 3:]          # steady state (skip warm-up / first-touch dispatches)
                f.write('%s,%s,%d,%.3f\n' % (k, c, len(v), sum(v) / len(v)))
fast = [k for k in agg if 'k_fast_cells' in k][0]
fv = agg[fast]['FETCH_SIZE']; fv = fv[len(fv) // 3:]
wv = agg[fast]['WRITE_SIZE']; wv = wv[len(wv) // 3:]
fetch, write = sum(fv) / len(fv) * 1024, sum(wv) / len(wv) * 1024
traffic = {
    'k_fast_cells_bytes_per_launch_b%d' % B: int(2 * fetch + write),
    'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over `python3 bench.py`, %d-frame launches), KB units; '
            'FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE reports half of wide coalesced reads); '
            'raw FETCH_SIZE*1024 = %d, WRITE_SIZE*1024 = %d' % (B, fetch, write),
    'source': 'profiles/%s_pmc_summary.csv' % tag,
}
json.dump(traffic, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print(json.dumps(bench)[:400])
print(traffic)
