"""Kernel / copy timeline of the LAST SearchByProjection call in a rocprofv3 --kernel-trace --memory-copy-trace run of
tools/config5_bench.py: per activity start offset, duration and the idle gap before it."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('orbfe::', '').replace('orbfe_match::', '').replace('void ', '')))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', r.get('Name', ''))))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'k_resolve' in r[2]]
if not ends:
    sys.exit('no k_resolve in the trace')
for which in (len(ends) // 2, len(ends) - 1):
    e = ends[which]
    s = e
    while s > 0 and 'k_window_match' not in rows[s][2]:
        s -= 1
    while s > 0 and rows[s - 1][2].startswith('copy') and rows[s][0] - rows[s - 1][1] < 50000:
        s -= 1
    t0, prev = rows[s][0], None
    last = e + 1 if e + 1 < len(rows) and rows[e + 1][2].startswith('copy') else e
    print('--- search #%d' % which)
    for a, b, n in rows[s:last + 1]:
        print('%-40s start %7.1f us  dur %7.1f us  gap before %6.1f us' % (n[:40], (a - t0) / 1e3, (b - a) / 1e3, (a - prev) / 1e3 if prev else 0.0))
        prev = b
    print('first start -> last end: %.1f us' % ((rows[last][1] - t0) / 1e3))
