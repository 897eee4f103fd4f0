"""Per-kernel start/end of ONE blocking single-frame extract from a rocprofv3 --kernel-trace csv: durations and the idle
gaps between consecutive kernels (GPU-side dependency + launch latency)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')))
rows.sort()
# last complete extract: find the last k_describe and walk back to the preceding k_resize run
end = max(i for i, r in enumerate(rows) if r[2].startswith('k_describe'))
start = end
pyr = lambda n: n.startswith('k_resize') or n.startswith('k_pyramid_cone')
while start > 0 and not (pyr(rows[start][2]) and not pyr(rows[start - 1][2])):
    start -= 1
prev_end = None
tot_k = tot_g = 0
for s, e, n in rows[start:end + 1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%-28s dur %7.1f us   gap before %6.1f us' % (n, (e - s) / 1e3, gap))
    tot_k += (e - s) / 1e3
    tot_g += gap
    prev_end = e
print('kernels %.1f us + gaps %.1f us = %.1f us from first kernel start to last kernel end' % (tot_k, tot_g, (rows[end][1] - rows[start][0]) / 1e3))
