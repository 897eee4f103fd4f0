"""Where do the occasional 8-12 ms steps come from?  Kernel trace of a bench run: longest kernels and longest
intervals without any kernel running.  usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...;
python3 tools/stall_probe.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')))
rows.sort()
print(len(rows), 'kernel launches, span %.1f ms' % ((rows[-1][1] - rows[0][0]) / 1e6))
long = sorted(rows, key=lambda r: r[1] - r[0], reverse=True)[:6]
for s, e, n in long:
    print('long kernel %-28s %8.3f ms at +%.1f ms' % (n, (e - s) / 1e6, (s - rows[0][0]) / 1e6))
# idle intervals: sweep
end = rows[0][1]
gaps = []
for s, e, n in rows[1:]:
    if s > end:
        gaps.append((s - end, end))
    end = max(end, e)
gaps.sort(reverse=True)
for g, at in gaps[:6]:
    print('idle %8.3f ms at +%.1f ms' % (g / 1e6, (at - rows[0][0]) / 1e6))
