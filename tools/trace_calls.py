"""Timeline (kernels + copies) of the last burst of GPU activity in a rocprofv3 --kernel-trace --memory-copy-trace csv dir."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('orbfe::', '').replace('void ', '')))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '')))
rows.sort()
gap = int(sys.argv[2]) if len(sys.argv) > 2 else 500000   # ns of idle that separates two calls
end = len(rows) - 1
start = end
while start > 0 and rows[start][0] - rows[start - 1][1] < gap:
    start -= 1
t0, prev = rows[start][0], None
for a, b, n in rows[start:end + 1]:
    print('%-34s start %7.1f us  dur %7.1f us  gap before %6.1f us' % (n[:34], (a - t0) / 1e3, (b - a) / 1e3, (a - prev) / 1e3 if prev else 0.0))
    prev = b
print('first start -> last end: %.1f us' % ((rows[end][1] - t0) / 1e3))
