"""Timeline of the LAST blocking one-frame extraction in a rocprofv3 --kernel-trace csv (tools/latency_trace.py under the profiler):
every kernel with start / end relative to the call's first kernel and the queue it ran on -- the two launch chains of the latency route."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '').replace('(anonymous namespace)::', '')
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', '?')))
rows.sort()
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ends = [i for i, r in enumerate(rows) if r[2].startswith('k_describe')]
# a call = everything from its first kernel (k_ingest, or the first kernel after the previous call's last k_describe) to its last k_describe
last = ends[-1]
i = last
while i > 0 and rows[i][0] - rows[i - 1][1] < 400_000 and rows[last][1] - rows[i - 1][0] < 400_000:   # calls are 2 ms apart
    i -= 1
t0 = rows[i][0]
for s, e, n, q in rows[i:last + 1]:
    print('%-34s queue %-3s  start %7.1f  end %7.1f  dur %6.1f us' % (n[:34], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print('first kernel start -> last kernel end: %.1f us' % ((max(r[1] for r in rows[i:last + 1]) - t0) / 1e3))
