"""Summarise a rocprofv3 --kernel-trace [--memory-copy-trace] run: per-kernel totals, union busy time, idle gaps."""
import csv
import glob
import sys

root = sys.argv[1]
kt = []
for f in glob.glob(root + '/**/*_kernel_trace.csv', recursive=True):
    kt += list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')) for r in kt)
# steady state: the last 60 % of the run
t0, t1 = ev[0][0], ev[-1][1]
lo = t0 + (t1 - t0) * 4 // 10
ev = [e for e in ev if e[0] >= lo]
span = ev[-1][1] - ev[0][0]
tot = {}
for s, e, n in ev:
    tot[n] = tot.get(n, 0) + (e - s)
busy = 0
cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('window %.2f ms, some kernel running %.1f %%' % (span / 1e6, 100.0 * busy / span))
for n in sorted(tot, key=lambda k: -tot[k]):
    print('  %-28s %7.2f ms  %5.1f %% of window' % (n, tot[n] / 1e6, 100.0 * tot[n] / span))
