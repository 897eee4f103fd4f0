"""Per-thread CPU use of one bench.py rank while the stream runs (which host threads a rank keeps busy)."""
import os, sys, subprocess, time, json
# run bench in-process-like: spawn and sample /proc/<pid>/task/*/stat near the end
p = subprocess.Popen([sys.executable, 'bench.py', '--steps', '4000', '--warmup', '5', '--no-latency', '--no-pcie', '--cpu-frames', '0', '--no-verify'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
def sample():
    out = {}
    for t in os.listdir('/proc/%d/task' % p.pid):
        try:
            f = open('/proc/%d/task/%s/stat' % (p.pid, t)).read()
            comm = f[f.index('(') + 1:f.rindex(')')]
            rest = f[f.rindex(')') + 2:].split()
            out[t] = (comm, int(rest[11]) + int(rest[12]))
        except Exception:
            pass
    return out
hz = os.sysconf('SC_CLK_TCK')
best = None
prev, tp = sample(), time.time()
while p.poll() is None:
    time.sleep(2)
    cur, tc = sample(), time.time()
    rows = [(cur[t][0] + ('(main)' if int(t) == p.pid else ''), round((cur[t][1] - prev[t][1]) / hz / (tc - tp), 2)) for t in cur if t in prev]
    tot = sum(r[1] for r in rows)
    if best is None or tot > best[0]: best = (tot, [r for r in rows if r[1] > 0.02])
    prev, tp = cur, tc
print('busiest 2-second window: %.2f cores' % best[0], best[1])
print(p.communicate()[0].decode()[-200:])
