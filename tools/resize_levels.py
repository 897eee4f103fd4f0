"""Per-level duration of the pyramid chain from a rocprofv3 --kernel-trace csv dir (GPU box).
usage: resize_levels.py <trace dir> [B]   -- groups the k_resize_fixed launches by grid size (= level) and prints
mean / min / max duration, bytes moved and the fraction of the 8 TB/s HBM roof per level and for the chain."""
import collections
import csv
import glob
import sys

B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_resize' not in n and 'k_pyramid' not in n:
            continue
        g = (n.split('(')[0].split('::')[-1][:40], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)),
             int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0))))
        rows[g].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
# level sizes at 1080p, scale 1.2 (cvRound of the level-0 size times the inverse scale factor)
W, H = 1920, 1080
sizes = [(W, H)]
s = 1.0
for l in range(1, 8):
    s *= 1.2
    inv = 1.0 / s
    sizes.append((int(round(W * inv)), int(round(H * inv))))
tot = 0.0
full = {}
for g, v in rows.items():
    top = max(len(x) for x in rows.values())
    full[g] = v
print('%-42s %10s %6s %8s %8s %8s' % ('kernel', 'grid', 'calls', 'mean us', 'min us', 'max us'))
for g in sorted(rows, key=lambda k: -k[1]):
    v = sorted(rows[g])
    # drop warm-up outliers: keep the central 80 %
    k = len(v) // 10
    c = v[k:len(v) - k] if len(v) > 10 else v
    print('%-42s %10d %6d %8.1f %8.1f %8.1f' % (g[0], g[1], len(v), sum(c) / len(c), v[0], v[-1]))
print('level sizes:', sizes)
for l in range(1, 8):
    rd = sizes[l - 1][0] * sizes[l - 1][1] * B
    wr = sizes[l][0] * sizes[l][1] * B
    tiles = ((sizes[l][0] + 63) // 64) * ((sizes[l][1] + 63) // 64)
    print('level %d: %dx%d  tiles/frame %d  blocks %d  read %.1f MB  write %.1f MB' % (l, sizes[l][0], sizes[l][1], tiles, tiles * B, rd / 1e6, wr / 1e6))
