"""Summary of tools/fast_ablation.sh: per ablation level the k_fast_tasks counters per cell-wave and its mean duration."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
names = {0: 'full kernel', 1: 'set-up + ROI load (+ score-tile zeroing)', 2: '+ stage 1: pre-test and ordered compaction',
         3: '+ stage 2: arc score', 5: 'full kernel without the second pass (minThFAST)'}
rows = {}
for a in (0, 1, 2, 3, 5):
    agg = collections.defaultdict(list)
    for f in glob.glob('%s/pmc_%d/**/*counter_collection.csv' % (d, a), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_fast_tasks' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    dur = []
    for f in glob.glob('%s/trace_%d/**/*kernel_trace.csv' % (d, a), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_fast_tasks' in r['Kernel_Name']:
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    if not agg.get('SQ_WAVES'):
        print('level %d: no counters' % a)
        continue
    top = max(agg['SQ_WAVES'])
    keep = [i for i, w in enumerate(agg['SQ_WAVES']) if w >= 0.8 * top]          # full 32-frame launches
    mean = lambda k: sum(agg[k][i] for i in keep) / len(keep) if agg.get(k) else float('nan')
    dur = sorted(x for x in dur if x >= 0.5 * max(dur)) if dur else [float('nan')]
    rows[a] = dict(waves=mean('SQ_WAVES'), valu=mean('SQ_INSTS_VALU') / mean('SQ_WAVES'), lds=mean('SQ_INSTS_LDS') / mean('SQ_WAVES'),
                   salu=mean('SQ_INSTS_SALU') / mean('SQ_WAVES'), cpi=4 * mean('SQ_ACTIVE_INST_VALU') / mean('SQ_INSTS_VALU'),
                   wait=mean('SQ_WAIT_ANY') / mean('SQ_WAVE_CYCLES'), us=dur[len(dur) // 2], n=len(keep))
print('k_fast_tasks stage ablation, 32 x 1080p frames per launch, kernel alone (tools/quick_bench.py 32); per cell-wave:')
print('%-46s %9s %7s %7s %7s %9s %9s' % ('stages run', 'VALU', 'LDS', 'SALU', 'cyc/VALU', 'wait frac', 'launch us'))
for a in (1, 2, 3, 5, 0):
    if a in rows:
        r = rows[a]
        print('%-46s %9.1f %7.1f %7.1f %7.2f %9.2f %9.1f   (%d launches, %d waves each)' % (names[a], r['valu'], r['lds'], r['salu'], r['cpi'], r['wait'], r['us'], r['n'], r['waves']))
if all(a in rows for a in range(4)):
    print('increments (VALU per cell-wave): set-up %.0f | stage 1 %.0f | stage 2 %.0f | stage 3 (NMS + emission) %.0f' % (
        rows[1]['valu'], rows[2]['valu'] - rows[1]['valu'], rows[3]['valu'] - rows[2]['valu'], rows[0]['valu'] - rows[3]['valu']))
if 5 in rows and 0 in rows:
    print('second pass (cells without a keypoint at iniThFAST, again at minThFAST): %.0f VALU per cell-wave on average; stage 3 of the first pass alone: %.0f' % (
        rows[0]['valu'] - rows[5]['valu'], rows[5]['valu'] - rows[3]['valu']))
