"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, average counter value per dispatch."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
only = sys.argv[2:] 
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].split('<')[0].replace('orbfe::', '').replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    if not k.startswith('k_') or (only and k not in only):
        continue
    d = agg[k]
    w = None
    print(k)
    for c in sorted(d):
        v = d[c][len(d[c]) // 2:]
        a = sum(v) / len(v)
        if c == 'SQ_WAVES':
            w = a
        print('   %-22s n=%3d avg=%14.0f' % (c, len(v), a))
    if w and 'SQ_INSTS_VALU' in d:
        mean = lambda c: (sum(d[c][len(d[c]) // 2:]) / len(d[c][len(d[c]) // 2:])) if d.get(c) else float('nan')
        print('   per wave: VALU %.0f SALU %.0f LDS %.0f SMEM %.1f' % tuple(mean(c) / w for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM')))
        if d.get('SQ_ACTIVE_INST_SCA') and d.get('SQ_BUSY_CYCLES'):
            # SQ_ACTIVE_INST_* count, summed over the chip's SQs, cycles (in units of 4) in which a wave executes an instruction of that kind;
            # SQ_BUSY_CYCLES counts per shader engine: busy fraction of a port = ACTIVE / waves-weighted time is not derivable from them alone,
            # so print the ratio of the two ports (same unit) and the instruction ratio
            print('   scalar port / vector port: active %.3f, instructions %.3f' % (mean('SQ_ACTIVE_INST_SCA') / mean('SQ_ACTIVE_INST_VALU'),
                                                                                  mean('SQ_INSTS_SALU') / mean('SQ_INSTS_VALU')))
