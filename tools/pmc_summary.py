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
        print('   per wave: VALU %.0f SALU %.0f LDS %.0f' % tuple(
            sum(d[c][len(d[c]) // 2:]) / len(d[c][len(d[c]) // 2:]) / w for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS')))
