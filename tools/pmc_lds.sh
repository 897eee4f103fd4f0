#!/bin/bash
# GPU box: LDS-side counters of the extractor kernels (one blocking 32-frame batch at a time).  bash tools/pmc_lds.sh <tag>
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_lds_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1 -- python3 $root/tools/quick_bench.py 32 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p2 -- python3 $root/tools/quick_bench.py 32 > $out/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1', 'p2'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('orbfe::', '').replace('(anonymous namespace)::', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(agg):
        top = max(len(v) for v in agg[k].values())
        line = []
        for c in sorted(agg[k]):
            v = agg[k][c]
            big = [x for x in v if x >= 0.8 * max(v)] or v
            line.append('%s=%.4g' % (c, sum(big) / len(big)))
        print(p, k[:36].ljust(36), ' '.join(line))
PY
