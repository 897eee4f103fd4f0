#!/bin/bash
# GPU box: latency-side counters of the extractor kernels (one blocking 32-frame batch at a time).  bash tools/pmc_lat.sh <tag>
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_lat_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH_LEVEL SQ_IFETCH SQ_LEVEL_WAVES SQ_WAVES --kernel-trace --output-format csv -d $out/p1 -- python3 $root/tools/quick_bench.py 32 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $out/p2 -- python3 $root/tools/quick_bench.py 32 > $out/p2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $out/p3 -- python3 $root/tools/quick_bench.py 32 > $out/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1', 'p2', 'p3'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('orbfe::', '').replace('(anonymous namespace)::', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(agg):
        line = []
        for c in sorted(agg[k]):
            v = agg[k][c]
            big = [x for x in v if x >= 0.8 * max(v)] or v
            line.append('%s=%.4g' % (c, sum(big) / len(big)))
        print(p, k[:30].ljust(30), ' '.join(line))
PY
