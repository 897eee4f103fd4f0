#!/bin/bash
# GPU box: what k_pyramid_cone (and the other kernels of the one-frame call) wait for -- instruction fetch, LDS, memory.  bash tools/pmc_cone.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_cone
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/p1 -- python3 $root/tools/latency_trace.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --kernel-trace --output-format csv -d $out/p2 -- python3 $root/tools/latency_trace.py > $out/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/p3 -- python3 $root/tools/latency_trace.py > $out/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1', 'p2', 'p3'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('orbfe::', '').replace('(anonymous namespace)::', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(agg):
        print('%-32s' % k[:32], '  '.join('%s=%.0f' % (c, sorted(v)[len(v) // 2]) for c, v in sorted(agg[k].items())))
PY
