#!/bin/bash
# GPU box: the pyramid chain alone (one blocking extract_batch of 32 resident frames at a time, nothing else in flight),
# kernel trace -> per-level durations.  Usage (through gpurun): bash tools/resize_levels.sh <tag>
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/resize_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/quick_bench.py 32 > $out/quick.log 2>&1
python3 $root/tools/resize_levels.py $out/trace 32 > $out/levels.txt 2>&1
cat $out/quick.log | tail -3
cat $out/levels.txt
