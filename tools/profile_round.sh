#!/bin/bash
# Round profile on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of the same command, and the
# HBM-traffic counters in separate --pmc passes.  Usage: bash tools/profile_round.sh <tag>   (e.g. r01e)
# Writes gpurun_out/prof_<tag>/...; tools/profile_collect.py turns that into profiles/<tag>_*.
tag=${1:-r01x}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 60 --warmup 6 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 30 --warmup 4 --cpu-frames 0 > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --steps 12 --warmup 3 --cpu-frames 0 > $out/pmc_$c.log 2>&1
done
ls -R $out | head -40
