#!/bin/bash
# GPU box: stage ablation of k_fast_tasks (ORBFE_FAST_ABLATE=0..3: full kernel / set-up + ROI load / + pre-test and
# compaction / + score stage), one blocking 32-frame 1080p batch at a time (tools/quick_bench.py 32): vector instructions
# per cell-wave from a --pmc pass, kernel duration from a separate --kernel-trace pass.  Usage (through gpurun):
#   bash tools/fast_ablation.sh <tag>     -> gpurun_out/fast_abl_<tag>/summary.txt
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/fast_abl_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3 5; do
  export ORBFE_FAST_ABLATE=$a
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc_$a -- python3 $root/tools/quick_bench.py 32 > $out/pmc_$a.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$a -- python3 $root/tools/quick_bench.py 32 > $out/trace_$a.log 2>&1
done
unset ORBFE_FAST_ABLATE
python3 $root/tools/fast_ablation_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
