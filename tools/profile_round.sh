#!/bin/bash
# Round profile on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of the same command, the
# HBM-traffic counters and the SQ instruction counters in separate --pmc passes (never combined with other trace
# domains), and the FETCH_SIZE calibration.  Usage: bash tools/profile_round.sh <tag>   (e.g. r02f)
# Writes gpurun_out/prof_<tag>/...; tools/profile_collect.py turns that into profiles/<tag>_* and profiles/counters.json.
tag=${1:-r02x}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 4 --warmup 1 --passes 2 --cpu-frames 0 --no-pcie --no-latency --prewarm-seconds 0 > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --steps 2 --warmup 1 --passes 1 --cpu-frames 0 --no-pcie --no-verify --no-latency --prewarm-seconds 0 > $out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_SQ -- python3 $root/bench.py --steps 2 --warmup 1 --passes 1 --cpu-frames 0 --no-pcie --no-verify --no-latency --prewarm-seconds 0 > $out/pmc_SQ.log 2>&1
# the scalar unit (round 5): its own pass
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc_SQ2 -- python3 $root/bench.py --steps 2 --warmup 1 --passes 1 --cpu-frames 0 --no-pcie --no-verify --no-latency --prewarm-seconds 0 > $out/pmc_SQ2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/calib -- $root/tools/ubench/fetch_calib > $out/calib.log 2>&1
python3 $root/tools/fetch_calib_summary.py $out/calib > $out/fetch_calibration.txt 2>&1
cat $out/fetch_calibration.txt
ls $out
