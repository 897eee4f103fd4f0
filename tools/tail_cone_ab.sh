#!/bin/bash
# Round 6, one bounded experiment: the tail of a batch's pyramid (levels base + 1 .. 7) in ONE k_pyramid_cone launch instead of one
# k_resize_fixed launch per level.  Needs the experiments build (tools/build_exp.sh).  Prints the resident stream rate per setting.
cd "$(dirname "$0")/.."
export ORBFE_LIB=$PWD/os1_amd/liborbfe_exp.so
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-pcie --no-latency --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-52s value %.0f  p50 %s  verified %s' % ('$*', d['value'], d.get('value_p50'), d['verified']))"
}
run ORBFE_TAIL_CONE_BASE=0
run ORBFE_TAIL_CONE_BASE=4 ORBFE_TAIL_CONE_TILE=32
run ORBFE_TAIL_CONE_BASE=4 ORBFE_TAIL_CONE_TILE=48
run ORBFE_TAIL_CONE_BASE=4 ORBFE_TAIL_CONE_TILE=64
run ORBFE_TAIL_CONE_BASE=5 ORBFE_TAIL_CONE_TILE=48
run ORBFE_TAIL_CONE_BASE=5 ORBFE_TAIL_CONE_TILE=96
run ORBFE_TAIL_CONE_BASE=0
