#!/bin/bash
# hardware-queue matrix: resident rate of the stream runner by GPU_MAX_HW_QUEUES x depth x frames per submission
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
for q in 4 8; do for cfg in "2 64" "3 64" "4 64" "2 128" "3 128"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 3 --depth $1 --batch $2 --no-pcie --no-latency --cpu-frames 0 --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('queues $q depth $1 batch $2: value %.0f  p50 %s' % (d['value'], d.get('value_p50')))" | tee -a gpurun_out/r06j/queue_matrix.txt
done; done
