#!/bin/bash
cd $GRAFT_REPO_ROOT
export ORBFE_LIB=$PWD/os1_amd/liborbfe_exp.so
for t in 0 24 25 26 27 28 30 32; do
  echo "== cone tile $t"
  ORBFE_CONE_TILE=$t python tools/latency_quick.py 2>&1 | grep -E "resident|page_locked  |Error|assert" | head -3
done
