#!/bin/bash
# A/B of the one-frame call (tools/latency_quick.py) under the switches of the latency route, each setting in a process of its own.
cd "$(dirname "$0")/.."
for env in "" "$@"; do
  echo "== ${env:-default}"
  env $env python tools/latency_quick.py
done
