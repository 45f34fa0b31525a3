#!/bin/bash
# GPU box: rebuild the kernels with an experiment switch and run the headline bench + passthrough
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f reforge_amd/csrc/build/rf_stream.o
  make -C reforge_amd/csrc ../librfhip.so HIPEXTRA="$v" > /dev/null 2>&1 || { echo "build failed"; continue; }
  bash scripts/bench_rpc.sh 0
  PROBE_UNFUSED=0 PROBE_REST=1 PROBE_RPC=0 timeout -k 10 200 python scripts/gpu_probe.py 2>&1 | grep -E "passthrough|8K" | cut -c1-120
done
rm -f reforge_amd/csrc/build/rf_stream.o
make -C reforge_amd/csrc ../librfhip.so > /dev/null 2>&1
