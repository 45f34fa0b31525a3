#!/bin/bash
# Exploration (GPU box): rebuild rf_kernels.o with an experiment switch and time the 4K chain.
# usage: variant_probe.sh "<HIPEXTRA flags>" [more variants...]
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f reforge_amd/csrc/build/rf_stream.o reforge_amd/csrc/build/rf_conv.o reforge_amd/csrc/build/rf_misc.o
  make -C reforge_amd/csrc ../librfhip.so HIPEXTRA="$v" > /dev/null 2>&1
  PROBE_UNFUSED=${PROBE_UNFUSED:-0} PROBE_REST=${PROBE_REST:-0} PROBE_RPC=${PROBE_RPC:-0} timeout -k 10 200 python scripts/gpu_probe.py 2>&1 | grep -E "fused|8K|passthrough" | cut -c1-160
done
rm -f reforge_amd/csrc/build/rf_stream.o reforge_amd/csrc/build/rf_conv.o reforge_amd/csrc/build/rf_misc.o
make -C reforge_amd/csrc ../librfhip.so > /dev/null 2>&1
