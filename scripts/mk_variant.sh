#!/bin/bash
# Exploration (this container): copy the tree to _exp/<name> and build librfhip.so there with experiment macros, for
# scripts/ab_probe.sh.  The macros are written as #defines at the top of rf_device.h of the COPY, so that kernels compiled
# at graph creation (hiprtc) see them too.  usage: mk_variant.sh <name> [MACRO[=value] ...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
DST="$ROOT/_exp/$NAME"
rm -rf "$DST"; mkdir -p "$DST"
(cd "$ROOT" && tar cf - --exclude='*.o' --exclude='*.so' --exclude=build --exclude=__pycache__ reforge_amd oracle include bench.py scripts tests/util.py tests/kat.py 2>/dev/null) | (cd "$DST" && tar xf -)
H="$DST/reforge_amd/csrc/rf_device.h"
for m in "$@"; do
  k="${m%%=*}"; v="${m#*=}"; [ "$k" = "$m" ] && v=1
  sed -i "1i #define $k $v" "$H"
done
make -C "$DST/reforge_amd/csrc" -j4 ../librfhip.so > "$DST/build.log" 2>&1 || { tail -30 "$DST/build.log"; exit 1; }
make -C "$DST/oracle" >> "$DST/build.log" 2>&1 || true
echo "built $DST ($*)"
