#!/bin/bash
# GPU box: the same walk_probe cases on two builds of the tree in ONE call (same box, interleaved twice) -- A/B of a kernel
# change against box-to-box variance.  usage: ab_probe.sh <other tree> case [case ...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OTHER="$1"; shift
for round in 1 2; do
  for tree in "$ROOT" "$ROOT/$OTHER"; do
    echo "== round $round: $tree"
    (cd "$tree" && timeout -k 10 200 python3 scripts/walk_probe.py "$@")
  done
done
