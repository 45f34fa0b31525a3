#!/bin/bash
# GPU box: the same walk_probe cases on SEVERAL builds of the tree in one call (same box, interleaved twice).
# usage: abn_probe.sh tree [tree ...] -- case [case ...]      (tree "." = this tree; others relative to the repo root)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TREES=()
while [ "$1" != "--" ]; do TREES+=("$1"); shift; done
shift
for round in 1 2; do
  for t in "${TREES[@]}"; do
    echo "== round $round: $t"
    (cd "$ROOT/$t" && WALK_T="${WALK_T:-1,2}" WALK_EX="${WALK_EX:-}" timeout -k 10 240 python3 scripts/walk_probe.py "$@")
  done
done
