#!/bin/bash
# GPU box: the bench headline at several forced chunk heights (RF_ROWS_PER_CHUNK)
cd "$(dirname "$0")/.."
for r in "$@"; do
  RF_ROWS_PER_CHUNK=$r timeout -k 10 200 python bench.py --skip-cpu-baseline ${BENCH_ARGS} 2>/dev/null > /tmp/b.json
  python - "$r" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("rpc", sys.argv[1], "value", d["value"], "ms_per_frame", d["config"]["ms_per_frame"], "launch_ms", d["roofline"]["launch_ms"])
PY
done
