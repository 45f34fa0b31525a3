#!/bin/bash
# GPU box (through gpurun): the round's standard check -- GPU tests, then the default bench line.
# usage: gpu_round.sh [pytest -k expression]
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
K="${1:-}"
if [ -n "$K" ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
else
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
fi
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print("power", d.get("power")); print("value", d["value"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "verified", d["verified"], "region_s", d["config"]["timed_region_s"])
for k, v in d.get("workloads", {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"]); continue
    if k == "conv31_8k":
        for p in ("valu", "mfma"):
            print(k, p, v[p]["ms_per_frame"], v[p]["roofline"]["frac"], v[p].get("verified"), v[p].get("power"))
    else:
        print(k, v["ms_per_frame"], v["roofline"]["frac"], v.get("frame_hbm_frac"), v.get("verified"), v.get("power"))
PY
