#!/bin/bash
# GPU box: rocprofv3 passes behind profiles/ (run from the repo root through gpurun).
#   pass 1  --kernel-trace --stats          per-kernel average duration
#   pass 2  --pmc FETCH_SIZE                HBM-side read requests   (own pass, MI355X_MICROARCH.md)
#   pass 3  --pmc WRITE_SIZE                HBM-side writes          (own pass)
# then scripts/summarize_profiles.py folds them into profiles/<tag>_kernel_stats.csv,
# profiles/<tag>_pmc.json and profiles/traffic.json.
# usage: profile_all.sh <round-tag> [workload ...]     e.g. profile_all.sh r01 chain3_4k chain3_4k_unfused
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
WORKLOADS="${@:-chain3_4k chain3_4k_cold chain3_4k_unfused gauss9_8k chain5_16k conv31_8k_valu conv31_8k_mfma chain3_4k_u8 chain3_8k_u8 gauss9_8k_u8 diamond_4k user_types_4k user_window_4k glsl_chain3_4k glsl_unsharp_4k glsl_fused_chain3_4k}"
export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
for wl in $WORKLOADS; do
  name="${wl%_unfused}"
  extra="--skip-workloads --no-power --no-cold"; [ "$wl" != "$name" ] && extra="$extra --no-fusion"
  case "$name" in
    chain3_4k_cold)   name=chain3_4k; extra="--cold-only --no-power" ;;   # the kernel trace of this pass holds cache-cold launches only (rotating frame slots)
    conv31_8k_valu)   name=conv31_8k; extra="$extra --conv-path 3" ;;
    conv31_8k_mfma)   name=conv31_8k; extra="$extra --conv-path 2" ;;
  esac
  case "$name" in
    chain3_4k|chain3_4k_u8|diamond_4k|user_types_4k|user_window_4k|glsl_chain3_4k|glsl_unsharp_4k|glsl_fused_chain3_4k)  steps=40; fps=8; psteps=4 ;;
    gauss9_8k|chain3_8k_u8|gauss9_8k_u8)  steps=20; fps=4; psteps=3 ;;
    chain5_16k) steps=5;  fps=2; psteps=2 ;;
    conv31_8k)  steps=5;  fps=1; psteps=2 ;;
  esac
  echo "=== $wl: kernel trace" >&2
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${wl}_kt" -o kt -- \
    python3 bench.py --workload "$name" $extra --steps $steps --warmup 2 --frames-per-step $fps --skip-cpu-baseline > "$OUT/${wl}_kt.json"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "=== $wl: pmc $ctr" >&2
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/${wl}_$ctr" -o pmc -- \
      python3 bench.py --workload "$name" $extra --steps $psteps --warmup 1 --frames-per-step 2 --skip-cpu-baseline > "$OUT/${wl}_$ctr.json"
  done
  kt=$(dirname "$(find "$OUT/${wl}_kt" -name '*kernel_stats.csv' | head -1)")
  fd=$(dirname "$(find "$OUT/${wl}_FETCH_SIZE" -name '*counter_collection.csv' | head -1)")
  wd=$(dirname "$(find "$OUT/${wl}_WRITE_SIZE" -name '*counter_collection.csv' | head -1)")
  python3 scripts/summarize_profiles.py "${TAG}_${wl}" "$wl" "$kt" "$fd" "$wd"
  mkdir -p "$ROOT/gpurun_out/profiles_new"
  cp profiles/${TAG}_${wl}_* profiles/traffic.json "$ROOT/gpurun_out/profiles_new/"
  rm -rf "$OUT/${wl}_kt" "$OUT/${wl}_FETCH_SIZE" "$OUT/${wl}_WRITE_SIZE"      # raw traces: gpurun copies back at most 64 MiB
done
