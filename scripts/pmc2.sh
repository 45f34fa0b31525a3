#!/bin/bash
# GPU box: counters of the kernels a probe program launches, one rocprofv3 --pmc pass per counter group (PMC_GROUPS: groups
# separated by ';').  Run from the tree whose library is to be profiled.  usage: pmc2.sh <tag> <kernel-name-substring> python3 <script> [args...]
set -e
export TMPDIR=/tmp
TAG="$1"; MATCH="$2"; shift 2
OUT=${PMC_OUT:-gpurun_out}/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
IFS=';' read -ra GROUPS_ <<< "$PMC_GROUPS"
i=0
for grp in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- "$@" > /dev/null 2>> $OUT/log.txt || echo "group $i failed: $grp"
  i=$((i+1))
done
MATCH="$MATCH" OUT="$OUT" python3 - <<'PY'
import csv, glob, collections, json, os
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
out_dir = os.environ["OUT"]
for f in glob.glob(out_dir + "/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ["MATCH"] in r["Kernel_Name"]:
            res[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out_dir + "/g0/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ["MATCH"] in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for k in out:
    out[k]["duration_us_under_pmc"] = sum(dur[k]) / max(1, len(dur[k]))
json.dump(out, open(out_dir + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/g*
