#!/bin/bash
# GPU box: SQ counters of whatever kernels a probe program launches, one rocprofv3 --pmc pass per
# counter group.  usage: pmc.sh <tag> <kernel-name-substring> python3 <script> [args...]
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG="$1"; MATCH="$2"; shift 2
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- "$@" > /dev/null 2>> $OUT/log.txt
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
            res[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out_dir + "/g0/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ["MATCH"] in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for k in out:
    out[k]["duration_us_under_pmc"] = sum(dur[k]) / max(1, len(dur[k]))
print(json.dumps(out, indent=1))
json.dump(out, open(out_dir + "/summary.json", "w"), indent=1)
PY
