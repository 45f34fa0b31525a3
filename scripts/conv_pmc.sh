#!/bin/bash
# GPU box: SQ counters of the conv kernels, one rocprofv3 --pmc pass per counter group.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/conv_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  for path in ${PATHS:-2 4}; do
    RF_CONV_PATH=$path rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p${path}_g$i -o pmc -- python3 scripts/conv_pmc.py > /dev/null 2>> $OUT/log.txt
  done
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/conv_pmc/*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/conv_pmc/*/*counter_collection.csv"):
    path = f.split("/")[2].split("_")[0]
    for r in csv.DictReader(open(f)):
        if "conv2d" in r["Kernel_Name"]:
            res[path + " " + r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/conv_pmc/summary.json", "w"), indent=1)
PY
