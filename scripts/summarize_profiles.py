#!/usr/bin/env python3
"""Turns the rocprofv3 output merged back under gpurun_out/ into the small, tracked
summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   the --kernel-trace --stats table (per-kernel avg duration)
  profiles/<tag>_pmc.json           per-kernel mean FETCH_SIZE / WRITE_SIZE (KiB, raw) and the
                                    corrected HBM-side bytes per launch
  profiles/traffic.json             {workload: bytes per launch of the dominant kernel}, read by
                                    bench.py for roofline.traffic

Correction (MI355X_MICROARCH.md, "HBM"): on gfx950 FETCH_SIZE reports exactly half of the
bytes of a 16-B/lane coalesced streaming read; WRITE_SIZE is exact.  The copy_kernel
rows of the same run calibrate this: a 256 MiB copy must read and write 262144 KiB.

usage: summarize_profiles.py <tag> <workload-key> <kt_dir> <fetch_dir> <write_dir>
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    tag, key, kt, fd, wd = sys.argv[1:6]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats_src = [f for f in os.listdir(kt) if f.endswith("kernel_stats.csv")][0]
    shutil.copy(os.path.join(kt, stats_src), os.path.join(prof, tag + "_kernel_stats.csv"))
    fetch = mean_counter(os.path.join(fd, [f for f in os.listdir(fd) if f.endswith("counter_collection.csv")][0]), "FETCH_SIZE")
    write = mean_counter(os.path.join(wd, [f for f in os.listdir(wd) if f.endswith("counter_collection.csv")][0]), "WRITE_SIZE")
    out = {"unit": "KiB per launch (raw counters); bytes = corrected HBM-side bytes per launch",
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests at 64 B); write bytes = WRITE_SIZE x 1024",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out["kernels"][k] = {"FETCH_SIZE_KiB": round(f, 2), "WRITE_SIZE_KiB": round(w, 2), "launches": [nf, nw],
                             "read_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024),
                             "bytes": int(2 * f * 1024 + w * 1024)}
    with open(os.path.join(prof, tag + "_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    # dominant kernel = largest total duration in the stats table, excluding the probes
    with open(os.path.join(prof, tag + "_kernel_stats.csv")) as fh:
        rows = [r for r in csv.DictReader(fh) if "copy_kernel" not in r["Name"] and "fill" not in r["Name"] and "rocclr" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    dom = rows[0]["Name"]
    # traffic.json = {"kernel_sources_sha16": identity of the kernel sources measured, "recorded": {workload: bytes}};
    # figures recorded on other sources are dropped (bench.py refuses them anyway)
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernel_sources_sha16()
    tpath = os.path.join(prof, "traffic.json")
    traffic = {"kernel_sources_sha16": sha, "recorded": {}}
    if os.path.exists(tpath):
        with open(tpath) as fh:
            old = json.load(fh)
        if old.get("kernel_sources_sha16") == sha:
            traffic["recorded"] = old.get("recorded", {})
    match = [k for k in out["kernels"] if k.split("(")[0] == dom.split("(")[0]]
    traffic["recorded"][key] = out["kernels"][match[0]]["bytes"] if match else None
    with open(tpath, "w") as fh:
        json.dump(traffic, fh, indent=1, sort_keys=True)
    print("dominant:", dom[:100], "avg ns", rows[0]["AverageNs"], "traffic", traffic["recorded"][key])


if __name__ == "__main__":
    main()
