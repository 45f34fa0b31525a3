"""Exploration (GPU box): shader clock and package power WHILE a workload runs (rocm-smi polled from a thread), to
tell an issue-bound kernel from a power-capped one.  usage: clock_probe.py [workload ...]"""
import json
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

samples = []
stop = threading.Event()


def poll():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            samples.append({k: v for k, v in card.items() if "sclk" in k.lower() or "ower" in k or "mclk" in k.lower() or "fclk" in k.lower()})
        except Exception as e:      # noqa: BLE001
            samples.append({"error": str(e)[:80]})
        time.sleep(0.2)


ctx = rf.Context(0)
names = sys.argv[1:] or ["chain5_16k", "gauss9_8k", "chain3_4k", "conv31_8k"]
for name in names:
    name, _, texels = name.partition(":")             # workload[:texels per lane]
    wl = bench.WORKLOADS[name]
    g = rf.Graph(ctx, rf.Config(wl["text"]), wl["W"], wl["H"], wl["fmt"], texels_per_lane=int(texels or 0))
    g.fill_synthetic(wl["seed"])
    g.execute(); g.wait()
    per = g.time_frames(3) / 3
    n = max(3, int(4000 / per))
    samples.clear()
    stop.clear()
    th = threading.Thread(target=poll)
    th.start()
    ms = g.time_frames(n) / n
    stop.set()
    th.join()
    print(name, "texels_per_lane", texels or "auto", "ms/frame %.4f over %d frames" % (ms, n))
    for s in samples[1:-1][:12]:
        print("   ", {k.replace(" clock speed:", "").replace("Current Socket Graphics Package ", ""): v for k, v in s.items() if "level" not in k})
    g.close()
time.sleep(1.0)
print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout[:600])
