"""Exploration (GPU box): package power and clocks (rocm-smi, polled from a thread) while a walk_probe-style case runs.
usage: power_probe.py name:WxH[:texels] ...   (names: walk_probe.TEXTS)"""
import json
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

TEXTS = {"chain5": bench.CHAIN5, "chain3": bench.CHAIN3, "gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "pass": "input -> passthrough -> output",
         "sharpen": "input -> sh -> output\nsh: sharpen { amount: 0.5 }", "gauss5": "input -> blur -> output\nblur: gaussian5 { sigma: 1.0 }",
         "grade": "input -> gg -> output\ngg: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }"}
samples = []
stop = threading.Event()


def poll():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            samples.append({k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "ower", "mclk", "fclk", "socclk"))})
        except Exception as e:      # noqa: BLE001
            samples.append({"error": str(e)[:80]})
        time.sleep(0.15)


ctx = rf.Context(0)
for sp in sys.argv[1:]:
    parts = sp.split(":")
    name = parts[0]
    fmt = 1
    if name.endswith("_u8"):
        name, fmt = name[:-3], 0
    W, H = map(int, parts[1].split("x"))
    t = int(parts[2]) if len(parts) > 2 else 0
    g = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, fmt, texels_per_lane=t)
    g.fill_synthetic(1)
    g.execute(); g.wait()
    per = g.time_frames(3) / 3
    n = max(3, int(3000 / per))
    samples.clear(); stop.clear()
    th = threading.Thread(target=poll); th.start()
    ms = g.time_frames(n) / n
    stop.set(); th.join()
    print(sp, "ms/frame %.4f over %d frames" % (ms, n), flush=True)
    for s in samples[2:-1][:6]:
        print("   ", {k.replace(" clock speed:", "").replace("Current Socket Graphics Package ", ""): v for k, v in s.items() if "level" not in k}, flush=True)
    g.close()
