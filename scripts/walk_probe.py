"""Exploration (GPU box): walk direction (alternating vs all top-down), texels per lane and chunk height.
usage: walk_probe.py name:WxH[:rpc,rpc,...] ..."""
import os
import sys
sys.path.insert(0, ".")
import bench
import reforge_amd as rf
ctx = rf.Context(0)
TEXTS = {"chain5": bench.CHAIN5, "chain3": bench.CHAIN3, "gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "pass": "input -> passthrough -> output",
         "sharpen": "input -> sh -> output\nsh: sharpen { amount: 0.5 }", "gauss5": "input -> blur -> output\nblur: gaussian5 { sigma: 1.0 }"}
for r in range(16):
    TEXTS["gr%d" % r] = "input -> gg -> output\ngg: gaussian { sigma: 2.0, radius: %d }" % r
TEXTS["diamond"] = bench.WORKLOADS["diamond_4k"]["text"]
TEXTS["grade"] = "input -> gg -> output\ngg: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }"
for sp in sys.argv[1:]:
    parts = sp.split(":")
    name, dims = parts[0], parts[1]
    fmt = 1
    if name.endswith("_u8"):
        name, fmt = name[:-3], 0
    rpcs = [int(x) for x in parts[2].split(",")] if len(parts) > 2 else [0]
    W, H = map(int, dims.split("x"))
    for t in [int(x) for x in os.environ.get("WALK_T", "1,2").split(",")]:
        line = []
        exs = {"alt": (rf.RF_EXEC_ALTERNATE,), "fwd": (rf.RF_EXEC_NO_ALTERNATE,), "auto": (0,)}.get(os.environ.get("WALK_EX", ""), (rf.RF_EXEC_ALTERNATE, rf.RF_EXEC_NO_ALTERNATE))
        for ex in exs:
            for rpc in rpcs:
                g = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, fmt, texels_per_lane=t, rows_per_chunk=rpc, exec_flags=ex | int(os.environ.get("WALK_FLAGS", "0"), 0))
                g.fill_synthetic(1)
                g.execute(); g.wait()
                n = max(4, int(20 / max(g.time_frames(2) / 2, 0.02)))
                ms = sorted(g.time_frames(n) / n for _ in range(5))
                line.append("%s%d:%.4f" % ("fwd" if ex == rf.RF_EXEC_NO_ALTERNATE else ("alt" if ex else "auto"), rpc, ms[0]))
                g.close()
        print(name, "u8" if fmt == 0 else "f32", dims, "T=%d" % t, " ".join(line), "| ns/px %.5f" % (1e6 * min(float(x.split(":")[1]) for x in line) / (W * H)), flush=True)
