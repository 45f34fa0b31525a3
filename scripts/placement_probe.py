"""Exploration (GPU box): the SAME launch timed on freshly allocated images, again and again in one process -- how much of the
frame time depends on where hipMalloc put the images.  usage: placement_probe.py name:WxH [repeats] ; PLACE_KEEP=1 keeps every
graph alive (new addresses each time), default frees each graph before the next is made (the allocator may hand the blocks back)."""
import os
import sys

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

ctx = rf.Context(0)
TEXTS = {"chain5": bench.CHAIN5, "chain3": bench.CHAIN3, "gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "pass": "input -> passthrough -> output"}
name, dims = sys.argv[1].split(":")
W, H = map(int, dims.split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
keep = []
times = []
for r in range(reps):
    g = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, rf.RF_FORMAT_RGBA32F)
    g.fill_synthetic(1)
    g.execute()
    g.wait()
    n = max(4, int(30 / max(g.time_frames(2) / 2, 0.02)))
    ms = sorted(g.time_frames(n) / n for _ in range(3))
    times.append(ms[0])
    print("%s %s graph %d: best %.4f ms  median %.4f" % (name, dims, r, ms[0], ms[1]), flush=True)
    if os.environ.get("PLACE_KEEP"):
        keep.append(g)
    else:
        g.close()
print("spread: min %.4f max %.4f (%.1f %%)" % (min(times), max(times), 100.0 * (max(times) / min(times) - 1.0)), flush=True)
