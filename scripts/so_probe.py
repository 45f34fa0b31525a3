"""Exploration (GPU box): time a few workloads on an alternative build of librfhip.so (argv[1]); prints min/median ms of 5 runs.
usage: so_probe.py <path/to/librfhip.so> [name:WxH:T:rpc ...]"""
import sys

sys.path.insert(0, ".")
import reforge_amd._lib as L

L.SO_PATH = sys.argv[1]
import bench
import reforge_amd as rf

ctx = rf.Context(0)
TEXTS = {"chain5": bench.CHAIN5, "chain3": bench.CHAIN3, "gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "pass": "input -> passthrough -> output",
         "diamond": bench.DIAMOND}
specs = sys.argv[2:] or ["chain5:16384x16384:1:0", "chain5:16384x16384:2:0", "chain3:16384x16384:1:0", "gauss9:7680x4320:1:0", "chain3:3840x2160:1:0", "pass:16384x16384:1:0"]
out = []
for sp in specs:
    name, dims, t, rpc = sp.split(":")
    W, H = map(int, dims.split("x"))
    g = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, 1, texels_per_lane=int(t), rows_per_chunk=int(rpc))
    g.fill_synthetic(1)
    g.execute(); g.wait()
    n = max(4, int(30 / max(g.time_frames(2) / 2, 0.02)))
    ms = sorted(g.time_frames(n) / n for _ in range(5))
    out.append("%s %.4f/%.4f" % (sp, ms[0], ms[2]))
    g.close()
print(sys.argv[1].split("/")[-1], " | ".join(out), flush=True)
