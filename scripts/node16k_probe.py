"""Exploration (GPU box): what the stream-kernel STRUCTURE costs on a 16384^2 rgba32f frame, node by node:
passthrough (no arithmetic) up to the fused 5-stage chain, one and two texels per lane, several chunk heights."""
import sys

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

ctx = rf.Context(0)
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
G5 = "input -> blur -> output\nblur: gaussian5 { sigma: 1.0 }"
G9 = "input -> blur -> output\nblur: gaussian9 { sigma: 2.0 }"
GR = "input -> gg -> output\ngg: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }"
SH = "input -> sh -> output\nsh: sharpen { amount: 0.5 }"
PT = "input -> passthrough -> output"
texts = [("passthrough", PT), ("grade", GR), ("sharpen", SH), ("gauss5", G5), ("gauss9", G9), ("chain3", bench.CHAIN3), ("chain5", bench.CHAIN5)]
print("copy_gbps 1GiB:", ctx.copy_bandwidth(1 << 30, 10), flush=True)
for name, text in texts:
    line = []
    for t in (1, 2):
        for rpc in (0, 32, 64, 128, 256):
            g = rf.Graph(ctx, rf.Config(text), W, H, 1, texels_per_lane=t, rows_per_chunk=rpc)
            g.fill_synthetic(1)
            g.execute(); g.wait()
            ms = min(g.time_frames(8) / 8 for _ in range(3))
            line.append("T%d/%d:%.3f" % (t, rpc, ms))
            g.close()
    print(name, " ".join(line), flush=True)
