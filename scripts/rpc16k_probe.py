"""Exploration (GPU box): chunk height and walk direction of the fused 5-stage chain on 16384^2."""
import sys
sys.path.insert(0, ".")
import bench
import reforge_amd as rf
ctx = rf.Context(0)
W = H = 16384
for name, text in (("chain5", bench.CHAIN5), ("chain3", bench.CHAIN3)):
    for t in (1, 2):
        line = []
        for ex in (0, rf.RF_EXEC_NO_ALTERNATE):
            for rpc in (128, 256, 512, 1024, 2048):
                g = rf.Graph(ctx, rf.Config(text), W, H, 1, texels_per_lane=t, rows_per_chunk=rpc, exec_flags=ex)
                g.fill_synthetic(1)
                g.execute(); g.wait()
                ms = sorted(g.time_frames(8) / 8 for _ in range(3))
                line.append("%s%d:%.3f" % ("fwd" if ex else "alt", rpc, ms[0]))
                g.close()
        print(name, "T=%d" % t, " ".join(line), flush=True)
