"""Exploration (GPU box): rgba8 timings of the 4K chain, fused and unfused."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util
ctx = rf.Context(0)
for fmt, name in ((util.U8, "rgba8"), (util.F32, "rgba32f")):
    for flags, tag in ((0, "fused"), (rf.RF_GRAPH_NO_FUSION, "unfused")):
        g = rf.Graph(ctx, rf.Config(util.CHAIN3), 3840, 2160, fmt, flags=flags)
        g.fill_synthetic(2); g.execute(); g.wait()
        g.time_frames(20)
        ms = g.time_frames(100) / 100
        print(json.dumps({"fmt": name, "mode": tag, "ms_frame": round(ms, 4), "Mpx_s": round(3840 * 2160 / ms / 1e3), "launches": [(l, round(t, 4)) for l, t in g.time_launches(50)]}), flush=True)
        g.close()
