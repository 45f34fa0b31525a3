"""Exploration (GPU box): the generic gaussian node at several radii, 4K, both formats."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util
ctx = rf.Context(0)
for fmt, fname in ((util.F32, "rgba32f"), (util.U8, "rgba8")):
    for r in (4, 7, 9, 10, 11, 12, 13, 15):
        text = "input -> gg -> output\ngg: gaussian { sigma: %.1f, radius: %d }" % (max(0.5, r / 2.5), r)
        g = rf.Graph(ctx, rf.Config(text), 3840, 2160, fmt)
        g.fill_synthetic(2); g.execute(); g.wait()
        g.time_frames(10)
        ms = g.time_frames(40) / 40
        print(json.dumps({"fmt": fname, "radius": r, "us": round(ms * 1e3, 1), "Mpx_s": round(3840 * 2160 / ms / 1e3)}), flush=True)
        g.close()
