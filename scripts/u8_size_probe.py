"""Exploration (GPU box): rgba8 graphs at several frame sizes (Mpx/s)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util
ctx = rf.Context(0)
G9 = "input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }"
G5 = "input -> gaussian5 -> output\ngaussian5: gaussian5 { sigma: 1.0 }"
SH = "input -> sharpen -> output\nsharpen: sharpen { amount: 0.5 }"
for name, text in (("chain3", util.CHAIN3), ("gauss5", G5), ("gauss9", G9), ("sharpen", SH), ("passthrough", "input -> passthrough -> output")):
    for W, H in ((1920, 1080), (3840, 2160), (7680, 4320), (16384, 8192)):
        g = rf.Graph(ctx, rf.Config(text), W, H, util.U8)
        g.fill_synthetic(2); g.execute(); g.wait()
        g.time_frames(5)
        n = 50 if W < 8000 else 10
        ms = g.time_frames(n) / n
        print(json.dumps({"graph": name, "W": W, "H": H, "us_frame": round(ms * 1e3, 2), "Mpx_s": round(W * H / ms / 1e3)}), flush=True)
        g.close()
