"""Exploration (GPU box): rgba8 4K chain vs rows per chunk (waves in flight vs halo re-read)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util
ctx = rf.Context(0)
for name, text in (("chain3", util.CHAIN3), ("passthrough", "input -> passthrough -> output"), ("grade", "input -> grade -> output\ngrade: grade { slope: 1.3, offset: -0.1, saturation: 0.6 }")):
    for rpc in (0, 8, 12, 16, 24, 32, 64):
        os.environ["RF_ROWS_PER_CHUNK"] = str(rpc)
        g = rf.Graph(ctx, rf.Config(text), 3840, 2160, util.U8)
        g.fill_synthetic(2); g.execute(); g.wait()
        g.time_frames(20)
        ms = g.time_frames(100) / 100
        print(json.dumps({"graph": name, "rpc": rpc, "us_frame": round(ms * 1e3, 2), "Mpx_s": round(3840 * 2160 / ms / 1e3)}), flush=True)
        g.close()
