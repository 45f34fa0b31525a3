"""Exploration (GPU box): frame time vs rows per chunk, several graphs and frame sizes (rgba32f)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util
ctx = rf.Context(0)
G9 = "input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }"
fmt = util.U8 if os.environ.get("SWEEP_FMT") == "u8" else util.F32
for name, text in (("chain3", util.CHAIN3), ("chain5", util.CHAIN5), ("gauss9", G9), ("sharpen", "input -> sharpen -> output\nsharpen: sharpen { amount: 0.5 }"), ("passthrough", "input -> passthrough -> output")):
    for W, H in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)):
        row = {}
        for rpc in (0, 8, 12, 16, 24, 32, 48, 64, 96, 128):
            os.environ["RF_ROWS_PER_CHUNK"] = str(rpc)
            g = rf.Graph(ctx, rf.Config(text), W, H, fmt)
            g.fill_synthetic(2); g.execute(); g.wait()
            g.time_frames(10)
            n = 60 if W < 7000 else 15
            row[rpc] = round(g.time_frames(n) / n * 1e3, 1)
            g.close()
        best = min((v, k) for k, v in row.items() if k)[1]
        print(json.dumps({"graph": name, "W": W, "H": H, "us_by_rpc": row, "best": best, "default_vs_best": round(row[0] / row[best], 2)}), flush=True)
