"""Exploration script (GPU box): copy bandwidth, per-launch times of the BASELINE chains."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf

CHAIN3 = """
input -> blur -> grade -> sharp -> output
blur:  gaussian5    { sigma: 1.0 }
grade: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp: sharpen      { amount: 0.5 }
"""

def run(ctx, text, W, H, fmt, flags, iters=50, label=""):
    g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags)
    g.fill_synthetic(0x5EED0002)
    g.execute(); g.wait()
    for _ in range(3):
        g.time_frames(5)
    ms = g.time_frames(iters) / iters
    launches = g.plan.launches()
    per = [g.time_launch(i, iters) for i in range(len(launches))]
    px = W * H
    bpp = 16 if fmt == rf.RF_FORMAT_RGBA32F else 4
    out = {"label": label, "W": W, "H": H, "ms_frame": ms, "Mpx_s": px / ms / 1e3,
           "launches": {l: {"ms": t, "GBs_alg": 2 * bpp * px / t / 1e6} for l, t in zip(launches, per)}}
    print(json.dumps(out), flush=True)
    g.close()
    return out

if __name__ == "__main__":
    ctx = rf.Context(0)
    print("arch", ctx.arch, flush=True)
    for nb in (256 << 20, 1 << 30):
        print("copy GB/s", nb >> 20, "MiB:", ctx.copy_bandwidth(nb, 20), flush=True)
    F = rf.RF_FORMAT_RGBA32F
    rpcs = os.environ.get("PROBE_RPC", "0").split(",")
    pfs = os.environ.get("PROBE_PF", "0").split(",")
    for pf in pfs:
        os.environ["RF_PREFETCH_ROWS"] = pf
        for rpc in rpcs:
            os.environ["RF_ROWS_PER_CHUNK"] = rpc
            if os.environ.get("PROBE_UNFUSED", "1") == "1":
                run(ctx, CHAIN3, 3840, 2160, F, rf.RF_GRAPH_NO_FUSION, label="4K unfused pf=%s rpc=%s" % (pf, rpc))
            run(ctx, CHAIN3, 3840, 2160, F, 0, label="4K fused pf=%s rpc=%s" % (pf, rpc))
    os.environ["RF_ROWS_PER_CHUNK"] = "0"
    os.environ["RF_PREFETCH_ROWS"] = "0"
    if os.environ.get("PROBE_REST", "1") != "1":
        sys.exit(0)
    run(ctx, "input -> passthrough -> output", 3840, 2160, F, 0, label="4K passthrough")
    run(ctx, "input -> gaussian9 -> output\ngaussian9: gaussian9 {sigma: 2.0}", 7680, 4320, F, 0, label="8K gaussian9")
    run(ctx, CHAIN3, 7680, 4320, F, 0, label="8K fused3")
    run(ctx, "input -> conv2d -> output\nconv2d: conv2d {ksize: 31, sigma: 5.0}", 1920, 1080, F, 0, iters=3, label="1080p conv31")
