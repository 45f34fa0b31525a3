"""Exploration (GPU box): small frames -- per-frame time back to back, with and without hipGraph."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util
ctx = rf.Context(0)
for name, text, W, H, fmt in (("passthrough 512^2 rgba8", "input -> passthrough -> output", 512, 512, util.U8),
                              ("passthrough 512^2 rgba32f", "input -> passthrough -> output", 512, 512, util.F32),
                              ("chain3 1080p rgba32f", util.CHAIN3, 1920, 1080, util.F32),
                              ("chain3 720p rgba32f", util.CHAIN3, 1280, 720, util.F32),
                              ("chain5 1080p rgba32f unfused", util.CHAIN5, 1920, 1080, util.F32)):
    for flags, tag in ((0, "streams"), (rf.RF_GRAPH_HIPGRAPH, "hipgraph"), (rf.RF_GRAPH_NO_FUSION, "unfused"), (rf.RF_GRAPH_NO_FUSION | rf.RF_GRAPH_HIPGRAPH, "unfused+hipgraph")):
        g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags)
        g.fill_synthetic(2); g.execute(); g.wait()
        for _ in range(50):
            g.execute()
        g.wait()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            g.execute()
        g.wait()
        us = (time.perf_counter() - t0) / n * 1e6
        # latency of one frame: execute + wait
        t0 = time.perf_counter()
        for _ in range(200):
            g.execute(); g.wait()
        lat = (time.perf_counter() - t0) / 200 * 1e6
        print(json.dumps({"graph": name, "mode": tag, "us_per_frame_pipelined": round(us, 2), "us_execute_plus_wait": round(lat, 2), "launches": len(g.plan.launches())}), flush=True)
        g.close()
