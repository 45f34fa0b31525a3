"""Exploration (GPU box): generated graphs through the sRGB boundary (upload_srgb8 -> graph ->
download_srgb8) against the oracle.  usage: fuzz_srgb.py <first seed> <count>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from oracle import graph as og
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
pixel.set_threads(min(16, os.cpu_count() or 1))
ctx = rf.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if os.environ.get("FUZZ_GEN") == "dag" else util.random_graph)(rng)
    W, H = int(rng.randint(1, 900)), int(rng.randint(1, 500))
    fmt = (util.F32, util.U8)[seed & 1]
    flags = (0, rf.RF_GRAPH_NO_FUSION)[(seed >> 1) & 1]
    rgba = pixel.fill_synthetic(W, H, util.U8, seed)
    try:
        ref = og.GraphOracle(text, W, H, fmt)
        ref.upload_srgb8(rgba)
        ref.execute()
        want = ref.download_srgb8()
        g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags)
        g.upload_srgb8(rgba)
        g.execute(); g.wait()
        got = g.download_srgb8()
        g.close()
        if got.tobytes() != want.tobytes():
            d = np.argwhere(got != want)
            raise AssertionError("%d bytes differ, first %s got %d want %d" % (len(d), d[0], got[tuple(d[0])], want[tuple(d[0])]))
    except Exception as e:
        bad += 1
        print("seed", seed, "flags", flags, "fmt", fmt, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
