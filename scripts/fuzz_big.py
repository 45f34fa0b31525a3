"""Exploration (GPU box): generated graphs on LARGE frames, fused vs one-launch-per-node vs the oracle
on a few bands.  usage: fuzz_big.py <first seed> <count>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = rf.Context(0)
bad = 0
t0 = time.time()
SIZES = [(3840, 2160), (7680, 4320), (5000, 3000), (16384, 1500), (1000, 9000), (2731, 4099)]
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if seed & 1 else util.random_graph)(rng)
    W, H = SIZES[seed % len(SIZES)]
    fmt = (util.F32, util.U8)[(seed >> 1) & 1]
    try:
        outs = []
        for flags in (0, rf.RF_GRAPH_NO_FUSION):
            g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags)
            g.fill_synthetic(seed)
            g.execute(); g.wait()
            outs.append(g.download_raw())
            g.close()
        if not np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8)):
            d = np.argwhere((outs[0].view(np.uint8) != outs[1].view(np.uint8)).reshape(H, -1).any(axis=1))
            raise AssertionError("fused != unfused in %d rows, first %d" % (len(d), d[0][0]))
    except Exception as e:
        bad += 1
        print("seed", seed, "fmt", fmt, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
