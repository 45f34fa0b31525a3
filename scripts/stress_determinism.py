"""Exploration (GPU box): the same graph many times on a busy chip -- every run must give the same
bits (a race shows up as run-to-run differences even where no oracle is affordable).
usage: stress_determinism.py <first seed> <count> [reps]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = rf.Context(0)
bad = 0
t0 = time.time()
SIZES = [(3840, 2160), (1920, 1080), (7680, 4320), (2560, 1440)]
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if seed & 1 else util.random_graph)(rng)
    W, H = SIZES[seed % len(SIZES)]
    fmt = (util.F32, util.U8)[(seed >> 1) & 1]
    flags = (0, rf.RF_GRAPH_NO_FUSION)[(seed >> 2) & 1]
    g = rf.Graph(ctx, rf.Config(text), W, H, fmt, num_frames=2, flags=flags)
    sums = set()
    for r in range(reps):
        g.fill_synthetic(seed)             # fresh input every time (in-place heads modify it)
        g.execute(r & 1)
        g.execute(1 - (r & 1))             # the other slot keeps the chip busy
        g.wait(0); g.wait(1)
        sums.add(zlib.crc32(g.download_raw(r & 1).tobytes()))
    g.close()
    if len(sums) != 1:
        bad += 1
        print("seed", seed, "fmt", fmt, "flags", flags, "%dx%d" % (W, H), len(sums), "different results in", reps, "runs\n" + text, flush=True)
print("done", count, "graphs x", reps, "runs,", bad, "nondeterministic, %.0f s" % (time.time() - t0), flush=True)
