"""Exploration (GPU box): flag combinations (timers, hipGraph, no-fusion), 1..3 frame slots, custom conv
weights (rf_graph_set_weights), intermediate downloads and per-node timers on generated graphs.
usage: fuzz_misc.py <first seed> <count>"""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
pixel.set_threads(min(16, os.cpu_count() or 1))
ctx = rf.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if seed & 1 else util.random_graph)(rng)
    if re.search(r"input -> \w+:image", text):
        continue                               # compounding in-place heads: covered by fuzz_strips
    W, H = int(rng.randint(1, 600)), int(rng.randint(1, 400))
    fmt = (util.F32, util.U8)[(seed >> 1) & 1]
    flags = [0, rf.RF_GRAPH_TIMERS, rf.RF_GRAPH_HIPGRAPH, rf.RF_GRAPH_NO_FUSION | rf.RF_GRAPH_TIMERS, rf.RF_GRAPH_NO_FUSION | rf.RF_GRAPH_HIPGRAPH,
             rf.RF_GRAPH_TIMERS | rf.RF_GRAPH_HIPGRAPH][seed % 6]
    slots = 1 + seed % 3
    weights = {}
    for m in re.finditer(r"^(\w+): conv2d \{ ksize: (\d+)", text, re.M):
        if rng.rand() < 0.6:
            k = int(m.group(2))
            weights[m.group(1)] = rng.uniform(-0.1, 0.1, (k, k)).astype(np.float32)
    x = pixel.fill_synthetic(W, H, fmt, seed)
    try:
        want = util.run_oracle(text, x, weights)
        g = rf.Graph(ctx, rf.Config(text), W, H, fmt, num_frames=slots, flags=flags)
        for node, w in weights.items():
            g.set_weights(node, w)
        g.upload_raw(x)
        for rep in range(2):
            for s in range(slots):
                g.execute(s)
        for s in range(slots):
            g.wait(s)
            util.assert_same(g.download_raw(s), want, "slot %d" % s)
        if flags & rf.RF_GRAPH_TIMERS and not (flags & rf.RF_GRAPH_HIPGRAPH):
            ts = g.times_string(0)
            names = sorted(g.plan.launches())
            assert [p.split(":")[0].strip() for p in ts.split(", ")] == names, (ts, names)
        g.close()
    except Exception as e:
        bad += 1
        print("seed", seed, "flags", flags, "slots", slots, "fmt", fmt, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
