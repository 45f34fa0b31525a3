"""Exploration (GPU box): generated graphs as row strips (over-fetch, every rank on GPU 0), through the
forced interior/boundary split, and on two frame slots in flight -- against the oracle, at 1080p.
usage: fuzz_strips.py <first seed> <count>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from oracle import graph as ograph
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
pixel.set_threads(min(16, os.cpu_count() or 1))
ctx0 = rf.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if os.environ.get("FUZZ_GEN") == "dag" else util.random_graph)(rng)
    W, H = int(rng.randint(200, 1920)), int(rng.randint(400, 1080))
    world = int(rng.randint(2, 5))
    flags = (0, rf.RF_GRAPH_NO_FUSION)[seed & 1]
    fmt = (util.F32, util.U8)[(seed >> 1) & 1]
    want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, seed))
    try:
        ghost = rf.Plan(rf.Config(text), flags).halo_schedule(False)[3]
        if ghost <= H // world:
            strips = []
            for rank in range(world):
                c = rf.Context(0, rank, world, None)
                g = rf.Graph(c, rf.Config(text), W, H, fmt, flags=flags | rf.RF_GRAPH_NO_HALO_XCHG)
                g.fill_synthetic(seed)
                g.execute(); g.wait()
                strips.append(g.download_raw())
                g.close(); c.close()
            util.assert_same(np.concatenate(strips, axis=0), want, "strips")
        os.environ["RF_FORCE_SPLIT"] = "1"
        x = pixel.fill_synthetic(W, H, fmt, seed)
        util.assert_same(util.run_hip(ctx0, text, x, flags=flags), want, "forced split")
        del os.environ["RF_FORCE_SPLIT"]
        # two slots in flight, interleaved executes
        g = rf.Graph(ctx0, rf.Config(text), W, H, fmt, num_frames=2, flags=flags)
        g.upload_raw(x)
        g.execute(0); g.execute(1)
        g.wait(0); g.wait(1)
        util.assert_same(g.download_raw(0), want, "slot 0")
        util.assert_same(g.download_raw(1), want, "slot 1")
        # a second frame on the same slots: what the oracle gives when it executes twice (a point op
        # written in place on rf:file-input grades its input again, as in the reference)
        o = ograph.GraphOracle(text, W, H, fmt)
        o.upload_raw(x); o.execute(); o.execute()
        g.execute(0); g.execute(1)
        g.wait(0); g.wait(1)
        util.assert_same(g.download_raw(0), o.download_raw(), "slot 0, second frame")
        util.assert_same(g.download_raw(1), o.download_raw(), "slot 1, second frame")
        g.close()
    except Exception as e:
        bad += 1
        os.environ.pop("RF_FORCE_SPLIT", None)
        print("seed", seed, "world", world, "flags", flags, "fmt", fmt, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
