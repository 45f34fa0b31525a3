"""Exploration (GPU box): parity of the default conv2d path on small frames (K = 7..31, both formats) and its time, power and clock
at 31x31 7680x4320 -- run in each of several builds of the tree (scripts/mk_variant.sh) to A/B a change of the conv kernels.
usage: conv_variant_probe.py [rounds]"""
import json, os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
import reforge_amd as rf
from tests import util

ctx = rf.Context(0)
bad = 0
for K in (7, 9, 15, 21, 31):
    text = "input -> conv2d -> output\nconv2d: conv2d { ksize: %d, sigma: %.1f }" % (K, K / 6.0)
    for fmt in (util.F32, util.U8):
        for (W, H) in ((97, 50), (130, 70), (300, 131)):
            x = util.synthetic(W, H, fmt)
            if util.run_hip(ctx, text, x).tobytes() != util.run_oracle(text, x).tobytes():
                bad += 1
                print("MISMATCH K=%d fmt=%d %dx%d" % (K, fmt, W, H), flush=True)
print("parity: %d mismatches" % bad, flush=True)
for K in (7, 9, 15, 21):
    g = rf.Graph(ctx, rf.Config("input -> conv2d -> output\nconv2d: conv2d { ksize: %d, sigma: %.1f }" % (K, K / 6.0)), 3840, 2160, util.F32)
    g.fill_synthetic(5)
    g.execute(); g.wait()
    print("K=%d 3840x2160: %.4f ms" % (K, min(g.time_frames(100) / 100 for _ in range(3))), flush=True)
    g.close()
text = "input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }"
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    g = rf.Graph(ctx, rf.Config(text), 7680, 4320, util.F32)
    g.fill_synthetic(5)
    g.execute(); g.wait()
    ms = min(g.time_frames(40) / 40 for _ in range(3))
    pw = bench.power_under_load(g, ms)
    print(json.dumps({"ms": round(ms, 4), "TFLOPs": round(2 * 961 * 4 * 7680 * 4320 / ms / 1e9, 1), "power": pw}), flush=True)
    g.close()
