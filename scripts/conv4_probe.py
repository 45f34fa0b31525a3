"""Exploration (GPU box): the VALU convolution with FOUR waves per SIMD (4 columns per lane, 16-wave workgroups; RF_CONV_PATH=4)
against the default shape (8 columns per lane, 8 waves): parity on small frames, then time, power and clock at 31x31 8K."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import reforge_amd as rf
from tests import util

ctx = rf.Context(0)
for K in (15, 21, 31):
    text = "input -> conv2d -> output\nconv2d: conv2d { ksize: %d, sigma: %.1f }" % (K, K / 6.0)
    for fmt in (util.F32, util.U8):
        for (W, H) in ((97, 50), (130, 70), (300, 131)):
            x = util.synthetic(W, H, fmt)
            want = util.run_oracle(text, x)
            for path in (3, 4):
                got = util.run_hip(ctx, text, x, conv_path=path)
                if got.tobytes() != want.tobytes():
                    d = np.argwhere(got != want)
                    print("MISMATCH K=%d fmt=%d %dx%d path=%d n=%d first=%s" % (K, fmt, W, H, path, len(d), d[0]), flush=True)
    print("K=%d parity checked" % K, flush=True)
text = "input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }"
for rnd in (1, 2):
    for path in (3, 4):
        g = rf.Graph(ctx, rf.Config(text), 7680, 4320, util.F32, conv_path=path)
        g.fill_synthetic(5)
        g.execute(); g.wait()
        ms = min(g.time_frames(40) / 40 for _ in range(3))
        pw = bench.power_under_load(g, ms)
        print(json.dumps({"round": rnd, "path": path, "ms": round(ms, 4), "TFLOPs": round(2 * 961 * 4 * 7680 * 4320 / ms / 1e9, 1), "power": pw}), flush=True)
        g.close()
