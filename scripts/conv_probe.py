"""Exploration (GPU box): conv2d paths, correctness vs oracle on a small frame + timing at 8K."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from tests import util

ctx = rf.Context(0)
for K in (3, 5, 9, 15, 31):
    text = "input -> conv2d -> output\nconv2d: conv2d { ksize: %d, sigma: %.1f }" % (K, K / 6.0)
    for fmt in (util.F32, util.U8):
        for (W, H) in ((97, 50), (64, 8), (130, 70)):
            x = util.synthetic(W, H, fmt)
            want = util.run_oracle(text, x)
            for path in ("1", "2", "3"):
                os.environ["RF_CONV_PATH"] = path
                got = util.run_hip(ctx, text, x)
                ok = got.tobytes() == want.tobytes()
                if not ok:
                    d = np.argwhere(got != want)
                    print("MISMATCH K=%d fmt=%d %dx%d path=%s n=%d first=%s got=%s want=%s" % (K, fmt, W, H, path, len(d), d[0], got[tuple(d[0])], want[tuple(d[0])]), flush=True)
    print("K=%d parity checked" % K, flush=True)
text = "input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }"
for path in ("1", "2", "3"):
    os.environ["RF_CONV_PATH"] = path
    for (W, H) in ((1920, 1080), (7680, 4320)):
        g = rf.Graph(ctx, rf.Config(text), W, H, util.F32)
        g.fill_synthetic(5)
        g.execute(); g.wait()
        ms = g.time_frames(3) / 3
        print(json.dumps({"path": path, "W": W, "H": H, "ms": ms, "Mpx_s": W * H / ms / 1e3, "TFLOPs": 2 * 961 * 4 * W * H / ms / 1e9}), flush=True)
        g.close()

for K in (3, 7, 9, 15):
    text = "input -> conv2d -> output\nconv2d: conv2d { ksize: %d, sigma: %.1f }" % (K, K / 6.0)
    for path in ("1", "2", "3"):
        os.environ["RF_CONV_PATH"] = path
        g = rf.Graph(ctx, rf.Config(text), 3840, 2160, util.F32)
        g.fill_synthetic(5)
        g.execute(); g.wait()
        ms = g.time_frames(5) / 5
        print(json.dumps({"K": K, "path": path, "ms_4k": ms}), flush=True)
        g.close()
