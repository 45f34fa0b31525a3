import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util
ctx = rf.Context(0)
pixel.set_threads(16)
W, H = 1920, 1080
x = pixel.fill_synthetic(W, H, util.F32, 20093)
def case(k, tail):
    return """input -> n00 -> n01 -> n02 -> output
n00: conv2d { ksize: %d, sigma: 1.65 }
n01: colour_grade { slope: 1.50, offset: 0.067, saturation: 1.15 }
n02: %s""" % (k, tail)
t2 = """input -> n01 -> n02 -> output
n01: colour_grade { slope: 1.50, offset: 0.067, saturation: 1.15 }
n02: gaussian9 { sigma: 1.35 }"""
t1 = """input -> n00 -> output
n00: conv2d { ksize: 3, sigma: 1.65 }"""
mid = util.run_hip(ctx, t1, x)
want = util.run_oracle(t2, mid)
for rpc in (None, 12, 16, 20, 24):
    tot = 0
    for rep in range(6):
        got = util.run_hip(ctx, t2, mid, rows_per_chunk=rpc)
        d = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(axis=2))
        tot += len(d)
        if len(d) and rpc in (None, 12):
            R = 12
            rows = sorted(set(d[:, 0]))
            for r in rows[:6]:
                xs = d[d[:, 0] == r][:, 1]
                print("   rpc", rpc, "row", r, "chunk", r // R, "row in chunk", r % R, "odd" if (r // R) & 1 else "even", "cols", xs.min(), "-", xs.max(), "n", len(xs), "strips", sorted(set(xs // 56))[:8])
    print("rpc", rpc, "total bad over 6 runs", tot, flush=True)
