"""Exploration (GPU box): extreme parameter values, node by node and in fused pairs, against the oracle
(sigma 0 / negative / huge, amount 0 / large / negative, slopes and saturations of any sign, radius 0,
ksize 1, mix outside [0,1], absent parameters), with special floats in the input for rgba32f."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util

ctx = rf.Context(0)
VALS = ["0", "0.0", "-1.5", "1000.0", "0.0001", "3", "1.0", "-0.0", "65504.0"]
NODES = []
for v in VALS:
    NODES += ["gaussian5 { sigma: %s }" % v, "gaussian9 { sigma: %s }" % v, "sharpen { amount: %s }" % v,
              "colour_grade { slope: %s, offset: %s, saturation: %s }" % (v, v, v), "conv2d { ksize: 5, sigma: %s }" % v]
NODES += ["gaussian { sigma: 2.0, radius: %s }" % r for r in ("0", "1", "15", "16", "100", "-3")]
NODES += ["conv2d { ksize: %s, sigma: 1.0 }" % k for k in ("1", "2", "4", "31", "33", "0")]
NODES += ["gaussian5 {}", "sharpen {}", "colour_grade {}", "conv2d {}", "gaussian {}", "colour_grade { slope: true }", "sharpen { amount: false }"]
bad = n = 0
rng = np.random.RandomState(1)
for i, decl in enumerate(NODES):
    for fmt in (util.F32, util.U8):
        W, H = int(rng.randint(1, 150)), int(rng.randint(1, 90))
        x = pixel.fill_synthetic(W, H, fmt, i)
        if fmt == util.F32 and W > 4 and H > 4:
            x[1, 1] = [np.inf, -np.inf, 1e30, -1e30]
            x[H // 2, W // 2] = [-0.0, 1e-45, -1e-40, 3.4e38]
        for text in ("input -> aa -> output\naa: %s" % decl, "input -> aa -> bb -> output\naa: %s\nbb: sharpen { amount: 0.3 }" % decl,
                     "input -> bb -> aa -> output\naa: %s\nbb: colour_grade { slope: 1.1, offset: 0.01, saturation: 0.9 }" % decl):
            try:
                want = util.run_oracle(text, x)
            except Exception as e:
                try:
                    util.run_hip(ctx, text, x)
                    bad += 1
                    print("oracle rejects, product accepts:", str(e)[:100], "\n" + text, flush=True)
                except rf.RfError:
                    pass
                continue
            for flags in (0, rf.RF_GRAPH_NO_FUSION):
                n += 1
                try:
                    got = util.run_hip(ctx, text, x, flags=flags)
                    same = (np.isnan(got) == np.isnan(want)).all() and (np.isnan(want) | (got.view(np.uint32) == want.view(np.uint32))).all() if fmt == util.F32 else got.tobytes() == want.tobytes()
                    if not same:
                        raise AssertionError("differs")
                except Exception as e:
                    bad += 1
                    print("fmt", fmt, "flags", flags, "%dx%d" % (W, H), str(e)[:200], "\n" + text, flush=True)
print("done", n, "runs,", bad, "failures", flush=True)
