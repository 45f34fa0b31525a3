"""Exploration (GPU box): the C++ CLI end to end on generated configs: PNG in -> reforge -> raw RGBA8 out,
against the oracle's sRGB path.  usage: fuzz_cli.py <first seed> <count>"""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import graph as og, pixel
from tests import util
from tests.test_cli_host import write_png, CLI

first, count = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if seed & 1 else util.random_graph)(rng)
    W, H = int(rng.randint(1, 300)), int(rng.randint(1, 200))
    fmt_name, fmt = (("rgba32f", util.F32), ("rgba8", util.U8))[(seed >> 1) & 1]
    rgba = pixel.fill_synthetic(W, H, util.U8, seed)
    rgba[..., 3] = 255
    src, dst, cfg = os.path.join(d, "in.png"), os.path.join(d, "out.rgba"), os.path.join(d, "g.cfg")
    write_png(src, rgba, lambda y: int(rng.randint(0, 5)))
    open(cfg, "w").write(text)
    extra = ["--no-fusion"] if (seed >> 2) & 1 else []
    r = subprocess.run([CLI, "-i", src, "--config", cfg, "-o", dst, "--shader-format", fmt_name] + extra, capture_output=True, text=True)
    try:
        assert r.returncode == 0, r.stderr[-300:]
        ref = og.GraphOracle(text, W, H, fmt)
        ref.upload_srgb8(rgba)
        ref.execute()
        got = np.fromfile(dst, np.uint8).reshape(H, W, 4)
        assert got.tobytes() == ref.download_srgb8().tobytes(), "output differs"
    except Exception as e:
        bad += 1
        print("seed", seed, fmt_name, extra, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
