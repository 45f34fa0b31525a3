"""Exploration (GPU box): parameter edits between frames (rf_graph_set_param) against the oracle
run on the edited config.  usage: fuzz_params.py <first seed> <count>"""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
pixel.set_threads(min(16, os.cpu_count() or 1))
ctx = rf.Context(0)
EDITS = {"colour_grade": ["slope", "offset", "saturation"], "grade": ["slope", "offset", "saturation"], "colour-grade": ["slope", "offset", "saturation"], "sharpen": ["amount"], "gaussian5": ["sigma"], "gaussian9": ["sigma"],
         "gaussian": ["sigma"], "conv2d": ["sigma"], "combination": ["mix"]}
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if os.environ.get("FUZZ_GEN") == "dag" else util.random_graph)(rng)
    W, H = int(rng.randint(50, 700)), int(rng.randint(50, 400))
    fmt = (util.F32, util.U8)[seed & 1]
    flags = (0, rf.RF_GRAPH_NO_FUSION, rf.RF_GRAPH_HIPGRAPH)[seed % 3]
    x = pixel.fill_synthetic(W, H, fmt, seed)
    try:
        g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags)
        g.upload_raw(x)
        g.execute(); g.wait()
        util.assert_same(g.download_raw(), util.run_oracle(text, x), "before the edit")
        lines = text.split("\n")
        cur = text
        for _ in range(3):
            idx = [i for i, l in enumerate(lines) if re.match(r"^\w+: ([\w-]+) \{ ", l)]
            if not idx:
                break
            i = idx[rng.randint(len(idx))]
            m = re.match(r"^(\w+): ([\w-]+) \{ (.*) \}$", lines[i])
            node, typ, body = m.group(1), m.group(2), m.group(3)
            name = EDITS[typ][rng.randint(len(EDITS[typ]))]
            val = round(float(rng.uniform(0.1, 2.0)), 2)
            kv = dict(p.split(": ") for p in body.split(", "))
            kv[name] = "%.2f" % val
            lines[i] = "%s: %s { %s }" % (node, typ, ", ".join("%s: %s" % p for p in kv.items()))
            cur = "\n".join(lines)
            g.set_param(node, name, float(np.float32(float("%.2f" % val))))
            if "image" in text.split("\n")[0] and text.startswith("input -> n00:image"):
                g.upload_raw(x)      # an in-place head grades its input again every frame: start from the same texels
            g.execute(); g.wait()
            want = util.run_oracle(cur, x)
            util.assert_same(g.download_raw(), want, "after %s.%s = %s" % (node, name, val))
        g.close()
    except Exception as e:
        bad += 1
        print("seed", seed, "flags", flags, "fmt", fmt, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
