"""GPU box: generated GLSL compute shaders (tests/glsl_gen.py, the programs of scripts/fuzz_glsl_mesa.py) run by the PRODUCT -- translated, compiled by
hiprtc for gfx950, dispatched by rf_graph_execute -- against the same translation compiled for the host with clang++ (tests/glsl_host.py; fma() one
rounding on both sides).  With fuzz_glsl_mesa.py (Mesa == host translation) this closes the triangle Mesa's GLSL compiler == translation on x86 ==
product on the GPU: correctly rounded division and square root, wrapping int arithmetic, conversions, the integer built-ins on gfx950.
usage: fuzz_glsl_gpu.py <first seed> <count> [statements]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import reforge_amd as rf  # noqa: E402
from tests import util  # noqa: E402
from tests.glsl_gen import generate  # noqa: E402
from tests.glsl_host import HostShader  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
statements = int(sys.argv[3]) if len(sys.argv) > 3 else 18
d = tempfile.mkdtemp()
rf.set_shader_path(d)
rf.set_type_lookup(True)
ctx = rf.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    text = generate(seed, statements)
    name = "gen%d" % seed
    with open(os.path.join(d, name + ".comp"), "w") as f:
        f.write(text)
    for fmt in (util.F32, util.U8):
        img = util.synthetic(150, 67, fmt, seed=seed)
        want = np.zeros_like(img)
        try:
            HostShader(name, text).run({"input_image": img, "output_image": want}, {"gain": 1.5, "shift": 3})
            got = util.run_hip(ctx, "input -> gn -> output\ngn: %s { gain: 1.5, shift: 3 }" % name, img)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("seed", seed, "fmt", fmt, "FAILED:", str(e)[:600], flush=True)
            continue
        same = got.tobytes() == want.tobytes()
        if not same:
            bad += 1
            diff = np.argwhere(got != want)
            y, x, c = diff[0]
            print("seed", seed, "fmt", fmt, "DIFF at", (x, y, c), got[y, x], want[y, x], "differing", len(diff), flush=True)
    if (seed - first) % 20 == 19:
        print("progress", seed - first + 1, "shaders,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
print("done", count, "shaders of", statements, "statements x 2 formats,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
