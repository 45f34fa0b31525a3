"""CPU (no GPU needed): generated GLSL compute shaders (tests/glsl_gen.py) through Mesa's GLSL compiler + llvmpipe and through rf_glsl.cpp's
translation compiled for the host, compared bit for bit.  usage: fuzz_glsl_mesa.py <first seed> <count> [statements per shader]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from tests.mesa_glsl import MesaShader, MesaCompileError, runner
from tests.glsl_host import HostShader
from tests.glsl_gen import generate
runner()
first, count = int(sys.argv[1]), int(sys.argv[2])
statements = int(sys.argv[3]) if len(sys.argv) > 3 else 14
imgs = [util.synthetic(37, 23, util.F32), util.synthetic(37, 23, util.U8)]      # rgba32f, and rgba8 (the UNORM8 load as llvmpipe's: split_fma)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    text = generate(seed, statements)
    for img in imgs:
        try:
            bm, bo = np.full(10, 7, np.uint32), np.full(10, 7, np.uint32)      # the Stats block (atomics): starts at 7s on both sides
            m = MesaShader("gen", text).run({"input_image": img, "output_image": np.zeros_like(img)}, {"gain": 1.5, "shift": 3}, {"Stats": bm})["output_image"]
        except MesaCompileError as e:
            bad += 1; print("seed", seed, "MESA REJECTS:", str(e)[-400:]); break
        try:
            o = np.zeros_like(img)
            HostShader("gen", text, split_fma=True).run({"input_image": img, "output_image": o}, {"gain": 1.5, "shift": 3}, {"Stats": bo.view(np.uint8)})
        except Exception as e:
            bad += 1; i = str(e).find("error:"); print("seed", seed, "TRANSLATION FAILS:", str(e)[max(i, 0):max(i, 0) + 500]); break
        if not np.array_equal(bm, bo):
            bad += 1
            print("seed", seed, img.dtype, "the Stats block differs:", bm, bo)
        elif m.tobytes() != o.tobytes():
            bad += 1
            diff = np.argwhere(m != o)
            y, x, ch = diff[0]
            print("seed", seed, img.dtype, "DIFF at", (x, y, ch), m[y, x], o[y, x], "differing", len(diff))
    if (seed - first) % 50 == 49:
        print("progress", seed - first + 1, "shaders,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
print("done", count, "shaders of", statements, "statements x 2 formats,", bad, "bad, %.0f s" % (time.time() - t0))
