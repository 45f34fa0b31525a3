"""GPU box: generated GLSL compute shaders that EXCHANGE their results through workgroup-shared memory (tests/glsl_gen.py's programs + a 16 x 16 tile in
`shared`, barrier(), a read of another invocation's value) -- run by the product on the GPU (the file's own workgroups: rf_glsl_dev.h's GROUPED dispatch) and
by Mesa's GLSL compiler + llvmpipe on the box's CPU (tests/mesa_glsl.py), compared bit for bit on rgba32f.  (The host harness cannot run shared memory; the
programs hold no fma(), every float is precise: neither side contracts, so the bits must agree.)
usage: fuzz_glsl_gpu_mesa.py <first seed> <count> [statements]"""
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
from tests.mesa_glsl import MesaShader, runner, why_not  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
statements = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if not runner():
    sys.exit("Mesa is not usable here: " + why_not())
d = tempfile.mkdtemp()
rf.set_shader_path(d)
rf.set_type_lookup(True)
ctx = rf.Context(0)
W, H = 160, 64      # whole workgroups: no invocation leaves before the barrier
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    text = generate(seed, statements)
    text = text.replace("void main()\n{", "shared vec4 tile[16][16];\nvoid main()\n{")
    text = text.replace("    imageStore(output_image, p, o);", "    uvec2 l = gl_LocalInvocationID.xy;\n    tile[l.y][l.x] = o;\n    barrier();\n"
                        "    precise vec4 both = o + tile[(l.y + 1u) & 15u][(l.x + 3u) & 15u];\n    imageStore(output_image, p, both);")
    assert "barrier()" in text and "shared vec4 tile" in text
    name = "grp%d" % seed
    with open(os.path.join(d, name + ".comp"), "w") as f:
        f.write(text)
    img = util.synthetic(W, H, util.F32, seed=seed)
    try:
        want = MesaShader(name, text).run({"input_image": img, "output_image": np.zeros_like(img)}, {"gain": 1.5, "shift": 3})["output_image"]
        got = util.run_hip(ctx, "input -> gn -> output\ngn: %s { gain: 1.5, shift: 3 }" % name, img)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED:", str(e)[-600:], flush=True)
        continue
    if got.tobytes() != want.tobytes():
        bad += 1
        diff = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        y, x, c = diff[0]
        print("seed", seed, "DIFF at", (x, y, c), got[y, x], want[y, x], "differing", len(diff), flush=True)
    if (seed - first) % 20 == 19:
        print("progress", seed - first + 1, "shaders,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
print("done", count, "shaders of", statements, "statements through shared memory,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
