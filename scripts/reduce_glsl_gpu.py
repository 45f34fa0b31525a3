"""GPU box: which variable of a generated shader (tests/glsl_gen.py) differs between the product and the translation compiled for the host.
usage: reduce_glsl_gpu.py <seed> <statements> <fmt: 0 | 1>"""
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import reforge_amd as rf  # noqa: E402
from tests import util  # noqa: E402
from tests.glsl_gen import Gen  # noqa: E402
from tests.glsl_host import HostShader  # noqa: E402

seed, statements, fmt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = Gen(seed)
text = g.shader(statements)
head, _ = text.split("    precise vec4 o = c * 0.0;")
d = tempfile.mkdtemp()
rf.set_shader_path(d)
rf.set_type_lookup(True)
ctx = rf.Context(0)
img = util.synthetic(150, 67, fmt, seed=seed)
conv = {"float": "vec4(%s)", "int": "vec4(float(%s & 0xffffff), float((%s >> 24) & 255), 0.0, 0.0)", "uint": "vec4(float(%s & 0xffffffu), float(%s >> 24), 0.0, 0.0)", "bool": "vec4(%s ? 1.0 : 0.0)",
        "vec2": "vec4(%s, %s)", "vec3": "vec4(%s, 1.0)", "vec4": "%s", "ivec2": "vec4(vec2(%s & 0xffff), vec2((%s >> 16) & 0xffff))"}
n = 0
for t, names in g.vars.items():
    for v in names:
        if v in ("c", "e", "s", "p"):
            continue
        e = conv[t] % ((v, v) if conv[t].count("%s") == 2 else v)
        body = head + "    imageStore(output_image, p, %s);\n}\n" % e
        n += 1
        name = "red%d" % n
        with open(os.path.join(d, name + ".comp"), "w") as f:
            f.write(body)
        want = np.zeros((67, 150, 4), np.float32)
        imgf = util.synthetic(150, 67, fmt, seed=seed)
        want = np.zeros_like(imgf)
        HostShader(name, body).run({"input_image": imgf, "output_image": want}, {"gain": 1.5, "shift": 3})
        got = util.run_hip(ctx, "input -> gn -> output\ngn: %s { gain: 1.5, shift: 3 }" % name, imgf)
        if got.tobytes() != want.tobytes():
            y, x, c = np.argwhere(got != want)[0]
            print(t, v, "DIFFERS at", (x, y), got[y, x], want[y, x], "differing", (got != want).sum(), flush=True)
            for line in head.split("\n"):
                if re.search(r"\b%s\b" % v, line):
                    print("    ", line.strip()[:500], flush=True)
print("checked", n, "variables")
