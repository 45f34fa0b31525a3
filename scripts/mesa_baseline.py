"""CPU: what the reference's CPU route would cost on this machine's cores.  SURVEY 8(f-4) asks for the reference on lavapipe (Mesa's CPU Vulkan
driver); there is no Vulkan loader or ICD in the image, but lavapipe's back end -- llvmpipe -- is, behind OpenGL: this script times the
dispatches of the headline graph's three filter files (shaders/gaussian5.comp, colour_grade.comp, sharpen.comp: what the reference would
compile and dispatch, one vkCmdDispatch per node) at 3840x2160 rgba32f through tests/native/mesa_glsl.c, beside the oracle.
usage: mesa_baseline.py [W H] [repeat]"""
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np  # noqa: E402

import reforge_amd as rf  # noqa: E402
from tests import util  # noqa: E402
from tests.mesa_glsl import MesaShader, runner, version, why_not  # noqa: E402
import glsl_weights  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 3
exe = runner()
if not exe:
    sys.exit("Mesa is not usable here: " + why_not())
print(version())
os.environ["RF_MESA_REPEAT"] = str(repeat)
img = util.synthetic(W, H, util.F32)
total = 0.0
nodes = [("gaussian5", dict({"sigma": 1.0}, **{"w%d" % i: w for i, w in enumerate(glsl_weights.weights(1.0, 2))})),
         ("colour_grade", {"slope": 1.1, "offset": -0.02, "saturation": 1.2}), ("sharpen", {"amount": 0.5})]
for t, params in nodes:
    sh = MesaShader(t, open(os.path.join(ROOT, "shaders", t + ".comp")).read())
    img = sh.run({"input_image": img, "output_image": np.zeros_like(img)}, params)["output_image"]
    seen = {"ms": float(re.search(r"dispatch_ms ([0-9.]+)", sh.last_log).group(1))}
    print("%-14s %8.2f ms per dispatch  (%.1f Mpx/s)" % (t + ".comp", seen["ms"], W * H / seen["ms"] / 1e3), flush=True)
    total += seen["ms"]
print("three dispatches: %.2f ms per frame = %.1f Mpx/s on %d cores (llvmpipe)" % (total, W * H / total / 1e3, os.cpu_count()))
