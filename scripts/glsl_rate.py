"""GPU box: rate of filter types run FROM THEIR GLSL FILES (shaders/*.comp translated by rf_glsl.cpp, rfglsl::glsl_node_kernel) at 4K,
beside the same config on the hand-written kernels.  Prints one JSON line per (type, format)."""
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import reforge_amd as rf  # noqa: E402
from tests import util  # noqa: E402
import glsl_weights  # noqa: E402

W, H = 3840, 2160
CASES = [   # (name, types as files, config, images moved per texel)
    ("passthrough", [], "input -> pp -> output\npp: passthrough {}", 2),
    ("invert", ["invert"], "input -> iv -> output\niv: invert { enabled: true, strength: 0.7 }", 2),
    ("colour_grade", ["colour_grade"], "input -> cg -> output\ncg: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }", 2),
    ("unsharp_mask", ["unsharp_mask"], "input -> bl -> um:blurred_image\ninput -> um:input_image\num -> output\nbl: passthrough {}\num: unsharp_mask { amount: 1.5, threshold: 0.02 }", 3),
    ("sharpen", ["sharpen"], "input -> sh -> output\nsh: sharpen { amount: 0.5 }", 2),
    ("edge_detect", ["edge_detect"], "input -> ed -> output\ned: edge_detect { scale: 1.5 }", 2),
    ("gaussian5", ["gaussian5"], "input -> gg -> output\ngg: gaussian5 { sigma: 1.0, %s }" % glsl_weights.as_params(1.0, 2), 2),
    ("local_contrast", ["local_contrast"], "input -> lc -> output\nlc: local_contrast { amount: 0.8 }", 2),
    ("gaussian9", ["gaussian9"], "input -> g9 -> output\ng9: gaussian9 { sigma: 2.0, %s }" % glsl_weights.as_params(2.0, 4), 2),
]
PASSTHROUGH = ("#version 450\nlayout (local_size_x = 16, local_size_y = 16) in;\nlayout (binding = 0, rgba8) uniform readonly image2D input_image;\n"
               "layout (binding = 1, rgba8) uniform writeonly image2D output_image;\nvoid main()\n{\n    vec4 res = imageLoad(input_image, ivec2(gl_GlobalInvocationID.xy));\n"
               "    imageStore(output_image, ivec2(gl_GlobalInvocationID.xy), res);\n}\n")

ctx = rf.Context(0)
d = tempfile.mkdtemp()
for f in os.listdir(os.path.join(ROOT, "shaders")):
    if f.endswith(".comp"):
        shutil.copy(os.path.join(ROOT, "shaders", f), d)
open(os.path.join(d, "passthrough.comp"), "w").write(PASSTHROUGH)
rf.set_shader_path(d)
ONLY = sys.argv[1:]
for fmt, fname, bpp in ((util.F32, "rgba32f", 16), (util.U8, "rgba8", 4)):
    for name, types, text, images in CASES:
        if ONLY and name not in ONLY:
            continue
        row = {"type": name, "fmt": fname}
        for mode in ("glsl", "built_in"):
            rf.set_type_lookup(mode == "glsl")
            g = rf.Graph(ctx, rf.Config(text), W, H, fmt)
            g.fill_synthetic(2)
            g.execute()
            g.wait()
            g.time_launches(5)
            label, ms = g.time_launches(40)[-1]      # the node itself is the last launch
            us = ms * 1e3
            row[mode + "_us"] = round(us, 1)
            row[mode + "_frac"] = round(images * W * H * bpp / (us * 1e-6) / 8e12, 3)
            row[mode + "_launch"] = label
            g.close()
        print(json.dumps(row), flush=True)
