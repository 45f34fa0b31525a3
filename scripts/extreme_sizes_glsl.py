import os, sys, shutil, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import reforge_amd as rf
from tests import util
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = tempfile.mkdtemp()
for t in ("gaussian5", "sharpen", "colour_grade", "local_contrast", "unsharp_mask"):
    shutil.copy(os.path.join(ROOT, "shaders", t + ".comp"), d)
rf.set_shader_path(d); rf.set_type_lookup(True)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import glsl_weights
text = "input -> gg -> cg -> sh -> output\ngg: gaussian5 { sigma: 1.0, %s }\ncg: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }\nsh: sharpen { amount: 0.5 }" % glsl_weights.as_params(1.0, 2)
ctx = rf.Context(0)
bad = 0
for W, H in ((65536, 8), (8, 65536), (1, 100000), (100000, 1), (3, 3), (2, 7), (16383, 17), (4097, 4097)):
    for fmt in (util.F32, util.U8):
        if W * H > 40_000_000 and fmt == util.F32: continue
        img = util.synthetic(W, H, fmt, seed=W + H)
        want = util.run_oracle(text, img)
        got = util.run_hip(ctx, text, img)
        ok = got.tobytes() == want.tobytes()
        bad += 0 if ok else 1
        print(W, H, fmt, "ok" if ok else "MISMATCH", flush=True)
print("done", bad, "failures")
