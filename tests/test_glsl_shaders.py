"""CPU: shaders/*.comp -- the GLSL 450 restatements of the authored node types in the reference's own
plugin form (src/config/config.rs:59-75).  Nothing here can compile GLSL, so they are checked as text:
scripts/check_glsl_taps.py compares the order of their multiply-adds with oracle/rf_oracle.c.  The
checker itself is checked by mutating copies of the shaders: each mutation must make it fail."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "scripts", "check_glsl_taps.py")
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def run_check(shaders=None):
    cmd = [sys.executable, CHECK] + (["--shaders", shaders] if shaders else [])
    return subprocess.run(cmd, capture_output=True, text=True)


def test_shaders_agree_with_the_oracle():
    r = run_check()
    assert r.returncode == 0, r.stderr[-2000:]


def test_every_registry_type_has_a_shader():
    import reforge_amd as rf
    have = {f[:-5] for f in os.listdir(os.path.join(ROOT, "shaders")) if f.endswith(".comp")}
    alias = {"colour-grade": "colour_grade", "grade": "colour_grade", "passthrough": None}    # passthrough.comp is the reference's own file
    for t in rf.registry_types():
        name = alias.get(t, t)
        assert name is None or name in have, "no shaders/%s.comp for registry type %s" % (name, t)


MUTATIONS = [
    ("sharpen.comp", "    acc = fma(vec4(ws), l, acc);\n    acc = fma(vec4(wc), c, acc);\n", "    acc = fma(vec4(wc), c, acc);\n    acc = fma(vec4(ws), l, acc);\n"),   # W and C swapped
    ("gaussian5.comp", "for (int i = -RADIUS; i <= RADIUS; ++i) {                 // H taps", "for (int i = RADIUS; i >= -RADIUS; --i) {                 // H taps"),   # descending taps
    ("gaussian9.comp", "o = fma(vec4(weight(j, w)), acc, o);", "o = fma(vec4(weight(j, w)), o, acc);"),                      # accumulator and operand swapped
    ("colour_grade.comp", "    luma = fma(0.7152, tg, luma);\n    luma = fma(0.0722, tb, luma);\n", "    luma = fma(0.0722, tb, luma);\n    luma = fma(0.7152, tg, luma);\n"),
    ("conv2d.comp", "for (int dy = -r; dy <= r; ++dy) {\n        int yy = clamp(p.y + dy, 0, size.y - 1);\n        for (int dx = -r; dx <= r; ++dx) {",
     "for (int dx = -r; dx <= r; ++dx) {\n        for (int dy = -r; dy <= r; ++dy) {\n        int yy = clamp(p.y + dy, 0, size.y - 1);"),      # loops interchanged
    ("combination.comp", "precise vec4 o = fma(vec4(mix), d, a);", "vec4 o = fma(vec4(mix), d, a);"),                         # not precise
    ("sharpen.comp", "layout (binding = 0, rgba32f) uniform readonly image2D input_image;", "layout (binding = 0, rgba32f) uniform readonly image2D source;"),   # binding name
]


@pytest.mark.parametrize("k", range(len(MUTATIONS)))
def test_the_checker_catches_a_mutated_shader(tmp_path, k):
    fname, old, new = MUTATIONS[k]
    d = tmp_path / "shaders"
    shutil.copytree(os.path.join(ROOT, "shaders"), d)
    text = (d / fname).read_text()
    assert old in text, "mutation %d no longer applies to %s" % (k, fname)
    (d / fname).write_text(text.replace(old, new))
    r = run_check(str(d))
    assert r.returncode != 0, "the checker accepted %s with mutation %d" % (fname, k)


def test_explicit_weights_round_trip_through_the_config_parser():
    """scripts/glsl_weights.py prints w0..wR with enough digits that the parsed f32 values are the
    host-derived weights bit for bit (the same members drive shaders/gaussian*.comp and librfhip.so)."""
    import glsl_weights
    from oracle import graph as og
    from oracle import pixel
    for sigma, radius, t in ((1.0, 2, "gaussian5"), (2.0, 4, "gaussian9"), (0.37, 2, "gaussian5"), (3.3, 7, "gaussian")):
        text = "input -> gg -> output\ngg: %s { sigma: 9.9, %s%s }" % (t, "radius: %d, " % radius if t == "gaussian" else "", glsl_weights.as_params(sigma, radius))
        g = og.GraphOracle(text, 8, 8, pixel.FMT_RGBA32F)
        p = g.infos["gg"].params
        got = np.array([p["w%d" % i] for i in range(radius + 1)], np.float32)
        assert got.tobytes() == pixel.gaussian_weights(sigma, radius).tobytes(), (sigma, radius)


def test_stage_files_and_their_glsl_twins_declare_the_same_interface():
    """shaders/T.stage.hip is what librfhip.so compiles for a user type, shaders/T.comp what reforge itself would: the same image
    variable names on the same bindings, the same uniform members (name, type, order) -- so one config drives both."""
    import re
    from oracle import user_stage
    shaders = os.path.join(ROOT, "shaders")
    twins = [f[:-10] for f in sorted(os.listdir(shaders)) if f.endswith(".stage.hip") and os.path.exists(os.path.join(shaders, f[:-10] + ".comp"))]
    assert set(twins) >= {"edge_detect", "invert", "unsharp_mask", "local_contrast"}
    for t in twins:
        ut = user_stage.UserType(t, os.path.join(shaders, t + ".stage.hip"))
        glsl = re.sub(r"//[^\n]*", "", open(os.path.join(shaders, t + ".comp")).read())
        images = {m.group(2): int(m.group(1)) for m in re.finditer(r"binding = (\d+), rgba32f\) uniform (?:readonly |writeonly )?image2D (\w+);", glsl)}
        assert images == ut.images, (t, images, ut.images)
        block = re.search(r"uniform Params \{(.*?)\}", glsl, re.S)
        members = [tuple(d.split()) for d in block.group(1).split(";") if d.split()] if block else []
        assert members == [({"f32": "float", "i32": "int", "bool": "bool"}[ty], n) for n, ty, _ in ut.params], (t, members)
