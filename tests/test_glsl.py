"""CPU: filter types that are GLSL compute shaders -- {shader_path}/{type}.comp, the reference's own plugin form
(src/config/config.rs:59-75; compiled and reflected there by shaderc + spirv-reflect, src/vulkan/shader.rs:73-160).

* reflection: what rf_glsl.cpp reads off a file is what the reference's reflection gives -- image variables by name, storage blocks by
  block type name, uniform members by member name -- and, for the shipped shaders, what the built-in registry says of the same type;
* semantics without a GPU: the translation of every shaders/*.comp, compiled for the HOST (tests/glsl_host.py), gives the bits of
  oracle/rf_oracle.c for the same node -- translator and GLSL prelude against code that shares nothing with them;
* the gfx950 code objects build (hiprtc, no device needed); files outside the subset are refused with file:line.
tests/test_gpu_glsl.py runs the same files on the GPU."""
import os
import shutil
import sys

import numpy as np
import pytest

import reforge_amd as rf
from oracle import graph as ograph
from oracle import pixel
from tests import util
from tests.glsl_host import HostShader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import glsl_weights  # noqa: E402

COMP = sorted(f[:-5] for f in os.listdir(SHADERS) if f.endswith(".comp"))


def text_of(t):
    return open(os.path.join(SHADERS, t + ".comp")).read()


@pytest.fixture
def glsl_dir(tmp_path):
    old = rf.shader_path()
    rf.set_shader_path(str(tmp_path))
    rf.set_type_lookup(True)
    yield tmp_path
    rf.set_type_lookup(False)
    rf.set_shader_path(old)


# ---- reflection ---------------------------------------------------------------------------------------------------------------------
def test_every_shipped_shader_translates_and_reflects_like_the_registry():
    assert len(COMP) >= 15
    for t in COMP:
        r = rf.glsl_reflect(t, text_of(t))
        assert r["local_size"] == [16, 16, 1] and not r["grouped"], t
        if t in rf.registry_types():
            for im in r["images"]:      # same variable on the same binding as the hand-written type (shader.rs:151-153)
                assert rf.registry_binding(t, im["name"]) == im["binding"], (t, im)
        src = rf.glsl_translate(t, text_of(t))
        assert "struct RfgShader" in src and "RFG void main()" in src and "#line 1 \"%s.comp\"" % t in src


def test_uniform_blocks_follow_std140_and_storage_blocks_std430():
    src = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform image2D image;
#define N 3
const int M = N + 1;
layout (binding = 1) uniform Params { float a; vec2 b; vec3 c; float d; float w[N]; mat3 m; int k; bool on; uvec4 u; } ;
layout (binding = 2) uniform More { float z; } more;
layout (std430, binding = 3) buffer Data { float f[M]; vec2 g; vec3 h[2]; vec4 i; mat2 j; uint n; } ;
layout (std430, binding = 4) readonly buffer Named { vec4 v[4]; float s; } named;
void main() { }
"""
    r = rf.glsl_reflect("layouts", src)
    p = {m["name"]: (m["offset"], m["stride"], m["bytes"]) for m in r["uniform_blocks"][0]["members"]}
    assert p == {"a": (0, 4, 4), "b": (8, 8, 8), "c": (16, 12, 12), "d": (28, 4, 4), "w": (32, 16, 48), "m": (80, 48, 48), "k": (128, 4, 4), "on": (132, 4, 4), "u": (144, 16, 16)}
    assert r["uniform_blocks"][0]["bytes"] == 160 and r["uniform_blocks"][1]["base"] == 160 and r["uniform_bytes"] == 176
    assert [m["name"] for m in r["uniform_blocks"][1]["members"]] == ["more.z"]      # the key of the reference's UBO map (pipeline_graph.rs:276-292)
    d = {m["name"]: (m["offset"], m["stride"], m["bytes"]) for m in r["storage_blocks"][0]["members"]}
    assert d == {"f": (0, 4, 16), "g": (16, 8, 8), "h": (32, 16, 32), "i": (64, 16, 16), "j": (80, 16, 16), "n": (96, 4, 4)}
    assert r["storage_blocks"][0]["bytes"] == 112 and not r["storage_blocks"][0]["readonly"]
    assert r["storage_blocks"][1]["readonly"] and r["storage_blocks"][1]["instance"] == "named" and r["storage_blocks"][1]["bytes"] == 80
    assert r["images"] == [{"name": "image", "binding": 0, "readonly": False, "writeonly": False}]
    src2 = rf.glsl_translate("layouts", src)
    assert "static constexpr int M = N + 1 ;" in src2 and "Named_t* named;" in src2 and "float* f;" in src2 and "vec2* rfg_p_g;" in src2


def test_what_the_translator_rewrites():
    src = rf.glsl_translate("tt", """#version 450
#extension GL_EXT_something : enable
precision highp float;
layout (local_size_x = 8, local_size_y = 8) in;
layout (binding = 0, rgba8) uniform readonly image2D input_image;
layout (binding = 1, rgba8) uniform writeonly image2D output_image;
struct Light { vec3 dir; float power; };
float helper(in float a, out float b, inout vec2 c, const in float d[2]);
#define HALF(v) ((v) * vec4(0.5))
float helper(in float a, out float b, inout vec2 c, const in float d[2]) { b = a * 2.; c.ts = c.st; return 1e-3 + d[1] + .5 + 3.0lf; }
void main()
{
    highp float x = 1.0; precise vec4 o = vec4(x, 0, 1u, true);
    float arr[2] = float[2](1.0, 2.0);
    Light l = Light(vec3(0.0, 1.0, 0.0), 2.0);
    vec2 c = vec2(0.25); float b;
    o.r = helper(x, b, c, arr) + l.power + float(0x1F);
    if (any(not(lessThan(o, vec4(0.5))))) o = HALF(o);
    imageStore(output_image, ivec2(gl_GlobalInvocationID.xy), o.bgra);
}
""")
    assert "#extension" not in src and "precision" not in src and "highp" not in src and "precise" not in src
    assert src.count("helper(") == 2      # the prototype is gone (a member function is declared once); definition + call remain
    assert "RFG float helper( float a,  float &b,  vec2 &c, const  rfg_arr<float, (2)> d)" in src      # arrays are values (a struct around the C array)
    assert "b = a * 2.f; c.yx = c.xy; return 1e-3f + d[1] + .5f + 3.0;" in src
    assert "float x = 1.0f;  vec4 o = mk_vec4(x, 0, 1u, true);" in src
    assert "rfg_arr<float, (2)> arr = rfg_arr<float, (2)>{1.0f, 2.0f};" in src
    assert "Light l = Light{mk_vec3(0.0f, 1.0f, 0.0f), 2.0f};" in src
    assert "#define HALF(v) ((v) * mk_vec4(0.5f))" in src and "#undef HALF" in src
    assert "rfg_not(lessThan(o, mk_vec4(0.5f)))" in src and "float(0x1F)" in src and "o.bgra" in src
    assert "LX = 8, LY = 8, LZ = 1, NIMG = 2, NBUF = 0, UBO = 0" in src and "GROUPED = false" in src


REFUSED = [
    ("layout (binding = 0) uniform samplerCube tex;\nvoid main() {}", "bad.comp:1", "samplerCube"),
    ("void main() { double x = 1.0; }", "bad.comp:1", "double"),
    ("layout (std430, binding = 0) buffer Data { float v[]; };\nvoid main() {}", "bad.comp:1", "unsized"),
    ("struct S { float a; };\nlayout (binding = 0) uniform P { S s; };\nvoid main() {}", "bad.comp:2", "nested structs"),
    ("layout (rgba32f) uniform image2D image;\nvoid main() {}", "bad.comp:1", "binding"),
    ("layout (binding = 1, rgba32f) uniform image2D aa;\nlayout (binding = 1, rgba32f) uniform image2D bb;\nvoid main() {}", "bad.comp:2", "used twice"),
    ("layout (binding = 0, rgba32f) uniform image2D image;\nfloat f() { return 1.0; }", "bad.comp", "no `void main()`"),
    ("#include \"common.glsl\"\nvoid main() {}", "bad.comp:1", "#include"),
    ("layout (set = 1, binding = 0, rgba32f) uniform image2D image;\nvoid main() {}", "bad.comp:1", "set 0"),
    ("layout (binding = 0, rgba32f) uniform image2D image;\nlayout (binding = 1) uniform P { float w[9]; float v[60]; };\nvoid main() {}", "bad.comp:2", "limit is 256"),
    ("layout (local_size_x = 64, local_size_y = 32) in;\nlayout (binding = 0, rgba32f) uniform image2D image;\nvoid main() {}", "bad.comp", "out of range"),
    ("layout (binding = 0, rgba32f) uniform readonly image2D image;\nvoid main() {}", "bad.comp", "nothing it does can be observed"),
    ("#pragma rf radius 1\nlayout (binding = 0, rgba32f) uniform image2D image;\nvoid main() {}", "bad.comp", "cannot run in place"),
]


@pytest.mark.parametrize("k", range(len(REFUSED)))
def test_files_outside_the_subset_are_refused_with_file_and_line(k):
    body, where, what = REFUSED[k]
    with pytest.raises(rf.RfError) as e:
        rf.glsl_translate("bad", body)
    assert where in str(e.value) and what in str(e.value), str(e.value)


# ---- semantics on the host: the shipped shaders against oracle/rf_oracle.c -----------------------------------------------------------------
def host_node(t, images, params=None, buffers=None):
    HostShader(t, text_of(t)).run(images, params, buffers)


def gparams(sigma, radius):
    return dict({"sigma": sigma}, **{"w%d" % i: w for i, w in enumerate(glsl_weights.weights(sigma, radius))})


@pytest.mark.parametrize("fmt", [util.F32, util.U8], ids=["rgba32f", "rgba8"])
def test_the_shipped_shaders_run_on_the_host_give_the_oracles_bits(fmt):
    W, H = 37, 23
    img = util.synthetic(W, H, fmt)
    other = util.synthetic(W, H, fmt, seed=77)

    def out():
        return np.zeros_like(img)

    o = out()
    host_node("gaussian5", {"input_image": img, "output_image": o}, gparams(1.0, 2))
    util.assert_same(o, pixel.gaussian(img, 2, sigma=1.0), "gaussian5")
    o = out()
    host_node("gaussian9", {"input_image": img, "output_image": o}, gparams(2.0, 4))
    util.assert_same(o, pixel.gaussian(img, 4, sigma=2.0), "gaussian9")
    o = out()
    host_node("gaussian", {"input_image": img, "output_image": o}, dict(gparams(2.5, 7), radius=7))
    util.assert_same(o, pixel.gaussian(img, 7, sigma=2.5), "gaussian radius 7")
    o = out()
    host_node("colour_grade", {"input_image": img, "output_image": o}, {"slope": 1.1, "offset": -0.02, "saturation": 1.2})
    util.assert_same(o, pixel.colour_grade(img, 1.1, -0.02, 1.2), "colour_grade")
    o = img.copy()
    host_node("colour_grade_inplace", {"image": o}, {"slope": 0.9, "offset": 0.03, "saturation": 0.4})
    util.assert_same(o, pixel.colour_grade(img, 0.9, 0.03, 0.4), "colour_grade_inplace")
    o = out()
    host_node("sharpen", {"input_image": img, "output_image": o}, {"amount": 0.75})
    util.assert_same(o, pixel.sharpen(img, 0.75), "sharpen")
    o = out()
    host_node("combination", {"input_image0": img, "input_image1": other, "output_image": o}, {"mix": 0.3})
    util.assert_same(o, pixel.mix(img, other, 0.3), "combination")
    luma, chroma = out(), out()
    host_node("split_luma", {"input_image": img, "luma_image": luma, "chroma_image": chroma})
    wl, wc = pixel.split_luma(img, np.zeros_like(img), np.zeros_like(img))
    util.assert_same(luma, wl, "split_luma luma")
    util.assert_same(chroma, wc, "split_luma chroma")
    K = 7
    w = ograph.default_conv_weights(K, 1.5).astype(np.float32)
    buf = np.zeros(961, np.float32)
    buf[:K * K] = w.ravel()
    o = out()
    host_node("conv2d", {"input_image": img, "output_image": o}, {"ksize": K}, {"ConvWeights": buf})
    util.assert_same(o, pixel.conv2d(img, w.reshape(K, K)), "conv2d 7x7 through its ConvWeights block")


def test_the_user_type_twins_run_on_the_host_match_their_stage_files():
    """edge_detect / invert / unsharp_mask / local_contrast exist as .stage.hip (what the oracle compiles for the host,
    oracle/user_stage.py) and as .comp: one config, the same bits"""
    old = util.register_user_types()
    try:
        W, H = 41, 19
        img = util.synthetic(W, H, util.F32)
        for t, params in (("invert", {"enabled": 1, "strength": 0.7}), ("edge_detect", {"scale": 1.5}), ("local_contrast", {"amount": 0.8})):
            text = "input -> nn -> output\nnn: %s { %s }" % (t, ", ".join("%s: %s" % (k, ("true" if v == 1 and k == "enabled" else v)) for k, v in params.items()))
            want = util.run_oracle(text, img)
            o = np.zeros_like(img)
            host_node(t, {"input_image": img, "output_image": o}, params)
            util.assert_same(o, want, t)
        blurred = pixel.gaussian(img, 4, sigma=2.0)
        want = util.run_oracle("input -> bl -> um:blurred_image\ninput -> um:input_image\num -> output\nbl: gaussian9 { sigma: 2.0 }\num: unsharp_mask { amount: 1.5, threshold: 0.02 }", img)
        o, m = np.zeros_like(img), np.zeros_like(img)
        host_node("unsharp_mask", {"input_image": img, "blurred_image": blurred, "output_image": o, "mask_image": m}, {"amount": 1.5, "threshold": 0.02})
        util.assert_same(o, want, "unsharp_mask")
    finally:
        rf.set_shader_path(old)


# ---- point shaders are row stages: they fuse with their neighbours ----------------------------------------------------------------------------
GAIN = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (binding = 2) uniform Params { float gain; float bias; };
void main()
{
    ivec2 size = imageSize(output_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (p.x >= size.x || p.y >= size.y) return;
    vec4 c = imageLoad(input_image, p);
    imageStore(output_image, p, vec4(c.rgb * gain + bias, c.a));
}
"""
NOT_POINT = {
    "reads a neighbour": GAIN.replace("imageLoad(input_image, p)", "imageLoad(input_image, p + ivec2(1, 0))"),
    "uses its position": GAIN.replace("c.rgb * gain + bias", "c.rgb * gain + bias * float(p.x)"),
    "returns early by position": GAIN.replace("    vec4 c =", "    if (p.x < 10) return;\n    vec4 c ="),
    "coordinate through a macro": GAIN.replace("void main()", "#define HERE ivec2(gl_GlobalInvocationID.xy)\nvoid main()").replace("imageLoad(input_image, p)", "imageLoad(input_image, HERE)"),
    "loads in a helper": GAIN.replace("void main()", "vec4 get(ivec2 q) { return imageLoad(input_image, q); }\nvoid main()").replace("imageLoad(input_image, p);", "get(p);"),
    "stores elsewhere": GAIN.replace("imageStore(output_image, p,", "imageStore(output_image, size - 1 - p,"),
    "uses the frame size": GAIN.replace("c.rgb * gain + bias", "c.rgb * gain + bias / float(size.x)"),
    "a local_size that covers part of the frame": GAIN.replace("local_size_x = 16", "local_size_x = 8"),
    "a second input": GAIN.replace("layout (binding = 2)", "layout (binding = 3, rgba32f) uniform readonly image2D other_image;\nlayout (binding = 2)"),
}


def test_point_shaders_are_recognised_conservatively():
    assert rf.glsl_reflect("gain", GAIN)["point"]
    assert rf.glsl_reflect("gain", GAIN.replace("if (p.x >= size.x || p.y >= size.y) return;", "if (any(greaterThanEqual(p, imageSize(input_image)))) { return; }"))["point"]
    assert rf.glsl_reflect("gain", GAIN.replace("    ivec2 size = imageSize(output_image);\n", "").replace("    if (p.x >= size.x || p.y >= size.y) return;\n", ""))["point"]      # the reference's passthrough has no guard
    for why, text in NOT_POINT.items():
        assert not rf.glsl_reflect("gain", text)["point"], why
    point = {t for t in COMP if rf.glsl_reflect(t, text_of(t))["point"]}
    assert point == {"colour_grade", "colour_grade_inplace", "invert", "pulse"}, point


def test_a_point_shader_fuses_with_its_neighbours(glsl_dir):
    rf.set_type_lookup(False)
    (glsl_dir / "gain.comp").write_text(GAIN)
    shutil.copy(os.path.join(SHADERS, "invert.comp"), glsl_dir / "invert.comp")
    (glsl_dir / "shifted.comp").write_text(NOT_POINT["reads a neighbour"])
    text = "input -> gg -> gn -> iv -> sh -> output\ngg: gaussian5 { sigma: 1.0 }\ngn: gain { gain: 1.25, bias: -0.125 }\niv: invert { enabled: true, strength: 0.5 }\nsh: sharpen { amount: 0.5 }"
    p = rf.Plan(rf.Config(text))
    assert p.launches() == ["gg+gn+iv+sh"] and p.needs_jit() == [True]      # ONE launch: two GLSL files between two hand-written stencils
    assert p.jit_compile(rf.RF_FORMAT_RGBA32F) > 8192 and p.jit_compile(rf.RF_FORMAT_RGBA8) > 8192
    q = rf.Plan(rf.Config(text.replace("gn: gain", "gn: shifted")))
    assert q.launches() == ["gg", "gn", "iv+sh"]                             # a shader that is not a point operation keeps a kernel of its own
    assert rf.Plan(rf.Config(text), rf.RF_GRAPH_NO_FUSION).launches() == ["gg", "gn", "iv", "sh"]


# ---- translation-invariant stencils run on the LDS-tiled window kernel -----------------------------------------------------------------------
BOX5 = """#version 450
#pragma rf radius 2
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (binding = 2) uniform Params { float gain; };
vec4 tap(ivec2 q, ivec2 size) { return imageLoad(input_image, clamp(q, ivec2(0), size - 1)); }
void main()
{
    ivec2 size = imageSize(output_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (p.x >= size.x || p.y >= size.y) return;
    vec4 acc = vec4(0.0);
    for (int dy = -2; dy <= 2; ++dy) {
        int yy = p.y + dy;
        for (int dx = -2; dx <= 2; ++dx) acc += tap(ivec2(p.x + dx, yy), size);
    }
    imageStore(output_image, p, acc * gain);
}
"""
NOT_STENCIL = {
    "the position in a float": BOX5.replace("acc * gain", "acc * gain * float(p.x & 1)"),
    "the frame size in a float": BOX5.replace("acc * gain", "acc * gain / float(size.x)"),
    "a loop bound by the position": BOX5.replace("dx <= 2;", "dx <= p.x;"),
    "a branch on the position": BOX5.replace("    vec4 acc", "    if (p.x == 100) { imageStore(output_image, p, vec4(1.0)); return; }\n    vec4 acc"),
    "the position through an integer into a float": BOX5.replace("    vec4 acc", "    int parity = (p.x + p.y) & 1;\n    vec4 acc").replace("acc * gain", "acc * gain * float(parity)"),
    "the position through a helper into a float": BOX5.replace("void main()", "float fade(int x) { return float(x) * 0.001; }\nvoid main()").replace("acc * gain", "acc * gain * fade(p.x)"),
    "a store somewhere else": BOX5.replace("imageStore(output_image, p,", "imageStore(output_image, ivec2(size.x - 1 - p.x, p.y),"),
    "an offset that follows a parameter": BOX5.replace("uniform Params { float gain; }", "uniform Params { float gain; int reach; }").replace("p.x + dx, yy", "p.x + dx * reach, yy"),
    "a loop bound by a parameter feeding the coordinate": BOX5.replace("uniform Params { float gain; }", "uniform Params { float gain; int reach; }").replace("dx <= 2;", "dx <= reach;"),
    "no stated radius": BOX5.replace("#pragma rf radius 2\n", ""),
    "an image read and written": BOX5.replace("uniform readonly image2D input_image", "uniform image2D input_image"),
}


def test_translation_invariant_stencils_are_recognised_conservatively():
    r = rf.glsl_reflect("box5", BOX5)
    assert r["stencil"] and not r["point"] and r["radius"] == 2
    assert r["stencil_why_not"] == ""
    says = {"the position in a float": "line 18: `p` carries", "a loop bound by the position": "line 16: `p`", "the position through a helper into a float": "line 8: `x` carries",
            "a store somewhere else": "line 18: a store somewhere else", "an offset that follows a parameter": "a coordinate follows a uniform", "no stated radius": "does not state `#pragma rf radius N`",
            "an image read and written": "`input_image` is read AND written"}
    for why, text in NOT_STENCIL.items():
        rr = rf.glsl_reflect("box5", text)
        assert not rr["stencil"] and rr["stencil_why_not"], why      # and the reflection says what kept it off the fast forms
        assert says.get(why, "") in rr["stencil_why_not"], (why, rr["stencil_why_not"])
    stencil = {t for t in COMP if rf.glsl_reflect(t, text_of(t))["stencil"]}
    assert stencil == {"gaussian5", "gaussian9", "sharpen", "edge_detect", "local_contrast"}, stencil      # (gaussian: 80 bytes of uniforms; conv2d: a storage block)


def test_a_three_by_three_stencil_is_a_row_stage_and_fuses(glsl_dir):
    rf.set_type_lookup(False)
    shutil.copy(os.path.join(SHADERS, "sharpen.comp"), glsl_dir / "sharp3.comp")
    shutil.copy(os.path.join(SHADERS, "edge_detect.comp"), glsl_dir / "edges3.comp")
    shutil.copy(os.path.join(SHADERS, "invert.comp"), glsl_dir / "invert.comp")
    for t in ("sharp3", "edges3"):
        r = rf.glsl_reflect(t, (glsl_dir / (t + ".comp")).read_text())
        assert r["stencil"] and r["box"] and not r["point"], t
    assert not rf.glsl_reflect("gaussian5", text_of("gaussian5"))["box"]      # radius 2
    text = ("input -> gg -> s3 -> iv -> e3 -> cg -> output\ngg: gaussian5 { sigma: 1.0 }\ns3: sharp3 { amount: 0.5 }\niv: invert { enabled: true, strength: 0.5 }\n"
            "e3: edges3 { scale: 1.5 }\ncg: colour_grade { slope: 1.1, offset: 0.0, saturation: 1.0 }")
    p = rf.Plan(rf.Config(text))
    assert p.launches() == ["gg+s3+iv+e3+cg"] and [l["radius"] for l in p.launch_info()] == [4]      # three GLSL files inside one launch
    assert p.jit_compile(rf.RF_FORMAT_RGBA32F) > 8192 and p.jit_compile(rf.RF_FORMAT_RGBA8) > 8192
    assert rf.Plan(rf.Config(text), rf.RF_GRAPH_GLSL_NODES).launches() == ["gg", "s3", "iv", "e3", "cg"]


def test_a_recognised_stencil_compiles_its_window_kernel_too(glsl_dir):
    (glsl_dir / "box5.comp").write_text(BOX5)
    p = rf.Plan(rf.Config("input -> bb -> output\nbb: box5 { gain: 0.04 }"))
    two = p.jit_compile(rf.RF_FORMAT_RGBA32F)
    (glsl_dir / "box5.comp").write_text(NOT_STENCIL["the position in a float"])
    os.utime(glsl_dir / "box5.comp", ns=(10 ** 18, 10 ** 18))
    one = rf.Plan(rf.Config("input -> bb -> output\nbb: box5 { gain: 0.04 }")).jit_compile(rf.RF_FORMAT_RGBA32F)
    assert two > one + 8192, (two, one)      # the generic kernel and rf::user_node_kernel over the generated window stage


# ---- == / != on vectors, specialisation constants, integer and packing built-ins --------------------------------------------------------------
EQUALITY = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform image2D image;
layout (constant_id = 3) const int MODE = 2;
#define SAME(a, b) ((a) == (b))
void main()
{
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (any(greaterThanEqual(p, imageSize(image)))) return;
    vec4 c = imageLoad(image, p);
    bool corner = p == ivec2(0) || p + 1 == imageSize(image);
    bool e = c.rgb != vec3(0.0) && MODE == 2 ? c.a <= 0.5 == false : SAME(c.r, c.g);
    int k = (p.x & 1) == 0 ? 1 : 2;
    if (corner != e) c.a = float(k);
    float[3] lift = float[3](0.25, 0.25, 0.25);
    for (int i = 0; i != lift.length(); ++i) c[i] += lift[i];
    if (mat2(1.0) == mat2(1.0, 0.0, 0.0, 1.0)) c.b = 1.0;
    uint code4 = packUnorm4x8(vec4(c.r - 0.25, 0.5, 2.0, -1.0));
    c.g = unpackUnorm4x8(code4).x + float(bitCount(code4 & 0xFFu)) + float(findMSB(uint(p.x + 1))) + float(bitfieldExtract(uint(p.y), 1, 3));
    imageStore(image, p, c);
}
"""


def equality(img):
    f = np.float32
    H, W, _ = img.shape
    out = img.copy()
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    corner = ((xs == 0) & (ys == 0)) | ((xs + 1 == W) & (ys + 1 == H))
    e = np.where(~np.all(img[..., :3] == 0, axis=-1), ~(img[..., 3] <= f(0.5)), img[..., 0] == img[..., 1])      # a vector != is "any component differs"
    k = np.where((xs & 1) == 0, 1, 2).astype(f)
    out[..., 3] = np.where(corner != e, k, img[..., 3])
    out[..., :3] = img[..., :3] + f(0.25)
    out[..., 2] = f(1.0)
    code = np.rint(np.clip(out[..., 0] - f(0.25), 0, 1) * f(255.0)).astype(np.int64)
    bits = np.array([bin(int(v)).count("1") for v in code.ravel()], f).reshape(code.shape)
    msb = np.floor(np.log2(xs + 1)).astype(f)
    out[..., 1] = ((code.astype(f) / f(255.0) + bits) + msb) + ((ys >> 1) & 7).astype(f)
    return out


def test_equality_of_vectors_is_one_bool_and_the_integer_built_ins_work():
    src = rf.glsl_translate("equality", EQUALITY)
    assert "rfg_eq(p , mk_ivec2(0)) || rfg_eq(p + 1 , imageSize(image))" in src
    assert "rfg_ne(c.rgb , mk_vec3(0.0f)) && rfg_eq(MODE , 2) ? rfg_eq(c.a <= 0.5f , false) : SAME(c.r, c.g)" in src
    assert "#define SAME(a, b) (rfg_eq((a) , (b)))" in src and "static constexpr int MODE = 2 ;" in src and "rfg_ne(i , rfg_length(lift))" in src and "rfg_arr<float, (3)> lift = rfg_arr<float, (3)>{0.25f, 0.25f, 0.25f};" in src
    img = util.synthetic(37, 21, util.F32)
    img[3, 5, :3] = 0.0
    img[4, 6, 0] = img[4, 6, 1]
    o = img.copy()
    HostShader("equality", EQUALITY).run({"image": o})
    util.assert_same(o, equality(img), "equality")


def test_a_shaders_own_names_do_not_meet_the_generated_ones():
    """the translation wraps the file in a struct with members and a bind function of its own: their names are reserved ones (rfg_*,
    Rfg*), so a file may call its things Shader, Info, Px, f, img, buf, ubo, i, t, T"""
    src = """#version 450
#define f 2.0
#define T 4
#define img input_image
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (binding = 2) uniform Params { float ubo; float buf[T]; bool t; };
struct Shader { float Px; };
struct Info { vec2 i; };
void main()
{
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    Shader s = Shader(ubo);
    Info info = Info(vec2(buf[0], t ? 1.0 : 0.0));
    imageStore(output_image, p, imageLoad(img, p) * f + vec4(s.Px, info.i, 0.0));
}
"""
    img = util.synthetic(33, 17, util.F32)
    o = np.zeros_like(img)
    HostShader("names", src).run({"input_image": img, "output_image": o}, {"ubo": 0.5, "t": 1})
    util.assert_same(o, img * np.float32(2.0) + np.array([0.5, 0.0, 1.0, 0.0], np.float32), "names")


# ---- combined image samplers (shader.rs:98): texture() through the graph's one sampler (vkutils.rs:358-365) ----------------------------------
RESAMPLE = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0) uniform sampler2D source;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (binding = 2) uniform Params { float shift_x; float shift_y; float zoom; };
void main()
{
    ivec2 size = textureSize(source, 0);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (p.x >= size.x || p.y >= size.y) return;
    vec2 uv = ((vec2(p) + vec2(0.5)) * zoom + vec2(shift_x, shift_y)) / vec2(size);
    vec4 filtered = texture(source, uv);
    vec4 exact = texelFetch(source, ivec2(size.x - 1 - p.x, p.y), 0);
    imageStore(output_image, p, vec4(filtered.rgb, exact.a));
}
"""


def resample(img, shift_x, shift_y, zoom):
    """texture() as rf_glsl_dev.h states it: unnormalised = uv * size - 0.5, four texels weighted (1-a)(1-b), a(1-b), (1-a)b, ab in
    float32, U clamped to the edge, V REPEATED (the reference's sampler never sets address_mode_v, vkutils.rs:358-365)"""
    f = np.float32
    H, W, _ = img.shape
    src = img.astype(f) if img.dtype == np.float32 else (img.astype(f) / f(255.0))
    xs, ys = np.meshgrid(np.arange(W, dtype=f), np.arange(H, dtype=f))
    uvx = ((xs + f(0.5)) * f(zoom) + f(shift_x)) / f(W)
    uvy = ((ys + f(0.5)) * f(zoom) + f(shift_y)) / f(H)
    u, v = uvx * f(W) - f(0.5), uvy * f(H) - f(0.5)
    fu, fv = np.floor(u), np.floor(v)
    a, b = (u - fu)[..., None], (v - fv)[..., None]
    i0, j0 = fu.astype(np.int64), fv.astype(np.int64)
    i1, j1 = np.clip(i0 + 1, 0, W - 1), np.mod(j0 + 1, H)
    i0, j0 = np.clip(i0, 0, W - 1), np.mod(j0, H)
    one = f(1.0)
    out = ((((one - a) * (one - b)) * src[j0, i0] + (a * (one - b)) * src[j0, i1]) + ((one - a) * b) * src[j1, i0]) + (a * b) * src[j1, i1]
    out[..., 3] = src[:, ::-1, 3]
    return out.astype(f)


@pytest.mark.parametrize("params", [(0.0, 0.0, 1.0), (0.37, -1.25, 1.0), (3.5, 40.0, 0.75)])
def test_a_sampler2D_is_filtered_by_the_graphs_sampler_on_the_host(params):
    r = rf.glsl_reflect("resample", RESAMPLE)
    assert r["images"][0] == {"name": "source", "binding": 0, "readonly": True, "writeonly": False, "sampled": True}
    img = util.synthetic(45, 23, util.F32)
    o = np.zeros_like(img)
    HostShader("resample", RESAMPLE).run({"source": img, "output_image": o}, {"shift_x": params[0], "shift_y": params[1], "zoom": params[2]})
    util.assert_same(o, resample(img, *params), "texture() %r" % (params,))
    if params == (0.0, 0.0, 1.0):      # sampled at the texel centres ((x + 0.5) / W * W - 0.5 rounds: the weights are 1, 0, 0, 0 up to an ulp or two)
        assert np.allclose(o[..., :3], img[..., :3], rtol=0, atol=2e-6)


def test_conv2d_weights_fills_its_block_from_invocations_beyond_a_small_frame():
    """the reference dispatches whole 16 x 16 workgroups (command.rs:167-168): on a 5 x 3 frame the 7 x 7 weights are written by
    invocations that lie outside the frame -- they must exist here too"""
    img = util.synthetic(5, 3, util.F32)
    o = np.zeros_like(img)
    buf = np.zeros(961, np.float32)
    host_node("conv2d_weights", {"input_image": img, "output_image": o}, {"ksize": 7, "sigma": 0.0}, {"ConvWeights": buf})
    want = np.zeros(49, np.float32)
    want[24] = 1.0      # sigma <= 0: the delta kernel
    assert np.array_equal(buf[:49], want) and np.array_equal(o, img)


CONSTRUCT_HEAD = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
"""
# body of the file -> the texel it stores at (5, 3), worked out by hand
CONSTRUCTS = {
    "nested structs": ("struct In { vec2 a; float k[2]; }; struct Out { In i; vec4 v[2]; };\nvoid main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); Out o; o.i.a = vec2(1.0, 2.0); o.i.k[1] = 3.0; "
                       "o.v[0] = vec4(o.i.a, o.i.k[1], 0.0); imageStore(output_image, p, o.v[0]); }", [1, 2, 3, 0]),
    "arrays of arrays": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); float m[2][3]; for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) m[i][j] = float(i * 3 + j); "
                         "imageStore(output_image, p, vec4(m[1][2], m[0][1], m.length(), m[0].length())); }", [5, 1, 2, 3]),
    "integer built-ins": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); ivec2 a = abs(p - 5); ivec2 b = min(a, 3); ivec2 c = max(b, ivec2(1)); int s = sign(p.x - 4); uvec2 u = uvec2(p) % 3u; "
                          "int k = clamp(p.x, 1, 4); imageStore(output_image, p, vec4(c, s + k, u.x + u.y)); }", [1, 2, 5, 2]),
    "loops": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); float s = 0.0; int i = 0; while (true) { if (i >= 8) break; ++i; if ((i & 1) == 0) continue; s += float(i); } "
              "for (;;) { s *= 0.5; if (s < 1.0) break; } do { s += 1.0; } while (s < 3.0); imageStore(output_image, p, vec4(s)); }", [3.5, 3.5, 3.5, 3.5]),
    "conversions": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); float f = -1.5 + float(p.x); int i = int(f); uint u = uint(max(i, 0)); bool b = bool(i); vec4 v = vec4(ivec4(i, u, b, 2)); "
                    "ivec3 iv = ivec3(vec3(1.7, -1.7, 2.5)); bvec2 bv = bvec2(p); imageStore(output_image, p, v + vec4(iv, float(bv.x) + float(bv.y))); }", [4, 2, 3, 4]),
    "overloads": ("float f(float x) { return x * 2.0; }\nvec2 f(vec2 x) { return x * 3.0; }\nint f(int x) { return x + 1; }\n"
                  "void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); imageStore(output_image, p, vec4(f(1.0), f(vec2(1.0)), f(p.x))); }", [2, 3, 3, 6]),
    "compound assignment": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); int a = p.x; a <<= 2; a |= 1; a ^= 6; a %= 7; a >>= 1; a &= 3; uint u = 5u; u *= 3u; u -= 1u; u /= 2u; "
                            "float f = 1.0; f /= 4.0; f -= 0.125; imageStore(output_image, p, vec4(a, u, f, !(a > 1) || (u < 3u && f > 0.0) ? 1.0 : 0.0)); }", [2, 7, 0.125, 0]),
    "built-ins with operands of different types": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); uint u = 7u; float a = max(p.x, 5.5) + min(2.5, p.y); uint b = max(p.x, u) + min(u, 9); "
                                                   "float c = clamp(p.x, 0.0, 4.5) + clamp(2.75, 0, 1) + clamp(p.y, 0u, 2.5); float d = mix(0, 10, 0.25) + pow(2, 3.0) + sqrt(p.x - 1) + step(4, 4.5) + mod(7, 4.0); "
                                                   "imageStore(output_image, p, vec4(a, b, c, d)); }", [8, 14, 8, 16.5]),
    "names that are C++'s keywords": ("struct Pick { float new; int or; };\nfloat xor(float delete, float char) { return delete - char; }\n"
                                      "void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); Pick auto = Pick(2.5, p.x); float and = xor(auto.new, 1.0); bool try = auto.or == 5; "
                                      "imageStore(output_image, p, vec4(and, auto.new, auto.or, try)); }", [1.5, 2.5, 5, 1]),
    "mixed constructors": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); vec2 a = vec2(p); vec4 b = vec4(a, 1, p.x); vec3 c = vec3(b); vec4 d = vec4(c.xy, ivec2(3, 4)); vec4 e = vec4(1u, 2, 3.0, true); "
                           "imageStore(output_image, p, b + d + e); }", [11, 8, 7, 10]),
    "array comparison": ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); int a[2] = int[](5, 2); int b[2] = int[2](p.x, 2); int c[2] = int[](p.y, 2); "
                         "imageStore(output_image, p, vec4(a == b ? 1.0 : 0.0, a != b ? 1.0 : 0.0, a == c ? 1.0 : 0.0, a != c ? 1.0 : 0.0)); }", [1, 0, 0, 1]),
    "struct comparison": ("struct In { ivec2 q; float k[2]; }; struct S { int a; In i; };\nvoid main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); S x = S(1, In(ivec2(5, 3), float[](1.0, 2.0))); "
                          "S y = S(1, In(p, float[](1.0, 2.0))); S z = y; z.i.k[1] = 2.5; imageStore(output_image, p, vec4(x == y ? 1.0 : 0.0, x != y ? 1.0 : 0.0, x == z ? 1.0 : 0.0, y != z ? 1.0 : 0.0)); }", [1, 0, 0, 1]),
    "a private global, macros with arguments, precision statements": (
        "#define SQ(x) ((x) * (x))\n#define TAPS 3\nprecision highp float;\nint counter = 0;\nconst int OFF[TAPS] = int[](-1, 0, 1);\nmat3x3 ident() { return mat3x3(1.0); }\n"
        "void bump(inout int n) { n++; counter += 2; }\n"
        "void main() { precision mediump int; ivec2 p = ivec2(gl_GlobalInvocationID.xy); int n = 0; for (int i = 0, j = 2; i < TAPS; ++i, --j) { bump(n); n += OFF[i] * OFF[j]; } mat3 m = ident(); m[1][2] = 0.5; "
        "vec3 r = m * vec3(1.0, 2.0, 4.0); switch (p.y) { case 3: r.x += 10.0; break; default: r.x = 0.0; } imageStore(output_image, p, vec4(r, SQ(n) + counter)); }", [11, 2, 5, 7]),
    "arrays are values": ("float[2] two(float x) { return float[2](x, x * 2.0); }\nvoid scribble(float w[2]) { w[0] = 100.0; }\nstruct Box { float k[2]; };\n"
                          "void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); float a[2] = two(3.0), b[2], c; b = a; b[1] += 1.0; scribble(a); c = a[0]; Box x; x.k = b; Box y = x; y.k[0] = 0.0; "
                          "float m[2][2] = float[2][2](a, b); m[0] = b; imageStore(output_image, p, vec4(c, b[1] + m[0][1], x.k[0] + y.k[0], a == two(3.0) ? 1.0 : 0.0)); }", [3, 14, 3, 1]),
    "arrays by value": ("vec4 total(vec4 v[3]) { return v[0] + v[1] + v[2]; }\nvoid second(float[2] w, out float s) { s = w[1]; }\n"
                        "void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); vec4 t[3]; for (int i = 0; i < 3; ++i) t[i] = vec4(float(i + p.x)); float s; second(float[](7.0, 9.0), s); "
                        "imageStore(output_image, p, total(t) + vec4(0, 0, 0, s)); }", [18, 18, 18, 27]),
}


@pytest.mark.parametrize("name", sorted(CONSTRUCTS))
def test_constructs_of_the_language_mean_what_glsl_says(name):
    body, want = CONSTRUCTS[name]
    img = util.synthetic(20, 9, util.F32)
    o = np.zeros_like(img)
    HostShader("construct", CONSTRUCT_HEAD + body).run({"input_image": img, "output_image": o})
    assert o[3, 5].tolist() == [float(x) for x in want], (name, o[3, 5])


def test_what_c_cannot_express_is_a_compile_error_with_the_files_line():
    """what the translation cannot carry is refused by the run-time compiler against the file's own line (never a silent difference):
    e.g. a storage block's array handed to a function as a whole (block members stay C arrays in the block's layout)"""
    body = ("layout (std430, binding = 2) readonly buffer Lut { float lut[4]; };\nfloat pick(float v[4], int i) { return v[i]; }\n"
            "void main() {\n float x = pick(lut, 1); imageStore(output_image, ivec2(0), vec4(x)); }")
    with pytest.raises(Exception) as e:
        HostShader("construct", CONSTRUCT_HEAD + body)
    assert "construct.comp:8" in str(e.value), str(e.value)[:400]
    # GLSL converts an integer vector to a float vector where the two meet; clang's vectors would REINTERPRET the bits unless told not to
    # (-flax-vector-conversions=integer, rf_jit.cpp): the file is refused and has to spell the constructor
    with pytest.raises(Exception) as e:
        HostShader("construct", CONSTRUCT_HEAD + "void main() {\n ivec2 p = ivec2(gl_GlobalInvocationID.xy);\n vec2 c = vec2(1.5) * p; imageStore(output_image, p, vec4(c, 0.0, 0.0)); }")
    assert "construct.comp:7" in str(e.value) and "cannot convert between vector values" in str(e.value), str(e.value)[:400]


HISTOGRAM = """#version 450
// luma histogram of the frame in 64 bins, the brightest code seen, and the count of invocations that ran (frame or not)
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (std430, binding = 2) buffer Hist { uint bins[64]; uint brightest; int darkest; uint invocations; uint first; };
void main()
{
    ivec2 size = imageSize(input_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    atomicAdd(invocations, 1);
    if (p.x >= size.x || p.y >= size.y) return;
    vec4 t = imageLoad(input_image, p);
    float y = clamp(dot(t.rgb, vec3(0.25, 0.5, 0.25)), 0.0, 1.0);
    uint code = uint(y * 255.0);
    atomicAdd(bins[code >> 2], 1u);
    atomicMax(brightest, code);
    atomicMin(darkest, int(code) - 300);
    atomicCompSwap(first, 0u, 7u);
    atomicOr(first, 8u);
    imageStore(output_image, p, t);
}
"""


def histogram_of(img):
    f = np.float32
    y = np.clip((img[..., 0] * f(0.25) + img[..., 1] * f(0.5)) + img[..., 2] * f(0.25), f(0), f(1))
    return (y * f(255.0)).astype(np.uint32)


def test_atomic_functions_on_a_storage_block():
    """GLSL's atomic memory functions (4.50 section 8.11) are the way invocations of a compute filter meet in a block: a histogram node"""
    W, H = 37, 23
    img = util.synthetic(W, H, util.F32)
    o = np.zeros_like(img)
    buf = np.zeros(68, np.uint32)
    HostShader("histogram", HISTOGRAM).run({"input_image": img, "output_image": o}, None, {"Hist": buf.view(np.uint8)})
    code = histogram_of(img)
    assert np.array_equal(buf[:64], np.bincount((code >> 2).ravel(), minlength=64))
    assert buf[64] == code.max() and buf[65].view(np.int32) == int(code.min()) - 300
    assert buf[66] == 48 * 32 and buf[67] == 15 and np.array_equal(o, img)      # every invocation of the 3 x 2 workgroups ran; 0 -> 7 once, then | 8
    with pytest.raises(Exception):      # float memory: not an atomic of GLSL 4.50 (a static_assert of the prelude, as glslc's type check)
        HostShader("histogram_bad", HISTOGRAM.replace("atomicAdd(bins[code >> 2], 1u);", "atomicAdd(t.x, 1.0);"))
    with pytest.raises(rf.RfError) as e:
        rf.glsl_translate("counter", HISTOGRAM.replace("void main()", "layout (binding = 5) uniform atomic_uint counter;\nvoid main()"))
    assert "counter.comp:" in str(e.value) and "atomic_uint" in str(e.value)


def test_an_integer_vector_where_a_float_vector_is_meant_is_refused_by_the_run_time_compiler_too(glsl_dir, tmp_path, monkeypatch):
    """the product's compile (hiprtc, rf_jit.cpp): the strict second parse finds what the default mode would turn into a reinterpretation of bits"""
    monkeypatch.setenv("RF_JIT_CACHE_DIR", str(tmp_path / "cache"))
    bad = GAIN.replace("vec4(c.rgb * gain + bias, c.a)", "vec4(c.rg * gain + bias, vec2(0.5) * p)")
    (glsl_dir / "scaled.comp").write_text(bad)
    with pytest.raises(rf.RfError) as e:
        rf.Plan(rf.Config("input -> ss -> output\nss: scaled { gain: 2.0 }"), rf.RF_GRAPH_GLSL_NODES).jit_compile(rf.RF_FORMAT_RGBA32F)
    assert "scaled.comp:12" in str(e.value) and "cannot convert between vector values" in str(e.value) and "vec2(p)" in str(e.value)
    (glsl_dir / "scaled.comp").write_text(bad.replace("vec2(0.5) * p", "vec2(0.5) * vec2(p)"))
    rf.Plan(rf.Config("input -> ss -> output\nss: scaled { gain: 2.0 }"), rf.RF_GRAPH_GLSL_NODES).jit_compile(rf.RF_FORMAT_RGBA32F)


# ---- planning and the gfx950 code objects ---------------------------------------------------------------------------------------------
def test_a_type_that_is_a_comp_file_plans_as_a_node_of_its_own(glsl_dir):
    for t in ("gaussian5", "colour_grade", "sharpen"):
        shutil.copy(os.path.join(SHADERS, t + ".comp"), glsl_dir / (t + ".comp"))
    assert rf.Plan(rf.Config(util.CHAIN3)).launches() == ["blur", "grade+sharp"]         # the FILES are the types (the reference's rule): gaussian5.comp a node
    assert rf.Plan(rf.Config(util.CHAIN3), rf.RF_GRAPH_GLSL_NODES).launches() == ["blur", "grade", "sharp"]      # (colour_grade.comp and sharpen.comp: row stages, fused)
    rf.set_type_lookup(False)
    assert rf.Plan(rf.Config(util.CHAIN3)).launches() == ["blur+grade+sharp"]             # default: the hand-written kernels, fused
    (glsl_dir / "wobble.comp").write_text(text_of("invert").replace("strength", "depth"))
    p = rf.Plan(rf.Config("input -> ww -> output\nww: wobble { enabled: true, depth: 0.5 }"))      # a name the registry lacks: found as a file either way
    assert p.launches() == ["ww"] and p.needs_jit() == [True]


def test_storage_blocks_of_a_shader_are_wired_by_block_type_name(glsl_dir):
    for t in ("conv2d", "conv2d_weights"):
        shutil.copy(os.path.join(SHADERS, t + ".comp"), glsl_dir / (t + ".comp"))
    p = rf.Plan(rf.Config("input -> kw -> cv -> output\nkw:ConvWeights -> cv:ConvWeights\nkw: conv2d_weights { ksize: 5, sigma: 1.0 }\ncv: conv2d { ksize: 5 }"))
    assert p.buffers() == {"kw:ConvWeights": 961 * 4}
    with pytest.raises(rf.RfError) as e:      # a block the shader only reads must be wired
        rf.Plan(rf.Config("input -> cv -> output\ncv: conv2d { ksize: 5 }")).halo_schedule()
    assert "needs a storage buffer wired to ConvWeights" in str(e.value)
    with pytest.raises(rf.RfError) as e:      # ... and on the side the shader's qualifier allows
        rf.Plan(rf.Config("input -> cv -> kw -> output\ncv:ConvWeights -> kw:ConvWeights\nkw: conv2d_weights { ksize: 5 }\ncv: conv2d { ksize: 5 }")).halo_schedule()
    assert "ConvWeights is the buffer conv2d" in str(e.value) and "the graph wires it as an" in str(e.value)


@pytest.mark.parametrize("t", COMP)
def test_every_shipped_shader_compiles_for_gfx950(glsl_dir, t):
    shutil.copy(os.path.join(SHADERS, t + ".comp"), glsl_dir / (t + ".comp"))
    r = rf.glsl_reflect(t, text_of(t))
    ins = [im["name"] for im in r["images"] if not im["writeonly"]]
    outs = [im["name"] for im in r["images"] if not im["readonly"]]
    lines = ["input -> nn:%s" % n for n in ins if n not in outs] + ["input -> nn:%s" % n for n in ins if n in outs]
    lines += ["nn:%s -> output" % outs[0]]
    for b in r["storage_blocks"]:
        if b["readonly"]:
            shutil.copy(os.path.join(SHADERS, "conv2d_weights.comp"), glsl_dir / "conv2d_weights.comp")
            lines = ["input -> kw -> nn:%s" % ins[0], "kw:%s -> nn:%s" % (b["type_name"], b["type_name"]), "nn:%s -> output" % outs[0], "kw: conv2d_weights { ksize: 3 }"]
    text = "\n".join(lines + ["nn: %s {}" % t])
    p = rf.Plan(rf.Config(text))
    assert p.jit_compile(rf.RF_FORMAT_RGBA32F) > 2048 and p.jit_compile(rf.RF_FORMAT_RGBA8) > 2048


# ---- the code objects: what makes a many-tap shader fast is that its loads are not waited for one by one ------------------------------------------
@pytest.mark.parametrize("compiler", ["this process", "a process that imported torch first"])
def test_glsl_kernels_have_no_scratch_and_batch_their_loads(tmp_path, compiler):
    """imageLoad is branch-free (rf_glsl_dev.h): a 25-tap shader keeps several loads in flight (counted waits) and drains the queue a handful of times -- under a
    bounds branch per load every load was waited for before the next one issued (profiles/EXPERIMENTS.md R4.7).  Checked on the
    disassembled code objects of both compiler builds a process may get (its own libhiprtc, or the one PyTorch bundles)."""
    import glob
    import subprocess
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_obj
    cmd = [_sys.executable, os.path.join(ROOT, "tests", "glsl_isa_compile.py"), str(tmp_path)] + (["torch"] if "torch" in compiler else [])
    if "torch" in compiler:
        import importlib.util
        if importlib.util.find_spec("torch") is None:
            pytest.skip("no PyTorch here")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "compiled with" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    files = sorted(glob.glob(str(tmp_path / "*.hsaco")))
    assert len(files) >= 10, files      # five graphs x two formats: a generic kernel each (colour_grade: a fused row stage), and the window kernels of gaussian5 / local_contrast
    many = windows = 0
    for f in files:
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True, check=True).stdout
        assert ".private_segment_fixed_size: 0" in notes, "a GLSL kernel uses scratch: " + f
        for name, ins in isa_obj.functions(f).items():
            if "user_node_kernel" in name:      # a window kernel: its taps are LDS reads at constant offsets -- no address arithmetic per tap survives
                windows += 1
                assert sum(1 for i in ins if i.op.startswith("ds_read")) >= 10 and not any(i.op.startswith(("global_load_dwordx4", "buffer_load")) for i in ins[len(ins) // 3:]), name
            if "glsl_node_kernel" not in name:
                continue
            loads = [i for i in ins if i.op.startswith(("global_load", "buffer_load"))]
            waits = [i for i in ins if i.op == "s_waitcnt" and "vmcnt(0)" in i.text]      # full drains (a counted wait leaves younger loads in flight)
            assert not any(i.op.startswith(("scratch_", "s_barrier")) for i in ins), name
            if len(loads) >= 20:      # gaussian5 (25 loads) and local_contrast (26): a handful of waits, not one per load
                many += 1
                assert len(waits) <= len(loads) // 3, (name, len(loads), len(waits))
    assert many == 4 and windows == 6, (many, windows)      # windows: gaussian5, local_contrast, sharpen x two formats
