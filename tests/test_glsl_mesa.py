"""shaders/*.comp and the translator's reading of GLSL against an INDEPENDENT GLSL implementation: Mesa's GLSL 4.50 compiler with its
llvmpipe CPU back end (tests/mesa_glsl.py, tests/native/mesa_glsl.c; the driver ships in the image, no X server is needed).

The reference compiles a filter file with shaderc and runs it on a Vulkan device (src/vulkan/shader.rs:73-93, command.rs:166-194); neither
exists here, and until this file the text of shaders/*.comp had only ever been read by rf_glsl.cpp.  What is pinned:
  * every shipped .comp is GLSL 4.50 that a GLSL compiler accepts;
  * on rgba32f images Mesa's result and the translation's are the SAME BITS for every shipped shader, once fma() is evaluated as
    llvmpipe evaluates it (a * b + c, two roundings: HostShader(split_fma=True)) -- so the only difference between Mesa and the product
    is the ONE rounding of fma() that DESIGN.md section 3 specifies (what GPUs with fused hardware do), and it stays within a few ulp;
  * the language-construct table of tests/test_glsl.py (hand-worked expectations), storage blocks, atomics, shared memory and barrier():
    Mesa computes what the table and the numpy models say;
  * rgba8 images: GL leaves the UNORM8 conversions to the implementation, so codes may differ by one (asserted: never more, and rarely)."""
import os
import sys

import numpy as np
import pytest

import reforge_amd as rf
from oracle import graph as ograph
from oracle import pixel
from tests import util
from tests.glsl_host import HostShader
from tests.mesa_glsl import MesaCompileError, MesaShader, runner, version, why_not

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import glsl_weights  # noqa: E402

@pytest.fixture(autouse=True, scope="module")
def mesa_or_skip():
    """(decided when the first test of this file runs, not when it is collected: a `-m gpu` session never builds or starts the runner)"""
    if runner() is None:
        pytest.skip("Mesa's software rasteriser is not usable here: " + why_not())


def text_of(t):
    with open(os.path.join(SHADERS, t + ".comp")) as f:
        return f.read()


def gparams(sigma, radius):
    return dict({"sigma": sigma}, **{"w%d" % i: w for i, w in enumerate(glsl_weights.weights(sigma, radius))})


def cases(fmt):
    """(type, images, params, buffers) for every shipped shader"""
    W, H = 37, 23
    img = util.synthetic(W, H, fmt)
    other = util.synthetic(W, H, fmt, seed=77)
    z = lambda: np.zeros_like(img)      # noqa: E731
    K = 7
    w = ograph.default_conv_weights(K, 1.5).astype(np.float32)
    buf = np.zeros(961, np.float32)
    buf[:K * K] = w.ravel()
    blurred = pixel.gaussian(img, 4, sigma=2.0)
    return [
        ("gaussian5", {"input_image": img, "output_image": z()}, gparams(1.0, 2), None),
        ("gaussian9", {"input_image": img, "output_image": z()}, gparams(2.0, 4), None),
        ("gaussian", {"input_image": img, "output_image": z()}, dict(gparams(2.5, 7), radius=7), None),
        ("colour_grade", {"input_image": img, "output_image": z()}, {"slope": 1.1, "offset": -0.02, "saturation": 1.2}, None),
        ("colour_grade_inplace", {"image": img.copy()}, {"slope": 0.9, "offset": 0.03, "saturation": 0.4}, None),
        ("sharpen", {"input_image": img, "output_image": z()}, {"amount": 0.75}, None),
        ("combination", {"input_image0": img, "input_image1": other, "output_image": z()}, {"mix": 0.3}, None),
        ("split_luma", {"input_image": img, "luma_image": z(), "chroma_image": z()}, None, None),
        ("conv2d", {"input_image": img, "output_image": z()}, {"ksize": K}, {"ConvWeights": buf}),
        ("invert", {"input_image": img, "output_image": z()}, {"enabled": 1, "strength": 0.7}, None),
        ("edge_detect", {"input_image": img, "output_image": z()}, {"scale": 1.5}, None),
        ("local_contrast", {"input_image": img, "output_image": z()}, {"amount": 0.8}, None),
        ("pulse", {"input_image": img, "output_image": z()}, {"_rf_time": 1234.0}, None),
        ("unsharp_mask", {"input_image": img, "blurred_image": blurred, "output_image": z(), "mask_image": z()}, {"amount": 1.5, "threshold": 0.02}, None),
    ]


def both_ways(t, text, images, params, buffers):
    """{image name: (Mesa's texels, the translation's with fma() split)} for every image the shader may write"""
    written = [i["name"] for i in rf.glsl_reflect(t, text)["images"] if not i["readonly"] and i["name"] in images]
    mesa = MesaShader(t, text).run({k: v.copy() for k, v in images.items()}, params, None if buffers is None else {k: v.copy() for k, v in buffers.items()})
    ours = {k: v.copy() for k, v in images.items()}
    HostShader(t, text, split_fma=True).run(ours, params, None if buffers is None else {k: v.copy() for k, v in buffers.items()})
    return {k: (mesa[k], ours[k]) for k in written}


def test_mesa_is_what_it_says():
    assert "llvmpipe" in version() and "GLSL 4." in version(), version()


def test_every_shipped_shader_is_covered():
    shipped = sorted(f[:-5] for f in os.listdir(SHADERS) if f.endswith(".comp"))
    assert sorted(set(c[0] for c in cases(util.F32)) | {"conv2d_weights"}) == shipped      # conv2d_weights: its own test below (it calls exp())


def test_mesa_and_the_translation_compute_the_same_bits_on_rgba32f():
    for t, images, params, buffers in cases(util.F32):
        for name, (mesa, ours) in both_ways(t, text_of(t), images, params, buffers).items():
            util.assert_same(mesa, ours, "%s.comp %s: Mesa llvmpipe vs rf_glsl.cpp's translation (fma split on both sides)" % (t, name))


def test_the_one_rounding_of_fma_is_all_that_separates_mesa_from_the_oracle():
    """the product (and the oracle) evaluate fma() with ONE rounding; llvmpipe with two.  edge_detect's fmas multiply by 2 (exact either way):
    there Mesa gives the oracle's bits.  Everywhere else the difference is a few ulp of the values involved."""
    old = util.register_user_types()
    try:
        img = util.synthetic(37, 23, util.F32)
        want = util.run_oracle("input -> nn -> output\nnn: edge_detect { scale: 1.5 }", img)
    finally:
        rf.set_shader_path(old)
    got = MesaShader("edge_detect", text_of("edge_detect")).run({"input_image": img, "output_image": np.zeros_like(img)}, {"scale": 1.5})["output_image"]
    util.assert_same(got, want, "edge_detect.comp on Mesa vs the oracle")
    for t, oracle in (("gaussian9", lambda: pixel.gaussian(img, 4, sigma=2.0)), ("sharpen", lambda: pixel.sharpen(img, 0.75)), ("colour_grade", lambda: pixel.colour_grade(img, 1.1, -0.02, 1.2))):
        c = next(c for c in cases(util.F32) if c[0] == t)
        got = MesaShader(t, text_of(t)).run(c[1], c[2])["output_image"]
        assert np.abs(got - oracle()).max() < 2e-6, t      # values of order 1: a few ulp


def test_unorm8_conversions_store_rounds_ties_to_even_load_is_the_implementations():
    """what an rgba8 image means to imageLoad / imageStore.  STORE: Mesa writes round-to-nearest-EVEN of clamp(v) * 255 -- on every tie
    (k + 0.5) / 255 and next to it: this library's rule, which had been its own choice until here.  LOAD: Mesa
    multiplies the code by fl(1 / 255), which is the correctly rounded c / 255 for 130 codes of 256; this library divides (what texture
    units do) -- the ONE other documented difference next to fma()"""
    head = CONSTRUCT_HEAD.replace("rgba32f", "rgba8")
    load = head + ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); uint b = floatBitsToUint(imageLoad(input_image, p).x); "
                   "imageStore(output_image, p, vec4(float(b & 255u), float((b >> 8) & 255u), float((b >> 16) & 255u), float(b >> 24)) / 255.0); }")
    img = np.zeros((16, 16, 4), np.uint8)
    img[..., 0] = np.arange(256).reshape(16, 16)
    out = MesaShader("load", load).run({"input_image": img, "output_image": np.zeros_like(img)})["output_image"].astype(np.uint32)
    got = (out[..., 0] | (out[..., 1] << 8) | (out[..., 2] << 16) | (out[..., 3] << 24)).reshape(-1).view(np.float32)      # the float imageLoad gave, byte by byte
    codes = np.arange(256)
    assert np.array_equal(got, codes.astype(np.float32) * np.float32(1.0 / 255.0))
    exact = (codes.astype(np.float64) / 255.0).astype(np.float32)
    assert (got == exact).sum() == 130 and np.abs(got.view(np.int32) - exact.view(np.int32)).max() == 1
    store = head + ("layout (binding = 2) uniform Params { float delta; };\nvoid main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); float k = float(p.y * 16 + p.x); "
                    "imageStore(output_image, p, vec4((k + 0.5 + delta) / 255.0, -k, k, 0.0)); }")
    for delta in (0.0, -0.01, 0.01):
        out = MesaShader("store", store).run({"input_image": img, "output_image": np.zeros_like(img)}, {"delta": delta})["output_image"]
        v = (codes.astype(np.float32) + np.float32(0.5) + np.float32(delta)) / np.float32(255.0)
        want = np.clip(np.rint(np.clip(v, 0, 1) * np.float32(255.0)), 0, 255).astype(np.uint8)      # np.rint: ties to even
        assert np.array_equal(out[..., 0].reshape(-1), want), delta
        assert not out[..., 1].any() and (out[..., 2].reshape(-1)[1:] == 255).all()      # clamps to [0, 1] first


def test_rgba8_images_give_the_same_codes_once_the_load_is_mesas():
    """every shipped shader on rgba8 images: Mesa and the translation -- fma() split, texels decoded by the reciprocal multiplication, both as
    llvmpipe does -- give the SAME CODES; with this library's own two choices the codes differ by at most one, in a few percent of texels"""
    for t, images, params, buffers in cases(util.U8):
        for name, (mesa, ours) in both_ways(t, text_of(t), images, params, buffers).items():
            util.assert_same(mesa, ours, "%s.comp %s on rgba8" % (t, name))
        if t in ("sharpen", "gaussian9", "colour_grade"):
            spec = {k: v.copy() for k, v in images.items()}
            HostShader(t, text_of(t)).run(spec, params)
            d = np.abs(spec["output_image"].astype(int) - mesa.astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 0.08, (t, d.max())


def test_conv2d_weights_fills_the_same_block_up_to_exp():
    buf_m, buf_o = np.zeros(961, np.float32), np.zeros(961, np.float32)
    img = util.synthetic(20, 9, util.F32)
    MesaShader("conv2d_weights", text_of("conv2d_weights")).run({"input_image": img, "output_image": np.zeros_like(img)}, {"ksize": 7, "sigma": 1.5}, {"ConvWeights": buf_m})
    HostShader("conv2d_weights", text_of("conv2d_weights"), split_fma=True).run({"input_image": img, "output_image": np.zeros_like(img)}, {"ksize": 7, "sigma": 1.5}, {"ConvWeights": buf_o})
    assert np.abs(buf_m - buf_o).max() < 1e-7 and abs(float(buf_m[:49].sum()) - 1.0) < 1e-6 and not buf_m[49:].any()      # exp() is each implementation's own


# ---- the language: the hand-worked construct table of tests/test_glsl.py, computed by Mesa ----------------------------------------------------
from tests.test_glsl import CONSTRUCT_HEAD, CONSTRUCTS, HISTOGRAM, histogram_of  # noqa: E402

# what GLSL 4.50 itself does not allow (this library's translation is more permissive there, never different)
NOT_GLSL_450 = set()


@pytest.mark.parametrize("name", sorted(CONSTRUCTS))
def test_mesa_computes_what_the_construct_table_says(name):
    body, want = CONSTRUCTS[name]
    img = util.synthetic(20, 9, util.F32)
    try:
        got = MesaShader("construct", CONSTRUCT_HEAD + body).run({"input_image": img, "output_image": np.zeros_like(img)})["output_image"]
    except MesaCompileError as e:
        if name in NOT_GLSL_450:
            pytest.skip("not GLSL 4.50: " + str(e)[-200:])
        raise
    assert got[3, 5].tolist() == [float(x) for x in want], (name, got[3, 5])


@pytest.mark.parametrize("first", [0, 12, 24])
def test_generated_shaders_give_the_same_bits_on_mesa_and_through_the_translation(first):
    """differential runs (tests/glsl_gen.py): typed random programs over operators by precedence, constructors, swizzles, conversions, integer
    and bit built-ins, exactly rounded float built-ins, control flow, arrays and structs as values.  Every float is `precise` -- GLSL lets an
    implementation contract a * b + c elsewhere, and Mesa does where this library never does.  scripts/fuzz_glsl_mesa.py: the long campaign."""
    from tests.glsl_gen import generate
    img = util.synthetic(37, 23, util.F32)
    for seed in range(first, first + 12):
        text = generate(seed, 18)
        block_mesa, block_ours = np.full(10, 7, np.uint32), np.full(10, 7, np.uint32)      # Stats: what the atomic functions of the program leave
        mesa = MesaShader("generated", text).run({"input_image": img, "output_image": np.zeros_like(img)}, {"gain": 1.5, "shift": 3}, {"Stats": block_mesa})["output_image"]
        ours = np.zeros_like(img)
        HostShader("generated", text, split_fma=True).run({"input_image": img, "output_image": ours}, {"gain": 1.5, "shift": 3}, {"Stats": block_ours.view(np.uint8)})
        assert np.isfinite(mesa).all() and np.array_equal(block_mesa, block_ours), (seed, block_mesa, block_ours)
        util.assert_same(mesa, ours, "generated shader %d: Mesa vs the translation\n%s" % (seed, text))


def test_vector_and_matrix_equality_and_the_pack_and_bit_built_ins_on_mesa():
    """tests/test_glsl.py's EQUALITY shader (== on vectors and matrices is ONE bool, a ternary between && and ==, .length(), packUnorm4x8, findMSB,
    bitfieldExtract) -- without its specialisation constant, which is Vulkan's GLSL"""
    from tests.test_glsl import EQUALITY, equality
    text = EQUALITY.replace("layout (constant_id = 3) const int MODE = 2;", "const int MODE = 2;")
    img = util.synthetic(37, 21, util.F32)
    img[3, 5, :3] = 0.0
    img[4, 6, 0] = img[4, 6, 1]
    mesa = MesaShader("equality", text).run({"image": img.copy()})["image"]
    ours = img.copy()
    HostShader("equality", text).run({"image": ours})
    util.assert_same(mesa, ours, "EQUALITY: Mesa vs the translation")
    util.assert_same(mesa, equality(img), "EQUALITY: Mesa vs the numpy model")


def test_the_shader_texts_of_the_recognition_tests_are_glsl_and_run_the_same():
    """GAIN / BOX5 and their near-misses (tests/test_glsl.py: what is and is not a point shader, a translation-invariant stencil): each is
    GLSL 4.50 to Mesa and gives the translation's bits -- whichever kernel form the product then picks for it"""
    from tests.test_glsl import BOX5, GAIN, NOT_POINT, NOT_STENCIL
    img = util.synthetic(40, 19, util.F32)
    texts = dict({"GAIN": GAIN, "BOX5": BOX5}, **{"not point: " + k: v for k, v in NOT_POINT.items()}, **{"not stencil: " + k: v for k, v in NOT_STENCIL.items()})
    for name, text in texts.items():
        try:
            rf.glsl_translate("recognition", text)
        except rf.RfError:
            continue      # a text the translator refuses (a stencil in place): nothing to compare
        images = {i["name"]: (img if i["readonly"] else np.zeros_like(img)) for i in rf.glsl_reflect("t", text)["images"]}
        for image_name, (mesa, ours) in both_ways("recognition", text, images, {"gain": 1.5, "bias": 0.25, "reach": 1}, None).items():
            d = np.abs(mesa.astype(np.float64) - ours.astype(np.float64)).max()
            assert d <= 4e-6, (name, image_name, d)      # (sums that are not `precise`: Mesa may contract or re-associate them)


def graph_texts():
    g5 = lambda sigma: "gaussian5 { sigma: %s, %s }" % (sigma, glsl_weights.as_params(sigma, 2))      # noqa: E731 -- (weights as parameters: the files' own exp() is each implementation's)
    g9 = "gaussian9 { sigma: 2.0, %s }" % glsl_weights.as_params(2.0, 4)
    return {
        "BASELINE configs[1]: 3-stage chain": util.CHAIN3.replace("gaussian5    { sigma: 1.0 }", g5(1.0)),
        "BASELINE configs[3]: 5-stage chain": util.CHAIN5.replace("gaussian5    { sigma: 1.0 }", g5(1.0)).replace("gaussian9    { sigma: 2.0 }", g9),
        "fork / join": util.DIAMOND.replace("gaussian5   { sigma: 1.5 }", g5(1.5)),
        "two output images": util.SPLIT2.replace("gaussian5 { sigma: 1.0 }", g5(1.0)),
        "a storage buffer edge": "input -> kw -> cv -> output\nkw:ConvWeights -> cv:ConvWeights\nkw: conv2d_weights { ksize: 7, sigma: 1.5 }\ncv: conv2d { ksize: 7 }",
        "a point node in place": "input -> aa -> cg:image -> bb -> output\naa: sharpen { amount: 0.5 }\nbb: sharpen { amount: 0.25 }\ncg: colour_grade_inplace { slope: 0.9, offset: 0.03, saturation: 0.4 }",
    }


@pytest.mark.parametrize("name", sorted(graph_texts()))
def test_whole_graphs_run_the_references_way_with_the_filter_files_as_kernels(name):
    """the closest thing to a run of the reference's executor that this image allows: the reference's planning (restated) + one dispatch of the
    node's GLSL file per node on an independent GLSL implementation.  Mesa and the translation give the same frame (fma() split on both
    sides; the storage-buffer graph up to exp()); the oracle -- one rounding per fma() -- is within 1e-5 of it"""
    from tests.mesa_glsl import FileGraph
    text = graph_texts()[name]
    img = util.synthetic(61, 35, util.F32)
    mesa = FileGraph(text, img, "mesa", SHADERS).result
    ours = FileGraph(text, img, "host", SHADERS).result
    if "storage buffer" in name:
        assert np.abs(mesa - ours).max() < 1e-6
    else:
        util.assert_same(mesa, ours, name + ": Mesa vs the translation, node by node through the graph")
    assert np.abs(mesa - util.run_oracle(text, img)).max() < 1e-5, name


CORNERS = {
    "negative division and shifts": "int a = -7 - p.x; int b = 2 + (p.y & 1); imageStore(output_image, p, vec4(a / b, (-a) / -b, a >> 1, (a >> 31) & 7));",
    "uint / int casts": "uint u = uint(-1 - p.x); int i = int(3000000000u + uint(p.x)); imageStore(output_image, p, vec4(float(u >> 8), float(i >> 8), float(uint(i) >> 30), float(u % 7u)));",
    "bool conversions": "bool b = bool(float(p.x) - 5.0); bool c = bool(p.y & 2); imageStore(output_image, p, vec4(b, c, float(b) + float(c), int(b) * 3));",
    "mod and fract of negatives": "precise float x = -2.75 + float(p.x) * 0.5; imageStore(output_image, p, vec4(mod(x, 1.5), fract(x), floor(x), x - 1.5 * floor(x / 1.5)));",
    "the rounding family": "precise float x = -2.5 + float(p.x) * 0.5; imageStore(output_image, p, vec4(roundEven(x), trunc(x), ceil(x), sign(x)));",
    "smoothstep step mix": "precise float x = float(p.x) * 0.125; precise vec4 o = vec4(smoothstep(0.25, 0.75, x), step(0.5, x), mix(2.0, 6.0, x), mix(1.0, 3.0, x > 0.5)); imageStore(output_image, p, o);",
    "integer vectors": "ivec3 a = ivec3(p, p.x - p.y) * ivec3(3, -2, 5); ivec3 b = a / ivec3(2, 3, 4) + (a & ivec3(6)) - (a >> ivec3(1)); imageStore(output_image, p, vec4(b, dot(vec3(b), vec3(1.0))));",
    "uint vectors wrap": "uvec2 u = uvec2(p) * 4000000000u + uvec2(17u); imageStore(output_image, p, vec4(vec2(u >> 16u), vec2(u & 65535u)));",
    "matrix arithmetic": "precise mat3 m = mat3(vec3(p, 1), vec3(0.5, 2.0, -1.0), vec3(1.5, 0.25, 3.0)); precise mat3 n = m * transpose(m); precise vec3 v = n * vec3(1.0, -1.0, 0.5); "
                         "precise vec3 w = vec3(0.5, 1.0, 2.0) * m; imageStore(output_image, p, vec4(v + w, determinant(mat2(m))));",
}


@pytest.mark.parametrize("name", sorted(CORNERS))
def test_corner_semantics_agree_with_mesa(name):
    """where C++ and GLSL could part: division and shifts of negative integers, wrapping conversions, bool(), mod / fract of negatives, ties in
    roundEven, matrix products by columns.  (What does NOT have to agree and does not: the sign of min(0, -0), and everything an implementation
    approximates -- normalize, inversesqrt, pow, exp / log, the trigonometric functions: Mesa's pow(1.5, 2.0) is 2.2500002.)"""
    text = CONSTRUCT_HEAD + "void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); " + CORNERS[name] + " }"
    img = util.synthetic(20, 9, util.F32)
    mesa = MesaShader("corner", text).run({"input_image": img, "output_image": np.zeros_like(img)})["output_image"]
    ours = np.zeros_like(img)
    HostShader("corner", text, split_fma=True).run({"input_image": img, "output_image": ours})
    util.assert_same(mesa, ours, name)


def test_block_layouts_are_mesas():
    """std140 / std430 as rf_glsl_reflect computes them (what spirv-reflect gives the reference: offsets, array strides, padded sizes --
    pipeline_graph.rs:163, :276-292) against the layout Mesa's linker reports through the program interface queries"""
    from tests.mesa_glsl import mesa_layout
    text = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform image2D image;
layout (std140, binding = 1) uniform Params { float a; vec3 b; float c; vec2 d; int e; float f[3]; mat3 m; vec4 g; bool h; mat2 n; uvec3 u; float tail; };
layout (std430, binding = 2) buffer Data { float x; vec3 y; float z[5]; vec2 w[3]; mat3 q; ivec4 r; vec3 t[2]; uint last; } data;
void main() { imageStore(image, ivec2(0), vec4(a + c + d.x + float(e) + f[1] + m[1].x + g.x + float(h) + n[1].y + float(u.x) + tail + b.x) + vec4(data.x + data.y.x + data.z[2] + data.w[1].x + data.q[2].z + float(data.r.w) + data.t[1].z + float(data.last))); }
"""
    members, blocks = mesa_layout(text)
    r = rf.glsl_reflect("layout", text)
    seen = 0
    for kind, blks in (("uniform", r["uniform_blocks"]), ("storage", r["storage_blocks"])):
        for blk in blks:
            assert blocks[(kind, blk["type_name"])] == (blk["binding"], blk["bytes"]), (blk["type_name"], blocks)
            for m in blk["members"]:
                name = (blk["type_name"] + "." if kind == "storage" else "") + m["name"] + ("[0]" if m["dims"] else "")
                offset, array_stride, matrix_stride = members[(kind, name)]
                assert offset == m["offset"], (name, offset, m)
                if m["dims"]:
                    assert array_stride == m["stride"], (name, array_stride, m)
                if m["cols"] > 1:
                    assert matrix_stride * m["cols"] == m["bytes"], (name, matrix_stride, m)
                seen += 1
    assert seen == 20 == len(members)


def test_logical_xor_binds_between_and_and_or():
    body = ("void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); bool a = p.x > 2, b = p.y > 2, c = p.x == 5; "
            "bool r0 = a ^^ b; bool r1 = a && b ^^ c && a; bool r2 = a ^^ b || c; bool r3 = a ^^ b ^^ c; bool r4 = p.x == 5 ^^ p.y != 3; "
            "imageStore(output_image, p, vec4(r0, r1, r2, r3) + vec4(r4 ? 0.5 : 0.0)); }")
    img = util.synthetic(20, 9, util.F32)
    mesa = MesaShader("xor", CONSTRUCT_HEAD + body).run({"input_image": img, "output_image": np.zeros_like(img)})["output_image"]
    ours = np.zeros_like(img)
    HostShader("xor", CONSTRUCT_HEAD + body).run({"input_image": img, "output_image": ours})
    util.assert_same(mesa, ours, "^^")
    assert len(np.unique(mesa.reshape(-1, 4), axis=0)) >= 5      # the frame exercises several truth assignments


def test_atomic_functions_on_a_storage_block_on_mesa():
    W, H = 37, 23
    img = util.synthetic(W, H, util.F32)
    buf = np.zeros(68, np.uint32)
    out = MesaShader("histogram", HISTOGRAM).run({"input_image": img, "output_image": np.zeros_like(img)}, None, {"Hist": buf})["output_image"]
    code = histogram_of(img)
    assert np.array_equal(buf[:64], np.bincount((code >> 2).ravel(), minlength=64))
    assert buf[64] == code.max() and buf[65].view(np.int32) == int(code.min()) - 300
    assert buf[66] == 48 * 32 and buf[67] == 15 and np.array_equal(out, img)


def test_shared_memory_barrier_and_the_kitchen_sink_on_mesa():
    from tests.test_gpu_glsl import HIST_APPLY, HIST_SHARED, KITCHEN, KITCHEN_FILL, TILE_BLUR, box3, hist_apply, kitchen
    img = util.synthetic(70, 37, util.F32)
    z = np.zeros_like(img)
    got = MesaShader("tile_blur", TILE_BLUR).run({"input_image": img, "output_image": z.copy()})["output_image"]
    util.assert_same(got, box3(img), "tile_blur (shared memory, barrier) on Mesa vs the numpy model the GPU test uses")
    # a histogram per workgroup in shared memory, flushed with atomics; the node that reads the block
    hist = np.zeros(66, np.uint32)
    MesaShader("histogram", HIST_SHARED).run({"input_image": img, "output_image": z.copy()}, None, {"Hist": hist})
    got = MesaShader("hist_apply", HIST_APPLY).run({"input_image": img, "output_image": z.copy()}, None, {"Hist": hist})["output_image"]
    util.assert_same(got, hist_apply(img), "histogram -> hist_apply on Mesa vs the numpy model")
    # structs, matrices, out parameters, array parameters, swizzles, bool / uint uniforms, two storage blocks, a second input image
    other = util.run_oracle("input -> gg -> output\ngg: colour_grade { slope: 0.5, offset: 0.25, saturation: 1.0 }", img)
    lut = np.zeros(12, np.float32)
    MesaShader("kitchen_fill", KITCHEN_FILL).run({"input_image": img, "output_image": z.copy()}, None, {"Lut": lut})
    assert lut.tolist() == [0.125 * i for i in range(8)] + [1.0, 2.0, 3.0, 0.5]
    for flip in (0, 1):
        got = MesaShader("kitchen", KITCHEN).run({"input_image": img, "other_image": other, "output_image": z.copy()}, {"gain": 1.5, "shift": 3, "flip": flip, "mask": 5, "bias": 0.25},
                                                 {"Lut": lut, "Stats": np.zeros(4, np.uint32)})["output_image"]
        util.assert_same(got, kitchen(img, other, 1.5, 3, bool(flip), 5, 0.25), "kitchen flip=%d on Mesa vs the numpy model" % flip)


@pytest.mark.parametrize("params", [(0.0, 0.0, 1.0), (0.37, -1.25, 1.0), (3.5, 40.0, 0.75)])
def test_the_graphs_sampler_filters_as_a_gl_sampler_with_the_same_parameters(params):
    """texture() / texelFetch() / textureSize() through reforge's one sampler -- LINEAR, U clamp-to-edge, V REPEAT (vkutils.rs:358-365: the
    second address mode is never set) -- against a GL sampler object set up the same way: the same texels with the same weights up to the
    filter's own arithmetic (llvmpipe's lerp rounds differently: 1e-6 of values of order 1; one code on rgba8)"""
    from tests.test_glsl import RESAMPLE, resample
    img = util.synthetic(45, 23, util.F32)
    run = lambda text, im: MesaShader("resample", text).run({"source": im, "output_image": np.zeros_like(im)}, {"shift_x": params[0], "shift_y": params[1], "zoom": params[2]})["output_image"]      # noqa: E731
    got = run(RESAMPLE, img)
    assert np.abs(got - resample(img, *params)).max() < 4e-6
    assert np.array_equal(got[..., 3], img[:, ::-1, 3])      # texelFetch: exact
    img8 = util.synthetic(45, 23, util.U8)
    got8 = run(RESAMPLE.replace("rgba32f", "rgba8"), img8)
    want8 = np.clip(np.rint(np.clip(resample(img8, *params), 0, 1) * 255), 0, 255).astype(np.uint8)
    assert np.abs(got8.astype(int) - want8.astype(int)).max() <= 1


def test_the_references_own_kernel_compiles_on_mesa_and_runs_like_its_translation():
    """shaders/passthrough.comp of the reference (its one kernel; read where it lies, never copied): its qualifier says rgba8 and the
    reference runs it on the graph's format (main.rs:60) -- here on both"""
    path = "/root/reference/shaders/passthrough.comp"
    if not os.path.exists(path):
        pytest.skip("the reference is not on this machine")
    with open(path) as f:
        text = f.read()
    for fmt in (util.F32, util.U8):
        img = util.synthetic(37, 23, fmt)
        (mesa, ours), = both_ways("passthrough", text, {"input_image": img, "output_image": np.zeros_like(img)}, None, None).values()
        util.assert_same(mesa, ours, "the reference's passthrough.comp")
        util.assert_same(mesa, img, "... is the identity")


def test_the_dispatch_and_the_frame_edges_are_the_references():
    """ceil(W/16) x ceil(H/16) workgroups whatever local_size the file declares (command.rs:167-168): a file with 8 x 4 covers part of the
    frame; loads outside the frame give zero and stores outside are dropped (GL's rule for images; what rf_glsl_dev.h does)"""
    text = CONSTRUCT_HEAD.replace("local_size_x = 16, local_size_y = 16", "local_size_x = 8, local_size_y = 4") + (
        "void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); precise vec4 a = imageLoad(input_image, p - ivec2(3, 2)) + imageLoad(input_image, p + ivec2(30, 20)); "
        "a = a + vec4(1.0); imageStore(output_image, p + ivec2(2, 1), a); }")      # (precise: Mesa re-associates sums that are not)
    img = util.synthetic(37, 23, util.F32)
    (mesa, ours), = both_ways("partial", text, {"input_image": img, "output_image": np.zeros_like(img)}, None, None).values()
    util.assert_same(mesa, ours, "partial coverage, out-of-frame loads and stores")
    assert (mesa[:, :, 0] != 0).sum() == 24 * 8      # 3 x 2 workgroups of 8 x 4 invocations, each storing inside the frame


def test_what_mesa_refuses_the_translator_refuses_or_the_run_time_compiler_does():
    """files outside GLSL 4.50: Mesa's compiler rejects them; here either rf_glsl.cpp does, or the translation fails to compile"""
    head = CONSTRUCT_HEAD
    for body in ("void main() { imageStore(output_image, ivec2(0), imageLoad(input_image, gl_GlobalInvocationID.xy)); }",      # a uvec2 coordinate
                 "void main() { vec4 v = imageLoad(input_image, ivec2(0)); atomicAdd(v.x, 1.0); imageStore(output_image, ivec2(0), v); }"):      # float atomics
        with pytest.raises(MesaCompileError):
            MesaShader("bad", head + body).run({"input_image": np.zeros((4, 4, 4), np.float32), "output_image": np.zeros((4, 4, 4), np.float32)})
        with pytest.raises(Exception):
            HostShader("bad", head + body)
