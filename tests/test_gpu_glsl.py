"""GPU: filter types in the reference's own file form -- {shader_path}/{type}.comp, GLSL 450 compute (src/config/config.rs:59-75,
src/vulkan/shader.rs:29-160) -- translated by rf_glsl.cpp, compiled by hiprtc at rf_graph_create, run by rfglsl::glsl_node_kernel.

Oracle: shaders/*.comp restate the authored node types in the oracle's tap order, so a graph run THROUGH THE GLSL FILES
must give the bits oracle/rf_oracle.c gives for the same config -- a check of the translator, the GLSL prelude and the kernel against
code that shares nothing with them.  Shaders that have no counterpart in the oracle (workgroup-shared memory, storage blocks updated
in place, matrices, structs, out parameters) are checked against numpy restatements written here."""
import os
import shutil
import sys

import numpy as np
import pytest

import reforge_amd as rf
from oracle import graph as ograph
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import glsl_weights  # noqa: E402


@pytest.fixture
def glsl_dir(tmp_path):
    """an empty shader directory with the FILE-first lookup of the reference; a test copies in the .comp files it wants run as GLSL"""
    old = rf.shader_path()
    rf.set_shader_path(str(tmp_path))
    rf.set_type_lookup(True)
    yield tmp_path
    rf.set_type_lookup(False)
    rf.set_shader_path(old)


def use(glsl_dir, *types):
    for t in types:
        shutil.copy(os.path.join(SHADERS, t + ".comp"), glsl_dir / (t + ".comp"))


def glsl_launches(text):
    return rf.Plan(rf.Config(text)).launches()


G5 = glsl_weights.as_params(1.0, 2)
G9 = glsl_weights.as_params(2.0, 4)
G7 = glsl_weights.as_params(2.5, 7)

# (types run as GLSL, config): every authored node type through its .comp file
TWINS = {
    "gaussian5": (["gaussian5"], "input -> gg -> output\ngg: gaussian5 { sigma: 1.0, %s }" % G5),
    "gaussian9": (["gaussian9"], "input -> gg -> output\ngg: gaussian9 { sigma: 2.0, %s }" % G9),
    "gaussian_r7": (["gaussian"], "input -> gg -> output\ngg: gaussian { sigma: 2.5, radius: 7, %s }" % G7),
    "colour_grade": (["colour_grade"], "input -> cg -> output\ncg: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }"),
    "colour_grade_inplace": (["colour_grade_inplace"], "input -> aa -> cg:image -> bb -> output\naa: passthrough {}\nbb: passthrough {}\ncg: colour_grade_inplace { slope: 0.9, offset: 0.03, saturation: 0.4 }"),
    "sharpen": (["sharpen"], "input -> sh -> output\nsh: sharpen { amount: 0.75 }"),
    "combination": (["combination"], "input -> aa -> mx:input_image0\ninput -> mx:input_image1\nmx -> output\naa: sharpen { amount: 1.0 }\nmx: combination { mix: 0.3 }"),
    "split_luma": (["split_luma"], util.SPLIT2),
    "conv2d_7": (["conv2d"], "input -> kw -> cv -> output\nkw:ConvWeights -> cv:ConvWeights\nkw: conv2d_weights { ksize: 7, sigma: 1.5 }\ncv: conv2d { ksize: 7 }"),
    "chain3": (["gaussian5", "colour_grade", "sharpen"], util.CHAIN3.replace("sigma: 1.0 }", "sigma: 1.0, %s }" % G5)),
}


@pytest.mark.parametrize("fmt", [util.F32, util.U8], ids=["rgba32f", "rgba8"])
@pytest.mark.parametrize("name", sorted(TWINS))
def test_a_graph_run_through_the_glsl_files_matches_the_oracle(ctx, glsl_dir, name, fmt):
    types, text = TWINS[name]
    use(glsl_dir, *types)
    for W, H in ((250, 131), (64, 4), (17, 13), (1, 1)):
        img = util.synthetic(W, H, fmt, seed=0x61 + W)
        want = util.run_oracle(text, img)
        got = util.run_hip(ctx, text, img)
        util.assert_same(got, want, "%s %dx%d" % (name, W, H))
        if W == 250:      # node by node: a point shader alone in its launch (colour_grade_inplace: the stream kernel reading and writing one image)
            util.assert_same(util.run_hip(ctx, text, img, flags=rf.RF_GRAPH_NO_FUSION), want, "%s node by node" % name)


def test_the_files_are_what_ran(ctx, glsl_dir):
    """with the file-first lookup the files are what runs: the 3-stage chain is gaussian5.comp as a node and the other two files as fused row stages"""
    use(glsl_dir, "gaussian5", "colour_grade", "sharpen")
    assert glsl_launches(util.CHAIN3) == ["blur", "grade+sharp"]      # gaussian5.comp: a node (window kernel); colour_grade.comp + sharpen.comp: row stages, fused
    rf.set_type_lookup(False)
    assert glsl_launches(util.CHAIN3) == ["blur+grade+sharp"]
    rf.set_type_lookup(True)
    assert rf.Plan(rf.Config(util.CHAIN3)).needs_jit() == [True, True]


def test_the_reference_passthrough_shader_text(ctx, glsl_dir):
    """the one shader the reference ships (shaders/passthrough.comp there: rgba8 qualifiers, no bounds guard, local_size 16 x 16):
    restated here character for character in its 13 lines' terms -- loads and stores outside the image are dropped"""
    (glsl_dir / "passthrough.comp").write_text(
        "#version 450\n\nlayout (local_size_x = 16, local_size_y = 16) in;\nlayout (binding = 0, rgba8) uniform readonly image2D input_image;\n"
        "layout (binding = 1, rgba8) uniform writeonly image2D output_image;\n\nvoid main()\n{\t\n    vec4 res = imageLoad(input_image, ivec2(gl_GlobalInvocationID.xy));\n\n\n"
        "    imageStore(output_image, ivec2(gl_GlobalInvocationID.xy), res);\n}\n")
    for fmt in (util.F32, util.U8):
        for W, H in ((512, 512), (250, 131), (3, 5)):
            img = util.synthetic(W, H, fmt)
            got = util.run_hip(ctx, "input -> pp -> output\npp: passthrough {}", img)
            util.assert_same(got, img, "passthrough %dx%d" % (W, H))


# ---- the reference's dispatch: ceil(W/16) x ceil(H/16) workgroups of WHATEVER local_size the file declares (command.rs:167-168) ------
def test_a_local_size_below_16_covers_part_of_the_frame_as_in_the_reference(ctx, glsl_dir):
    (glsl_dir / "half.comp").write_text("""#version 450
layout (local_size_x = 8, local_size_y = 4) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); imageStore(output_image, p, imageLoad(input_image, p) + vec4(1.0)); }
""")
    W, H = 100, 70
    img = util.synthetic(W, H, util.F32)
    got = util.run_hip(ctx, "input -> hh -> output\nhh: half {}", img)
    cx, cy = ((W + 15) // 16) * 8, ((H + 15) // 16) * 4        # invocations that exist
    assert cx < W and cy < H
    util.assert_same(got[:cy, :cx], img[:cy, :cx] + np.float32(1.0), "covered")
    want = img + np.float32(1.0)
    assert not (got[cy:] == want[cy:]).any() and not (got[:, cx:] == want[:, cx:]).any()      # never written


# ---- workgroup-shared memory and barrier(): the file's own workgroups -----------------------------------------------------------------
TILE_BLUR = """#version 450
// a 3x3 box mean through a workgroup tile: 16 x 16 invocations stage an 18 x 18 tile (clamp-to-edge) in shared memory
#pragma rf radius 1
#define T 16
layout (local_size_x = T, local_size_y = T) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
const int HALO = 1;
shared vec4 tile[T + 2 * HALO][T + 2 * HALO];

vec4 fetch(ivec2 q, ivec2 size) { return imageLoad(input_image, clamp(q, ivec2(0), size - 1)); }

void main()
{
    ivec2 size = imageSize(input_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    ivec2 l = ivec2(gl_LocalInvocationID.xy);
    ivec2 org = ivec2(gl_WorkGroupID.xy) * T - HALO;
    for (uint i = gl_LocalInvocationIndex; i < uint((T + 2) * (T + 2)); i += gl_WorkGroupSize.x * gl_WorkGroupSize.y) {
        ivec2 t = ivec2(int(i) % (T + 2), int(i) / (T + 2));
        tile[t.y][t.x] = fetch(org + t, size);
    }
    barrier();
    if (p.x >= size.x || p.y >= size.y) return;
    precise vec4 acc = vec4(0.0);
    for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) acc += tile[l.y + dy][l.x + dx];
    imageStore(output_image, p, acc * (1.0 / 9.0));
}
"""


def box3(img):
    H, W, _ = img.shape
    ys = np.clip(np.arange(-1, H + 1), 0, H - 1)
    xs = np.clip(np.arange(-1, W + 1), 0, W - 1)
    pad = img[ys][:, xs]
    acc = np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            acc = acc + pad[dy:dy + H, dx:dx + W]
    return acc * np.float32(1.0 / 9.0)


def test_shared_memory_and_barrier_run_in_the_files_own_workgroups(ctx, glsl_dir):
    (glsl_dir / "tile_blur.comp").write_text(TILE_BLUR)
    r = rf.glsl_reflect("tile_blur", TILE_BLUR)
    assert r["grouped"] and r["radius"] == 1 and r["local_size"] == [16, 16, 1]
    for W, H in ((250, 131), (16, 16), (33, 7)):
        img = util.synthetic(W, H, util.F32)
        got = util.run_hip(ctx, "input -> tb -> output\ntb: tile_blur {}", img)
        util.assert_same(got, box3(img), "tile_blur %dx%d" % (W, H))


# ---- language coverage: structs, out parameters, matrices, swizzles, integer / bool / vector uniforms, several storage blocks ------------
KITCHEN = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 4, rgba32f) uniform readonly image2D other_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (binding = 2) uniform Params {
    float gain;
    int   shift;
    bool  flip;
    uint  mask;
    vec3  tint;          // not a parameter the host sets (render.rs:169-185 knows scalars): stays zero
    float bias;
};
layout (std430, binding = 3) readonly buffer Lut { float lut[8]; vec4 corner; };
layout (std430, binding = 5) buffer Stats { uint hits[4]; } stats;

struct Pair { vec3 a; float k; };
const mat3 TO_YUV = mat3(0.299, -0.147, 0.615, 0.587, -0.289, -0.515, 0.114, 0.436, -0.100);

void split(vec4 t, out vec3 rgb, out float a) { rgb = t.rgb; a = t.a; }
float pick(float v[3], int i) { return v[i]; }
Pair make(vec3 a, float k) { return Pair(a, k); }

void main()
{
    ivec2 size = imageSize(output_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (any(greaterThanEqual(p, size))) return;
    ivec2 q = flip ? ivec2(size.x - 1 - p.x, p.y) : p;
    vec3 rgb; float a;
    split(imageLoad(input_image, q), rgb, a);
    vec3 yuv = TO_YUV * rgb;
    Pair pr = make(yuv.zyx, gain);
    float w[3] = float[](pr.a.x, pr.a.y, pr.a.z);
    vec4 o = vec4(pick(w, 2), pick(w, 1), pick(w, 0), a);
    o.rg = o.gr * pr.k;
    o.b += lut[(p.x + shift) & 7] + bias + tint.x;
    uvec2 u = uvec2(p) & mask;
    o.a = float(u.x + u.y) + corner.w + imageLoad(other_image, p).s;
    if (p.x < 4 && p.y == 0) stats.hits[p.x] = stats.hits[p.x] + uint(p.x) + 1u;
    imageStore(output_image, p, mix(o, o.wzyx, bvec4(false, true, false, true)));
}
"""

KITCHEN_FILL = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (std430, binding = 3) writeonly buffer Lut { float lut[8]; vec4 corner; };
void main()
{
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (p.y == 0 && p.x < 8) lut[p.x] = 0.125 * float(p.x);
    if (p.x == 0 && p.y == 0) corner = vec4(1.0, 2.0, 3.0, 0.5);
    imageStore(output_image, p, imageLoad(input_image, p));
}
"""


def kitchen(img, other, gain, shift, flip, mask, bias):
    H, W, _ = img.shape
    f = np.float32
    src = img[:, ::-1] if flip else img
    m = np.array([[0.299, -0.147, 0.615], [0.587, -0.289, -0.515], [0.114, 0.436, -0.100]], f)      # columns of TO_YUV
    rgb = src[..., :3]
    yuv = (m[0] * rgb[..., 0:1] + m[1] * rgb[..., 1:2]) + m[2] * rgb[..., 2:3]      # columns scaled, summed left to right
    zyx = yuv[..., ::-1]
    o = np.empty_like(img)
    o[..., 0], o[..., 1], o[..., 2], o[..., 3] = zyx[..., 2], zyx[..., 1], zyx[..., 0], src[..., 3]
    r, g = o[..., 1] * f(gain), o[..., 0] * f(gain)
    o[..., 0], o[..., 1] = r, g
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    lut = (f(0.125) * np.arange(8, dtype=f))
    o[..., 2] = (o[..., 2] + ((lut[(xs + shift) & 7] + f(bias)) + f(0.0))).astype(f)
    o[..., 3] = ((xs & mask) + (ys & mask)).astype(f) + f(0.5) + other[..., 0]
    out = o.copy()
    out[..., 1], out[..., 3] = o[..., 2], o[..., 0]      # mix(o, o.wzyx, (F, T, F, T)): y <- z, w <- x
    return out


@pytest.mark.parametrize("flip", [False, True])
def test_structs_matrices_out_parameters_and_several_storage_blocks(ctx, glsl_dir, flip):
    (glsl_dir / "kitchen.comp").write_text(KITCHEN)
    (glsl_dir / "kitchen_fill.comp").write_text(KITCHEN_FILL)
    text = """
input -> kf -> kk:input_image
input -> gg -> kk:other_image
kf:Lut -> kk:Lut
kk -> output
gg: colour_grade { slope: 0.5, offset: 0.25, saturation: 1.0 }
kf: kitchen_fill {}
kk: kitchen { gain: 1.5, shift: 3, flip: %s, mask: 5, bias: 0.25 }
""" % ("true" if flip else "false")
    r = rf.glsl_reflect("kitchen", KITCHEN)
    off = {m["name"]: m["offset"] for m in r["uniform_blocks"][0]["members"]}
    assert off == {"gain": 0, "shift": 4, "flip": 8, "mask": 12, "tint": 16, "bias": 28} and r["uniform_bytes"] == 32      # std140: the float packs behind the vec3
    lut = r["storage_blocks"][0]
    assert [m["offset"] for m in lut["members"]] == [0, 32] and lut["bytes"] == 48
    W, H = 70, 37
    img = util.synthetic(W, H, util.F32)
    other = util.run_oracle("input -> gg -> output\ngg: colour_grade { slope: 0.5, offset: 0.25, saturation: 1.0 }", img)
    got = util.run_hip(ctx, text, img)
    util.assert_same(got, kitchen(img, other, 1.5, 3, flip, 5, 0.25), "kitchen flip=%s" % flip)


# ---- atomic memory functions: invocations meeting in a storage block (a histogram node and a node that reads it) -----------------------------
HIST_GLOBAL = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (std430, binding = 2) buffer Hist { uint bins[64]; uint brightest; uint texels; };
void main()
{
    ivec2 size = imageSize(input_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (p.x >= size.x || p.y >= size.y) return;
    vec4 t = imageLoad(input_image, p);
    uint code = uint(clamp(dot(t.rgb, vec3(0.25, 0.5, 0.25)), 0.0, 1.0) * 255.0);
    atomicAdd(bins[code >> 2], 1u);
    atomicMax(brightest, code);
    atomicAdd(texels, 1u);
    imageStore(output_image, p, t);
}
"""

# the same block filled the way compute filters usually do it: a histogram per workgroup in shared memory, flushed by 64 invocations
HIST_SHARED = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (std430, binding = 2) buffer Hist { uint bins[64]; uint brightest; uint texels; };
shared uint local_bins[64];
shared uint local_max;
void main()
{
    ivec2 size = imageSize(input_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    uint l = gl_LocalInvocationIndex;
    if (l < 64u) local_bins[l] = 0u;
    if (l == 64u) local_max = 0u;
    barrier();
    bool inside = p.x < size.x && p.y < size.y;
    if (inside) {
        vec4 t = imageLoad(input_image, p);
        uint code = uint(clamp(dot(t.rgb, vec3(0.25, 0.5, 0.25)), 0.0, 1.0) * 255.0);
        atomicAdd(local_bins[code >> 2], 1u);
        atomicMax(local_max, code);
        imageStore(output_image, p, t);
    }
    barrier();
    if (l < 64u && local_bins[l] != 0u) { atomicAdd(bins[l], local_bins[l]); atomicAdd(texels, local_bins[l]); }
    if (l == 64u) atomicMax(brightest, local_max);
}
"""

HIST_APPLY = """#version 450
// out = in scaled so the brightest luma code of the frame becomes 255, rgb further scaled by the share of texels at or below the texel's bin
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (std430, binding = 2) readonly buffer Hist { uint bins[64]; uint brightest; uint texels; };
void main()
{
    ivec2 size = imageSize(output_image);
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (p.x >= size.x || p.y >= size.y) return;
    vec4 t = imageLoad(input_image, p);
    uint code = uint(clamp(dot(t.rgb, vec3(0.25, 0.5, 0.25)), 0.0, 1.0) * 255.0);
    uint below = 0u;
    for (uint b = 0u; b <= (code >> 2); ++b) below += bins[b];
    float share = float(below) / float(texels);
    float gain = 255.0 / float(max(brightest, 1u));
    imageStore(output_image, p, vec4(t.rgb * gain * share, t.a));
}
"""


def hist_apply(img, frames=1):
    f = np.float32
    y = np.clip((img[..., 0] * f(0.25) + img[..., 1] * f(0.5)) + img[..., 2] * f(0.25), f(0), f(1))
    code = (y * f(255.0)).astype(np.uint32)
    bins = np.bincount((code >> 2).ravel(), minlength=64).astype(np.uint64) * frames      # the block is never cleared (nor is the reference's): frames add up
    below = np.cumsum(bins)[code >> 2]
    share = (below.astype(f) / f(code.size * frames)).astype(f)
    gain = f(255.0) / f(max(int(code.max()), 1))
    out = img.copy()
    out[..., :3] = (img[..., :3] * gain) * share[..., None]
    return out


@pytest.mark.parametrize("source", ["global", "shared"])
def test_a_histogram_node_fills_a_block_with_atomics_and_the_next_node_reads_it(ctx, glsl_dir, source):
    text = HIST_GLOBAL if source == "global" else HIST_SHARED
    (glsl_dir / "histogram.comp").write_text(text)
    (glsl_dir / "hist_apply.comp").write_text(HIST_APPLY)
    r = rf.glsl_reflect("histogram", text)
    assert r["grouped"] == (source == "shared") and not r["point"] and not r["stencil"]
    cfg = "input -> hh -> ha -> output\nhh:Hist -> ha:Hist\nhh: histogram {}\nha: hist_apply {}"
    assert glsl_launches(cfg) == ["hh", "ha"]
    for W, H in ((250, 131), (64, 64), (7, 3)):
        img = util.synthetic(W, H, util.F32, seed=W)
        util.assert_same(util.run_hip(ctx, cfg, img), hist_apply(img), "%s histogram %dx%d" % (source, W, H))
    # a second frame through the same graph: the block keeps what the first frame added (every invocation counted exactly once per frame)
    img = util.synthetic(120, 50, util.F32, seed=5)
    g = rf.Graph(ctx, rf.Config(cfg), 120, 50, util.F32)
    g.upload_raw(img)
    g.execute()
    g.execute()
    g.wait()
    util.assert_same(g.download_raw(), hist_apply(img, frames=2), "two frames")
    g.close()


# ---- generated programs: the product on gfx950 against the same translation compiled for the host ----------------------------------------------
def test_generated_shaders_give_the_host_translations_bits(ctx, glsl_dir):
    """tests/glsl_gen.py's typed random programs (the ones tests/test_glsl_mesa.py runs on Mesa's GLSL compiler): translated, compiled by hiprtc,
    dispatched by rf_graph_execute -- against the translation compiled with clang++ for x86.  Correctly rounded division and square root,
    wrapping int arithmetic, conversions, the integer built-ins on the GPU.  (scripts/fuzz_glsl_gpu.py: the long campaign)"""
    from tests.glsl_gen import generate
    from tests.glsl_host import HostShader
    for seed in range(900, 906):
        text = generate(seed, 18)
        name = "gen%d" % seed
        (glsl_dir / (name + ".comp")).write_text(text)
        for fmt in (util.F32, util.U8):
            img = util.synthetic(150, 67, fmt, seed=seed)
            want = np.zeros_like(img)
            HostShader(name, text).run({"input_image": img, "output_image": want}, {"gain": 1.5, "shift": 3})
            got = util.run_hip(ctx, "input -> gn -> output\ngn: %s { gain: 1.5, shift: 3 }" % name, img)
            util.assert_same(got, want, "generated shader %d, format %d" % (seed, fmt))


def test_generated_shaders_through_shared_memory_give_mesas_bits(ctx, glsl_dir):
    """the product against an INDEPENDENT GLSL implementation, directly: generated programs (no fma(), every float precise: neither side
    contracts) that hand their result to another invocation through a `shared` tile and barrier() -- dispatched in the file's own
    workgroups on the GPU, and compiled and run by Mesa's GLSL compiler + llvmpipe on the host's cores (tests/mesa_glsl.py).
    (scripts/fuzz_glsl_gpu_mesa.py: the campaign)"""
    from tests.glsl_gen import generate
    from tests.mesa_glsl import MesaShader, runner, why_not
    if runner() is None:
        pytest.skip("Mesa's software rasteriser is not usable here: " + why_not())
    for seed in range(700, 704):
        text = generate(seed, 16).replace("void main()\n{", "shared vec4 tile[16][16];\nvoid main()\n{")
        text = text.replace("    imageStore(output_image, p, o);", "    uvec2 l = gl_LocalInvocationID.xy;\n    tile[l.y][l.x] = o;\n    barrier();\n"
                            "    precise vec4 both = o + tile[(l.y + 1u) & 15u][(l.x + 3u) & 15u];\n    imageStore(output_image, p, both);")
        name = "grp%d" % seed
        (glsl_dir / (name + ".comp")).write_text(text)
        assert rf.glsl_reflect(name, text)["grouped"]
        img = util.synthetic(160, 64, util.F32, seed=seed)      # whole workgroups: no invocation leaves before the barrier
        want = MesaShader(name, text).run({"input_image": img, "output_image": np.zeros_like(img)}, {"gain": 1.5, "shift": 3})["output_image"]
        got = util.run_hip(ctx, "input -> gn -> output\ngn: %s { gain: 1.5, shift: 3 }" % name, img)
        util.assert_same(got, want, "generated shader %d through shared memory: the product vs Mesa" % seed)


def test_the_fast_forms_of_a_stencil_file_give_mesas_bits(ctx, glsl_dir):
    """a 5 x 5 box (window kernel on LDS tiles + border ring) and a 3 x 3 box between two point shaders (a fused 3 x 3 row stage), all from GLSL
    files whose sums are `precise` (no contraction, no re-association on either side): the product's fast forms against Mesa's GLSL compiler +
    llvmpipe running the same files node by node"""
    from tests.mesa_glsl import MesaShader, runner, why_not
    from tests.test_glsl import BOX5, GAIN
    if runner() is None:
        pytest.skip("Mesa's software rasteriser is not usable here: " + why_not())
    box5 = BOX5.replace("    vec4 acc = vec4(0.0);", "    precise vec4 acc = vec4(0.0);").replace("imageStore(output_image, p, acc * gain);", "acc = acc * gain; imageStore(output_image, p, acc);")
    box3 = box5.replace("#pragma rf radius 2", "#pragma rf radius 1").replace("dy = -2; dy <= 2", "dy = -1; dy <= 1").replace("dx = -2; dx <= 2", "dx = -1; dx <= 1")
    gain = GAIN.replace("vec4(c.rgb * gain + bias, c.a)", "vec4(c.rgb * gain, c.a - bias)")      # (one operation per channel: nothing to contract)
    for name, text in (("box5", box5), ("box3", box3), ("gainp", gain)):
        (glsl_dir / (name + ".comp")).write_text(text)
    r5, r3 = rf.glsl_reflect("box5", box5), rf.glsl_reflect("box3", box3)
    assert r5["stencil"] and r5["radius"] == 2 and r3["stencil"] and r3["box"] and rf.glsl_reflect("gainp", gain)["point"]
    cfg = "input -> aa -> bb -> cc -> dd -> output\naa: gainp { gain: 0.5, bias: 0.125 }\nbb: box3 { gain: 0.125 }\ncc: gainp { gain: 1.5, bias: -0.25 }\ndd: box5 { gain: 0.0625 }"
    assert glsl_launches(cfg) == ["aa+bb+cc", "dd"]      # a fused chain with the 3 x 3 row stage inside; the 5 x 5 box on the window kernel
    for W, H in ((250, 131), (64, 48)):
        img = util.synthetic(W, H, util.F32, seed=W)
        want = img
        for t, text, params in (("gainp", gain, {"gain": 0.5, "bias": 0.125}), ("box3", box3, {"gain": 0.125}), ("gainp", gain, {"gain": 1.5, "bias": -0.25}), ("box5", box5, {"gain": 0.0625})):
            want = MesaShader(t, text).run({"input_image": want, "output_image": np.zeros_like(want)}, params)["output_image"]
        util.assert_same(util.run_hip(ctx, cfg, img), want, "fused row stages + window kernel vs Mesa node by node, %dx%d" % (W, H))


# ---- row strips: the launch split into interior and boundary rows (the geometry of the halo exchange, one GPU) ----------------------------
def test_a_stencil_shader_split_into_row_ranges_gives_the_same_frame(ctx, glsl_dir, monkeypatch):
    """both files state `#pragma rf radius 2`: the launch radius of a row-strip partition"""
    use(glsl_dir, "local_contrast", "gaussian5")
    text = "input -> gg -> lc -> output\ngg: gaussian5 { sigma: 1.0, %s }\nlc: local_contrast { amount: 0.8 }" % G5
    assert [l["radius"] for l in rf.Plan(rf.Config(text)).launch_info()] == [2, 2]
    img = util.synthetic(200, 90, util.F32)
    whole = util.run_hip(ctx, text, img)
    monkeypatch.setenv("RF_FORCE_SPLIT", "1")
    split = util.run_hip(ctx, text, img)
    util.assert_same(split, whole, "split launches")
    rf.set_type_lookup(False)
    from tests.util import register_user_types
    old = register_user_types()
    try:
        want = util.run_oracle(text, img)
    finally:
        rf.set_shader_path(old)
    util.assert_same(whole, want, "against the oracle (stage-file twin of local_contrast)")


@pytest.mark.parametrize("fmt", [util.F32, util.U8], ids=["rgba32f", "rgba8"])
def test_point_shaders_fused_between_hand_written_stages(ctx, glsl_dir, fmt):
    """gaussian5 -> gain.comp -> invert.comp -> sharpen as ONE launch (the two files are row stages of the stream kernel), against the
    same graph node by node and against the oracle's restatement of each node (invert: its stage-file twin compiled for the host)"""
    from oracle import pixel
    from tests.test_glsl import GAIN
    rf.set_type_lookup(False)
    (glsl_dir / "gain.comp").write_text(GAIN)
    use(glsl_dir, "invert")
    text = "input -> gg -> gn -> iv -> sh -> output\ngg: gaussian5 { sigma: 1.0 }\ngn: gain { gain: 1.25, bias: -0.125 }\niv: invert { enabled: true, strength: 0.5 }\nsh: sharpen { amount: 0.5 }"
    assert rf.Plan(rf.Config(text)).launches() == ["gg+gn+iv+sh"]
    if "user" not in ograph.NODE_TYPES.get("invert", {}):
        ograph.register_user_type("invert", os.path.join(SHADERS, "invert.stage.hip"))
    f = np.float32
    for W, H in ((250, 131), (64, 4), (17, 13), (1, 1)):
        img = util.synthetic(W, H, fmt, seed=0x71 + W)
        a = pixel.gaussian(img, 2, sigma=1.0)
        af = a.astype(f) / f(255.0) if fmt == util.U8 else a
        b = af.copy()
        b[..., :3] = af[..., :3] * f(1.25) + f(-0.125)
        if fmt == util.U8:      # the node boundary of an rgba8 graph: store as UNORM8 (clamp, x 255, round to nearest even)
            b = np.rint(np.clip(b, 0, 1) * f(255.0)).astype(np.uint8)
        want = util.run_oracle("input -> iv -> sh -> output\niv: invert { enabled: true, strength: 0.5 }\nsh: sharpen { amount: 0.5 }", b)
        fused = util.run_hip(ctx, text, img)
        util.assert_same(fused, want, "fused %dx%d" % (W, H))
        util.assert_same(util.run_hip(ctx, text, img, flags=rf.RF_GRAPH_NO_FUSION), want, "node by node %dx%d" % (W, H))


def box5(img, gain):
    H, W, _ = img.shape
    ys = np.clip(np.arange(-2, H + 2), 0, H - 1)
    xs = np.clip(np.arange(-2, W + 2), 0, W - 1)
    pad = img[ys][:, xs]
    acc = np.zeros_like(img)
    for dy in range(5):
        for dx in range(5):
            acc = acc + pad[dy:dy + H, dx:dx + W]
    return acc * np.float32(gain)


def test_a_stencil_shader_on_the_window_kernel_and_on_its_generic_kernel(ctx, glsl_dir):
    """BOX5 (tests/test_glsl.py): a 5 x 5 box through a helper function, recognised as a translation-invariant stencil -- the LDS-tiled window
    kernel computes the frame, the generic kernel its border ring; both ways (RF_EXEC_GLSL_NO_WINDOW) against numpy, bit for bit"""
    from tests.test_glsl import BOX5
    (glsl_dir / "box5.comp").write_text(BOX5)
    text = "input -> bb -> output\nbb: box5 { gain: 0.04 }"
    for W, H in ((250, 131), (64, 5), (5, 64), (4, 4), (96, 64)):
        img = util.synthetic(W, H, util.F32, seed=W)
        want = box5(img, 0.04)
        g = rf.Graph(ctx, rf.Config(text), W, H, util.F32)
        try:
            assert g.note == "", g.note      # the self-test of rf_graph_create accepted the window kernel
            g.upload_raw(img)
            g.execute()
            g.wait()
            util.assert_same(g.download_raw(), want, "window kernel %dx%d" % (W, H))
        finally:
            g.close()
        util.assert_same(util.run_hip(ctx, text, img, exec_flags=rf.RF_EXEC_GLSL_NO_WINDOW), want, "generic kernel %dx%d" % (W, H))


@pytest.mark.parametrize("fmt", [util.F32, util.U8], ids=["rgba32f", "rgba8"])
def test_three_by_three_stencil_shaders_fused_as_row_stages(ctx, glsl_dir, fmt):
    """gaussian5 -> sharpen.comp -> invert.comp -> edge_detect.comp -> colour_grade: ONE launch (the 3 x 3 files are StUser row stages of radius 1 in
    a virtual 3 x 3 frame, checked against their generic kernels when the graph is created), against the oracle's restatement of every node"""
    rf.set_type_lookup(False)
    for src, dst in (("sharpen", "sharp3"), ("edge_detect", "edges3"), ("invert", "invert")):
        shutil.copy(os.path.join(SHADERS, src + ".comp"), glsl_dir / (dst + ".comp"))
    for t in ("invert", "edge_detect"):
        if "user" not in ograph.NODE_TYPES.get(t, {}):
            ograph.register_user_type(t, os.path.join(SHADERS, t + ".stage.hip"))
    text = ("input -> gg -> s3 -> iv -> e3 -> cg -> output\ngg: gaussian5 { sigma: 1.0 }\ns3: sharp3 { amount: 0.5 }\niv: invert { enabled: true, strength: 0.5 }\n"
            "e3: edges3 { scale: 1.5 }\ncg: colour_grade { slope: 1.1, offset: 0.0, saturation: 1.0 }")
    oracle_text = text.replace("sharp3", "sharpen").replace("edges3", "edge_detect")
    for W, H in ((250, 131), (64, 4), (17, 13), (1, 1)):
        img = util.synthetic(W, H, fmt, seed=0x81 + W)
        want = util.run_oracle(oracle_text, img)
        g = rf.Graph(ctx, rf.Config(text), W, H, fmt)
        try:
            assert g.note == "" and g.plan.launches() == ["gg+s3+iv+e3+cg"], (g.note, g.plan.launches())
            g.upload_raw(img)
            g.execute()
            g.wait()
            util.assert_same(g.download_raw(), want, "fused %dx%d" % (W, H))
        finally:
            g.close()
        util.assert_same(util.run_hip(ctx, text, img, flags=rf.RF_GRAPH_GLSL_NODES), want, "files as nodes %dx%d" % (W, H))


def test_a_three_by_three_shader_with_its_own_idea_of_the_edges_is_not_fused(ctx, glsl_dir):
    """a 3 x 3 box WITHOUT clamps darkens the frame's edges (zeros outside the image); a row stage of the stream kernel sees clamp-to-edge
    neighbours there.  rf_graph_create finds the two to differ, says so, and the type is a node with a kernel of its own from then on"""
    rf.set_type_lookup(False)
    src = """#version 450
#pragma rf radius 1
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
void main()
{
    ivec2 p = ivec2(gl_GlobalInvocationID.xy);
    if (any(greaterThanEqual(p, imageSize(output_image)))) return;
    vec4 acc = vec4(0.0);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) acc += imageLoad(input_image, p + ivec2(dx, dy));
    imageStore(output_image, p, acc * 0.125);
}
"""
    assert rf.glsl_reflect("box3z", src)["box"]
    (glsl_dir / "box3z.comp").write_text(src)
    text = "input -> gg -> bz -> output\ngg: gaussian5 { sigma: 1.0 }\nbz: box3z {}"
    assert rf.Plan(rf.Config(text)).launches() == ["gg+bz"]      # before any graph has looked: a row stage
    img = util.synthetic(120, 70, util.F32)
    g = rf.Graph(ctx, rf.Config(text), 120, 70, util.F32)
    try:
        assert "box3z.comp) is not fused" in g.note and g.plan.launches() == ["gg", "bz"], (g.note, g.plan.launches())
        g.upload_raw(img)
        g.execute()
        g.wait()
        from oracle import pixel
        a = pixel.gaussian(img, 2, sigma=1.0)
        pad = np.zeros((72, 122, 4), np.float32)
        pad[1:-1, 1:-1] = a
        acc = np.zeros_like(a)
        for dy in range(3):
            for dx in range(3):
                acc = acc + pad[dy:dy + 70, dx:dx + 120]
        util.assert_same(g.download_raw(), acc * np.float32(0.125), "zero-padded 3x3 box")
    finally:
        g.close()
    assert rf.Plan(rf.Config(text)).launches() == ["gg", "bz"]      # the process remembers


def test_a_stencil_that_relies_on_zeros_outside_the_image(ctx, glsl_dir):
    """the same box WITHOUT clamps: imageLoad outside the image returns zero, so the frame's edges darken.  The window kernel's tiles hold
    clamp-to-edge copies there -- which is why the border ring is the generic kernel's: against numpy (zero padding), both ways"""
    from tests.test_glsl import BOX5
    src = BOX5.replace("clamp(q, ivec2(0), size - 1)", "q")
    assert src != BOX5 and rf.glsl_reflect("box5z", src)["stencil"]
    (glsl_dir / "box5z.comp").write_text(src)
    text = "input -> bb -> output\nbb: box5z { gain: 0.04 }"
    for W, H in ((250, 131), (3, 70), (70, 3), (1, 1)):
        img = util.synthetic(W, H, util.F32, seed=W + 5)
        pad = np.zeros((H + 4, W + 4, 4), np.float32)
        pad[2:-2, 2:-2] = img
        acc = np.zeros_like(img)
        for dy in range(5):
            for dx in range(5):
                acc = acc + pad[dy:dy + H, dx:dx + W]
        want = acc * np.float32(0.04)
        g = rf.Graph(ctx, rf.Config(text), W, H, util.F32)
        try:
            assert g.note == "", g.note
            g.upload_raw(img)
            g.execute()
            g.wait()
            util.assert_same(g.download_raw(), want, "window kernel %dx%d" % (W, H))
        finally:
            g.close()
        util.assert_same(util.run_hip(ctx, text, img, exec_flags=rf.RF_EXEC_GLSL_NO_WINDOW), want, "generic kernel %dx%d" % (W, H))


def test_a_shader_that_reads_further_than_it_states_keeps_its_generic_kernel(ctx, glsl_dir):
    """gaussian9.comp reads 4 texels out; with `#pragma rf radius 2` its window kernel cannot agree with the generic one: rf_graph_create
    finds out on a random frame, says so, and the graph gives the right frame on the generic kernel"""
    src = open(os.path.join(SHADERS, "gaussian9.comp")).read()
    assert "#pragma rf radius 4" in src
    (glsl_dir / "gaussian9.comp").write_text(src.replace("#pragma rf radius 4", "#pragma rf radius 2"))
    text = TWINS["gaussian9"][1]
    img = util.synthetic(200, 90, util.F32)
    g = rf.Graph(ctx, rf.Config(text), 200, 90, util.F32)
    try:
        assert "keeps its generic kernel" in g.note and "gaussian9.comp" in g.note, g.note
        g.upload_raw(img)
        g.execute()
        g.wait()
        util.assert_same(g.download_raw(), util.run_oracle(text, img), "understated radius")
    finally:
        g.close()


def test_equality_of_vectors_and_the_integer_built_ins_on_the_gpu(ctx, glsl_dir):
    from tests.test_glsl import EQUALITY, equality
    (glsl_dir / "equality.comp").write_text(EQUALITY)
    img = util.synthetic(250, 131, util.F32)
    img[3, 5, :3] = 0.0
    img[4, 6, 0] = img[4, 6, 1]
    got = util.run_hip(ctx, "input -> pp -> eq:image -> output\npp: passthrough {}\neq: equality {}", img)
    util.assert_same(got, equality(img), "equality.comp")


def test_a_shader_on_a_frame_of_four_gibibytes(ctx, glsl_dir):
    """16384 x 16384 rgba32f: an image is exactly 4 GiB, beyond a 32-bit byte offset -- the kernel variant with 64-bit addresses; rows
    at both ends of the frame against the input generator"""
    from oracle import pixel
    (glsl_dir / "shift.comp").write_text("""#version 450
#pragma rf radius 0
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
void main() { ivec2 p = ivec2(gl_GlobalInvocationID.xy); imageStore(output_image, p, imageLoad(input_image, p + ivec2(1, 0)) + vec4(0.5)); }
""")
    assert not rf.glsl_reflect("shift", (glsl_dir / "shift.comp").read_text())["point"]
    W = H = 16384
    g = rf.Graph(ctx, rf.Config("input -> sf -> output\nsf: shift {}"), W, H, util.F32)
    try:
        g.fill_synthetic(0x5EED0042)
        g.execute()
        g.wait()
        for y0, y1 in ((0, 2), (8191, 8193), (H - 2, H)):
            src = pixel.fill_synthetic(W, y1 - y0, util.F32, 0x5EED0042, y0=y0)
            want = np.zeros_like(src)
            want[:, :-1] = src[:, 1:]      # the texel to the right; beyond the frame's last column: zero
            want += np.float32(0.5)
            util.assert_same(g.download_rows(y0, y1), want, "rows %d..%d" % (y0, y1))
    finally:
        g.close()


# ---- combined image samplers ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", [util.F32, util.U8], ids=["rgba32f", "rgba8"])
def test_a_sampler2D_is_filtered_by_the_graphs_sampler(ctx, glsl_dir, fmt):
    """`uniform sampler2D` (a combined image sampler, shader.rs:98) read with texture() / texelFetch(): LINEAR, U clamp-to-edge, V repeat
    (vkutils.rs:358-365 as written), against the numpy restatement of tests/test_glsl.py"""
    from tests.test_glsl import RESAMPLE, resample
    (glsl_dir / "resample.comp").write_text(RESAMPLE)
    img = util.synthetic(250, 131, fmt)
    for sx, sy, zoom in ((0.0, 0.0, 1.0), (0.37, -1.25, 1.0), (3.5, 40.0, 0.75)):
        got = util.run_hip(ctx, "input -> gg -> rs:source\nrs -> output\ngg: passthrough {}\nrs: resample { shift_x: %s, shift_y: %s, zoom: %s }" % (sx, sy, zoom), img)
        want = resample(img, sx, sy, zoom)
        if fmt == util.U8:      # imageStore's conversion of the restatement's floats: clamp, x 255, round to nearest even
            want = np.rint(np.clip(want, 0, 1) * np.float32(255.0)).astype(np.uint8)
        util.assert_same(got, want, "texture() shift %s %s zoom %s" % (sx, sy, zoom))


# ---- row strips over several ranks (processes sharing GPU 0 over the RCCL test double of tests/test_gpu_exchange.py) ------------------------
from tests.test_gpu_exchange import fake_rccl_dir, run_ranks  # noqa: E402,F401


@pytest.mark.parametrize("flags", [0, rf.RF_GRAPH_NO_HALO_XCHG], ids=["exchange", "overfetch"])
def test_glsl_nodes_split_into_row_strips(fake_rccl_dir, glsl_dir, tmp_path, monkeypatch, flags):
    """the headline graph through its three .comp files on three ranks: each file states `#pragma rf radius` (2, 0, 1), which is
    the halo a rank exchanges before -- or over-fetches for -- the node; shaders address the FRAME (imageSize, gl_GlobalInvocationID)"""
    use(glsl_dir, "gaussian5", "colour_grade", "sharpen")
    text = TWINS["chain3"][1]
    monkeypatch.setenv("RF_TEST_SHADER_PATH", str(glsl_dir))
    monkeypatch.setenv("RF_TEST_FILES_FIRST", "1")
    W, H, seed = 333, 257, 0x5EED0011
    for fmt in (util.F32, util.U8):
        sub = tmp_path / ("fmt%d" % fmt)
        sub.mkdir()
        got = run_ranks(fake_rccl_dir, sub, text, 3, W, H, fmt, flags, seed)
        from oracle import pixel
        util.assert_same(got, util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, seed)), "GLSL chain on 3 ranks, flags=%d fmt=%d" % (flags, fmt))


def test_a_shader_with_shared_memory_split_into_row_strips(fake_rccl_dir, glsl_dir, tmp_path, monkeypatch):
    """TILE_BLUR (16 x 16 workgroups, an 18 x 18 tile in shared memory, barrier()) on three ranks: a rank runs the workgroups that hold rows
    of its strip, loads reach one row into the neighbours' (#pragma rf radius 1), stores outside the strip are dropped"""
    (glsl_dir / "tile_blur.comp").write_text(TILE_BLUR)
    monkeypatch.setenv("RF_TEST_SHADER_PATH", str(glsl_dir))
    monkeypatch.setenv("RF_TEST_FILES_FIRST", "1")
    from oracle import pixel
    W, H, seed = 333, 257, 0x5EED0012
    for flags in (0, rf.RF_GRAPH_NO_HALO_XCHG):
        sub = tmp_path / ("flags%d" % flags)
        sub.mkdir()
        got = run_ranks(fake_rccl_dir, sub, "input -> tb -> output\ntb: tile_blur {}", 3, W, H, util.F32, flags, seed)
        util.assert_same(got, box3(pixel.fill_synthetic(W, H, util.F32, seed)), "tile_blur on 3 ranks, flags=%d" % flags)


def test_a_shader_that_does_not_state_its_radius_is_not_split(glsl_dir):
    """one GPU: any shader runs; more than one rank: the node must say how far it reads -- and may not fill a storage block"""
    (glsl_dir / "mystery.comp").write_text(open(os.path.join(SHADERS, "sharpen.comp")).read().replace("#pragma rf radius 1", ""))
    use(glsl_dir, "conv2d_weights", "conv2d")
    c2 = rf.Context(0, 0, 2, None)      # a rank of two without a communicator: over-fetch graphs only
    try:
        with pytest.raises(rf.RfError) as e:
            rf.Graph(c2, rf.Config("input -> my -> output\nmy: mystery { amount: 0.5 }"), 64, 64, util.F32, flags=rf.RF_GRAPH_NO_HALO_XCHG)
        assert "mystery.comp does not say `#pragma rf radius N`" in str(e.value)
        with pytest.raises(rf.RfError) as e:
            rf.Graph(c2, rf.Config("input -> kw -> cv -> output\nkw:ConvWeights -> cv:ConvWeights\nkw: conv2d_weights { ksize: 3 }\ncv: conv2d { ksize: 3 }"), 64, 64, util.F32,
                     flags=rf.RF_GRAPH_NO_HALO_XCHG)
        assert "conv2d_weights.comp writes a storage block (ConvWeights)" in str(e.value)
    finally:
        c2.close()


def test_an_edited_shader_is_translated_again(ctx, glsl_dir):
    src = open(os.path.join(SHADERS, "invert.comp")).read()
    (glsl_dir / "invert.comp").write_text(src)
    text = "input -> iv -> output\niv: invert { enabled: true, strength: 1.0 }"
    img = util.synthetic(64, 20, util.F32)
    a = util.run_hip(ctx, text, img)
    t0 = rf.lib().rf_user_stage_mtime(b"invert")
    assert t0 > 0
    (glsl_dir / "invert.comp").write_text(src.replace("void main()", "void main_body()").replace("#version 450", "#version 450\nvoid main_body();", 1) +
                                         "\nvoid main() { main_body(); ivec2 p = ivec2(gl_GlobalInvocationID.xy); imageStore(output_image, p, imageLoad(input_image, p) * 0.5); }\n")
    os.utime(glsl_dir / "invert.comp", ns=(10 ** 18, 10 ** 18))
    b = util.run_hip(ctx, text, img)
    util.assert_same(b, img * np.float32(0.5), "edited")
    assert not np.array_equal(a, b) and rf.lib().rf_user_stage_mtime(b"invert") == 10 ** 18


def test_a_file_outside_the_subset_is_refused_with_its_line(ctx, glsl_dir):
    (glsl_dir / "bad.comp").write_text("#version 450\nlayout (local_size_x = 16, local_size_y = 16) in;\nlayout (binding = 0) uniform samplerCube tex;\nvoid main() {}\n")
    with pytest.raises(rf.RfError) as e:
        rf.Plan(rf.Config("input -> bb -> output\nbb: bad {}"))
    assert "bad.comp:3" in str(e.value)
    (glsl_dir / "worse.comp").write_text("#version 450\nlayout (local_size_x = 16) in;\nlayout (binding = 0, rgba32f) uniform readonly image2D input_image;\n"
                                        "layout (binding = 1, rgba32f) uniform writeonly image2D output_image;\nvoid main() { imageStore(output_image, ivec2(0), undefined_thing); }\n")
    img = util.synthetic(16, 16, util.F32)
    with pytest.raises(rf.RfError) as e:
        util.run_hip(ctx, "input -> ww -> output\nww: worse {}", img)
    assert "worse.comp" in str(e.value) and "undefined_thing" in str(e.value)
