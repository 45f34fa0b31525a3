"""GPU: filter types that are FILES ({shader_path}/{type}.stage.hip) -- dropped into a directory, fused with built-in nodes,
checked against an independent restatement (exact rationals, one rounding per operation: tests/golden/exact_eval.py),
edited and reloaded.  Ref: src/config/config.rs:59-75, src/vulkan/shader.rs:29-59, src/render.rs:225-249."""
import os
import shutil
import struct
from fractions import Fraction

import numpy as np
import pytest

import reforge_amd as rf
from tests import util
from tests.golden import exact_eval as ex

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")


@pytest.fixture(scope="module")
def ctx():
    c = rf.Context(0)
    yield c
    c.close()


@pytest.fixture
def stage_dir(tmp_path):
    for f in ("edge_detect.stage.hip", "invert.stage.hip"):
        shutil.copy(os.path.join(SHADERS, f), tmp_path / f)
    old = rf.shader_path()
    rf.set_shader_path(str(tmp_path))
    yield tmp_path
    rf.set_shader_path(old)


# ---- the restatement: shaders/edge_detect.stage.hip and invert.stage.hip in exact arithmetic --------------------------
def to_img(a):
    """numpy frame (f32 values or u8 codes) -> exact_eval image"""
    if a.dtype == np.uint8:
        return [[[Fraction(int(c)) for c in t] for t in row] for row in a]
    return [[[Fraction(float(c)) for c in t] for t in row] for row in a]


def from_img(img, dtype):
    fmt = "u8" if dtype == np.uint8 else "f32"
    H, W = len(img), len(img[0])
    return np.frombuffer(ex.to_bytes(img, fmt), dtype=dtype).reshape(H, W, 4).copy()


def sobel(img, scale):
    scale = ex.f32(scale)
    two = Fraction(2)
    out = []
    for y in range(len(img)):
        row = []
        for x in range(len(img[0])):
            n = [[ex.at(img, x + dx, y + dy) for dx in (-1, 0, 1)] for dy in (-1, 0, 1)]
            t = []
            for c in range(3):
                right = ex.rn(ex.fma(two, n[1][2][c], n[0][2][c]) + n[2][2][c])
                left = ex.rn(ex.fma(two, n[1][0][c], n[0][0][c]) + n[2][0][c])
                below = ex.rn(ex.fma(two, n[2][1][c], n[2][0][c]) + n[2][2][c])
                above = ex.rn(ex.fma(two, n[0][1][c], n[0][0][c]) + n[0][2][c])
                g = ex.rn(abs(ex.rn(right - left)) + abs(ex.rn(below - above)))
                t.append(ex.clamp01(ex.rn(scale * g)))
            row.append(t + [n[1][1][3]])
        out.append(row)
    return out


def invert(img, enabled, strength):
    if not enabled:
        return img
    s = ex.f32(strength)
    return [[[ex.fma(s, ex.rn(ex.rn(1 - t[c]) - t[c]), t[c]) for c in range(3)] + [t[3]] for t in row] for row in img]


def want_chain(x, fmt, scale, strength):
    a = ex.node(ex.gaussian, fmt, [to_img(x)], 1.0, 2)
    b = ex.node(sobel, fmt, [a], scale)
    return from_img(ex.node(invert, fmt, [b], True, strength), x.dtype)


CHAIN = """input -> blur -> edges -> neg -> output
blur:  gaussian5   { sigma: 1.0 }
edges: edge_detect { scale: %s }
neg:   invert      { enabled: true, strength: %s }
"""


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_user_stages_fused_and_unfused_match_the_restatement(ctx, stage_dir, fmt, tag):
    W, H = 83, 37                                             # ragged: not a multiple of anything
    x = util.synthetic(W, H, fmt, 77)
    want = want_chain(x, tag, 0.75, 1.0)
    text = CHAIN % ("0.75", "1.0")
    assert rf.Plan(rf.Config(text)).launches() == ["blur+edges+neg"]
    for flags in (0, rf.RF_GRAPH_NO_FUSION, rf.RF_GRAPH_NO_JIT):
        for ex_flags in (0, rf.RF_EXEC_NO_ALTERNATE):
            got = util.run_hip(ctx, text, x, flags=flags, exec_flags=ex_flags, rows_per_chunk=12)
            util.assert_same(got, want, "user chain %s flags=%d exec=%d" % (tag, flags, ex_flags))


def test_bool_parameter_and_single_nodes(ctx, stage_dir):
    x = util.synthetic(70, 29, util.F32, 5)
    off = util.run_hip(ctx, "input -> nn -> output\nnn: invert { enabled: false, strength: 1.0 }", x)
    assert off.tobytes() == x.tobytes()
    half = util.run_hip(ctx, "input -> nn -> output\nnn: invert { enabled: true, strength: 0.5 }", x)
    util.assert_same(half, from_img(invert(to_img(x), True, 0.5), np.float32), "invert 0.5")
    edges = util.run_hip(ctx, "input -> ee -> output\nee: edge_detect { scale: 2.0 }", x, rows_per_chunk=8)
    util.assert_same(edges, from_img(sobel(to_img(x), 2.0), np.float32), "edge_detect alone")


def test_the_reference_diamond_with_the_real_edge_detect(ctx, stage_dir):
    """pipeline_graph.rs:462-468: gaussian || edge_detect -> combination, as ONE fork/join launch"""
    text = ("input -> gaussian -> combination:input_image0\ninput -> edge_detect -> combination:input_image1\ncombination -> output\n"
            "gaussian: gaussian5 { sigma: 1.5 }\nedge_detect: edge_detect { scale: 1.0 }\ncombination: combination { mix: 0.25 }")
    x = util.synthetic(96, 33, util.F32, 9)
    xi = to_img(x)
    want = from_img(ex.combination(ex.gaussian(xi, 1.5, 2), sobel(xi, 1.0), 0.25), np.float32)
    for flags in (0, rf.RF_GRAPH_NO_FUSION):
        util.assert_same(util.run_hip(ctx, text, x, flags=flags), want, "diamond flags=%d" % flags)


def test_an_edited_stage_file_is_picked_up_and_a_broken_one_keeps_the_old_graph(ctx, stage_dir, tmp_path):
    W, H = 64, 24
    cfg_path = tmp_path / "graph.rf"
    cfg_path.write_text("input -> nn -> output\nnn: invert { enabled: true, strength: 1.0 }")
    r = rf.Render(rf.RenderInfo(W, H, config_path=str(cfg_path), shader_path=str(stage_dir), format=rf.RF_FORMAT_RGBA32F), ctx)
    x = util.synthetic(W, H, util.F32, 3)
    r.graph.upload_raw(x)
    r.graph.execute(); r.graph.wait()
    util.assert_same(r.graph.download_raw(), from_img(invert(to_img(x), True, 1.0), np.float32), "before the edit")
    # edit: the negative becomes a passthrough-with-offset; the reload loop sees the new mtime and rebuilds
    f = stage_dir / "invert.stage.hip"
    f.write_text(f.read_text().replace("fmaf(p.strength, (1.0f - c.x) - c.x, c.x)", "c.x + 0.25f"))
    st = os.stat(f)
    os.utime(f, ns=(st.st_mtime_ns + 10 ** 9, st.st_mtime_ns + 10 ** 9))
    assert r.trigger_reloads() is True
    r.graph.upload_raw(x)
    r.graph.execute(); r.graph.wait()
    got = r.graph.download_raw()
    assert np.array_equal(got[..., 0], (x[..., 0] + np.float32(0.25)).astype(np.float32)) and not np.array_equal(got[..., 1], x[..., 1])
    # a file that no longer compiles: the rebuild fails, the graph that was running keeps running (render.rs:121-136)
    good = f.read_text()
    f.write_text(good.replace("c.x + 0.25f", "c.x + nonsense"))
    st = os.stat(f)
    os.utime(f, ns=(st.st_mtime_ns + 10 ** 9, st.st_mtime_ns + 10 ** 9))
    old_graph = r.graph
    assert r.trigger_reloads() is False and r.graph is old_graph
    r.graph.execute(); r.graph.wait()
    assert np.array_equal(r.graph.download_raw()[..., 0], got[..., 0])
    with pytest.raises(rf.RfError) as e:
        rf.Graph(ctx, rf.Config(cfg_path.read_text()), W, H, rf.RF_FORMAT_RGBA32F)
    assert e.value.status == 3 and "nonsense" in str(e.value)
    r.graph.close()


def test_user_stage_on_a_large_frame_as_strips_of_chunks(ctx, stage_dir):
    """a 1080p frame, fused with a gaussian, against the same graph run node by node (the restatement above is exact but slow)"""
    text = CHAIN % ("1.5", "0.5")
    x = util.synthetic(1920, 1080, util.F32, 11)
    a = util.run_hip(ctx, text, x)
    b = util.run_hip(ctx, text, x, flags=rf.RF_GRAPH_NO_FUSION)
    util.assert_same(a, b, "fused vs unfused at 1080p")


LUT_STAGE = """// a point op that indexes a LOCAL ARRAY with a value only the run knows: the array cannot live in registers
struct Params { float gain; int shift; };
static constexpr int RADIUS = 0;
RF_STAGE f4 apply(const Params& p, f4 c)
{
    float t[64];
    for (int i = 0; i < 64; ++i) t[i] = (c.x * (float)i) * p.gain;
    const int k = ((int)(c.y * 63.0f) + p.shift) & 63;
    return make_float4(t[k], c.y, c.z, c.w);
}
"""


def test_a_user_stage_that_needs_scratch_memory_still_runs(ctx, stage_dir):
    """stream_prepare refused every run-time compiled kernel with scratch, so a valid stage file whose apply() indexes
    a local array at run time made rf_graph_create fail -- the reference runs every shader that compiles (shader.rs:29-93).  A
    launch that is ONE user stage is now kept whatever it spills (rf_graph_note says so); only fused chains are cut on a spill."""
    (stage_dir / "lut.stage.hip").write_text(LUT_STAGE)
    text = "input -> lut -> output\nlut: lut { gain: 0.5, shift: 3 }"
    W, H = 211, 37
    x = util.synthetic(W, H, util.F32, seed=41)
    g = rf.Graph(ctx, rf.Config(text), W, H, util.F32)
    try:
        assert g.note == "" or "spills" in g.note, g.note
        g.upload_raw(x)
        g.execute()
        g.wait()
        got = g.download_raw()
    finally:
        g.close()
    k = (((x[..., 1] * np.float32(63.0)).astype(np.int32) + 3) & 63).astype(np.float32)
    want = x.copy()
    want[..., 0] = (x[..., 0] * k) * np.float32(0.5)
    util.assert_same(got, want, "lut stage")
    # fused behind a built-in node the chain is either one launch or, when the fused kernel spills, cut -- the result is the same
    text2 = "input -> gg -> lut -> output\ngg: colour_grade { slope: 1.0, offset: 0.0, saturation: 1.0 }\nlut: lut { gain: 0.5, shift: 3 }"
    g = rf.Graph(ctx, rf.Config(text2), W, H, util.F32)
    try:
        g.upload_raw(x)
        g.execute()
        g.wait()
        got2 = g.download_raw()
    finally:
        g.close()
    ref = rf.Graph(ctx, rf.Config(text2), W, H, util.F32, flags=rf.RF_GRAPH_NO_FUSION)
    try:
        ref.upload_raw(x)
        ref.execute()
        ref.wait()
        util.assert_same(got2, ref.download_raw(), "lut stage fused vs unfused")
    finally:
        ref.close()
