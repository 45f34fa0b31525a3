"""GPU: user NODES -- stage files that declare their images (RF_INPUTS / RF_OUTPUTS): several input images, several output
images, written in place when a name is on both sides.  Checked against an independent restatement in exact rationals
(tests/golden/exact_eval.py), against numpy where numpy's float32 arithmetic IS the specification (one subtraction), and
through size-independent properties at 4K.  Ref: src/vulkan/shader.rs:151-153, src/vulkan/pipeline_graph.rs:205-236,:402-406."""
import json
import os
import shutil

import numpy as np
import pytest

import reforge_amd as rf
from tests import util
from tests.golden import exact_eval as ex
from tests.test_gpu_user_stage import from_img, to_img
from tests.test_user_node import BOX_GRAPH, BOX_WEIGHTS, CURVE, GUIDED, TINT, TINT_GRAPH, UNSHARP, UNSHARP_BOTH, WINDOW_GRAPH

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
    for f in ("unsharp_mask.stage.hip", "tone_curve.stage.hip", "apply_curve.stage.hip", "local_contrast.stage.hip", "streak.stage.hip"):
        shutil.copy(os.path.join(SHADERS, f), tmp_path / f)
    (tmp_path / "box_weights.stage.hip").write_text(BOX_WEIGHTS)
    (tmp_path / "guided.stage.hip").write_text(GUIDED)
    (tmp_path / "tint.stage.hip").write_text(TINT)
    old = rf.shader_path()
    rf.set_shader_path(str(tmp_path))
    yield tmp_path
    rf.set_shader_path(old)


# ---- the restatements: shaders/unsharp_mask.stage.hip and the `tint` of tests/test_user_node.py in exact arithmetic --------
def unsharp(img, blurred, amount, threshold):
    """-> (output_image, mask_image)"""
    amount, threshold = ex.f32(amount), ex.f32(threshold)
    out, mask = [], []
    for ri, rb in zip(img, blurred):
        o_row, m_row = [], []
        for t, b in zip(ri, rb):
            o, m = [], []
            for c in range(3):
                d = ex.rn(t[c] - b[c])
                on = abs(d) >= threshold
                o.append(ex.fma(amount, d, t[c]) if on else t[c])
                m.append(abs(d) if on else ex.ZERO)
            o_row.append(o + [t[3]])
            m_row.append(m + [ex.ONE])
        out.append(o_row)
        mask.append(m_row)
    return out, mask


def tint(img, other, strength):
    s = ex.f32(strength)
    return [[[ex.fma(s, ex.rn(b[c] - a[c]), a[c]) for c in range(3)] + [a[3]] for a, b in zip(ra, rb)] for ra, rb in zip(img, other)]


def tone_curve(gamma, lift):
    """shaders/tone_curve.stage.hip: the 256 floats fill() produces"""
    gamma, lift, k = ex.f32(gamma), ex.f32(lift), ex.f32(0.003921569)
    out = []
    for i in range(256):
        t = ex.rn(i * k)
        c = ex.fma(gamma, ex.rn(ex.rn(t * t) - t), t)
        out.append(ex.fma(lift, ex.rn(1 - c), c))
    return out


def apply_curve(img, curve, strength):
    """shaders/apply_curve.stage.hip"""
    s = ex.f32(strength)

    def through(c):
        x = ex.rn(ex.clamp01(c) * 255)
        i = min(int(x), 254)                       # x >= 0: truncation is floor
        f = ex.rn(x - i)
        m = ex.fma(f, ex.rn(curve[i + 1] - curve[i]), curve[i])
        return ex.fma(s, ex.rn(m - c), c)
    return [[[through(t[c]) for c in range(3)] + [t[3]] for t in row] for row in img]


def want_curve(x, fmt, gamma, lift, strength):
    xi = to_img(x)
    through = ex.store(ex.load(xi, fmt), fmt)       # tone_curve passes its image through (a load and a store)
    return from_img(ex.store(apply_curve(ex.load(through, fmt), tone_curve(gamma, lift), strength), fmt), x.dtype)


def local_contrast(img, amount):
    """shaders/local_contrast.stage.hip: RADIUS 2, the 5x5 mean accumulated row by row, left to right"""
    amount, k = ex.f32(amount), ex.f32(0.04)
    out = []
    for y in range(len(img)):
        row = []
        for x in range(len(img[0])):
            acc = [ex.ZERO] * 3
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    t = ex.at(img, x + dx, y + dy)
                    acc = [ex.fma(k, t[c], acc[c]) for c in range(3)]
            c0 = img[y][x]
            row.append([ex.fma(amount, ex.rn(c0[c] - acc[c]), c0[c]) for c in range(3)] + [c0[3]])
        out.append(row)
    return out


def streak(img, amount):
    """shaders/streak.stage.hip: RADIUS 7, the 28 texels of a four-pointed star, distance by distance: left, right, above, below"""
    amount, k = ex.f32(amount), ex.f32(1.0 / 28.0)
    out = []
    for y in range(len(img)):
        row = []
        for x in range(len(img[0])):
            acc = [ex.ZERO] * 3
            for d in range(1, 8):
                for t in (ex.at(img, x - d, y), ex.at(img, x + d, y), ex.at(img, x, y - d), ex.at(img, x, y + d)):
                    acc = [ex.fma(k, t[c], acc[c]) for c in range(3)]
            c0 = img[y][x]
            row.append([ex.fma(amount, ex.rn(acc[c] - c0[c]), c0[c]) for c in range(3)] + [c0[3]])
        out.append(row)
    return out


def guided(base, guide, strength):
    """the GUIDED file of tests/test_user_node.py: two inputs read through windows"""
    s = ex.f32(strength)
    out = []
    for y in range(len(base)):
        row = []
        for x in range(len(base[0])):
            c0, e, w, so, n = base[y][x], ex.at(guide, x + 1, y), ex.at(guide, x - 1, y), ex.at(guide, x, y + 1), ex.at(guide, x, y - 1)
            row.append([ex.fma(s, ex.rn(ex.rn(e[c] - w[c]) + ex.rn(so[c] - n[c])), c0[c]) for c in range(3)] + [c0[3]])
        out.append(row)
    return out


def want_window_graph(x, fmt):
    xi = to_img(x)
    gg = ex.node(ex.gaussian, fmt, [xi], 1.0, 2)
    lc = ex.node(local_contrast, fmt, [gg], 0.8)
    sh = ex.node(ex.sharpen, fmt, [xi], 0.4)
    return from_img(ex.node(guided, fmt, [lc, sh], 0.25), x.dtype)


def want_unsharp_both(x, fmt):
    xi = to_img(x)
    blur = ex.node(ex.gaussian, fmt, [xi], 1.0, 2)
    o, m = unsharp(ex.load(xi, fmt), ex.load(blur, fmt), 0.8, 0.05)
    o, m = ex.store(o, fmt), ex.store(m, fmt)
    gg = ex.node(ex.colour_grade, fmt, [m], 1.5, 0.0, 1.0)
    return from_img(ex.node(ex.combination, fmt, [o, gg], 0.25), x.dtype)


def want_tint(x, fmt):
    xi = to_img(x)
    aa = ex.node(ex.gaussian, fmt, [xi], 1.0, 2)
    cc = ex.node(ex.colour_grade, fmt, [xi], 0.5, 0.1, 0.0)
    tt = ex.node(tint, fmt, [aa, cc], 0.3)
    return from_img(ex.node(ex.sharpen, fmt, [tt], 0.5), x.dtype)


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_two_inputs_two_outputs_match_the_restatement(ctx, stage_dir, fmt, tag):
    W, H = 71, 23                                             # ragged: the last workgroup of a row is partly outside the frame
    x = util.synthetic(W, H, fmt, 41)
    want = want_unsharp_both(x, tag)
    for flags in (0, rf.RF_GRAPH_NO_FUSION, rf.RF_GRAPH_HIPGRAPH):
        util.assert_same(util.run_hip(ctx, UNSHARP_BOTH, x, flags=flags), want, "unsharp_mask, both outputs read, %s flags=%d" % (tag, flags))
    # a later frame slot, and layers on side streams: the node's launch takes its images from the slot it runs in
    util.assert_same(util.run_hip(ctx, UNSHARP_BOTH, x, num_frames=3, slot=2, exec_flags=rf.RF_EXEC_CONCURRENT_LAYERS), want, "slot 2, concurrent layers")


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_an_output_the_graph_leaves_unwired_is_not_stored(ctx, stage_dir, fmt, tag):
    x = util.synthetic(64, 19, fmt, 43)
    xi = to_img(x)
    blur = ex.node(ex.gaussian, tag, [xi], 2.0, 4)
    o, m = unsharp(ex.load(xi, tag), ex.load(blur, tag), 1.5, 0.02)
    util.assert_same(util.run_hip(ctx, UNSHARP, x), from_img(ex.store(o, tag), x.dtype), "output_image only")
    only_mask = UNSHARP.replace("um -> output", "um:mask_image -> output")
    util.assert_same(util.run_hip(ctx, only_mask, x), from_img(ex.store(m, tag), x.dtype), "mask_image only")


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_a_node_that_writes_its_first_input_in_place(ctx, stage_dir, fmt, tag):
    x = util.synthetic(67, 21, fmt, 47)
    want = want_tint(x, tag)
    assert rf.Plan(rf.Config(TINT_GRAPH)).launch_info()[-2]["outputs"] == rf.Plan(rf.Config(TINT_GRAPH)).launch_info()[-2]["inputs"][:1]
    for flags in (0, rf.RF_GRAPH_NO_FUSION):
        util.assert_same(util.run_hip(ctx, TINT_GRAPH, x, flags=flags), want, "tint in place %s flags=%d" % (tag, flags))
    # the second frame of an in-place graph starts from a fresh upload, not from what frame one left in the image
    g = rf.Graph(ctx, rf.Config(TINT_GRAPH), 67, 21, fmt)
    try:
        for _ in range(2):
            g.upload_raw(x)
            g.execute(); g.wait()
            util.assert_same(g.download_raw(), want, "frame after frame")
    finally:
        g.close()


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_nodes_that_read_through_windows_match_the_restatement(ctx, stage_dir, fmt, tag):
    """RADIUS 2 (local_contrast: one input) and RADIUS 1 with two declared inputs (guided): clamp-to-edge at all four frame edges
    (the frame is narrower than a workgroup and only 14 rows high), the interior / boundary split of an exchange-mode strip"""
    W, H = 53, 14
    x = util.synthetic(W, H, fmt, 71)
    want = want_window_graph(x, tag)
    for flags, ex_flags in ((0, 0), (rf.RF_GRAPH_NO_FUSION, 0), (rf.RF_GRAPH_HIPGRAPH, 0), (0, rf.RF_EXEC_FORCE_SPLIT), (0, rf.RF_EXEC_CONCURRENT_LAYERS)):
        util.assert_same(util.run_hip(ctx, WINDOW_GRAPH, x, flags=flags, exec_flags=ex_flags), want, "window nodes %s flags=%d exec=%d" % (tag, flags, ex_flags))
    alone = util.run_hip(ctx, "input -> lc -> output\nlc: local_contrast { amount: 1.5 }", x)
    util.assert_same(alone, from_img(ex.store(local_contrast(ex.load(to_img(x), tag), 1.5), tag), x.dtype), "local_contrast alone")


@pytest.mark.parametrize("text,world", [(UNSHARP_BOTH, 3), (TINT_GRAPH, 2), (WINDOW_GRAPH, 3), (WINDOW_GRAPH, 2)])
def test_user_nodes_in_row_strips(stage_dir, ctx, text, world):
    """over-fetch row strips on one GPU standing in for N ranks: the user node produces the ghost rows its readers
    want like any point op; the stacked strips equal the whole frame"""
    W, H = 150, 61
    for fmt in (util.F32, util.U8):
        whole = rf.Graph(ctx, rf.Config(text), W, H, fmt)
        whole.fill_synthetic(0x5EED0004)
        whole.execute(); whole.wait()
        want = whole.download_raw()
        whole.close()
        strips = []
        for rank in range(world):
            c = rf.Context(0, rank, world, None)
            g = rf.Graph(c, rf.Config(text), W, H, fmt, flags=rf.RF_GRAPH_NO_HALO_XCHG)
            g.fill_synthetic(0x5EED0004)
            g.execute(); g.wait()
            strips.append(g.download_raw())
            g.close()
            c.close()
        util.assert_same(np.concatenate(strips, axis=0), want, "world=%d fmt=%d" % (world, fmt))


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_a_buffer_filled_by_one_user_node_and_read_by_another(ctx, stage_dir, fmt, tag):
    """storage blocks by TYPE name (shader.rs:144-147): tone_curve fills ToneCurve on the device every frame, apply_curve reads it"""
    W, H = 77, 19
    x = util.synthetic(W, H, fmt, 61)
    want = want_curve(x, tag, 0.6, 0.05, 0.8)
    for flags in (0, rf.RF_GRAPH_HIPGRAPH):
        util.assert_same(util.run_hip(ctx, CURVE, x, flags=flags), want, "tone curve %s flags=%d" % (tag, flags))
    # a parameter of the FILLING node changes what the reading node computes on the next frame; with two frames in flight the
    # frames of such a graph are ordered one behind the other (the buffer is one per graph), so each slot sees its own frame's curve
    other = want_curve(x, tag, -0.4, 0.0, 0.8)
    g = rf.Graph(ctx, rf.Config(CURVE), W, H, fmt, num_frames=2)
    try:
        g.upload_raw(x)
        g.execute(0)
        g.set_param("tc", "gamma", -0.4)
        g.set_param("tc", "lift", 0.0)
        g.execute(1)
        g.wait(0); g.wait(1)
        util.assert_same(g.download_raw(0), want, "slot 0: the curve of the parameters it was submitted with")
        util.assert_same(g.download_raw(1), other, "slot 1: the edited curve")
    finally:
        g.close()


def test_user_written_weights_feed_the_builtin_conv2d(ctx, stage_dir):
    """a user node fills a ConvWeights block (961 floats, the leading K x K used); the built-in conv2d reads it through the edge
    `bw:ConvWeights -> cv:ConvWeights` exactly as it reads conv2d_weights' -- the oracle runs conv2d with the same weights"""
    for fmt in (util.F32, util.U8):
        x = util.synthetic(90, 41, fmt, 67)
        want = util.run_oracle("input -> cv -> output\ncv: conv2d { ksize: 5 }", x, weights={"cv": np.full((5, 5), np.float32(0.04), np.float32)})
        util.assert_same(util.run_hip(ctx, BOX_GRAPH, x), want, "box weights fmt=%d" % fmt)


def test_parameters_of_a_user_node_are_uniform_members(ctx, stage_dir):
    """rf_graph_set_param reaches the node's Params block (render.rs:167-210): amount 0 makes the node an identity on input_image"""
    x = util.synthetic(80, 16, util.F32, 53)
    g = rf.Graph(ctx, rf.Config(UNSHARP), 80, 16, util.F32)
    try:
        g.upload_raw(x)
        g.execute(); g.wait()
        sharpened = g.download_raw()
        assert sharpened.tobytes() != x.tobytes()
        g.set_param("um", "amount", 0.0)
        g.execute(); g.wait()
        assert g.download_raw().tobytes() == x.tobytes()
        g.set_param("um", "amount", 1.5)
        g.execute(); g.wait()
        assert g.download_raw().tobytes() == sharpened.tobytes()
    finally:
        g.close()


def test_user_node_at_4k_and_the_rate_it_streams_at(ctx, stage_dir):
    """3840 x 2160 rgba32f.  The mask is ONE float32 subtraction per channel, which numpy computes as the device does; the
    blurred image comes from the oracle.  A huge threshold turns the node into a copy of input_image (every texel checked).
    Prints the launch's rate: 2 images read + 2 written, every byte once."""
    W, H = 3840, 2160
    x = util.synthetic(W, H, util.F32, 59)
    blurred = util.run_oracle("input -> blur -> output\nblur: gaussian9 { sigma: 2.0 }", x)
    d = x - blurred
    thr = np.float32(0.02)
    on = np.abs(d) >= thr
    want_mask = np.where(on, np.abs(d), np.float32(0)).astype(np.float32)
    want_mask[..., 3] = 1.0
    got_mask = util.run_hip(ctx, UNSHARP.replace("um -> output", "um:mask_image -> output"), x)
    util.assert_same(got_mask, want_mask, "mask_image at 4K")
    ident = util.run_hip(ctx, UNSHARP.replace("threshold: 0.02", "threshold: 1000.0"), x)
    assert ident.tobytes() == x.tobytes()
    # where the mask is off the output is the input, bit for bit; where it is on it differs from it (amount 1.5, d != 0)
    out = util.run_hip(ctx, UNSHARP, x)
    assert np.array_equal(out[..., :3][~on[..., :3]], x[..., :3][~on[..., :3]]) and np.array_equal(out[..., 3], x[..., 3])
    assert np.all(out[..., :3][on[..., :3]] != x[..., :3][on[..., :3]])
    g = rf.Graph(ctx, rf.Config(UNSHARP_BOTH), W, H, util.F32)
    try:
        g.fill_synthetic(1)
        g.execute(); g.wait()
        k = g.plan.launches().index("um")
        ms = min(g.time_launch(k, 50) for _ in range(3))
        gbps = 4 * W * H * 16 / (ms * 1e-3) / 1e9
        print("\nuser node unsharp_mask 3840x2160 rgba32f: %.4f ms per launch, %.0f GB/s (2 images in, 2 out)" % (ms, gbps))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "user_node_rate.json"), "w") as f:
            json.dump({"workload": "unsharp_mask user node, 3840x2160 rgba32f, 2 images in + 2 out", "ms_per_launch": round(ms, 5), "gbps": round(gbps, 1),
                       "algorithmic_bytes": 4 * W * H * 16}, f)
        assert gbps > 1500.0                                  # a point op that moved every byte twice, or ran one wave per row, would not
    finally:
        g.close()
    # the window fallback (RADIUS 2: 25 cached loads per texel) for the record: one image in, one out
    g = rf.Graph(ctx, rf.Config("input -> lc -> output\nlc: local_contrast { amount: 0.8 }"), W, H, util.F32)
    try:
        g.fill_synthetic(1)
        g.execute(); g.wait()
        ms = min(g.time_launch(0, 30) for _ in range(3))
        print("user node local_contrast (RADIUS 2, window) 3840x2160 rgba32f: %.4f ms per launch, %.0f GB/s algorithmic" % (ms, 2 * W * H * 16 / (ms * 1e-3) / 1e9))
        rec = json.load(open(os.path.join(ROOT, "gpurun_out", "user_node_rate.json")))
        rec["local_contrast_radius2_window"] = {"ms_per_launch": round(ms, 5), "gbps_algorithmic": round(2 * W * H * 16 / (ms * 1e-3) / 1e9, 1)}
        with open(os.path.join(ROOT, "gpurun_out", "user_node_rate.json"), "w") as f:
            json.dump(rec, f)
    finally:
        g.close()


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_window_nodes_across_tiles(ctx, stage_dir, fmt, tag):
    """the LDS-tiled kernel of window nodes (rf_user_dev.h): frames of several 64 x TH tiles in both directions with partial
    tiles at the right and bottom edges, RADIUS 2 (register window slid down a thread's outputs) and RADIUS 7 (taps read from
    the tile), whole and as the interior / boundary parts of a split launch; against the exact restatements"""
    for W, H in ((200, 77), (64, 32), (65, 33), (129, 16)):
        x = util.synthetic(W, H, fmt, 5 + W)
        xi = ex.load(to_img(x), tag)
        want_lc = from_img(ex.store(local_contrast(xi, 0.8), tag), x.dtype)
        want_st = from_img(ex.store(streak(xi, 0.6), tag), x.dtype)
        for ex_flags in (0, rf.RF_EXEC_FORCE_SPLIT):
            util.assert_same(util.run_hip(ctx, "input -> lc -> output\nlc: local_contrast { amount: 0.8 }", x, exec_flags=ex_flags), want_lc, "local_contrast %dx%d %s" % (W, H, tag))
            util.assert_same(util.run_hip(ctx, "input -> st -> output\nst: streak { amount: 0.6 }", x, exec_flags=ex_flags), want_st, "streak %dx%d %s" % (W, H, tag))


def test_window_node_rates_at_4k(ctx, stage_dir):
    """the rates DESIGN.md quotes for window nodes (3840 x 2160 rgba32f, 2 x 132.7 MB per launch): recorded, and held to a floor
    that the round-3 kernel (a clamped global load per tap: 0.30 of 8 TB/s) did not reach"""
    import json
    out = {}
    for name, text in (("local_contrast", "input -> lc -> output\nlc: local_contrast { amount: 0.8 }"), ("streak", "input -> st -> output\nst: streak { amount: 0.6 }")):
        g = rf.Graph(ctx, rf.Config(text), 3840, 2160, util.F32)
        try:
            g.fill_synthetic(3)
            g.execute()
            g.wait()
            ms = min(g.time_launch(0, 50) for _ in range(3))
        finally:
            g.close()
        out[name] = {"ms": round(ms, 5), "frac_of_8TBs": round(2 * 3840 * 2160 * 16 / (ms * 1e-3) / 8e12, 4)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "user_window_rate.json"), "w"), indent=1)
    print(out)
    assert out["local_contrast"]["frac_of_8TBs"] > 0.45 and out["streak"]["frac_of_8TBs"] > 0.35, out
