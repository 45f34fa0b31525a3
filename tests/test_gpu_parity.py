"""GPU (MI355X): the HIP path, called through the C ABI (librfhip.so), against the CPU
oracle on the same seeded inputs, against hand-derivable known answers and against the
committed golden vectors.  Bit-exact for rgba8 AND rgba32f (the stated bar is
pixel-exact rgba8 and <= 1 ulp rgba32f; the kernels use the oracle's exact fmaf order,
so the tests hold them to 0 ulp).  One process, one rf_ctx for the whole session."""
import os

import numpy as np
import pytest

import reforge_amd as rf
from oracle import graph as og
from oracle import pixel
from tests import kat, util

pytestmark = pytest.mark.gpu

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))
NF = rf.RF_GRAPH_NO_FUSION


def test_the_hip_library_is_what_runs(ctx):
    assert ctx.arch.startswith("gfx950")
    with open("/proc/self/maps") as fh:
        maps = fh.read()
    assert "librfhip.so" in maps


def test_rccl_loads_and_round_trips():
    """One-rank communicator, grouped send+recv to self: librccl is loadable on the box and
    is called with the right ABI.  (The multi-rank pattern is tests/test_dist_gloo.py.)"""
    rf.comm_selftest(0, 3 * 3840 * 16)


# ---- known answers (same checks the oracle passes in test_oracle.py) ----------------
@pytest.fixture()
def run(ctx):
    def _run(text, img, weights=None):
        return util.run_hip(ctx, text, img, weights=weights)
    return _run


@pytest.mark.parametrize("fmt", [util.F32, util.U8])
@pytest.mark.parametrize("W,H", kat.PASSTHROUGH_SIZES)
def test_passthrough_identity(run, fmt, W, H):
    kat.check_passthrough_identity(run, fmt, W, H)


def test_passthrough_special_floats(run):
    kat.check_passthrough_preserves_special_floats(run)


def test_unorm8_decode_all_codes(run):
    """imageLoad of every UNORM8 code, observed through a gaussian delta kernel on rgba32f-free
    arithmetic: an rgba8 grade with slope 1 reproduces c/255 -> store exactly."""
    c = np.arange(256, dtype=np.uint8).reshape(1, 64, 4)
    out = run("input -> gaussian5 -> output", np.ascontiguousarray(c))      # sigma absent: delta
    assert (out == c).all()


def test_impulse_responses(run):
    kat.check_gaussian_impulse(run, GOLDEN)
    kat.check_gaussian9_weights(run, GOLDEN)
    kat.check_sharpen_impulse(run)
    kat.check_conv_impulse_is_flipped_kernel(run)


@pytest.mark.parametrize("fmt", [util.F32, util.U8])
def test_degenerate_parameters_are_identity(run, fmt):
    kat.check_gaussian_delta_is_identity(run, fmt)
    kat.check_sharpen_zero_is_identity(run, fmt)


def test_grade_properties(run):
    kat.check_grade_saturation_zero_is_grey(run)
    kat.check_unorm8_store_rounds_to_even(run)


# ---- golden vectors -------------------------------------------------------------------
@pytest.mark.parametrize("flags", [0, NF])
@pytest.mark.parametrize("tag", ["f32", "u8"])
def test_golden_vectors(ctx, tag, flags):
    x = GOLDEN["in_" + tag]
    for name, text in (("chain3", util.CHAIN3), ("chain5", util.CHAIN5), ("diamond", util.DIAMOND),
                       ("gauss9", "input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }"),
                       ("conv7", "input -> conv2d -> output\nconv2d: conv2d { ksize: 7, sigma: 1.5 }")):
        util.assert_same(util.run_hip(ctx, text, x, flags=flags), GOLDEN["%s_%s" % (name, tag)], "%s %s flags=%d" % (name, tag, flags))
    from tests.golden import exact_eval as ex
    for name, (_fn, text) in ex.MORE_GRAPHS.items():
        util.assert_same(util.run_hip(ctx, text or util.SPLIT2, x, flags=flags), GOLDEN["%s_%s" % (name, tag)], "%s %s flags=%d" % (name, tag, flags))


# ---- oracle parity on seeded frames: every node type, ragged sizes, chunk seams ---------
NODES = {
    "passthrough": "input -> passthrough -> output",
    "gaussian5": "input -> gaussian5 -> output\ngaussian5: gaussian5 { sigma: 1.0 }",
    "gaussian9": "input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }",
    "gaussian_r1": "input -> gg -> output\ngg: gaussian { sigma: 0.8, radius: 1 }",
    "gaussian_r7": "input -> gg -> output\ngg: gaussian { sigma: 3.0, radius: 7 }",
    "gaussian_r15": "input -> gg -> output\ngg: gaussian { sigma: 6.0, radius: 15 }",
    "grade": "input -> grade -> output\ngrade: grade { slope: 1.3, offset: -0.1, saturation: 0.6 }",
    "sharpen": "input -> sharpen -> output\nsharpen: sharpen { amount: 1.25 }",
    "conv3": "input -> conv2d -> output\nconv2d: conv2d { ksize: 3, sigma: 0.7 }",
    "conv9": "input -> conv2d -> output\nconv2d: conv2d { ksize: 9, sigma: 2.0 }",
    "chain3": util.CHAIN3,
    "chain5": util.CHAIN5,
    "chain5_split": util.CHAIN5_SPLIT,
    "diamond": util.DIAMOND,
    "inplace": "input -> gaussian5 -> colour_grade:image -> sharpen -> output\n"
               "gaussian5: gaussian5 { sigma: 1.2 }\ncolour_grade: colour_grade { slope: 0.8, offset: 0.1, saturation: 1.5 }\n"
               "sharpen: sharpen { amount: 0.4 }",
}
# widths around the strip seams (a 64-lane strip yields 64-2r columns), heights around chunks
SIZES = [(1, 1), (2, 3), (5, 1), (1, 9), (17, 13), (59, 7), (60, 33), (61, 64), (121, 35), (250, 131)]


@pytest.mark.parametrize("fmt", [util.F32, util.U8])
@pytest.mark.parametrize("name", sorted(NODES))
def test_node_parity_ragged_sizes(ctx, name, fmt):
    for W, H in SIZES:
        x = util.synthetic(W, H, fmt, seed=0x5EED0000 + W * 131 + H)
        want = util.run_oracle(NODES[name], x)
        util.assert_same(util.run_hip(ctx, NODES[name], x), want, "%s %dx%d fused" % (name, W, H))
        util.assert_same(util.run_hip(ctx, NODES[name], x, flags=NF), want, "%s %dx%d unfused" % (name, W, H))


PAIR_TYPES = {
    "g5": ("gaussian5", "{ sigma: 1.1 }"), "g9": ("gaussian9", "{ sigma: 2.2 }"),
    "gr": ("colour_grade", "{ slope: 0.9, offset: 0.04, saturation: 1.4 }"), "sh": ("sharpen", "{ amount: 0.7 }"),
}


@pytest.mark.parametrize("first", sorted(PAIR_TYPES))
@pytest.mark.parametrize("second", sorted(PAIR_TYPES))
def test_every_fused_pair(ctx, first, second):
    """Every pair of fusable node kinds runs as ONE launch and equals node-at-a-time execution
    and the oracle, for both formats, across chunk seams (both walk directions)."""
    text = "input -> n1 -> n2 -> output\nn1: %s %s\nn2: %s %s" % (PAIR_TYPES[first] + PAIR_TYPES[second])
    assert rf.Plan(rf.Config(text), 0).launches() == ["n1+n2"]
    for fmt in (util.F32, util.U8):
        x = util.synthetic(157, 83, fmt, seed=31)
        want = util.run_oracle(text, x)
        util.assert_same(util.run_hip(ctx, text, x, rows_per_chunk=11), want, "%s+%s fused" % (first, second))
        util.assert_same(util.run_hip(ctx, text, x, flags=NF), want, "%s+%s unfused" % (first, second))


@pytest.mark.parametrize("rows_per_chunk", [1, 2, 3, 7, 16, 1000])
@pytest.mark.parametrize("name", ["gaussian9", "chain3", "chain5", "sharpen"])
def test_chunk_seams(ctx, name, rows_per_chunk):
    """Every vertical chunk re-primes its rolling windows: seams at every possible phase."""
    W, H = 130, 45
    for fmt in (util.F32, util.U8):
        x = util.synthetic(W, H, fmt, seed=11)
        want = util.run_oracle(NODES[name], x)
        util.assert_same(util.run_hip(ctx, NODES[name], x, rows_per_chunk=rows_per_chunk), want,
                         "%s rpc=%d" % (name, rows_per_chunk))


def test_medium_frame_all_paths(ctx):
    """640x360: large enough for several workgroups per strip row and many chunks."""
    for fmt in (util.F32, util.U8):
        x = util.synthetic(640, 360, fmt)
        for text in (util.CHAIN3, util.CHAIN5, util.CHAIN5_SPLIT, util.DIAMOND):
            want = util.run_oracle(text, x)
            for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH, NF | rf.RF_GRAPH_HIPGRAPH, rf.RF_GRAPH_TIMERS):
                util.assert_same(util.run_hip(ctx, text, x, flags=flags), want, "flags=%d" % flags)


def test_structured_frame_and_fills(ctx):
    """Device-side generators equal the oracle's, and the ramp+impulse frame goes through the chain."""
    for fmt in (util.F32, util.U8):
        W, H = 300, 170
        g = rf.Graph(ctx, rf.Config("input -> passthrough -> output"), W, H, fmt)
        g.fill_synthetic(0x5EED0001)
        g.execute(); g.wait()
        assert g.download_raw().tobytes() == pixel.fill_synthetic(W, H, fmt, 0x5EED0001).tobytes()
        g.fill_structured()
        g.execute(); g.wait()
        s = pixel.fill_structured(W, H, fmt)
        assert g.download_raw().tobytes() == s.tobytes()
        g.close()
        util.assert_same(util.run_hip(ctx, util.CHAIN3, s), util.run_oracle(util.CHAIN3, s), "structured chain3")


def test_conv2d_31x31_and_custom_weights(ctx):
    rng = np.random.RandomState(3)
    w = rng.uniform(-0.05, 0.05, (31, 31)).astype(np.float32)
    text = "input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }"
    for fmt in (util.F32, util.U8):
        x = util.synthetic(97, 50, fmt)
        util.assert_same(util.run_hip(ctx, text, x), util.run_oracle(text, x), "conv31 default weights")
        util.assert_same(util.run_hip(ctx, text, x, weights={"conv2d": w}), util.run_oracle(text, x, {"conv2d": w}), "conv31 custom")


def test_non_finite_texels_propagate_like_the_oracle(ctx):
    """inf / nan / -0 / subnormal texels through the stencil and point stages: the same texels
    are NaN on both sides and every other texel is bit-identical (NaN payloads are not compared;
    the clamp of the colour grade maps NaN to 0 on both sides)."""
    x = util.synthetic(90, 40, util.F32, seed=99)
    x[5, 7] = [np.inf, -np.inf, np.nan, 1.0]
    x[20, 60] = [-0.0, 1e-42, -1e-42, 3e38]
    x[39, 89] = [np.nan, 0.5, np.inf, -np.inf]
    for name in ("gaussian5", "gaussian9", "sharpen", "grade", "chain3", "chain5", "passthrough", "conv3", "conv9", "gaussian_r7"):
        want = util.run_oracle(NODES[name], x)
        for flags in (0, NF):
            got = util.run_hip(ctx, NODES[name], x, flags=flags)
            assert (np.isnan(got) == np.isnan(want)).all(), name
            ok = np.isnan(want) | (got.view(np.uint32) == want.view(np.uint32))
            assert ok.all(), "%s flags=%d: %d finite texels differ" % (name, flags, (~ok).sum())


@pytest.mark.parametrize("path", [1, 2, 3])
def test_conv2d_every_kernel_path(ctx, path):
    """The three conv2d kernels (1 = 16x16 LDS tile, 2 = banded MFMA, 3 = register-blocked VALU)
    are all bit-identical to the oracle: ragged widths around the 64/128-column strips, heights
    around the 16/32-row steps (several steps per chunk: the register-prefetched ring refill),
    frame edges inside the halo.  K < 9 has no MFMA kernel: path 2 then takes the VALU kernel.
    Selected through rf_graph_options."""
    for K, sigma in ((3, 0.8), (5, 1.0), (9, 1.5), (13, 2.0), (21, 3.5), (31, 5.0)):
        text = "input -> conv2d -> output\nconv2d: conv2d { ksize: %d, sigma: %.1f }" % (K, sigma)
        rng = np.random.RandomState(K)
        w = rng.uniform(-0.03, 0.03, (K, K)).astype(np.float32)
        for fmt in (util.F32, util.U8):
            for W, H in ((7, 5), (65, 9), (129, 17), (200, 45), (131, 150)):
                x = util.synthetic(W, H, fmt, seed=K * 100 + W)
                util.assert_same(util.run_hip(ctx, text, x, weights={"conv2d": w}, conv_path=path, rows_per_chunk=64),
                                 util.run_oracle(text, x, {"conv2d": w}), "conv path %s K=%d %dx%d fmt=%d" % (path, K, W, H, fmt))


# ---- the sRGB boundary (render.rs:264-313, :406-433) ------------------------------------
def test_srgb_identity_and_lut(ctx):
    c = np.zeros((4, 256, 4), np.uint8)
    c[:, :, :] = np.arange(256)[None, :, None]
    for fmt, want_rgb in ((util.F32, np.arange(256)), (util.U8, GOLDEN["srgb_rgba8_roundtrip"])):
        g = rf.Graph(ctx, rf.Config("input -> passthrough -> output"), 256, 4, fmt)
        g.upload_srgb8(c)
        g.execute(); g.wait()
        out = g.download_srgb8()
        assert (out[..., 0] == want_rgb[None, :]).all() and (out[..., 3] == np.arange(256)[None, :]).all()
        lin = g.download_raw()
        assert lin.tobytes() == pixel.upload_srgb8(c, fmt).tobytes()
        g.close()


def test_srgb_chain_matches_oracle(ctx):
    rgba = pixel.fill_synthetic(211, 97, util.U8, 0xABCDEF)
    for fmt in (util.F32, util.U8):
        from oracle import graph as og
        ref = og.GraphOracle(util.CHAIN3, 211, 97, fmt)
        ref.upload_srgb8(rgba)
        ref.execute()
        g = rf.Graph(ctx, rf.Config(util.CHAIN3), 211, 97, fmt)
        g.upload_srgb8(rgba)
        g.execute(); g.wait()
        assert g.download_srgb8().tobytes() == ref.download_srgb8().tobytes()
        g.close()


def test_srgb_encode_edge_values(ctx):
    x = np.zeros((1, 8, 4), np.float32)
    x[0, :, 0] = [np.nan, -1.0, 0.0, 1.0, 2.0, np.inf, -np.inf, 0.5]
    x[0, :, 3] = [np.nan, -1.0, 0.0, 1.0, 2.0, 0.5, 0.25, 0.75]
    g = rf.Graph(ctx, rf.Config("input -> passthrough -> output"), 8, 1, util.F32)
    g.upload_raw(x)
    g.execute(); g.wait()
    assert g.download_srgb8().tobytes() == pixel.download_srgb8(x).tobytes()
    g.close()


# ---- executor behaviour ------------------------------------------------------------------
def test_parameters_timers_and_frames_in_flight(ctx):
    W, H, fmt = 200, 120, util.F32
    x = util.synthetic(W, H, fmt)
    g = rf.Graph(ctx, rf.Config(util.CHAIN3), W, H, fmt, num_frames=2, flags=rf.RF_GRAPH_TIMERS | NF)
    assert g.node_times(0) == []                       # nothing recorded yet (vkutils.rs:107-109)
    g.upload_raw(x)
    for slot in (0, 1, 0):
        g.execute(slot)
    g.wait(0); g.wait(1)
    want = util.run_oracle(util.CHAIN3, x)
    util.assert_same(g.download_raw(0), want, "slot 0")
    util.assert_same(g.download_raw(1), want, "slot 1")
    times = g.node_times(0)
    assert [n for n, _ in times] == ["blur", "grade", "sharp"] and all(0.0 < t < 50.0 for _, t in times)
    s = g.times_string(0)
    assert s.startswith("blur: ") and s.count("ms") == 3 and ", grade: " in s
    # update a uniform member (render.rs:167-210) and re-run
    g.set_param("sharp", "amount", 1.0)
    g.execute(0); g.wait(0)
    text2 = util.CHAIN3.replace("amount: 0.5", "amount: 1.0")
    util.assert_same(g.download_raw(0), util.run_oracle(text2, x), "after set_param")
    with pytest.raises(rf.RfError) as e:
        g.set_param("sharp", "nosuch", 1.0)
    assert e.value.status == 16
    # the intermediate images of the unfused graph are observable
    two = "input -> blur -> grade -> output\nblur: gaussian5 { sigma: 1.0 }\ngrade: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }"
    util.assert_same(g.download_image("grade:output_image", 1), util.run_oracle(two, x), "intermediate image")
    assert g.plan.resolve("rf:final-output") == "blur:output_image"     # the output reuses the first ping-pong image
    g.close()


def test_radius_change_needs_a_new_graph(ctx):
    g = rf.Graph(ctx, rf.Config("input -> gg -> output\ngg: gaussian { sigma: 1.0, radius: 2 }"), 64, 64, util.F32)
    with pytest.raises(rf.RfError) as e:
        g.set_param("gg", "radius", 3)
    assert e.value.status == 6
    g.set_param("gg", "sigma", 2.0)
    g.close()


def test_graph_errors(ctx):
    with pytest.raises(rf.RfError) as e:                # no input image wired (pipeline_graph.rs:236 panics)
        rf.Graph(ctx, rf.Config("sharpen -> output", expects_input=False), 32, 32, util.F32)
    assert e.value.status == 3
    with pytest.raises(rf.RfError):
        rf.Graph(ctx, rf.Config("input -> passthrough -> output"), 0, 32, util.F32)


def test_render_host_mirror(ctx, tmp_path):
    """The Render object drives the same call order as main.rs:134-182 (headless)."""
    cfg = tmp_path / "pipe.cfg"
    cfg.write_text(util.CHAIN3)
    info = rf.RenderInfo(width=96, height=64, config_path=str(cfg), format=util.F32)
    r = rf.Render(info, ctx)
    rgba = pixel.fill_synthetic(96, 64, util.U8, 42)
    r.staging_buffer()[...] = rgba
    out = r.render_frame().copy()
    from oracle import graph as og
    ref = og.GraphOracle(util.CHAIN3, 96, 64, util.F32)
    ref.upload_srgb8(rgba); ref.execute()
    assert out.tobytes() == ref.download_srgb8().tobytes()
    assert "ms" in r.last_frame_gpu_times()
    # hot reload: a broken config keeps the old graph; a fixed one replaces it (render.rs:121-165)
    cfg.write_text("input -> nosuchfilter -> output")
    os.utime(cfg, (1, 1))
    assert r.trigger_reloads() is False and r.graph is not None
    cfg.write_text("input -> passthrough -> output")
    os.utime(cfg, (2, 2))
    assert r.trigger_reloads() is True
    r.staging_buffer()[...] = rgba
    out2 = r.render_frame()
    assert out2.tobytes() == rgba.tobytes()             # sRGB identity through rgba32f
    r.graph.close()


# ---- row strips on real kernels (one GPU standing in for N ranks) --------------------------
@pytest.mark.parametrize("text,world,flags", [
    (util.CHAIN5, 2, 0), (util.CHAIN5, 3, NF), (util.CHAIN5_SPLIT, 3, 0), (util.CHAIN3, 4, 0), (util.DIAMOND, 2, 0),
    (util.SPLIT2, 3, 0), (util.SPLIT2, 2, NF),       # a node with two output images: both carry the ghost rows their readers want
])
def test_row_strips_overfetch_on_one_gpu(text, world, flags):
    """The N>1 path of rf_graph.cpp in over-fetch mode (RF_GRAPH_NO_HALO_XCHG: strips carry
    their cumulative halo, no communication per frame): every rank's context lives on GPU 0
    here, without a communicator; each generates its strip + ghost rows, runs the graph, and
    the stacked strips must equal the whole frame computed by the oracle."""
    W, H = 190, 97
    for fmt in (util.F32, util.U8):
        want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, 0x5EED0004))
        strips = []
        for rank in range(world):
            c = rf.Context(0, rank, world, None)
            g = rf.Graph(c, rf.Config(text), W, H, fmt, flags=flags | rf.RF_GRAPH_NO_HALO_XCHG)
            assert g.strip == rf.strip_rows(H, world, rank)
            g.fill_synthetic(0x5EED0004)
            g.execute(); g.wait()
            strips.append(g.download_raw())
            g.close()
            c.close()
        util.assert_same(np.concatenate(strips, axis=0), want, "world=%d flags=%d fmt=%d" % (world, flags, fmt))


def test_explicit_gaussian_weights_and_the_in_place_grade_type(ctx):
    """The members shaders/gaussian*.comp add to the plugin contract: explicit weights `w0 .. wR` (given => they replace
    the kernel derived from sigma) and the `colour_grade_inplace` type (one read-write image named `image`).  The HIP
    path must honour both exactly as the oracle does, fused and unfused."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import glsl_weights
    same = "input -> gg -> output\ngg: gaussian9 { sigma: 7.0, %s }" % glsl_weights.as_params(2.0, 4)          # the weights of sigma 2, not 7
    derived = "input -> gg -> output\ngg: gaussian9 { sigma: 2.0 }"
    odd = "input -> gg -> sh -> output\ngg: gaussian5 { sigma: 1.0, w0: 0.5, w1: 0.125, w2: 0.375 }\nsh: sharpen { amount: 0.3 }"   # not even normalised
    inplace = "input -> blur -> gg:image -> sh -> output\nblur: gaussian5 { sigma: 1.2 }\ngg: colour_grade_inplace { slope: 1.2, offset: 0.01, saturation: 0.7 }\nsh: sharpen { amount: 0.4 }"
    for fmt in (util.F32, util.U8):
        x = util.synthetic(150, 97, fmt, seed=11)
        util.assert_same(util.run_hip(ctx, same, x), util.run_oracle(derived, x), "explicit weights equal to the derived ones")
        for text in (same, odd, inplace):
            want = util.run_oracle(text, x)
            for flags in (0, NF):
                util.assert_same(util.run_hip(ctx, text, x, flags=flags), want, "flags %d\n%s" % (flags, text))


JIT_CHAINS = [
    "input -> aa -> bb -> cc -> dd -> ee -> output\naa: gaussian { sigma: 1.5, radius: 3 }\nbb: passthrough {}\ncc: sharpen { amount: 0.4 }\n"
    "dd: colour_grade { slope: 1.1, offset: 0.01, saturation: 1.3 }\nee: gaussian5 { sigma: 0.8 }",
    "input -> aa -> bb -> cc -> dd -> output\naa: sharpen { amount: 0.9 }\nbb: gaussian9 { sigma: 1.7 }\ncc: sharpen { amount: 0.2 }\ndd: gaussian { sigma: 0.7, radius: 1 }",
    "input -> aa -> bb:image -> cc -> dd -> ee -> ff -> output\naa: gaussian5 { sigma: 1.1 }\nbb: colour_grade { slope: 0.9, offset: 0.02, saturation: 0.5 }\n"
    "cc: passthrough {}\ndd: passthrough {}\nee: gaussian { sigma: 2.0, radius: 5 }\nff: colour_grade { slope: 1.0, offset: 0.0, saturation: 1.6 }",
    "input -> aa -> bb -> cc -> output\naa: passthrough {}\nbb: passthrough {}\ncc: passthrough {}",
    "input -> aa -> bb -> cc -> output\naa: gaussian { sigma: 2.5, radius: 7 }\nbb: colour_grade { slope: 1.2, offset: -0.05, saturation: 1.1 }\ncc: gaussian { sigma: 0.0, radius: 0 }",
]


@pytest.mark.parametrize("k", range(len(JIT_CHAINS)))
def test_chains_compiled_at_graph_creation(ctx, k):
    """Chains the ahead-of-time catalogue lacks: ONE launch whose kernel rf_graph_create compiled (rf_jit.cpp) from the
    library's own device source.  Bit-identical to the oracle and to catalogue-only / unfused execution, both
    formats, ragged sizes, several chunk heights, two texels per lane, a second graph reusing the loaded kernel."""
    text = JIT_CHAINS[k]
    p = rf.Plan(rf.Config(text))
    assert len(p.launches()) == 1 and (p.needs_jit() == [True] or k == 3)      # (all-passthrough = the copy kernel)
    before = rf.lib().rf_jit_compile_count()
    for fmt in (util.F32, util.U8):
        for W, H in ((67, 41), (300, 77), (129, 130)):
            x = util.synthetic(W, H, fmt, seed=W)
            want = util.run_oracle(text, x)
            for kw in (dict(), dict(rows_per_chunk=19), dict(flags=rf.RF_GRAPH_NO_JIT), dict(flags=NF), dict(texels_per_lane=2),
                       dict(exec_flags=rf.RF_EXEC_NO_ALTERNATE), dict(exec_flags=rf.RF_EXEC_FORCE_SPLIT)):
                util.assert_same(util.run_hip(ctx, text, x, **kw), want, "jit chain %d %dx%d fmt=%d %r" % (k, W, H, fmt, kw))
    assert rf.lib().rf_jit_compile_count() - before <= 4        # per format: the kernel and its two-texel variant, compiled once


SSBO = """input -> kw -> cv -> output
kw:ConvWeights -> cv:ConvWeights
kw: conv2d_weights { ksize: 5, sigma: 1.2 }
cv: conv2d { ksize: 5, sigma: 9.0 }"""


def test_conv_weights_through_a_storage_buffer_edge(ctx):
    """`kw:ConvWeights -> cv:ConvWeights`: the conv2d node reads its K x K weights from the storage buffer the conv2d_weights
    node writes (found by the block type name, shader.rs:144-147), not from its own sigma.  Same bits as the oracle and as a
    conv2d that derives those weights itself; rf_graph_set_weights / set_param on the WRITER change what the conv reads."""
    direct = "input -> cv -> output\ncv: conv2d { ksize: 5, sigma: 1.2 }"
    for fmt in (util.F32, util.U8):
        x = util.synthetic(97, 61, fmt, seed=9)
        want = util.run_oracle(SSBO, x)
        util.assert_same(want, util.run_oracle(direct, x), "oracle: wired == direct")
        for flags in (0, NF):
            util.assert_same(util.run_hip(ctx, SSBO, x, flags=flags), want, "buffer edge flags=%d" % flags)
        g = rf.Graph(ctx, rf.Config(SSBO), 97, 61, fmt)
        o = og.GraphOracle(SSBO, 97, 61, fmt)
        g.upload_raw(x); o.upload_raw(x)
        w = np.random.RandomState(3).uniform(-0.05, 0.09, (5, 5)).astype(np.float32)
        g.set_weights("kw", w); o.set_weights("kw", w)
        g.execute(); g.wait(); o.execute()
        util.assert_same(g.download_raw(), o.download_raw(), "weights set on the writer")
        with pytest.raises(rf.RfError):
            g.set_weights("cv", w)                       # a wired conv2d has no weights of its own
        g.set_param("kw", "sigma", 0.7); o.buffers.pop("kw"); o.set_param("kw", "sigma", float(np.float32(0.7)))
        g.execute(); g.wait(); o.execute()
        util.assert_same(g.download_raw(), o.download_raw(), "sigma edited on the writer")
        g.close()


def test_rf_time_drives_the_pulse_node(ctx):
    """update_ubos (render.rs:212-223): every uniform member whose name ends in `_rf_time` receives the frame time.  The
    `pulse` type has one; rf_graph_set_time must change what the NEXT frame computes, fused into a chain or alone, with
    hipGraph replay too, exactly as the oracle's set_time does."""
    text = "input -> pp -> gg -> output\npp: pulse { amount: 0.5, phase_rf_time: 0.25 }\ngg: gaussian5 { sigma: 1.0 }"
    for fmt in (util.F32, util.U8):
        x = util.synthetic(131, 45, fmt, seed=4)
        for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH):
            g = rf.Graph(ctx, rf.Config(text), 131, 45, fmt, flags=flags)
            o = og.GraphOracle(text, 131, 45, fmt)
            g.upload_raw(x); o.upload_raw(x)
            outs = []
            for t in (None, 0.0, 3.25, 7.9990234375, 1e9):
                if t is not None:
                    g.set_time(t); o.set_time(t)
                g.execute(); g.wait(); o.execute()
                util.assert_same(g.download_raw(), o.download_raw(), "pulse at t=%r flags=%d" % (t, flags))
                outs.append(g.download_raw().tobytes())
            assert outs[0] != outs[1] and outs[1] != outs[2]          # the config's own 0.25, then 0 (no pulse), then 3.25
            g.close()
    # the Render mirror: update_ubos is part of every frame (main.rs:134-182)
    info = rf.RenderInfo(width=32, height=16, format=util.F32)      # (through rgba32f the sRGB round trip is the identity)
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as fh:
        fh.write("input -> pp -> output\npp: pulse { amount: 1.0 }")
    info.config_path = fh.name
    r = rf.Render(info, ctx=ctx)
    r.staging[...] = 100
    a = r.render_frame(0.0).copy()
    b = r.render_frame(0.5).copy()
    assert (a[..., :3] == 100).all() and (b[..., :3] > 100).all() and (b[..., 3] == a[..., 3]).all()
    os.unlink(fh.name)


FORKS = [
    util.DIAMOND,
    # an empty branch: the unsharp-mask shape (the join reads the forked image itself)
    "input -> blur -> mx:input_image0\ninput -> mx:input_image1\nmx -> output\nblur: gaussian9 { sigma: 3.0 }\nmx: combination { mix: -0.7 }",
    # the forked image is produced by a node, both branches have several nodes, something follows the join
    "input -> n0 -> aa -> bb -> mx:input_image0\nn0 -> cc -> dd -> mx:input_image1\nmx -> post -> output\nn0: gaussian5 { sigma: 1.0 }\naa: sharpen { amount: 0.5 }\n"
    "bb: colour_grade { slope: 1.1, offset: 0.0, saturation: 1.3 }\ncc: gaussian9 { sigma: 2.0 }\ndd: passthrough {}\npost: sharpen { amount: 0.2 }\nmx: combination { mix: 0.4 }",
    # branches swapped over the inputs; radius-0 and point-only branches
    "input -> gg -> mx:input_image1\ninput -> sh -> mx:input_image0\nmx -> output\ngg: colour_grade { slope: 0.8, offset: 0.05, saturation: 0.4 }\nsh: gaussian { sigma: 1.0, radius: 0 }\nmx: combination { mix: 0.5 }",
]


@pytest.mark.parametrize("k", range(len(FORKS)))
def test_fork_join_in_one_launch(ctx, k):
    """A `combination` whose inputs descend from one image is fused with both branches into ONE launch (a pair of rows
    travels the stage chain; the branch that is not being worked on rides a delay line).  Bit-identical to the oracle's
    node-at-a-time execution: both formats, chunk seams, frame edges inside the halo, both walk directions, the three-part
    split of exchange mode, and as row strips."""
    text = FORKS[k]
    fused = [l for l in rf.Plan(rf.Config(text)).launch_info() if any(l["member_slots"])]
    assert len(fused) == 1, rf.Plan(rf.Config(text)).launches()
    for fmt in (util.F32, util.U8):
        for W, H in ((61, 47), (200, 90), (130, 33)):
            x = util.synthetic(W, H, fmt, seed=W * 3 + k)
            want = util.run_oracle(text, x)
            for kw in (dict(), dict(rows_per_chunk=11), dict(flags=NF), dict(exec_flags=rf.RF_EXEC_NO_ALTERNATE), dict(exec_flags=rf.RF_EXEC_ALTERNATE, rows_per_chunk=8),
                       dict(exec_flags=rf.RF_EXEC_FORCE_SPLIT), dict(flags=rf.RF_GRAPH_HIPGRAPH)):
                util.assert_same(util.run_hip(ctx, text, x, **kw), want, "fork %d %dx%d fmt=%d %r" % (k, W, H, fmt, kw))
    # as over-fetch row strips (every rank's context on GPU 0, no communicator)
    W, H, world = 97, 120, 3
    want = util.run_oracle(text, pixel.fill_synthetic(W, H, util.F32, 77))
    strips = []
    for rank in range(world):
        c = rf.Context(0, rank, world, None)
        g = rf.Graph(c, rf.Config(text), W, H, util.F32, flags=rf.RF_GRAPH_NO_HALO_XCHG)
        g.fill_synthetic(77)
        g.execute(); g.wait()
        strips.append(g.download_raw())
        g.close()
        c.close()
    util.assert_same(np.concatenate(strips, axis=0), want, "fork %d as strips" % k)


@pytest.mark.parametrize("t", [1, 2])
@pytest.mark.parametrize("walk", ["alternate", "top-down"])
def test_texels_per_lane_and_walk_direction(ctx, t, walk):
    """One or two texels per lane (64- / 128-wide strips; the two-texel launch moves its last strip left so
    that every store of a row has an active lane) and both walk policies (odd chunks bottom-up with the
    vertical taps in window form, or every chunk top-down with the taps in scatter form): all four
    combinations must reproduce the oracle bit for bit -- several chunks per strip, ragged widths around
    the 128-column strips, heights that end inside a chunk, frame edges inside the halo."""
    ex = rf.RF_EXEC_ALTERNATE if walk == "alternate" else rf.RF_EXEC_NO_ALTERNATE
    for text in (util.CHAIN5, util.CHAIN3, NODES["gaussian9"], NODES["sharpen"], NODES["gaussian_r7"], util.DIAMOND):
        for W, H in ((256, 37), (257, 70), (371, 45), (640, 121)):
            x = util.synthetic(W, H, util.F32, seed=W + H)
            want = util.run_oracle(text, x)
            for rpc in (0, 16, 29):
                g = rf.Graph(ctx, rf.Config(text), W, H, util.F32, rows_per_chunk=rpc, exec_flags=ex, texels_per_lane=t)
                g.upload_raw(x)
                g.execute(); g.wait()
                got = g.download_raw()
                g.close()
                util.assert_same(got, want, "T=%d %s %dx%d rpc=%d" % (t, walk, W, H, rpc))


def test_interior_boundary_split(ctx):
    """Exchange mode overlaps the halo exchange with the interior rows by issuing a stencil
    launch in three parts (interior, top r rows, bottom r rows).  RF_FORCE_SPLIT=1 takes that
    path on one rank without the exchange: the three parts must tile the frame exactly."""
    for fmt in (util.F32, util.U8):
        x = util.synthetic(211, 157, fmt, seed=5)
        for text in (util.CHAIN3, util.CHAIN5, util.DIAMOND, NODES["gaussian_r7"], NODES["conv9"]):
            want = util.run_oracle(text, x)
            for flags in (0, NF, rf.RF_GRAPH_TIMERS):
                util.assert_same(util.run_hip(ctx, text, x, flags=flags, exec_flags=rf.RF_EXEC_FORCE_SPLIT), want, "split flags=%d" % flags)


def test_a_node_with_two_output_images(ctx):
    """split_luma writes luma_image and chroma_image (one image per output binding, pipeline_graph.rs:205-224): both outputs
    against the oracle, each consumed by its own chain and joined; one output left unwired; every intermediate image too."""
    for fmt in (util.F32, util.U8):
        x = util.synthetic(157, 61, fmt, seed=21)
        for text in (util.SPLIT2, "input -> sp\nsp:chroma_image -> sharpen -> output\nsp: split_luma {}",
                     "input -> sp\nsp:luma_image -> output\nsp: split_luma {}"):
            want = util.run_oracle(text, x)
            for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH):
                for ex in (0, rf.RF_EXEC_FORCE_SPLIT, rf.RF_EXEC_CONCURRENT_LAYERS):
                    util.assert_same(util.run_hip(ctx, text, x, flags=flags, exec_flags=ex), want, "split fmt=%d flags=%d ex=%d" % (fmt, flags, ex))
        # the two images of the split node themselves
        from oracle import graph as og
        o = og.GraphOracle(util.SPLIT2, 157, 61, fmt)
        o.upload_raw(x)
        o.execute()
        g = rf.Graph(ctx, rf.Config(util.SPLIT2), 157, 61, fmt, flags=NF)
        g.upload_raw(x)
        g.execute(); g.wait()
        for res in ("sp:luma_image", "sp:chroma_image"):
            util.assert_same(g.download_image(g.plan.resolve(res)), o.images[o.image_of(res)], res)
        g.close()


def test_exchange_mode_needs_a_communicator():
    c = rf.Context(0, 0, 2, None)
    with pytest.raises(rf.RfError) as e:
        rf.Graph(c, rf.Config(util.CHAIN3), 64, 64, util.F32)         # exchange mode, no unique id
    assert e.value.status == 6 and "communicator" in str(e.value)
    # uploads in over-fetch mode would need the neighbours' rows: refused without a communicator
    g = rf.Graph(c, rf.Config(util.CHAIN3), 64, 64, util.F32, flags=rf.RF_GRAPH_NO_HALO_XCHG)
    with pytest.raises(rf.RfError):
        g.upload_raw(np.zeros((32, 64, 4), np.float32))
    g.close()
    c.close()


# ---- random graphs --------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(40))
def test_random_graphs(ctx, seed):
    """Planner (layering, aliasing, in-place point ops, fusion with greedy splitting) and kernels
    together, on graphs nobody wrote by hand: fused, unfused and hipGraph execution all equal the
    oracle, both formats -- with the layers in plan order (default) and, every other seed, with
    hazard-free layers forked onto side streams (RF_CONCURRENT_LAYERS=1)."""
    rng = np.random.RandomState(1000 + seed)
    text = (util.random_graph if seed < 24 else util.random_dag)(rng)
    W, H = int(rng.randint(1, 200)), int(rng.randint(1, 120))
    ex = rf.RF_EXEC_CONCURRENT_LAYERS if seed & 1 else 0
    for fmt in (util.F32, util.U8):
        x = util.synthetic(W, H, fmt, seed=seed)
        want = util.run_oracle(text, x)
        for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH):
            util.assert_same(util.run_hip(ctx, text, x, flags=flags, exec_flags=ex), want, "seed %d flags %d %dx%d\n%s" % (seed, flags, W, H, text))


@pytest.mark.parametrize("seed", range(24))
def test_random_dags_with_two_output_nodes(ctx, seed):
    """random DAGs in which some nodes write TWO images (split_luma): aliasing of both outputs, their readers in later layers,
    fused and unfused, both formats; every third seed also as 2-3 row strips in over-fetch mode"""
    rng = np.random.RandomState(7000 + seed)
    text = util.random_dag(rng, split=True)
    W, H = int(rng.randint(1, 200)), int(rng.randint(40, 120))
    for fmt in (util.F32, util.U8):
        x = util.synthetic(W, H, fmt, seed=seed)
        want = util.run_oracle(text, x)
        for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH):
            util.assert_same(util.run_hip(ctx, text, x, flags=flags, exec_flags=rf.RF_EXEC_CONCURRENT_LAYERS if seed & 1 else 0), want,
                             "seed %d flags %d %dx%d\n%s" % (seed, flags, W, H, text))
    if seed % 3 == 0:
        world = 2 + seed % 2
        p = rf.Plan(rf.Config(text))
        if p.halo_schedule(False)[3] <= H // world:
            want = util.run_oracle(text, pixel.fill_synthetic(W, H, util.F32, 0x5EED0004))
            strips = []
            for rank in range(world):
                c = rf.Context(0, rank, world, None)
                g = rf.Graph(c, rf.Config(text), W, H, util.F32, flags=rf.RF_GRAPH_NO_HALO_XCHG)
                g.fill_synthetic(0x5EED0004)
                g.execute(); g.wait()
                strips.append(g.download_raw())
                g.close()
                c.close()
            util.assert_same(np.concatenate(strips, axis=0), want, "strips seed %d world %d\n%s" % (seed, world, text))


@pytest.mark.parametrize("seed", range(10))
def test_random_graphs_as_row_strips(seed, monkeypatch):
    """Random graphs again, cut into 2..4 row strips in over-fetch mode (every rank's context on
    GPU 0, no communicator) and, on one rank, through the three-part interior/boundary launch of
    exchange mode (RF_FORCE_SPLIT): the cumulative-halo schedule of arbitrary graphs."""
    rng = np.random.RandomState(2000 + seed)
    text = util.random_graph(rng)
    W, H, world = int(rng.randint(8, 160)), int(rng.randint(70, 140)), int(rng.randint(2, 5))
    flags = (0, NF)[seed & 1]
    ghost = rf.Plan(rf.Config(text), flags).halo_schedule(False)[3]
    if ghost > H // world:
        pytest.skip("strips shorter than the cumulative halo")
    for fmt in (util.F32, util.U8):
        want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, 0x5EED0000 + seed))
        strips = []
        for rank in range(world):
            c = rf.Context(0, rank, world, None)
            g = rf.Graph(c, rf.Config(text), W, H, fmt, flags=flags | rf.RF_GRAPH_NO_HALO_XCHG)
            g.fill_synthetic(0x5EED0000 + seed)
            g.execute(); g.wait()
            strips.append(g.download_raw())
            g.close()
            c.close()
        util.assert_same(np.concatenate(strips, axis=0), want, "strips seed=%d world=%d flags=%d\n%s" % (seed, world, flags, text))
    monkeypatch.setenv("RF_FORCE_SPLIT", "1")          # the environment form of RF_EXEC_FORCE_SPLIT (it overrides the options)
    c = rf.Context(0)
    x = util.synthetic(W, H, util.F32, seed=seed)
    util.assert_same(util.run_hip(c, text, x, flags=flags), util.run_oracle(text, x), "forced split seed=%d\n%s" % (seed, text))
    c.close()


def test_second_frame_of_an_in_place_graph(ctx):
    """A point op written in place on rf:file-input grades the input again on every frame (the
    reference uploads once per slot, main.rs:164-170, and aliases the node onto the input image):
    frame 2 equals the oracle executed twice, fused or not, on both slots in flight."""
    text = "input -> aa:image -> bb -> cc -> output\naa: colour_grade { slope: 0.8, offset: 0.05, saturation: 1.3 }\nbb: gaussian5 { sigma: 1.2 }\ncc: sharpen { amount: 0.4 }"
    from oracle import graph as ograph
    for fmt in (util.F32, util.U8):
        x = util.synthetic(300, 170, fmt, seed=9)
        o = ograph.GraphOracle(text, 300, 170, fmt)
        o.upload_raw(x); o.execute()
        first = o.download_raw().copy()
        o.execute()
        second = o.download_raw()
        assert first.tobytes() != second.tobytes()
        for flags in (0, NF):
            g = rf.Graph(ctx, rf.Config(text), 300, 170, fmt, num_frames=2, flags=flags)
            g.upload_raw(x)
            g.execute(0); g.execute(1); g.wait(0); g.wait(1)
            util.assert_same(g.download_raw(0), first, "frame 1 slot 0")
            g.execute(0); g.execute(1); g.wait(0); g.wait(1)
            util.assert_same(g.download_raw(0), second, "frame 2 slot 0")
            util.assert_same(g.download_raw(1), second, "frame 2 slot 1")
            g.close()


def test_conv_sigma_edit_regenerates_the_weights(ctx):
    """rf_graph_set_param on a conv2d node's sigma: the default weights are derived from sigma, so the
    next frame equals a graph created with the new value (found by scripts/fuzz_params.py)."""
    t0 = "input -> cc -> gg -> output\ncc: conv2d { ksize: 5, sigma: 1.70 }\ngg: colour_grade { slope: 1.10, offset: 0.00, saturation: 1.00 }"
    t1 = t0.replace("sigma: 1.70", "sigma: 0.60").replace("slope: 1.10", "slope: 0.70")
    for fmt in (util.F32, util.U8):
        x = util.synthetic(97, 61, fmt, seed=4)
        for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH):
            g = rf.Graph(ctx, rf.Config(t0), 97, 61, fmt, flags=flags)
            g.upload_raw(x)
            g.execute(); g.wait()
            util.assert_same(g.download_raw(), util.run_oracle(t0, x), "before")
            g.set_param("cc", "sigma", float(np.float32(0.60)))
            g.set_param("gg", "slope", float(np.float32(0.70)))
            g.execute(); g.wait()
            util.assert_same(g.download_raw(), util.run_oracle(t1, x), "after the edits, flags %d" % flags)
            g.close()


@pytest.fixture
def user_types():
    old = util.register_user_types()
    yield
    rf.set_shader_path(old)


@pytest.mark.parametrize("seed", range(30))
def test_random_dags_with_user_types(ctx, user_types, seed):
    """random DAGs in which about a third of the nodes are USER types (shaders/*.stage.hip): invert / edge_detect fused with
    whatever surrounds them, unsharp_mask with two inputs and up to two outputs read, tone_curve -> apply_curve over a
    storage-buffer edge; split_luma nodes on even seeds.  The oracle runs the same files compiled for the host
    (oracle/user_stage.py): this checks the EXECUTOR around user types (fusion, aliasing, layering, strips), the files'
    arithmetic is checked against exact restatements in test_gpu_user_stage.py / test_gpu_user_node.py."""
    rng = np.random.RandomState(9000 + seed)
    text = util.random_dag(rng, split=seed % 2 == 0, user=True)
    W, H = int(rng.randint(1, 200)), int(rng.randint(40, 120))
    for fmt in (util.F32, util.U8):
        x = util.synthetic(W, H, fmt, seed=seed)
        want = util.run_oracle(text, x)
        for flags in (0, NF, rf.RF_GRAPH_HIPGRAPH, rf.RF_GRAPH_NO_JIT):
            util.assert_same(util.run_hip(ctx, text, x, flags=flags, exec_flags=rf.RF_EXEC_CONCURRENT_LAYERS if seed & 1 else 0), want,
                             "seed %d flags %d %dx%d\n%s" % (seed, flags, W, H, text))
    if seed % 3 == 0:
        world = 2 + seed % 2
        p = rf.Plan(rf.Config(text))
        if p.halo_schedule(False)[3] <= H // world:
            want = util.run_oracle(text, pixel.fill_synthetic(W, H, util.F32, 0x5EED0004))
            strips = []
            for rank in range(world):
                c = rf.Context(0, rank, world, None)
                g = rf.Graph(c, rf.Config(text), W, H, util.F32, flags=rf.RF_GRAPH_NO_HALO_XCHG)
                g.fill_synthetic(0x5EED0004)
                g.execute(); g.wait()
                strips.append(g.download_raw())
                g.close()
                c.close()
            util.assert_same(np.concatenate(strips, axis=0), want, "strips seed %d world %d\n%s" % (seed, world, text))
