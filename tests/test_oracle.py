"""CPU: the oracle against independent known answers and the committed golden vectors.
(The reference holds no golden vectors of its own -- SURVEY.md section 4.)"""
import os

import numpy as np
import pytest

from oracle import graph as ograph
from oracle import pixel
from tests import kat, util

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))


def run(text, img, weights=None):
    return util.run_oracle(text, img, weights)


def test_hash32_matches_independent_python():
    for (s, i, c), want in zip(GOLDEN["hash_args"], GOLDEN["hash_vals"]):
        assert pixel.hash32(int(s), int(i), int(c)) == int(want)


def test_synthetic_fill_layout():
    x = pixel.fill_synthetic(5, 3, util.F32, 0x5EED0002)
    u = pixel.hash32(0x5EED0002, 1 * 5 + 2, 3)
    assert x[1, 2, 3] == np.float32((u >> 8) * 2.0 ** -24) and 0.0 <= x.min() and x.max() < 1.0
    y = pixel.fill_synthetic(5, 3, util.U8, 0x5EED0002)
    assert y[1, 2, 3] == u >> 24
    # a strip generated with a row offset equals the rows of the full frame
    strip = pixel.fill_synthetic(5, 2, util.F32, 0x5EED0002, y0=1)
    assert strip.tobytes() == x[1:3].tobytes()


def test_srgb_tables_match_independent_float64():
    eotf, thr = pixel.srgb_tables()
    assert eotf.tobytes() == GOLDEN["srgb_eotf"].tobytes()
    assert thr.tobytes() == GOLDEN["srgb_thr"].tobytes()


def test_srgb_roundtrip_identity_rgba32f():
    """enc(dec(c)) == c for all 256 codes through an rgba32f graph."""
    c = np.zeros((1, 256, 4), np.uint8)
    c[0, :, :] = np.arange(256)[:, None]
    assert (pixel.download_srgb8(pixel.upload_srgb8(c, util.F32)) == c).all()


def test_srgb_roundtrip_lut_rgba8():
    """through an rgba8 graph the round trip is a fixed lossy LUT."""
    c = np.zeros((1, 256, 4), np.uint8)
    c[0, :, :] = np.arange(256)[:, None]
    rt = pixel.download_srgb8(pixel.upload_srgb8(c, util.U8))
    assert (rt[0, :, 0] == GOLDEN["srgb_rgba8_roundtrip"]).all()
    assert (rt[0, :, 3] == np.arange(256)).all()           # alpha is linear: exact


def test_gaussian_weights_match_independent_float64():
    assert pixel.gaussian_weights(1.0, 2).tobytes() == GOLDEN["gauss_w_s1_r2"].tobytes()
    assert pixel.gaussian_weights(2.0, 4).tobytes() == GOLDEN["gauss_w_s2_r4"].tobytes()
    assert pixel.gaussian_weights(5.0, 15).tobytes() == GOLDEN["gauss_w_s5_r15"].tobytes()
    assert pixel.gaussian_weights(0.0, 4).tolist() == [1, 0, 0, 0, 0]


def test_unorm8_conversions_exact():
    c = np.arange(256, dtype=np.uint8).reshape(1, 64, 4)
    assert (pixel.passthrough(c) == c).all()


@pytest.mark.parametrize("fmt", [util.F32, util.U8])
@pytest.mark.parametrize("W,H", kat.PASSTHROUGH_SIZES)
def test_passthrough_identity(fmt, W, H):
    kat.check_passthrough_identity(run, fmt, W, H)


def test_passthrough_special_floats():
    kat.check_passthrough_preserves_special_floats(run)


def test_gaussian_impulse():
    kat.check_gaussian_impulse(run, GOLDEN)
    kat.check_gaussian9_weights(run, GOLDEN)


@pytest.mark.parametrize("fmt", [util.F32, util.U8])
def test_degenerate_parameters_are_identity(fmt):
    kat.check_gaussian_delta_is_identity(run, fmt)
    kat.check_sharpen_zero_is_identity(run, fmt)


def test_sharpen_impulse():
    kat.check_sharpen_impulse(run)


def test_conv_impulse():
    kat.check_conv_impulse_is_flipped_kernel(run)


def test_grade_properties():
    kat.check_grade_saturation_zero_is_grey(run)
    kat.check_unorm8_store_rounds_to_even(run)


def test_separable_gaussian_equals_dense_conv_on_interior_within_rounding():
    """The dense outer-product kernel and the separable passes agree to a few ulp
    (different association), a sanity check on both restatements."""
    x = pixel.fill_synthetic(48, 40, util.F32, 3)
    a = run("input -> gaussian5 -> output\ngaussian5: gaussian5 { sigma: 1.0 }", x)
    b = run("input -> conv2d -> output\nconv2d: conv2d { ksize: 5, sigma: 1.0 }", x)
    np.testing.assert_allclose(a, b, rtol=0, atol=4e-7)


@pytest.mark.parametrize("tag,fmt", [("f32", util.F32), ("u8", util.U8)])
def test_golden_part_b_from_the_exact_evaluator(tag, fmt):
    """golden.npz part B is produced by tests/golden/exact_eval.py (exact rationals, one rounding per operation, no code
    shared with oracle/): the oracle is checked against it here, the kernels in test_gpu_parity.py::test_golden_vectors."""
    x = GOLDEN["in_" + tag]
    assert x.tobytes() == pixel.fill_synthetic(40, 24, fmt, 0x5EED0002).tobytes()
    util.assert_same(run(util.CHAIN3, x), GOLDEN["chain3_" + tag], "chain3")
    util.assert_same(run(util.CHAIN5, x), GOLDEN["chain5_" + tag], "chain5")
    util.assert_same(run(util.DIAMOND, x), GOLDEN["diamond_" + tag], "diamond")
    util.assert_same(run("input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }", x), GOLDEN["gauss9_" + tag], "gauss9")
    util.assert_same(run("input -> conv2d -> output\nconv2d: conv2d { ksize: 7, sigma: 1.5 }", x), GOLDEN["conv7_" + tag], "conv7")
    from tests.golden import exact_eval as ex
    for name, (_fn, text) in ex.MORE_GRAPHS.items():
        util.assert_same(run(text or util.SPLIT2, x), GOLDEN["%s_%s" % (name, tag)], name)


def test_aliasing_plan_does_not_change_results():
    """pipeline_graph.rs:358-427: results must not depend on image reuse.  Execute the
    5-chain with and without the remapping."""
    x = pixel.fill_synthetic(33, 21, util.F32, 5)
    g = ograph.GraphOracle(util.CHAIN5, 33, 21, util.F32)
    assert g.reuse                                   # the chain does alias
    g.upload_raw(x)
    g.execute()
    aliased = g.download_raw()
    g2 = ograph.GraphOracle(util.CHAIN5, 33, 21, util.F32)
    g2.reuse = {}
    g2.images = {}
    for layer in g2.layers:
        for n in layer:
            for r, _ in g2.infos[n].input_images + g2.infos[n].output_images:
                g2.images.setdefault(r, pixel.new_image(33, 21, util.F32))
    g2.upload_raw(x)
    g2.execute()
    assert aliased.tobytes() == g2.download_raw().tobytes()


def test_in_place_point_op():
    """`-> colour_grade:image ->` uses one binding for input and output: in place
    (pipeline_graph.rs:400-411), same result as the two-image form."""
    x = pixel.fill_synthetic(30, 9, util.U8, 9)
    inst = "\ngg: colour_grade { slope: 0.9, offset: 0.05, saturation: 1.3 }\nbb: gaussian5 { sigma: 1.0 }\nss: sharpen { amount: 0.3 }"
    a = run("input -> bb -> gg -> ss -> output" + inst, x)
    b = run("input -> bb -> gg:image -> ss -> output" + inst, x)
    assert a.tobytes() == b.tobytes()
