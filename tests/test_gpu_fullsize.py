"""GPU (MI355X): BASELINE.json's full sizes.  The oracle cannot run whole 4K/8K frames
in seconds, so full frames are checked through size-independent properties:
  * passthrough == identity (checksum of the whole frame against the generator);
  * fused execution == one-launch-per-node execution, bit for bit, whole frame;
  * the output does not depend on how rows are chunked across waves;
  * BAND CHECK: for bands of rows at the top edge, the bottom edge and the interior, the
    oracle is run on the band plus its halo (full width) and must equal the GPU rows
    bit for bit -- any row of the frame could be chosen, so this samples the full-size
    result against the oracle itself."""
import os

import numpy as np
import pytest

import reforge_amd as rf
from oracle import pixel
from tests import util

pytestmark = pytest.mark.gpu
NF = rf.RF_GRAPH_NO_FUSION

GAUSS9 = "input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }"
CONV31 = "input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }"


def gpu_frame(ctx, text, W, H, fmt, seed, flags=0, rpc=None, conv_path=0):
    g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags, rows_per_chunk=rpc or 0, conv_path=conv_path)
    g.fill_synthetic(seed)
    g.execute()
    g.wait()
    out = g.download_raw()
    g.close()
    return out


def band_check(out, text, W, H, fmt, seed, radius, bands):
    for b0, b1 in bands:
        lo, hi = max(0, b0 - radius), min(H, b1 + radius)
        src = pixel.fill_synthetic(W, hi - lo, fmt, seed, y0=lo)       # rows lo..hi of the frame
        want = util.run_oracle(text, src)[b0 - lo:b1 - lo]
        util.assert_same(np.ascontiguousarray(out[b0:b1]), np.ascontiguousarray(want), "rows %d..%d" % (b0, b1))


def test_config1_passthrough_512_rgba8(ctx):
    """BASELINE config 1: passthrough, 512x512 rgba8."""
    out = gpu_frame(ctx, "input -> passthrough -> output", 512, 512, util.U8, 0x5EED0001)
    assert out.tobytes() == pixel.fill_synthetic(512, 512, util.U8, 0x5EED0001).tobytes()


def test_config2_chain3_4k_rgba32f(ctx):
    """BASELINE config 2 (the headline metric): gaussian5 -> colour-grade -> sharpen, 3840x2160 rgba32f."""
    W, H, seed = 3840, 2160, 0x5EED0002
    fused = gpu_frame(ctx, util.CHAIN3, W, H, util.F32, seed)
    unfused = gpu_frame(ctx, util.CHAIN3, W, H, util.F32, seed, flags=NF)
    assert fused.tobytes() == unfused.tobytes()
    assert gpu_frame(ctx, util.CHAIN3, W, H, util.F32, seed, rpc=61).tobytes() == fused.tobytes()
    band_check(fused, util.CHAIN3, W, H, util.F32, seed, 3, [(0, 24), (1068, 1092), (2136, 2160)])


def test_config2_chain3_4k_rgba8(ctx):
    W, H, seed = 3840, 2160, 0x5EED0002
    fused = gpu_frame(ctx, util.CHAIN3, W, H, util.U8, seed)
    assert fused.tobytes() == gpu_frame(ctx, util.CHAIN3, W, H, util.U8, seed, flags=NF).tobytes()
    band_check(fused, util.CHAIN3, W, H, util.U8, seed, 3, [(0, 16), (2144, 2160)])


def test_passthrough_4k_identity(ctx):
    W, H = 3840, 2160
    out = gpu_frame(ctx, "input -> passthrough -> output", W, H, util.F32, 7)
    assert out.tobytes() == pixel.fill_synthetic(W, H, util.F32, 7).tobytes()


def test_config3_gaussian9_8k(ctx):
    """BASELINE config 3: 9x9 separable gaussian, 7680x4320 rgba32f."""
    W, H, seed = 7680, 4320, 0x5EED0003
    out = gpu_frame(ctx, GAUSS9, W, H, util.F32, seed)
    assert out.tobytes() == gpu_frame(ctx, GAUSS9, W, H, util.F32, seed, rpc=97).tobytes()
    band_check(out, GAUSS9, W, H, util.F32, seed, 4, [(0, 12), (2000, 2012), (4308, 4320)])


def test_config4_chain5_tall_strip(ctx):
    """BASELINE config 4 is 16384^2 over 8 GPUs; one rank's share is a 16384 x 2048 strip.
    Run that strip-sized frame on one GPU, fused vs unfused, plus band checks."""
    W, H, seed = 16384, 2048, 0x5EED0004
    fused = gpu_frame(ctx, util.CHAIN5, W, H, util.F32, seed)
    assert fused.tobytes() == gpu_frame(ctx, util.CHAIN5, W, H, util.F32, seed, flags=NF).tobytes()
    band_check(fused, util.CHAIN5, W, H, util.F32, seed, 7, [(0, 8), (1020, 1028), (2040, 2048)])


def test_config5_conv31_8k_band(ctx):
    """BASELINE config 5: 31x31 dense convolution on 7680x4320 rgba32f."""
    W, H, seed = 7680, 4320, 0x5EED0005
    out = gpu_frame(ctx, CONV31, W, H, util.F32, seed)
    band_check(out, CONV31, W, H, util.F32, seed, 15, [(0, 2), (2159, 2161), (4318, 4320)])
    # every large-K kernel at the full size: the banded MFMA contraction north_star names and the
    # register-blocked VALU kernel must produce the same frame
    for path in (rf.RF_CONV_MFMA, rf.RF_CONV_VALU):
        assert gpu_frame(ctx, CONV31, W, H, util.F32, seed, conv_path=path).tobytes() == out.tobytes(), "conv path %d at 8K" % path
    # linearity in the input survives at full size: conv(x) of a constant frame is that
    # constant times the kernel sum, identical at every pixel away from nothing (clamp-to-edge
    # keeps a constant frame constant everywhere)
    g = rf.Graph(ctx, rf.Config(CONV31), 512, 512, util.F32)
    g.upload_raw(np.full((512, 512, 4), 0.5, np.float32))
    g.execute(); g.wait()
    c = g.download_raw()
    g.close()
    assert (c == c[0, 0, 0]).all() and abs(float(c[0, 0, 0]) - 0.5) < 1e-5


@pytest.mark.parametrize("seed", range(16))
def test_random_graphs_1080p_whole_frame(ctx, seed):
    """Generated graphs at 1920x1080, the whole frame against the oracle, both formats: enough
    strips and chunks for every seam, both walk directions and multi-round launches -- and enough
    waves in flight for timing-dependent faults to show (this test found the in-place/fork race of
    a layer, the fused in-place head and a ds_read that an LDS-DMA refill could overtake)."""
    text = (util.random_graph if seed < 8 else util.random_dag)(np.random.RandomState(7000 + seed))
    ex = rf.RF_EXEC_CONCURRENT_LAYERS if seed & 1 else 0      # hazard-free layers forked onto side streams
    pixel.set_threads(min(16, os.cpu_count() or 1))
    try:
        for fmt in (util.F32, util.U8):
            x = pixel.fill_synthetic(1920, 1080, fmt, 0x5EED0000 + seed)
            want = util.run_oracle(text, x)
            util.assert_same(util.run_hip(ctx, text, x, exec_flags=ex), want, "1080p seed %d fused\n%s" % (seed, text))
            util.assert_same(util.run_hip(ctx, text, x, flags=NF, exec_flags=ex), want, "1080p seed %d unfused\n%s" % (seed, text))
    finally:
        pixel.set_threads(1)


STORE_HAZARD_GRAPHS = [
    # scripts/fuzz_graphs.py seeds 7054 / 7063 / 7113: fork/join pipelines compiled at graph creation whose schedule put a
    # VALU write to the store's data registers one wait state behind the (inline-asm) global_store_dwordx4 -- gfx940+ needs
    # two for a store of more than 8 bytes, and the hazard recogniser does not look into asm (PxF32::store_row)
    "input -> n00 -> n01 -> mx:input_image0\nn00 -> n02 -> mx:input_image1\nmx -> output\nn00: colour_grade { slope: 1.20, offset: 0.065, saturation: 0.46 }\n"
    "n01: colour_grade { slope: 1.17, offset: 0.086, saturation: 0.22 }\nn02: sharpen { amount: 1.35 }\nmx: combination { mix: 0.60 }",
    "input -> n00 -> n01 -> mx:input_image0\nn00 -> n02 -> mx:input_image1\nmx -> output\nn00: colour_grade { slope: 0.63, offset: -0.030, saturation: 0.93 }\n"
    "n01: gaussian5 { sigma: 2.74 }\nn02: gaussian5 { sigma: 1.48 }\nmx: combination { mix: 0.99 }",
    "input -> n00 -> n01 -> mx:input_image0\nn00 -> n02 -> mx:input_image1\nmx -> output\nn00: conv2d { ksize: 3, sigma: 1.51 }\n"
    "n01: colour_grade { slope: 0.98, offset: -0.080, saturation: 1.29 }\nn02: gaussian { sigma: 1.73, radius: 3 }\nmx: combination { mix: 0.31 }",
]


@pytest.mark.parametrize("k", range(len(STORE_HAZARD_GRAPHS)))
def test_store_data_hazard_graphs_1080p(ctx, k):
    """The graphs that exposed the store-data hazard of the asm store (a few thousand wrong texels per 1080p frame, different
    ones each run), whole frame against the oracle, three runs each."""
    text = STORE_HAZARD_GRAPHS[k]
    pixel.set_threads(min(16, os.cpu_count() or 1))
    try:
        x = pixel.fill_synthetic(1920, 1080, util.F32, 7054 + k)
        want = util.run_oracle(text, x)
        for rep in range(3):
            util.assert_same(util.run_hip(ctx, text, x), want, "store hazard graph %d run %d" % (k, rep))
        util.assert_same(util.run_hip(ctx, text, x, flags=rf.RF_GRAPH_HIPGRAPH), want, "store hazard graph %d replayed" % k)
    finally:
        pixel.set_threads(1)


def test_config4_chain5_whole_16k_frame_one_gpu(ctx):
    """BASELINE config 4's whole 16384 x 16384 rgba32f frame on ONE GPU (what bench.py --workload
    chain5_16k times at N=1): 4 GiB per image, byte offsets beyond 2^31 and 2^32.  Fused (one
    launch) == one launch per node over the whole frame, and bands at the top, across the 2 GiB
    and 4 GiB-offset rows and at the bottom equal the oracle."""
    W, H, seed = 16384, 16384, 0x5EED0004
    fused = gpu_frame(ctx, util.CHAIN5, W, H, util.F32, seed)
    band_check(fused, util.CHAIN5, W, H, util.F32, seed, 7, [(0, 8), (8188, 8196), (16376, 16384)])
    unfused = gpu_frame(ctx, util.CHAIN5, W, H, util.F32, seed, flags=NF)
    assert np.array_equal(fused.view(np.uint32), unfused.view(np.uint32))
