"""Shared helpers of the parity tests: run the same config text through the CPU oracle
and through librfhip.so (C ABI), return the raw output texels."""
import os

import numpy as np

import reforge_amd as rf
from oracle import graph as ograph
from oracle import pixel

F32 = rf.RF_FORMAT_RGBA32F
U8 = rf.RF_FORMAT_RGBA8

CHAIN3 = """
input -> blur -> grade -> sharp -> output
blur:  gaussian5    { sigma: 1.0 }
grade: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp: sharpen      { amount: 0.5 }
"""

# the 5-stage chain of BASELINE config 4
CHAIN5 = """
input -> blur -> grade -> sharp -> wide -> finish -> output
blur:   gaussian5    { sigma: 1.0 }
grade:  colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp:  sharpen      { amount: 0.5 }
wide:   gaussian9    { sigma: 2.0 }
finish: colour_grade { slope: 0.95, offset: 0.01, saturation: 0.9 }
"""

# same shape with a gaussian5 in the middle: fuses as 3 + 2 launches (two fused launches per frame)
CHAIN5_SPLIT = CHAIN5.replace("gaussian9    { sigma: 2.0 }", "gaussian5    { sigma: 2.0 }")

DIAMOND = """
input -> blur -> mixer:input_image0
input -> sharp -> mixer:input_image1
mixer -> output
blur:  gaussian5   { sigma: 1.5 }
sharp: sharpen     { amount: 0.75 }
mixer: combination { mix: 0.25 }
"""


def run_oracle(text, img, weights=None):
    H, W, _ = img.shape
    g = ograph.GraphOracle(text, W, H, pixel.fmt_of(img))
    for node, w in (weights or {}).items():
        g.set_weights(node, w)
    g.upload_raw(img)
    g.execute()
    return g.download_raw()


def run_hip(ctx, text, img, flags=0, weights=None, rows_per_chunk=None, num_frames=1, slot=0):
    H, W, _ = img.shape
    old = os.environ.get("RF_ROWS_PER_CHUNK")
    if rows_per_chunk is not None:
        os.environ["RF_ROWS_PER_CHUNK"] = str(rows_per_chunk)
    try:
        g = rf.Graph(ctx, rf.Config(text), W, H, pixel.fmt_of(img), num_frames=num_frames, flags=flags)
    finally:
        if rows_per_chunk is not None:
            if old is None:
                del os.environ["RF_ROWS_PER_CHUNK"]
            else:
                os.environ["RF_ROWS_PER_CHUNK"] = old
    try:
        for node, w in (weights or {}).items():
            g.set_weights(node, w)
        g.upload_raw(img)
        g.execute(slot)
        g.wait(slot)
        return g.download_raw(slot)
    finally:
        g.close()


def assert_same(got, want, what=""):
    """Bit-exact for both formats (the bar is pixel-exact rgba8 / <= 1 ulp rgba32f; the
    kernels are built to be bit-identical, so the tests hold them to 0 ulp)."""
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    if got.tobytes() == want.tobytes():
        return
    if got.dtype == np.float32:
        a, b = got.view(np.uint32).astype(np.int64), want.view(np.uint32).astype(np.int64)
    else:
        a, b = got.astype(np.int64), want.astype(np.int64)
    bad = np.argwhere(a != b)
    y, x, c = bad[0]
    raise AssertionError("%s: %d of %d values differ; first at (y=%d,x=%d,c=%d): got %r want %r; max |diff| = %d (ulp or codes)" % (
        what, len(bad), a.size, y, x, c, got[y, x, c], want[y, x, c], np.abs(a - b).max()))


def synthetic(W, H, fmt, seed=0x5EED0002):
    return pixel.fill_synthetic(W, H, fmt, seed)
