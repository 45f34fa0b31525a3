"""Generates tests/golden/golden.npz.

The reference (calkhaz/reforge) holds no tests, fixtures or golden vectors and cannot
be built or run here (SURVEY.md sections 4 and 8c), so there is nothing of the
reference's to import.  The vectors are of two kinds:

  (A) KNOWN ANSWERS derived here INDEPENDENTLY of oracle/ (pure Python / numpy float64
      restatements of the published formulas): the hash, the sRGB tables, the rgba8
      round-trip LUT, gaussian weights.  They pin the oracle.
  (B) REGRESSION PINS produced by the oracle itself on small seeded frames: they pin
      nothing about the reference (parity is unpinned for the authored nodes), they
      only freeze the authored specification so later rounds cannot drift silently.

Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pixel  # noqa: E402
from tests import util  # noqa: E402


def hash32_py(seed, idx, c):
    m = 0xFFFFFFFF
    h = (seed ^ (((idx * 4 + c) & m) * 0x9E3779B1)) & m
    h ^= h >> 16
    h = (h * 0x7FEB352D) & m
    h ^= h >> 15
    h = (h * 0x846CA68B) & m
    h ^= h >> 16
    return h


def eotf(cs):
    return cs / 12.92 if cs <= 0.04045 else math.pow((cs + 0.055) / 1.055, 2.4)


def oetf(v):
    return 12.92 * v if v <= 0.0031308 else 1.055 * math.pow(v, 1.0 / 2.4) - 0.055


def main():
    out = {}
    # ---- (A) independent known answers -------------------------------------------
    out["hash_args"] = np.array([(0x5EED0001, 0, 0), (0x5EED0001, 1, 3), (0x5EED0002, 8294399, 2),
                                 (0x5EED0004, 268435455, 3), (0, 0, 0), (0xFFFFFFFF, 12345, 1)], dtype=np.uint64)
    out["hash_vals"] = np.array([hash32_py(int(s), int(i), int(c)) for s, i, c in out["hash_args"]], dtype=np.uint32)
    out["srgb_eotf"] = np.array([eotf(c / 255.0) for c in range(256)], dtype=np.float64).astype(np.float32)
    out["srgb_thr"] = np.array([eotf((q + 0.5) / 255.0) for q in range(255)], dtype=np.float64).astype(np.float32)
    # rgba8 graph: code -> decode -> quantise to UNORM8 -> linear/255 -> encode;
    # encode by the exact OETF in float64, round half to even
    lut = []
    for c in range(256):
        q = int(np.rint(np.float32(np.float32(eotf(c / 255.0)) * np.float32(255.0))))
        v = float(np.float32(q) / np.float32(255.0))
        lut.append(int(np.rint(oetf(v) * 255.0)))
    out["srgb_rgba8_roundtrip"] = np.array(lut, dtype=np.uint8)
    for name, sigma, r in (("gauss_w_s1_r2", 1.0, 2), ("gauss_w_s2_r4", 2.0, 4), ("gauss_w_s5_r15", 5.0, 15)):
        e = [math.exp(-(i * i) / (2.0 * sigma * sigma)) for i in range(r + 1)]
        tot = e[0] + sum(2.0 * v for v in e[1:])
        out[name] = np.array([v / tot for v in e], dtype=np.float64).astype(np.float32)

    # ---- (B) regression pins of the authored specification -------------------------
    W, H = 40, 24
    for tag, fmt in (("f32", util.F32), ("u8", util.U8)):
        x = pixel.fill_synthetic(W, H, fmt, 0x5EED0002)
        out["in_" + tag] = x
        out["chain3_" + tag] = util.run_oracle(util.CHAIN3, x)
        out["chain5_" + tag] = util.run_oracle(util.CHAIN5, x)
        out["diamond_" + tag] = util.run_oracle(util.DIAMOND, x)
        out["gauss9_" + tag] = util.run_oracle("input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }", x)
        out["conv7_" + tag] = util.run_oracle("input -> conv2d -> output\nconv2d: conv2d { ksize: 7, sigma: 1.5 }", x)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz"), **out)
    print("wrote golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
