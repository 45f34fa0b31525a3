"""Generates tests/golden/golden.npz.

The reference (calkhaz/reforge) holds no tests, fixtures or golden vectors and cannot
be built or run here (SURVEY.md sections 4 and 8c), so there is nothing of the
reference's to import.  The vectors are of two kinds:

  (A) KNOWN ANSWERS derived here INDEPENDENTLY of oracle/ (pure Python / numpy float64
      restatements of the published formulas): the hash, the sRGB tables, the rgba8
      round-trip LUT, gaussian weights.  They pin the oracle.
  (B) the authored node SPECIFICATION evaluated on small seeded frames by
      tests/golden/exact_eval.py: exact rational arithmetic, one round-to-nearest-even to
      binary32 per operation, pure Python, no code shared with oracle/ or the kernels.  The
      oracle AND the HIP kernels are then both checked against it (tests/test_oracle.py,
      tests/test_gpu_parity.py::test_golden_vectors).  It pins nothing about the reference
      (parity is unpinned for the authored nodes); it removes the risk of the oracle
      grading itself.  This script imports nothing from oracle/.

Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.golden import exact_eval as ex  # noqa: E402


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

    # ---- (B) the specification, evaluated exactly (NOT by the oracle) ---------------
    W, H = 40, 24
    for tag, dt in (("f32", np.float32), ("u8", np.uint8)):
        x = ex.synthetic(W, H, tag, 0x5EED0002)
        out["in_" + tag] = np.frombuffer(ex.to_bytes(x, tag), dtype=dt).reshape(H, W, 4).copy()
        for name, fn in list(ex.GRAPHS.items()) + [(k, v[0]) for k, v in ex.MORE_GRAPHS.items()]:
            out["%s_%s" % (name, tag)] = np.frombuffer(ex.to_bytes(fn(x, tag), tag), dtype=dt).reshape(H, W, 4).copy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz"), **out)
    print("wrote golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
