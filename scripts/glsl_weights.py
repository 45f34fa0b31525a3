#!/usr/bin/env python3
"""Prints the explicit gaussian weights `w0: ..., w1: ...` of a (sigma, radius) pair as config parameters.

librfhip.so and the oracle derive w_i = (float)(exp(-i^2 / (2 sigma^2)) / S) on the host in double
(DESIGN.md 3); shaders/gaussian*.comp cannot (GLSL exp() is single precision and not correctly
rounded), so a config that must produce the same bits on the reference's Vulkan path passes the
weights explicitly, in plain decimals (the grammar has no exponents).  Nine significant digits round-trip an f32 through Rust's str::parse::<f32>
(src/render.rs:173) and through this repository's parser.

usage: glsl_weights.py <sigma> <radius>        e.g.  glsl_weights.py 1.0 2
"""
import math
import struct
import sys


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def weights(sigma, radius):
    sigma = f32(sigma)
    if not sigma > 0.0:
        return [1.0] + [0.0] * radius
    s2 = 2.0 * sigma * sigma
    e = [math.exp(-(i * i) / s2) for i in range(radius + 1)]
    total = e[0] + 2.0 * sum(e[1:])
    return [f32(v / total) for v in e]


def fixed(w):
    """Nine significant digits in plain decimal notation: the config grammar has no exponents
    (config_grammar.lalrpop:74-78: [0-9]+ or -?[0-9]+\\.[0-9]+)."""
    if w == 0.0:
        return "0.0"
    from decimal import Decimal
    d = Decimal(repr(float(w)))
    digits = max(1, 9 - d.adjusted() - 1)
    s = format(d, ".%df" % digits)
    return s if "." in s else s + ".0"


def as_params(sigma, radius):
    return ", ".join("w%d: %s" % (i, fixed(w)) for i, w in enumerate(weights(sigma, radius)))


if __name__ == "__main__":
    print(as_params(float(sys.argv[1]), int(sys.argv[2])))
