#!/usr/bin/env python3
"""can a fused rgba8 chain carry the UNORM8 CODE (an integer-valued float) across a node boundary and fold the
/255 of the next node's load into that node's first operation?  The specification loads fl(code / 255) and
then applies the node's first fmaf; a fold replaces  fmaf(w, fl(c / 255), acc)  by  fmaf(fl(w / 255), c, acc)  (or, for the
colour grade,  fmaf(fl(c / 255), slope, offset)  by  fmaf(c, fl(slope / 255), offset)).  Both are single-rounding fmas of
DIFFERENT real products, so they can only agree by luck; this script counts, in exact arithmetic (tests/golden/exact_eval.py:
rationals, one rounding per operation), how often they do for the parameters of the BASELINE chains -- every one of the 256
codes, every tap weight, with the accumulator at the values a frame really produces (0 for the first tap).
usage: python3 scripts/u8_fold_exactness.py > profiles/r04_rgba8_fold_exactness.txt"""
import os
import sys
from fractions import Fraction

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.golden import exact_eval as ex  # noqa: E402


def gauss_weights(sigma, radius):
    import math
    e = [math.exp(-(i * i) / (2.0 * sigma * sigma)) for i in range(radius + 1)]
    s = e[0] + 2 * sum(e[1:])
    return [ex.f32(v / s) for v in e]


def main():
    c255 = Fraction(255)
    decoded = [ex.rn(Fraction(c) / c255) for c in range(256)]
    print("fold of the /255 of an rgba8 load into the first fmaf of the next node: codes (of 256) for which the folded form gives")
    print("the bits of the specified one.  acc = 0: the first tap of a sum; acc = 1/3: a running sum of ordinary size.\n")
    cases = []
    for name, sigma, radius in (("gaussian5 sigma 1.0", 1.0, 2), ("gaussian9 sigma 2.0", 2.0, 4)):
        for i, w in enumerate(gauss_weights(sigma, radius)):
            cases.append(("%s tap w%d = %.9g" % (name, i, float(w)), w))
    cases.append(("sharpen amount 0.5, centre weight 3.0", ex.f32(3.0)))
    cases.append(("sharpen amount 0.5, side weight -0.5", ex.f32(-0.5)))
    for label, w in cases:
        wf = ex.rn(w / c255)                       # the folded weight, rounded once
        for acc in (Fraction(0), ex.f32(1.0 / 3.0)):
            same = sum(ex.fma(w, decoded[c], acc) == ex.fma(wf, Fraction(c), acc) for c in range(256))
            print("  %-44s acc %-9s: %3d / 256 agree" % (label, "0" if acc == 0 else "0.3333", same))
    print()
    for slope, offset in ((1.1, -0.02), (0.95, 0.01)):
        s, o = ex.f32(slope), ex.f32(offset)
        sf = ex.rn(s / c255)
        same = sum(ex.fma(decoded[c], s, o) == ex.fma(Fraction(c), sf, o) for c in range(256))
        print("  colour grade slope %.2f offset %+.2f: t = fmaf(in, slope, offset)      : %3d / 256 agree" % (slope, offset, same))
    # the one fold that IS exact: a weight whose /255 is itself exact in binary32 -- only a weight that is a multiple of 255
    print("\nA weight w folds exactly for every code only if w / 255 is a binary32 number and no product rounds differently: w = 255 k.")
    print("No tap weight of a normalised gaussian, no sharpen weight and no slope a user would write is one.")


if __name__ == "__main__":
    main()
