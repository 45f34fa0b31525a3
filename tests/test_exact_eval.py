"""CPU: the exact evaluator behind golden.npz part B (tests/golden/exact_eval.py) -- its rounding primitive against
IEEE arithmetic done another way, and the committed vectors against a fresh evaluation.  Neither imports oracle/:
the oracle (test_oracle.py) and the kernels (test_gpu_parity.py) are checked AGAINST these vectors."""
import os
import random
import struct
from fractions import Fraction

import numpy as np

from tests.golden import exact_eval as ex

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))


def test_rn_is_ieee_round_to_nearest_even():
    rnd = random.Random(7)
    for _ in range(4000):
        a = np.float32(rnd.uniform(-4, 4) * 2.0 ** rnd.randint(-30, 10))
        b = np.float32(rnd.uniform(-4, 4) * 2.0 ** rnd.randint(-30, 10))
        # a 24 x 24-bit product is exact in float64; the cast to float32 then rounds once
        want = np.float32(np.float64(a) * np.float64(b))
        got = ex.rn(Fraction(float(a)) * Fraction(float(b)))
        assert float(got) == float(want)
    # ties go to even; subnormals; the smallest normal
    ulp = Fraction(1, 1 << 23)
    assert ex.rn(1 + ulp / 2) == 1 and ex.rn(1 + 3 * ulp / 2) == 1 + 2 * ulp and ex.rn(1 + ulp / 2 + ulp / 1024) == 1 + ulp
    assert ex.rn(Fraction(1, 1 << 150)) == 0 and ex.rn(Fraction(3, 1 << 150)) == Fraction(1, 1 << 148)
    assert ex.rn(Fraction(1, 1 << 126)) == Fraction(1, 1 << 126)
    assert ex.rn(-(1 + ulp / 2)) == -1
    assert ex.f32(0.1) == Fraction(struct.unpack("<f", struct.pack("<f", 0.1))[0])


def test_unorm8_rules():
    img = [[[Fraction(c) for c in (0, 1, 127, 255)]]]
    dec = ex.load(img, "u8")
    assert [float(v) for v in dec[0][0]] == [float(np.float32(c) / np.float32(255)) for c in (0, 1, 127, 255)]
    # store: clamp, x255, ties to even (0.5/255 -> 0, 1.5/255 -> 2), round trip of every code
    half = [[[ex.rn(Fraction(1, 510)), ex.rn(Fraction(3, 510)), Fraction(-1), Fraction(2)]]]
    want = [int(np.rint(np.float32(float(v)) * np.float32(255))) for v in half[0][0][:2]] + [0, 255]
    assert [int(v) for v in ex.store(half, "u8")[0][0]] == want
    codes = [[[Fraction(c)] * 4 for c in range(256)]]
    assert ex.store(ex.load(codes, "u8"), "u8") == codes


def test_committed_vectors_are_what_the_evaluator_produces():
    W, H = 40, 24
    for tag, names in (("f32", ("chain3", "gauss9")), ("u8", ("diamond",))):
        x = ex.synthetic(W, H, tag, 0x5EED0002)
        assert ex.to_bytes(x, tag) == GOLDEN["in_" + tag].tobytes()
        for name in names:
            assert ex.to_bytes(ex.GRAPHS[name](x, tag), tag) == GOLDEN["%s_%s" % (name, tag)].tobytes(), (name, tag)


def test_evaluator_does_not_import_the_oracle():
    import re
    for path in (ex.__file__, os.path.join(os.path.dirname(ex.__file__), "make_golden.py")):
        imports = [line for line in open(path).read().splitlines() if re.match(r"\s*(import|from)\s", line)]
        assert imports and not any(re.search(r"\boracle\b|\butil\b|reforge_amd", line) for line in imports), (path, imports)
