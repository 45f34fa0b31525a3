"""An EXACT evaluator of the authored node specification, independent of oracle/.

Purpose: part B of golden.npz used to be produced by the oracle itself, so the
oracle graded its own output.  This file shares no code with oracle/ or with the kernels: every value is a
`fractions.Fraction`, every arithmetic operation of the specification is evaluated exactly in rationals and
rounded ONCE to the nearest binary32 (ties to even) by integer arithmetic (`rn`), which is what an IEEE
`fmaf` / multiply / subtract does.  No numpy, no C, no float arithmetic on pixel values.  The five golden
graphs are written out as explicit node sequences (no config parser, no planner).

It does not pin the reference (nothing can here: SURVEY.md section 8c) -- it removes the common-mode risk of
one author writing the oracle and the kernels to one misunderstanding of the specification.

Pure Python: a 40x24 frame through the 5-stage chain takes a few seconds.
"""
import math
import struct
from fractions import Fraction

ZERO = Fraction(0)
ONE = Fraction(1)


def rn(q):
    """q (a Fraction, finite, |q| < 2^128) rounded to the nearest binary32, ties to even; result a Fraction."""
    if q == 0:
        return ZERO
    neg = q < 0
    a = -q if neg else q
    # exponent e with 2^23 <= a / 2^e < 2^24 for normal numbers; subnormals stop at e = -149
    e = a.numerator.bit_length() - a.denominator.bit_length() - 24
    while a >= Fraction(2) ** (e + 24):
        e += 1
    while a < Fraction(2) ** (e + 23):
        e -= 1
    if e < -149:
        e = -149
    scaled = a / Fraction(2) ** e
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    r = Fraction(n) * Fraction(2) ** e
    return -r if neg else r


def f32(x):
    """a Python float (double) rounded to binary32 -- the `(float)` cast of the specification; as a Fraction"""
    return Fraction(struct.unpack("<f", struct.pack("<f", x))[0])


def fma(a, b, c):
    return rn(a * b + c)


def clamp01(v):
    return ZERO if v < 0 else (ONE if v > 1 else v)


def hash32(seed, idx, c):
    m = 0xFFFFFFFF
    h = (seed ^ (((idx * 4 + c) & m) * 0x9E3779B1)) & m
    h ^= h >> 16
    h = (h * 0x7FEB352D) & m
    h ^= h >> 15
    h = (h * 0x846CA68B) & m
    h ^= h >> 16
    return h


# ---- images: a list of rows, a row a list of texels, a texel a list of four Fractions ----------------------------
def synthetic(W, H, fmt, seed):
    """DESIGN 3.2; fmt 'f32': (u >> 8) * 2^-24, 'u8': the code u >> 24 (kept as an integer Fraction)"""
    img = []
    for y in range(H):
        row = []
        for x in range(W):
            u = [hash32(seed, y * W + x, c) for c in range(4)]
            row.append([Fraction(v >> 8, 1 << 24) for v in u] if fmt == "f32" else [Fraction(v >> 24) for v in u])
        img.append(row)
    return img


def load(img, fmt):
    """imageLoad of a whole image: rgba32f bit copy; UNORM8 code / 255 correctly rounded"""
    if fmt == "f32":
        return img
    return [[[rn(c / 255) for c in t] for t in row] for row in img]


def store(img, fmt):
    """imageStore: rgba32f bit copy; UNORM8 clamp, x255 (one rounding), round to nearest even"""
    if fmt == "f32":
        return img
    out = []
    for row in img:
        o = []
        for t in row:
            codes = []
            for v in t:
                s = rn(clamp01(v) * 255)
                n = s.numerator // s.denominator
                rem = s - n
                if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
                    n += 1
                codes.append(Fraction(n))
            o.append(codes)
        out.append(o)
    return out


def at(img, x, y):
    H, W = len(img), len(img[0])
    return img[min(max(y, 0), H - 1)][min(max(x, 0), W - 1)]


# ---- node arithmetic ----------------------------------------------------------------------------------
def gaussian_weights(sigma, r):
    if sigma <= 0:
        return [ONE] + [ZERO] * r
    e = [math.exp(-(i * i) / (2.0 * sigma * sigma)) for i in range(r + 1)]
    tot = e[0] + sum(2.0 * v for v in e[1:])
    return [f32(v / tot) for v in e]


def gaussian(img, sigma, r):
    w = gaussian_weights(sigma, r)
    H, W = len(img), len(img[0])
    tmp = []
    for y in range(H):
        row = []
        for x in range(W):
            acc = [ZERO] * 4
            for i in range(-r, r + 1):
                t = at(img, x + i, y)
                acc = [fma(w[abs(i)], t[c], acc[c]) for c in range(4)]
            row.append(acc)
        tmp.append(row)
    out = []
    for y in range(H):
        row = []
        for x in range(W):
            acc = [ZERO] * 4
            for j in range(-r, r + 1):
                t = at(tmp, x, y + j)
                acc = [fma(w[abs(j)], t[c], acc[c]) for c in range(4)]
            row.append(acc)
        out.append(row)
    return out


def colour_grade(img, slope, offset, saturation):
    slope, offset, saturation = f32(slope), f32(offset), f32(saturation)
    kr, kg, kb = f32(0.2126), f32(0.7152), f32(0.0722)
    out = []
    for row in img:
        o = []
        for t in row:
            tr, tg, tb = (fma(t[c], slope, offset) for c in range(3))
            luma = fma(kb, tb, fma(kg, tg, rn(kr * tr)))
            o.append([clamp01(fma(saturation, rn(tc - luma), luma)) for tc in (tr, tg, tb)] + [t[3]])
        out.append(o)
    return out


def sharpen(img, amount):
    amount = f32(amount)
    wc = fma(Fraction(4), amount, ONE)
    ws = -amount
    H, W = len(img), len(img[0])
    out = []
    for y in range(H):
        row = []
        for x in range(W):
            acc = [ZERO] * 4
            for (dx, dy, w) in ((0, -1, ws), (-1, 0, ws), (0, 0, wc), (1, 0, ws), (0, 1, ws)):      # N, W, C, E, S
                t = at(img, x + dx, y + dy)
                acc = [fma(w, t[c], acc[c]) for c in range(4)]
            row.append(acc)
        out.append(row)
    return out


def conv2d(img, ksize, sigma):
    r = ksize // 2
    if sigma <= 0:
        g = [1.0] + [0.0] * r
    else:
        e = [math.exp(-(i * i) / (2.0 * sigma * sigma)) for i in range(r + 1)]
        tot = e[0] + sum(2.0 * v for v in e[1:])
        g = [v / tot for v in e]
    w = [[f32(g[abs(dy)] * g[abs(dx)]) for dx in range(-r, r + 1)] for dy in range(-r, r + 1)]
    H, W = len(img), len(img[0])
    out = []
    for y in range(H):
        row = []
        for x in range(W):
            acc = [ZERO] * 4
            for dy in range(-r, r + 1):
                for dx in range(-r, r + 1):
                    t = at(img, x + dx, y + dy)
                    acc = [fma(w[dy + r][dx + r], t[c], acc[c]) for c in range(4)]
            row.append(acc)
        out.append(row)
    return out


def split_luma(img):
    """one input, two outputs: luma = fma(0.0722, b, fma(0.7152, g, 0.2126 r)); (l, l, l, a) and (fma(0.5, c - l, 0.5) ..., a)"""
    kr, kg, kb, half = f32(0.2126), f32(0.7152), f32(0.0722), Fraction(1, 2)
    luma, chroma = [], []
    for row in img:
        lr, cr = [], []
        for t in row:
            l = fma(kb, t[2], fma(kg, t[1], rn(kr * t[0])))
            lr.append([l, l, l, t[3]])
            cr.append([fma(half, rn(t[c] - l), half) for c in range(3)] + [t[3]])
        luma.append(lr)
        chroma.append(cr)
    return luma, chroma


def combination(a, b, mix):
    mix = f32(mix)
    return [[[fma(mix, rn(tb[c] - ta[c]), ta[c]) for c in range(4)] for ta, tb in zip(ra, rb)] for ra, rb in zip(a, b)]


# ---- the golden graphs as explicit node sequences; between nodes an image is stored and loaded in its format ---------
def node(fn, fmt, stored_inputs, *params):
    return store(fn(*[load(i, fmt) for i in stored_inputs], *params), fmt)


def chain3(x, fmt):
    a = node(gaussian, fmt, [x], 1.0, 2)
    b = node(colour_grade, fmt, [a], 1.1, -0.02, 1.2)
    return node(sharpen, fmt, [b], 0.5)


def chain5(x, fmt):
    c = chain3(x, fmt)
    d = node(gaussian, fmt, [c], 2.0, 4)
    return node(colour_grade, fmt, [d], 0.95, 0.01, 0.9)


def diamond(x, fmt):
    a = node(gaussian, fmt, [x], 1.5, 2)
    b = node(sharpen, fmt, [x], 0.75)
    return node(combination, fmt, [a, b], 0.25)


def gauss9(x, fmt):
    return node(gaussian, fmt, [x], 2.0, 4)


def conv7(x, fmt):
    return node(conv2d, fmt, [x], 7, 1.5)


def split2(x, fmt):
    """tests/util.py SPLIT2: split_luma -> (gaussian5 on the luma image | colour_grade on the chroma image) -> combination"""
    luma, chroma = split_luma(load(x, fmt))
    luma, chroma = store(luma, fmt), store(chroma, fmt)
    a = node(gaussian, fmt, [luma], 1.0, 2)
    b = node(colour_grade, fmt, [chroma], 1.2, -0.05, 1.1)
    return node(combination, fmt, [a, b], 0.4)


def gauss_r7_sharp(x, fmt):
    """a wide gaussian (radius 7, sigma 3) followed by a strong sharpen"""
    a = node(gaussian, fmt, [x], 3.0, 7)
    return node(sharpen, fmt, [a], 1.25)


GRAPHS = {"chain3": chain3, "chain5": chain5, "diamond": diamond, "gauss9": gauss9, "conv7": conv7}
# further graphs in golden.npz part B (round 3): name -> (function, config text the oracle / the kernels run)
MORE_GRAPHS = {
    "split2": (split2, None),              # text: tests/util.py SPLIT2
    "gauss_r7_sharp": (gauss_r7_sharp, "input -> gg -> sh -> output\ngg: gaussian { sigma: 3.0, radius: 7 }\nsh: sharpen { amount: 1.25 }"),
}


def to_bytes(img, fmt):
    """stored image -> the bytes of the frame (rgba32f little-endian floats / rgba8 codes)"""
    out = bytearray()
    for row in img:
        for t in row:
            for v in t:
                if fmt == "f32":
                    out += struct.pack("<f", float(v))      # v is exactly representable: the conversion is exact
                else:
                    out.append(int(v))
    return bytes(out)
