#!/usr/bin/env python3
"""Text-level check of shaders/*.comp against oracle/rf_oracle.c: the ORDER of the multiply-adds.

Nothing in this environment can compile GLSL, so the restatement shaders are checked as text: for
every node the sequence of fma() calls -- which loops enclose each one, in which direction they run,
which operand is the weight, which the accumulator -- must be the same as the sequence of fmaf()
calls in the oracle's function for that node, after the per-channel loops of the C code (`k`) are
dropped (GLSL works on vec4) and names are mapped to roles.  Also checked: every accumulator of an
fma is declared `precise` (GLSL may split an fma otherwise), the binding names of
shaders/passthrough.comp (`input_image`, `output_image`) and the 16x16 local size.

usage: check_glsl_taps.py [-v] [--shaders DIR]      exit code 0 = all shaders agree with the oracle
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")      # --shaders DIR: check another copy (the tests mutate one to see the checker fail)


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"(?m)^[ \t]*#[^\n]*", " ", text)     # preprocessor lines (#pragma omp ... for, #define, #version)


def tokenize(text):
    return re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.\d*f?|\.\d+f?|\d+f?|->|\+\+|--|<=|>=|==|!=|&&|\|\||[-+*/%<>=!&|^~?:;,.(){}\[\]#]", text)


class Parser:
    """Just enough C/GLSL structure: for-loops (braced or single statement) and fma calls."""

    def __init__(self, toks):
        self.t, self.i, self.loops, self.out = toks, 0, [], []

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def take(self):
        self.i += 1
        return self.t[self.i - 1]

    def balanced(self, open_, close):
        """tokens of a balanced group starting at the opening token (consumed), without the delimiters"""
        assert self.take() == open_
        depth, got = 1, []
        while depth:
            tok = self.take()
            depth += tok == open_
            depth -= tok == close
            if depth:
                got.append(tok)
        return got

    def statement(self):
        tok = self.peek()
        if tok is None:
            return
        if tok == "{":
            self.take()
            while self.peek() != "}":
                self.statement()
            self.take()
        elif tok == "for":
            self.take()
            head = self.balanced("(", ")")
            self.loops.append(self.loop_of(head))
            self.statement()
            self.loops.pop()
        elif tok in ("if", "while"):
            self.take()
            self.balanced("(", ")")
            self.statement()
            if self.peek() == "else":
                self.take()
                self.statement()
        else:
            stmt = []
            while self.peek() not in (";", None):
                if self.peek() == "{":           # initialiser list / compound literal
                    stmt += ["{"] + self.balanced("{", "}") + ["}"]
                else:
                    stmt.append(self.take())
            self.take()
            self.scan(stmt)

    @staticmethod
    def loop_of(head):
        s = " ".join(head)
        m = re.match(r"(?:int )?(\w+) = (.+?) ; \1 (<=|<) (.+?) ; (\+\+ \1|\1 \+\+)$", s)
        return (m.group(1), m.group(2), m.group(3), m.group(4)) if m else ("?", s, "", "")

    def scan(self, stmt):
        s = " ".join(stmt)
        for m in re.finditer(r"(?:(\w[\w .\[\]]*?) = )?\bfmaf? \(", s):
            dest = (m.group(1) or "").strip()
            args, depth, cur = [], 1, []
            for tok in s[m.end():].split():
                if tok == "(":
                    depth += 1
                elif tok == ")":
                    depth -= 1
                    if depth == 0:
                        break
                if tok == "," and depth == 1:
                    args.append(" ".join(cur))
                    cur = []
                else:
                    cur.append(tok)
            args.append(" ".join(cur))
            self.out.append({"loops": list(self.loops), "dest": dest, "args": args})


def function_body(text, name):
    m = re.search(r"\b%s\s*\([^)]*\)\s*\{" % re.escape(name), text)
    assert m, name
    i, depth = m.end(), 1
    while depth:
        depth += text[i] == "{"
        depth -= text[i] == "}"
        i += 1
    return text[m.end() - 1:i]


def fmas(text, func):
    p = Parser(tokenize(function_body(strip_comments(text), func)))
    p.statement()
    return p.out


def norm(s, names):
    """names: regex -> role, applied to a normalised token string"""
    s = re.sub(r"\s+", " ", s).strip()
    s = re.sub(r"\bvec4 \( (.+) \)$", r"\1", s)          # vec4(w) broadcast of a scalar weight
    s = re.sub(r" ?\. c \[ k \]| ?\[ k \]", "", s)        # per-channel index of the C code
    s = re.sub(r"(\d)f\b", r"\1", s)
    for pat, role in names:
        s = re.sub(pat, role, s)
    return s


def signature(entries, names, drop_loops=("k",)):
    sig = []
    for e in entries:
        loops = tuple((norm(v, names), norm(a, names), op, norm(b, names)) for v, a, op, b in e["loops"] if v not in drop_loops)
        sig.append((loops, tuple(norm(a, names) for a in e["args"]), norm(e["dest"], names)))
    return sig


# (shader, oracle function, name -> role maps): the maps say WHICH variable plays which role; the ORDER is what is compared
COMMON = [(r"\bradius\b|\bRADIUS\b|\bR\b|\br\b", "R")]
CHECKS = [
    ("gaussian5.comp", "rfo_gaussian",
     COMMON + [(r"weight \( i , w \)|w \[ i < 0 \? - i : i \]|\bwi\b", "W(i)"), (r"weight \( j , w \)|w \[ j < 0 \? - j : j \]|\bwj\b", "W(j)"),
               (r"\bp\b", "TEXEL"), (r"^t$", "HROW"), (r"^o$", "OUT"), (r"^acc$", "HACC")],
     COMMON + [(r"weight \( i , w \)", "W(i)"), (r"weight \( j , w \)", "W(j)"), (r"^t$", "TEXEL"), (r"^acc$", "HACC"), (r"^o$", "OUT")]),
]


def check_gaussian(shader, verbose):
    """H taps inside V taps in the shader; two separate passes over an f32 intermediate in the oracle.  Per output
    value the chains are the same: for each j ascending { for each i ascending: hacc = fma(W(i), texel, hacc) } out = fma(W(j), hacc, out)."""
    oracle = fmas(open(os.path.join(ROOT, "oracle", "rf_oracle.c")).read(), "rfo_gaussian")
    glsl = fmas(open(os.path.join(SHADERS, shader)).read(), "main")
    gl = [e for e in glsl if "exp" not in " ".join(e["args"])]
    assert len(oracle) == 2 and len(gl) == 2, (shader, len(oracle), len(gl))

    def shape(e, wvar):
        loops = [l for l in e["loops"] if l[0] != "k"]
        v, a, op, b = loops[-1]
        rad = r"radius|RADIUS|R"
        assert re.fullmatch(r"- (%s)" % rad, a) and op == "<=" and re.fullmatch(rad, b), (shader, loops[-1])    # ascending -R .. +R
        w, x, acc = e["args"]
        assert re.search(r"\b%s\b" % v, w) or w == wvar, (shader, w, v)                                    # the weight is indexed by the loop variable
        assert norm(acc, []) == norm(e["dest"], []), (shader, "accumulates into its own destination", acc, e["dest"])
        return v
    # oracle: H pass over i (weight wi), then V pass over j (weight wj)
    assert shape(oracle[0], "wi") == "i" and shape(oracle[1], "wj") == "j"
    assert "tmp" in open(os.path.join(ROOT, "oracle", "rf_oracle.c")).read()
    # shader: the i loop (H) nested in the j loop (V); V accumulates the finished H sum
    assert shape(gl[0], None) == "i" and [l[0] for l in gl[0]["loops"]] == ["j", "i"], (shader, gl[0]["loops"])
    assert shape(gl[1], None) == "j" and [l[0] for l in gl[1]["loops"]] == ["j"], (shader, gl[1]["loops"])
    assert gl[1]["args"][1].strip() == gl[0]["dest"], (shader, "V tap must take the H accumulator")
    if verbose:
        print(shader, "H: i ascending into", gl[0]["dest"], "| V: j ascending into", gl[1]["dest"])


def ordered_args(text, func, names):
    return [tuple(norm(a, names) for a in e["args"]) for e in fmas(text, func)]


def main():
    global SHADERS
    verbose = "-v" in sys.argv
    if "--shaders" in sys.argv:
        SHADERS = sys.argv[sys.argv.index("--shaders") + 1]
    oracle_c = open(os.path.join(ROOT, "oracle", "rf_oracle.c")).read()

    def sh(name):
        return open(os.path.join(SHADERS, name)).read()

    for g in ("gaussian5.comp", "gaussian9.comp", "gaussian.comp"):
        check_gaussian(g, verbose)

    # sharpen: five taps N, W, C, E, S with weights ws, ws, wc, ws, ws (+ the weight derivation fma(4, amount, 1))
    o = ordered_args(oracle_c, "rfo_sharpen", [(r"^n$", "N"), (r"^l$", "W"), (r"^c$", "C"), (r"^r$", "E"), (r"^s$", "S")])
    o_w = ordered_args(oracle_c, "rfo_sharpen_weights", [])
    g = ordered_args(sh("sharpen.comp"), "main", [(r"^n$", "N"), (r"^l$", "W"), (r"^c$", "C"), (r"^r$", "E"), (r"^s$", "S"), (r"\.0\b", ".0")])
    want = [("ws", "N", "acc"), ("ws", "W", "acc"), ("wc", "C", "acc"), ("ws", "E", "acc"), ("ws", "S", "acc")]
    assert o == want, o
    assert [tuple(x.replace("4.0", "4").replace("1.0", "1") for x in a) for a in g] == [("4", "amount", "1")] + want, g
    assert [tuple(x.replace("4.0", "4").replace("1.0", "1") for x in a) for a in o_w] == [("4", "amount", "1")], o_w
    # which texel each name holds: N = (x, y-1), W = (x-1, y), C, E = (x+1, y), S = (x, y+1)
    for src, pats in ((strip_comments(oracle_c), (r"n = load_px\([^;]*x,\s*ym\)", r"l = load_px\([^;]*xm,\s*y\)", r"r = load_px\([^;]*xp,\s*y\)", r"s = load_px\([^;]*x,\s*yp\)")),
                      (strip_comments(sh("sharpen.comp")), (r"n = imageLoad\(input_image, ivec2\(p\.x, ym\)\)", r"l = imageLoad\(input_image, ivec2\(xm, p\.y\)\)",
                                                            r"r = imageLoad\(input_image, ivec2\(xp, p\.y\)\)", r"s = imageLoad\(input_image, ivec2\(p\.x, yp\)\)"))):
        for pat in pats:
            assert re.search(pat, src), pat

    # colour grade: t_c = fma(in_c, slope, offset); luma = fma(.0722, tb, fma(.7152, tg, .2126 tr)); out_c = fma(saturation, t_c - luma, luma)
    names = [(r"p \. c \[ 0 \]|c \. r", "IN_R"), (r"p \. c \[ 1 \]|c \. g", "IN_G"), (r"p \. c \[ 2 \]|c \. b", "IN_B"),
             (r"tr - luma|\bdr\b", "DR"), (r"tg - luma|\bdg\b", "DG"), (r"tb - luma|\bdb\b", "DB")]
    o = ordered_args(oracle_c, "rfo_colour_grade", names)
    for fname in ("colour_grade.comp", "colour_grade_inplace.comp"):
        g = ordered_args(sh(fname), "main", names)
        flat_o = [a for a in o if "fmaf" not in " ".join(a)]            # the oracle nests the two luma fmas in one expression
        assert ("IN_R", "slope", "offset") in o and ("IN_G", "slope", "offset") in o and ("IN_B", "slope", "offset") in o
        assert g[:3] == [("IN_R", "slope", "offset"), ("IN_G", "slope", "offset"), ("IN_B", "slope", "offset")], g
        assert g[3:5] == [("0.7152", "tg", "luma"), ("0.0722", "tb", "luma")], g          # inner fma first, then the outer one
        assert re.search(r"luma = 0\.2126 \* tr", strip_comments(sh(fname)))
        assert re.search(r"fmaf\(0\.0722f, tb, fmaf\(0\.7152f, tg, 0\.2126f \* tr\)\)", strip_comments(oracle_c))
        assert g[5:] == [("saturation", "DR", "luma"), ("saturation", "DG", "luma"), ("saturation", "DB", "luma")], g
        assert [a for a in flat_o if a[0] == "saturation"] == [("saturation", "DR", "luma"), ("saturation", "DG", "luma"), ("saturation", "DB", "luma")], flat_o

    # pulse: slope = fma(amount, frac(t), 1), then the grade arithmetic with (slope, 0, 1)
    o = ordered_args(oracle_c, "rfo_pulse_slope", [(r"\.0\b", "")])
    g = ordered_args(sh("pulse.comp"), "main", [(r"c \. r", "IN_R"), (r"c \. g", "IN_G"), (r"c \. b", "IN_B"), (r"\bdr\b", "DR"), (r"\bdg\b", "DG"), (r"\bdb\b", "DB"), (r"\.0\b", "")])
    assert o == [("amount", "f", "1")], o
    assert g[0] == ("amount", "f", "1") and g[1:4] == [("IN_R", "slope", "0"), ("IN_G", "slope", "0"), ("IN_B", "slope", "0")], g
    assert g[4:6] == [("0.7152", "tg", "luma"), ("0.0722", "tb", "luma")] and g[6:] == [("1", "DR", "luma"), ("1", "DG", "luma"), ("1", "DB", "luma")], g
    assert re.search(r"f = t - floorf\(t\)", strip_comments(oracle_c)) and re.search(r"f = phase_rf_time - floor\(phase_rf_time\)", strip_comments(sh("pulse.comp")))

    # conv2d: dy outer, dx inner, both ascending -r .. r; weight [(dy+r)*K + (dx+r)]
    for text, func in ((oracle_c, "rfo_conv2d"), (sh("conv2d.comp"), "main")):
        e = [x for x in fmas(text, func)]
        assert len(e) == 1, (func, len(e))
        loops = [l for l in e[0]["loops"] if l[0] not in ("k", "x", "y")]      # the C code's per-pixel and per-channel loops = GLSL invocations / vec4
        assert [(l[0], l[1], l[2], l[3]) for l in loops] == [("dy", "- r", "<=", "r"), ("dx", "- r", "<=", "r")], loops
        w = norm(e[0]["args"][0], [(r"\bwt\b", "weights [ ( dy + r ) * K + ( dx + r ) ]")])
        assert w == "weights [ ( dy + r ) * K + ( dx + r ) ]", w
        assert norm(e[0]["args"][2], []) == norm(e[0]["dest"], []) == "o", e[0]
    assert re.search(r"wt = weights\[\(dy \+ r\) \* K \+ \(dx \+ r\)\]", strip_comments(oracle_c))

    # combination: out = fma(mix, b - a, a)
    o = ordered_args(oracle_c, "rfo_mix", [(r"pb - pa", "D"), (r"\bpa\b", "A")])
    g = ordered_args(sh("combination.comp"), "main", [(r"^d$", "D"), (r"^a$", "A")])
    assert o == g == [("mix", "D", "A")], (o, g)
    assert re.search(r"d = b - a", strip_comments(sh("combination.comp")))

    # split_luma: two output images; luma = fma(0.0722, b, fma(0.7152, g, 0.2126 r)), chroma = fma(0.5, c - luma, 0.5)
    oc, gs = strip_comments(oracle_c), strip_comments(sh("split_luma.comp"))
    assert re.search(r"l = fmaf\(0\.0722f, p\.c\[2\], fmaf\(0\.7152f, p\.c\[1\], 0\.2126f \* p\.c\[0\]\)\)", oc)
    assert re.search(r"l = 0\.2126 \* c\.r;\s*l = fma\(0\.7152, c\.g, l\);\s*l = fma\(0\.0722, c\.b, l\);", gs)
    assert len(re.findall(r"fmaf\(0\.5f, p\.c\[\d\] - l, 0\.5f\)", oc)) == 3 and len(re.findall(r"fma\(0\.5, d[rgb], 0\.5\)", gs)) == 3
    assert re.search(r"binding = 1, rgba32f\) uniform writeonly image2D luma_image;", gs) and re.search(r"binding = 2, rgba32f\) uniform writeonly image2D chroma_image;", gs)

    # every shader: the contract of shaders/passthrough.comp, and `precise` on every fma destination
    for f in sorted(os.listdir(SHADERS)):
        if not f.endswith(".comp"):
            continue
        text = strip_comments(sh(f))
        assert "#version 450" in sh(f) and re.search(r"local_size_x = 16, local_size_y = 16", text), f
        if f == "combination.comp":
            assert "input_image0" in text and "input_image1" in text and "output_image" in text
        elif f == "colour_grade_inplace.comp":
            assert re.search(r"uniform image2D image;", text)
        elif f == "split_luma.comp":
            assert re.search(r"binding = 0, rgba32f\) uniform readonly image2D input_image;", text), f
        elif f == "unsharp_mask.comp":
            # the twin of shaders/unsharp_mask.stage.hip: the bindings librfhip.so gives the declared names (inputs 0.., outputs after them)
            for b, decl in enumerate(("readonly image2D input_image", "readonly image2D blurred_image", "writeonly image2D output_image", "writeonly image2D mask_image")):
                assert re.search(r"binding = %d, rgba32f\) uniform %s;" % (b, decl), text), (f, decl)
        else:
            assert re.search(r"binding = 0, rgba32f\) uniform readonly image2D input_image;", text), f
            assert re.search(r"binding = 1, rgba32f\) uniform writeonly image2D output_image;", text), f
        for e in fmas(sh(f), "main"):
            if "exp" in " ".join(e["args"]):
                continue
            base = re.split(r"[ .\[]", e["dest"].replace("precise ", "").replace("vec4 ", "").replace("float ", "").strip())[0]
            assert re.search(r"precise (?:vec4|float) [^;]*\b%s\b" % re.escape(base), text), (f, "fma destination not precise:", e["dest"])
        if verbose:
            print(f, "ok")
    print("shaders/*.comp: tap order agrees with oracle/rf_oracle.c")


if __name__ == "__main__":
    main()
