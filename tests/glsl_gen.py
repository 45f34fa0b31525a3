"""Test infrastructure: generated GLSL compute shaders for differential runs -- the same text through Mesa's GLSL compiler (tests/mesa_glsl.py)
and through rf_glsl.cpp's translation (tests/glsl_host.py).  A shader is a list of typed assignments over the texel, four neighbours, the
invocation's coordinates and two uniforms; every operation used is ONE correctly rounded IEEE operation (or integer arithmetic), so two
conforming implementations must agree bit for bit; float values are kept in a bounded range (no NaN, no infinity, conversions in range),
integer divisors and shift counts are made safe in the text itself.  Statement forms cover: operators by precedence with and without
parentheses, vector constructors and swizzles (also as l-values), compound assignment, the ternary operator, int / uint / float / bool
conversions, integer and bit built-ins, the exactly rounded float built-ins, if / else, for loops, helper functions with out parameters,
arrays (also as values), structs."""
import random

HEAD = """#version 450
layout (local_size_x = 16, local_size_y = 16) in;
layout (binding = 0, rgba32f) uniform readonly image2D input_image;
layout (binding = 1, rgba32f) uniform writeonly image2D output_image;
layout (binding = 2) uniform Params { float gain; int shift; };
layout (std430, binding = 3) buffer Stats { uint counters[8]; int low; uint high; };
"""

SWZ = "xyzw"
SKIP_FLOAT_FORMS = set()      # forms of Gen.f() left out (bisection of a difference)
SKIP_VEC_FORMS = set()
# GLSL lets an implementation CONTRACT a * b + c into one fused operation unless the result is `precise`; two implementations only have to
# agree bit for bit on precise code (Mesa fuses where it may, this library never does): every float the generator declares is precise
PRECISE = "precise "


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        self.vars = {"float": [], "int": [], "uint": [], "bool": [], "vec2": [], "vec3": [], "vec4": [], "ivec2": []}
        self.lines = []
        self.n = 0

    def name(self):
        self.n += 1
        return "v%d" % self.n

    def pick(self, t):
        return self.r.choice(self.vars[t])

    # ---- expressions ---------------------------------------------------------------------------------------------------------------------
    def f(self, d=0):
        """a float expression"""
        r = self.r
        k = r.randrange(16 if d < 3 else 3)
        if k in SKIP_FLOAT_FORMS:
            k = 3
        if k == 0 or not self.vars["float"]:
            return r.choice(["0.5", "1.25", "2.0", "0.1", "3.0", "gain", "0.75"])
        if k == 1:
            return self.pick("float")
        if k == 2:
            v = self.pick(r.choice([t for t in ("vec2", "vec3", "vec4") if self.vars[t]]))
            return "%s.%s" % (v, r.choice(SWZ[:int(self.vtype(v)[-1])]))
        if k == 3:
            return "%s %s %s" % (self.f(d + 1), r.choice("+-*"), self.f(d + 1))
        if k == 4:
            return "(%s %s %s)" % (self.f(d + 1), r.choice("+-*"), self.f(d + 1))
        if k == 5:
            return "%s / (abs(%s) + 0.5)" % (self.f(d + 1), self.f(d + 1))
        if k == 6:
            return "%s(%s)" % (r.choice(["abs", "floor", "ceil", "fract", "trunc", "sign", "roundEven"]), self.f(d + 1))
        if k == 7:
            return "sqrt(abs(%s))" % self.f(d + 1)
        if k == 8:
            return "%s(%s, %s)" % (r.choice(["min", "max", "step", "mod"]), self.f(d + 1), "abs(%s) + 0.25" % self.f(d + 1))
        if k == 9:
            return "clamp(%s, %s, %s)" % (self.f(d + 1), r.choice(["0.0", "-1.0", "0.25"]), r.choice(["1.0", "2.5", "4.0"]))
        if k == 10:
            return "float(%s)" % (self.i(d + 1) if r.random() < 0.6 else self.u(d + 1))
        if k == 11:
            return "(%s ? %s : %s)" % (self.b(d + 1), self.f(d + 1), self.f(d + 1))
        if k == 12:
            return "-%s" % self.atom_f()
        if k == 13:
            n = r.choice([2, 3, 4])
            return "dot(%s, %s)" % (self.v(n, d + 1), self.v(n, d + 1))
        if k == 14:
            # (not fma(): Mesa evaluates it with two roundings or with one depending on its operands -- fma(x, 1.25, 3.0) fused where
            # fma(x, 1.25, y) is split, seed 20178 of profiles/r04_fuzz_glsl_mesa.txt; the shipped shaders' fma()s are covered one by one)
            return "(%s * %s + %s)" % (self.f(d + 1), self.f(d + 1), self.f(d + 1))
        return "mix(%s, %s, %s)" % (self.f(d + 1), self.f(d + 1), r.choice(["0.25", "0.5", "gain * 0.125"]))

    def atom_i(self, d):
        return "(%s)" % self.i(d)

    def atom_f(self):
        return self.pick("float") if self.vars["float"] else "gain"

    def i(self, d=0):
        r = self.r
        k = r.randrange(12 if d < 3 else 3)
        if k == 0 or not self.vars["int"]:
            return r.choice(["1", "3", "7", "-2", "shift", "p.x", "p.y", "12"])
        if k == 1:
            return self.pick("int")
        if k == 2:
            return "%s.%s" % (self.pick("ivec2"), r.choice("xy")) if self.vars["ivec2"] else "p.x"
        if k == 3:
            return "%s %s %s" % (self.i(d + 1), r.choice(["+", "-", "*", "&", "|", "^"]), self.i(d + 1))
        if k == 4:
            return "(%s %s %s)" % (self.i(d + 1), r.choice(["+", "-", "*", "&", "|", "^"]), self.i(d + 1))
        if k == 5:
            return "(%s & 0xffff) %s ((%s & 7) + 1)" % (self.atom_i(d + 1), r.choice("/%"), self.atom_i(d + 1))      # operands of % must not be negative (undefined in GLSL); abs() would keep INT_MIN
        if k == 6:
            return "((%s) %s ((%s) & 7))" % (self.i(d + 1), r.choice(["<<", ">>"]), self.i(d + 1))
        if k == 7:
            return "%s(%s, %s)" % (r.choice(["min", "max"]), self.i(d + 1), self.i(d + 1))
        if k == 8:
            return "int(clamp(%s, -100.0, 100.0))" % self.f(d + 1)
        if k == 9:
            return "(%s ? %s : %s)" % (self.b(d + 1), self.i(d + 1), self.i(d + 1))
        if k == 10:
            return "%s(%s)" % (r.choice(["abs", "sign", "bitCount", "findMSB", "findLSB", "~", "-"]), self.i(d + 1))
        return "int((%s) & 0xffffu)" % self.u(d + 1)

    def u(self, d=0):
        r = self.r
        k = r.randrange(9 if d < 3 else 2)
        if k == 0 or not self.vars["uint"]:
            return r.choice(["1u", "5u", "255u", "0x80000001u", "uint(p.x)", "gl_GlobalInvocationID.y"])
        if k == 1:
            return self.pick("uint")
        if k == 2:
            return "%s %s %s" % (self.u(d + 1), r.choice(["+", "-", "*", "&", "|", "^"]), self.u(d + 1))
        if k == 3:
            return "((%s) %s ((%s) & 15u))" % (self.u(d + 1), r.choice(["<<", ">>"]), self.u(d + 1))
        if k == 4:
            return "(%s) %s ((%s) %% 9u + 1u)" % (self.u(d + 1), r.choice("/%"), self.u(d + 1))
        if k == 5:
            return "uint(%s)" % self.i(d + 1)
        if k == 6:
            return "bitfieldExtract(%s, %d, %d)" % (self.u(d + 1), r.randrange(0, 12), r.randrange(1, 12))
        if k == 7:
            return "uint(%s(%s))" % (r.choice(["bitCount", "findMSB"]), self.u(d + 1))
        return "floatBitsToUint(%s)" % self.f(d + 1)

    def b(self, d=0):
        r = self.r
        k = r.randrange(7 if d < 3 else 2)
        if k == 0:
            return "%s %s %s" % (self.f(d + 1), r.choice(["<", ">", "<=", ">="]), self.f(d + 1))
        if k == 1:
            return "%s %s %s" % (self.atom_i(d + 1), r.choice(["<", ">", "==", "!=", "<=", ">="]), self.atom_i(d + 1))      # (& | ^ bind looser than comparisons)
        if k == 2 and self.vars["bool"]:
            return self.pick("bool")
        if k == 3:
            return "(%s %s %s)" % (self.b(d + 1), r.choice(["&&", "||", "^^"]), self.b(d + 1))
        if k == 4:
            return "!(%s)" % self.b(d + 1)
        if k == 5:
            n = r.choice([2, 3, 4])
            return "%s(%s(%s, %s))" % (r.choice(["any", "all"]), r.choice(["lessThan", "greaterThanEqual", "equal", "notEqual"]), self.v(n, d + 1), self.v(n, d + 1))
        return "(%s) == (%s)" % (self.u(d + 1), self.u(d + 1))

    def vtype(self, name):
        return next(t for t, vs in self.vars.items() if name in vs)

    def v(self, n, d=0):
        """a vecN expression"""
        r = self.r
        t = "vec%d" % n
        k = r.randrange(10 if d < 3 else 3)
        if k in SKIP_VEC_FORMS:
            k = 3
        if k == 0 or not self.vars[t]:
            if n == 2:
                return r.choice(["vec2(%s)" % self.f(d + 1), "vec2(%s, %s)" % (self.f(d + 1), self.f(d + 1)), "vec2(p)"])
            if n == 3:
                return r.choice(["vec3(%s)" % self.f(d + 1), "vec3(%s, %s)" % (self.v(2, d + 1), self.f(d + 1)), "c.rgb", "vec3(%s, %s)" % (self.f(d + 1), self.v(2, d + 1))])
            return r.choice(["vec4(%s)" % self.f(d + 1), "vec4(%s, %s)" % (self.v(2, d + 1), self.v(2, d + 1)), "c", "vec4(%s, %s)" % (self.v(3, d + 1), self.f(d + 1)), "vec4(p, p)"])
        if k == 1:
            return self.pick(t)
        if k == 2:
            src = r.choice([s for s in ("vec2", "vec3", "vec4") if self.vars[s]] or [t])
            if not self.vars[src]:
                return self.v(n, d + 1)
            m = int(src[-1])
            return "%s.%s" % (self.pick(src), "".join(r.choice(SWZ[:m]) for _ in range(n)))
        if k == 3:
            return "%s %s %s" % (self.v(n, d + 1), r.choice("+-*"), self.v(n, d + 1))
        if k == 4:
            return "(%s %s %s)" % (self.v(n, d + 1), r.choice("+-*"), self.f(d + 1))
        if k == 5:
            return "%s * %s" % (self.f(d + 1), self.v(n, d + 1))
        if k == 6:
            return "%s / (abs(%s) + %s(0.5))" % (self.v(n, d + 1), self.v(n, d + 1), t)
        if k == 7:
            return "%s(%s)" % (r.choice(["abs", "floor", "fract", "sign", "ceil"]), self.v(n, d + 1))
        if k == 8:
            return "%s(%s, %s)" % (r.choice(["min", "max"]), self.v(n, d + 1), r.choice([self.v(n, d + 1), self.f(d + 1)]))
        return "clamp(%s, 0.0, 1.0)" % self.v(n, d + 1)

    # ---- statements ----------------------------------------------------------------------------------------------------------------------
    def bounded(self, t, e):
        if t == "float":
            return "clamp(%s, -8.0, 8.0)" % e
        if t.startswith("vec"):
            return "clamp(%s, %s(-8.0), %s(8.0))" % (e, t, t)
        if t == "int":
            return "(%s) & 0xffff" % e
        return e

    def declare(self, t, e, indent="    "):
        n = self.name()
        self.lines.append("%s%s%s %s = %s;" % (indent, PRECISE if t == "float" or t.startswith("vec") else "", t, n, self.bounded(t, e)))
        self.vars[t].append(n)
        return n

    def statement(self):
        r = self.r
        k = r.randrange(19)
        if k == 18:      # invocations meet in a storage block: atomic functions whose result does not depend on their order
            op = r.randrange(4)      # a counter only ever meets ONE kind of operation (Add then Max is not Max then Add)
            self.lines.append("    atomic%s(counters[%d + (%s & 1)], %s & 255u);" % (["Add", "Max", "Or", "Xor"][op], 2 * op, self.atom_i(2), "(%s)" % self.u(2)))
            self.lines.append("    atomicMin(low, %s); atomicMax(high, %s);" % (self.atom_i(2), "(%s)" % self.u(2)))
            return
        if k <= 2:
            self.declare("float", self.f())
        elif k == 3:
            self.declare("int", self.i())
        elif k == 4:
            self.declare("uint", self.u())
        elif k == 5:
            self.declare("bool", self.b())
        elif k <= 7:
            n = r.choice([2, 3, 4])
            self.declare("vec%d" % n, self.v(n))
        elif k == 8:
            self.declare("ivec2", r.choice(["p + ivec2(%s, %s)" % (self.i(2), self.i(2)), "(ivec2(%s) & 0xffff) %% 50" % self.i(1), "ivec2(clamp(vec2(%s, %s), vec2(-100.0), vec2(100.0)))" % (self.f(2), self.f(2)), "abs(p - ivec2(%s))" % self.i(2)]))
        elif k == 9 and self.vars["vec4"]:      # swizzle l-values and compound assignment
            v = self.pick("vec4")
            sw = "".join(r.sample(SWZ, r.choice([1, 2, 3])))
            rhs = self.f(1) if len(sw) == 1 else self.v(len(sw), 1)
            self.lines.append("    %s.%s %s %s;" % (v, sw, r.choice(["=", "+=", "*=", "-="]), rhs))
            self.lines.append("    %s = clamp(%s, vec4(-8.0), vec4(8.0));" % (v, v))
        elif k == 10 and self.vars["float"]:      # if / else
            v = self.pick("float")
            self.lines.append("    if (%s) { %s = %s; } else { %s %s %s; }" % (self.b(), v, self.bounded("float", self.f(1)), v, r.choice(["+=", "-="]), "0.125"))
        elif k == 11:      # a loop accumulating into a new float and a new int
            a, n = self.name(), self.name()
            self.lines.append("    precise float %s = 0.0; int %s = 0;" % (a, n))
            self.lines.append("    for (int k = 0; k < %d; ++k) { if ((k & 1) == %d) continue; %s += %s * float(k); %s += k ^ (%s); if (%s > 6.0) break; }" % (
                r.randrange(2, 7), r.randrange(2), a, self.bounded("float", self.f(2)), n, self.i(2), a))
            self.lines.append("    %s = clamp(%s, -8.0, 8.0); %s = %s & 1023;" % (a, a, n, n))
            self.vars["float"].append(a)
            self.vars["int"].append(n)
        elif k == 12:      # arrays as values, a helper with an out parameter
            a, b2, o = self.name(), self.name(), self.name()
            self.lines.append("    float %s[3] = float[3](%s, %s, %s);" % (a, self.f(2), self.f(2), self.f(2)))
            self.lines.append("    float %s[3] = %s; %s[%s] = %s;" % (b2, a, b2, "(%s & 0xffff) %% 3" % self.atom_i(2), self.f(2)))
            self.lines.append("    precise float %s; swap_sum(%s, %s);" % (o, b2, o))
            self.lines.append("    %s = clamp(%s + (%s == %s ? 1.0 : 0.0), -8.0, 8.0);" % (o, o, a, b2))
            self.vars["float"].append(o)
        elif k == 14:      # matrices: constructors, products, transpose, columns
            m, n2 = self.name(), self.name()
            dim = r.choice([2, 3])
            vt = "vec%d" % dim
            cols = ", ".join(self.v(dim, 2) for _ in range(dim))
            self.lines.append("    mat%d %s = mat%d(%s);" % (dim, m, dim, cols))
            self.lines.append("    mat%d %s = %s;" % (dim, n2, r.choice(["transpose(%s)" % m, "%s * mat%d(0.5)" % (m, dim), "mat%d(%s)" % (dim, self.f(2)), "%s * 0.25 + %s" % (m, m)])))
            self.lines.append("    precise %s %s_c = %s[%d];" % (vt, m, n2, r.randrange(dim)))
            self.lines.append("    %s_c = clamp(%s_c, %s(-8.0), %s(8.0));" % (m, m, vt, vt))
            self.vars[vt].append(m + "_c")
        elif k == 15:      # switch, while, a helper with an inout parameter and an early return
            a = self.name()
            self.lines.append("    precise float %s = %s;" % (a, self.bounded("float", self.f(1))))
            self.lines.append("    switch (%s & 3) { case 0: %s += 0.5; break; case 1: %s = -%s; case 2: %s *= 0.5; break; default: bump(%s, %s); }" % (
                self.atom_i(2), a, a, a, a, a, self.i(2)))
            self.lines.append("    { int guard = 0; while (%s > 0.25 && guard < 6) { %s *= 0.5; ++guard; } }" % (a, a))
            self.vars["float"].append(a)
        elif k == 16:      # integer vectors
            q = self.name()
            self.lines.append("    ivec3 %s = ivec3(%s, %s) %s ivec3(%s);" % (q, self.pick("ivec2"), self.i(2), r.choice(["+", "-", "*", "&", "|", "^"]), self.i(2)))
            self.lines.append("    uvec2 %s_u = uvec2(%s.zx) >> uvec2(%s & 7, 3) ;" % (q, q, self.atom_i(2)))
            self.declare("int", "%s.x + %s.y - %s.z + int(%s_u.x & 255u) + int(bitfieldInsert(%s, %s, %d, %d) & 0xfffu)" % (q, q, q, q, self.u(2), self.u(2), r.randrange(0, 8), r.randrange(1, 8)))
        else:      # structs
            s, o = self.name(), self.name()
            self.lines.append("    Pair %s = Pair(%s, %s);" % (s, self.v(2, 1), self.i(1)))
            self.lines.append("    Pair %s = %s; %s.n += 1;" % (o, s, o))
            self.declare("float", "%s.a.x + %s.a.y * 0.5 + float(%s.n & 7) + (%s == %s ? 2.0 : 0.0)" % (s, o, o, s, o))

    def shader(self, statements=14):
        self.lines = [
            "    ivec2 size = imageSize(output_image);",
            "    ivec2 p = ivec2(gl_GlobalInvocationID.xy);",
            "    if (p.x >= size.x || p.y >= size.y) return;",
            "    vec4 c = imageLoad(input_image, p);",
            "    vec4 e = imageLoad(input_image, clamp(p + ivec2(1, 0), ivec2(0), size - 1));",
            "    vec4 s = imageLoad(input_image, clamp(p + ivec2(0, 1), ivec2(0), size - 1));",
        ]
        self.vars["vec4"] += ["c", "e", "s"]
        self.vars["ivec2"] += ["p"]
        for _ in range(statements):
            self.statement()
        # everything computed ends in the stored texel
        acc = ["precise vec4 o = c * 0.0;"]
        for i, v in enumerate(self.vars["float"]):
            acc.append("o.%s += %s;" % (SWZ[i % 4], v))
        for i, v in enumerate(self.vars["int"]):
            acc.append("o.%s += float(%s & 1023) * 0.0009765625;" % (SWZ[i % 4], v))
        for i, v in enumerate(self.vars["uint"]):
            acc.append("o.%s += float(%s & 1023u) * 0.0009765625;" % (SWZ[(i + 1) % 4], v))
        for i, v in enumerate(self.vars["bool"]):
            acc.append("o.%s += %s ? 0.5 : 0.0;" % (SWZ[(i + 2) % 4], v))
        for v in self.vars["vec2"]:
            acc.append("o.xy += %s; o.zw -= %s.yx;" % (v, v))
        for v in self.vars["vec3"]:
            acc.append("o.xyz += %s; o.w += %s.z;" % (v, v))
        for v in self.vars["vec4"][3:]:
            acc.append("o += %s;" % v)
        for v in self.vars["ivec2"][1:]:
            acc.append("o.xy += vec2(%s & 127) * 0.0078125;" % v)
        body = "\n".join(self.lines + ["    " + a for a in acc] + ["    imageStore(output_image, p, o);"])
        helpers = ("struct Pair { vec2 a; int n; };\n"
                   "void bump(inout float x, int n) { if (n < 0) { x -= 0.25; return; } for (int i = 0; i < (n & 3); ++i) x += 0.125; }\n"
                   "void swap_sum(float w[3], out float total) { float t = w[0]; w[0] = w[2]; w[2] = t; precise float half_of = w[1] * 0.5; precise float quarter = w[2] * 0.25; "
                   "precise float r = w[0] - half_of; r = r + quarter; total = r; }\n")
        return HEAD + helpers + "void main()\n{\n" + body + "\n}\n"


def generate(seed, statements=14):
    return Gen(seed).shader(statements)
