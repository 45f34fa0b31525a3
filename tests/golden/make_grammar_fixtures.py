#!/usr/bin/env python3
"""Golden vectors for the config DSL, derived from the REFERENCE'S OWN GRAMMAR FILE.

    python3 tests/golden/make_grammar_fixtures.py [/root/reference/src/config/config_grammar.lalrpop]
        -> tests/golden/grammar_fixtures.json.gz

The one artefact of the reference that pins the parser is `src/config/config_grammar.lalrpop` (lines 7-81): the
productions, and the terminals LALRPOP builds its lexer from.  This script READS that file -- it holds no copy of the
grammar -- and turns it into a recogniser by general means:

  * a reader for the subset of LALRPOP's own syntax the file uses (`Name: Type = { alt => action, ... };`,
    `<name:Symbol>` captures, "literal" and r"regex" terminals, the `( ... )*` macro), which keeps the symbols of
    every alternative and skips the Rust types and action code;
  * LALRPOP's generated lexer, as documented and as lalrpop-util 0.20 implements it: white space (`\\s`) between tokens
    is skipped; at each position every terminal is tried, the LONGEST match wins, a quoted literal beats a regex of
    the same length (two regexes of the same length would have been rejected when the reference was built); no match
    is an InvalidToken error.  The regex dialect of the file's five patterns is common to Rust's `regex` crate and
    Python's `re` (classes, escapes, greedy repetition, leftmost-first alternation);
  * an Earley recogniser over the token stream (the grammar is left-recursive; LALR(1), hence unambiguous: the one
    derivation is kept), start symbol = the `pub` nonterminal;
  * the value of a derivation, following the ten action bodies of the file (`vec![..]`/`push`, the (name, Option)
    pair, `Pipeline { name, pipeline_type, parameters }`, the parameter list in source order, `to_string`): the
    reader checks that every production still has the captures those actions name, and fails loudly otherwise.

With it, >= 2000 generated texts (well-formed configs, mutated ones, token soup, hand-picked lexer traps) are
classified accept / reject and, when accepted, written out with their syntax tree:

    {"exprs": [["pipeline", name, type, [[key, value], ...]] | ["graph", [[name, descriptor | null], ...]] | ["comment", text]]}

`tests/test_grammar_fixtures.py` holds BOTH `reforge_amd/csrc/rf_config.cpp` (through rf_config_syntax) and
`oracle/graph.py` to these vectors; nothing of the reference travels (the .json.gz holds texts and trees only)."""
import gzip
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_GRAMMAR = "/root/reference/src/config/config_grammar.lalrpop"
OUT = os.path.join(HERE, "grammar_fixtures.json.gz")


# --------------------------------------------------------------------------------------------------------------
# 1. the grammar file -> productions
# --------------------------------------------------------------------------------------------------------------
def _grammar_tokens(src):
    """tokens of the .lalrpop text: ('id', s) ('lit', s) ('re', s) ('p', punctuation)"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c.isspace():
            i += 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif c == "r" and i + 1 < n and src[i + 1] == '"':
            j = i + 2
            while src[j] != '"':
                j += 1                                   # a Rust raw string: no escapes, ends at the next quote
            out.append(("re", src[i + 2:j]))
            i = j + 1
        elif c == '"':
            j, s = i + 1, ""
            while src[j] != '"':
                if src[j] == "\\":
                    j += 1
                s += src[j]
                j += 1
            out.append(("lit", s))
            i = j + 1
        elif c.isalpha() or c == "_":
            j = i
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            out.append(("id", src[i:j]))
            i = j
        elif src.startswith("=>", i) or src.startswith("::", i) or src.startswith("->", i):
            out.append(("p", src[i:i + 2]))
            i += 2
        else:
            out.append(("p", c))
            i += 1
    return out


class Grammar:
    def __init__(self):
        self.start = None
        self.prods = {}          # nonterminal -> [alternative]; alternative = [(capture name | None, symbol)]
        self.literals = []       # quoted terminals, in order of appearance
        self.regexes = []        # regex terminals, in order of appearance
        self._fresh = 0

    def terminal(self, kind, text):
        pool = self.literals if kind == "lit" else self.regexes
        if text not in pool:
            pool.append(text)
        return (kind, text)

    def fresh(self, base):
        self._fresh += 1
        return "%s#%d" % (base, self._fresh)


def read_grammar(path):
    src = open(path, encoding="utf-8").read()
    t = _grammar_tokens(src)
    g = Grammar()
    i = 0

    def at(k, v=None):
        return i < len(t) and t[i][0] == k and (v is None or t[i][1] == v)

    # preamble: `use ...;` lines and `grammar;`
    while at("id", "use"):
        while not at("p", ";"):
            i += 1
        i += 1
    assert at("id", "grammar"), "expected `grammar;`"
    i += 1
    assert at("p", ";")
    i += 1

    def symbols(closers):
        """a sequence of symbols up to (not including) one of the closing punctuation marks"""
        nonlocal i
        seq = []
        while not (t[i][0] == "p" and t[i][1] in closers):
            cap = None
            if at("p", "<"):
                i += 1
                if t[i][0] == "id" and t[i + 1] == ("p", ":"):
                    cap = t[i][1]
                    i += 2
                sym = atom()
                assert at("p", ">"), "unclosed capture"
                i += 1
                if cap is None:
                    cap = "_%d" % len(seq)
            else:
                sym = atom()
            seq.append((cap, sym))
        return seq

    def atom():
        nonlocal i
        k, v = t[i]
        if k == "id":
            i += 1
            sym = ("nt", v)
        elif k in ("lit", "re"):
            i += 1
            sym = g.terminal(k, v)
        elif (k, v) == ("p", "("):
            i += 1
            inner = symbols({")"})
            i += 1
            name = g.fresh("group")
            g.prods[name] = [inner]
            sym = ("nt", name)
        else:
            raise AssertionError("unexpected %r in an alternative" % (t[i],))
        while t[i][0] == "p" and t[i][1] in "*+?":
            op = t[i][1]
            i += 1
            name = g.fresh({"*": "star", "+": "plus", "?": "opt"}[op])
            if op == "*":
                g.prods[name] = [[], [("list", ("nt", name)), ("item", sym)]]
            elif op == "+":
                g.prods[name] = [[("item", sym)], [("list", ("nt", name)), ("item", sym)]]
            else:
                g.prods[name] = [[], [("item", sym)]]
            sym = ("nt", name)
        return sym

    def skip_action():
        """Rust code after `=>`: up to the `,` or `}` that ends the alternative (nesting depth 0)"""
        nonlocal i
        depth = 0
        while True:
            k, v = t[i]
            if k == "p" and v in "([{":
                depth += 1
            elif k == "p" and v in ")]}":
                if depth == 0:
                    return
                depth -= 1
            elif k == "p" and v in ",;" and depth == 0:
                return
            i += 1

    def alternative(closers):
        nonlocal i
        seq = symbols(closers | {"=>"})
        if at("p", "=>"):
            i += 1
            skip_action()
        return seq

    while i < len(t):
        is_pub = at("id", "pub")
        if is_pub:
            i += 1
        assert t[i][0] == "id", t[i]
        name = t[i][1]
        i += 1
        assert at("p", ":")
        i += 1
        depth = 0
        while not (at("p", "=") and depth == 0):          # the Rust type of the nonterminal
            if t[i] == ("p", "<") or t[i] == ("p", "("):
                depth += 1
            elif t[i] == ("p", ">") or t[i] == ("p", ")"):
                depth -= 1
            i += 1
        i += 1
        alts = []
        if at("p", "{"):
            i += 1
            while not at("p", "}"):
                alts.append(alternative({",", "}"}))
                if at("p", ","):
                    i += 1
            i += 1
        else:
            alts.append(alternative({";"}))
        assert at("p", ";"), "production %s is not closed" % name
        i += 1
        g.prods[name] = alts
        if is_pub:
            assert g.start is None, "two pub nonterminals"
            g.start = name
    assert g.start, "no pub nonterminal"
    return g


# --------------------------------------------------------------------------------------------------------------
# 2. LALRPOP's lexer over the file's terminals
# --------------------------------------------------------------------------------------------------------------
class LexError(Exception):
    pass


class ParseError(Exception):
    pass


def make_lexer(g):
    rx = [(("re", p), re.compile(p)) for p in g.regexes]
    lits = [("lit", s) for s in g.literals]

    def lex(text):
        toks, i, n = [], 0, len(text)
        ws = re.compile(r"\s+")
        while i < n:
            m = ws.match(text, i)
            if m:
                i = m.end()
                continue
            best, kind = 0, None
            for term in lits:
                if text.startswith(term[1], i) and len(term[1]) > best:
                    best, kind = len(term[1]), term
            for term, r in rx:
                m = r.match(text, i)
                if m and m.end() - i > best:              # strictly longer: a literal wins a tie
                    best, kind = m.end() - i, term
            if kind is None:
                raise LexError("invalid token at offset %d" % i)
            toks.append((kind, text[i:i + best]))
            i += best
        return toks

    return lex


# --------------------------------------------------------------------------------------------------------------
# 3. Earley recogniser with derivation
# --------------------------------------------------------------------------------------------------------------
def make_parser(g):
    rules = []                                    # (lhs, alt index, [(capture, symbol)])
    by_lhs = {}
    for lhs, alts in g.prods.items():
        for k, alt in enumerate(alts):
            by_lhs.setdefault(lhs, []).append(len(rules))
            rules.append((lhs, k, alt))

    def parse(toks):
        n = len(toks)
        chart = [dict() for _ in range(n + 1)]    # item (rule, dot, origin) -> derivation: list of children so far
        agenda = [[] for _ in range(n + 1)]

        def add(pos, item, deriv):
            if item not in chart[pos]:
                chart[pos][item] = deriv
                agenda[pos].append(item)

        for r in by_lhs[g.start]:
            add(0, (r, 0, 0), ())
        for pos in range(n + 1):
            done_here = {}                        # nonterminals completed with origin == pos (empty derivations)
            while agenda[pos]:
                item = agenda[pos].pop()
                r, dot, origin = item
                lhs, _k, alt = rules[r]
                deriv = chart[pos][item]
                if dot == len(alt):               # complete
                    node = ("node", lhs, rules[r][1], deriv)
                    if origin == pos:
                        done_here[lhs] = node
                    for other, d2 in list(chart[origin].items()):
                        r2, dot2, o2 = other
                        alt2 = rules[r2][2]
                        if dot2 < len(alt2) and alt2[dot2][1] == ("nt", lhs):
                            add(pos, (r2, dot2 + 1, o2), d2 + ((alt2[dot2][0], node),))
                    continue
                cap, sym = alt[dot]
                if sym[0] == "nt":
                    for r2 in by_lhs[sym[1]]:
                        add(pos, (r2, 0, pos), ())
                    if sym[1] in done_here:       # a nullable nonterminal completed at this position earlier
                        add(pos, (r, dot + 1, origin), deriv + ((cap, done_here[sym[1]]),))
                elif pos < n and toks[pos][0] == sym:
                    add(pos + 1, (r, dot + 1, origin), deriv + ((cap, ("tok", toks[pos][1])),))
        for r in by_lhs[g.start]:
            item = (r, len(rules[r][2]), 0)
            if item in chart[n]:
                return ("node", g.start, rules[r][1], chart[n][item])
        raise ParseError("no derivation")

    return parse


# --------------------------------------------------------------------------------------------------------------
# 4. the value of a derivation (the file's action bodies, lines 8-81)
# --------------------------------------------------------------------------------------------------------------
EXPECTED_CAPTURES = {            # nonterminal -> per alternative, the captures its action body names
    "ExprList": [["pipeline_expr"], ["pipeline_exprs", "pipeline_expr"]],
    "Expr": [["pipeline"], ["graph"], [], []],
    "GraphExpr": [["pipeline0", "pipeline1"], ["graph", "pipeline"]],
    "GraphMember": [["pipeline"], ["pipeline", "descriptor"]],
    "PipelineField": [["name", "pipeline_type", "parameters"]],
    "PipelineParams": [["map"], []],
    "ParamFieldList": [["head", "tail"]],
    "ParamField": [["key", "value"]],
    "BoolLiteral": [[], []],
    "ParamValueOp": [["val"], ["val"], ["val"]],
    "Str": [["s"]],
}


def check_shape(g):
    for nt, alts in EXPECTED_CAPTURES.items():
        assert nt in g.prods, "the grammar has no nonterminal %s any more" % nt
        got = [[c for c, _s in alt if c is not None] for alt in g.prods[nt]]
        assert got == alts, "captures of %s changed: %r (this script follows %r)" % (nt, got, alts)
    extra = [nt for nt in g.prods if "#" not in nt and nt not in EXPECTED_CAPTURES]
    assert not extra, "nonterminals this script has no action for: %r" % extra


def value(node):
    if node[0] == "tok":
        return node[1]
    _n, nt, k, children = node
    c = dict(children)
    v = {name: value(sub) for name, sub in children if name is not None}
    if "#" in nt:                                           # macro expansions: lists / options / groups
        if nt.startswith("group"):
            return [value(sub) for _name, sub in children]
        if not children:
            return [] if not nt.startswith("opt") else None
        if "list" in c:
            return v["list"] + [v["item"]]
        return [v["item"]] if not nt.startswith("opt") else v["item"]
    if nt == "ExprList":
        return [v["pipeline_expr"]] if k == 0 else v["pipeline_exprs"] + [v["pipeline_expr"]]
    if nt == "Expr":
        if k == 0:
            return ["pipeline"] + v["pipeline"]
        if k == 1:
            return ["graph", v["graph"]]
        return ["comment", "".join(value(sub) for _name, sub in children)]       # Expr::Ignore(0): the text is dropped; kept here as evidence of the token
    if nt == "GraphExpr":
        return [v["pipeline0"], v["pipeline1"]] if k == 0 else v["graph"] + [v["pipeline"]]
    if nt == "GraphMember":
        return [v["pipeline"], None] if k == 0 else [v["pipeline"], v["descriptor"]]
    if nt == "PipelineField":
        return [v["name"], v["pipeline_type"], v["parameters"]]
    if nt == "PipelineParams":
        return v["map"] if k == 0 else []
    if nt == "ParamFieldList":
        return [v["head"]] + [pair[-1] for pair in v["tail"]]                   # tail: ("," ParamField)* -> the fields, in source order
    if nt == "ParamField":
        return [v["key"], v["value"]]
    if nt == "BoolLiteral":
        return value(children[0][1])
    if nt == "ParamValueOp":
        return v["val"]
    if nt == "Str":
        return v["s"]
    raise AssertionError(nt)


# --------------------------------------------------------------------------------------------------------------
# 5. texts
# --------------------------------------------------------------------------------------------------------------
TRAPS = [
    "input -> passthrough -> output", "input->passthrough->output", "input -> aa->bb -> output", "aa -> bb", "aa->bb", "a -> bb", "aa -> b",
    "aa: bb {}", "aa: bb { }", "aa: bb {  }", "aa : bb{}", "aa:bb{}", "aa: bb {sigma: 1}", "aa: bb { sigma: 1.5, amount: 2 }", "aa: bb { sigma: 1.5, }",
    "aa: bb { sigma: -1.5 }", "aa: bb { sigma: -1 }", "aa: bb { sigma: .5 }", "aa: bb { sigma: 5. }", "aa: bb { sigma: 1e3 }", "aa: bb { sigma: true, tt: false }",
    "aa: bb { sigma: truex }", "aa: bb { true: 1 }", "aa: bb { sigma: 1, sigma: 2 }", "aa: bb { sigma: 01 }", "aa: bb { sigma: 1.50 }", "aa: bb { sigma 1 }",
    "aa: bb { sigma: 1 amount: 2 }", "aa: bb {} cc: dd {}", "aa: bb {}\ncc -> dd", "aa: bb", "aa:", ":", "->", "aa ->", "-> aa", "aa -> -> bb", "aa -> bb ->",
    "aa:xx -> bb:yy -> cc", "aa:xx:yy -> bb", "aa -> bb:yy:zz", "aa: bb: cc {}", "input -> aa:image -> output", "// only a comment", "// c\n", "// c\r\n// d\r\naa -> bb",
    "/* c */", "/* c */ aa -> bb", "aa -> bb /* c */", "aa /* c */ -> bb", "/* a */ aa -> bb /* b */", "/* a */ aa -> bb /* b */ cc -> dd", "/**/", "/***/", "/* * */", "/* / */",
    "/* unterminated", "unterminated */", "/* a /* b */ c */", "aa -> bb // tail", "aa -> bb // tail\ncc -> dd", "aa // c\n-> bb", "", " ", "\n\n", "\t", "aa", "aa bb", "aa, bb",
    "aa -> bb, cc", "{}", "{ }", "{", "}", "aa: bb {}}", "aa: bb {{}", "aa_b -> c-d", "a-b -> c_d", "aa- -> bb", "aa--bb -> cc", "-aa -> bb", "_a -> _b", "__ -> --", "a1 -> b2",
    "1a -> bb", "aa -> 12", "12 -> aa", "aa -> 1.5", "true -> false", "truee -> falsee", "tru -> fals", "aa: true {}", "true: aa {}", "aa: bb { cc: dd }", "aa: bb { cc: 1 } -> dd",
    "aa -> bbé", "é", "aa → bb", "aa -> bb\x0b", "aa\x0c->\x0cbb", "aa -> bb\x00", "aa: bb { sigma: 1 }\n\n\ninput -> aa -> output\n",
    "input -> blur -> grade -> sharp -> output\nblur:  gaussian5    { sigma: 1.0 }\ngrade: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }\nsharp: sharpen      { amount: 0.5 }\n",
    "input -> blur -> mixer:input_image0\ninput -> sharp -> mixer:input_image1\nmixer -> output\nblur: gaussian5 { sigma: 1.5 }\nsharp: sharpen { amount: 0.75 }\nmixer: combination { mix: 0.25 }",
]


def generate_texts(count, seed=20261005):
    rng = np.random.RandomState(seed)
    vocab = ["input", "output", "aa", "bb", "cc", "blur", "a", "x1", "gaussian5", "sharpen", "colour_grade", "->", "->", "->", ":", "{", "}", "{}", ",",
             "sigma", "amount", "1.5", "2", "-0.5", "-3", "true", "false", "1e3", "// note\n", "/* c */", "\n", "\n", " ", "  ", "\t", "image", "input_image",
             "_x", "a-b", "-", ">", "*/", "/*", "é", "0", ".5", "5.", "\r\n", "/", "*", "//", "{}", "{ }", "truefalse", "true1", "-", "--", "->>", "=>", ";"]
    idents = ["aa", "bb", "cc", "blur", "x1", "_x", "a-b", "gaussian5", "sharpen", "grade", "input_image", "image", "n0", "colour-grade", "A_b", "zz9", "tr", "truth"]
    values = ["1.5", "2", "-0.5", "true", "false", "0", "10.25", "-3", "1e3", ".5", "007", "3.", "-0.0", "123456789", "0.000001"]

    def pick(xs):
        return xs[rng.randint(len(xs))]

    def well_formed():
        exprs = []
        for _ in range(rng.randint(1, 5)):
            kind = rng.randint(4)
            if kind == 0:
                mid = [pick(idents) + (":" + pick(idents) if rng.rand() < 0.25 else "") for _ in range(rng.randint(1, 4))]
                arrow = pick([" -> ", " -> ", " ->\n  ", "  ->  ", " -> /* x */ "])
                exprs.append(pick(["input", pick(idents)]) + arrow + arrow.join(mid) + pick([" -> output", " -> output", ""]))
            elif kind == 1:
                kv = ["%s%s %s" % (pick(idents), pick([":", ": ", " : "]), pick(values)) for _ in range(rng.randint(0, 4))]
                exprs.append("%s: %s %s" % (pick(idents), pick(idents), "{ " + pick([", ", ",", " , ", ",\n  "]).join(kv) + " }" if kv else pick(["{}", "{ }", "{}"])))
            elif kind == 2:
                exprs.append(pick(["// a note", "/* block */", "/* two\nlines */", "// x -> y", "/* a: b {} */", "/** doc **/"]))
            else:
                exprs.append("input -> aa -> bb -> output")
        return pick(["\n", "\n\n", " \n", " ", "\r\n"]).join(exprs)

    texts = list(TRAPS)
    i = 0
    while len(texts) < count:
        if i % 4 == 3:
            n = rng.randint(1, 14)
            text = "".join(pick(vocab) + ("" if rng.rand() < 0.3 else " ") for _ in range(n))
        else:
            text = well_formed()
            for _ in range(rng.randint(0, 4) if rng.rand() < 0.55 else 0):
                at = rng.randint(len(text) + 1)
                text = text[:at] + pick(vocab) + text[at + rng.randint(0, 3):]
        texts.append(text)
        i += 1
    return texts


def classify(texts, grammar_path=DEFAULT_GRAMMAR):
    g = read_grammar(grammar_path)
    check_shape(g)
    lex, parse = make_lexer(g), make_parser(g)
    out = []
    for text in texts:
        rec = {"t": text}
        try:
            toks = lex(text)
            rec["ok"] = True
            rec["exprs"] = value(parse(toks))
        except LexError:
            rec["ok"], rec["why"] = False, "lex"
        except ParseError:
            rec["ok"], rec["why"] = False, "parse"
        out.append(rec)
    meta = {"grammar_sha256": hashlib.sha256(open(grammar_path, "rb").read()).hexdigest(), "start": g.start,
            "literals": g.literals, "regexes": g.regexes,
            "productions": {nt: len(alts) for nt, alts in g.prods.items() if "#" not in nt}}
    return meta, out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_GRAMMAR
    meta, recs = classify(generate_texts(2400), path)
    doc = {"meta": meta, "cases": recs}
    raw = json.dumps(doc, ensure_ascii=True, separators=(",", ":"), sort_keys=True).encode("ascii")
    with open(OUT, "wb") as f:
        with gzip.GzipFile(filename="", mode="wb", fileobj=f, mtime=0) as z:      # mtime 0: the same bytes every time
            z.write(raw)
    ok = sum(r["ok"] for r in recs)
    print("%d texts: %d accepted, %d invalid token, %d no derivation -> %s (%d bytes)" % (
        len(recs), ok, sum(r.get("why") == "lex" for r in recs), sum(r.get("why") == "parse" for r in recs), OUT, os.path.getsize(OUT)))
    print("terminals read from the file: literals %r, regexes %r" % (meta["literals"], meta["regexes"]))


if __name__ == "__main__":
    main()
