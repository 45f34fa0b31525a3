// rf_glsl.cpp -- see rf_glsl.h.  Host only, no dependency beyond the standard library.
#include "rf_glsl.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>

namespace rf {

namespace {

enum TokKind { T_ID, T_NUM, T_PUNCT, T_PP, T_END };
struct Tok {
    TokKind k = T_END;
    std::string s;       // the text that is emitted (rewritten in place)
    std::string ws;      // what stood in front of it: white space; comments are reduced to their line breaks
    std::string pre, post;   // inserted around it (emitted even when the token itself is dropped)
    std::string outer;       // inserted in front of `pre`: the opening of a call that encloses what `pre` opens (rfg_xor( around rfg_eq( )
    int line = 1;
    bool drop = false;   // emit the white space only
};

struct Fail {
    int line;
    std::string msg;
};

bool id_start(char c) { return std::isalpha((unsigned char)c) || c == '_'; }
bool id_char(char c) { return std::isalnum((unsigned char)c) || c == '_'; }

// ---- lexer ------------------------------------------------------------------------------------------------------------------
std::vector<Tok> lex(const std::string& t)
{
    std::vector<Tok> out;
    size_t i = 0;
    int line = 1;
    bool line_start = true;
    std::string ws;
    while (i < t.size()) {
        const char c = t[i];
        if (c == '\n') { ws += c; ++line; ++i; line_start = true; continue; }
        if (c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v') { ws += c; ++i; continue; }
        if (c == '\\' && i + 1 < t.size() && t[i + 1] == '\n') { ws += '\n'; ++line; i += 2; continue; }      // a spliced line outside a directive
        if (t.compare(i, 2, "//") == 0) { while (i < t.size() && t[i] != '\n') ++i; continue; }
        if (t.compare(i, 2, "/*") == 0) {
            size_t e = t.find("*/", i + 2);
            if (e == std::string::npos) e = t.size() - 2;
            for (size_t k = i; k < e + 2 && k < t.size(); ++k)
                if (t[k] == '\n') { ws += '\n'; ++line; }
            ws += ' ';
            i = e + 2;
            continue;
        }
        Tok k;
        k.ws = ws;
        k.line = line;
        ws.clear();
        if (c == '#' && line_start) {
            // a directive: to the end of the line, spliced lines included; comments inside removed
            std::string d;
            while (i < t.size() && t[i] != '\n') {
                if (t[i] == '\\' && i + 1 < t.size() && t[i + 1] == '\n') { d += ' '; ws += '\n'; ++line; i += 2; continue; }
                if (t.compare(i, 2, "//") == 0) { while (i < t.size() && t[i] != '\n') ++i; break; }
                if (t.compare(i, 2, "/*") == 0) {
                    size_t e = t.find("*/", i + 2);
                    if (e == std::string::npos) e = t.size() - 2;
                    for (size_t q = i; q < e + 2 && q < t.size(); ++q)
                        if (t[q] == '\n') { ws += '\n'; ++line; }
                    d += ' ';
                    i = e + 2;
                    continue;
                }
                d += t[i++];
            }
            k.k = T_PP;
            k.s = d;
            out.push_back(k);
            continue;      // ws now holds the line breaks the directive swallowed
        }
        line_start = false;
        if (id_start(c)) {
            size_t e = i;
            while (e < t.size() && id_char(t[e])) ++e;
            k.k = T_ID;
            k.s = t.substr(i, e - i);
            i = e;
        } else if (std::isdigit((unsigned char)c) || (c == '.' && i + 1 < t.size() && std::isdigit((unsigned char)t[i + 1]))) {
            size_t e = i;
            bool is_float = false, hex = false;
            if (c == '0' && e + 1 < t.size() && (t[e + 1] == 'x' || t[e + 1] == 'X')) {
                hex = true;
                e += 2;
                while (e < t.size() && std::isxdigit((unsigned char)t[e])) ++e;
            } else {
                while (e < t.size() && std::isdigit((unsigned char)t[e])) ++e;
                if (e < t.size() && t[e] == '.') { is_float = true; ++e; while (e < t.size() && std::isdigit((unsigned char)t[e])) ++e; }
                if (e < t.size() && (t[e] == 'e' || t[e] == 'E')) {
                    size_t q = e + 1;
                    if (q < t.size() && (t[q] == '+' || t[q] == '-')) ++q;
                    if (q < t.size() && std::isdigit((unsigned char)t[q])) { is_float = true; e = q; while (e < t.size() && std::isdigit((unsigned char)t[e])) ++e; }
                }
            }
            std::string num = t.substr(i, e - i), suffix;
            while (e < t.size() && id_char(t[e])) suffix += t[e++];
            i = e;
            k.k = T_NUM;
            if (!hex && (is_float || suffix == "f" || suffix == "F" || suffix == "lf" || suffix == "LF")) {
                // GLSL: a literal with a point or an exponent is a float (C++: a double, and the arithmetic around it with it)
                if (suffix == "lf" || suffix == "LF") k.s = num + (is_float ? "" : ".0");
                else k.s = num + (is_float ? "f" : ".0f");
            } else {
                k.s = num + suffix;      // integers: 1, 1u, 0x1Fu mean the same in C++
            }
        } else {
            k.k = T_PUNCT;
            k.s = std::string(1, c);
            ++i;
        }
        out.push_back(k);
    }
    // matNxN is another spelling of matN (GLSL 4.50 4.1.6); the non-square matrices stay outside the subset
    for (size_t k = 0; k < out.size(); ++k)
        if (out[k].k == T_ID && !(k > 0 && out[k - 1].k == T_PUNCT && out[k - 1].s == ".")) {
            if (out[k].s == "mat2x2") out[k].s = "mat2";
            else if (out[k].s == "mat3x3") out[k].s = "mat3";
            else if (out[k].s == "mat4x4") out[k].s = "mat4";
        }
    Tok e;
    e.k = T_END;
    e.ws = ws;
    e.line = line;
    out.push_back(e);
    return out;
}

bool is(const Tok& t, const char* s) { return (t.k == T_ID || t.k == T_PUNCT) && t.s == s; }

const std::set<std::string>& vector_types()
{
    static const std::set<std::string> s = {"vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4", "bvec2", "bvec3", "bvec4", "mat2", "mat3", "mat4"};
    return s;
}
const std::set<std::string>& scalar_types()
{
    static const std::set<std::string> s = {"float", "int", "uint", "bool"};
    return s;
}
const std::set<std::string>& dropped_qualifiers()
{
    static const std::set<std::string> s = {"precise", "highp", "mediump", "lowp", "invariant"};
    return s;
}
// words that are names a GLSL file may give its things but that C++ keeps for itself (GLSL reserves most of C++'s others itself): renamed
const std::set<std::string>& cpp_only_keywords()
{
    static const std::set<std::string> s = {"new", "delete", "char", "auto", "register", "signed", "private", "protected", "friend", "operator", "virtual", "try", "catch", "throw",
                                            "mutable", "explicit", "typename", "and", "or", "xor", "bitand", "bitor", "compl", "and_eq", "or_eq", "xor_eq", "not_eq", "asm", "wchar_t",
                                            "constexpr", "decltype", "nullptr", "alignas", "alignof", "static_assert", "thread_local", "typeid", "const_cast", "dynamic_cast",
                                            "reinterpret_cast", "static_cast", "noexcept", "export", "concept", "requires", "char16_t", "char32_t", "final", "override"};
    return s;
}
const std::set<std::string>& atomic_functions()
{
    static const std::set<std::string> s = {"atomicAdd", "atomicMin", "atomicMax", "atomicAnd", "atomicOr", "atomicXor", "atomicExchange", "atomicCompSwap"};
    return s;
}
const std::set<std::string>& unsupported_words()
{
    static const std::set<std::string> s = {"double",      "dvec2",       "dvec3",       "dvec4",      "dmat2",       "dmat3",          "dmat4",        "sampler1D",
                                            "sampler3D",   "samplerCube", "sampler2DArray", "sampler2DShadow", "textureGather", "image1D",        "image3D",      "imageCube",
                                            "image2DArray", "iimage2D",   "uimage2D",    "imageAtomicAdd", "imageAtomicMin", "imageAtomicMax", "imageAtomicExchange", "imageAtomicCompSwap",
                                            "atomic_uint", "atomicCounter", "atomicCounterIncrement", "atomicCounterDecrement", "subroutine", "mat2x3",     "mat2x4",      "mat3x2",         "mat3x4",
                                            "mat4x2",      "mat4x3",      "push_constant"};
    return s;
}

struct Ctx {
    std::string type;
    std::set<std::string> structs, struct_members, ssbo_scalars, ssbo_instances, macros;
};

size_t match(const std::vector<Tok>& v, size_t open, size_t end)
{
    const std::string o = v[open].s, c = o == "(" ? ")" : (o == "[" ? "]" : "}");
    int depth = 0;
    for (size_t i = open; i < end; ++i) {
        if (v[i].k != T_PUNCT) continue;
        if (v[i].s == o) ++depth;
        else if (v[i].s == c && --depth == 0) return i;
    }
    throw Fail{v[open].line, "unbalanced `" + o + "`"};
}

size_t next_live(const std::vector<Tok>& v, size_t i, size_t end)
{
    while (i < end && v[i].drop) ++i;
    return i;
}

// `a == b` / `a != b`: GLSL compares vectors, matrices and structs as wholes and gives ONE bool; a C++ comparison of clang vectors
// gives a vector.  Every equality operator becomes a call -- rfg_eq(a, b) / rfg_ne(a, b), rf_glsl_dev.h: plain == for scalars, "all
// components equal" for vectors and matrices -- around its operands, found by precedence: everything that binds tighter than ==
// (postfix, unary, * / %, + -, << >>, < > <= >=) belongs to the operand; the scan stops at the enclosing bracket, at `,` `;` `?` `:`,
// at & ^ | (and so && ||), at an assignment and at another equality operator.
void rewrite_equality(std::vector<Tok>& v, size_t b, size_t e)
{
    auto live = [&](size_t i) { return !v[i].drop; };
    auto punct = [&](size_t i, char c) { return live(i) && v[i].k == T_PUNCT && v[i].s.size() == 1 && v[i].s[0] == c; };
    auto glued = [&](size_t i) { return v[i].ws.empty() && v[i].pre.empty(); };      // directly behind the token in front of it
    for (size_t i = b; i + 1 < e; ++i) {
        if (!(punct(i, '=') || punct(i, '!')) || !punct(i + 1, '=') || !glued(i + 1)) continue;
        if (punct(i, '=') && i > b && glued(i) && v[i - 1].k == T_PUNCT && live(i - 1) && std::string("<>!=+-*/%&|^").find(v[i - 1].s[0]) != std::string::npos) continue;   // <=, >=, +=, ... or the tail of ==
        const bool ne = v[i].s == "!";
        // left operand: back to the first token that binds looser than ==
        size_t lb = i;
        int depth = 0;
        while (lb > b) {
            const size_t k = lb - 1;
            if (!live(k)) { --lb; continue; }
            if (v[k].k == T_PUNCT) {
                const char c = v[k].s[0];
                if (c == ')' || c == ']' || c == '}') ++depth;
                else if (c == '(' || c == '[' || c == '{') { if (depth == 0) break; --depth; }
                else if (depth == 0) {
                    if (c == ',' || c == ';' || c == '?' || c == ':' || c == '&' || c == '|' || c == '^') break;
                    if (c == '=') {
                        const bool relational = k > b && live(k - 1) && v[k - 1].k == T_PUNCT && (v[k - 1].s == "<" || v[k - 1].s == ">") && glued(k);
                        if (!relational) break;      // an assignment, a compound assignment, or another == / != (left-associative: it is the left operand's end)
                    }
                }
            } else if (depth == 0 && v[k].k == T_ID && (v[k].s == "return" || v[k].s == "case")) break;
            --lb;
        }
        while (lb < i && !live(lb)) ++lb;
        // right operand
        size_t re = i + 2;
        depth = 0;
        while (re < e) {
            if (live(re) && v[re].k == T_PUNCT) {
                const char c = v[re].s[0];
                if (c == '(' || c == '[' || c == '{') ++depth;
                else if (c == ')' || c == ']' || c == '}') { if (depth == 0) break; --depth; }
                else if (depth == 0) {
                    if (c == ',' || c == ';' || c == '?' || c == ':' || c == '&' || c == '|' || c == '^') break;
                    if ((c == '=' || c == '!') && re + 1 < e && punct(re + 1, '=') && glued(re + 1)) break;      // the next equality operator
                    if (c == '=' && !(re > 0 && live(re - 1) && (v[re - 1].s == "<" || v[re - 1].s == ">") && glued(re))) break;
                }
            }
            ++re;
        }
        if (lb >= i || re <= i + 2) throw Fail{v[i].line, "cannot find the operands of this comparison"};
        size_t last = re - 1;
        while (last > i + 1 && !live(last)) --last;
        v[lb].pre = (ne ? "rfg_ne(" : "rfg_eq(") + v[lb].pre;
        v[i].s = ",";
        v[i + 1].drop = true;
        v[last].post += ")";
    }
}

// a ^^ b (GLSL's logical exclusive or; C++ has no such operator) -> rfg_xor(a, b).  It binds tighter than || only: the operands reach
// back / forward to the enclosing bracket, `,` `;` `?` `:`, an assignment, `||` or another ^^ (left-associative: the LAST one of a chain is
// rewritten first, its left operand the chain in front of it).  Runs before rewrite_equality, whose scans stop at the comma written here.
void rewrite_logical_xor(std::vector<Tok>& v, size_t b, size_t e)
{
    auto live = [&](size_t i) { return !v[i].drop; };
    auto punct = [&](size_t i, char c) { return i < e && live(i) && v[i].k == T_PUNCT && v[i].s.size() == 1 && v[i].s[0] == c; };
    auto glued = [&](size_t i) { return v[i].ws.empty() && v[i].pre.empty() && v[i].outer.empty(); };
    for (size_t i = e; i-- > b + 1;) {
        if (!punct(i - 1, '^') || !punct(i, '^') || !glued(i)) continue;
        const size_t op = i - 1;
        size_t lb = op;
        int depth = 0;
        while (lb > b) {
            const size_t k = lb - 1;
            if (!live(k)) { --lb; continue; }
            if (v[k].k == T_PUNCT) {
                const char c = v[k].s[0];
                if (c == ')' || c == ']' || c == '}') ++depth;
                else if (c == '(' || c == '[' || c == '{') { if (depth == 0) break; --depth; }
                else if (depth == 0) {
                    if (c == ',' || c == ';' || c == '?' || c == ':') break;
                    if (c == '|' && k > b && punct(k - 1, '|') && glued(k)) break;      // ||
                    if (c == '=') {
                        const bool compare = (k > b && live(k - 1) && v[k - 1].k == T_PUNCT && glued(k) && std::string("=!<>").find(v[k - 1].s[0]) != std::string::npos) ||
                                             (punct(k + 1, '=') && glued(k + 1));
                        if (!compare) break;      // an assignment
                    }
                }
            } else if (depth == 0 && v[k].k == T_ID && (v[k].s == "return" || v[k].s == "case")) break;
            --lb;
        }
        while (lb < op && !live(lb)) ++lb;
        size_t re = i + 1;
        depth = 0;
        while (re < e) {
            if (live(re) && v[re].k == T_PUNCT) {
                const char c = v[re].s[0];
                if (c == '(' || c == '[' || c == '{') ++depth;
                else if (c == ')' || c == ']' || c == '}') { if (depth == 0) break; --depth; }
                else if (depth == 0) {
                    if (c == ',' || c == ';' || c == '?' || c == ':') break;
                    if (c == '|' && punct(re + 1, '|') && glued(re + 1)) break;
                    if (c == '^' && punct(re + 1, '^') && glued(re + 1)) break;
                }
            }
            ++re;
        }
        if (lb >= op || re <= i + 1) throw Fail{v[op].line, "cannot find the operands of this ^^"};
        size_t last = re - 1;
        while (last > i && !live(last)) --last;
        v[lb].outer += "rfg_xor(";
        v[op].s = ",";
        v[i].drop = true;
        v[last].post += ")";
        i = op;      // (the loop's i-- moves in front of the operator)
    }
}

// the rewrites that need no knowledge of where a token stands: [b, e)
void rewrite(std::vector<Tok>& v, size_t b, size_t e, const Ctx& cx)
{
    for (size_t i = b; i < e; ++i) {
        Tok& t = v[i];
        if (t.drop || t.k != T_ID) continue;
        if (cpp_only_keywords().count(t.s)) t.s += "_rfg";      // (also behind a dot: a struct member of that name)
        const size_t n = next_live(v, i + 1, e);
        const bool call = n < e && is(v[n], "(");
        const bool after_dot = i > b && is(v[i - 1], ".");
        if (after_dot && t.s == "length" && call && i >= b + 2 && v[i - 2].k == T_ID && !(i >= b + 3 && is(v[i - 3], "."))) {
            // name.length(): the number of elements of an array (of components of a vector) -- a constant either way
            const size_t close = next_live(v, n + 1, e);
            if (close < e && is(v[close], ")")) {
                v[i - 2].pre += "rfg_length(";
                v[i - 1].drop = true;      // the dot
                t.drop = true;
                v[n].drop = true;
                continue;                  // the closing parenthesis stays: rfg_length(name)
            }
        }
        if (after_dot && t.s == "length" && call && i >= b + 2 && is(v[i - 2], "]")) {
            // name[i].length() (an array of arrays): back over the subscripts to the name
            const size_t close = next_live(v, n + 1, e);
            size_t a = i - 2;
            bool ok = close < e && is(v[close], ")");
            while (ok && is(v[a], "]")) {
                int depth = 0;
                size_t k = a;
                for (;; --k) {
                    if (is(v[k], "]")) ++depth;
                    else if (is(v[k], "[") && --depth == 0) break;
                    if (k == b) { ok = false; break; }
                }
                if (!ok || k == b) { ok = false; break; }
                a = k - 1;
            }
            if (ok && v[a].k == T_ID && !(a > b && is(v[a - 1], "."))) {
                v[a].pre += "rfg_length(";
                v[i - 1].drop = true;
                t.drop = true;
                v[n].drop = true;
                continue;
            }
        }
        if (after_dot) {
            // a swizzle spelled with texture coordinates: clang's vectors know xyzw and rgba
            bool stpq = !t.s.empty() && t.s.size() <= 4 && !cx.struct_members.count(t.s);
            for (char c : t.s) stpq = stpq && (c == 's' || c == 't' || c == 'p' || c == 'q');
            if (stpq)
                for (char& c : t.s) c = c == 's' ? 'x' : (c == 't' ? 'y' : (c == 'p' ? 'z' : 'w'));
            continue;
        }
        if (unsupported_words().count(t.s)) throw Fail{t.line, "`" + t.s + "` is outside the GLSL subset this library translates (rf_glsl.h)"};
        if (dropped_qualifiers().count(t.s)) { t.drop = true; continue; }
        if (t.s == "precision") {      // `precision highp float;` inside a function: a statement without meaning here
            size_t q = i;
            while (q < e && !is(v[q], ";")) ++q;
            if (q < e) { for (size_t k = i; k <= q; ++k) v[k].drop = true; i = q; continue; }
        }
        const bool basic = vector_types().count(t.s) || scalar_types().count(t.s);
        const bool is_type = basic || cx.structs.count(t.s);
        // Arrays are VALUES in GLSL (assigned, returned, passed and compared as wholes): a sized array type T[n] becomes rfg_arr<T, (n)>
        // (rf_glsl_dev.h: a struct around the C array).  Subscripts [first, q): their sizes, outermost first; false if one is `[]`.
        auto subscripts = [&](size_t first, size_t& q, std::vector<std::string>& dims) {
            bool sized = true;
            q = first;
            while (q < e && is(v[q], "[")) {
                const size_t rb = match(v, q, e);
                std::string d;
                for (size_t k = q + 1; k < rb; ++k)
                    if (!v[k].drop) d += (d.empty() ? "" : " ") + v[k].s;
                sized = sized && !d.empty();
                dims.push_back(d);
                q = next_live(v, rb + 1, e);
            }
            return sized;
        };
        auto array_type = [](const std::string& base, const std::vector<std::string>& dims) {
            std::string ty = base;
            for (size_t k = dims.size(); k-- > 0;) ty = "rfg_arr<" + ty + ", (" + dims[k] + ")>";
            return ty;
        };
        if (is_type && n < e && is(v[n], "[")) {
            size_t q = n;
            std::vector<std::string> dims;
            const bool sized = subscripts(n, q, dims);
            if (q < e && is(v[q], "(")) {
                // T[](a, b, c) / T[3](a, b, c): an array constructor -> {a, b, c} / rfg_arr<T, (3)>{a, b, c}
                const size_t close = match(v, q, e);
                for (size_t k = n; k < q; ++k) v[k].drop = true;
                if (sized) t.s = array_type(t.s, dims);
                else t.drop = true;
                v[q].s = "{";
                v[close].s = "}";
                continue;
            }
            if (q < e && v[q].k == T_ID) {
                // float[3] w  (GLSL's other spelling of  float w[3]), or the return type of a function
                if (sized) {
                    for (size_t k = n; k < q; ++k) v[k].drop = true;
                    t.s = array_type(t.s, dims);
                } else {      // float[] w = float[](...): a C array whose initialiser says its size (not a value: cannot be assigned as a whole)
                    std::string text;
                    for (size_t k = n; k < q; ++k) { if (!v[k].drop) text += v[k].s; v[k].drop = true; }
                    v[q].post = text + v[q].post;
                }
                continue;
            }
        }
        if (is_type && n < e && v[n].k == T_ID) {
            // T a[3], b, c[2][2] = ...;  -- each declarator with subscripts gets the array type; a list whose declarators differ is split
            const size_t after_name = next_live(v, n + 1, e);
            if (after_name < e && !is(v[after_name], "(")) {
                const std::string base = t.s;
                size_t prev = i;
                while (prev > b && v[prev - 1].drop) --prev;
                const bool is_const = prev > b && v[prev - 1].k == T_ID && v[prev - 1].s == "const";
                size_t name = n, sep = i;      // sep: the token that carries this declarator's type (the type token, then the commas)
                bool first = true;
                std::string first_ty;
                for (;;) {
                    size_t q = next_live(v, name + 1, e);
                    std::vector<std::string> dims;
                    const size_t d0 = q;
                    const bool sized = subscripts(d0, q, dims);
                    std::string ty = base;
                    if (!dims.empty() && sized) {
                        for (size_t k = d0; k < q; ++k) v[k].drop = true;
                        ty = array_type(base, dims);
                    }
                    if (first) v[sep].s = first_ty = ty;
                    else if (ty != first_ty) v[sep].s = std::string("; ") + (is_const ? "const " : "") + ty;
                    first = false;
                    // over the initialiser to the `,` that starts another declarator
                    int depth = 0;
                    while (q < e) {
                        if (!v[q].drop) {
                            if (is(v[q], "(") || is(v[q], "[") || is(v[q], "{")) ++depth;
                            else if (is(v[q], ")") || is(v[q], "]") || is(v[q], "}")) { if (depth-- == 0) break; }
                            else if (depth == 0 && (is(v[q], ",") || is(v[q], ";"))) break;
                        }
                        ++q;
                    }
                    if (q >= e || !is(v[q], ",")) break;
                    const size_t nn = next_live(v, q + 1, e);
                    if (nn >= e || v[nn].k != T_ID || vector_types().count(v[nn].s) || scalar_types().count(v[nn].s) || cx.structs.count(v[nn].s) || v[nn].s == "const" ||
                        v[nn].s == "in" || v[nn].s == "out" || v[nn].s == "inout" || dropped_qualifiers().count(v[nn].s))
                        break;      // a parameter list: the next parameter starts with its type
                    const size_t an = next_live(v, nn + 1, e);
                    if (an >= e || !(is(v[an], "[") || is(v[an], "=") || is(v[an], ",") || is(v[an], ";"))) break;
                    sep = q;
                    name = nn;
                }
                // (the declarators are visited again as identifiers by this loop: nothing of them is a type)
            }
        }
        if (call && vector_types().count(t.s)) { t.s = "mk_" + t.s; continue; }
        if (call && cx.structs.count(t.s)) {      // Light(a, b) -> Light{a, b}
            const size_t q = match(v, n, e);
            v[n].s = "{";
            v[q].s = "}";
            continue;
        }
        if (call && t.s == "not") { t.s = "rfg_not"; continue; }
        if (call && atomic_functions().count(t.s)) { t.s = "rfg_" + t.s; continue; }      // HIP's functions of these names take pointers
        if (cx.ssbo_scalars.count(t.s)) { t.s = "(*rfg_p_" + t.s + ")"; continue; }
        if (cx.ssbo_instances.count(t.s) && n < e && is(v[n], ".")) { v[n].s = "->"; continue; }
    }
    rewrite_logical_xor(v, b, e);
    rewrite_equality(v, b, e);
}

std::string emit(const std::vector<Tok>& v, size_t b, size_t e)
{
    std::string o;
    for (size_t i = b; i < e; ++i) {
        o += v[i].ws + v[i].outer + v[i].pre;
        if (!v[i].drop) o += v[i].s;
        o += v[i].post;
    }
    return o;
}

void blank(std::vector<Tok>& v, size_t b, size_t e)
{
    for (size_t i = b; i < e; ++i) v[i].drop = true;
}

// a directive: what is kept of it, translated ("" = dropped)
std::string directive(const Tok& t, Ctx& cx, GlslShader& sh)
{
    size_t i = 1;
    const std::string& d = t.s;
    while (i < d.size() && (d[i] == ' ' || d[i] == '\t')) ++i;
    size_t e = i;
    while (e < d.size() && id_char(d[e])) ++e;
    const std::string name = d.substr(i, e - i), rest = d.substr(e);
    if (name == "version" || name == "extension" || name == "line") return "";
    if (name == "pragma") {
        std::vector<Tok> r = lex(rest);
        if (r.size() >= 3 && is(r[0], "rf") && is(r[1], "radius")) {
            if (r[2].k != T_NUM || r.size() != 4) throw Fail{t.line, "#pragma rf radius N: N is an integer 0.." + std::to_string(1 << 12)};
            sh.radius = std::atoi(r[2].s.c_str());
            if (sh.radius < 0 || sh.radius > (1 << 12)) throw Fail{t.line, "#pragma rf radius " + r[2].s + " is out of range"};
        }
        return "";
    }
    if (name == "include") throw Fail{t.line, "#include is not available (the reference compiles the file alone, shader.rs:73-93)"};
    if (name == "define") {
        std::vector<Tok> r = lex(rest);
        if (r.empty() || r[0].k != T_ID) throw Fail{t.line, "#define without a name"};
        cx.macros.insert(r[0].s);
        size_t body = 1;
        if (r.size() > 2 && is(r[1], "(") && r[1].ws.empty()) body = match(r, 1, r.size()) + 1;      // a function-like macro: its parameter list stays
        rewrite(r, body, r.size() - 1, cx);
        return "#define" + emit(r, 0, r.size());
    }
    return d;      // #undef, #if, #ifdef, #ifndef, #else, #elif, #endif, #error: the C++ preprocessor reads them the same way
}

struct TypeInfo {
    char base;
    int comps, cols;
};
bool block_member_type(const std::string& s, TypeInfo& ti)
{
    static const std::map<std::string, TypeInfo> m = {
        {"float", {'f', 1, 1}}, {"int", {'i', 1, 1}},   {"uint", {'u', 1, 1}},  {"bool", {'b', 1, 1}},  {"vec2", {'f', 2, 1}},  {"vec3", {'f', 3, 1}},
        {"vec4", {'f', 4, 1}},  {"ivec2", {'i', 2, 1}}, {"ivec3", {'i', 3, 1}}, {"ivec4", {'i', 4, 1}}, {"uvec2", {'u', 2, 1}}, {"uvec3", {'u', 3, 1}},
        {"uvec4", {'u', 4, 1}}, {"mat2", {'f', 2, 2}},  {"mat3", {'f', 3, 3}},  {"mat4", {'f', 4, 4}}};
    auto it = m.find(s);
    if (it == m.end()) return false;
    ti = it->second;
    return true;
}
std::string cpp_type(const GlslMember& m)
{
    if (m.cols > 1) return "mat" + std::to_string(m.cols);
    const std::string p = m.base == 'f' ? "vec" : (m.base == 'i' ? "ivec" : "uvec");
    if (m.comps > 1) return p + std::to_string(m.comps);
    return m.base == 'f' ? "float" : (m.base == 'i' ? "int" : (m.base == 'u' ? "uint" : "bool"));
}
int cpp_elem_bytes(const GlslMember& m)      // sizeof of one array element as the C++ side lays it out
{
    const int v = m.comps == 1 ? (m.base == 'b' ? 1 : 4) : (m.comps == 2 ? 8 : 16);
    return v * m.cols;
}
int round_up(int x, int a) { return (x + a - 1) / a * a; }

// an integer constant expression of a declaration: literals, the macros and constants seen so far, + - * / ( )
struct ConstEval {
    const std::map<std::string, long>& names;
    const std::vector<Tok>& v;
    size_t i, e;
    long primary()
    {
        if (i >= e) throw Fail{v[e].line, "an array size is missing"};
        const Tok& t = v[i];
        if (is(t, "(")) { ++i; const long r = sum(); if (i >= e || !is(v[i], ")")) throw Fail{t.line, "unbalanced ( in an array size"}; ++i; return r; }
        if (is(t, "-")) { ++i; return -primary(); }
        if (is(t, "+")) { ++i; return primary(); }
        if (t.k == T_NUM) {
            const bool hex = t.s.size() > 1 && (t.s[1] == 'x' || t.s[1] == 'X');
            if (!hex && t.s.find_first_of(".eEfF") != std::string::npos) throw Fail{t.line, "an array size is an integer (`" + t.s + "` is not)"};
            ++i;
            return std::strtol(t.s.c_str(), nullptr, 0);
        }
        if (t.k == T_ID) {
            auto it = names.find(t.s);
            if (it == names.end()) throw Fail{t.line, "the array size `" + t.s + "` of a block member is not a constant this translator can evaluate (a literal, an object-like #define or a const int of literals)"};
            ++i;
            return it->second;
        }
        throw Fail{t.line, "cannot evaluate the array size"};
    }
    long product()
    {
        long r = primary();
        while (i < e && (is(v[i], "*") || is(v[i], "/") || is(v[i], "%"))) {
            const char op = v[i].s[0];
            ++i;
            const long b = primary();
            if (op != '*' && b == 0) throw Fail{v[i - 1].line, "division by zero in an array size"};
            r = op == '*' ? r * b : (op == '/' ? r / b : r % b);
        }
        return r;
    }
    long sum()
    {
        long r = product();
        while (i < e && (is(v[i], "+") || is(v[i], "-"))) {
            const char op = v[i].s[0];
            ++i;
            const long b = product();
            r = op == '+' ? r + b : r - b;
        }
        return r;
    }
};

struct Translator {
    const std::string& type;
    GlslShader& sh;
    Ctx cx;
    std::vector<Tok> v;
    std::string hoisted;                       // in front of the struct: directives, constants, shared variables
    std::string members, bind, bind_ubo;       // generated members of Shader, the body of rfg_bind; bind_ubo: its uniform-block part (shared with rfg_bind_point)
    std::map<std::string, long> int_consts;    // object-like macros and const ints whose value is a literal expression
    std::set<int> bindings;
    bool has_main = false;

    Translator(const std::string& ty, GlslShader& s) : type(ty), sh(s) { cx.type = ty; }

    [[noreturn]] void fail(int line, const std::string& m) { throw Fail{line, m}; }

    void claim_binding(int b, int line, const std::string& what)
    {
        if (b < 0) fail(line, what + " has no layout(binding = N)");
        if (!bindings.insert(b).second) fail(line, what + ": binding " + std::to_string(b) + " is used twice (one descriptor set, one resource per binding: shader.rs:122-128)");
    }

    // layout ( id [= value], ... ) starting at v[i] == "layout": returns the index behind ")"
    size_t parse_layout(size_t i, std::map<std::string, std::string>& q)
    {
        if (!is(v[i + 1], "(")) fail(v[i].line, "layout without (");
        const size_t e = match(v, i + 1, v.size());
        for (size_t k = i + 2; k < e;) {
            if (v[k].k != T_ID) fail(v[k].line, "cannot read the layout qualifier `" + v[k].s + "`");
            const std::string id = v[k].s;
            ++k;
            std::string val;
            if (k < e && is(v[k], "=")) {
                ++k;
                const size_t vb = k;
                while (k < e && !is(v[k], ",")) ++k;
                ConstEval ce{int_consts, v, vb, k};
                val = std::to_string(ce.sum());
            }
            q[id] = val;
            if (k < e && is(v[k], ",")) ++k;
        }
        return e + 1;
    }

    // [dims] at v[i]: returns the index behind them
    size_t parse_dims(size_t i, std::vector<int>& dims)
    {
        while (is(v[i], "[")) {
            const size_t e = match(v, i, v.size());
            if (e == i + 1) fail(v[i].line, "unsized arrays are not supported (the reference sizes a buffer from its block's members, pipeline_graph.rs:158-175: an unsized array has none)");
            ConstEval ce{int_consts, v, i + 1, e};
            const long n = ce.sum();
            if (ce.i != e) fail(v[i].line, "cannot evaluate the array size");
            if (n < 1 || n > (1 << 24)) fail(v[i].line, "array size " + std::to_string(n) + " is out of range");
            dims.push_back((int)n);
            i = e + 1;
        }
        return i;
    }

    // Block { members } [instance] ;  with v[i] == the block's type name, v[i + 1] == "{"
    size_t parse_block(size_t i, bool is_ubo, bool std140, GlslBlock& blk)
    {
        blk.type_name = v[i].s;
        const size_t close = match(v, i + 1, v.size());
        int cur = 0, max_align = 4;
        for (size_t k = i + 2; k < close;) {
            while (k < close && v[k].k == T_ID && (dropped_qualifiers().count(v[k].s) || v[k].s == "readonly" || v[k].s == "writeonly" || v[k].s == "coherent" || v[k].s == "restrict" || v[k].s == "volatile")) ++k;
            if (k >= close) break;
            if (is(v[k], "layout")) fail(v[k].line, "layout() on a block member (explicit offsets) is not supported");
            TypeInfo ti;
            if (v[k].k != T_ID || !block_member_type(v[k].s, ti)) fail(v[k].line, "block member of type `" + v[k].s + "`: members are float, int, uint, bool, vectors and matrices of them, and arrays (nested structs are not supported)");
            ++k;
            for (;;) {
                if (v[k].k != T_ID) fail(v[k].line, "cannot read the member name `" + v[k].s + "`");
                GlslMember m;
                m.name = v[k].s;
                m.base = ti.base;
                m.comps = ti.comps;
                m.cols = ti.cols;
                cx.struct_members.insert(m.name);
                k = parse_dims(k + 1, m.dims);
                // std140 / std430 (OpenGL 4.6 7.6.2.2): a vec3 is aligned like a vec4; an array element (and a matrix column) is
                // aligned to 16 bytes in std140, to its own alignment in std430
                const int vec_bytes = m.comps * 4, vec_align = m.comps == 3 ? 16 : vec_bytes;
                const bool arrayish = !m.dims.empty() || m.cols > 1;
                const int elem_align = arrayish && std140 ? 16 : vec_align;
                const int col_stride = arrayish ? round_up(vec_bytes, elem_align) : vec_bytes;
                long count = m.cols;
                for (int d : m.dims) count *= d;
                m.stride = m.cols > 1 ? col_stride * m.cols : col_stride;
                m.offset = round_up(cur, elem_align);
                m.bytes = arrayish ? (int)std::min<long>(count * col_stride, 1L << 30) : vec_bytes;
                if (arrayish && count * col_stride > (1L << 28)) fail(v[k].line, "block member `" + m.name + "` is larger than 256 MiB");
                cur = m.offset + m.bytes;
                max_align = std::max(max_align, elem_align);
                blk.members.push_back(m);
                if (is(v[k], ",")) { ++k; continue; }
                if (!is(v[k], ";")) fail(v[k].line, "expected `;` after the member `" + m.name + "`");
                ++k;
                break;
            }
        }
        blk.bytes = round_up(cur, std140 ? 16 : max_align);
        size_t k = close + 1;
        if (v[k].k == T_ID) { blk.instance = v[k].s; ++k; }
        if (is(v[k], "[")) fail(v[k].line, "arrays of blocks are not supported");
        if (!is(v[k], ";")) fail(v[k].line, "expected `;` after the block " + blk.type_name);
        if (blk.members.empty()) fail(v[i].line, "the block " + blk.type_name + " has no members");
        (void)is_ubo;
        return k + 1;
    }

    std::string dims_text(const std::vector<int>& d, size_t from = 0)
    {
        std::string s;
        for (size_t i = from; i < d.size(); ++i) s += "[" + std::to_string(d[i]) + "]";
        return s;
    }

    void add_ubo(GlslBlock blk, int line)
    {
        blk.ubo_base = round_up(sh.ubo_bytes, 16);
        sh.ubo_bytes = blk.ubo_base + blk.bytes;
        if (sh.ubo_bytes > kGlslMaxUniformBytes) fail(line, "the uniform blocks of this shader take " + std::to_string(sh.ubo_bytes) + " bytes; the limit is " + std::to_string(kGlslMaxUniformBytes));
        const std::string pre = blk.instance.empty() ? "" : blk.instance + ".";
        if (!blk.instance.empty()) members += "    struct " + blk.type_name + "_t {\n";
        for (auto& m : blk.members) {
            const std::string ty = cpp_type(m);
            members += std::string(blk.instance.empty() ? "    " : "        ") + ty + " " + m.name + dims_text(m.dims) + ";\n";
            const int at = blk.ubo_base + m.offset;
            long count = 1;
            for (int d : m.dims) count *= d;
            const std::string lv = pre + m.name;
            if (m.base == 'b') {
                if (m.comps != 1) fail(line, "boolean vectors in a uniform block are not supported");
                bind_ubo += "        for (int rfg_i = 0; rfg_i < " + std::to_string(count) + "; ++rfg_i) { unsigned rfg_t; __builtin_memcpy(&rfg_t, rfg_ubo + " + std::to_string(at) + " + rfg_i * " + std::to_string(m.stride) +
                        ", 4); reinterpret_cast<bool*>(&" + lv + ")[rfg_i] = rfg_t != 0u; }\n";
            } else if (m.cols > 1) {
                // a matrix: `cols` columns per element, a column every (stride / cols) bytes
                bind_ubo += "        for (int rfg_i = 0; rfg_i < " + std::to_string(count * m.cols) + "; ++rfg_i) __builtin_memcpy(reinterpret_cast<char*>(&" + lv + ") + rfg_i * " + std::to_string(m.comps == 2 ? 8 : 16) + ", rfg_ubo + " +
                        std::to_string(at) + " + rfg_i * " + std::to_string(m.stride / m.cols) + ", " + std::to_string(m.comps * 4) + ");\n";
            } else {
                bind_ubo += "        for (int rfg_i = 0; rfg_i < " + std::to_string(count) + "; ++rfg_i) __builtin_memcpy(reinterpret_cast<char*>(&" + lv + ") + rfg_i * " + std::to_string(cpp_elem_bytes(m)) + ", rfg_ubo + " + std::to_string(at) +
                        " + rfg_i * " + std::to_string(m.dims.empty() ? 0 : m.stride) + ", " + std::to_string(m.comps * 4) + ");\n";
            }
            m.name = pre + m.name;      // the key the reference's UBO map carries (pipeline_graph.rs:276-292)
        }
        if (!blk.instance.empty()) members += "    } " + blk.instance + ";\n";
        sh.ubos.push_back(blk);
    }

    void add_ssbo(GlslBlock blk, bool std140, int line)
    {
        if ((int)sh.ssbos.size() >= kGlslMaxBuffers) fail(line, "more than " + std::to_string(kGlslMaxBuffers) + " storage buffers");
        const std::string slot = std::to_string(sh.ssbos.size());
        if (!blk.instance.empty()) {
            members += "    struct " + blk.type_name + "_t {\n";
            for (const auto& m : blk.members) members += "        " + cpp_type(m) + " " + m.name + dims_text(m.dims) + ";\n";
            members += "    };\n    " + blk.type_name + "_t* " + blk.instance + ";\n";
            for (const auto& m : blk.members)
                members += "    static_assert(__builtin_offsetof(" + blk.type_name + "_t, " + m.name + ") == " + std::to_string(m.offset) + ", \"" + type + ".comp: the member " + m.name + " of the storage block " +
                           blk.type_name + " does not lie where " + (std140 ? "std140" : "std430") + " puts it (a scalar behind a vec3?): not supported\");\n";
            bind += "        " + blk.instance + " = static_cast<" + blk.type_name + "_t*>(rfg_buf[" + slot + "]);\n";
            cx.ssbo_instances.insert(blk.instance);
        }
        for (const auto& m : blk.members) {
            const bool arrayish = !m.dims.empty();
            if (m.base == 'b') fail(line, "bool members of a storage block are not supported");
            if ((arrayish || m.cols > 1) && m.stride != cpp_elem_bytes(m))
                fail(line, "the member " + m.name + " of the storage block " + blk.type_name + " has a " + std::to_string(m.stride) + "-byte array stride (std140); declare the block std430");
            if (!blk.instance.empty()) continue;
            const std::string ty = cpp_type(m), at = "static_cast<char*>(rfg_buf[" + slot + "]) + " + std::to_string(m.offset);
            if (!arrayish) {
                members += "    " + ty + "* rfg_p_" + m.name + ";\n";
                bind += "        rfg_p_" + m.name + " = reinterpret_cast<" + ty + "*>(" + at + ");\n";
                cx.ssbo_scalars.insert(m.name);
            } else if (m.dims.size() == 1) {
                members += "    " + ty + "* " + m.name + ";\n";
                bind += "        " + m.name + " = reinterpret_cast<" + ty + "*>(" + at + ");\n";
            } else {
                members += "    " + ty + " (*" + m.name + ")" + dims_text(m.dims, 1) + ";\n";
                bind += "        " + m.name + " = reinterpret_cast<" + ty + " (*)" + dims_text(m.dims, 1) + ">(" + at + ");\n";
            }
        }
        sh.ssbos.push_back(blk);
    }

    // a function definition whose name is v[name]; v[name + 1] == "(".  `first` = first token of the statement.
    size_t function(size_t first, size_t name)
    {
        const size_t close = match(v, name + 1, v.size());
        // parameters
        size_t pb = name + 2;
        while (pb < close) {
            size_t pe = pb;
            int depth = 0;
            while (pe < close && !(depth == 0 && is(v[pe], ","))) {
                if (is(v[pe], "(") || is(v[pe], "[")) ++depth;
                if (is(v[pe], ")") || is(v[pe], "]")) --depth;
                ++pe;
            }
            bool by_ref = false;
            size_t pname = pe;
            int d2 = 0;
            for (size_t k = pb; k < pe; ++k) {
                if (is(v[k], "[")) ++d2;
                if (is(v[k], "]")) --d2;
                if (d2 == 0 && v[k].k == T_ID) {
                    if (v[k].s == "out" || v[k].s == "inout") { by_ref = true; v[k].drop = true; }
                    else if (v[k].s == "in") v[k].drop = true;
                    else if (v[k].s != "const" && !dropped_qualifiers().count(v[k].s)) pname = k;      // the last identifier outside brackets: the name
                }
            }
            if (by_ref) {      // (an array parameter: its subscripts become part of the type, rewrite() -- `rfg_arr<float, (3)> &w`; by value it is a copy, as in GLSL)
                if (pname >= pe) fail(v[pb].line, "an out parameter without a name");
                v[pname].s = "&" + v[pname].s;
            }
            pb = pe + 1;
        }
        rewrite(v, first, close + 1, cx);
        size_t after = close + 1;
        if (is(v[after], ";")) {      // a prototype: member functions need none (and must not be declared twice)
            blank(v, first, after + 1);
            return after + 1;
        }
        if (!is(v[after], "{")) fail(v[after].line, "expected the body of " + v[name].s);
        const size_t end = match(v, after, v.size());
        rewrite(v, after, end + 1, cx);
        size_t f = first;
        while (v[f].drop) ++f;
        v[f].s = "RFG " + v[f].s;
        if (v[name].s == "main") has_main = true;
        return end + 1;
    }

    void run()
    {
        v = lex(sh.source);
        sh.source.clear();
        for (const auto& t : v)
            if (t.k == T_ID && (t.s == "gl_LocalInvocationID" || t.s == "gl_LocalInvocationIndex" || t.s == "gl_WorkGroupID" || t.s == "gl_NumWorkGroups" || t.s == "shared" || t.s == "barrier" ||
                                t.s == "memoryBarrierShared" || t.s == "groupMemoryBarrier"))
                sh.grouped = true;
        bool local_size = false;
        size_t i = 0;
        while (v[i].k != T_END) {
            Tok& t = v[i];
            if (t.k == T_PP) {
                const std::string d = directive(t, cx, sh);
                // an object-like macro whose body is an integer expression can size arrays of blocks
                if (d.compare(0, 7, "#define") == 0) {
                    std::vector<Tok> r = lex(d.substr(7));
                    if (r.size() >= 3 && r[0].k == T_ID && !(is(r[1], "(") && r[1].ws.empty())) {
                        try {
                            ConstEval ce{int_consts, r, 1, r.size() - 1};
                            const long val = ce.sum();
                            if (ce.i == r.size() - 1) int_consts[r[0].s] = val;
                        } catch (const Fail&) {
                        }
                    }
                }
                t.s = d;
                if (d.empty()) t.drop = true;
                else hoisted += d + "\n";
                ++i;
                continue;
            }
            if (is(t, ";")) { ++i; continue; }
            const size_t first = i;
            std::map<std::string, std::string> lq;
            bool has_layout = false;
            if (is(v[i], "layout")) { i = parse_layout(i, lq); has_layout = true; }
            bool q_uniform = false, q_buffer = false, q_ro = false, q_wo = false, q_shared = false, q_const = false, q_in = false;
            for (;; ++i) {
                if (v[i].k != T_ID) break;
                const std::string& s = v[i].s;
                if (s == "uniform") q_uniform = true;
                else if (s == "buffer") q_buffer = true;
                else if (s == "readonly") q_ro = true;
                else if (s == "writeonly") q_wo = true;
                else if (s == "shared") q_shared = true;
                else if (s == "const") q_const = true;
                else if (s == "in") q_in = true;
                else if (s == "coherent" || s == "volatile" || s == "restrict" || dropped_qualifiers().count(s)) {}
                else break;
            }
            if (q_in && has_layout) {      // layout(local_size_x = ...) in;
                if (!is(v[i], ";")) fail(v[i].line, "expected `;` after `in`");
                for (const auto& kv : lq) {
                    int* dst = kv.first == "local_size_x" ? &sh.lx : (kv.first == "local_size_y" ? &sh.ly : (kv.first == "local_size_z" ? &sh.lz : nullptr));
                    if (!dst) fail(v[first].line, "layout(" + kv.first + ") in: only local_size_x / _y / _z are read");
                    *dst = std::atoi(kv.second.c_str());
                }
                local_size = true;
                blank(v, first, i + 1);
                ++i;
                continue;
            }
            if (v[i].k == T_ID && v[i].s == "precision") {
                while (!is(v[i], ";")) { if (v[i].k == T_END) fail(v[first].line, "precision statement without `;`"); ++i; }
                blank(v, first, i + 1);
                ++i;
                continue;
            }
            if (q_uniform || q_buffer) {
                if (lq.count("set") && lq["set"] != "0") fail(v[first].line, "descriptor set " + lq["set"] + ": only set 0 is supported (shader.rs:125-127)");
                const int binding = lq.count("binding") ? std::atoi(lq["binding"].c_str()) : -1;
                if (v[i].k != T_ID) fail(v[i].line, "cannot read this declaration");
                if (q_uniform && (v[i].s == "image2D" || v[i].s == "sampler2D")) {
                    const bool sampled = v[i].s == "sampler2D";
                    size_t k = i + 1;
                    if (v[k].k != T_ID) fail(v[k].line, "an image variable needs a name");
                    GlslImageVar im;
                    im.name = v[k].s;
                    im.binding = binding;
                    im.readonly = q_ro || sampled;
                    im.writeonly = q_wo && !sampled;
                    im.sampled = sampled;
                    if (!is(v[k + 1], ";")) fail(v[k + 1].line, is(v[k + 1], "[") ? "arrays of images are not supported" : "expected `;` after the image variable " + im.name);
                    claim_binding(binding, v[first].line, std::string(sampled ? "sampler2D " : "image2D ") + im.name);
                    if ((int)sh.images.size() >= kGlslMaxImages) fail(v[first].line, "more than " + std::to_string(kGlslMaxImages) + " image variables");
                    for (const auto& o : sh.images)
                        if (o.name == im.name) fail(v[k].line, "the image variable " + im.name + " is declared twice");
                    members += std::string("    ") + (sampled ? "sampler2D" : "image2D") + "<RfgPx> " + im.name + ";\n";
                    bind += "        " + im.name + (sampled ? ".im" : "") + " = image2D<RfgPx>{rfg_img[" + std::to_string(sh.images.size()) + "].base, rfg_img[" + std::to_string(sh.images.size()) + "].pitch, rfg_f.W, rfg_f.H, rfg_f.row_lo, rfg_f.row_hi, rfg_f.y0, rfg_f.y1 - 1, rfg_f.zero};\n        " + im.name + (sampled ? ".im" : "") + ".finish();\n";
                    sh.images.push_back(im);
                    blank(v, first, k + 2);
                    i = k + 2;
                    continue;
                }
                if (!is(v[i + 1], "{")) fail(v[i].line, "`" + v[i].s + "`: a uniform is a storage image (image2D), a combined image sampler (sampler2D) or a block; a buffer is a block");
                GlslBlock blk;
                blk.binding = binding;
                blk.readonly = q_ro;
                blk.writeonly = q_wo;
                const bool std140 = q_uniform ? true : lq.count("std140") > 0;
                if (q_uniform && lq.count("std430")) fail(v[first].line, "a uniform block cannot be std430");
                const size_t end = parse_block(i, q_uniform, std140, blk);
                claim_binding(binding, v[first].line, "block " + blk.type_name);
                if (q_uniform) add_ubo(blk, v[first].line);
                else {
                    for (const auto& o : sh.ssbos)
                        if (o.type_name == blk.type_name) fail(v[first].line, "the storage block " + blk.type_name + " is declared twice (buffers are found by their block's type name, shader.rs:144-147)");
                    add_ssbo(blk, std140, v[first].line);
                }
                blank(v, first, end);
                i = end;
                continue;
            }
            if (has_layout && !lq.count("constant_id")) fail(v[first].line, "this layout() declaration is not supported");
            if (has_layout) blank(v, first, i), rewrite(v, first, first, cx);      // layout(constant_id = N) const T x = default; -> the default
            if (v[i].k == T_ID && v[i].s == "struct") {
                if (v[i + 1].k != T_ID || !is(v[i + 2], "{")) fail(v[i].line, "cannot read this struct");
                cx.structs.insert(v[i + 1].s);
                const size_t close = match(v, i + 2, v.size());
                std::string same;      // GLSL compares structs member by member (== and != are calls here: rfg_eq finds this operator)
                for (size_t k = i + 3; k < close; ++k)
                    if (v[k].k == T_ID && (is(v[k + 1], ";") || is(v[k + 1], ",") || is(v[k + 1], "["))) {
                        cx.struct_members.insert(v[k].s);
                        if (!vector_types().count(v[k].s) && !scalar_types().count(v[k].s) && !cx.structs.count(v[k].s))
                        {
                            const std::string m = v[k].s + (cpp_only_keywords().count(v[k].s) ? "_rfg" : "");      // as rewrite() renames it
                            same += (same.empty() ? "" : " && ") + std::string("rfg_eq(") + m + ", rfg_o." + m + ")";
                        }
                    }
                v[close].pre += "RFG bool operator==(const " + v[i + 1].s + "& rfg_o) const { return " + (same.empty() ? "true" : same) + "; } ";
                size_t e = close + 1;
                while (!is(v[e], ";")) { if (v[e].k == T_END) fail(v[i].line, "struct without `;`"); ++e; }
                rewrite(v, i, e + 1, cx);
                i = e + 1;
                continue;
            }
            // [const] type name ...
            if (v[i].k != T_ID) fail(v[i].line, "cannot read this declaration (at `" + v[i].s + "`)");
            const size_t ty = i;
            size_t k = i + 1;
            if (is(v[k], "[")) k = match(v, k, v.size()) + 1;      // float[3] name
            if (v[k].k != T_ID) fail(v[k].line, "cannot read this declaration (at `" + v[k].s + "`)");
            if (is(v[k + 1], "(")) { i = function(first, k); continue; }
            // a global variable: to its `;`
            size_t e = k;
            int depth = 0;
            while (!(depth == 0 && is(v[e], ";"))) {
                if (v[e].k == T_END) fail(v[first].line, "declaration without `;`");
                if (is(v[e], "(") || is(v[e], "[") || is(v[e], "{")) ++depth;
                if (is(v[e], ")") || is(v[e], "]") || is(v[e], "}")) --depth;
                ++e;
            }
            rewrite(v, first, e + 1, cx);
            if (q_shared) {
                // LDS of the workgroup: a variable in front of the struct (a class cannot hold one)
                std::string text;
                for (size_t q = first; q <= e; ++q)
                    if (!v[q].drop && !(v[q].k == T_ID && v[q].s == "shared")) text += (text.empty() ? "" : " ") + v[q].s;
                hoisted += "__shared__ " + text + "\n";
                blank(v, first, e + 1);
            } else if (q_const && scalar_types().count(v[ty].s) && is(v[k + 1], "=")) {
                bool calls = false;
                for (size_t q = k + 2; q < e; ++q) calls = calls || (v[q].k == T_ID && is(v[q + 1], "("));
                if (!calls) {
                    // a constant of literals: in front of the struct, where array sizes (also of shared variables) can use it
                    std::string text;
                    for (size_t q = first; q <= e; ++q)
                        if (!v[q].drop && !(v[q].k == T_ID && v[q].s == "const")) text += (text.empty() ? "" : " ") + v[q].s;
                    hoisted += "static constexpr " + text + "\n";
                    if (v[ty].s == "int" || v[ty].s == "uint") {
                        try {
                            ConstEval ce{int_consts, v, k + 2, e};
                            const long val = ce.sum();
                            if (ce.i == e) int_consts[v[k].s] = val;
                        } catch (const Fail&) {
                        }
                    }
                    blank(v, first, e + 1);
                }
            }
            i = e + 1;
        }
        if (!has_main) fail(v.back().line, "no `void main()`");
        if (!local_size) { sh.lx = sh.ly = sh.lz = 1; }
        if (sh.lx < 1 || sh.ly < 1 || sh.lz < 1 || (long)sh.lx * sh.ly * sh.lz > 1024) fail(1, "local_size " + std::to_string(sh.lx) + " x " + std::to_string(sh.ly) + " x " + std::to_string(sh.lz) + " is out of range (1..1024 invocations)");
    }
};

// Is the shader a POINT operation on one image -- every invocation reads only its own texel of one image and writes only its own
// texel of one image (or of the same one), and nothing it computes depends on where the invocation is?  Then it can run as a ROW
// STAGE of the stream kernel (StUser, rf_stream_dev.h) and fuse with its neighbours, as a {type}.stage.hip of RADIUS 0 does.
// Decided on the file's own tokens, conservatively (a file that is not recognised is simply a node with a kernel of its own):
//   * one readonly + one writeonly image2D, or one image2D that is neither; no storage block, no sampler, no workgroup built-in;
//     local_size at least 16 x 16 (the reference's dispatch then covers the frame);
//   * imageLoad / imageStore / imageSize / gl_GlobalInvocationID / the image variables appear in main() only, and not in a #define;
//   * the coordinate of every imageLoad and imageStore is `ivec2(gl_GlobalInvocationID.xy)` or a variable declared as exactly that;
//   * such a variable, and one declared `ivec2 S = imageSize(image);`, is otherwise used only in the frame guard
//     `if (C.x >= S.x || C.y >= S.y) return;` / `if (any(greaterThanEqual(C, S))) return;`.
bool point_shader(const std::string& text, const GlslShader& sh, std::string& in_name, std::string& out_name)
{
    if (sh.grouped || !sh.ssbos.empty() || sh.radius > 0) return false;
    if (sh.lx < 16 || sh.ly < 16) return false;      // the reference dispatches ceil(W/16) x ceil(H/16) workgroups: a smaller local_size covers part of the frame only
    for (const auto& im : sh.images)
        if (im.sampled) return false;
    if (sh.images.size() == 1 && !sh.images[0].readonly && !sh.images[0].writeonly) in_name = out_name = sh.images[0].name;
    else if (sh.images.size() == 2 && sh.images[0].readonly != sh.images[1].readonly && sh.images[0].writeonly != sh.images[1].writeonly &&
             (sh.images[0].readonly || sh.images[0].writeonly) && (sh.images[1].readonly || sh.images[1].writeonly)) {
        in_name = sh.images[sh.images[0].readonly ? 0 : 1].name;
        out_name = sh.images[sh.images[0].readonly ? 1 : 0].name;
    } else return false;
    std::vector<Tok> t;
    try {
        t = lex(text);
    } catch (const Fail&) {
        return false;
    }
    auto special = [&](const std::string& s) {
        return s == "gl_GlobalInvocationID" || s == "imageLoad" || s == "imageStore" || s == "imageSize" || s == in_name || s == out_name;
    };
    static const char* forbidden[] = {"gl_WorkGroupID", "gl_LocalInvocationID", "gl_LocalInvocationIndex", "gl_NumWorkGroups", "gl_WorkGroupSize"};
    size_t mb = 0, me = 0;
    for (size_t i = 0; i + 4 < t.size(); ++i)
        if (t[i].k == T_ID && t[i].s == "main" && is(t[i + 1], "(") && i > 0 && is(t[i - 1], "void")) {
            size_t k = match(t, i + 1, t.size()) + 1;
            if (!is(t[k], "{")) return false;
            mb = k + 1;
            me = match(t, k, t.size());
        }
    if (me == 0) return false;
    for (size_t i = 0; i < t.size(); ++i) {
        if (t[i].k == T_PP) {
            std::vector<Tok> d = lex(t[i].s.substr(1));
            for (const auto& q : d)
                if (q.k == T_ID && special(q.s)) return false;
            continue;
        }
        if (t[i].k != T_ID) continue;
        for (const char* f : forbidden)
            if (t[i].s == f) return false;
        if ((i < mb || i >= me) && special(t[i].s)) {
            // outside main(): only the declarations of the images themselves (`uniform ... image2D name;`)
            if ((t[i].s == in_name || t[i].s == out_name) && i > 0 && is(t[i - 1], "image2D")) continue;
            return false;
        }
    }
    std::vector<char> ok(t.size(), 0);
    std::set<std::string> coord, size;
    auto seq = [&](size_t i, std::initializer_list<const char*> pat) {
        size_t k = i;
        for (const char* p : pat) {
            if (k >= me) return false;
            if (std::string(p) == "$ID") { if (t[k].k != T_ID) return false; }
            else if (!is(t[k], p)) return false;
            ++k;
        }
        return true;
    };
    auto mark = [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) ok[k] = 1; };
    // inline coordinate `ivec2 ( gl_GlobalInvocationID . xy )` or `ivec2 ( gl_GlobalInvocationID )` at i: its length, 0 if none
    auto inline_coord = [&](size_t i) -> size_t {
        if (seq(i, {"ivec2", "(", "gl_GlobalInvocationID", ".", "xy", ")"})) return 6;
        if (seq(i, {"ivec2", "(", "gl_GlobalInvocationID", ")"})) return 4;
        return 0;
    };
    for (size_t i = mb; i < me; ++i) {      // declarations of coordinate and size variables
        if (!is(t[i], "ivec2") || t[i + 1].k != T_ID || !is(t[i + 2], "=")) continue;
        const size_t n = inline_coord(i + 3);
        if (n && is(t[i + 3 + n], ";")) { coord.insert(t[i + 1].s); mark(i, i + 4 + n); continue; }
        if (seq(i + 3, {"imageSize", "(", "$ID", ")", ";"}) && (t[i + 5].s == in_name || t[i + 5].s == out_name)) { size.insert(t[i + 1].s); mark(i, i + 8); }
    }
    // a coordinate operand at i: a coordinate variable or the inline form; returns its length
    auto coord_at = [&](size_t i) -> size_t { return (t[i].k == T_ID && coord.count(t[i].s)) ? 1 : inline_coord(i); };
    auto size_at = [&](size_t i) -> size_t {
        if (t[i].k == T_ID && size.count(t[i].s)) return 1;
        return (seq(i, {"imageSize", "(", "$ID", ")"}) && (t[i + 2].s == in_name || t[i + 2].s == out_name)) ? 4 : 0;
    };
    for (size_t i = mb; i < me; ++i) {
        if (t[i].k != T_ID) continue;
        if ((t[i].s == "imageLoad" || t[i].s == "imageStore") && is(t[i + 1], "(") && t[i + 2].k == T_ID && is(t[i + 3], ",")) {
            const bool load = t[i].s == "imageLoad";
            if (t[i + 2].s != (load ? in_name : out_name)) return false;
            const size_t n = coord_at(i + 4);
            if (!n || !is(t[i + 4 + n], load ? ")" : ",")) return false;
            mark(i, i + 4 + n);
        } else if (t[i].s == "if" && is(t[i + 1], "(")) {
            // the frame guard, in one of its two spellings, followed by `return;` or `{ return; }`
            const size_t close = match(t, i + 1, me);
            size_t k = close + 1;
            const bool braces = is(t[k], "{");
            if (braces) ++k;
            if (!(is(t[k], "return") && is(t[k + 1], ";") && (!braces || is(t[k + 2], "}")))) continue;
            size_t q = i + 2;
            bool guard = false;
            if (seq(q, {"any", "(", "greaterThanEqual", "("})) {
                size_t a = q + 4;
                const size_t n1 = coord_at(a);
                if (n1 && is(t[a + n1], ",")) {
                    const size_t n2 = size_at(a + n1 + 1);
                    guard = n2 && seq(a + n1 + 1 + n2, {")", ")", ")"}) && a + n1 + 1 + n2 + 2 == close;
                }
            } else {
                // C.x >= S.x || C.y >= S.y (either order)
                auto half = [&](size_t a, char want, size_t& end) {
                    const size_t n1 = coord_at(a);
                    if (!n1 || !is(t[a + n1], ".") || t[a + n1 + 1].s != std::string(1, want)) return false;
                    size_t b = a + n1 + 2;
                    if (!(is(t[b], ">") && is(t[b + 1], "=") && t[b + 1].ws.empty())) return false;
                    const size_t n2 = size_at(b + 2);
                    if (!n2 || !is(t[b + 2 + n2], ".") || t[b + 2 + n2 + 1].s != std::string(1, want)) return false;
                    end = b + 2 + n2 + 2;
                    return true;
                };
                for (int order = 0; order < 2 && !guard; ++order) {
                    size_t e1 = 0, e2 = 0;
                    if (half(q, order ? 'y' : 'x', e1) && is(t[e1], "|") && is(t[e1 + 1], "|") && half(e1 + 2, order ? 'x' : 'y', e2) && e2 == close) guard = true;
                }
            }
            if (guard) mark(i, close + 1);
        }
    }
    for (size_t i = mb; i < me; ++i)
        if (!ok[i] && t[i].k == T_ID && (special(t[i].s) || coord.count(t[i].s) || size.count(t[i].s))) return false;
    return true;
}

// Is the shader a TRANSLATION-INVARIANT STENCIL -- what an invocation writes depends on the texels around its own position and on nothing
// else about that position or the frame?  Then (with `#pragma rf radius R`) it can run on the LDS-tiled window kernel the stage files use
// (user_node_kernel, rf_user_dev.h): in a virtual frame of (2R+1)^2 texels around the invocation every coordinate the shader computes is a
// constant of the unrolled code, so a tap is a register or an LDS read at a fixed offset.  Decided on the file's tokens, conservatively:
// position and frame size (gl_GlobalInvocationID, imageSize and every integer variable or helper parameter computed from them) may be
// used ONLY to compute further integer coordinates, as the coordinate of an imageLoad, in the frame guard, and as the coordinate of an
// imageStore (the invocation's own).  A use anywhere else -- a float, a condition, a loop bound -- makes the shader position-dependent
// as far as this analysis can tell, and it keeps its generic kernel.  (The radius itself and the border behaviour are not proven here: the
// frame's border texels are computed by the generic kernel, and rf_graph_create runs both kernels on a small random frame and keeps the
// window kernel only if the two agree bit for bit.)
bool stencil_shader(const std::string& text, const GlslShader& sh, std::string& why)
{
    auto no = [&](const std::string& reason) { why = reason; return false; };
    if (sh.radius < 1) return no(sh.radius < 0 ? "it does not state `#pragma rf radius N`" : "its stated radius is 0");
    if (sh.radius > 15) return no("its stated radius is above 15");
    if (sh.grouped) return no("it uses workgroup built-ins, shared variables or barrier()");
    if (!sh.ssbos.empty()) return no("it has a storage block");
    if (sh.lx < 16 || sh.ly < 16) return no("its local_size is below 16 x 16 (the reference's dispatch then covers part of the frame only)");
    if (sh.ubo_bytes > 56) return no("its uniform blocks take more than 56 bytes");
    std::set<std::string> in_names, out_names;
    for (const auto& im : sh.images) {
        if (im.sampled) return no("it samples `" + im.name + "`");
        if (im.readonly == im.writeonly) return no("`" + im.name + "` is read AND written");      // every image is read or written, never both
        (im.readonly ? in_names : out_names).insert(im.name);
    }
    if (in_names.empty() || out_names.empty() || in_names.size() > 4 || out_names.size() > 4) return no("it needs one to four images read and one to four written");
    const std::vector<Tok> t = lex(text);
    const size_t N = t.size();
    auto is_image = [&](const std::string& s) { return in_names.count(s) || out_names.count(s); };
    static const std::set<std::string> int_types = {"int", "uint", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4"};
    static const char* forbidden[] = {"gl_WorkGroupID", "gl_LocalInvocationID", "gl_LocalInvocationIndex", "gl_NumWorkGroups", "gl_WorkGroupSize"};
    for (const auto& q : t) {
        if (q.k == T_PP) {
            const std::vector<Tok> d = lex(q.s.substr(1));
            for (const auto& w : d)
                if (w.k == T_ID && (w.s == "gl_GlobalInvocationID" || w.s == "imageSize" || w.s == "imageLoad" || w.s == "imageStore" || is_image(w.s))) return no("line " + std::to_string(q.line) + ": a #define mentions `" + w.s + "`");
        } else if (q.k == T_ID) {
            for (const char* f : forbidden)
                if (q.s == f) return no("line " + std::to_string(q.line) + ": it uses `" + q.s + "`");
        }
    }
    // ---- functions -----------------------------------------------------------------------------------------------------------------
    struct Fn { std::vector<std::string> pname, ptype; std::vector<bool> byref; std::vector<size_t> pat; size_t b = 0, e = 0; std::string ret; };
    std::map<std::string, Fn> fns;
    {
        int depth = 0;
        for (size_t i = 0; i + 3 < N; ++i) {
            if (is(t[i], "{")) ++depth;
            if (is(t[i], "}")) --depth;
            if (depth != 0 || t[i].k != T_ID || t[i + 1].k != T_ID || !is(t[i + 2], "(")) continue;
            const size_t close = match(t, i + 2, N);
            if (!is(t[close + 1], "{")) continue;
            Fn f;
            f.ret = t[i].s;
            f.b = close + 2;
            f.e = match(t, close + 1, N);
            size_t pb = i + 3;
            while (pb < close) {
                size_t pe = pb;
                int d = 0;
                while (pe < close && !(d == 0 && is(t[pe], ","))) { if (is(t[pe], "(") || is(t[pe], "[")) ++d; if (is(t[pe], ")") || is(t[pe], "]")) --d; ++pe; }
                std::string ty, nm;
                size_t at = 0;
                bool ref = false;
                int bd = 0;
                for (size_t k = pb; k < pe; ++k) {
                    if (is(t[k], "[")) ++bd;
                    if (is(t[k], "]")) --bd;
                    if (bd || t[k].k != T_ID) continue;
                    if (t[k].s == "out" || t[k].s == "inout") ref = true;
                    else if (t[k].s == "in" || t[k].s == "const" || dropped_qualifiers().count(t[k].s)) {}
                    else if (ty.empty()) ty = t[k].s;
                    else { nm = t[k].s; at = k; }
                }
                if (!ty.empty() && ty != "void") { f.ptype.push_back(ty); f.pname.push_back(nm); f.byref.push_back(ref); f.pat.push_back(at); }
                pb = pe + 1;
            }
            fns[t[i + 1].s] = f;
            i = close;
        }
    }
    if (!fns.count("main")) return no("no main()");
    // ---- integer variables, each with the range of tokens it is visible in -------------------------------------------------------
    // position reaches a variable (`pos`) / a uniform reaches it (`var`): flags of the DECLARATION, so that the `i` of one loop is not the
    // `i` of another
    struct Decl { std::string name; size_t at, b, e; bool pos = false, var = false; };
    std::vector<Decl> decls;
    for (const auto& kv : fns) {
        const Fn& f = kv.second;
        for (size_t p = 0; p < f.pname.size(); ++p)
            if (int_types.count(f.ptype[p]) && !f.pname[p].empty()) decls.push_back(Decl{f.pname[p], f.pat[p], f.b, f.e});
        std::vector<size_t> block_end;      // innermost last: where the enclosing { } ends
        block_end.push_back(f.e);
        std::vector<std::pair<size_t, size_t>> for_scope;      // (init range end, statement end) of the `for` statements we are inside the header of
        for (size_t i = f.b; i < f.e; ++i) {
            if (is(t[i], "{")) block_end.push_back(match(t, i, f.e));
            else if (is(t[i], "}") && block_end.size() > 1) block_end.pop_back();
            if (t[i].k != T_ID || !int_types.count(t[i].s) || t[i + 1].k != T_ID || is(t[i + 1], "(")) continue;
            // the scope: the enclosing block -- or, in the header of a `for`, that statement
            size_t scope_end = block_end.back();
            {
                size_t k = i;      // are we inside `for ( ... ;` ?  walk back over the init expression to the opening parenthesis
                int d = 0;
                while (k > f.b) {
                    --k;
                    if (is(t[k], ")") || is(t[k], "]")) ++d;
                    else if (is(t[k], "(") || is(t[k], "[")) { if (d == 0) break; --d; }
                    else if (d == 0 && (is(t[k], ";") || is(t[k], "{") || is(t[k], "}"))) { k = 0; break; }
                }
                if (k > f.b && is(t[k], "(") && t[k - 1].k == T_ID && t[k - 1].s == "for") {
                    const size_t close = match(t, k, f.e);
                    size_t b1 = close + 1;
                    if (is(t[b1], "{")) scope_end = match(t, b1, f.e) + 1;
                    else { int d2 = 0; while (b1 < f.e && !(d2 == 0 && is(t[b1], ";"))) { if (is(t[b1], "(") || is(t[b1], "{")) ++d2; if (is(t[b1], ")") || is(t[b1], "}")) --d2; ++b1; } scope_end = b1 + 1; }
                }
            }
            size_t k = i + 1;
            for (;;) {      // INTTYPE a = e, b[3], c = e2;
                if (t[k].k != T_ID) break;
                decls.push_back(Decl{t[k].s, k, k, scope_end});
                size_t q = k + 1;
                while (is(t[q], "[")) q = match(t, q, f.e) + 1;
                if (is(t[q], "=")) {
                    int d = 0;
                    ++q;
                    while (q < f.e && !(d == 0 && (is(t[q], ",") || is(t[q], ";") || is(t[q], ")")))) { if (is(t[q], "(") || is(t[q], "[") || is(t[q], "{")) ++d; if (is(t[q], ")") || is(t[q], "]") || is(t[q], "}")) --d; ++q; }
                }
                if (is(t[q], ",")) { k = q + 1; continue; }
                break;
            }
        }
    }
    auto resolve = [&](size_t i) -> Decl* {      // the declaration an identifier at i names: the innermost one in scope
        Decl* best = nullptr;
        for (auto& d : decls)
            if (d.name == t[i].s && d.b <= i && i < d.e && (!best || d.b >= best->b)) best = &d;
        return best;
    };
    std::set<std::string> uniforms;
    for (const auto& blk : sh.ubos) {
        if (!blk.instance.empty()) uniforms.insert(blk.instance);
        for (const auto& m : blk.members) uniforms.insert(m.name.substr(m.name.rfind('.') == std::string::npos ? 0 : m.name.rfind('.') + 1));
    }
    std::set<std::string> int_returning;      // helpers that return an integer computed from the position: their calls are such values
    auto names = [&](size_t i) { return t[i].k == T_ID && !(i > 0 && is(t[i - 1], ".")); };
    auto pos_at = [&](size_t i) {
        if (!names(i)) return false;
        if (t[i].s == "gl_GlobalInvocationID" || t[i].s == "imageSize" || int_returning.count(t[i].s)) return true;
        const Decl* d = resolve(i);
        return d && d->pos;
    };
    auto var_at = [&](size_t i) {
        if (!names(i)) return false;
        const Decl* d = resolve(i);
        return d ? d->var : uniforms.count(t[i].s) > 0;
    };
    auto has_pos = [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) if (pos_at(k)) return true; return false; };
    auto has_var = [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) if (var_at(k)) return true; return false; };
    auto arg_end = [&](size_t a, size_t limit) {
        int d = 0;
        size_t k = a;
        for (; k < limit; ++k) {
            if (t[k].k != T_PUNCT) continue;
            const char c = t[k].s[0];
            if (c == '(' || c == '[' || c == '{') ++d;
            else if (c == ')' || c == ']' || c == '}') { if (d == 0) break; --d; }
            else if (d == 0 && (c == ',' || c == ';')) break;
        }
        return k;
    };
    // pure coordinate / size variables of main (the frame guard and the store coordinate are written with these)
    const Fn& mainf = fns["main"];
    std::set<std::string> pure_coord, pure_size;
    auto seq = [&](size_t i, std::initializer_list<const char*> pat) {
        size_t k = i;
        for (const char* p : pat) {
            if (k >= N) return false;
            if (std::string(p) == "$ID") { if (t[k].k != T_ID) return false; }
            else if (!is(t[k], p)) return false;
            ++k;
        }
        return true;
    };
    auto inline_coord = [&](size_t i) -> size_t {
        if (seq(i, {"ivec2", "(", "gl_GlobalInvocationID", ".", "xy", ")"})) return 6;
        if (seq(i, {"ivec2", "(", "gl_GlobalInvocationID", ")"})) return 4;
        return 0;
    };
    for (size_t i = mainf.b; i < mainf.e; ++i) {
        if (!is(t[i], "ivec2") || t[i + 1].k != T_ID || !is(t[i + 2], "=")) continue;
        const size_t n = inline_coord(i + 3);
        if (n && is(t[i + 3 + n], ";")) pure_coord.insert(t[i + 1].s);
        else if (seq(i + 3, {"imageSize", "(", "$ID", ")", ";"}) && is_image(t[i + 5].s)) pure_size.insert(t[i + 1].s);
    }
    auto coord_at = [&](size_t i) -> size_t { return (t[i].k == T_ID && pure_coord.count(t[i].s)) ? 1 : inline_coord(i); };
    auto size_at = [&](size_t i) -> size_t {
        if (t[i].k == T_ID && pure_size.count(t[i].s)) return 1;
        return (seq(i, {"imageSize", "(", "$ID", ")"}) && is_image(t[i + 2].s)) ? 4 : 0;
    };
    std::vector<char> ok, controlled;
    auto flags = [&]() { size_t n = int_returning.size(); for (const auto& d : decls) n += (d.pos ? 1 : 0) + (d.var ? 2 : 0); return n; };
    for (int round = 0; round < 24; ++round) {
        const size_t before = flags();
        ok.assign(N, 0);
        controlled.assign(N, 0);
        auto mark = [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) ok[k] = 1; };
        // statements whose execution depends on a uniform: what they assign follows the uniform too
        for (const auto& kv : fns) {
            const Fn& f = kv.second;
            for (size_t i = f.b; i < f.e; ++i) {
                if (t[i].k != T_ID || !(t[i].s == "if" || t[i].s == "for" || t[i].s == "while") || !is(t[i + 1], "(")) continue;
                const size_t close = match(t, i + 1, f.e);
                size_t c0 = i + 2, c1 = close;
                if (t[i].s == "for") {
                    int d = 0;
                    std::vector<size_t> semi;
                    for (size_t k = i + 2; k < close; ++k) {
                        if (is(t[k], "(") || is(t[k], "[")) ++d;
                        if (is(t[k], ")") || is(t[k], "]")) --d;
                        if (d == 0 && is(t[k], ";")) semi.push_back(k);
                    }
                    if (semi.size() == 2) { c0 = semi[0] + 1; c1 = semi[1]; }
                }
                if (!has_var(c0, c1)) continue;
                for (size_t k = c0; k < c1; ++k)
                    if (names(k)) if (Decl* d = resolve(k)) d->var = true;
                size_t b0 = close + 1, b1;
                for (;;) {
                    if (is(t[b0], "{")) b1 = match(t, b0, f.e) + 1;
                    else { b1 = b0; int d = 0; while (b1 < f.e && !(d == 0 && is(t[b1], ";"))) { if (is(t[b1], "(") || is(t[b1], "{")) ++d; if (is(t[b1], ")") || is(t[b1], "}")) --d; ++b1; } ++b1; }
                    for (size_t k = (t[i].s == "for" ? i : b0); k < b1 && k < f.e; ++k) controlled[k] = 1;
                    if (b1 < f.e && t[b1].k == T_ID && t[b1].s == "else") { b0 = b1 + 1; continue; }
                    break;
                }
            }
        }
        for (const auto& kv : fns) {
            const Fn& f = kv.second;
            for (size_t i = f.b; i < f.e; ++i) {
                if (t[i].k != T_ID) continue;
                const std::string& w = t[i].s;
                if (int_types.count(w) && t[i + 1].k == T_ID && !is(t[i + 1], "(")) {
                    size_t k = i + 1;
                    for (;;) {
                        if (t[k].k != T_ID) break;
                        Decl* d = nullptr;
                        for (auto& q : decls) if (q.at == k) d = &q;
                        if (!d) return false;
                        if (controlled[k]) d->var = true;
                        size_t q = k + 1;
                        while (is(t[q], "[")) q = match(t, q, f.e) + 1;
                        if (is(t[q], "=")) {
                            const size_t e = arg_end(q + 1, f.e);
                            if (has_var(q + 1, e)) d->var = true;
                            if (has_pos(q + 1, e)) d->pos = true;
                            mark(i, e);
                            q = e;
                        }
                        if (is(t[q], ",")) { k = q + 1; continue; }
                        break;
                    }
                } else if (names(i) && resolve(i) && (is(t[i - 1], ";") || is(t[i - 1], "{") || is(t[i - 1], "}") || is(t[i - 1], ")"))) {
                    // NAME = e;  NAME += e;  NAME.x = e;  ++ / -- are left alone  -- an assignment to an integer variable
                    Decl* d = resolve(i);
                    size_t q = i + 1;
                    while (is(t[q], ".") || (t[q].k == T_ID && is(t[q - 1], "."))) ++q;
                    if (t[q].k == T_PUNCT && std::string("+-*/%&|^").find(t[q].s[0]) != std::string::npos && is(t[q + 1], "=")) ++q;
                    if (is(t[q], "=") && !is(t[q + 1], "=")) {
                        const size_t e = arg_end(q + 1, f.e);
                        if (has_var(q + 1, e) || controlled[i]) d->var = true;
                        if (has_pos(q + 1, e)) d->pos = true;
                        if (d->pos) mark(i, e);
                    }
                } else if ((w == "imageLoad" || w == "imageStore") && is(t[i + 1], "(") && t[i + 2].k == T_ID && is(t[i + 3], ",")) {
                    const bool load = w == "imageLoad";
                    if (!(load ? in_names.count(t[i + 2].s) : out_names.count(t[i + 2].s))) return false;
                    const size_t e = arg_end(i + 4, f.e);
                    if (has_var(i + 4, e)) return no("line " + std::to_string(t[i].line) + ": a coordinate follows a uniform (parameters change after the graph is created)");
                    if (!load) {
                        const size_t n = coord_at(i + 4);      // a store: in main(), at the invocation's own coordinate
                        if (kv.first != "main" || !n || i + 4 + n != e) return no("line " + std::to_string(t[i].line) + ": a store somewhere else than the invocation's own texel (`ivec2(gl_GlobalInvocationID.xy)` or a variable declared as that), or outside main()");
                    }
                    mark(i, e);
                } else if (w == "imageSize" && is(t[i + 1], "(") && t[i + 2].k == T_ID && is_image(t[i + 2].s) && is(t[i + 3], ")")) {
                    ok[i + 2] = 1;
                } else if (w == "if" && is(t[i + 1], "(") && kv.first == "main") {
                    const size_t close = match(t, i + 1, f.e);
                    size_t k = close + 1;
                    const bool braces = is(t[k], "{");
                    if (braces) ++k;
                    if (!(is(t[k], "return") && is(t[k + 1], ";") && (!braces || is(t[k + 2], "}")))) continue;
                    const size_t q = i + 2;
                    bool guard = false;
                    if (seq(q, {"any", "(", "greaterThanEqual", "("})) {
                        const size_t a = q + 4, n1 = coord_at(a);
                        if (n1 && is(t[a + n1], ",")) {
                            const size_t n2 = size_at(a + n1 + 1);
                            guard = n2 && seq(a + n1 + 1 + n2, {")", ")", ")"}) && a + n1 + 1 + n2 + 2 == close;
                        }
                    } else {
                        auto half = [&](size_t a, char want, size_t& end) {
                            const size_t n1 = coord_at(a);
                            if (!n1 || !is(t[a + n1], ".") || t[a + n1 + 1].s != std::string(1, want)) return false;
                            const size_t b = a + n1 + 2;
                            if (!(is(t[b], ">") && is(t[b + 1], "=") && t[b + 1].ws.empty())) return false;
                            const size_t n2 = size_at(b + 2);
                            if (!n2 || !is(t[b + 2 + n2], ".") || t[b + 2 + n2 + 1].s != std::string(1, want)) return false;
                            end = b + 2 + n2 + 2;
                            return true;
                        };
                        for (int order = 0; order < 2 && !guard; ++order) {
                            size_t e1 = 0, e2 = 0;
                            if (half(q, order ? 'y' : 'x', e1) && is(t[e1], "|") && is(t[e1 + 1], "|") && half(e1 + 2, order ? 'x' : 'y', e2) && e2 == close) guard = true;
                        }
                    }
                    if (guard) mark(i, close + 1);
                } else if (fns.count(w) && is(t[i + 1], "(") && w != kv.first) {
                    // a helper call: what an argument carries lands in the parameter (an integer, passed by value)
                    const Fn& h = fns.at(w);
                    const size_t close = match(t, i + 1, f.e);
                    size_t a = i + 2;
                    bool any = false;
                    for (size_t p = 0; a < close; ++p) {
                        const size_t e = arg_end(a, close);
                        Decl* pd = nullptr;
                        if (p < h.pname.size())
                            for (auto& q : decls) if (q.at == h.pat[p] && q.b == h.b) pd = &q;
                        if (has_var(a, e) && pd) pd->var = true;
                        if (has_pos(a, e)) {
                            if (!pd || h.byref[p]) return no("line " + std::to_string(t[a].line) + ": the position is handed to `" + w + "` in a parameter that is not an integer passed by value");
                            pd->pos = true;
                            mark(a, e);
                            any = true;
                        }
                        a = e + 1;
                    }
                    if (any && int_types.count(h.ret)) int_returning.insert(w);
                }
            }
        }
        if (flags() == before && round > 0) break;
    }
    // a variable the position AND a uniform reach cannot be vouched for
    for (const auto& d : decls)
        if (d.pos && d.var) return no("line " + std::to_string(t[d.at].line) + ": `" + d.name + "` follows both the position and a uniform");
    // the verdict: the position outside every allowed place?  an image variable that is not the first argument of an image function?
    for (const auto& kv : fns)
        for (size_t i = kv.second.b; i < kv.second.e; ++i) {
            if (!names(i)) continue;
            if (is_image(t[i].s) && !(i >= 2 && is(t[i - 1], "(") && (t[i - 2].s == "imageLoad" || t[i - 2].s == "imageStore" || t[i - 2].s == "imageSize"))) return false;
            if (pos_at(i) && !ok[i]) return no("line " + std::to_string(t[i].line) + ": `" + t[i].s + "` carries the position or the frame size into something else than an integer coordinate, the frame guard or the coordinate of a load / of the store");
        }
    {
        std::vector<char> inside(N, 0);
        for (const auto& kv : fns)
            for (size_t i = kv.second.b; i < kv.second.e; ++i) inside[i] = 1;
        for (size_t i = 0; i < N; ++i)
            if (!inside[i] && t[i].k == T_ID && (t[i].s == "gl_GlobalInvocationID" || t[i].s == "imageSize" || t[i].s == "imageLoad" || t[i].s == "imageStore")) return false;
    }
    return true;
}

}  // namespace

bool glsl_translate(const std::string& type, const std::string& text, const std::string& ident, GlslShader& out, std::string& err)
{
    out = GlslShader();
    out.source = text;
    Translator tr(type, out);
    try {
        tr.run();
    } catch (const Fail& f) {
        err = type + ".comp:" + std::to_string(f.line) + ": " + f.msg;
        out = GlslShader();
        return false;
    }
    std::string s = "\nnamespace rfglsl { namespace " + ident + " {\n";
    s += tr.hoisted;
    s += "template <class RfgPx> struct RfgShader {\n"
         "    uvec3 gl_NumWorkGroups, gl_WorkGroupID, gl_LocalInvocationID, gl_GlobalInvocationID;\n"
         "    uint gl_LocalInvocationIndex;\n"
         "    const uvec3 gl_WorkGroupSize = uvec3{" + std::to_string(out.lx) + "u, " + std::to_string(out.ly) + "u, " + std::to_string(out.lz) + "u};\n";
    s += tr.members;
    int n_read = 0;
    for (const auto& im : out.images) n_read += im.writeonly ? 0 : 1;
    s += "    typedef RfgPx PxT;\n";
    s += "    RFG void rfg_bind(const GlslFrame& rfg_f, const GlslImage* rfg_img, void* const* rfg_buf, const unsigned char* rfg_ubo)\n    {\n        (void)rfg_f; (void)rfg_img; (void)rfg_buf; (void)rfg_ubo;\n" + tr.bind_ubo + tr.bind + "    }\n";
    std::string pin, pout;
    try {
        out.point = point_shader(text, out, pin, pout);
    } catch (const Fail&) {      // (a bracket the analysis could not pair: not recognised, that is all)
        out.point = false;
    }
    try {
        out.stencil = !out.point && stencil_shader(text, out, out.stencil_why);
        if (out.stencil || out.point) out.stencil_why.clear();
    } catch (const Fail&) {
        out.stencil = false;
    }
    out.box = out.stencil && out.radius == 1 && out.images.size() == 2;      // (a stencil has read-only and write-only images only: one of each)
    if (out.box) {
        const std::string& bi = out.images[out.images[0].readonly ? 0 : 1].name, & bo = out.images[out.images[0].readonly ? 1 : 0].name;
        s += "    template <class RfgN> RFG void rfg_bind_box(const unsigned char* rfg_ubo, const RfgN& rfg_n)\n    {\n        (void)rfg_ubo;\n" + tr.bind_ubo + "        " + bi + ".n = rfg_n;\n        " + bo +
             ".n = rfg_n;\n        " + bo + ".value = vec4{0.0f, 0.0f, 0.0f, 0.0f};\n    }\n    RFG vec4 rfg_box_result() const { return " + bo + ".value; }\n";
    }
    if (out.point) {
        // the shader as a row stage: its image variables hold one texel (image2D<PointPx>, rf_glsl_dev.h)
        s += "    RFG void rfg_bind_point(const unsigned char* rfg_ubo, vec4 rfg_c)\n    {\n        (void)rfg_ubo;\n" + tr.bind_ubo + "        " + pout + ".value = vec4{0.0f, 0.0f, 0.0f, 0.0f};\n        " + pin +
             ".value = rfg_c;\n    }\n    RFG vec4 rfg_result() const { return " + pout + ".value; }\n";
    }
    std::vector<std::string> win_in, win_out;
    if (out.stencil) {
        // the shader on the window kernel: its readable images are windows around the invocation, its written images one texel each
        std::string b = "    template <class RfgW> RFG void rfg_bind_win(const unsigned char* rfg_ubo, const RfgW* rfg_in)\n    {\n        (void)rfg_ubo; (void)rfg_in;\n" + tr.bind_ubo;
        std::string o = "    RFG vec4 rfg_out(int rfg_o) const\n    {\n";
        for (const auto& im : out.images) {
            if (im.readonly) { b += "        " + im.name + ".w = &rfg_in[" + std::to_string(win_in.size()) + "];\n"; win_in.push_back(im.name); }
            else { b += "        " + im.name + ".value = vec4{0.0f, 0.0f, 0.0f, 0.0f};\n"; o += "        if (rfg_o == " + std::to_string(win_out.size()) + ") return " + im.name + ".value;\n"; win_out.push_back(im.name); }
        }
        s += b + "    }\n" + o + "        return vec4{0.0f, 0.0f, 0.0f, 0.0f};\n    }\n";
    }
    s += "#line 1 \"" + type + ".comp\"\n";
    s += emit(tr.v, 0, tr.v.size());
    s += "\n};\n";
    for (const auto& m : tr.cx.macros) s += "#undef " + m + "\n";
    s += "struct RfgInfo {\n    static constexpr int LX = " + std::to_string(out.lx) + ", LY = " + std::to_string(out.ly) + ", LZ = " + std::to_string(out.lz) + ", NIMG = " + std::to_string(out.images.size()) +
         ", NBUF = " + std::to_string(out.ssbos.size()) + ", UBO = " + std::to_string(out.ubo_bytes) + ", RADIUS = " + std::to_string(out.radius < 0 ? 0 : out.radius) + ", NREAD = " + std::to_string(n_read) + ";\n    static constexpr bool GROUPED = " +
         (out.grouped ? "true" : "false") + ";\n};\n} }\n";
    if (out.point) {
        const std::string n = std::to_string(out.ubo_bytes > 0 ? out.ubo_bytes : 1);
        s += "#ifdef RFGLSL_KERNEL\nnamespace rfglsl { namespace " + ident + " {\nstruct RfgParams { unsigned char b[" + n + "]; };\nstruct RfgStage {\n    typedef RfgParams P;\n    static constexpr int R = 0;\n"
             "    template <class Q> static RFG rf::f4 point(const Q& p, rf::f4 c)\n    {\n        RfgShader<PointPx> s;\n        s.gl_NumWorkGroups = s.gl_WorkGroupID = s.gl_LocalInvocationID = s.gl_GlobalInvocationID = uvec3{0u, 0u, 0u};\n"
             "        s.gl_LocalInvocationIndex = 0u;\n        s.rfg_bind_point(p.b, vec4{c.x, c.y, c.z, c.w});\n        s.main();\n        const vec4 r = s.rfg_result();\n        return make_float4(r.x, r.y, r.z, r.w);\n    }\n"
             "    template <class Q> static RFG rf::f4 box(const Q&, const rf::f4 (&n)[3][3]) { return n[1][1]; }\n};\n} }\n"
             "namespace rfuser { namespace " + ident + " { typedef rfglsl::" + ident + "::RfgStage Stage; } }\n#endif\n";
    }
    if (out.box) {
        const std::string n = std::to_string(out.ubo_bytes > 0 ? out.ubo_bytes : 1);
        s += "#ifdef RFGLSL_KERNEL\nnamespace rfglsl { namespace " + ident + " {\nstruct RfgBParams { unsigned char b[" + n + "]; };\nstruct RfgBStage {\n    typedef RfgBParams P;\n    static constexpr int R = 1;\n"
             "    template <class Q> static RFG rf::f4 point(const Q&, rf::f4 c) { return c; }\n"
             "    template <class Q> static RFG rf::f4 box(const Q& p, const rf::f4 (&n)[3][3])\n    {\n        RfgShader<BoxPx> s;\n        s.gl_NumWorkGroups = uvec3{1u, 1u, 1u};\n        s.gl_WorkGroupID = uvec3{0u, 0u, 0u};\n"
             "        s.gl_LocalInvocationID = s.gl_GlobalInvocationID = uvec3{1u, 1u, 0u};\n        s.gl_LocalInvocationIndex = 0u;\n        s.rfg_bind_box(p.b, n);\n        s.main();\n"
             "        const vec4 r = s.rfg_box_result();\n        return make_float4(r.x, r.y, r.z, r.w);\n    }\n};\n} }\n"
             "namespace rfuser { namespace " + ident + " { typedef rfglsl::" + ident + "::RfgBStage BStage; } }\n#endif\n";
    }
    if (out.stencil) {
        const std::string n = std::to_string(out.ubo_bytes > 0 ? out.ubo_bytes : 1), R = std::to_string(out.radius);
        s += "#ifdef RFGLSL_KERNEL\nnamespace rfglsl { namespace " + ident + " {\nstruct RfgWParams { unsigned char b[" + n + "]; };\nstruct RfgWStage {\n    typedef RfgWParams P;\n    static constexpr int R = " + R +
             ", NI = " + std::to_string(win_in.size()) + ", NO = " + std::to_string(win_out.size()) + ", FILL = 0;\n"
             "    template <class Q, class In> static RFG void node(const Q& p, const In (&in)[NI], rf::f4 (&out)[NO], const float*)\n    {\n        RfgShader<WinPx<In, R>> s;\n"
             "        s.gl_NumWorkGroups = uvec3{1u, 1u, 1u};\n        s.gl_WorkGroupID = uvec3{0u, 0u, 0u};\n        s.gl_LocalInvocationID = s.gl_GlobalInvocationID = uvec3{" + R + "u, " + R + "u, 0u};\n"
             "        s.gl_LocalInvocationIndex = 0u;\n        s.rfg_bind_win(p.b, in);\n        s.main();\n#pragma unroll\n        for (int o = 0; o < NO; ++o) { const vec4 r = s.rfg_out(o); out[o] = make_float4(r.x, r.y, r.z, r.w); }\n    }\n};\n} }\n"
             "namespace rfuser { namespace " + ident + " { typedef rfglsl::" + ident + "::RfgWStage Stage; } }\n#endif\n";
    }
    out.source = s;
    return true;
}

static std::string json_escape(const std::string& x)
{
    std::string o;
    for (char c : x) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if (c == '\n') o += "\\n";
        else o += c;
    }
    return o;
}

std::string glsl_reflection_json(const GlslShader& s)
{
    auto q = [](const std::string& x) { return "\"" + x + "\""; };
    std::string j = std::string("{\"point\": ") + (s.point ? "true" : "false") + ", \"stencil\": " + (s.stencil ? "true" : "false") + ", \"box\": " + (s.box ? "true" : "false") + ", \"stencil_why_not\": \"" + json_escape(s.stencil_why) + "\", \"local_size\": [" + std::to_string(s.lx) + ", " + std::to_string(s.ly) + ", " + std::to_string(s.lz) + "], \"grouped\": " + (s.grouped ? "true" : "false") +
                    ", \"radius\": " + std::to_string(s.radius) + ", \"uniform_bytes\": " + std::to_string(s.ubo_bytes) + ", \"images\": [";
    for (size_t i = 0; i < s.images.size(); ++i)
        j += std::string(i ? ", " : "") + "{\"name\": " + q(s.images[i].name) + ", \"binding\": " + std::to_string(s.images[i].binding) + ", \"readonly\": " + (s.images[i].readonly ? "true" : "false") +
             ", \"writeonly\": " + (s.images[i].writeonly ? "true" : "false") + (s.images[i].sampled ? ", \"sampled\": true" : "") + "}";
    auto blocks = [&](const std::vector<GlslBlock>& bl) {
        std::string o;
        for (size_t i = 0; i < bl.size(); ++i) {
            const GlslBlock& b = bl[i];
            o += std::string(i ? ", " : "") + "{\"type_name\": " + q(b.type_name) + ", \"instance\": " + q(b.instance) + ", \"binding\": " + std::to_string(b.binding) + ", \"bytes\": " + std::to_string(b.bytes) +
                 ", \"base\": " + std::to_string(b.ubo_base) + ", \"readonly\": " + (b.readonly ? "true" : "false") + ", \"writeonly\": " + (b.writeonly ? "true" : "false") + ", \"members\": [";
            for (size_t k = 0; k < b.members.size(); ++k) {
                const GlslMember& m = b.members[k];
                std::string dims;
                for (size_t d = 0; d < m.dims.size(); ++d) dims += std::string(d ? ", " : "") + std::to_string(m.dims[d]);
                o += std::string(k ? ", " : "") + "{\"name\": " + q(m.name) + ", \"base\": " + q(std::string(1, m.base)) + ", \"comps\": " + std::to_string(m.comps) + ", \"cols\": " + std::to_string(m.cols) +
                     ", \"dims\": [" + dims + "], \"offset\": " + std::to_string(m.offset) + ", \"stride\": " + std::to_string(m.stride) + ", \"bytes\": " + std::to_string(m.bytes) + "}";
            }
            o += "]}";
        }
        return o;
    };
    j += "], \"uniform_blocks\": [" + blocks(s.ubos) + "], \"storage_blocks\": [" + blocks(s.ssbos) + "]}";
    return j;
}

}  // namespace rf
