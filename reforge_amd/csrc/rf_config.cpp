// rf_config.cpp -- see rf_config.h.  Host only; no GPU involved.
#include "rf_config.h"

#include <cctype>
#include <cstdio>
#include <cstring>

namespace rf {

const char* const kFileInput = "rf:file-input";
const char* const kFinalOutput = "rf:final-output";

const std::string& Config::type_of(const std::string& node) const
{
    auto it = pipeline_instances.find(node);
    return it != pipeline_instances.end() ? it->second.pipeline_type : node;
}

const std::map<std::string, std::string>& Config::params_of(const std::string& node) const
{
    static const std::map<std::string, std::string> empty;
    auto it = pipeline_instances.find(node);
    return it != pipeline_instances.end() ? it->second.parameters : empty;
}

namespace {

// Terminals of config_grammar.lalrpop.  LALRPOP's generated lexer skips whitespace,
// takes the longest match, and prefers a literal over a regex on a tie.
enum Tok {
    T_ARROW, T_EMPTY_BRACES, T_LBRACE, T_RBRACE, T_COLON, T_COMMA, T_TRUE, T_FALSE,   // literals
    T_LINE_COMMENT,   // r"//[^\n\r]*[\n\r]*"                                   :24
    T_BLOCK_COMMENT,  // r"/\*([^\*]*\*+[^\*/])*([^\*]*\*+|[^\*])*\*/"          :27
    T_INT,            // r"[0-9]+"                                              :75
    T_DEC,            // r"-?[0-9]+\.[0-9]+"                                    :76
    T_STR             // r"[a-zA-Z_][a-zA-Z0-9_-]+"  (two characters minimum)   :81
};

struct Token {
    Tok kind;
    std::string text;
    size_t pos;
};

const char* tok_name(Tok t)
{
    switch (t) {
        case T_ARROW: return "'->'";
        case T_EMPTY_BRACES: return "'{}'";
        case T_LBRACE: return "'{'";
        case T_RBRACE: return "'}'";
        case T_COLON: return "':'";
        case T_COMMA: return "','";
        case T_TRUE: return "'true'";
        case T_FALSE: return "'false'";
        case T_INT: return "'[0-9]+'";
        case T_DEC: return "'-?[0-9]+\\.[0-9]+'";
        case T_STR: return "'[a-zA-Z_][a-zA-Z0-9_-]+'";
        default: return "comment";
    }
}

bool is_digit(char c) { return c >= '0' && c <= '9'; }
bool is_ident_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
bool is_ident_rest(char c) { return is_ident_start(c) || is_digit(c) || c == '-'; }

size_t match_digits(const std::string& s, size_t i)
{
    size_t j = i;
    while (j < s.size() && is_digit(s[j])) ++j;
    return j - i;
}

// line number (1-based) and the text of the line holding byte offset `pos`
void locate(const std::string& s, size_t pos, size_t& line_no, std::string& line)
{
    line_no = 1;
    size_t start = 0;
    for (size_t i = 0; i < pos && i < s.size(); ++i)
        if (s[i] == '\n') { ++line_no; start = i + 1; }
    size_t end = s.find('\n', start);
    line = s.substr(start, end == std::string::npos ? std::string::npos : end - start);
}

bool lex(const std::string& s, std::vector<Token>& out, std::string& err)
{
    static const struct { const char* text; Tok kind; } literals[] = {
        {"->", T_ARROW}, {"{}", T_EMPTY_BRACES}, {"{", T_LBRACE}, {"}", T_RBRACE},
        {":", T_COLON},  {",", T_COMMA},         {"true", T_TRUE}, {"false", T_FALSE}};
    size_t i = 0;
    const size_t n = s.size();
    while (i < n) {
        if (std::isspace((unsigned char)s[i])) { ++i; continue; }
        size_t best = 0;
        Tok kind = T_STR;
        for (const auto& l : literals) {
            size_t len = std::strlen(l.text);
            if (s.compare(i, len, l.text) == 0 && len > best) { best = len; kind = l.kind; }
        }
        auto offer = [&](size_t len, Tok k) { if (len > best) { best = len; kind = k; } };   // literals win ties
        if (s.compare(i, 2, "//") == 0) {
            size_t j = i + 2;
            while (j < n && s[j] != '\n' && s[j] != '\r') ++j;
            while (j < n && (s[j] == '\n' || s[j] == '\r')) ++j;
            offer(j - i, T_LINE_COMMENT);
        }
        if (s.compare(i, 2, "/*") == 0) {
            // The regex accepts every string that starts "/*" and ends "*/" (its second
            // group generates arbitrary text), and the lexer keeps the LONGEST match:
            // the comment runs to the last "*/" of the input.
            size_t j = s.rfind("*/");
            if (j != std::string::npos && j >= i + 2) offer(j + 2 - i, T_BLOCK_COMMENT);
        }
        {
            size_t d = match_digits(s, i);
            if (d > 0) offer(d, T_INT);
            size_t k = i + (s[i] == '-' ? 1 : 0);
            size_t d1 = match_digits(s, k);
            if (d1 > 0 && k + d1 < n && s[k + d1] == '.') {
                size_t d2 = match_digits(s, k + d1 + 1);
                if (d2 > 0) offer(k + d1 + 1 + d2 - i, T_DEC);
            }
        }
        if (is_ident_start(s[i])) {
            size_t j = i + 1;
            while (j < n && is_ident_rest(s[j])) ++j;
            if (j - i >= 2) offer(j - i, T_STR);
        }
        if (best == 0) {
            size_t line_no;
            std::string line;
            locate(s, i, line_no, line);
            err = "Invalid token '" + std::string(1, s[i]) + "' at line " + std::to_string(line_no) + ": " + line;
            return false;
        }
        out.push_back({kind, s.substr(i, best), i});
        i += best;
    }
    return true;
}

// ast.rs:5-17
struct AstPipeline {
    std::string name, pipeline_type;
    std::map<std::string, std::string> parameters;
    std::vector<std::pair<std::string, std::string>> fields;   // the same in source order, duplicates kept (parse_syntax)
};
struct AstExpr {
    bool is_graph = false;
    bool is_comment = false;                                  // Expr::Ignore: kept only when the parser is asked to (parse_syntax)
    std::string comment;
    std::vector<std::pair<std::string, std::string>> graph;   // (name, descriptor); "" = None
    std::vector<bool> has_desc;
    AstPipeline pipeline;
};

struct Parser {
    const std::string& src;
    const std::vector<Token>& t;
    size_t i = 0;
    std::string err;
    bool keep_comments = false;

    bool fail_at(size_t at, const char* expected)
    {
        if (at >= t.size()) {
            err = std::string("Error while parsing: unexpected end of input, expected ") + expected;
            return false;
        }
        size_t line_no;
        std::string line;
        locate(src, t[at].pos, line_no, line);
        err = "Unrecognized token '" + t[at].text + "' at line " + std::to_string(line_no) + ": " + line +
              "\nExpected to find: " + expected;
        return false;
    }
    bool peek(Tok k, size_t at) const { return at < t.size() && t[at].kind == k; }
    bool need(Tok k, size_t at) { return peek(k, at) ? true : fail_at(at, tok_name(k)); }

    // ExprList, config_grammar.lalrpop:7-14
    bool parse(std::vector<AstExpr>& out)
    {
        if (t.empty()) {
            err = "Error while parsing: unexpected end of input";
            return false;
        }
        while (i < t.size()) {
            if (t[i].kind == T_LINE_COMMENT || t[i].kind == T_BLOCK_COMMENT) {   // Expr::Ignore
                if (keep_comments) {
                    AstExpr c;
                    c.is_comment = true;
                    c.comment = t[i].text;
                    out.push_back(c);
                }
                ++i;
                continue;
            }
            if (!need(T_STR, i)) return false;
            std::string name = t[i].text;
            ++i;
            std::string desc;
            bool has_desc = false;
            if (peek(T_COLON, i)) {
                if (!need(T_STR, i + 1)) return false;
                std::string second = t[i + 1].text;
                i += 2;
                if (peek(T_LBRACE, i) || peek(T_EMPTY_BRACES, i)) {
                    // PipelineField: name ":" type PipelineParams     :44-51
                    AstExpr e;
                    e.pipeline.name = name;
                    e.pipeline.pipeline_type = second;
                    if (peek(T_EMPTY_BRACES, i)) {
                        ++i;
                    } else {
                        ++i;
                        for (;;) {   // ParamFieldList :53-64
                            if (!need(T_STR, i) || !need(T_COLON, i + 1)) return false;
                            if (!(peek(T_INT, i + 2) || peek(T_DEC, i + 2) || peek(T_TRUE, i + 2) || peek(T_FALSE, i + 2)))
                                return fail_at(i + 2, "'[0-9]+', '-?[0-9]+\\.[0-9]+', 'true', 'false'");
                            e.pipeline.parameters[t[i].text] = t[i + 2].text;   // HashMap::insert: the last one wins
                            e.pipeline.fields.push_back({t[i].text, t[i + 2].text});
                            i += 3;
                            if (peek(T_COMMA, i)) { ++i; continue; }
                            if (!need(T_RBRACE, i)) return false;
                            ++i;
                            break;
                        }
                    }
                    out.push_back(e);
                    continue;
                }
                desc = second;
                has_desc = true;
            }
            // GraphExpr: member ("->" member)+     :30-42
            AstExpr e;
            e.is_graph = true;
            e.graph.push_back({name, desc});
            e.has_desc.push_back(has_desc);
            if (!need(T_ARROW, i)) return false;
            while (peek(T_ARROW, i)) {
                if (!need(T_STR, i + 1)) return false;
                std::string mname = t[i + 1].text, mdesc;
                bool mhas = false;
                i += 2;
                if (peek(T_COLON, i)) {
                    if (!need(T_STR, i + 1)) return false;
                    mdesc = t[i + 1].text;
                    mhas = true;
                    i += 2;
                }
                e.graph.push_back({mname, mdesc});
                e.has_desc.push_back(mhas);
            }
            out.push_back(e);
        }
        return true;
    }
};

bool only_whitespace(const std::string& s)
{
    for (char c : s)
        if (!std::isspace((unsigned char)c)) return false;
    return true;
}

}  // namespace

bool parse_config(const std::string& text, bool expects_input, Config& config, std::string& err)
{
    config = Config();
    if (only_whitespace(text)) {   // config.rs:99-102
        err = "Empty configuration given to parse";
        return false;
    }
    std::vector<Token> toks;
    if (!lex(text, toks, err)) return false;
    std::vector<AstExpr> exprs;
    Parser p{text, toks, 0, std::string()};
    if (!p.parse(exprs)) {
        err = p.err;
        return false;
    }

    bool found_input = false, found_output = false;
    for (const AstExpr& e : exprs) {
        if (!e.is_graph) {   // config.rs:191-195
            config.pipeline_instances[e.pipeline.name] = {e.pipeline.pipeline_type, e.pipeline.parameters};
            continue;
        }
        const auto& graph = e.graph;   // config.rs:149-190
        for (size_t i = 0; i < graph.size(); ++i) {
            const std::string& name = graph[i].first;
            if (name == "input") { found_input = true; continue; }
            if (name == "output") { found_output = true; continue; }
            GraphPipeline& info = config.graph_pipelines[name];
            if (i > 0) {
                const std::string& prev = graph[i - 1].first;
                std::string descriptor = e.has_desc[i] ? graph[i].second : "input_image";
                std::string resource = prev == "input"
                                           ? std::string(kFileInput)
                                           : prev + ":" + (e.has_desc[i - 1] ? graph[i - 1].second : "output_image");
                info.inputs.push_back({resource, descriptor});
            }
            if (i + 1 < graph.size()) {
                const std::string& next = graph[i + 1].first;
                std::string descriptor = e.has_desc[i] ? graph[i].second : "output_image";
                std::string resource = next == "output" ? std::string(kFinalOutput) : name + ":" + descriptor;
                info.outputs.push_back({resource, descriptor});
            }
        }
    }
    if (config.graph_pipelines.empty()) {   // config.rs:200
        err = "Configuration had an empty graph";
        return false;
    }
    if (found_input && !expects_input) {    // config.rs:201
        err = "Found 'input' in pipeline configuration but no input image was specified";
        return false;
    }
    if (!found_output) {                    // config.rs:202
        err = "'output' is never used in the pipeline configuration";
        return false;
    }
    return true;
}

// The syntax tree of a config text as JSON -- what LALRPOP's parser hands config::parse (config.rs:105), before any of its
// checks: {"exprs": [["pipeline", name, type, [[key, value], ...]] | ["graph", [[name, descriptor | null], ...]] |
// ["comment", text]]}, parameters in source order with duplicates.  tests/test_grammar_fixtures.py holds it to vectors derived
// from the reference's grammar file itself (tests/golden/make_grammar_fixtures.py).
static void json_string(const std::string& s, std::string& out)
{
    out += '"';
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
        else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof(b), "\\u%04x", c); out += b; }
        else out += (char)c;
    }
    out += '"';
}

bool parse_syntax(const std::string& text, std::string& json, std::string& err)
{
    std::vector<Token> toks;
    if (!lex(text, toks, err)) return false;
    std::vector<AstExpr> exprs;
    Parser p{text, toks, 0, std::string(), true};
    if (!p.parse(exprs)) {
        err = p.err;
        return false;
    }
    json = "{\"exprs\":[";
    for (size_t k = 0; k < exprs.size(); ++k) {
        const AstExpr& e = exprs[k];
        if (k) json += ',';
        if (e.is_comment) {
            json += "[\"comment\",";
            json_string(e.comment, json);
            json += ']';
        } else if (e.is_graph) {
            json += "[\"graph\",[";
            for (size_t m = 0; m < e.graph.size(); ++m) {
                if (m) json += ',';
                json += '[';
                json_string(e.graph[m].first, json);
                json += ',';
                if (e.has_desc[m]) json_string(e.graph[m].second, json);
                else json += "null";
                json += ']';
            }
            json += "]]";
        } else {
            json += "[\"pipeline\",";
            json_string(e.pipeline.name, json);
            json += ',';
            json_string(e.pipeline.pipeline_type, json);
            json += ",[";
            for (size_t m = 0; m < e.pipeline.fields.size(); ++m) {
                if (m) json += ',';
                json += '[';
                json_string(e.pipeline.fields[m].first, json);
                json += ',';
                json_string(e.pipeline.fields[m].second, json);
                json += ']';
            }
            json += "]]";
        }
    }
    json += "]}";
    return true;
}

bool single_node_config(const std::string& type_name, bool expects_input, Config& out, std::string& err)
{
    // config.rs:81-84
    std::string text = expects_input ? "input -> " + type_name + " -> output" : type_name + " -> output";
    return parse_config(text, expects_input, out, err);
}

}  // namespace rf
