// rf_user.cpp -- see rf_user.h.  Host only.
#include "rf_user.h"
#include "rf_glsl.h"

#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>

namespace rf {

namespace {

std::mutex g_mu;
std::string g_dir;
bool g_files_first = false;
std::deque<UserStage> g_stages;                        // stable addresses: NodeType pointers are handed out
std::map<std::string, int> g_latest;                   // type name -> newest entry
std::map<const NodeType*, int> g_by_type;

uint64_t fnv1a(const std::string& s)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

std::string strip_comments(const std::string& t)
{
    std::string o;
    for (size_t i = 0; i < t.size();) {
        if (t.compare(i, 2, "//") == 0) { while (i < t.size() && t[i] != '\n') ++i; }
        else if (t.compare(i, 2, "/*") == 0) { size_t e = t.find("*/", i + 2); i = e == std::string::npos ? t.size() : e + 2; o += ' '; }
        else o += t[i++];
    }
    return o;
}

bool ident_ok(const std::string& s)
{
    if (s.empty() || !(std::isalpha((unsigned char)s[0]) || s[0] == '_')) return false;
    for (char c : s)
        if (!(std::isalnum((unsigned char)c) || c == '_')) return false;
    return true;
}

long long file_mtime_ns(const std::string& path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return -1;
    return (long long)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
}

}  // namespace

bool parse_user_stage(const std::string& type, const std::string& text, UserStage& out, std::string& err)
{
    out = UserStage();
    out.type_name = type;
    out.text = text;
    char hx[32];
    std::snprintf(hx, sizeof(hx), "u_%016llx", (unsigned long long)fnv1a(type + "\n" + text));
    out.ident = hx;
    const std::string t = strip_comments(text);
    // static constexpr int RADIUS = N;
    size_t r = t.find("RADIUS");
    if (r == std::string::npos) { err = type + ".stage.hip: no `static constexpr int RADIUS = 0 | 1;`"; return false; }
    size_t eq = t.find('=', r);
    if (eq == std::string::npos) { err = type + ".stage.hip: RADIUS has no value"; return false; }
    out.radius = std::atoi(t.c_str() + eq + 1);
    if (out.radius < 0 || out.radius > kMaxRadius) { err = type + ".stage.hip: RADIUS must be 0 (point op), 1 (3x3 neighbourhood) or 2.." + std::to_string(kMaxRadius) + " (a window read through Window::at)"; return false; }
    // struct Params { <type> <name>; ... };
    size_t sp = t.find("struct Params");
    if (sp == std::string::npos) { err = type + ".stage.hip: no `struct Params { ... };`"; return false; }
    size_t ob = t.find('{', sp), cb = ob == std::string::npos ? ob : t.find('}', ob);
    if (ob == std::string::npos || cb == std::string::npos) { err = type + ".stage.hip: struct Params is not closed"; return false; }
    std::stringstream body(t.substr(ob + 1, cb - ob - 1));
    std::string decl;
    int off = 0, align = 1;
    while (std::getline(body, decl, ';')) {
        std::stringstream ds(decl);
        std::string ty, name, extra;
        if (!(ds >> ty)) continue;                    // whitespace only
        if (!(ds >> name) || (ds >> extra)) { err = type + ".stage.hip: cannot read the Params member `" + decl + "` (expected `float|int|bool name;`)"; return false; }
        UserParam p;
        if (ty == "float") { p.type = PARAM_F32; p.size = 4; }
        else if (ty == "int") { p.type = PARAM_I32; p.size = 4; }
        else if (ty == "bool") { p.type = PARAM_BOOL; p.size = 1; }
        else { err = type + ".stage.hip: Params member `" + name + "` has type `" + ty + "`; uniform members are float, int or bool (render.rs:169-185)"; return false; }
        if (!ident_ok(name)) { err = type + ".stage.hip: `" + name + "` is not a member name"; return false; }
        for (const auto& q : out.params)
            if (q.name == name) { err = type + ".stage.hip: Params member `" + name + "` is declared twice"; return false; }
        off = (off + p.size - 1) / p.size * p.size;
        p.offset = off;
        off += p.size;
        align = std::max(align, p.size);
        p.name = name;
        out.params.push_back(p);
    }
    if (out.params.size() > 14 || off > 56) { err = type + ".stage.hip: Params is limited to 56 bytes"; return false; }
    out.params_size = std::max(1, (off + align - 1) / align * align);
    if (t.find("apply") == std::string::npos) { err = type + ".stage.hip: no `RF_STAGE f4 apply(const Params&, ...)`"; return false; }
    // RF_INPUTS(a, b, ...) / RF_OUTPUTS(c, ...): the image variables of a node with a kernel of its own
    out.inputs = {"input_image"};                      // passthrough.comp:4-5
    out.outputs = {"output_image"};
    for (int side = 0; side < 2; ++side) {
        const char* kw = side == 0 ? "RF_INPUTS" : "RF_OUTPUTS";
        size_t at = t.find(kw);
        if (at == std::string::npos) continue;
        if (t.find(kw, at + 1) != std::string::npos) { err = type + ".stage.hip: " + kw + " is declared twice"; return false; }
        size_t op = t.find('(', at), cp = op == std::string::npos ? op : t.find(')', op);
        if (op == std::string::npos || cp == std::string::npos) { err = type + ".stage.hip: " + kw + "(...) is not closed"; return false; }
        std::vector<std::string> names;
        std::stringstream list(t.substr(op + 1, cp - op - 1));
        std::string item;
        while (std::getline(list, item, ',')) {
            std::stringstream is(item);
            std::string name, extra;
            if (!(is >> name) || (is >> extra) || !ident_ok(name)) { err = type + ".stage.hip: " + kw + ": `" + item + "` is not an image variable name"; return false; }
            if (std::find(names.begin(), names.end(), name) != names.end()) { err = type + ".stage.hip: " + kw + " lists `" + name + "` twice"; return false; }
            names.push_back(name);
        }
        if (names.empty() || names.size() > (size_t)kMaxUserImages) { err = type + ".stage.hip: " + kw + " takes 1 to " + std::to_string(kMaxUserImages) + " image names"; return false; }
        (side == 0 ? out.inputs : out.outputs) = names;
        out.multi = true;
    }
    for (int side = 0; side < 2; ++side) {
        const char* kw = side == 0 ? "RF_BUFFER_IN" : "RF_BUFFER_OUT";
        size_t at = t.find(kw);
        if (at == std::string::npos) continue;
        if (t.find(kw, at + 1) != std::string::npos) { err = type + ".stage.hip: " + kw + " is declared twice (one buffer read, one written)"; return false; }
        size_t op = t.find('(', at), cp = op == std::string::npos ? op : t.find(')', op);
        if (op == std::string::npos || cp == std::string::npos) { err = type + ".stage.hip: " + kw + "(...) is not closed"; return false; }
        const std::string inside = t.substr(op + 1, cp - op - 1);
        const size_t comma = inside.find(',');
        std::stringstream ns(inside.substr(0, comma)), cs(comma == std::string::npos ? std::string() : inside.substr(comma + 1));
        UserStage::Buffer b;
        std::string extra;
        long count = 0;
        if (!(ns >> b.name) || (ns >> extra) || !ident_ok(b.name) || !(cs >> count) || (cs >> extra) || count < 1 || count > 65536) {
            err = type + ".stage.hip: " + kw + "(BlockTypeName, number of floats 1..65536), got `" + inside + "`";
            return false;
        }
        b.count = (int)count;
        b.bytes = (size_t)count * sizeof(float);
        (side == 0 ? out.buf_in : out.buf_out).push_back(b);
        out.multi = true;
    }
    if (!out.buf_in.empty() && !out.buf_out.empty() && out.buf_in[0].name == out.buf_out[0].name) {
        err = type + ".stage.hip: " + out.buf_in[0].name + " is both read and written; a node reads one buffer and fills another";
        return false;
    }
    if (!out.buf_out.empty() && t.find("fill") == std::string::npos) { err = type + ".stage.hip: RF_BUFFER_OUT needs `RF_STAGE float fill(const Params&, int i)`"; return false; }
    if (out.radius >= 2) out.multi = true;            // no row stage of that radius: a node with a kernel of its own, its inputs read through windows
    if (out.multi && out.radius > 0) {
        for (const auto& o : out.outputs)
            if (std::find(out.inputs.begin(), out.inputs.end(), o) != out.inputs.end()) { err = type + ".stage.hip: `" + o + "` is read through a window (RADIUS " + std::to_string(out.radius) + ") and cannot be written in place"; return false; }
    }
    for (size_t i = 0; i < out.inputs.size(); ++i) out.in_binding.push_back((int)i);
    int next = (int)out.inputs.size();
    for (const auto& o : out.outputs) {
        auto it = std::find(out.inputs.begin(), out.inputs.end(), o);
        out.out_binding.push_back(it != out.inputs.end() ? (int)(it - out.inputs.begin()) : next++);     // same name = same binding = in place
    }
    for (auto& b : out.buf_in) b.binding = next++;
    for (auto& b : out.buf_out) b.binding = next++;
    return true;
}

std::string UserStage::wrapper() const
{
    if (glsl) return glsl_source;
    std::string w = "\nnamespace rfuser { namespace " + ident + " {\nusing rf::f4;\nusing rf::Window;\n#define RF_STAGE static __device__ __forceinline__\n"
                    "#define RF_INPUTS(...) static_assert(true, \"\")\n#define RF_OUTPUTS(...) static_assert(true, \"\")\n"
                    "#define RF_BUFFER_IN(...) static_assert(true, \"\")\n#define RF_BUFFER_OUT(...) static_assert(true, \"\")\n#line 1 \"" +
                    type_name + ".stage.hip\"\n" + text + "\n#undef RF_STAGE\n#undef RF_INPUTS\n#undef RF_OUTPUTS\n#undef RF_BUFFER_IN\n#undef RF_BUFFER_OUT\nstruct Stage {\n    typedef Params P;\n    static constexpr int R = RADIUS;\n"
                    "    static_assert(R == " + std::to_string(radius) + ", \"RADIUS is not the value the host read\");\n"
                    "    static_assert(sizeof(Params) == " + std::to_string(params_size) + ", \"struct Params is not laid out as the host computed\");\n";
    for (const auto& p : params)
        w += "    static_assert(__builtin_offsetof(Params, " + p.name + ") == " + std::to_string(p.offset) + ", \"struct Params is not laid out as the host computed\");\n";
    if (multi) {
        w += "    static constexpr int NI = " + std::to_string(inputs.size()) + ", NO = " + std::to_string(outputs.size()) +
             ", FILL = " + std::to_string(buf_out.empty() ? 0 : buf_out[0].count) + ";\n"
             "    template <class Q, class In> static __device__ __forceinline__ void node(const Q& p, const In (&in)[NI], f4 (&out)[NO], const float* buf) { apply(p, in, out" +
             std::string(buf_in.empty() ? "" : ", buf") + "); }\n";
        if (!buf_out.empty()) w += "    template <class Q> static __device__ __forceinline__ float fill_at(const Q& p, int i) { return fill(p, i); }\n";
        w += "};\n} }\n";
        return w;
    }
    // templates: only the form the file defines is ever instantiated (apply is found by argument-dependent lookup)
    w += "    template <class Q> static __device__ __forceinline__ f4 point(const Q& p, f4 c) { return apply(p, c); }\n"
         "    template <class Q> static __device__ __forceinline__ f4 box(const Q& p, const f4 (&n)[3][3]) { return apply(p, n); }\n};\n} }\n";
    return w;
}

bool parse_glsl_stage(const std::string& type, const std::string& text, UserStage& out, std::string& err)
{
    out = UserStage();
    out.type_name = type;
    out.text = text;
    out.glsl = true;
    out.multi = true;
    char hx[40];
    std::snprintf(hx, sizeof(hx), "g%d_%016llx", kGlslTranslatorVersion, (unsigned long long)fnv1a(type + "\n" + text));
    out.ident = hx;
    GlslShader sh;
    if (!glsl_translate(type, text, out.ident, sh, err)) return false;
    out.glsl_source = sh.source;
    // a point operation on one image: a ROW STAGE of the stream kernel like a {type}.stage.hip of RADIUS 0 -- it fuses with its neighbours
    // (RF_GLSL_NO_FUSE=1: every .comp file stays a node with a kernel of its own, for A/B measurements and tests of that path)
    static const bool no_fuse = [] { const char* e = std::getenv("RF_GLSL_NO_FUSE"); return e && std::atoi(e) != 0; }();
    if ((sh.point || sh.box) && !no_fuse) out.multi = false;
    if (sh.box && !sh.point) out.row_stage = "BStage";      // (Stage is the window stage of the same file)
    static const bool no_window = [] { const char* e = std::getenv("RF_GLSL_NO_WINDOW"); return e && std::atoi(e) != 0; }();
    out.glsl_window = sh.stencil && !no_window;      // (which graphs use it: glsl_wants_window, rf_graph.cpp)
    out.radius = sh.radius < 0 ? 0 : sh.radius;
    out.radius_stated = sh.radius >= 0;
    out.glsl_grouped = sh.grouped;
    out.glsl_groups[0] = sh.lx; out.glsl_groups[1] = sh.ly; out.glsl_groups[2] = sh.lz;
    out.inputs.clear();
    out.outputs.clear();
    for (const auto& im : sh.images) {
        out.glsl_images.push_back(im.name);
        out.glsl_image_binding.push_back(im.binding);
        out.glsl_image_written.push_back(!im.readonly);
        if (!im.writeonly) { out.inputs.push_back(im.name); out.in_binding.push_back(im.binding); }
        if (!im.readonly) { out.outputs.push_back(im.name); out.out_binding.push_back(im.binding); }
    }
    if (out.outputs.empty() && sh.ssbos.empty()) { err = type + ".comp: the shader writes no image and has no storage block: nothing it does can be observed"; return false; }
    if (out.radius > 0)
        for (const auto& im : sh.images)
            if (!im.readonly && !im.writeonly) { err = type + ".comp: `" + im.name + "` is read and written (no readonly / writeonly) by a shader that reads its neighbourhood (#pragma rf radius " + std::to_string(out.radius) + "): a stencil cannot run in place"; return false; }
    for (size_t k = 0; k < sh.ssbos.size(); ++k) {
        const GlslBlock& b = sh.ssbos[k];
        UserStage::Buffer ub;
        ub.name = b.type_name;
        ub.count = b.bytes / 4;
        ub.bytes = (size_t)b.bytes;
        ub.binding = b.binding;
        ub.slot = (int)k;
        if (!b.writeonly) out.buf_in.push_back(ub);
        if (!b.readonly) out.buf_out.push_back(ub);
    }
    out.glsl_buffers = (int)sh.ssbos.size();
    // uniform members the host can set: scalars (render.rs:169-185: FLOAT, INT -- spirv-reflect's flag for both signednesses --, BOOL);
    // every other member stays zero, as there (render.rs:200-203)
    for (const auto& blk : sh.ubos)
        for (const auto& m : blk.members) {
            if (m.comps != 1 || m.cols != 1 || !m.dims.empty()) continue;
            UserParam p;
            p.name = m.name;
            p.type = m.base == 'f' ? PARAM_F32 : (m.base == 'b' ? PARAM_BOOL : PARAM_I32);
            p.offset = blk.ubo_base + m.offset;
            p.size = 4;
            for (const auto& q : out.params)
                if (q.name == p.name) { err = type + ".comp: two uniform members are called `" + p.name + "` (parameters are matched by member name, pipeline_graph.rs:276-292)"; return false; }
            out.params.push_back(p);
        }
    out.params_size = std::max(1, sh.ubo_bytes);
    return true;
}

void set_files_first(bool on)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_files_first = on;
}

bool files_first()
{
    std::lock_guard<std::mutex> lock(g_mu);
    return g_files_first;
}

void set_shader_path(const std::string& dir)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_dir = dir;
}

const std::string& shader_path()
{
    std::lock_guard<std::mutex> lock(g_mu);
    static thread_local std::string copy;
    copy = g_dir;
    return copy;
}

const UserStage* user_stage_for_type(const std::string& type, std::string& err)
{
    std::lock_guard<std::mutex> lock(g_mu);
    err.clear();
    if (g_dir.empty() || !ident_ok(type)) return nullptr;
    std::string path = g_dir + "/" + type + ".stage.hip";
    long long mt = file_mtime_ns(path);
    bool glsl = false;
    if (mt < 0) {      // the reference's own form: {type}.comp (config.rs:59-75)
        path = g_dir + "/" + type + ".comp";
        mt = file_mtime_ns(path);
        glsl = true;
    }
    if (mt < 0) return nullptr;                       // no such filter (Shader::from_path -> None, utils.rs:23)
    auto it = g_latest.find(type);
    if (it != g_latest.end() && g_stages[(size_t)it->second].mtime_ns == mt && g_stages[(size_t)it->second].path == path) return &g_stages[(size_t)it->second];
    std::ifstream f(path, std::ios::binary);
    if (!f) return nullptr;
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (it != g_latest.end() && g_stages[(size_t)it->second].text == text && g_stages[(size_t)it->second].path == path) {
        g_stages[(size_t)it->second].mtime_ns = mt;   // touched, not edited
        return &g_stages[(size_t)it->second];
    }
    UserStage st;
    if (!(glsl ? parse_glsl_stage(type, text, st, err) : parse_user_stage(type, text, st, err))) return nullptr;
    st.path = path;
    st.mtime_ns = mt;
    st.id = (int)g_stages.size();
    g_stages.push_back(std::move(st));
    UserStage& s = g_stages.back();
    s.node_type.name = s.type_name.c_str();
    s.node_type.kind = s.multi ? OP_USERN : OP_USER;
    s.node_type.fixed_radius = s.radius;
    // image variable -> binding (the names point into this entry, which never moves)
    for (size_t i = 0; i < s.inputs.size(); ++i) s.node_type.images.push_back({s.inputs[i].c_str(), s.in_binding[i]});
    for (size_t o = 0; o < s.outputs.size(); ++o)
        if (std::find(s.inputs.begin(), s.inputs.end(), s.outputs[o]) == s.inputs.end()) s.node_type.images.push_back({s.outputs[o].c_str(), s.out_binding[o]});
    for (const auto& b : s.buf_in) s.node_type.buffers.push_back(NodeType::BufferDef{b.name.c_str(), b.binding, b.bytes});
    for (const auto& b : s.buf_out) {
        bool listed = false;
        for (const auto& a : s.buf_in) listed = listed || a.binding == b.binding;      // a block updated in place: one binding, listed once
        if (!listed) s.node_type.buffers.push_back(NodeType::BufferDef{b.name.c_str(), b.binding, b.bytes});
    }
    for (const auto& p : s.params) s.node_type.params.push_back(ParamDef{p.name.c_str(), p.type});
    if (s.glsl && !s.multi) {      // a row stage: the same type as a node with a kernel of its own
        s.node_type_alt = s.node_type;
        s.node_type_alt.kind = OP_USERN;
        s.has_alt = true;
        g_by_type[&s.node_type_alt] = s.id;
    }
    g_latest[type] = s.id;
    g_by_type[&s.node_type] = s.id;
    return &s;
}

namespace { std::set<int> g_no_row_stage; }
const NodeType* user_stage_node_type(const UserStage* u, bool want_node)
{
    std::lock_guard<std::mutex> lock(g_mu);
    return u->has_alt && (want_node || g_no_row_stage.count(u->id)) ? &u->node_type_alt : &u->node_type;
}

void user_stage_give_up_row_stage(int id)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_no_row_stage.insert(id);
}

const UserStage* user_stage_by_id(int id)
{
    std::lock_guard<std::mutex> lock(g_mu);
    return id >= 0 && id < (int)g_stages.size() ? &g_stages[(size_t)id] : nullptr;
}

const UserStage* user_stage_of(const NodeType* t)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_by_type.find(t);
    return it == g_by_type.end() ? nullptr : &g_stages[(size_t)it->second];
}

}  // namespace rf
