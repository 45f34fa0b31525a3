// rf_jit.cpp -- see rf_jit.h.
#include "rf_jit.h"
#include "rf_user.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <vector>

namespace rf {

namespace {

// the device source: rf_device.h + rf_stream_dev.h + rf_user_dev.h + rf_glsl_dev.h, generated into build/ by the Makefile
const char kSource[] =
#include "build/rf_jit_source.inc"
    ;

typedef struct _hiprtcProgram* RtcProgram;
struct Rtc {
    void* handle = nullptr;
    int (*CreateProgram)(RtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*AddNameExpression)(RtcProgram, const char*) = nullptr;
    int (*CompileProgram)(RtcProgram, int, const char**) = nullptr;
    int (*GetProgramLogSize)(RtcProgram, size_t*) = nullptr;
    int (*GetProgramLog)(RtcProgram, char*) = nullptr;
    int (*GetCodeSize)(RtcProgram, size_t*) = nullptr;
    int (*GetCode)(RtcProgram, char*) = nullptr;
    int (*GetLoweredName)(RtcProgram, const char*, const char**) = nullptr;
    int (*DestroyProgram)(RtcProgram*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*Version)(int*, int*) = nullptr;      // optional: part of the disk-cache key
};

Rtc* rtc()
{
    static Rtc lib;
    static bool tried = false;
    if (tried) return lib.handle ? &lib : nullptr;
    tried = true;
    for (const char* n : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
        lib.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib.handle) break;
    }
    if (!lib.handle) return nullptr;
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(lib.handle, name);
        if (!p) ok = false;
        return p;
    };
    lib.CreateProgram = (decltype(lib.CreateProgram))sym("hiprtcCreateProgram");
    lib.AddNameExpression = (decltype(lib.AddNameExpression))sym("hiprtcAddNameExpression");
    lib.CompileProgram = (decltype(lib.CompileProgram))sym("hiprtcCompileProgram");
    lib.GetProgramLogSize = (decltype(lib.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
    lib.GetProgramLog = (decltype(lib.GetProgramLog))sym("hiprtcGetProgramLog");
    lib.GetCodeSize = (decltype(lib.GetCodeSize))sym("hiprtcGetCodeSize");
    lib.GetCode = (decltype(lib.GetCode))sym("hiprtcGetCode");
    lib.GetLoweredName = (decltype(lib.GetLoweredName))sym("hiprtcGetLoweredName");
    lib.DestroyProgram = (decltype(lib.DestroyProgram))sym("hiprtcDestroyProgram");
    lib.GetErrorString = (decltype(lib.GetErrorString))sym("hiprtcGetErrorString");
    lib.Version = (decltype(lib.Version))dlsym(lib.handle, "hiprtcVersion");
    if (!ok) {
        dlclose(lib.handle);
        lib.handle = nullptr;
        return nullptr;
    }
    return &lib;
}

struct Compiled {
    std::vector<char> code;
    std::string lowered;     // mangled kernel name
    std::string disk;        // path stem of the disk-cache entry this came from ("" = compiled by this process)
};

const char kArch[] = "gfx950";

std::mutex g_mu;
std::map<std::string, Compiled> g_code;                       // name expression -> code object (process cache)
std::map<std::string, JitKernel> g_loaded;                    // "dev|name expression" -> loaded function
int g_compiled = 0;

std::string name_expression(int fmt, int pf, int texels, const StageList& sl)
{
    return std::string("rf::stream_kernel<") + (fmt == kFmtRGBA8 ? "rf::PxU8" : (fmt == kPxF32Stream ? "rf::PxF32NT" : "rf::PxF32")) + ", " + std::to_string(pf) + ", " +
           std::to_string(texels) + ", " + sl.type_list() + ">";
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull)
{
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

// Optional disk cache (RF_JIT_CACHE_DIR), shared by concurrent processes (the ranks of one job, pytest workers):
//   <dir>/rfjit_<key>.hsaco   the code object, a plain ELF (llvm-objdump reads it: tests/test_jit_isa.py)
//   <dir>/rfjit_<key>.name    "<mangled kernel name>\n<code bytes> <fnv1a of the code>\n"
// key = hash of device source + name expression + waves per block + target + hiprtc version.  Both files are written to
// a temporary name and renamed; an entry whose .name does not match its .hsaco (torn, stale, truncated) is deleted
// and treated as a miss, and so is one the driver refuses to load (jit_compile): a bad entry costs a recompile on the
// process that finds it, never a silent fallback on ONE rank.
std::string cache_path(const std::string& expr, int waves_per_block)
{
    const char* dir = std::getenv("RF_JIT_CACHE_DIR");
    if (!dir || !*dir) return "";
    int major = 0, minor = 0;
    std::string where;
    if (Rtc* r = rtc()) {
        if (r->Version) (void)r->Version(&major, &minor);
        // WHICH libhiprtc: a process that imported PyTorch first gets the copy PyTorch bundles (same SONAME, another
        // compiler build, other code for the same source) -- its objects must not be served to a process on /opt/rocm's
        Dl_info info;
        if (dladdr((void*)r->CompileProgram, &info) && info.dli_fname) where = info.dli_fname;
    }
    const std::string salt = expr + "|w" + std::to_string(waves_per_block) + "|" + kArch + "|rtc" + std::to_string(major) + "." + std::to_string(minor) + "|" + where;
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%016llx", (unsigned long long)fnv1a(salt, fnv1a(kSource)));
    return std::string(dir) + "/rfjit_" + buf;
}

bool read_file(const std::string& p, std::vector<char>& out)
{
    std::ifstream f(p, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return !out.empty();
}

bool write_file_atomically(const std::string& path, const char* data, size_t n)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        f.write(data, (std::streamsize)n);
        if (!f) { (void)std::remove(tmp.c_str()); return false; }
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { (void)std::remove(tmp.c_str()); return false; }
    return true;
}

void cache_drop(const std::string& stem)
{
    (void)std::remove((stem + ".name").c_str());
    (void)std::remove((stem + ".hsaco").c_str());
}

bool cache_load(const std::string& stem, Compiled& c)
{
    std::vector<char> nm;
    if (!read_file(stem + ".name", nm) || !read_file(stem + ".hsaco", c.code)) return false;
    const std::string text(nm.begin(), nm.end());
    const size_t nl = text.find('\n');
    unsigned long long size = 0, hash = 0;
    if (nl == std::string::npos || std::sscanf(text.c_str() + nl + 1, "%llu %llx", &size, &hash) != 2 || size != c.code.size() ||
        hash != (unsigned long long)fnv1a(std::string(c.code.begin(), c.code.end()))) {
        cache_drop(stem);
        return false;
    }
    c.lowered = text.substr(0, nl);
    c.disk = stem;
    return !c.lowered.empty();
}

void cache_store(const std::string& stem, const Compiled& c)
{
    const char* dir = std::getenv("RF_JIT_CACHE_DIR");
    (void)mkdir(dir, 0755);
    char tail[64];
    std::snprintf(tail, sizeof(tail), "\n%llu %llx\n", (unsigned long long)c.code.size(),
                  (unsigned long long)fnv1a(std::string(c.code.begin(), c.code.end())));
    const std::string name = c.lowered + tail;
    if (write_file_atomically(stem + ".hsaco", c.code.data(), c.code.size())) (void)write_file_atomically(stem + ".name", name.data(), name.size());
}

// RF_GLSL_BUFFER_LOADS=0: .comp nodes address their texels with 64-bit pointers whatever the image size (A/B measurements)
bool glsl_buffer_loads()
{
    static const bool on = [] { const char* e = std::getenv("RF_GLSL_BUFFER_LOADS"); return !(e && *e && std::atoi(e) == 0); }();
    return on;
}

std::string node_expression(int fmt, const UserStage& u, bool wide = false)
{
    if (u.glsl) {      // {type}.comp: rfglsl::glsl_node_kernel<RfgShader, texel format, RfgInfo> (rf_glsl_dev.h)
        const std::string ns = "rfglsl::" + u.ident + "::";
        return "rfglsl::glsl_node_kernel<" + ns + "RfgShader, " + (fmt == kFmtRGBA8 ? "rf::PxU8" : "rf::PxF32") + ", " + ns + "RfgInfo, " + (wide || !glsl_buffer_loads() ? "false" : "true") + ">";
    }
    return std::string("rf::user_node_kernel<") + (fmt == kFmtRGBA8 ? "rf::PxU8" : "rf::PxF32") + ", rfuser::" + u.ident + "::Stage>";
}

// the window kernel of a .comp stencil: user_node_kernel over the stage rf_glsl.cpp generates (rfuser::<ident>::Stage)
std::string window_expression(int fmt, const UserStage& u)
{
    return std::string("rf::user_node_kernel<") + (fmt == kFmtRGBA8 ? "rf::PxU8" : "rf::PxF32") + ", rfuser::" + u.ident + "::Stage>";
}

std::string fill_expression(const UserStage& u) { return "rf::user_fill_kernel<rfuser::" + u.ident + "::Stage>"; }

// expr: the kernel instantiation to build; users: the user stages (rf_user.h ids) whose wrappers the translation unit needs
const Compiled* compile_expr(const std::string& expr, const std::vector<int>& users, int waves_per_block, std::string& err)
{
    auto it = g_code.find(expr);
    if (it != g_code.end()) return &it->second;
    const std::string cpath = cache_path(expr, waves_per_block);
    if (!cpath.empty()) {
        Compiled c;
        if (cache_load(cpath, c)) return &(g_code[expr] = std::move(c));
    }
    Rtc* r = rtc();
    if (!r) { err = "libhiprtc.so could not be loaded"; return nullptr; }
    RtcProgram prog = nullptr;
    const std::string wpb = "-DRF_WAVES_PER_BLOCK=" + std::to_string(waves_per_block);
    // the stages of user types ({shader_path}/{type}.stage.hip) this list names, wrapped into their namespaces (rf_user.cpp)
    std::string source_with_users;
    const char* source = kSource;
    if (!users.empty()) {
        source_with_users = kSource;
        for (int id : users) {
            const UserStage* u = user_stage_by_id(id);
            if (!u) { err = "a user stage of this launch is no longer registered"; return nullptr; }
            if (source_with_users.find("namespace " + u->ident + " {") == std::string::npos) source_with_users += u->wrapper();
        }
        source = source_with_users.c_str();
    }
    int rc = r->CreateProgram(&prog, source, "rf_stream_jit.hip", 0, nullptr, nullptr);
    if (rc != 0) { err = std::string("hiprtcCreateProgram: ") + r->GetErrorString(rc); return nullptr; }
    // the flags of the ahead-of-time build (Makefile): explicit fmaf only, no contraction -- bit-identical to the oracle
    // A user NODE reads its neighbourhood through Window::at inside loops of the file's own (rf_user_dev.h): with small radii the
    // window is a register copy, which only pays when those loops are unrolled -- a 5 x 5 tap loop over an inlined body is beyond
    // the compiler's default threshold, so it is raised for these kernels (their bodies are a few hundred instructions).
    const bool node = expr.find("user_node_kernel") != std::string::npos || expr.find("glsl_node_kernel") != std::string::npos;
    // -fwrapv (units that hold a GLSL file only): GLSL's int arithmetic wraps; C++ calls the overflow undefined and may optimise on that
    bool glsl = false;
    for (int id : users) {
        const UserStage* u = user_stage_by_id(id);
        glsl = glsl || (u && u->glsl);
    }
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", wpb.c_str()};
    if (node) { opts.push_back("-mllvm"); opts.push_back("-unroll-threshold=6000"); }
    if (glsl) opts.push_back("-fwrapv");
    if (glsl) {
        // A float vector and an integer vector of the same size convert into each other by REINTERPRETING their bits under this clang's
        // default (-flax-vector-conversions=all): `vec2(p) / imageSize(image)` of a GLSL file would compile and divide by garbage, where GLSL
        // converts the values.  The strict mode cannot be the mode of the build (the run-time compiler's own header needs the lax one),
        // so the unit is parsed once more in strict mode and only the errors that lie in a .comp file count: the file is refused and
        // told to spell the constructor.  (int and uint vectors still mix, as in GLSL.)
        RtcProgram check = nullptr;
        if (r->CreateProgram(&check, source, "rf_stream_jit.hip", 0, nullptr, nullptr) == 0) {
            std::vector<const char*> strict = opts;
            strict.push_back("-flax-vector-conversions=integer");
            strict.push_back("-ferror-limit=0");
            r->AddNameExpression(check, expr.c_str());
            if (r->CompileProgram(check, (int)strict.size(), strict.data()) != 0) {
                size_t ls = 0;
                r->GetProgramLogSize(check, &ls);
                std::string log(ls, 0);
                if (ls) r->GetProgramLog(check, &log[0]);
                std::string found;
                size_t at = 0;
                while (at < log.size()) {
                    size_t nl = log.find('\n', at);
                    if (nl == std::string::npos) nl = log.size();
                    const std::string line = log.substr(at, nl - at);
                    const size_t comp = line.find(".comp:"), er = line.find(": error:");
                    if (comp != std::string::npos && er != std::string::npos && comp < er) {
                        found = line;
                        // the line of the file and the caret under it
                        for (int extra = 0; extra < 2 && nl < log.size(); ++extra) {
                            const size_t n2 = log.find('\n', nl + 1);
                            found += "\n" + log.substr(nl + 1, (n2 == std::string::npos ? log.size() : n2) - nl - 1);
                            nl = n2 == std::string::npos ? log.size() : n2;
                        }
                        break;
                    }
                    at = nl + 1;
                }
                if (!found.empty()) {
                    r->DestroyProgram(&check);
                    r->DestroyProgram(&prog);
                    err = found + "\n(GLSL converts an integer vector to a float vector where the two meet; here the conversion has to be written: vec2(p), vec4(size, size), ...)";
                    return nullptr;
                }
            }
            r->DestroyProgram(&check);
        }
    }
    rc = r->AddNameExpression(prog, expr.c_str());
    if (rc == 0) rc = r->CompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != 0) {
        size_t ls = 0;
        r->GetProgramLogSize(prog, &ls);
        std::string log(ls, 0);
        if (ls) r->GetProgramLog(prog, &log[0]);
        err = std::string("hiprtc: ") + r->GetErrorString(rc) + " compiling " + expr + "\n" + log.substr(0, 2000);
        r->DestroyProgram(&prog);
        return nullptr;
    }
    Compiled c;
    size_t cs = 0;
    r->GetCodeSize(prog, &cs);
    c.code.resize(cs);
    r->GetCode(prog, c.code.data());
    const char* low = nullptr;
    r->GetLoweredName(prog, expr.c_str(), &low);
    c.lowered = low ? low : "";
    r->DestroyProgram(&prog);
    if (c.lowered.empty() || c.code.empty()) { err = "hiprtc produced no code for " + expr; return nullptr; }
    ++g_compiled;
    if (!cpath.empty()) cache_store(cpath, c);
    return &(g_code[expr] = std::move(c));
}

const Compiled* compile(int fmt, int pf, int texels, const StageList& sl, int waves_per_block, std::string& err)
{
    std::vector<int> users;
    for (int i = 0; i < sl.n; ++i)
        if (sl.st[i].kind == ST_USER) users.push_back(sl.st[i].user);
    return compile_expr(name_expression(fmt, pf, texels, sl), users, waves_per_block, err);
}

std::string loaded_key(const std::string& expr)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return std::to_string(dev) + "|" + expr;
}

}  // namespace

bool jit_available()
{
    if (const char* e = std::getenv("RF_NO_JIT"))
        if (std::atoi(e) != 0) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    return rtc() != nullptr;
}

std::string jit_library()
{
    std::lock_guard<std::mutex> lock(g_mu);
    Rtc* r = rtc();
    Dl_info info;
    if (r && dladdr((void*)r->CompileProgram, &info) && info.dli_fname) return info.dli_fname;
    return "";
}

int jit_compile_count()
{
    std::lock_guard<std::mutex> lock(g_mu);
    return g_compiled;
}

size_t jit_compile_only(int fmt, int pf, int texels, const StageList& sl, int waves_per_block, std::string& err)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const Compiled* c = compile(fmt, pf, texels, sl, waves_per_block, err);
    return c ? c->code.size() : 0;
}

namespace {
// compile (or fetch) `expr` and load it on the current device; g_mu held
bool load_expr(const std::string& expr, const std::vector<int>& users, int texels, int waves_per_block, std::string& err)
{
    const std::string lk = loaded_key(expr);
    if (g_loaded.count(lk)) return true;
    JitKernel k;
    for (int attempt = 0;; ++attempt) {
        const Compiled* c = compile_expr(expr, users, waves_per_block, err);
        if (!c) return false;
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, c->code.data());
        std::string what;
        if (e != hipSuccess) what = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        if (e == hipSuccess) {
            e = hipModuleGetFunction(&k.fn, mod, c->lowered.c_str());
            if (e != hipSuccess) {
                what = std::string("hipModuleGetFunction(") + c->lowered + "): " + hipGetErrorString(e);
                (void)hipModuleUnload(mod);
            }
        }
        if (e == hipSuccess) break;
        // an entry of the disk cache the driver refuses (stale, damaged in a way the checksum of its own bytes cannot
        // show): delete it and compile here, once -- never fall back on this process alone
        if (attempt == 0 && !c->disk.empty()) {
            cache_drop(c->disk);
            g_code.erase(expr);
            continue;
        }
        err = what;
        return false;
    }
    k.texels = texels;
    (void)hipFuncGetAttribute(&k.vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k.fn);
    (void)hipFuncGetAttribute(&k.scratch_bytes, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k.fn);
    int per_cu = 0, cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn, 64 * waves_per_block, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    k.resident_workgroups = per_cu * (cus > 0 ? cus : 256);
    g_loaded[lk] = k;      // the module stays loaded for the life of the process (kernels are shared by every graph)
    return true;
}
}  // namespace

bool jit_compile(int fmt, int pf, int texels, const StageList& sl, int waves_per_block, std::string& err)
{
    std::lock_guard<std::mutex> lock(g_mu);
    std::vector<int> users;
    for (int i = 0; i < sl.n; ++i)
        if (sl.st[i].kind == ST_USER) users.push_back(sl.st[i].user);
    return load_expr(name_expression(fmt, pf, texels, sl), users, texels, waves_per_block, err);
}

// ---- user nodes (rf_user_dev.h): user_node_kernel<Px, rfuser::<ident>::Stage>, 256 threads per workgroup ----------------
bool jit_compile_user_node(int fmt, int user_id, std::string& err, bool wide)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const UserStage* u = user_stage_by_id(user_id);
    if (!u || !(u->multi || u->has_alt)) { err = "not a user node"; return false; }
    if (!u->glsl && !u->buf_out.empty() && !load_expr(fill_expression(*u), {user_id}, 1, 4, err)) return false;      // RF_BUFFER_OUT: its fill kernel
    if (u->glsl && u->glsl_window && !load_expr(window_expression(fmt, *u), {user_id}, 1, 4, err)) return false;
    return load_expr(node_expression(fmt, *u, wide), {user_id}, 1, 4, err);
}

const JitKernel* jit_lookup_glsl_window(int fmt, int user_id)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const UserStage* u = user_stage_by_id(user_id);
    if (!u || !u->glsl_window) return nullptr;
    auto it = g_loaded.find(loaded_key(window_expression(fmt, *u)));
    return it == g_loaded.end() ? nullptr : &it->second;
}

const JitKernel* jit_lookup_user_fill(int user_id)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const UserStage* u = user_stage_by_id(user_id);
    if (!u || u->glsl || u->buf_out.empty()) return nullptr;
    auto it = g_loaded.find(loaded_key(fill_expression(*u)));
    return it == g_loaded.end() ? nullptr : &it->second;
}

const JitKernel* jit_lookup_user_node(int fmt, int user_id, bool wide)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const UserStage* u = user_stage_by_id(user_id);
    if (!u) return nullptr;
    auto it = g_loaded.find(loaded_key(node_expression(fmt, *u, wide)));
    return it == g_loaded.end() ? nullptr : &it->second;
}

size_t jit_compile_only_user_node(int fmt, int user_id, std::string& err)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const UserStage* u = user_stage_by_id(user_id);
    if (!u || !(u->multi || u->has_alt)) { err = "not a user node"; return 0; }
    size_t total = 0;
    if (!u->glsl && !u->buf_out.empty()) {
        const Compiled* f = compile_expr(fill_expression(*u), {user_id}, 4, err);
        if (!f) return 0;
        total += f->code.size();
    }
    if (u->glsl && u->glsl_window) {
        const Compiled* w = compile_expr(window_expression(fmt, *u), {user_id}, 4, err);
        if (!w) return 0;
        total += w->code.size();
    }
    const Compiled* c = compile_expr(node_expression(fmt, *u), {user_id}, 4, err);
    return c ? total + c->code.size() : 0;
}

const JitKernel* jit_lookup(int fmt, int pf, int texels, const StageList& sl)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_loaded.find(loaded_key(name_expression(fmt, pf, texels, sl)));
    return it == g_loaded.end() || it->second.disabled ? nullptr : &it->second;
}

// The entry STAYS (g_loaded never erases: jit_lookup hands out pointers into it, and an erased entry would be compiled and its
// module loaded again by the next graph that plans the same chain -- one leaked module per graph creation in a live-reload loop);
// it is only marked, jit_lookup no longer returns it and load_expr finds it present.
void jit_forget(int fmt, int pf, int texels, const StageList& sl)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_loaded.find(loaded_key(name_expression(fmt, pf, texels, sl)));
    if (it != g_loaded.end()) it->second.disabled = true;
}

hipError_t jit_launch(const JitKernel& k, unsigned grid, unsigned block, void* args, size_t size, hipStream_t stream)
{
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(k.fn, grid, 1, 1, block, 1, 1, 0, stream, nullptr, config);
}

}  // namespace rf
