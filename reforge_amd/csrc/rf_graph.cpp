// rf_graph.cpp -- device half of the C ABI (see include/rfhip.h, rf_runtime.h).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "rf_jit.h"
#include "rf_rccl_abi.h"
#include "rf_runtime.h"
#include "rf_user.h"
#include "rf_glsl.h"
#include "rf_user_dev.h"

using namespace rf;

// ---------------------------------------------------------------------------------
// RCCL, loaded on first use: a single-GPU context never touches it, and the library
// stays loadable on machines without it.
// ---------------------------------------------------------------------------------
namespace {

struct RcclLib {
    void* handle = nullptr;
    std::string path;                       // the file the symbols came from (dladdr)
    NcclGetUniqueIdFn GetUniqueId = nullptr;
    NcclCommInitRankFn CommInitRank = nullptr;
    NcclCommDestroyFn CommDestroy = nullptr;
    NcclSendFn Send = nullptr;
    NcclRecvFn Recv = nullptr;
    NcclGroupFn GroupStart = nullptr;
    NcclGroupFn GroupEnd = nullptr;
    NcclGetErrorStringFn GetErrorString = nullptr;
};

RcclLib* rccl_lib(std::string& err)
{
    static RcclLib lib;
    static bool tried = false;
    if (lib.handle) return &lib;
    if (tried) { err = "librccl.so could not be loaded"; return nullptr; }
    tried = true;
    // RF_RCCL_LIBRARY = path of the library to bind instead (a specific RCCL build; the test double under a host that has
    // already mapped its own librccl.so.1, e.g. PyTorch: a dlopen by SONAME would return that copy)
    const char* forced = std::getenv("RF_RCCL_LIBRARY");
    const char* names[] = {forced && *forced ? forced : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        lib.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib.handle) break;
        if (forced && *forced && n == names[0]) { err = std::string("dlopen(RF_RCCL_LIBRARY): ") + dlerror(); return nullptr; }
    }
    if (!lib.handle) { err = std::string("dlopen(librccl): ") + dlerror(); return nullptr; }
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(lib.handle, name);
        if (!p) ok = false;
        return p;
    };
    lib.GetUniqueId = (NcclGetUniqueIdFn)sym("ncclGetUniqueId");
    lib.CommInitRank = (NcclCommInitRankFn)sym("ncclCommInitRank");
    lib.CommDestroy = (NcclCommDestroyFn)sym("ncclCommDestroy");
    lib.Send = (NcclSendFn)sym("ncclSend");
    lib.Recv = (NcclRecvFn)sym("ncclRecv");
    lib.GroupStart = (NcclGroupFn)sym("ncclGroupStart");
    lib.GroupEnd = (NcclGroupFn)sym("ncclGroupEnd");
    lib.GetErrorString = (NcclGetErrorStringFn)sym("ncclGetErrorString");
    if (!ok) {
        err = "librccl.so lacks a required symbol";
        dlclose(lib.handle);
        lib.handle = nullptr;
        return nullptr;
    }
    // dlopen by SONAME returns a copy that is ALREADY mapped (PyTorch ships its own librccl.so with
    // SONAME librccl.so.1: under bench.py both communicators then live in that one library instance,
    // which is what RCCL expects); record the file for rf_comm_library()
    Dl_info info;
    if (dladdr((void*)lib.Send, &info) && info.dli_fname) lib.path = info.dli_fname;
    return &lib;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
            return RF_ERR_DEVICE;                                                          \
        }                                                                                  \
    } while (0)

#define NCCL_TRY(lib, expr)                                                                \
    do {                                                                                   \
        int r_ = (expr);                                                                   \
        if (r_ != 0) {                                                                     \
            set_error(std::string(#expr) + ": " + (lib)->GetErrorString(r_));              \
            return RF_ERR_DEVICE;                                                          \
        }                                                                                  \
    } while (0)

rf_status fail(rf_status st, const std::string& msg)
{
    set_error(msg);
    return st;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Row pitch of a device image.  EXPERIMENT (RF_PITCH_PAD = bytes added to every row): how the pitch maps rows onto the HBM
// channels decides how well a column-strip walk streams (scripts/walk_probe.py, profiles/r03_pitch_*).
size_t image_pitch(size_t row_bytes)
{
    size_t pitch = align_up(row_bytes, 256);
    if (const char* e = std::getenv("RF_PITCH_PAD")) pitch += align_up((size_t)std::atol(e), 256);
    return pitch;
}

RcclLib* ctx_rccl(const rf_ctx* ctx) { return (RcclLib*)ctx->rccl; }

int strip_rows_of(const rf_graph* g) { return g->strip_y1 - g->strip_y0; }

// clamp bounds / output rows of a launch in local rows
Geom launch_geom(const rf_graph* g, const Launch& L)
{
    const int Hs = strip_rows_of(g), H = g->opt.height;
    Geom geo;
    geo.W = g->opt.width;
    geo.row_lo = std::max(-L.need_src, -g->strip_y0);
    geo.row_hi = std::min(Hs - 1 + L.need_src, H - 1 - g->strip_y0);
    geo.y0 = std::max(-L.need_dst, -g->strip_y0);
    geo.y1 = std::min(Hs + L.need_dst, H - g->strip_y0);
    geo.nt_store = L.result_only && g->nt_stores;
    return geo;
}

// neighbour exchange of `r` rows of `img` above and below the strip:
// my top r rows -> rank-1's bottom ghost, my bottom r rows -> rank+1's top ghost.
rf_status exchange_rows(rf_graph* g, const DeviceImage& img, int r, hipStream_t stream)
{
    rf_ctx* ctx = g->ctx;
    if (ctx->world <= 1 || r <= 0) return RF_OK;
    RcclLib* lib = ctx_rccl(ctx);
    if (!lib || !ctx->comm)
        return fail(RF_ERR_UNSUPPORTED, "this rank has no RCCL communicator (rf_ctx_create_dist was given no unique id): "
                                        "use RF_GRAPH_NO_HALO_XCHG with rf_graph_fill_*, or create the context with an id");
    const int Hs = strip_rows_of(g);
    const size_t bytes = (size_t)r * img.pitch;
    NCCL_TRY(lib, lib->GroupStart());
    // every exit below closes the group: a return from inside an open ncclGroupStart would leave
    // the communicator collecting the NEXT caller's operations into this half-built group
    int rc = 0;
    const char* what = "";
    auto step = [&](int r_, const char* w) {
        if (rc == 0 && r_ != 0) { rc = r_; what = w; }
    };
    if (ctx->rank > 0) {
        step(lib->Send(img.base, bytes, kNcclChar, ctx->rank - 1, ctx->comm, stream), "ncclSend(up)");
        if (rc == 0) step(lib->Recv(img.base - (ptrdiff_t)bytes, bytes, kNcclChar, ctx->rank - 1, ctx->comm, stream), "ncclRecv(up)");
    }
    if (rc == 0 && ctx->rank < ctx->world - 1) {
        step(lib->Send(img.base + (size_t)(Hs - r) * img.pitch, bytes, kNcclChar, ctx->rank + 1, ctx->comm, stream), "ncclSend(down)");
        if (rc == 0) step(lib->Recv(img.base + (size_t)Hs * img.pitch, bytes, kNcclChar, ctx->rank + 1, ctx->comm, stream), "ncclRecv(down)");
    }
    const int rc_end = lib->GroupEnd();
    if (rc != 0) return fail(RF_ERR_DEVICE, std::string(what) + ": " + lib->GetErrorString(rc));
    if (rc_end != 0) return fail(RF_ERR_DEVICE, std::string("ncclGroupEnd: ") + lib->GetErrorString(rc_end));
    // The first exchange of a GRAPH is where a mis-paired send/recv (ranks that disagree about the
    // plan: every graph has a plan of its own) would hang forever.  Wait for it with a deadline instead: poll the
    // stream, give up after RF_XCHG_TIMEOUT_S seconds (default 60) with RF_ERR_DEVICE, never block without bound.
    if (!g->exchanged_once) {
        double limit_s = 60.0;
        if (const char* e = std::getenv("RF_XCHG_TIMEOUT_S")) limit_s = std::atof(e) > 0 ? std::atof(e) : limit_s;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail(RF_ERR_DEVICE, std::string("halo exchange: ") + hipGetErrorString(q));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
                return fail(RF_ERR_DEVICE, "halo exchange: the first neighbour send/recv of rank " + std::to_string(ctx->rank) + " of " +
                                               std::to_string(ctx->world) + " did not complete in " + std::to_string((int)limit_s) +
                                               " s (ranks disagree about the graph, or a rank is missing)");
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        g->exchanged_once = true;
    }
    return RF_OK;
}

bool exchange_mode(const rf_graph* g) { return g->ctx->world > 1 && !(g->opt.flags & RF_GRAPH_NO_HALO_XCHG); }

// 16 zero bytes every .comp launch of a graph can point at: what an imageLoad outside an image reads (rf_glsl_dev.h)
static const char kGlslZeroTexel[] = "\x01glsl:zero-texel";
// name of the scratch block behind a storage buffer a .comp node declares and the graph leaves unwired (the shader writes it
// whether anybody reads it or not): allocated with the graph's other buffers
static std::string unwired_buffer_name(const std::string& label, const std::string& block) { return "\x01unwired:" + label + ":" + block; }

// a node whose type is {shader_path}/{type}.comp (rf_glsl.h): rfglsl::glsl_node_kernel, one invocation of the translated
// main() per thread.  The dispatch is the reference's: ceil(W/16) x ceil(H/16) workgroups of the file's local_size
// (command.rs:167-168), in FRAME coordinates; a row strip runs the invocations of its own rows.
static hipError_t launch_glsl_node(rf_graph* g, const Launch& L, FrameSlot& f, const Geom& geo, hipStream_t stream, const UserStage* u, const JitKernel* k, int ring = 0)
{
    const Op& op = L.ops[0];
    const int y_org = g->strip_y0;
    struct Frame { int W, H, row_lo, row_hi, y0, y1, groups_x, groups_y; const char* zero; int ring, pad; } fr;
    static_assert(sizeof(Frame) == 48, "GlslFrame of rf_glsl_dev.h");
    fr.W = geo.W;
    fr.H = g->opt.height;
    fr.row_lo = geo.row_lo + y_org;
    fr.row_hi = geo.row_hi + y_org;
    fr.y0 = geo.y0 + y_org;
    fr.y1 = geo.y1 + y_org;
    fr.groups_x = (fr.W + 15) / 16;
    fr.groups_y = (fr.H + 15) / 16;
    fr.ring = ring;
    fr.pad = 0;
    {
        auto z = g->dev_buffers.find(kGlslZeroTexel);
        if (z == g->dev_buffers.end()) return hipErrorInvalidValue;
        fr.zero = reinterpret_cast<const char*>(z->second);
    }
    // the launch that runs the frame's last row also runs the invocations below it that the dispatch covers (they exist in the
    // reference; their image accesses fall outside the frame and are dropped, their storage-block writes are not)
    if (fr.y1 == fr.H && ring == 0) fr.y1 = std::max(fr.H, fr.groups_y * u->glsl_groups[1]);
    // the kernel's argument block (GlslArgs): 48 + 16 x 32 + 8 x 32 + 256 bytes at most, built on the stack every frame
    struct ArgBlock {
        alignas(8) unsigned char b[sizeof(Frame) + 16 * kGlslMaxImages + 8 * kGlslMaxBuffers + kGlslMaxUniformBytes + 8];
        size_t n = 0;
        unsigned char* data() { return b; }
        size_t size() const { return n; }
    } args;
    bool fits = true;
    auto push = [&](const void* p, size_t n) {
        if (args.n + n > sizeof(args.b)) { fits = false; return; }
        std::memcpy(args.b + args.n, p, n);
        args.n += n;
    };
    push(&fr, sizeof(Frame));
    const size_t n_img = std::max<size_t>(u->glsl_images.size(), 1);
    for (size_t i = 0; i < n_img; ++i) {
        struct { char* base; unsigned long long pitch; } im{nullptr, 0ull};
        if (i < u->glsl_images.size()) {
            const DeviceImage* di = nullptr;
            for (size_t d = 0; d < L.dsts.size(); ++d)
                if (u->glsl_image_written[i] && L.dst_bindings[d] == u->glsl_image_binding[i]) di = &f.images.at(L.dsts[d]);
            if (!di) {
                auto at = std::find(u->inputs.begin(), u->inputs.end(), u->glsl_images[i]);
                if (at != u->inputs.end() && (size_t)(at - u->inputs.begin()) < L.src.size()) di = &f.images.at(L.src[(size_t)(at - u->inputs.begin())]);
            }
            if (!di && std::find(u->inputs.begin(), u->inputs.end(), u->glsl_images[i]) != u->inputs.end()) return hipErrorInvalidValue;      // a readable image is always wired (build_launches)
            if (di) {      // address of FRAME row 0 (an address this rank may not own: the shader's loads and stores are bounded by row_lo .. row_hi)
                im.base = di->base - (ptrdiff_t)y_org * (ptrdiff_t)di->pitch;
                im.pitch = di->pitch;
            }
        }
        push(&im, sizeof(im));
    }
    const size_t n_buf = std::max<size_t>((size_t)u->glsl_buffers, 1);
    for (size_t b = 0; b < n_buf; ++b) {
        void* ptr = nullptr;
        if ((int)b < u->glsl_buffers) {
            const UserStage::Buffer* decl = nullptr;
            for (const auto* list : {&u->buf_in, &u->buf_out})
                for (const auto& x : *list)
                    if (x.slot == (int)b) decl = &x;
            if (!decl) return hipErrorInvalidValue;
            std::string name;
            for (size_t q = 0; q < L.out_buffers.size(); ++q)
                if (L.out_buffer_bindings[q] == decl->binding) name = L.out_buffers[q];
            for (size_t q = 0; q < L.in_buffers.size() && name.empty(); ++q)
                if (L.in_buffer_bindings[q] == decl->binding) name = L.in_buffers[q];
            if (name.empty()) name = unwired_buffer_name(L.label, decl->name);
            auto it = g->dev_buffers.find(name);
            if (it == g->dev_buffers.end()) return hipErrorInvalidValue;
            ptr = it->second;
        }
        push(&ptr, sizeof(ptr));
    }
    const size_t ubo = std::max<size_t>(((size_t)u->params_size + 7) / 8 * 8, 8);
    if (ubo > sizeof(op.user_params)) return hipErrorInvalidValue;
    push(op.user_params, ubo);
    if (!fits) return hipErrorInvalidValue;
    if (u->glsl_grouped) {
        const unsigned threads = (unsigned)(u->glsl_groups[0] * u->glsl_groups[1] * u->glsl_groups[2]);
        return jit_launch(*k, (unsigned)fr.groups_x * (unsigned)fr.groups_y, threads, args.data(), args.size(), stream);
    }
    if (ring > 0) {
        // the border ring of these rows (the kernel enumerates it the same way): whole rows within `ring` of the frame's top and bottom,
        // 2 x ring columns of the rows between
        const int R = ring, W = fr.W, H = fr.H;
        const long nt = std::max(0, std::min(fr.y1, R) - std::max(fr.y0, 0)), nb = std::max(0, std::min(fr.y1, H) - std::max(fr.y0, H - R));
        const long nm = std::max(0, std::min(fr.y1, H - R) - std::max(fr.y0, R));
        const long total = (nt + nb) * W + nm * 2 * R;
        if (total <= 0) return hipSuccess;
        return jit_launch(*k, (unsigned)(((total + 255) / 256 + 7) / 8 * 8), 256, args.data(), args.size(), stream);
    }
    // 64 x 4 invocations per workgroup over x < groups_x * LX, the rows of this launch (the kernel aligns the first to 4)
    const unsigned tiles_x = ((unsigned)(fr.groups_x * u->glsl_groups[0]) + 63u) / 64u;
    const unsigned y_first = (unsigned)fr.y0 & ~3u;
    const unsigned tiles_y = ((unsigned)fr.y1 - y_first + 3u) / 4u;
    const unsigned long tiles = (unsigned long)tiles_x * tiles_y;
    return jit_launch(*k, (unsigned)((tiles + 7) / 8 * 8), 256, args.data(), args.size(), stream);
}

// Does a recognised .comp stencil run on its window kernel in this graph?  Radius >= 2: yes.  Radius 1: on rgba8 only -- measured at 4K
// (profiles/r04_glsl_radius1_ab.txt): rgba32f sharpen.comp 39.3 us generic against 47.5 window + border ring, edge_detect.comp 51.0 / 50.2;
// rgba8 35.4 / 29.8 and 51.3 / 36.9 (there the generic kernel's per-load conversions are what the window's registers save).
static bool glsl_wants_window(const rf_graph* g, const UserStage* u)
{
    static const int min_radius = [] { const char* e = std::getenv("RF_GLSL_WINDOW_MIN_RADIUS"); return e && std::atoi(e) > 0 ? std::atoi(e) : 0; }();      // (A/B)
    if (!u->glsl || !u->glsl_window || g->glsl_no_window) return false;
    if (min_radius > 0) return u->radius >= min_radius;
    return u->radius >= 2 || g->opt.format == RF_FORMAT_RGBA8;
}

// an image of this graph spans 4 GiB or more (16384^2 rgba32f): beyond a 32-bit byte offset
static bool wide_images(const rf_graph* g)
{
    return image_pitch((size_t)g->opt.width * bytes_per_pixel(g->opt.format)) * (size_t)g->opt.height >= 0xFFFFFFF0ull;
}

// a user NODE (rf_user.h, a stage file that declares its images): user_node_kernel of rf_user_dev.h, compiled at graph creation
static hipError_t launch_user_node(rf_graph* g, const Launch& L, FrameSlot& f, const Geom& geo, hipStream_t stream)
{
    if (geo.y1 <= geo.y0 || geo.W <= 0) return hipSuccess;
    const int fmt = g->opt.format;
    const Op& op = L.ops[0];
    const UserStage* u = user_stage_by_id(op.user_id);
    const JitKernel* k = jit_lookup_user_node(fmt, op.user_id, wide_images(g));
    if (!u || !k || L.src.size() != u->inputs.size()) return hipErrorInvalidDeviceFunction;      // graph_build compiled it: cannot happen
    // a .comp file: its generic kernel -- unless it is a recognised stencil whose window kernel rf_graph_create has checked against the
    // generic one (glsl_window_ok): then the LDS-tiled kernel computes the frame and the generic kernel only the border ring, where the
    // shader's own treatment of the frame's edges decides (interior texels cannot tell the two apart)
    bool ring_after = false;
    if (u->glsl) {
        const bool window = glsl_wants_window(g, u) && g->glsl_window_ok.count(L.label) && geo.W > 2 * u->radius && g->opt.height > 2 * u->radius;
        if (!window) return launch_glsl_node(g, L, f, geo, stream, u, k);
        ring_after = true;
    }
    const JitKernel* generic = k;
    if (ring_after) {
        k = jit_lookup_glsl_window(fmt, op.user_id);
        if (!k) return hipErrorInvalidDeviceFunction;
    }
    UserNodeArgs A;
    std::memset(&A, 0, sizeof(A));
    for (size_t i = 0; i < L.src.size(); ++i) {
        const Image v = f.images.at(L.src[i]).view();
        A.src[i] = static_cast<const char*>(v.base);
        A.src_pitch[i] = v.pitch;
    }
    for (size_t k2 = 0; k2 < L.dsts.size(); ++k2) {
        // the output variable this image is wired to (outputs the graph leaves unwired stay null: not stored)
        for (size_t o = 0; o < u->outputs.size(); ++o) {
            if (u->out_binding[o] != L.dst_bindings[k2]) continue;
            const Image v = f.images.at(L.dsts[k2]).view();
            A.dst[o] = static_cast<char*>(v.base);
            A.dst_pitch[o] = v.pitch;
        }
    }
    A.W = geo.W;
    A.y0 = geo.y0;
    A.y1 = geo.y1;
    A.grid_x = (geo.W + 255) / 256;
    A.row_lo = geo.row_lo;
    A.row_hi = geo.row_hi;
    static_assert(sizeof(A.params) <= sizeof(op.user_params), "Params block");
    std::memcpy(A.params, op.user_params, sizeof(A.params));
    // storage buffers by block type name (shader.rs:144-147): the one it reads must be wired (build_launches), the one it fills
    // is filled only when the graph wires it to something
    if (!u->buf_in.empty()) {
        auto it = L.in_buffers.empty() ? g->dev_buffers.end() : g->dev_buffers.find(L.in_buffers[0]);
        if (it == g->dev_buffers.end()) return hipErrorInvalidValue;
        A.buf_in = it->second;
    }
    if (!u->buf_out.empty() && !L.out_buffers.empty()) {
        auto it = g->dev_buffers.find(L.out_buffers[0]);
        const JitKernel* fk = jit_lookup_user_fill(op.user_id);
        if (it == g->dev_buffers.end() || !fk) return hipErrorInvalidValue;
        A.buf_out = it->second;
        hipError_t e = jit_launch(*fk, (unsigned)((u->buf_out[0].count + 255) / 256), 256, &A, sizeof(A), stream);
        if (e != hipSuccess) return e;
    }
    const int rows = geo.y1 - geo.y0;
    // a node that reads through windows: one workgroup per 64 x TH tile of the output, its inputs staged in LDS (rf_user_dev.h,
    // UserTile); the grid a multiple of the 8 XCDs (the kernel gives each a contiguous range of tiles)
    const UserTile tile = user_tile((int)bytes_per_pixel(fmt), u->radius, (int)u->inputs.size());
    hipError_t e;
    if (u->radius > 0 && tile.lds) {
        A.grid_x = (geo.W + 63) / 64;
        const long tiles = (long)A.grid_x * ((rows + tile.th - 1) / tile.th);
        e = jit_launch(*k, (unsigned)((tiles + 7) / 8 * 8), 256, &A, sizeof(A), stream);
    } else {
        const unsigned gy = (unsigned)(rows > 1024 ? 1024 : rows);
        e = jit_launch(*k, (unsigned)A.grid_x * gy, 256, &A, sizeof(A), stream);
    }
    if (e != hipSuccess || !ring_after) return e;
    return launch_glsl_node(g, L, f, geo, stream, u, generic, u->radius);      // behind it on the same stream: the border ring, exactly as the file treats it
}

// the kernel(s) of one launch over output rows [y0, y1) of `geo`
rf_status launch_rows(rf_graph* g, FrameSlot& f, const Launch& L, Geom geo, int y0, int y1, hipStream_t stream)
{
    if (y1 <= y0) return RF_OK;
    geo.y0 = y0;
    geo.y1 = y1;
    const DeviceImage& dst = f.images.at(L.dst);
    if (L.ops.size() == 1 && L.ops[0].kind == OP_SPLIT) {
        // a node with several output bindings: one allocated image each (pipeline_graph.rs:205-224)
        Image luma{}, chroma{};
        for (size_t k = 0; k < L.dsts.size(); ++k) (L.dst_bindings[k] == 1 ? luma : chroma) = f.images.at(L.dsts[k]).view();
        HIP_TRY(launch_split_luma(g->opt.format, f.images.at(L.src[0]).view(), luma, chroma, geo, stream));
    } else if (L.ops.size() == 1 && L.ops[0].kind == OP_USERN) {
        HIP_TRY(launch_user_node(g, L, f, geo, stream));
    } else if (L.ops.size() == 1 && L.ops[0].kind == OP_MIX) {
        HIP_TRY(launch_mix(g->opt.format, f.images.at(L.src[0]).view(), f.images.at(L.src[1]).view(), dst.view(), geo,
                           L.ops[0].slope, stream));
    } else {
        HIP_TRY(launch_ops(g->opt.format, L.ops.data(), (int)L.ops.size(), f.images.at(L.src[0]).view(), dst.view(), geo,
                           g->tune, stream));
    }
    return RF_OK;
}

// One launch of the frame.  In exchange mode a stencil launch is split so that the halo flies
// while the interior computes: the rows that read no ghost row start at once on
// the launch's stream, the neighbour exchange runs on the slot's comm stream behind an event,
// and the r top + r bottom rows follow once the ghost rows have landed.  RF_FORCE_SPLIT=1
// takes the same three-part path on a single rank (no exchange) so the geometry is testable
// on one GPU.
rf_status run_launch(rf_graph* g, FrameSlot& f, size_t li, hipStream_t stream, bool timers)
{
    const Launch& L = g->launches[li];
    const Geom geo = launch_geom(g, L);
    const bool xchg = exchange_mode(g) && L.radius > 0;
    const int r = L.radius;
    const bool split = (xchg || (g->force_split && r > 0)) && (geo.y1 - geo.y0) > 4 * r;
    if (timers) HIP_TRY(hipEventRecord(f.t0[li], stream));
    if (!split) {
        if (xchg) {
            for (const auto& s : L.src) {
                rf_status st = exchange_rows(g, f.images.at(s), r, stream);
                if (st != RF_OK) return st;
            }
        }
        rf_status st = launch_rows(g, f, L, geo, geo.y0, geo.y1, stream);
        if (st != RF_OK) return st;
    } else {
        if (xchg) {
            HIP_TRY(hipEventRecord(f.src_ready, stream));            // everything that wrote src is ordered before this
            HIP_TRY(hipStreamWaitEvent(f.comm, f.src_ready, 0));
            for (const auto& s : L.src) {
                rf_status st = exchange_rows(g, f.images.at(s), r, f.comm);
                if (st != RF_OK) return st;
            }
            HIP_TRY(hipEventRecord(f.halo_ready, f.comm));
        }
        rf_status st = launch_rows(g, f, L, geo, geo.y0 + r, geo.y1 - r, stream);   // reads rows [y0, y1): no ghost row
        if (st != RF_OK) return st;
        if (xchg) HIP_TRY(hipStreamWaitEvent(stream, f.halo_ready, 0));
        // the r top and r bottom rows: ONE launch over two row ranges for the stream kernels (two slivers of r rows leave
        // most of the chip idle, and each launch costs its start-up and its drain), two launches for the kernels of their own
        const bool own_kernel = L.ops.size() == 1 && own_kernel_kind(L.ops[0].kind);
        if (own_kernel) {
            st = launch_rows(g, f, L, geo, geo.y0, geo.y0 + r, stream);
            if (st != RF_OK) return st;
            st = launch_rows(g, f, L, geo, geo.y1 - r, geo.y1, stream);
        } else {
            Geom two = geo;
            two.yb0 = geo.y1 - r;
            two.yb1 = geo.y1;
            st = launch_rows(g, f, L, two, geo.y0, geo.y0 + r, stream);
        }
        if (st != RF_OK) return st;
    }
    if (timers) HIP_TRY(hipEventRecord(f.t1[li], stream));
    if (g->sync_launches) HIP_TRY(hipStreamSynchronize(stream));   // RF_SYNC_LAUNCHES=1: debugging aid
    return RF_OK;
}

// command::execute_pipeline_graph, command.rs:166-242: layer by layer, in plan order on the
// frame's stream (stream order is the per-layer barrier of command.rs:226-240).  With
// RF_CONCURRENT_LAYERS=1 the nodes of a hazard-free layer run on side streams between a fork
// and a join event pair instead.
rf_status issue_frame(rf_graph* g, FrameSlot& f, bool timers)
{
    size_t li = 0;
    while (li < g->launches.size()) {
        size_t end = li;
        while (end < g->launches.size() && g->launches[end].layer == g->launches[li].layer) ++end;
        const size_t m = end - li;
        if (m == 1 || g->launches[li].serial || !g->concurrent_layers) {
            // Plan order on the frame's stream.  The nodes of a layer are independent and the reference
            // lets them overlap, but they are memory-bound: two at once evict each other's working set
            // and pay the fork/join events (4K diamond 174 us concurrent, 164 us in order; 1080p 49.5 vs
            // 39.2 us).  RF_CONCURRENT_LAYERS=1 puts hazard-free layers on side streams instead.
            for (size_t j = 0; j < m; ++j) {
                rf_status st = run_launch(g, f, li + j, f.stream, timers);
                if (st != RF_OK) return st;
            }
        } else {
            HIP_TRY(hipEventRecord(f.fork, f.stream));
            for (size_t j = 0; j < m; ++j) {
                hipStream_t s = j == 0 ? f.stream : f.aux[j - 1];
                if (j > 0) HIP_TRY(hipStreamWaitEvent(s, f.fork, 0));
                rf_status st = run_launch(g, f, li + j, s, timers);
                if (st != RF_OK) return st;
                if (j > 0) {
                    HIP_TRY(hipEventRecord(f.join[j - 1], s));
                    HIP_TRY(hipStreamWaitEvent(f.stream, f.join[j - 1], 0));
                }
            }
        }
        li = end;
    }
    return RF_OK;
}

void rebuild_ops(rf_graph* g)
{
    for (auto& L : g->launches) {
        L.ops = ops_of_members(g->plan.plan, L.members, L.member_slot, &g->weights_of);
    }
}

void destroy_graph_exec(FrameSlot& f)
{
    if (f.graph_exec) { (void)hipGraphExecDestroy(f.graph_exec); f.graph_exec = nullptr; }
    if (f.graph) { (void)hipGraphDestroy(f.graph); f.graph = nullptr; }
}


}  // namespace

// ---------------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------------
static rf_status ctx_create_common(int device, rf_ctx** out)
{
    if (!out) return fail(RF_ERR_INVALID, "rf_ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RF_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                                          " (librfhip has no CPU fallback)");
    if (device < 0 || device >= n) return fail(RF_ERR_INVALID, "rf_ctx_create: device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    std::string arch = prop.gcnArchName;
    if (arch.compare(0, 6, "gfx950") != 0)
        return fail(RF_ERR_NO_DEVICE, "device " + std::to_string(device) + " is " + arch + "; librfhip is built for gfx950 (MI355X) only");
    rf_ctx* ctx = new rf_ctx();
    ctx->device = device;
    ctx->arch = arch.substr(0, arch.find(':'));
    if (hipStreamCreateWithFlags(&ctx->util_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(RF_ERR_DEVICE, "hipStreamCreate failed");
    }
    float tables[256 + 255];
    srgb_tables(tables, tables + 256);
    if (hipMalloc((void**)&ctx->d_tables, sizeof(tables)) != hipSuccess ||
        hipMemcpy(ctx->d_tables, tables, sizeof(tables), hipMemcpyHostToDevice) != hipSuccess) {
        rf_ctx_destroy(ctx);
        return fail(RF_ERR_DEVICE, "sRGB table upload failed");
    }
    *out = ctx;
    return RF_OK;
}

extern "C" rf_status rf_ctx_create(int device, rf_ctx** out) { return ctx_create_common(device, out); }

extern "C" rf_status rf_comm_unique_id(void* id128)
{
    if (!id128) return fail(RF_ERR_INVALID, "rf_comm_unique_id: null buffer");
    std::string err;
    RcclLib* lib = rccl_lib(err);
    if (!lib) return fail(RF_ERR_DEVICE, err);
    NcclId id;
    NCCL_TRY(lib, lib->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return RF_OK;
}

extern "C" const char* rf_comm_library(void)
{
    std::string err;
    RcclLib* lib = rccl_lib(err);
    return lib ? lib->path.c_str() : "";
}

extern "C" rf_status rf_ctx_create_dist(int device, int rank, int world, const void* id128, rf_ctx** out)
{
    if (world < 1 || rank < 0 || rank >= world) return fail(RF_ERR_INVALID, "rf_ctx_create_dist: bad rank/world");
    rf_status st = ctx_create_common(device, out);
    if (st != RF_OK) return st;
    rf_ctx* ctx = *out;
    ctx->rank = rank;
    ctx->world = world;
    if (world == 1) return RF_OK;
    // no unique id: a rank without a communicator.  Its graphs must carry their halo
    // themselves (RF_GRAPH_NO_HALO_XCHG with a generated input); any exchange fails loudly.
    if (!id128) return RF_OK;
    std::string err;
    RcclLib* lib = rccl_lib(err);
    if (!lib) { rf_ctx_destroy(ctx); *out = nullptr; return fail(RF_ERR_DEVICE, err); }
    ctx->rccl = (const RcclApi*)lib;
    NcclId id;
    std::memcpy(&id, id128, sizeof(id));
    int r = lib->CommInitRank(&ctx->comm, world, id, rank);
    if (r != 0) {
        std::string msg = std::string("ncclCommInitRank: ") + lib->GetErrorString(r);
        ctx->comm = nullptr;
        rf_ctx_destroy(ctx);
        *out = nullptr;
        return fail(RF_ERR_DEVICE, msg);
    }
    return RF_OK;
}

extern "C" void rf_ctx_destroy(rf_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->comm && ctx->rccl) ctx_rccl(ctx)->CommDestroy(ctx->comm);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->util_stream) (void)hipStreamDestroy(ctx->util_stream);
    delete ctx;
}

extern "C" rf_status rf_ctx_synchronize(rf_ctx* ctx)
{
    if (!ctx) return fail(RF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    return RF_OK;
}

extern "C" int rf_ctx_rank(const rf_ctx* ctx) { return ctx ? ctx->rank : -1; }
extern "C" int rf_ctx_world(const rf_ctx* ctx) { return ctx ? ctx->world : -1; }
extern "C" const char* rf_ctx_device_arch(const rf_ctx* ctx) { return ctx ? ctx->arch.c_str() : ""; }

extern "C" rf_status rf_ctx_copy_bandwidth(rf_ctx* ctx, size_t bytes, int iters, float* gbps)
{
    if (!ctx || !gbps || iters < 1) return fail(RF_ERR_INVALID, "rf_ctx_copy_bandwidth: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int W = 7680;
    const size_t pitch = image_pitch((size_t)W * 16);
    const int rows = (int)std::min<size_t>(bytes / pitch, 1u << 20);
    if (rows < 1) return fail(RF_ERR_INVALID, "rf_ctx_copy_bandwidth: fewer bytes than one row");
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    rf_status st = RF_OK;
    float ms = 0.f;
    auto body = [&]() -> rf_status {
        HIP_TRY(hipMalloc(&a, (size_t)rows * pitch));
        HIP_TRY(hipMalloc(&b, (size_t)rows * pitch));
        HIP_TRY(hipMemsetAsync(a, 0x3c, (size_t)rows * pitch, ctx->util_stream));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        Op pass;
        Geom geo;
        geo.W = W;
        geo.row_lo = 0;
        geo.row_hi = rows - 1;
        geo.y0 = 0;
        geo.y1 = rows;
        const StreamTuning tune;
        for (int i = 0; i < 3; ++i) HIP_TRY(launch_ops(kFmtRGBA32F, &pass, 1, Image{a, pitch}, Image{b, pitch}, geo, tune, ctx->util_stream));
        HIP_TRY(hipEventRecord(e0, ctx->util_stream));
        for (int i = 0; i < iters; ++i) HIP_TRY(launch_ops(kFmtRGBA32F, &pass, 1, Image{a, pitch}, Image{b, pitch}, geo, tune, ctx->util_stream));
        HIP_TRY(hipEventRecord(e1, ctx->util_stream));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        return RF_OK;
    };
    st = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (st != RF_OK) return st;
    *gbps = (float)(2.0 * (double)W * 16.0 * (double)rows * iters / ((double)ms * 1e-3) / 1e9);
    return RF_OK;
}

// ---------------------------------------------------------------------------------
// Graph
// ---------------------------------------------------------------------------------
// tuning and executor variants: rf_graph_options first, the environment overrides when set
static void read_tuning(rf_graph* g)
{
    const rf_graph_options& opt = g->opt;
    g->tune.rows_per_chunk = opt.rows_per_chunk;
    g->tune.conv_path = opt.conv_path;
    g->tune.texels_per_lane = opt.texels_per_lane;
    g->tune.walk = (opt.exec_flags & RF_EXEC_NO_ALTERNATE) ? 2 : (opt.exec_flags & RF_EXEC_ALTERNATE) ? 1 : 0;
    g->sync_launches = (opt.exec_flags & RF_EXEC_SYNC_LAUNCHES) != 0;
    g->concurrent_layers = (opt.exec_flags & RF_EXEC_CONCURRENT_LAYERS) != 0;
    g->force_split = (opt.exec_flags & RF_EXEC_FORCE_SPLIT) != 0;
    g->glsl_no_window = (opt.exec_flags & RF_EXEC_GLSL_NO_WINDOW) != 0;
    if (const char* e = std::getenv("RF_ROWS_PER_CHUNK")) g->tune.rows_per_chunk = std::atoi(e);
    if (const char* e = std::getenv("RF_CONV_PATH")) g->tune.conv_path = std::atoi(e);
    if (g->tune.conv_path < 0 || g->tune.conv_path > 3) g->tune.conv_path = -1;      // refused by rf_graph_create (RF_ERR_INVALID): there is no such kernel
    if (const char* e = std::getenv("RF_TEXELS_PER_LANE")) g->tune.texels_per_lane = std::atoi(e);
    if (const char* e = std::getenv("RF_SYNC_LAUNCHES")) g->sync_launches = std::atoi(e) != 0;
    if (const char* e = std::getenv("RF_CONCURRENT_LAYERS")) g->concurrent_layers = std::atoi(e) != 0;
    if (const char* e = std::getenv("RF_FORCE_SPLIT")) g->force_split = std::atoi(e) != 0;
    if (const char* e = std::getenv("RF_NT_STORE")) g->nt_stores = std::atoi(e) != 0;
    if (const char* e = std::getenv("RF_NO_ALTERNATE")) g->tune.walk = std::atoi(e) ? 2 : 1;
    // Exchange mode shares ONE comm stream and one src_ready/halo_ready event pair per slot: two
    // stencils of a layer reading the same source would each re-exchange its ghost rows while the
    // other's boundary kernels may still read them.  Plan order on one stream keeps it ordered.
    if (exchange_mode(g)) g->concurrent_layers = false;
}

// ---- .comp files: the fast forms of a shader are held against its generic kernel before a graph uses them ------------------------------
// The generic kernel IS the file as the reference would run it.  The LDS-tiled window kernel and the 3 x 3 row stage rest on what
// rf_glsl.cpp read off the file's text (translation invariance), on the radius the file states and -- the row stage -- on the file
// treating the frame's edges as clamp-to-edge.  So the node is run alone on a small random frame with the node's own parameters, both
// ways, every wired output, and compared bit for bit.  Graph creation is where the reference compiles its shaders
// (pipeline_graph.rs:509-545): a few milliseconds here are in the right place.
static thread_local bool t_in_selftest = false;
static bool glsl_same_both_ways(rf_graph* g, const UserStage* u, const NodeParams& np, const std::vector<int>* wired, uint32_t flags_a, uint32_t exec_a,
                                uint32_t flags_b, uint32_t exec_b, std::string& why)
{
    t_in_selftest = true;
    bool same = true;
    const int W = 96, H = 64;
    const size_t bpp = bytes_per_pixel(g->opt.format);
    rf_ctx* ctx = nullptr;
    if (rf_ctx_create(g->ctx->device, &ctx) != RF_OK) { same = false; why = last_error(); }
    for (size_t o = 0; same && o < u->outputs.size(); ++o) {
        if (wired && std::find(wired->begin(), wired->end(), u->out_binding[o]) == wired->end()) continue;      // not wired in this graph
        std::string text;
        for (size_t k = 0; k < u->inputs.size(); ++k) {
            if (k == 0) text += "input -> nn:" + u->inputs[k] + "\n";
            else text += "input -> t" + std::to_string(k) + "x -> nn:" + u->inputs[k] + "\nt" + std::to_string(k) + "x: grade { slope: 0." + std::to_string(5 + k) + ", offset: 0.0" + std::to_string(k) + ", saturation: 1.0 }\n";
        }
        text += "nn:" + u->outputs[o] + " -> output\nnn: " + u->type_name + " {}\n";
        rf_config* cfg = nullptr;
        if (rf_config_parse(text.c_str(), 1, &cfg) != RF_OK) { same = false; why = last_error(); break; }
        std::vector<unsigned char> got[2];
        for (int variant = 0; same && variant < 2; ++variant) {
            rf_graph_options opt{};
            opt.width = W;
            opt.height = H;
            opt.format = g->opt.format;
            opt.num_frames = 1;
            opt.flags = variant ? flags_b : flags_a;
            opt.exec_flags = variant ? exec_b : exec_a;
            rf_graph* t = nullptr;
            if (rf_graph_create(ctx, cfg, &opt, &t) != RF_OK) { same = false; why = last_error(); break; }
            for (const auto& p : u->params) {
                auto it = np.values.find(p.name);
                ParamValue v;
                v.i = 0;
                if (it != np.values.end()) v = it->second;
                (void)rf_graph_set_param(t, "nn", p.name.c_str(), p.type == PARAM_F32 ? RF_PARAM_F32 : (p.type == PARAM_I32 ? RF_PARAM_I32 : RF_PARAM_BOOL), &v);
            }
            got[variant].resize((size_t)W * H * bpp);
            if (rf_graph_fill_synthetic(t, 0x5E1F7E57u + (uint32_t)o) != RF_OK || rf_graph_execute(t, 0) != RF_OK || rf_graph_wait(t, 0) != RF_OK ||
                rf_graph_download_raw(t, 0, got[variant].data(), (size_t)W * bpp) != RF_OK) { same = false; why = last_error(); }
            rf_graph_destroy(t);
        }
        rf_config_destroy(cfg);
        if (same && got[0] != got[1]) { same = false; why = "it differs from the file's generic kernel on a random 96 x 64 frame (output " + u->outputs[o] + ")"; }
    }
    if (ctx) rf_ctx_destroy(ctx);
    t_in_selftest = false;
    (void)hipSetDevice(g->ctx->device);
    return same;
}

// the window kernels of the graph's launches
static void glsl_window_selftests(rf_graph* g)
{
    if (g->glsl_no_window) return;
    for (const auto& L : g->launches) {
        if (L.ops.size() != 1 || L.ops[0].kind != OP_USERN) continue;
        const UserStage* u = user_stage_by_id(L.ops[0].user_id);
        if (!u || !glsl_wants_window(g, u)) continue;
        if (t_in_selftest) { g->glsl_window_ok.insert(L.label); continue; }      // the inner graph of a check: this IS the kernel under test
        std::string why;
        if (glsl_same_both_ways(g, u, g->plan.plan.nodes.at(L.members[0]), &L.dst_bindings, 0u, 0u, 0u, RF_EXEC_GLSL_NO_WINDOW, why)) g->glsl_window_ok.insert(L.label);
        else g->jit_note += (g->jit_note.empty() ? "" : "; ") + std::string("node '") + L.label + "' (" + u->file_name() + ", #pragma rf radius " + std::to_string(u->radius) + ") keeps its generic kernel: its window kernel: " + why;
    }
}

// the 3 x 3 row stages the plan holds: true if one was given up (the caller plans again: the type is a node from now on)
static bool glsl_row_stage_selftests(rf_graph* g)
{
    if (t_in_selftest) return false;
    bool gave_up = false;
    std::set<int> seen;
    for (const auto& kv : g->plan.plan.nodes) {
        if (kv.second.type->kind != OP_USER) continue;
        const UserStage* u = user_stage_of(kv.second.type);
        if (!u || !u->glsl || u->radius != 1 || !seen.insert(u->id).second) continue;      // (a point shader is a row stage by a proof of its own)
        std::string why;
        if (glsl_same_both_ways(g, u, kv.second, nullptr, RF_GRAPH_NO_FUSION, 0u, RF_GRAPH_GLSL_NODES, RF_EXEC_GLSL_NO_WINDOW, why)) continue;
        user_stage_give_up_row_stage(u->id);
        gave_up = true;
        g->jit_note += (g->jit_note.empty() ? "" : "; ") + std::string("type '") + u->type_name + "' (" + u->file_name() + ") is not fused: as a 3 x 3 row stage " + why;
    }
    return gave_up;
}

static rf_status graph_build(rf_graph* g, const rf_config* cfg)
{
    rf_ctx* ctx = g->ctx;
    const rf_graph_options& opt = g->opt;
    std::string err;
    uint32_t plan_flags = opt.flags;
    if (!build_plan(cfg->cfg, plan_flags, g->plan.plan, err)) return fail(RF_ERR_GRAPH, err);
    HIP_TRY(hipSetDevice(ctx->device));
    if (!(plan_flags & kPlanGlslNodes) && glsl_row_stage_selftests(g)) {
        g->plan = rf_plan();
        if (!build_plan(cfg->cfg, plan_flags, g->plan.plan, err)) return fail(RF_ERR_GRAPH, err);
    }
    g->plan.index();
    read_tuning(g);
    if (g->tune.conv_path < 0) return fail(RF_ERR_INVALID, "rf_graph_create: conv_path must be 0 (auto), 1 (LDS tile), 2 (MFMA band) or 3 (VALU)");
    // Kernels of fused chains the ahead-of-time catalogue lacks are compiled HERE, where the reference compiles its
    // shaders (PipelineGraph::new -> Pipeline::new_compute, pipeline_graph.rs:509-545), never on the frame path.  If a
    // chain cannot be compiled (no libhiprtc, or the compiler rejects it) the graph is planned again with
    // catalogue-only fusion: same results, more launches.
    int Hs0, Hs1;
    strip_rows(opt.height, ctx->world, ctx->rank, Hs0, Hs1);
    // prepare the kernels of the plan's launches; user_only: just the launches that hold a user stage
    auto prepare = [&](bool user_only, std::string& jerr) {
        for (const auto& d : g->plan.launches) {
            std::vector<Op> ops = ops_of_members(g->plan.plan, d.members, d.member_slot, nullptr);
            bool has_user = false;
            for (const auto& o : ops) has_user = has_user || o.kind == OP_USER;
            if (ops.size() == 1 && ops[0].kind == OP_USERN) {
                // a user NODE (its file declares its images): a kernel of its own, rf_user_dev.h
                if (user_only && !jit_compile_user_node(opt.format, ops[0].user_id, jerr, wide_images(g))) return false;
                continue;
            }
            if (user_only ? !has_user : (ops.size() < 2 && !has_user)) continue;      // single built-in nodes are in the catalogue
            std::string note;
            if (!stream_prepare(opt.format, ops.data(), (int)ops.size(), opt.width, Hs1 - Hs0, g->tune, d.result_only && g->nt_stores, jerr, &note)) return false;
            if (!note.empty()) g->jit_note += (g->jit_note.empty() ? "" : "; ") + note;
        }
        return true;
    };
    if (!(plan_flags & (kPlanNoJit | kPlanNoFusion)) && g->plan.launch_error.empty()) {
        std::string jerr;
        if (!prepare(false, jerr)) {
            // With neighbour exchanges the launch list fixes how many rows every ncclSend/Recv carries and how many
            // exchanges a frame has: a rank that fell back on its own would mis-pair them with its neighbours'.  The
            // fallback is a LOCAL decision, so it is not taken there: the graph is refused and the caller decides for
            // all ranks (RF_GRAPH_NO_JIT on every rank, or RF_GRAPH_NO_HALO_XCHG).
            if (ctx->world > 1 && !(opt.flags & RF_GRAPH_NO_HALO_XCHG))
                return fail(RF_ERR_UNSUPPORTED, "a fused launch of this graph could not be compiled on rank " + std::to_string(ctx->rank) + " (" + jerr +
                                                    "); in exchange mode the ranks must agree on the launch list: create the graph with RF_GRAPH_NO_JIT on every rank");
            g->jit_note = "catalogue-only fusion: " + jerr;
            plan_flags |= kPlanNoJit;
            g->plan = rf_plan();
            if (!build_plan(cfg->cfg, plan_flags, g->plan.plan, err)) return fail(RF_ERR_GRAPH, err);
            g->plan.index();
        }
    }
    // A node whose type is a FILE ({shader_path}/{type}.stage.hip, rf_user.h) is compiled here whatever the fusion flags say --
    // this IS the reference's shader compile (Shader::from_path, shader.rs:29-59): a file that does not compile refuses the
    // graph with the compiler's message, and the caller keeps the graph it has (render.rs:121-136).
    if (g->plan.launch_error.empty()) {
        std::string jerr;
        if (!prepare(true, jerr)) return fail(RF_ERR_GRAPH, jerr);
    }
    const Plan& plan = g->plan.plan;

    strip_rows(opt.height, ctx->world, ctx->rank, g->strip_y0, g->strip_y1);
    const int Hs = strip_rows_of(g);
    if (Hs < 1) return fail(RF_ERR_INVALID, "frame has fewer rows than ranks");

    // storage buffers (PipelineGraphFrame::new, pipeline_graph.rs:142-175,:249-260): one zero-filled device buffer per
    // allocated name, sized to the largest block any user declares
    for (const auto& name : plan.buffers) {
        float* d = nullptr;
        const size_t bytes = std::max<size_t>(plan.buffer_bytes.at(name), 4);
        HIP_TRY(hipMalloc((void**)&d, bytes));
        g->dev_buffers[name] = d;
        HIP_TRY(hipMemset(d, 0, bytes));
    }
    // conv2d: K x K weights -- through a ConvWeights buffer edge when the graph wires one, else the node's own (default
    // weights derived from sigma on the host; rf_graph_set_weights overrides).  conv2d_weights: the node that WRITES such a
    // buffer; its content is host-derived too, so it is written here and whenever its parameters or weights change (the
    // reference's node would rewrite the same values every frame).
    for (const auto& kv : plan.nodes) {
        const int kind = kv.second.type->kind;
        if (kind != OP_CONV2D && kind != OP_WEIGHTS) continue;
        const PipelineInfo* info = nullptr;
        for (const auto& u : plan.infos)
            if (std::find(u.second.members.begin(), u.second.members.end(), kv.first) != u.second.members.end()) info = &u.second;
        const int K = kv.second.conv_ksize();
        std::vector<float> w((size_t)K * K);
        auto it = kv.second.values.find("sigma");
        default_conv_weights(K, it == kv.second.values.end() ? 0.f : it->second.f, w.data());
        if (kind == OP_WEIGHTS) {
            if (!info) continue;
            for (const auto& out : info->output_ssbos) {
                float* d = g->dev_buffers.at(plan.resolve_buffer(out.first));
                g->written_by[kv.first].push_back(d);
                HIP_TRY(hipMemcpy(d, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            continue;
        }
        if (info && !info->input_ssbos.empty()) {
            auto bit = g->dev_buffers.find(plan.resolve_buffer(info->input_ssbos[0].first));
            if (bit == g->dev_buffers.end()) return fail(RF_ERR_GRAPH, "No buffer found for input " + info->input_ssbos[0].first);   // pipeline_graph.rs:269
            g->weights_of[kv.first] = bit->second;
            continue;
        }
        float* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, w.size() * sizeof(float)));
        g->dev_weights[kv.first] = d;
        g->weights_of[kv.first] = d;
        HIP_TRY(hipMemcpy(d, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // launches in execution order + the ghost rows each one reads beyond the strip
    {
        if (!g->plan.launch_error.empty()) return fail(RF_ERR_GRAPH, g->plan.launch_error);
        std::vector<LaunchDesc> descs = g->plan.launches;
        halo_schedule(descs, ctx->world > 1, exchange_mode(g), g->need_input, g->ghost);
        if (exchange_mode(g) && g->ghost > 0 && !ctx->comm)
            return fail(RF_ERR_UNSUPPORTED, "this rank has no RCCL communicator: create the graph with RF_GRAPH_NO_HALO_XCHG "
                                            "or the context with a unique id");
        for (const auto& d : descs) {
            Launch L;
            static_cast<LaunchDesc&>(L) = d;
            g->launches.push_back(L);
        }
    }
    rebuild_ops(g);
    for (const auto& L : g->launches) {
        if (L.ops.size() != 1 || L.ops[0].kind != OP_USERN) continue;
        const UserStage* u = user_stage_by_id(L.ops[0].user_id);
        if (u && u->glsl) {
            if (ctx->world > 1 && !u->buf_out.empty())
                return fail(RF_ERR_UNSUPPORTED, "node '" + L.label + "': " + u->file_name() + " writes a storage block (" + u->buf_out[0].name + "); storage buffers are per rank, so a shader that "
                                                "fills one from the invocations of some rows cannot be split into row strips");
            // a .comp node addresses the FRAME: split over ranks it needs to say how far it reads (#pragma rf radius N, rf_glsl.h)
            if (ctx->world > 1 && !u->radius_stated)
                return fail(RF_ERR_UNSUPPORTED, "node '" + L.label + "': " + u->file_name() + " does not say `#pragma rf radius N` (rows an invocation reads above / below its own); "
                                                "without it the node cannot be split into row strips");
            if (!g->dev_buffers.count(kGlslZeroTexel)) {
                float* z = nullptr;
                HIP_TRY(hipMalloc((void**)&z, 16));
                g->dev_buffers[kGlslZeroTexel] = z;
                HIP_TRY(hipMemset(z, 0, 16));
            }
            // the blocks it declares and the graph leaves unwired: the shader writes them all the same
            for (const auto* list : {&u->buf_in, &u->buf_out})
                for (const auto& b : *list) {
                    if (std::find(L.in_buffer_bindings.begin(), L.in_buffer_bindings.end(), b.binding) != L.in_buffer_bindings.end() ||
                        std::find(L.out_buffer_bindings.begin(), L.out_buffer_bindings.end(), b.binding) != L.out_buffer_bindings.end()) continue;
                    const std::string name = unwired_buffer_name(L.label, b.name);
                    if (g->dev_buffers.count(name)) continue;
                    float* d = nullptr;
                    HIP_TRY(hipMalloc((void**)&d, std::max<size_t>(b.bytes, 4)));
                    g->dev_buffers[name] = d;
                    HIP_TRY(hipMemset(d, 0, std::max<size_t>(b.bytes, 4)));
                }
        }
        if (u && !u->buf_out.empty() && (!L.out_buffers.empty() || u->glsl)) g->fills_buffers = true;
    }
    if (g->fills_buffers && opt.num_frames > 1) HIP_TRY(hipEventCreateWithFlags(&g->buffers_idle, hipEventDisableTiming));
    g->input_image = std::find(plan.images.begin(), plan.images.end(), kFileInput) != plan.images.end() ? kFileInput : "";
    g->output_image = plan.resolve(kFinalOutput);
    if (std::find(plan.images.begin(), plan.images.end(), g->output_image) == plan.images.end())
        return fail(RF_ERR_GRAPH, "the graph never writes rf:final-output");
    if (ctx->world > 1 && g->ghost > opt.height / ctx->world)
        return fail(RF_ERR_UNSUPPORTED, "strip height " + std::to_string(opt.height / ctx->world) +
                                            " is smaller than the halo " + std::to_string(g->ghost));

    // per-frame images, streams, events (PipelineGraphFrame::new, Frame::new)
    const size_t pitch = image_pitch((size_t)opt.width * bytes_per_pixel(opt.format));
    size_t max_layer = 1;
    {
        std::map<int, size_t> per_layer;
        for (const auto& L : g->launches) max_layer = std::max(max_layer, ++per_layer[L.layer]);
    }
    const bool timers = (opt.flags & RF_GRAPH_TIMERS) != 0;
    g->frames.resize((size_t)opt.num_frames);
    for (auto& f : g->frames) {
        for (const auto& name : plan.images) {
            DeviceImage img;
            img.pitch = pitch;
            const size_t rows = (size_t)Hs + 2 * (size_t)g->ghost;
            HIP_TRY(hipMalloc(&img.alloc, rows * pitch));
            img.base = (char*)img.alloc + (size_t)g->ghost * pitch;
            f.images[name] = img;
            if (std::getenv("RF_TRACE_SHAPE")) std::fprintf(stderr, "rf image: %s at %p (%zu bytes, pitch %zu)\n", name.c_str(), img.alloc, rows * pitch, pitch);
        }
        HIP_TRY(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
        for (size_t j = 1; j < max_layer; ++j) {
            hipStream_t s;
            hipEvent_t e;
            HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            f.aux.push_back(s);
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            f.join.push_back(e);
        }
        HIP_TRY(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
        if (exchange_mode(g) || g->force_split) {
            HIP_TRY(hipStreamCreateWithFlags(&f.comm, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&f.src_ready, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&f.halo_ready, hipEventDisableTiming));
        }
        // the frame fence (Frame.fence, frame.rs:47, created SIGNALED) is the slot's stream itself:
        // an idle stream synchronises immediately
        if (timers) {
            for (size_t k = 0; k < g->launches.size(); ++k) {
                hipEvent_t a, b;
                HIP_TRY(hipEventCreate(&a));
                HIP_TRY(hipEventCreate(&b));
                f.t0.push_back(a);
                f.t1.push_back(b);
            }
        }
    }
    // the fills above (storage buffers) ran on the null stream; the frames run on non-blocking streams of their own
    HIP_TRY(hipDeviceSynchronize());
    return RF_OK;
}

extern "C" rf_status rf_graph_create(rf_ctx* ctx, const rf_config* cfg, const rf_graph_options* opt, rf_graph** out)
{
    if (!ctx || !cfg || !opt || !out) return fail(RF_ERR_INVALID, "rf_graph_create: null argument");
    *out = nullptr;
    if (opt->width < 1 || opt->height < 1) return fail(RF_ERR_INVALID, "rf_graph_create: width and height must be >= 1");
    if (opt->format != RF_FORMAT_RGBA8 && opt->format != RF_FORMAT_RGBA32F) return fail(RF_ERR_INVALID, "rf_graph_create: unknown format");
    if (opt->num_frames < 1) return fail(RF_ERR_INVALID, "rf_graph_create: num_frames must be >= 1");
    rf_graph* g = new rf_graph();
    g->ctx = ctx;
    g->opt = *opt;
    rf_status st = graph_build(g, cfg);
    if (st != RF_OK) {
        std::string keep = last_error();
        rf_graph_destroy(g);
        set_error(keep);
        return st;
    }
    glsl_window_selftests(g);
    *out = g;
    return RF_OK;
}

extern "C" void rf_graph_destroy(rf_graph* g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipDeviceSynchronize();   // pipeline_graph.rs:610
    for (auto& f : g->frames) {
        destroy_graph_exec(f);
        for (auto& kv : f.images)
            if (kv.second.alloc) (void)hipFree(kv.second.alloc);
        for (auto e : f.t0) (void)hipEventDestroy(e);
        for (auto e : f.t1) (void)hipEventDestroy(e);
        for (auto e : f.join) (void)hipEventDestroy(e);
        if (f.fork) (void)hipEventDestroy(f.fork);
        if (f.src_ready) (void)hipEventDestroy(f.src_ready);
        if (f.halo_ready) (void)hipEventDestroy(f.halo_ready);
        if (f.comm) (void)hipStreamDestroy(f.comm);
        for (auto s : f.aux) (void)hipStreamDestroy(s);
        if (f.stream) (void)hipStreamDestroy(f.stream);
    }
    for (auto& kv : g->dev_weights) (void)hipFree(kv.second);
    for (auto& kv : g->dev_buffers) (void)hipFree(kv.second);
    if (g->buffers_idle) (void)hipEventDestroy(g->buffers_idle);
    if (g->d_staging) (void)hipFree(g->d_staging);
    delete g;
}

extern "C" const rf_plan* rf_graph_plan(const rf_graph* g) { return g ? &g->plan : nullptr; }

extern "C" rf_status rf_graph_strip(const rf_graph* g, int* y0, int* y1)
{
    if (!g || !y0 || !y1) return fail(RF_ERR_INVALID, "rf_graph_strip: null argument");
    *y0 = g->strip_y0;
    *y1 = g->strip_y1;
    return RF_OK;
}

static void invalidate_captures(rf_graph* g)
{
    for (auto& f : g->frames) {
        (void)hipStreamSynchronize(f.stream);
        destroy_graph_exec(f);
    }
}

extern "C" rf_status rf_graph_set_param(rf_graph* g, const char* node, const char* name, rf_param_type type, const void* value)
{
    if (!g || !node || !name || !value) return fail(RF_ERR_INVALID, "rf_graph_set_param: null argument");
    auto it = g->plan.plan.nodes.find(node);
    if (it == g->plan.plan.nodes.end()) return fail(RF_ERR_INVALID, std::string("no node named '") + node + "'");
    NodeParams& np = it->second;
    const ParamDef* pd = np.type->param(name);
    if (!pd) return fail(RF_WARN_UNKNOWN_PARAM, std::string("node type '") + np.type->name + "' has no parameter '" + name + "'");
    double v = 0.0;
    switch (type) {
        case RF_PARAM_F32: v = *(const float*)value; break;
        case RF_PARAM_I32: v = *(const int32_t*)value; break;
        case RF_PARAM_BOOL: v = *(const int32_t*)value ? 1.0 : 0.0; break;
        default: return fail(RF_ERR_INVALID, "rf_graph_set_param: bad type");
    }
    ParamValue nv;
    nv.i = 0;
    if (pd->type == PARAM_F32) nv.f = (type == RF_PARAM_F32) ? *(const float*)value : (float)v;
    else if (pd->type == PARAM_I32) nv.i = (int32_t)v;
    else nv.b = v != 0.0 ? 1 : 0;
    const ParamValue old = np.values[name];
    const Op before = np.to_op(nullptr);
    np.values[name] = nv;
    const Op after = np.to_op(nullptr);
    if (before.radius != after.radius) {
        // changes the halo, the fusion pattern and possibly the allocation: the reference
        // rebuilds the whole graph on such edits (recreate_graph, render.rs:121-136)
        np.values[name] = old;
        return fail(RF_ERR_UNSUPPORTED, std::string("parameter '") + name + "' changes the stencil radius: destroy and re-create the graph");
    }
    invalidate_captures(g);
    if ((np.type->kind == OP_CONV2D || np.type->kind == OP_WEIGHTS) && std::string(name) == "sigma") {
        // the weights are DERIVED from sigma until rf_graph_set_weights replaces them: an edit of sigma regenerates
        // them, as creating the graph with that sigma would (a conv2d fed through a buffer edge has none of its own)
        std::vector<float*> targets;
        auto wit = g->dev_weights.find(node);
        if (wit != g->dev_weights.end()) targets.push_back(wit->second);
        auto bit = g->written_by.find(node);
        if (bit != g->written_by.end()) targets = bit->second;
        if (!targets.empty()) {
            const int K = np.conv_ksize();
            std::vector<float> w((size_t)K * K);
            default_conv_weights(K, nv.f, w.data());
            HIP_TRY(hipSetDevice(g->ctx->device));
            for (auto& f : g->frames) HIP_TRY(hipStreamSynchronize(f.stream));
            for (float* d : targets) HIP_TRY(hipMemcpy(d, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    rebuild_ops(g);
    return RF_OK;
}

extern "C" rf_status rf_graph_set_weights(rf_graph* g, const char* node, const float* weights, int count)
{
    if (!g || !node || !weights) return fail(RF_ERR_INVALID, "rf_graph_set_weights: null argument");
    auto nit = g->plan.plan.nodes.find(node);
    if (nit == g->plan.plan.nodes.end()) return fail(RF_ERR_INVALID, std::string("no node named '") + node + "'");
    std::vector<float*> targets;
    auto it = g->dev_weights.find(node);
    if (it != g->dev_weights.end()) targets.push_back(it->second);
    auto bit = g->written_by.find(node);
    if (bit != g->written_by.end()) targets = bit->second;
    if (targets.empty()) {
        if (nit->second.type->kind == OP_CONV2D)
            return fail(RF_ERR_INVALID, std::string("node '") + node + "' takes its weights through a ConvWeights buffer edge: set them on the node that writes it");
        return fail(RF_ERR_INVALID, std::string("node '") + node + "' is not a conv2d or conv2d_weights node");
    }
    const int K = nit->second.conv_ksize();
    if (count != K * K) return fail(RF_ERR_INVALID, "rf_graph_set_weights: expected " + std::to_string(K * K) + " weights");
    HIP_TRY(hipSetDevice(g->ctx->device));
    for (auto& f : g->frames) HIP_TRY(hipStreamSynchronize(f.stream));
    for (float* d : targets) HIP_TRY(hipMemcpy(d, weights, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
    return RF_OK;
}

extern "C" rf_status rf_graph_set_time(rf_graph* g, float seconds)
{
    if (!g) return fail(RF_ERR_INVALID, "null graph");
    bool any = false;
    for (auto& kv : g->plan.plan.nodes) {
        for (auto& pv : kv.second.values) {
            const std::string& n = pv.first;
            if (n.size() >= 8 && n.compare(n.size() - 8, 8, "_rf_time") == 0) { pv.second.f = seconds; any = true; }
        }
    }
    if (any) { invalidate_captures(g); rebuild_ops(g); }
    return RF_OK;
}

// ---- input -----------------------------------------------------------------------
static rf_status after_input_write(rf_graph* g, FrameSlot& f, hipStream_t stream)
{
    // over-fetch mode: the neighbours' rows the whole frame will read, exchanged once
    if (g->ctx->world > 1 && !exchange_mode(g) && g->need_input > 0)
        return exchange_rows(g, f.images.at(g->input_image), g->need_input, stream);
    return RF_OK;
}

static rf_status need_input(rf_graph* g, const char* who)
{
    if (!g) return fail(RF_ERR_INVALID, std::string(who) + ": null graph");
    if (g->input_image.empty()) return fail(RF_ERR_INVALID, std::string(who) + ": the graph has no 'input'");
    HIP_TRY(hipSetDevice(g->ctx->device));
    return RF_OK;
}

extern "C" rf_status rf_graph_upload_raw(rf_graph* g, const void* texels, size_t row_stride)
{
    rf_status st = need_input(g, "rf_graph_upload_raw");
    if (st != RF_OK) return st;
    if (!texels) return fail(RF_ERR_INVALID, "rf_graph_upload_raw: null buffer");
    const size_t row_bytes = (size_t)g->opt.width * bytes_per_pixel(g->opt.format);
    if (row_stride < row_bytes) return fail(RF_ERR_INVALID, "rf_graph_upload_raw: row_stride smaller than a row");
    for (auto& f : g->frames) {
        const DeviceImage& img = f.images.at(g->input_image);
        HIP_TRY(hipMemcpy2DAsync(img.base, img.pitch, texels, row_stride, row_bytes, (size_t)strip_rows_of(g), hipMemcpyHostToDevice, f.stream));
        st = after_input_write(g, f, f.stream);
        if (st != RF_OK) return st;
        HIP_TRY(hipStreamSynchronize(f.stream));   // the caller's buffer is free on return
    }
    return RF_OK;
}

extern "C" rf_status rf_graph_upload_srgb8(rf_graph* g, const uint8_t* rgba, size_t row_stride)
{
    rf_status st = need_input(g, "rf_graph_upload_srgb8");
    if (st != RF_OK) return st;
    if (!rgba) return fail(RF_ERR_INVALID, "rf_graph_upload_srgb8: null buffer");
    const size_t row_bytes = (size_t)g->opt.width * 4;
    if (row_stride < row_bytes) return fail(RF_ERR_INVALID, "rf_graph_upload_srgb8: row_stride smaller than a row");
    const int Hs = strip_rows_of(g);
    const size_t need = row_bytes * (size_t)Hs;   // staging buffer, render.rs:552
    if (g->staging_bytes < need) {
        if (g->d_staging) (void)hipFree(g->d_staging);
        g->d_staging = nullptr;
        g->staging_bytes = 0;
        HIP_TRY(hipMalloc((void**)&g->d_staging, need));
        g->staging_bytes = need;
    }
    for (auto& f : g->frames) {
        HIP_TRY(hipMemcpy2DAsync(g->d_staging, row_bytes, rgba, row_stride, row_bytes, (size_t)Hs, hipMemcpyHostToDevice, f.stream));
        HIP_TRY(launch_upload_srgb8(g->opt.format, g->d_staging, row_bytes, f.images.at(g->input_image).view(), g->opt.width, Hs,
                                    g->ctx->d_tables, f.stream));
        st = after_input_write(g, f, f.stream);
        if (st != RF_OK) return st;
        HIP_TRY(hipStreamSynchronize(f.stream));
    }
    return RF_OK;
}

static void fill_rows(const rf_graph* g, int& lo, int& hi)
{
    // generated fills can write the ghost rows directly: no exchange needed
    const int Hs = g->strip_y1 - g->strip_y0, need = g->ctx->world > 1 ? g->ghost : 0;
    lo = std::max(-need, -g->strip_y0);
    hi = std::min(Hs + need, g->opt.height - g->strip_y0);
}

extern "C" rf_status rf_graph_fill_synthetic(rf_graph* g, uint32_t seed)
{
    rf_status st = need_input(g, "rf_graph_fill_synthetic");
    if (st != RF_OK) return st;
    int lo, hi;
    fill_rows(g, lo, hi);
    for (auto& f : g->frames) {
        HIP_TRY(launch_fill_synthetic(g->opt.format, f.images.at(g->input_image).view(), g->opt.width, lo, hi, g->strip_y0, seed, f.stream));
        HIP_TRY(hipStreamSynchronize(f.stream));
    }
    return RF_OK;
}

extern "C" rf_status rf_graph_fill_structured(rf_graph* g)
{
    rf_status st = need_input(g, "rf_graph_fill_structured");
    if (st != RF_OK) return st;
    int lo, hi;
    fill_rows(g, lo, hi);
    for (auto& f : g->frames) {
        HIP_TRY(launch_fill_structured(g->opt.format, f.images.at(g->input_image).view(), g->opt.width, lo, hi, g->strip_y0, g->opt.height, f.stream));
        HIP_TRY(hipStreamSynchronize(f.stream));
    }
    return RF_OK;
}

// ---- execute -----------------------------------------------------------------------
static rf_status slot_of(rf_graph* g, int slot, FrameSlot** out, const char* who)
{
    if (!g) return fail(RF_ERR_INVALID, std::string(who) + ": null graph");
    if (slot < 0 || slot >= (int)g->frames.size()) return fail(RF_ERR_INVALID, std::string(who) + ": frame slot out of range");
    HIP_TRY(hipSetDevice(g->ctx->device));
    *out = &g->frames[(size_t)slot];
    return RF_OK;
}

static bool use_hipgraph(const rf_graph* g)
{
    return (g->opt.flags & RF_GRAPH_HIPGRAPH) && !(g->opt.flags & RF_GRAPH_TIMERS) && g->ctx->world == 1;
}

static rf_status submit_frame_unordered(rf_graph* g, FrameSlot& f);

static rf_status submit_frame(rf_graph* g, FrameSlot& f)
{
    // device-filled storage buffers are shared by the frame slots: this frame starts when the previous one (any slot) is done
    const bool ordered = g->fills_buffers && g->frames.size() > 1 && g->buffers_idle;
    if (ordered && g->buffers_idle_set) HIP_TRY(hipStreamWaitEvent(f.stream, g->buffers_idle, 0));
    rf_status st = submit_frame_unordered(g, f);
    if (st == RF_OK && ordered) {
        HIP_TRY(hipEventRecord(g->buffers_idle, f.stream));
        g->buffers_idle_set = true;
    }
    return st;
}

static rf_status submit_frame_unordered(rf_graph* g, FrameSlot& f)
{
    const bool timers = (g->opt.flags & RF_GRAPH_TIMERS) != 0;
    if (use_hipgraph(g)) {
        if (!f.graph_exec) {
            HIP_TRY(hipStreamBeginCapture(f.stream, hipStreamCaptureModeThreadLocal));
            rf_status st = issue_frame(g, f, false);
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(f.stream, &graph);
            if (st != RF_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
            if (e != hipSuccess) return fail(RF_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            f.graph = graph;
            HIP_TRY(hipGraphInstantiate(&f.graph_exec, graph, nullptr, nullptr, 0));
        }
        HIP_TRY(hipGraphLaunch(f.graph_exec, f.stream));
        return RF_OK;
    }
    rf_status st = issue_frame(g, f, timers);
    if (st == RF_OK && timers) f.timed_once = true;
    return st;
}

extern "C" rf_status rf_graph_execute(rf_graph* g, int frame_slot)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_execute");
    if (st != RF_OK) return st;
    // the fence of this slot is the slot's own stream (every side stream joins it before the
    // frame ends): no event packet between consecutive frames
    return submit_frame(g, *f);
}

extern "C" rf_status rf_graph_wait(rf_graph* g, int frame_slot)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_wait");
    if (st != RF_OK) return st;
    HIP_TRY(hipStreamSynchronize(f->stream));
    return RF_OK;
}

// ---- output ------------------------------------------------------------------------
static rf_status download_image(rf_graph* g, FrameSlot& f, const std::string& name, void* texels, size_t row_stride)
{
    const size_t row_bytes = (size_t)g->opt.width * bytes_per_pixel(g->opt.format);
    if (!texels) return fail(RF_ERR_INVALID, "download: null buffer");
    if (row_stride < row_bytes) return fail(RF_ERR_INVALID, "download: row_stride smaller than a row");
    auto it = f.images.find(name);
    if (it == f.images.end()) return fail(RF_ERR_INVALID, "no image named '" + name + "'");
    HIP_TRY(hipStreamSynchronize(f.stream));
    HIP_TRY(hipMemcpy2D(texels, row_stride, it->second.base, it->second.pitch, row_bytes, (size_t)strip_rows_of(g), hipMemcpyDeviceToHost));
    return RF_OK;
}

extern "C" rf_status rf_graph_download_raw(rf_graph* g, int frame_slot, void* texels, size_t row_stride)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_download_raw");
    if (st != RF_OK) return st;
    return download_image(g, *f, g->output_image, texels, row_stride);
}

extern "C" rf_status rf_graph_download_rows(rf_graph* g, int frame_slot, int y0, int y1, void* texels, size_t row_stride)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_download_rows");
    if (st != RF_OK) return st;
    const size_t row_bytes = (size_t)g->opt.width * bytes_per_pixel(g->opt.format);
    if (!texels) return fail(RF_ERR_INVALID, "rf_graph_download_rows: null buffer");
    if (row_stride < row_bytes) return fail(RF_ERR_INVALID, "rf_graph_download_rows: row_stride smaller than a row");
    if (y0 < 0 || y1 > strip_rows_of(g) || y0 >= y1) return fail(RF_ERR_INVALID, "rf_graph_download_rows: rows outside the strip");
    const DeviceImage& img = f->images.at(g->output_image);
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipMemcpy2D(texels, row_stride, img.base + (size_t)y0 * img.pitch, img.pitch, row_bytes, (size_t)(y1 - y0), hipMemcpyDeviceToHost));
    return RF_OK;
}

extern "C" rf_status rf_graph_download_image(rf_graph* g, int frame_slot, const char* resource, void* texels, size_t row_stride)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_download_image");
    if (st != RF_OK) return st;
    if (!resource) return fail(RF_ERR_INVALID, "rf_graph_download_image: null resource name");
    return download_image(g, *f, g->plan.plan.resolve(resource), texels, row_stride);
}

extern "C" rf_status rf_graph_download_srgb8(rf_graph* g, int frame_slot, uint8_t* rgba, size_t row_stride)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_download_srgb8");
    if (st != RF_OK) return st;
    if (!rgba) return fail(RF_ERR_INVALID, "rf_graph_download_srgb8: null buffer");
    const size_t row_bytes = (size_t)g->opt.width * 4;
    if (row_stride < row_bytes) return fail(RF_ERR_INVALID, "rf_graph_download_srgb8: row_stride smaller than a row");
    const int Hs = strip_rows_of(g);
    const size_t need = row_bytes * (size_t)Hs;
    if (g->staging_bytes < need) {
        if (g->d_staging) (void)hipFree(g->d_staging);
        g->d_staging = nullptr;
        g->staging_bytes = 0;
        HIP_TRY(hipMalloc((void**)&g->d_staging, need));
        g->staging_bytes = need;
    }
    HIP_TRY(launch_download_srgb8(g->opt.format, f->images.at(g->output_image).view(), g->d_staging, row_bytes, g->opt.width, Hs,
                                  g->ctx->d_tables, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipMemcpy2D(rgba, row_stride, g->d_staging, row_bytes, row_bytes, (size_t)Hs, hipMemcpyDeviceToHost));
    return RF_OK;
}

// ---- timing ------------------------------------------------------------------------
extern "C" rf_status rf_graph_node_times(rf_graph* g, int frame_slot, const char** names, float* ms, int* n)
{
    FrameSlot* f;
    rf_status st = slot_of(g, frame_slot, &f, "rf_graph_node_times");
    if (st != RF_OK) return st;
    if (!n) return fail(RF_ERR_INVALID, "rf_graph_node_times: null count");
    if (!(g->opt.flags & RF_GRAPH_TIMERS)) return fail(RF_ERR_INVALID, "graph was created without RF_GRAPH_TIMERS");
    const int cap = *n;
    *n = 0;
    if (!f->timed_once) return RF_OK;   // no frame recorded yet (current_query_index == 0, vkutils.rs:107-109)
    HIP_TRY(hipStreamSynchronize(f->stream));
    std::vector<std::pair<std::string, float>> rows;
    for (size_t k = 0; k < g->launches.size(); ++k) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, f->t0[k], f->t1[k]));
        rows.push_back({g->launches[k].label, t});
    }
    std::sort(rows.begin(), rows.end());   // BTreeMap order, vkutils.rs:50
    g->time_names.clear();
    for (auto& r : rows) g->time_names.push_back(r.first);
    int count = 0;
    for (size_t k = 0; k < rows.size() && count < cap; ++k, ++count) {
        if (names) names[count] = g->time_names[k].c_str();
        if (ms) ms[count] = rows[k].second;
    }
    *n = count;
    return RF_OK;
}

extern "C" rf_status rf_graph_times_string(rf_graph* g, int frame_slot, char* buf, size_t cap)
{
    if (!buf || cap == 0) return fail(RF_ERR_INVALID, "rf_graph_times_string: null buffer");
    buf[0] = 0;
    const char* names[256];
    float ms[256];
    int n = 256;
    rf_status st = rf_graph_node_times(g, frame_slot, names, ms, &n);
    if (st != RF_OK) return st;
    std::string s;
    char tmp[64];
    for (int i = 0; i < n; ++i) {   // "{}: {:.3}ms, " vkutils.rs:126
        std::snprintf(tmp, sizeof(tmp), "%.3fms", ms[i]);
        s += std::string(names[i]) + ": " + tmp + (i + 1 < n ? ", " : "");
    }
    std::snprintf(buf, cap, "%s", s.c_str());
    return RF_OK;
}

extern "C" rf_status rf_graph_time_frames(rf_graph* g, int iters, float* total_ms)
{
    FrameSlot* f;
    rf_status st = slot_of(g, 0, &f, "rf_graph_time_frames");
    if (st != RF_OK) return st;
    if (iters < 1 || !total_ms) return fail(RF_ERR_INVALID, "rf_graph_time_frames: bad argument");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto body = [&]() -> rf_status {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, f->stream));
        for (int i = 0; i < iters; ++i) {
            rf_status s2 = submit_frame(g, *f);
            if (s2 != RF_OK) return s2;
        }
        HIP_TRY(hipEventRecord(e1, f->stream));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(total_ms, e0, e1));
        return RF_OK;
    };
    st = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return st;
}

// Frames that rotate over ALL frame slots -- frame i reads and writes slot i % num_frames's images -- submitted to ONE
// queue (slot 0's stream), which is how the reference runs its frames in flight: a command buffer and images per frame,
// one queue (src/main.rs:164-170, src/vulkan/core.rs:123).  With enough slots no cache line is touched twice within the
// Infinity Cache's 256 MiB: the cache-cold figure of bench.py.
extern "C" rf_status rf_graph_time_frames_rotating(rf_graph* g, int iters, float* total_ms)
{
    FrameSlot* f0;
    rf_status st = slot_of(g, 0, &f0, "rf_graph_time_frames_rotating");
    if (st != RF_OK) return st;
    if (iters < 1 || !total_ms) return fail(RF_ERR_INVALID, "rf_graph_time_frames_rotating: bad argument");
    if (use_hipgraph(g) || exchange_mode(g)) return fail(RF_ERR_UNSUPPORTED, "rf_graph_time_frames_rotating: not with RF_GRAPH_HIPGRAPH or a halo exchange (their streams belong to a slot)");
    for (auto& f : g->frames) HIP_TRY(hipStreamSynchronize(f.stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t q = f0->stream;
    auto body = [&]() -> rf_status {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, q));
        for (int i = 0; i < iters; ++i) {
            FrameSlot& f = g->frames[(size_t)i % g->frames.size()];
            hipStream_t own = f.stream;
            f.stream = q;                                  // this slot's images, the shared queue
            rf_status s2 = submit_frame(g, f);
            f.stream = own;
            if (s2 != RF_OK) return s2;
        }
        HIP_TRY(hipEventRecord(e1, q));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(total_ms, e0, e1));
        return RF_OK;
    };
    st = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return st;
}

extern "C" rf_status rf_graph_time_each_frame(rf_graph* g, int iters, float* ms_each)
{
    FrameSlot* f;
    rf_status st = slot_of(g, 0, &f, "rf_graph_time_each_frame");
    if (st != RF_OK) return st;
    if (iters < 1 || !ms_each) return fail(RF_ERR_INVALID, "rf_graph_time_each_frame: bad argument");
    std::vector<hipEvent_t> ev(2 * (size_t)iters, nullptr);
    auto body = [&]() -> rf_status {
        for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
        for (int i = 0; i < iters; ++i) {
            HIP_TRY(hipEventRecord(ev[2 * (size_t)i], f->stream));
            rf_status s2 = submit_frame(g, *f);
            if (s2 != RF_OK) return s2;
            HIP_TRY(hipEventRecord(ev[2 * (size_t)i + 1], f->stream));
        }
        HIP_TRY(hipStreamSynchronize(f->stream));
        for (int i = 0; i < iters; ++i) HIP_TRY(hipEventElapsedTime(&ms_each[i], ev[2 * (size_t)i], ev[2 * (size_t)i + 1]));
        return RF_OK;
    };
    st = body();
    for (auto e : ev)
        if (e) (void)hipEventDestroy(e);
    return st;
}

extern "C" const char* rf_graph_note(const rf_graph* g) { return g ? g->jit_note.c_str() : ""; }

extern "C" rf_status rf_graph_time_launch(rf_graph* g, int launch, int iters, float* avg_ms)
{
    FrameSlot* f;
    rf_status st = slot_of(g, 0, &f, "rf_graph_time_launch");
    if (st != RF_OK) return st;
    if (launch < 0 || launch >= (int)g->launches.size() || iters < 1 || !avg_ms) return fail(RF_ERR_INVALID, "rf_graph_time_launch: bad argument");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    auto body = [&]() -> rf_status {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        const Launch& L = g->launches[(size_t)launch];
        const Geom geo = launch_geom(g, L);
        auto once = [&]() -> hipError_t {
            if (L.ops.size() == 1 && L.ops[0].kind == OP_SPLIT) {
                Image luma{}, chroma{};
                for (size_t k = 0; k < L.dsts.size(); ++k) (L.dst_bindings[k] == 1 ? luma : chroma) = f->images.at(L.dsts[k]).view();
                return launch_split_luma(g->opt.format, f->images.at(L.src[0]).view(), luma, chroma, geo, f->stream);
            }
            if (L.ops.size() == 1 && L.ops[0].kind == OP_USERN) return launch_user_node(g, L, *f, geo, f->stream);
            if (L.ops.size() == 1 && L.ops[0].kind == OP_MIX)
                return launch_mix(g->opt.format, f->images.at(L.src[0]).view(), f->images.at(L.src[1]).view(), f->images.at(L.dst).view(), geo,
                                  L.ops[0].slope, f->stream);
            return launch_ops(g->opt.format, L.ops.data(), (int)L.ops.size(), f->images.at(L.src[0]).view(), f->images.at(L.dst).view(), geo,
                              g->tune, f->stream);
        };
        HIP_TRY(once());
        HIP_TRY(hipEventRecord(e0, f->stream));
        for (int i = 0; i < iters; ++i) HIP_TRY(once());
        HIP_TRY(hipEventRecord(e1, f->stream));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        return RF_OK;
    };
    st = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st == RF_OK) *avg_ms = ms / (float)iters;
    return st;
}

// `iters` whole frames on slot 0 with a hipEvent pair around EVERY launch (recorded on
// the stream that launch runs on): average milliseconds per launch, in execution order.
extern "C" rf_status rf_graph_time_launches(rf_graph* g, int iters, float* avg_ms, int n)
{
    FrameSlot* f;
    rf_status st = slot_of(g, 0, &f, "rf_graph_time_launches");
    if (st != RF_OK) return st;
    const size_t nl = g->launches.size();
    if (iters < 1 || !avg_ms || n < (int)nl) return fail(RF_ERR_INVALID, "rf_graph_time_launches: bad argument");
    std::vector<hipEvent_t> ev(2 * nl * (size_t)iters, nullptr);
    std::vector<hipEvent_t> keep0 = f->t0, keep1 = f->t1;
    auto body = [&]() -> rf_status {
        for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
        f->t0.assign(nl, nullptr);
        f->t1.assign(nl, nullptr);
        for (int it = 0; it < iters; ++it) {
            for (size_t k = 0; k < nl; ++k) {
                f->t0[k] = ev[2 * ((size_t)it * nl + k)];
                f->t1[k] = ev[2 * ((size_t)it * nl + k) + 1];
            }
            rf_status s2 = issue_frame(g, *f, true);
            if (s2 != RF_OK) return s2;
        }
        HIP_TRY(hipStreamSynchronize(f->stream));
        for (size_t k = 0; k < nl; ++k) {
            double sum = 0.0;
            for (int it = 0; it < iters; ++it) {
                float t = 0.f;
                HIP_TRY(hipEventElapsedTime(&t, ev[2 * ((size_t)it * nl + k)], ev[2 * ((size_t)it * nl + k) + 1]));
                sum += t;
            }
            avg_ms[k] = (float)(sum / iters);
        }
        return RF_OK;
    };
    st = body();
    f->t0 = keep0;
    f->t1 = keep1;
    for (auto e : ev)
        if (e) (void)hipEventDestroy(e);
    return st;
}

// One-rank RCCL round trip on `device`: communicator of world 1, a grouped send+recv to
// self of `bytes` bytes, result compared.  Proves that librccl loads and that the entry
// points are called with the right ABI on this machine; the multi-rank pattern itself is
// covered by tests/test_dist_gloo.py.
extern "C" rf_status rf_comm_selftest(int device, size_t bytes)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RF_ERR_NO_DEVICE, "no HIP device (librfhip has no CPU fallback)");
    if (device < 0 || device >= n || bytes == 0) return fail(RF_ERR_INVALID, "rf_comm_selftest: bad argument");
    HIP_TRY(hipSetDevice(device));
    std::string err;
    RcclLib* lib = rccl_lib(err);
    if (!lib) return fail(RF_ERR_DEVICE, err);
    NcclId id;
    NCCL_TRY(lib, lib->GetUniqueId(&id));
    void* comm = nullptr;
    NCCL_TRY(lib, lib->CommInitRank(&comm, 1, id, 0));
    char *a = nullptr, *b = nullptr;
    hipStream_t s = nullptr;
    std::vector<char> host(bytes), back(bytes);
    for (size_t i = 0; i < bytes; ++i) host[i] = (char)(i * 131u + 7u);
    auto body = [&]() -> rf_status {
        HIP_TRY(hipMalloc((void**)&a, bytes));
        HIP_TRY(hipMalloc((void**)&b, bytes));
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipMemcpyAsync(a, host.data(), bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(b, 0, bytes, s));
        NCCL_TRY(lib, lib->GroupStart());
        NCCL_TRY(lib, lib->Send(a, bytes, kNcclChar, 0, comm, s));
        NCCL_TRY(lib, lib->Recv(b, bytes, kNcclChar, 0, comm, s));
        NCCL_TRY(lib, lib->GroupEnd());
        HIP_TRY(hipMemcpyAsync(back.data(), b, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (back != host) return fail(RF_ERR_DEVICE, "rf_comm_selftest: received bytes differ from the bytes sent");
        return RF_OK;
    };
    rf_status st = body();
    std::string keep = last_error();
    if (s) (void)hipStreamDestroy(s);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    lib->CommDestroy(comm);
    if (st != RF_OK) set_error(keep);
    return st;
}
