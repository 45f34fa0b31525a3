// rf_glsl.h -- filter types in the reference's OWN plugin form: `{shader_path}/{type}.comp`, a GLSL 450 compute shader.
//
// In the reference that file is all a filter type is (src/config/config.rs:59-75): shaderc compiles it
// (src/vulkan/shader.rs:73-93), spirv-reflect lists its bindings -- storage images by variable NAME, storage buffers by
// block TYPE name, uniform-block members by member name (shader.rs:106-160, pipeline_graph.rs:276-292) -- and
// vkCmdDispatch(ceil(W/16), ceil(H/16), 1) runs it (src/vulkan/command.rs:166-194).  There is no GLSL compiler in this
// image and none is needed: the subset a compute filter uses is C with vector types, so the file is TRANSLATED to HIP
// device source here (host only, no dependency) and compiled by hiprtc at rf_graph_create together with
// rf_glsl_dev.h, which supplies the GLSL types and built-ins and the kernel that runs one invocation per thread.
//
// Reflection (what spirv-reflect gives the reference) falls out of the same parse:
//   layout(binding = B, <format>) uniform [readonly|writeonly] image2D name;      -> image variable `name`
//   layout(binding = B) uniform sampler2D name;                                   -> the same, read through the graph's sampler (texture(), texelFetch())
//   layout(binding = B) uniform Block { float|int|uint|bool|vecN|... members; } [instance];  -> uniform members (std140)
//   layout(std430, binding = B) [readonly|writeonly] buffer Block { members } [instance];    -> storage buffer `Block` (std430)
//   layout(local_size_x = X, local_size_y = Y, local_size_z = Z) in;
//   #pragma rf radius N     (ignored by a GLSL compiler) rows above / below its own that an invocation reads: what a row-strip
//                           partition must exchange or over-fetch for the node.  A file without it runs on one GPU only.
// Translation: globals become members of `template <class RfgPx> struct RfgShader`, functions its member functions (so every
// function sees every uniform, image and built-in variable, in any order); `vecN(...)` constructors become mk_vecN(...),
// array constructors braces, sized arrays `rfg_arr<T, (n)>` (values, as in GLSL), `out` / `inout` parameters references, literals `float`, `a == b` a call (one bool also for vectors); `shared` variables move in front of
// the struct as LDS variables; `precise`, precision qualifiers and prototypes go.  Not translated (the file is refused with
// a message, as the reference refuses a file that does not compile: Option::None + warning, shader.rs:92): samplers other
// than sampler2D, images other than image2D, nested structs in blocks, unsized arrays, double precision, image atomics and atomic counters
// (atomicAdd ... atomicCompSwap on storage-block and shared integers are translated: HIP's atomics behind reference parameters).
#pragma once

#include <string>
#include <vector>

namespace rf {

struct GlslMember {
    std::string name;             // as the reference keys it: "member", or "instance.member" for a named block instance
    char base = 'f';              // 'f' float, 'i' int, 'u' uint, 'b' bool
    int comps = 1;                // 1 scalar, 2..4 vector
    int cols = 1;                 // 2..4: a matrix of `cols` columns of `comps` rows
    std::vector<int> dims;        // array dimensions, outermost first
    int offset = 0;               // bytes from the start of the block
    int stride = 0;               // bytes between array elements (innermost dimension)
    int bytes = 0;                // padded size of the member
};

struct GlslBlock {
    std::string type_name, instance;
    int binding = -1;
    bool readonly = false, writeonly = false;
    std::vector<GlslMember> members;
    int bytes = 0;                // the block's size (std140 for uniform blocks, std430 for storage blocks unless it says std140)
    int ubo_base = 0;             // uniform blocks: where the block starts in the node's uniform bytes
};

struct GlslImageVar {
    std::string name;
    int binding = -1;
    bool readonly = false, writeonly = false;
    bool sampled = false;         // `uniform sampler2D`: bound as a combined image sampler (shader.rs:98), read through texture() / texelFetch()
};

struct GlslShader {
    std::vector<GlslImageVar> images;      // declaration order = the order of GlslArgs::img
    std::vector<GlslBlock> ubos, ssbos;    // ssbos: declaration order = the order of GlslArgs::buf
    int lx = 1, ly = 1, lz = 1;
    bool grouped = false;                  // uses workgroup built-ins, shared variables or barrier(): dispatched in the file's own workgroups
    int radius = -1;                       // #pragma rf radius N; -1 = not stated
    bool point = false;                    // recognised as a point operation on one image: also a row stage of the stream kernel (fuses)
    bool stencil = false;                  // recognised as a translation-invariant stencil of the stated radius: also runs on the LDS-tiled window kernel
    std::string stencil_why;               // why not (the first reason the analysis met; "" for a point shader or a recognised stencil)
    bool box = false;                      // ... of radius 1 with one image in and one out: also a 3 x 3 row stage of the stream kernel (fuses)
    int ubo_bytes = 0;
    std::string source;                    // namespace rfglsl { namespace <ident> { ... RfgShader<Px> ... RfgInfo ... } }
};

constexpr int kGlslMaxImages = 32, kGlslMaxBuffers = 32, kGlslMaxUniformBytes = 256;
constexpr int kGlslTranslatorVersion = 3;      // part of a shader's identity (rf_user.cpp): code objects cached on disk follow the translator

// `ident`: the namespace the translation lives in (unique per file text).  false + err ("type.comp:LINE: ...") if the file
// uses something outside the subset.
bool glsl_translate(const std::string& type, const std::string& text, const std::string& ident, GlslShader& out, std::string& err);

// the reflection as JSON (rf_glsl_reflect of the C ABI; tests compare it with the stage files' and with hand-written expectations)
std::string glsl_reflection_json(const GlslShader& s);

}  // namespace rf
