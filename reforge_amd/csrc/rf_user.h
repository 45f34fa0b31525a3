// rf_user.h -- filter types that are FILES: `{shader_path}/{type}.stage.hip`.
//
// In the reference a filter type T is the file {shader_path}/T.comp (src/config/config.rs:59-75), compiled and reflected when
// the graph is built (src/vulkan/shader.rs:29-59, :106-160) and rebuilt when it changes (src/render.rs:225-249,
// src/vulkan/pipeline_graph.rs:329-356).  Here the built-in types are a registry of hand-written stages (rf_plan.cpp); a type
// the registry lacks is looked for as {shader_path}/T.stage.hip: a small HIP fragment (struct Params, RADIUS, apply()) that
// becomes a row stage of the stream kernel (rf_stream_dev.h, StUser) and is compiled by hiprtc at rf_graph_create together
// with whatever it is fused with.  "Reflection" = parsing `struct Params { ... }`: the member names are the config's
// parameter names, f32 / i32 / bool as in render.rs:169-185, `_rf_time` members honoured (render.rs:190,:212-223).
//
// A file may also DECLARE its images -- `RF_INPUTS(base_image, detail_image); RF_OUTPUTS(output_image, mask_image);` -- the
// counterpart of a .comp file's `uniform image2D` variables, found by name (shader.rs:151-153; a config wires them with
// `-> node:detail_image`).  Such a type is a NODE with up to 4 input and 4 output images and a kernel of its own
// (rf_user_dev.h, user_node_kernel; a point op: RADIUS 0), `apply(const Params&, const f4 (&in)[NI], f4 (&out)[NO])`; a name
// listed on both sides is ONE binding, i.e. written in place (pipeline_graph.rs:228,:402-406).  With RADIUS R > 0 such a node reads
// its inputs through WINDOWS -- `apply(const Params&, const Window (&in)[NI], f4 (&out)[NO])`, `in[k].at(dx, dy)` for |dx|, |dy| <= R,
// clamp-to-edge -- the counterpart of a shader that calls imageLoad at neighbouring coordinates; RADIUS 2..15 always takes this form
// (a file without declarations then has the one input and output every shader has, passthrough.comp:4-5).  A node may also read one storage
// buffer and write one, found by the block's TYPE name like the reference's (shader.rs:144-147, vkutils.rs:165-170):
// `RF_BUFFER_IN(ToneCurve, 256);` adds a `const float*` argument to apply(); `RF_BUFFER_OUT(ToneCurve, 256);` asks for
// `RF_STAGE float fill(const Params&, int i)`, evaluated for i = 0..255 in front of the node's own kernel, every frame.
#pragma once

#include <string>
#include <vector>

#include "rf_plan.h"

namespace rf {

struct UserParam {
    std::string name;
    ParamType type;
    int offset, size;
};

struct UserStage {
    int id = -1;                  // index in the process-wide table (an edited file gets a NEW entry: old graphs keep theirs)
    std::string type_name, path, text;
    std::string ident;            // "u_<hash of the text>": the namespace its wrapper lives in
    int radius = 0;
    bool multi = false;           // declares RF_INPUTS / RF_OUTPUTS: a node with a kernel of its own (OP_USERN), not a row stage
    std::vector<std::string> inputs, outputs;      // image variable names in declaration order ({input_image} / {output_image} by default)
    std::vector<int> in_binding, out_binding;      // their bindings: inputs 0.., outputs after them; a name on both sides shares one
    // storage buffers of a node (RF_BUFFER_IN / RF_BUFFER_OUT: block TYPE name + number of floats; shader.rs:144-147): at most one
    // read (handed to apply() as `const float*`) and one written (element i = fill(params, i), by a small kernel in front of the node's)
    struct Buffer { std::string name; int count = 0, binding = -1; size_t bytes = 0; int slot = -1; };     // slot: GLSL, the index in GlslArgs::buf
    std::vector<Buffer> buf_in, buf_out;
    // {type}.comp, the reference's own plugin form (rf_glsl.h): translated GLSL, always a node with a kernel of its own
    // (rfglsl::glsl_node_kernel), any number of images and storage blocks; a block that is neither readonly nor writeonly is on
    // both lists (one binding: updated in place, pipeline_graph.rs:240-247)
    bool glsl = false;
    bool radius_stated = true;    // GLSL: the file says `#pragma rf radius N` (what a row-strip partition needs to know)
    std::vector<std::string> glsl_images;          // image variables in declaration order (= GlslArgs::img)
    std::vector<int> glsl_image_binding;
    std::vector<bool> glsl_image_written;          // the variable is not readonly
    int glsl_buffers = 0, glsl_groups[3] = {1, 1, 1};
    bool glsl_grouped = false;
    bool glsl_window = false;     // recognised as a translation-invariant stencil: its interior runs on user_node_kernel (LDS tiles), its border ring on the generic kernel
    std::string glsl_source;      // the translation
    std::string file_name() const { return type_name + (glsl ? ".comp" : ".stage.hip"); }
    std::vector<UserParam> params;
    int params_size = 1;          // sizeof(Params) as the device compiler lays it out (checked there by static_assert)
    long long mtime_ns = 0;
    NodeType node_type;           // the registry entry (names point into `params` / `type_name`)
    // a .comp file that is a ROW STAGE (a point shader, a 3 x 3 stencil) is also a node with a kernel of its own: the same entry with kind
    // OP_USERN, handed out when a plan asks for it (kPlanGlslNodes) or after rf_graph_create found the stage and the generic kernel to differ
    NodeType node_type_alt;
    bool has_alt = false;
    std::string row_stage = "Stage";      // rfuser::<ident>::<row_stage>: the struct StUser is instantiated with
    std::string wrapper() const;  // device source appended to the run-time compiler's translation unit
};

// [host] directory searched for {type}.stage.hip ("" = none); process-wide like Render's shader_path (render.rs:537-588)
void set_shader_path(const std::string& dir);
const std::string& shader_path();
// the stage of a type the built-in registry lacks: loads / reloads {shader_path}/{type}.stage.hip, or, when there is no such
// file, {shader_path}/{type}.comp.  nullptr: no such file (err empty) or a file that does not parse (err set)
const UserStage* user_stage_for_type(const std::string& type, std::string& err);
// the registry entry of a stage: node_type, or node_type_alt when the caller wants a node (or the row stage has been given up)
const NodeType* user_stage_node_type(const UserStage* u, bool want_node);
// rf_graph_create found the row stage of this file to differ from its generic kernel: from now on the type is a node (process-wide)
void user_stage_give_up_row_stage(int id);
const UserStage* user_stage_by_id(int id);
const UserStage* user_stage_of(const NodeType* t);
// parse one stage file's text (exposed for the tests of the parser)
bool parse_user_stage(const std::string& type, const std::string& text, UserStage& out, std::string& err);
// the same for {type}.comp: translate (rf_glsl.h) and fill the entry from the shader's reflection
bool parse_glsl_stage(const std::string& type, const std::string& text, UserStage& out, std::string& err);
// 0 (default): a built-in type wins over a file of the same name; 1: a file in {shader_path} wins -- the reference's rule, where
// the file IS the type (config.rs:59-75): a reforge user's shader directory then runs as it is, built-in names included
void set_files_first(bool on);
bool files_first();

}  // namespace rf
