// rf_plan.h -- node-type registry + graph planner (host only).
//
// Registry: what the reference learns by compiling `{shader_path}/{type}.comp` and
// reflecting its SPIR-V (src/vulkan/shader.rs:29-160, vkutils.rs:140-196): image
// variable name -> binding index, uniform-block members -> (type).  Here a node type
// is a hand-written HIP stage list instead of a GLSL file, so the table is static.
//
// Planner: PipelineGraph::order_by_execution (pipeline_graph.rs:429-497) and
// PipelineGraph::reusable_image_remapping (pipeline_graph.rs:358-427), applied to the
// node graph -- or, with fusion enabled, to the graph whose nodes are fused chains.
#pragma once

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "rf_config.h"
#include "rf_kernels.h"

namespace rf {

enum ParamType { PARAM_F32 = 0, PARAM_I32 = 1, PARAM_BOOL = 2 };   // render.rs:171-184

struct ParamDef {
    const char* name;
    ParamType type;
};

struct NodeType {
    const char* name;
    OpKind kind;
    int fixed_radius;                                     // -1: from the `radius` / `ksize` parameter
    std::vector<std::pair<const char*, int>> images;      // variable name -> binding (shader.rs:151-153)
    std::vector<ParamDef> params;                         // uniform members (pipeline_graph.rs:276-292)
    // storage buffers, found by the block TYPE name (shader.rs:144-147): (name, binding, bytes of the block's members)
    struct BufferDef { const char* name; int binding; size_t bytes; };
    std::vector<BufferDef> buffers;
    int binding(const std::string& descriptor) const;     // image variable -> binding, -1 if absent
    const BufferDef* buffer(const std::string& descriptor) const;
    const ParamDef* param(const std::string& name) const;
};

const std::vector<NodeType>& registry();
// built-in types first, then {shader_path}/{name}.stage.hip (rf_user.h); *why = what is wrong with such a file, if it exists
// glsl_nodes: a {type}.comp file that could be a row stage of the stream kernel is wanted as a node with a kernel of its own (kPlanGlslNodes)
const NodeType* find_type(const std::string& name, std::string* why = nullptr, bool glsl_nodes = false);

// one uniform member's value (the UBO bytes of render.rs:167-210)
union ParamValue {
    float f;
    int32_t i;
    int32_t b;
};
// Rust str::parse::<f32|i32|bool>; failure or absence -> 0 (render.rs:173,:177,:181,:200-203)
ParamValue parse_param(const std::string* text, ParamType type);

// pipeline.rs:17-25 restricted to images
struct PipelineInfo {
    std::string name;                                      // node name, or "a+b+c" for a fused chain
    std::vector<std::string> members;                      // node names in execution order (1 unless fused)
    std::vector<int> member_slot;                          // per member: 0, or 1 / 2 = node of the branch feeding input_image0 / 1 of a fused fork/join (empty = all 0)
    std::vector<std::pair<std::string, int>> input_images; // (resource name, binding)
    std::vector<std::pair<std::string, int>> output_images;
    std::vector<std::pair<std::string, int>> input_ssbos;  // storage-buffer edges (pipeline.rs:23-24)
    std::vector<std::pair<std::string, int>> output_ssbos;
};

// per-node resolved parameters -> the device op
struct NodeParams {
    const NodeType* type = nullptr;
    std::map<std::string, ParamValue> values;   // every member of the type, zero-filled if absent
    Op to_op(const float* dev_weights) const;   // derives weights (gaussian/sharpen) on the host
    int conv_ksize() const;                     // conv2d: `ksize` clamped to odd [1,31]
};

struct Plan {
    std::map<std::string, NodeParams> nodes;               // every config node
    std::map<std::string, PipelineInfo> infos;             // planned units (nodes or fused chains)
    std::vector<std::vector<std::string>> layers;          // order_by_execution, name-sorted inside a layer
    std::map<std::string, std::string> reuse;              // image_reuse_remapping
    std::vector<std::string> images;                       // allocated per frame, sorted
    // storage buffers (pipeline_graph.rs:142-175, :240-265): size = max over users; an output on the binding of an input is
    // the same buffer (buffer_reuse); `buffers` = the names actually allocated, sorted
    std::map<std::string, std::string> buffer_reuse;
    std::map<std::string, size_t> buffer_bytes;
    std::vector<std::string> buffers;
    const std::string& resolve_buffer(const std::string& resource) const;
    bool fused = false;

    const std::string& resolve(const std::string& resource) const;   // remap_resource_name, :75-79
    // planned units in execution order (layer by layer)
    std::vector<std::string> launch_order() const;
};

// One kernel launch of a frame, in execution order.
struct LaunchDesc {
    std::string label;              // planned unit: node name or "a+b+c"
    std::vector<std::string> members;
    std::vector<int> member_slot;   // as PipelineInfo::member_slot
    int layer = 0;
    std::vector<std::string> src;   // allocated image names (1; 2 for OP_MIX in binding order)
    std::string dst;                // allocated image name (the first of `dsts`)
    std::vector<std::string> dsts;  // every allocated image the launch writes: 1, or up to 2 for a multi-output node (split_luma)
    std::vector<int> dst_bindings;  // the output binding each entry of `dsts` is wired to
    std::vector<std::string> in_buffers, out_buffers;   // allocated storage-buffer names the launch reads / writes
    std::vector<int> in_buffer_bindings, out_buffer_bindings;   // the binding each of them is wired to
    int radius = 0;                 // vertical halo read beyond the rows written
    int need_src = 0;               // ghost rows of src the launch reads (multi-rank)
    int need_dst = 0;               // ghost rows of dst the launch must also produce (over-fetch mode)
    bool result_only = false;       // no launch of the frame reads an image this one writes (the graph's result): rgba32f stream
                                    // kernels store it with the non-temporal hint (Geom::nt_store)
    bool serial = false;            // its layer holds a launch that writes an image another launch of the layer touches: the
                                    // layer runs in plan (name) order on one stream instead of concurrently
};

// device ops of a planned unit's members, in order, with their fork/join slots (weights: node name -> device pointer, may be null)
std::vector<Op> ops_of_members(const Plan& plan, const std::vector<std::string>& members, const std::vector<int>& member_slot,
                               const std::map<std::string, float*>* dev_weights);

// planned units -> launches; validates what the kernels can execute (one input image,
// two for `combination`; one output image; no in-place stencil)
bool build_launches(const Plan& plan, std::vector<LaunchDesc>& out, std::string& err);

// Ghost-row schedule of a row-strip partition (no reference counterpart).
//   exchange = true : before a launch of vertical radius r its input's r edge rows are
//                     exchanged with the neighbour ranks (need_src = r, need_dst = 0);
//   exchange = false: over-fetch -- the input carries the cumulative halo of everything
//                     downstream and each launch also produces the ghost rows its
//                     consumers read (need_dst), so no per-launch communication.
// need_input = ghost rows of rf:file-input the frame reads; ghost = rows to allocate.
void halo_schedule(std::vector<LaunchDesc>& launches, bool multi_rank, bool exchange, int& need_input, int& ghost);

constexpr uint32_t kPlanNoFusion = 0x2u;   // == RF_GRAPH_NO_FUSION
constexpr uint32_t kPlanNoJit = 0x10u;     // == RF_GRAPH_NO_JIT: fuse only what the ahead-of-time kernel catalogue holds
constexpr uint32_t kPlanGlslNodes = 0x20u; // == RF_GRAPH_GLSL_NODES: every {type}.comp file is a node with a kernel of its own, never a row stage

bool build_plan(const Config& cfg, uint32_t flags, Plan& out, std::string& err);

// rows [y0,y1) of `height` owned by `rank` of `world`: contiguous, sizes differ by <= 1
void strip_rows(int height, int world, int rank, int& y0, int& y1);

}  // namespace rf
