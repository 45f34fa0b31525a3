// rf_abi.cpp -- host-only half of the C ABI (include/rfhip.h): errors, the config DSL,
// the planner views and the node-type registry.  Nothing here touches a GPU.
#include <algorithm>
#include <cstring>

#include "rf_jit.h"
#include "rf_user.h"
#include "rf_glsl.h"
#include "rf_runtime.h"

namespace rf {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }

}  // namespace rf

using namespace rf;

static rf_status fail(rf_status st, const std::string& msg)
{
    set_error(msg);
    return st;
}

extern "C" const char* rf_last_error(void) { return last_error(); }
extern "C" int rf_abi_version(void) { return RF_ABI_VERSION; }

// ---------------------------------------------------------------------------------
// Config
// ---------------------------------------------------------------------------------
static void index_config(rf_config* c)
{
    c->node_names.clear();
    for (const auto& kv : c->cfg.graph_pipelines) c->node_names.push_back(kv.first);
}

extern "C" rf_status rf_config_parse(const char* text, int expects_input, rf_config** out)
{
    if (!text || !out) return fail(RF_ERR_INVALID, "rf_config_parse: null argument");
    *out = nullptr;
    rf_config* c = new rf_config();
    std::string err;
    if (!parse_config(text, expects_input != 0, c->cfg, err)) {
        delete c;
        return fail(RF_ERR_CONFIG, err);
    }
    index_config(c);
    *out = c;
    return RF_OK;
}

extern "C" rf_status rf_config_syntax(const char* text, char* buf, size_t cap, size_t* len)
{
    if (!text || !len || (!buf && cap)) return fail(RF_ERR_INVALID, "rf_config_syntax: null argument");
    std::string json, err;
    if (!parse_syntax(text, json, err)) return fail(RF_ERR_CONFIG, err);
    *len = json.size();
    if (json.size() + 1 > cap) return fail(RF_ERR_INVALID, "rf_config_syntax: the buffer is too small");
    std::memcpy(buf, json.c_str(), json.size() + 1);
    return RF_OK;
}

extern "C" rf_status rf_config_single(const char* type_name, int expects_input, rf_config** out)
{
    if (!type_name || !out) return fail(RF_ERR_INVALID, "rf_config_single: null argument");
    *out = nullptr;
    rf_config* c = new rf_config();
    std::string err;
    if (!single_node_config(type_name, expects_input != 0, c->cfg, err)) {
        delete c;
        return fail(RF_ERR_CONFIG, err);
    }
    index_config(c);
    *out = c;
    return RF_OK;
}

extern "C" void rf_config_destroy(rf_config* cfg) { delete cfg; }

extern "C" int rf_config_num_nodes(const rf_config* cfg) { return cfg ? (int)cfg->node_names.size() : 0; }

static const GraphPipeline* node_at(const rf_config* cfg, int node)
{
    if (!cfg || node < 0 || node >= (int)cfg->node_names.size()) return nullptr;
    return &cfg->cfg.graph_pipelines.at(cfg->node_names[(size_t)node]);
}

extern "C" const char* rf_config_node_name(const rf_config* cfg, int node)
{
    return node_at(cfg, node) ? cfg->node_names[(size_t)node].c_str() : nullptr;
}

extern "C" const char* rf_config_node_type(const rf_config* cfg, int node)
{
    return node_at(cfg, node) ? cfg->cfg.type_of(cfg->node_names[(size_t)node]).c_str() : nullptr;
}

extern "C" int rf_config_node_num_inputs(const rf_config* cfg, int node)
{
    const GraphPipeline* p = node_at(cfg, node);
    return p ? (int)p->inputs.size() : 0;
}

extern "C" int rf_config_node_num_outputs(const rf_config* cfg, int node)
{
    const GraphPipeline* p = node_at(cfg, node);
    return p ? (int)p->outputs.size() : 0;
}

static const ConfigDescriptor* desc_at(const rf_config* cfg, int node, int i, bool input)
{
    const GraphPipeline* p = node_at(cfg, node);
    if (!p) return nullptr;
    const auto& v = input ? p->inputs : p->outputs;
    return (i >= 0 && i < (int)v.size()) ? &v[(size_t)i] : nullptr;
}

extern "C" const char* rf_config_node_input_resource(const rf_config* cfg, int node, int i)
{
    const ConfigDescriptor* d = desc_at(cfg, node, i, true);
    return d ? d->resource_name.c_str() : nullptr;
}
extern "C" const char* rf_config_node_input_descriptor(const rf_config* cfg, int node, int i)
{
    const ConfigDescriptor* d = desc_at(cfg, node, i, true);
    return d ? d->descriptor_name.c_str() : nullptr;
}
extern "C" const char* rf_config_node_output_resource(const rf_config* cfg, int node, int i)
{
    const ConfigDescriptor* d = desc_at(cfg, node, i, false);
    return d ? d->resource_name.c_str() : nullptr;
}
extern "C" const char* rf_config_node_output_descriptor(const rf_config* cfg, int node, int i)
{
    const ConfigDescriptor* d = desc_at(cfg, node, i, false);
    return d ? d->descriptor_name.c_str() : nullptr;
}

extern "C" int rf_config_node_num_params(const rf_config* cfg, int node)
{
    return node_at(cfg, node) ? (int)cfg->cfg.params_of(cfg->node_names[(size_t)node]).size() : 0;
}

static const std::pair<const std::string, std::string>* param_at(const rf_config* cfg, int node, int i)
{
    if (!node_at(cfg, node)) return nullptr;
    const auto& m = cfg->cfg.params_of(cfg->node_names[(size_t)node]);
    if (i < 0 || i >= (int)m.size()) return nullptr;
    auto it = m.begin();
    std::advance(it, i);
    return &*it;
}

extern "C" const char* rf_config_node_param_key(const rf_config* cfg, int node, int i)
{
    auto* p = param_at(cfg, node, i);
    return p ? p->first.c_str() : nullptr;
}
extern "C" const char* rf_config_node_param_value(const rf_config* cfg, int node, int i)
{
    auto* p = param_at(cfg, node, i);
    return p ? p->second.c_str() : nullptr;
}

// ---------------------------------------------------------------------------------
// Plan
// ---------------------------------------------------------------------------------
extern "C" rf_status rf_plan_create(const rf_config* cfg, uint32_t flags, rf_plan** out)
{
    if (!cfg || !out) return fail(RF_ERR_INVALID, "rf_plan_create: null argument");
    *out = nullptr;
    rf_plan* p = new rf_plan();
    std::string err;
    if (!build_plan(cfg->cfg, flags, p->plan, err)) {
        delete p;
        return fail(RF_ERR_GRAPH, err);
    }
    p->index();
    *out = p;
    return RF_OK;
}

extern "C" void rf_plan_destroy(rf_plan* plan) { delete plan; }

extern "C" int rf_plan_num_layers(const rf_plan* p) { return p ? (int)p->plan.layers.size() : 0; }
extern "C" int rf_plan_layer_size(const rf_plan* p, int layer)
{
    return (p && layer >= 0 && layer < (int)p->plan.layers.size()) ? (int)p->plan.layers[(size_t)layer].size() : 0;
}
extern "C" const char* rf_plan_layer_node(const rf_plan* p, int layer, int i)
{
    if (!p || layer < 0 || layer >= (int)p->plan.layers.size()) return nullptr;
    const auto& l = p->plan.layers[(size_t)layer];
    return (i >= 0 && i < (int)l.size()) ? l[(size_t)i].c_str() : nullptr;
}
extern "C" int rf_plan_num_aliases(const rf_plan* p) { return p ? (int)p->aliases.size() : 0; }
extern "C" const char* rf_plan_alias_from(const rf_plan* p, int i)
{
    return (p && i >= 0 && i < (int)p->aliases.size()) ? p->aliases[(size_t)i].first.c_str() : nullptr;
}
extern "C" const char* rf_plan_alias_to(const rf_plan* p, int i)
{
    return (p && i >= 0 && i < (int)p->aliases.size()) ? p->aliases[(size_t)i].second.c_str() : nullptr;
}
extern "C" int rf_plan_num_images(const rf_plan* p) { return p ? (int)p->plan.images.size() : 0; }
extern "C" const char* rf_plan_image_name(const rf_plan* p, int i)
{
    return (p && i >= 0 && i < (int)p->plan.images.size()) ? p->plan.images[(size_t)i].c_str() : nullptr;
}
extern "C" const char* rf_plan_resolve(const rf_plan* p, const char* resource)
{
    if (!p || !resource) return nullptr;
    // the returned pointer must outlive the call: it is a key/value of the plan's own
    // maps, or the caller's string when nothing remaps it
    std::string r = resource;
    auto it = p->plan.reuse.find(r);
    if (it == p->plan.reuse.end()) return resource;
    return p->plan.resolve(it->second).c_str();
}
void rf_plan::index()
{
    aliases.assign(plan.reuse.begin(), plan.reuse.end());
    launch_error.clear();
    if (!build_launches(plan, launches, launch_error)) launches.clear();
}

static const LaunchDesc* launch_at(const rf_plan* p, int i)
{
    return (p && i >= 0 && i < (int)p->launches.size()) ? &p->launches[(size_t)i] : nullptr;
}

extern "C" int rf_plan_num_launches(const rf_plan* p) { return p ? (int)p->launches.size() : 0; }
extern "C" const char* rf_plan_launch_label(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? l->label.c_str() : nullptr;
}
extern "C" int rf_plan_launch_layer(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? l->layer : -1;
}
extern "C" int rf_plan_launch_num_members(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? (int)l->members.size() : 0;
}
extern "C" const char* rf_plan_launch_member(const rf_plan* p, int i, int k)
{
    const LaunchDesc* l = launch_at(p, i);
    return (l && k >= 0 && k < (int)l->members.size()) ? l->members[(size_t)k].c_str() : nullptr;
}
extern "C" int rf_plan_launch_num_inputs(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? (int)l->src.size() : 0;
}
extern "C" const char* rf_plan_launch_input(const rf_plan* p, int i, int k)
{
    const LaunchDesc* l = launch_at(p, i);
    return (l && k >= 0 && k < (int)l->src.size()) ? l->src[(size_t)k].c_str() : nullptr;
}
extern "C" const char* rf_plan_launch_output(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? l->dst.c_str() : nullptr;
}
extern "C" int rf_plan_launch_num_outputs(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? (int)l->dsts.size() : -1;
}
extern "C" const char* rf_plan_launch_output_at(const rf_plan* p, int i, int k)
{
    const LaunchDesc* l = launch_at(p, i);
    return l && k >= 0 && k < (int)l->dsts.size() ? l->dsts[(size_t)k].c_str() : nullptr;
}
extern "C" int rf_plan_launch_radius(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? l->radius : -1;
}
extern "C" int rf_plan_launch_serial(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    return l ? (l->serial ? 1 : 0) : -1;
}
extern "C" uint64_t rf_plan_signature(const rf_plan* p)
{
    if (!p) return 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const std::string& t) {
        for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; }
        h ^= 0xff; h *= 1099511628211ull;
    };
    mix(p->launch_error);
    for (const auto& l : p->launches) {
        mix(l.label);
        mix(std::to_string(l.radius) + "/" + std::to_string(l.layer));
        for (const auto& d : l.dsts) mix(d);
    }
    return h ? h : 1;
}
extern "C" rf_status rf_plan_halo_schedule(const rf_plan* p, int exchange, int* need_src, int* need_dst, int n,
                                           int* need_input, int* ghost)
{
    if (!p || !need_src || !need_dst || !need_input || !ghost) return fail(RF_ERR_INVALID, "rf_plan_halo_schedule: null argument");
    if (!p->launch_error.empty()) return fail(RF_ERR_GRAPH, p->launch_error);
    if (n < (int)p->launches.size()) return fail(RF_ERR_INVALID, "rf_plan_halo_schedule: arrays shorter than rf_plan_num_launches");
    std::vector<LaunchDesc> l = p->launches;
    halo_schedule(l, true, exchange != 0, *need_input, *ghost);
    for (size_t k = 0; k < l.size(); ++k) {
        need_src[k] = l[k].need_src;
        need_dst[k] = l[k].need_dst;
    }
    return RF_OK;
}

// ---- storage buffers -------------------------------------------------------------------------------
extern "C" int rf_plan_num_buffers(const rf_plan* p) { return p ? (int)p->plan.buffers.size() : -1; }
extern "C" const char* rf_plan_buffer_name(const rf_plan* p, int i)
{
    return (p && i >= 0 && i < (int)p->plan.buffers.size()) ? p->plan.buffers[(size_t)i].c_str() : nullptr;
}
extern "C" size_t rf_plan_buffer_bytes(const rf_plan* p, int i)
{
    return (p && i >= 0 && i < (int)p->plan.buffers.size()) ? p->plan.buffer_bytes.at(p->plan.buffers[(size_t)i]) : 0;
}
extern "C" const char* rf_plan_resolve_buffer(const rf_plan* p, const char* resource)
{
    return (p && resource) ? p->plan.resolve_buffer(resource).c_str() : nullptr;
}
extern "C" int rf_registry_buffer_binding(const char* type_name, const char* block_type_name)
{
    if (!type_name || !block_type_name) return -1;
    const NodeType* t = find_type(type_name);
    const NodeType::BufferDef* b = t ? t->buffer(block_type_name) : nullptr;
    return b ? b->binding : -1;
}

// ---- filter types that are files (rf_user.h) -----------------------------------------------------
extern "C" rf_status rf_set_shader_path(const char* dir)
{
    set_shader_path(dir ? dir : "");
    return RF_OK;
}
extern "C" const char* rf_shader_path(void)
{
    static thread_local std::string s;
    s = shader_path();
    return s.c_str();
}
extern "C" long long rf_user_stage_mtime(const char* type_name)
{
    if (!type_name) return -1;
    std::string err;
    const UserStage* u = user_stage_for_type(type_name, err);
    return u ? u->mtime_ns : -1;
}

extern "C" rf_status rf_set_type_lookup(int files_first_) { set_files_first(files_first_ != 0); return RF_OK; }
extern "C" int rf_type_lookup(void) { return files_first() ? 1 : 0; }

// ---- {type}.comp: GLSL compute shaders, translated (rf_glsl.cpp) ------------------------------------
static rf_status glsl_text_out(const char* what, const std::string& text, char* buf, size_t cap, size_t* len)
{
    *len = text.size();
    if (text.size() + 1 > cap) return fail(RF_ERR_INVALID, std::string(what) + ": the buffer is too small");
    std::memcpy(buf, text.c_str(), text.size() + 1);
    return RF_OK;
}
extern "C" rf_status rf_glsl_translate(const char* type_name, const char* text, char* buf, size_t cap, size_t* len)
{
    if (!type_name || !text || !len || (!buf && cap)) return fail(RF_ERR_INVALID, "rf_glsl_translate: null argument");
    UserStage st;
    std::string err;
    if (!parse_glsl_stage(type_name, text, st, err)) return fail(RF_ERR_GRAPH, err);
    return glsl_text_out("rf_glsl_translate", "// " + st.ident + "\n" + st.glsl_source, buf, cap, len);
}
extern "C" rf_status rf_glsl_reflect(const char* type_name, const char* text, char* buf, size_t cap, size_t* len)
{
    if (!type_name || !text || !len || (!buf && cap)) return fail(RF_ERR_INVALID, "rf_glsl_reflect: null argument");
    GlslShader sh;
    std::string err;
    if (!glsl_translate(type_name, text, "reflect", sh, err)) return fail(RF_ERR_GRAPH, err);
    return glsl_text_out("rf_glsl_reflect", glsl_reflection_json(sh), buf, cap, len);
}

// ---- kernels compiled at graph creation (rf_jit.cpp) --------------------------------------------
extern "C" int rf_jit_available(void) { return jit_available() ? 1 : 0; }
extern "C" int rf_jit_compile_count(void) { return jit_compile_count(); }
extern "C" const char* rf_jit_library(void)
{
    static thread_local std::string s;
    s = jit_library();
    return s.c_str();
}

extern "C" int rf_plan_launch_member_slot(const rf_plan* p, int i, int k)
{
    const LaunchDesc* l = launch_at(p, i);
    if (!l || k < 0 || k >= (int)l->members.size()) return -1;
    return k < (int)l->member_slot.size() ? l->member_slot[(size_t)k] : 0;
}

extern "C" int rf_plan_launch_needs_jit(const rf_plan* p, int i)
{
    const LaunchDesc* l = launch_at(p, i);
    if (!l) return -1;
    std::vector<Op> ops = ops_of_members(p->plan, l->members, l->member_slot, nullptr);
    StageList sl;
    if (ops.size() == 1 && ops[0].kind == OP_USERN) return 1;      // a user node: always compiled at graph creation
    if ((ops.size() < 2 && !(ops.size() == 1 && ops[0].kind == OP_USER)) || !ops_to_stages(ops.data(), (int)ops.size(), sl)) return 0;
    return stream_in_catalogue(sl) ? 0 : 1;
}

extern "C" rf_status rf_plan_jit_compile(const rf_plan* p, int format, size_t* code_bytes) { return rf_plan_jit_compile_texels(p, format, 1, code_bytes); }

extern "C" rf_status rf_plan_jit_compile_texels(const rf_plan* p, int format, int texels_per_lane, size_t* code_bytes)
{
    if (!p) return fail(RF_ERR_INVALID, "rf_plan_jit_compile: null plan");
    if (texels_per_lane != 1 && texels_per_lane != 2) return fail(RF_ERR_INVALID, "rf_plan_jit_compile: texels_per_lane must be 1 or 2");
    if (!p->launch_error.empty()) return fail(RF_ERR_GRAPH, p->launch_error);
    if (format != RF_FORMAT_RGBA8 && format != RF_FORMAT_RGBA32F) return fail(RF_ERR_INVALID, "rf_plan_jit_compile: unknown format");
    size_t total = 0;
    for (const auto& l : p->launches) {
        std::vector<Op> ops = ops_of_members(p->plan, l.members, l.member_slot, nullptr);
        StageList sl;
        std::string err;
        if (ops.size() == 1 && ops[0].kind == OP_USERN) {
            if (texels_per_lane != 1) continue;
            const size_t n = jit_compile_only_user_node(format, ops[0].user_id, err);
            if (n == 0) return fail(RF_ERR_UNSUPPORTED, err);
            total += n;
            continue;
        }
        if ((ops.size() < 2 && !(ops.size() == 1 && ops[0].kind == OP_USER)) || !ops_to_stages(ops.data(), (int)ops.size(), sl) || stream_in_catalogue(sl)) continue;
        if (texels_per_lane == 2 && (sl.sum_rh() > 7 || sl.max_rv() > 4 || sl.pair())) continue;      // no two-texel variant of such a list (choose_texels)
        // the variant rf_graph_create would build: rgba32f with non-temporal stores where no launch reads the result (stream_prepare)
        const int kc = stream_kernel_code(format, sl, l.result_only);
        const size_t n = jit_compile_only(kc, 4, texels_per_lane, sl, 4, err);
        if (n == 0) return fail(RF_ERR_UNSUPPORTED, err);
        total += n;
    }
    if (code_bytes) *code_bytes = total;
    return RF_OK;
}

// ---------------------------------------------------------------------------------
// Registry
// ---------------------------------------------------------------------------------
extern "C" int rf_registry_num_types(void) { return (int)registry().size(); }
extern "C" const char* rf_registry_type_name(int t)
{
    return (t >= 0 && t < (int)registry().size()) ? registry()[(size_t)t].name : nullptr;
}
extern "C" int rf_registry_binding(const char* type_name, const char* descriptor)
{
    if (!type_name || !descriptor) return -1;
    const NodeType* t = find_type(type_name);
    return t ? t->binding(descriptor) : -1;
}
extern "C" int rf_registry_radius(const char* type_name)
{
    if (!type_name) return -1;
    const NodeType* t = find_type(type_name);
    if (!t) return -1;
    return t->fixed_radius >= 0 ? t->fixed_radius : kMaxRadius;
}

extern "C" rf_status rf_strip_rows(int height, int world, int rank, int* y0, int* y1)
{
    if (!y0 || !y1 || height < 1 || world < 1 || rank < 0 || rank >= world) return fail(RF_ERR_INVALID, "rf_strip_rows: bad argument");
    strip_rows(height, world, rank, *y0, *y1);
    return RF_OK;
}
