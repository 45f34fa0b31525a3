// rf_plan.cpp -- see rf_plan.h.  Host only; no GPU involved.
#include "rf_plan.h"
#include "rf_user.h"

#include <algorithm>
#include <functional>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>

namespace rf {

// ---------------------------------------------------------------------------------
// Registry
// ---------------------------------------------------------------------------------
int NodeType::binding(const std::string& descriptor) const
{
    for (const auto& im : images)
        if (descriptor == im.first) return im.second;
    return -1;
}

const NodeType::BufferDef* NodeType::buffer(const std::string& descriptor) const
{
    for (const auto& b : buffers)
        if (descriptor == b.name) return &b;
    return nullptr;
}

const ParamDef* NodeType::param(const std::string& pname) const
{
    for (const auto& p : params)
        if (pname == p.name) return &p;
    return nullptr;
}

const std::vector<NodeType>& registry()
{
    // passthrough.comp:4-5 declares input_image at binding 0 and output_image at 1; the
    // authored types keep that layout.  colour_grade adds a read-write `image` so a
    // config can run it in place (`-> colour_grade:image ->`, pipeline_graph.rs:400-411).
    static const std::vector<std::pair<const char*, int>> io = {{"input_image", 0}, {"output_image", 1}};
    static const std::vector<std::pair<const char*, int>> io_rw = {{"input_image", 0}, {"output_image", 1}, {"image", 2}};
    static const std::vector<std::pair<const char*, int>> io2 = {{"input_image0", 0}, {"input_image1", 1}, {"output_image", 2}};
    static const std::vector<std::pair<const char*, int>> rw = {{"image", 0}};   // shaders/colour_grade_inplace.comp: one read-write image
    static const std::vector<ParamDef> grade = {{"slope", PARAM_F32}, {"offset", PARAM_F32}, {"saturation", PARAM_F32}};
    // gaussian types: sigma (+ radius) and the OPTIONAL explicit weights w0 .. wR of shaders/gaussian*.comp: all zero
    // (not given) => derived from sigma on the host; given => used as they are (scripts/glsl_weights.py prints them)
    static const char* wn[kMaxRadius + 1] = {"w0", "w1", "w2", "w3", "w4", "w5", "w6", "w7", "w8", "w9", "w10", "w11", "w12", "w13", "w14", "w15"};
    auto gauss_params = [&](int r, bool with_radius) {
        std::vector<ParamDef> v = {{"sigma", PARAM_F32}};
        if (with_radius) v.push_back({"radius", PARAM_I32});
        for (int i = 0; i <= r; ++i) v.push_back({wn[i], PARAM_F32});
        return v;
    };
    static const std::vector<NodeType> types = {
        {"passthrough", OP_PASSTHROUGH, 0, io, {}, {}},
        {"gaussian5", OP_GAUSSIAN, 2, io, gauss_params(2, false), {}},
        {"gaussian9", OP_GAUSSIAN, 4, io, gauss_params(4, false), {}},
        {"gaussian", OP_GAUSSIAN, -1, io, gauss_params(kMaxRadius, true), {}},
        {"colour_grade", OP_GRADE, 0, io_rw, grade, {}},
        {"colour_grade_inplace", OP_GRADE, 0, rw, grade, {}},
        {"colour-grade", OP_GRADE, 0, io_rw, grade, {}},
        {"grade", OP_GRADE, 0, io_rw, grade, {}},
        {"sharpen", OP_SHARPEN, 1, io, {{"amount", PARAM_F32}}, {}},
        // conv2d may take its K x K weights through a storage-buffer edge (`kw:ConvWeights -> conv:ConvWeights`); the block
        // `ConvWeights { float weights[961]; }` is found by its type name (shaders/conv2d.comp, shader.rs:144-147)
        {"conv2d", OP_CONV2D, -1, io, {{"ksize", PARAM_I32}, {"sigma", PARAM_F32}}, {{"ConvWeights", 3, 961 * sizeof(float)}}},
        {"conv2d_weights", OP_WEIGHTS, 0, io, {{"ksize", PARAM_I32}, {"sigma", PARAM_F32}}, {{"ConvWeights", 3, 961 * sizeof(float)}}},
        {"combination", OP_MIX, 0, io2, {{"mix", PARAM_F32}}, {}},
        // a node with TWO output images: the reference allocates one image per output binding (pipeline_graph.rs:205-224)
        {"split_luma", OP_SPLIT, 0, {{"input_image", 0}, {"luma_image", 1}, {"chroma_image", 2}}, {}, {}},
        // a member ending in `_rf_time` receives the seconds since start every frame (render.rs:190,:212-223)
        {"pulse", OP_PULSE, 0, io, {{"amount", PARAM_F32}, {"phase_rf_time", PARAM_F32}}, {}},
    };
    return types;
}

const NodeType* find_type(const std::string& name, std::string* why, bool glsl_nodes)
{
    std::string err;
    if (files_first()) {      // the reference's rule: the file IS the type (config.rs:59-75) -- a file that does not translate is an error, not a fallback
        const UserStage* u = user_stage_for_type(name, err);
        if (u) return user_stage_node_type(u, glsl_nodes);
        if (!err.empty()) { if (why) *why = err; return nullptr; }
    }
    for (const auto& t : registry())
        if (name == t.name) return &t;
    // not built in: a type that is a file, {shader_path}/{name}.stage.hip or {name}.comp (config.rs:59-75; rf_user.h, rf_glsl.h)
    const UserStage* u = user_stage_for_type(name, err);
    if (why) *why = err;
    return u ? user_stage_node_type(u, glsl_nodes) : nullptr;
}

// ---------------------------------------------------------------------------------
// Parameters (render.rs:167-210)
// ---------------------------------------------------------------------------------
static bool rust_parse_i32(const std::string& s, int32_t& out)
{
    // i32::from_str: optional sign, then decimal digits only; overflow is an error
    size_t i = 0;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) ++i;
    if (i >= s.size()) return false;
    for (size_t j = i; j < s.size(); ++j)
        if (s[j] < '0' || s[j] > '9') return false;
    errno = 0;
    long long v = std::strtoll(s.c_str(), nullptr, 10);
    if (errno != 0 || v < INT32_MIN || v > INT32_MAX) return false;
    out = (int32_t)v;
    return true;
}

static bool rust_parse_f32(const std::string& s, float& out)
{
    // the grammar only lets [0-9]+ , -?[0-9]+\.[0-9]+ , true, false through
    // (config_grammar.lalrpop:74-78); "true"/"false" fail f32::from_str
    if (s.empty()) return false;
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-') ++i;
    bool digits = false, dot = false;
    for (; i < s.size(); ++i) {
        if (s[i] >= '0' && s[i] <= '9') digits = true;
        else if (s[i] == '.' && !dot) dot = true;
        else return false;
    }
    if (!digits) return false;
    out = std::strtof(s.c_str(), nullptr);   // correctly rounded decimal -> f32, like Rust
    return true;
}

ParamValue parse_param(const std::string* text, ParamType type)
{
    ParamValue v;
    v.i = 0;
    if (!text) return v;   // absent: zero-filled (render.rs:200-203)
    switch (type) {
        case PARAM_F32: {
            float f;
            if (rust_parse_f32(*text, f)) v.f = f;
            break;
        }
        case PARAM_I32: {
            int32_t i;
            if (rust_parse_i32(*text, i)) v.i = i;
            break;
        }
        case PARAM_BOOL:
            v.b = (*text == "true") ? 1 : 0;
            break;
    }
    return v;
}

static float pf(const std::map<std::string, ParamValue>& m, const char* k)
{
    auto it = m.find(k);
    return it == m.end() ? 0.0f : it->second.f;
}
static int pi(const std::map<std::string, ParamValue>& m, const char* k)
{
    auto it = m.find(k);
    return it == m.end() ? 0 : it->second.i;
}

int NodeParams::conv_ksize() const
{
    int k = pi(values, "ksize");
    if (k < 1) k = 1;
    if (k > 2 * kMaxRadius + 1) k = 2 * kMaxRadius + 1;
    if (k % 2 == 0) k -= 1;
    return k;
}

// slope of the `pulse` node: fma(amount, frac(t), 1) -- the same single rounding as oracle/rf_oracle.c rfo_pulse_slope
static float pulse_slope(float amount, float t)
{
    float f = t - std::floor(t);
    if (!(f >= 0.0f && f < 1.0f)) f = 0.0f;
    return std::fmaf(amount, f, 1.0f);
}

Op NodeParams::to_op(const float* dev_weights) const
{
    Op op;
    op.kind = type->kind;
    switch (type->kind) {
        case OP_WEIGHTS:              // its device work is the copy of its image; the buffer is written by the host side (rf_graph.cpp)
            op.kind = OP_PASSTHROUGH;
            break;
        case OP_PULSE:
            op.kind = OP_GRADE;
            op.slope = pulse_slope(pf(values, "amount"), pf(values, "phase_rf_time"));
            op.offset = 0.0f;
            op.saturation = 1.0f;
            break;
        case OP_GAUSSIAN: {
            int r = type->fixed_radius;
            if (r < 0) r = std::min(std::max(pi(values, "radius"), 0), kMaxRadius);
            op.radius = r;
            gaussian_weights(pf(values, "sigma"), r, op.w);
            {   // explicit weights (shaders/gaussian*.comp: `w0 .. wR`): any non-zero member replaces the derived kernel
                static const char* wn[kMaxRadius + 1] = {"w0", "w1", "w2", "w3", "w4", "w5", "w6", "w7", "w8", "w9", "w10", "w11", "w12", "w13", "w14", "w15"};
                bool given = false;
                for (int i = 0; i <= r; ++i) given = given || pf(values, wn[i]) != 0.0f;
                if (given)
                    for (int i = 0; i <= r; ++i) op.w[i] = pf(values, wn[i]);
            }
            break;
        }
        case OP_GRADE:
            op.slope = pf(values, "slope");
            op.offset = pf(values, "offset");
            op.saturation = pf(values, "saturation");
            break;
        case OP_SHARPEN:
            op.radius = 1;
            sharpen_weights(pf(values, "amount"), &op.wc, &op.ws);
            break;
        case OP_CONV2D:
            op.radius = conv_ksize() / 2;
            op.dev_weights = dev_weights;
            break;
        case OP_MIX:
            op.slope = pf(values, "mix");
            break;
        case OP_USER:
        case OP_USERN:
            if (const UserStage* u = user_stage_of(type)) {
                op.user_id = u->id;
                op.radius = u->radius;
                for (const auto& p : u->params) {
                    auto it = values.find(p.name);
                    ParamValue v;
                    v.i = 0;
                    if (it != values.end()) v = it->second;
                    if (p.type == PARAM_BOOL) op.user_params[p.offset] = v.b ? 1 : 0;
                    else std::memcpy(op.user_params + p.offset, &v, 4);
                }
            }
            break;
        default:
            break;
    }
    return op;
}

// ---------------------------------------------------------------------------------
// Plan
// ---------------------------------------------------------------------------------
const std::string& Plan::resolve_buffer(const std::string& resource) const
{
    const std::string* name = &resource;
    for (;;) {
        auto it = buffer_reuse.find(*name);
        if (it == buffer_reuse.end()) return *name;
        name = &it->second;
    }
}

const std::string& Plan::resolve(const std::string& resource) const
{
    const std::string* name = &resource;
    for (;;) {
        auto it = reuse.find(*name);
        if (it == reuse.end()) return *name;
        name = &it->second;
    }
}

std::vector<std::string> Plan::launch_order() const
{
    std::vector<std::string> out;
    for (const auto& layer : layers)
        for (const auto& n : layer) out.push_back(n);
    return out;
}

// order_by_execution, pipeline_graph.rs:429-497
static bool order_by_execution(const std::map<std::string, PipelineInfo>& infos,
                               std::vector<std::vector<std::string>>& layers, std::string& err)
{
    std::set<std::string> unexecuted;
    for (const auto& kv : infos) unexecuted.insert(kv.first);

    auto input_nodes = [&](const PipelineInfo& info) {
        std::vector<std::string> nodes;
        for (const auto& cand : infos) {
            bool feeds = false;
            for (const auto& out : cand.second.output_images)
                for (const auto& in : info.input_images)
                    if (out.first == in.first) feeds = true;
            for (const auto& out : cand.second.output_ssbos)     // storage-buffer edges order nodes too (:438,:443)
                for (const auto& in : info.input_ssbos)
                    if (out.first == in.first) feeds = true;
            if (feeds) nodes.push_back(cand.first);
        }
        return nodes;
    };

    while (!unexecuted.empty()) {
        const std::set<std::string> snapshot = unexecuted;   // readiness is judged against the snapshot (:470,:479)
        std::vector<std::string> layer;
        for (const auto& node : snapshot) {
            bool ready = true;
            for (const auto& dep : input_nodes(infos.at(node)))
                if (snapshot.count(dep)) ready = false;
            if (ready) {
                unexecuted.erase(node);
                layer.push_back(node);
            }
        }
        if (snapshot.size() == unexecuted.size()) {   // :487-490
            err = "Graph incorrectly constructed. Failed to add nodes into execution: [";
            bool first = true;
            for (const auto& n : snapshot) {
                err += (first ? "\"" : ", \"") + n + "\"";
                first = false;
            }
            err += "]";
            return false;
        }
        layers.push_back(layer);
    }
    return true;
}

// reusable_image_remapping, pipeline_graph.rs:358-427
static std::map<std::string, std::string> reusable_image_remapping(
    const std::vector<std::vector<std::string>>& layers, const std::map<std::string, PipelineInfo>& infos)
{
    std::vector<std::string> free_images;
    std::set<std::string> images;
    std::map<std::string, std::string> reuse;

    // "Found a remap of the allocation in use" (:364-369).  The reference looks ONE level up
    // (image_reuse.get(image_name) == name); this follows the alias chain to the allocation.
    // With one level, an image read through a two-level alias (a point op written in place on
    // an image that is itself a recycled allocation) counts as free while a later node still
    // reads it, and that node's output lands on the image it reads: a stencil in place,
    // undefined output in the reference.  Following the chain only keeps an image allocated
    // longer; every plan the reference gets right is unchanged (tests/test_config_plan.py).
    auto root_of = [&](const std::string& n) {
        const std::string* cur = &n;
        for (auto it = reuse.find(*cur); it != reuse.end(); it = reuse.find(*cur)) cur = &it->second;
        return *cur;
    };
    auto has_remap = [&](const std::string& name, const std::vector<std::pair<std::string, int>>& imgs) {
        for (const auto& im : imgs)
            if (reuse.count(im.first) && root_of(im.first) == name) return true;
        return false;
    };
    auto node_uses = [&](const PipelineInfo& node, const std::string& name) {
        for (const auto& im : node.input_images)
            if (im.first == name) return true;
        for (const auto& im : node.output_images)
            if (im.first == name) return true;
        return has_remap(name, node.input_images) || has_remap(name, node.output_images);
    };
    auto still_in_use = [&](const std::string& name, size_t start_layer) {
        for (size_t l = start_layer; l < layers.size(); ++l)
            for (const auto& n : layers[l])
                if (node_uses(infos.at(n), name)) return true;
        return false;
    };

    for (size_t li = 0; li < layers.size(); ++li) {
        for (const auto& name : images) {
            if (std::find(free_images.begin(), free_images.end(), name) != free_images.end()) continue;
            if (!still_in_use(name, li)) free_images.push_back(name);
        }
        for (const auto& n : layers[li]) {
            const PipelineInfo& node = infos.at(n);
            for (const auto& out : node.output_images) {
                bool point_op = false;   // same binding as an input: run in place (:400-411)
                for (const auto& in : node.input_images) {
                    if (out.second == in.second) {
                        point_op = true;
                        reuse[out.first] = in.first;
                    }
                }
                if (point_op) continue;
                if (free_images.empty()) {
                    images.insert(out.first);
                } else {
                    reuse[out.first] = free_images.back();
                    free_images.pop_back();
                }
            }
        }
    }
    return reuse;
}

static bool is_simple(const PipelineInfo& p)
{
    return p.input_images.size() == 1 && p.output_images.size() == 1 && p.input_ssbos.empty() && p.output_ssbos.empty();   // a node with a buffer edge keeps a launch of its own
}

// Fuse chains of single-input/single-output nodes whose intermediate image has exactly
// one producer and one consumer and is not the graph output (no reference counterpart:
// the reference materialises every edge, pipeline_graph.rs:219-221).  A chain fuses whole when its
// kernel is in the ahead-of-time catalogue or -- allow_jit -- can be compiled when the graph is
// created (rf_jit.cpp: any list of passthrough / gaussian / colour_grade / sharpen nodes whose state
// fits the register file); otherwise it is cut greedily into the longest pieces that do.
static void fuse_chains(const Plan& plan, std::map<std::string, PipelineInfo>& infos, bool allow_jit)
{
    std::map<std::string, std::vector<std::string>> producers, consumers;
    for (const auto& kv : infos) {
        for (const auto& o : kv.second.output_images) producers[o.first].push_back(kv.first);
        for (const auto& i : kv.second.input_images) consumers[i.first].push_back(kv.first);
    }
    auto next_of = [&](const std::string& n) -> std::string {
        const PipelineInfo& p = infos.at(n);
        if (!is_simple(p) || p.members.size() != 1) return "";          // (a fused fork/join unit stays a launch of its own)
        const std::string& r = p.output_images[0].first;
        if (r == kFinalOutput || r == kFileInput) return "";
        if (producers[r].size() != 1 || consumers[r].size() != 1) return "";
        const std::string& c = consumers[r][0];
        if (c == n || !is_simple(infos.at(c)) || infos.at(c).members.size() != 1) return "";
        return c;
    };
    std::map<std::string, std::string> prev_of;
    for (const auto& kv : infos) {
        std::string nx = next_of(kv.first);
        if (!nx.empty()) prev_of[nx] = kv.first;
    }
    // A point op written in place (same binding for input and output, :400-411) MODIFIES the image
    // it reads -- and, through a run of in-place nodes, the allocation at the root of that run.  A
    // fused launch writes only its last member's output, so an in-place member may be fused only if
    // nobody outside the group can see its write: every node that reads a resource living on that
    // root allocation is in the group, and the root is not rf:file-input (which persists from frame
    // to frame -- the reference grades its input again every frame in that case, and so does this
    // path).  Otherwise the node keeps a launch of its own.
    auto in_place = [&](const PipelineInfo& p) { return is_simple(p) && p.input_images[0].second == p.output_images[0].second; };
    std::map<std::string, std::string> root_memo;
    std::function<std::string(const std::string&)> root_of = [&](const std::string& res) -> std::string {
        auto it = root_memo.find(res);
        if (it != root_memo.end()) return it->second;
        root_memo[res] = res;                                      // cycle guard
        std::string r = res;
        auto pr = producers.find(res);
        if (pr != producers.end() && pr->second.size() == 1 && in_place(infos.at(pr->second[0])))
            r = root_of(infos.at(pr->second[0]).input_images[0].first);
        return root_memo[res] = r;
    };
    std::map<std::string, std::set<std::string>> readers_of_root;
    for (const auto& kv : infos)
        for (const auto& in : kv.second.input_images) readers_of_root[root_of(in.first)].insert(kv.first);
    auto side_effects_stay_inside = [&](const std::vector<std::string>& group) {
        for (const auto& m : group) {
            const PipelineInfo& p = infos.at(m);
            if (!in_place(p)) continue;
            const std::string root = root_of(p.input_images[0].first);
            if (root == kFileInput) return false;
            for (const auto& reader : readers_of_root[root])
                if (std::find(group.begin(), group.end(), reader) == group.end()) return false;
        }
        return true;
    };
    // ---- fork/join: a `combination` node whose two inputs descend from ONE image through chains of simple nodes is ONE
    // launch (rf_stream_dev.h "Fork / join in ONE launch"): branch results are never stored, the forked image is read once.
    // Conservative: no in-place node in a branch, every branch image has one producer and one consumer.
    struct Diamond { std::vector<std::string> pre, a, b, post; std::string mix, src; };
    std::vector<Diamond> diamonds;
    std::set<std::string> in_diamond;
    for (const auto& kv : infos) {
        const PipelineInfo& m = kv.second;
        if (plan.nodes.at(kv.first).type->kind != OP_MIX || m.input_images.size() != 2 || m.output_images.size() != 1) continue;
        std::string in0, in1;
        for (const auto& in : m.input_images) (in.second == 0 ? in0 : in1) = in.first;
        if (in0.empty() || in1.empty() || in0 == in1) continue;
        auto trace = [&](std::string res, std::vector<std::string>& nodes, std::vector<std::string>& sources) {
            // walk upstream while the image has ONE simple, not-in-place producer and this walk is its only consumer
            sources.push_back(res);
            for (;;) {
                if (res == kFileInput || producers[res].size() != 1 || consumers[res].size() != 1) return;
                const std::string& n = producers[res][0];
                const PipelineInfo& pi = infos.at(n);
                const int kind = plan.nodes.at(n).type->kind;
                if (!is_simple(pi) || in_place(pi) || own_kernel_kind(kind)) return;
                nodes.insert(nodes.begin(), n);
                res = pi.input_images[0].first;
                sources.push_back(res);
            }
        };
        std::vector<std::string> na, nb, sa, sb;
        trace(in0, na, sa);
        trace(in1, nb, sb);
        // the nearest common image: branch a = nodes below it on a's walk, same for b
        std::string common;
        size_t ia = 0, ib = 0;
        for (ia = 0; ia < sa.size() && common.empty(); ++ia)
            for (ib = 0; ib < sb.size(); ++ib)
                if (sa[ia] == sb[ib]) { common = sa[ia]; break; }
        if (common.empty()) continue;
        --ia;                                         // sa[ia] == sb[ib] == common; nodes below: the last ia (ib) of na (nb)
        Diamond d;
        d.a.assign(na.end() - (ptrdiff_t)ia, na.end());
        d.b.assign(nb.end() - (ptrdiff_t)ib, nb.end());
        if (d.a.empty() && d.b.empty()) continue;
        d.mix = kv.first;
        d.src = common;
        // the common image is read by BOTH walks: consumers[common] holds two entries for them; anything inside a branch
        // must have been single-consumer (trace), which the common image itself need not be
        bool clash = false;
        for (const auto& n : d.a) clash = clash || in_diamond.count(n);
        for (const auto& n : d.b) clash = clash || in_diamond.count(n);
        if (clash || in_diamond.count(d.mix)) continue;
        // nodes before the fork and after the join ride along (StSolo stages) when nobody else sees their images: upstream
        // while the forked image has ONE simple producer and no reader outside the group; downstream while the joined image has
        // ONE simple consumer and is not the graph output.  The longest admissible extension wins; the bare diamond is the floor.
        std::vector<std::string> pre, post;
        {
            std::string res = common;
            std::set<std::string> inside(d.a.begin(), d.a.end());
            inside.insert(d.b.begin(), d.b.end());
            inside.insert(d.mix);
            for (;;) {
                if (res == kFileInput || producers[res].size() != 1) break;
                bool all_inside = true;
                for (const auto& c : consumers[res]) all_inside = all_inside && inside.count(c);
                const std::string& n = producers[res][0];
                const PipelineInfo& pi = infos.at(n);
                const int kind = plan.nodes.at(n).type->kind;
                if (!all_inside || !is_simple(pi) || in_place(pi) || own_kernel_kind(kind) || in_diamond.count(n)) break;
                pre.insert(pre.begin(), n);
                inside.insert(n);
                res = pi.input_images[0].first;
            }
            res = m.output_images[0].first;
            for (;;) {
                if (res == kFinalOutput || producers[res].size() != 1 || consumers[res].size() != 1) break;
                const std::string& n = consumers[res][0];
                const PipelineInfo& pi = infos.at(n);
                const int kind = plan.nodes.at(n).type->kind;
                if (!is_simple(pi) || in_place(pi) || own_kernel_kind(kind) || in_diamond.count(n) || inside.count(n)) break;
                post.push_back(n);
                inside.insert(n);
                res = pi.output_images[0].first;
            }
        }
        std::vector<std::string> members;
        std::vector<int> slots;
        bool ok = false;
        for (;;) {
            members = pre;
            members.insert(members.end(), d.a.begin(), d.a.end());
            members.insert(members.end(), d.b.begin(), d.b.end());
            members.push_back(d.mix);
            members.insert(members.end(), post.begin(), post.end());
            slots.assign(pre.size(), 0);
            slots.insert(slots.end(), d.a.size(), 1);
            slots.insert(slots.end(), d.b.size(), 2);
            slots.insert(slots.end(), 1 + post.size(), 0);
            if (members.size() <= (size_t)kMaxFusedOps) {
                std::vector<Op> ops = ops_of_members(plan, members, slots, nullptr);
                if (stream_supported(ops.data(), (int)ops.size(), allow_jit)) { ok = true; break; }
            }
            if (!post.empty()) post.pop_back();                 // shrink: nodes after the join first, then those before the fork
            else if (!pre.empty()) pre.erase(pre.begin());
            else break;
        }
        if (!ok) continue;
        d.pre = pre;
        d.post = post;
        for (const auto& n : members) in_diamond.insert(n);
        diamonds.push_back(d);
    }
    for (const auto& d : diamonds) {
        PipelineInfo f;
        for (const auto& n : d.pre) { f.members.push_back(n); f.member_slot.push_back(0); }
        for (const auto& n : d.a) { f.members.push_back(n); f.member_slot.push_back(1); }
        for (const auto& n : d.b) { f.members.push_back(n); f.member_slot.push_back(2); }
        f.members.push_back(d.mix);
        f.member_slot.push_back(0);
        for (const auto& n : d.post) { f.members.push_back(n); f.member_slot.push_back(0); }
        for (size_t k = 0; k < f.members.size(); ++k) f.name += (k ? "+" : "") + f.members[k];
        f.input_images = {{d.pre.empty() ? d.src : infos.at(d.pre.front()).input_images[0].first, 1000}};
        f.output_images = {{infos.at(d.post.empty() ? d.mix : d.post.back()).output_images[0].first, 1001}};
        for (const auto& n : f.members) infos.erase(n);
        infos[f.name] = f;
    }
    if (!diamonds.empty()) {                          // the maps below describe the graph with the fused units in it
        producers.clear();
        consumers.clear();
        for (const auto& kv : infos) {
            for (const auto& o : kv.second.output_images) producers[o.first].push_back(kv.first);
            for (const auto& i : kv.second.input_images) consumers[i.first].push_back(kv.first);
        }
        prev_of.clear();
        for (const auto& kv : infos) {
            std::string nx = next_of(kv.first);
            if (!nx.empty()) prev_of[nx] = kv.first;
        }
        root_memo.clear();
        readers_of_root.clear();
        for (const auto& kv : infos)
            for (const auto& in : kv.second.input_images) readers_of_root[root_of(in.first)].insert(kv.first);
    }
    std::vector<std::vector<std::string>> groups;
    for (const auto& kv : infos) {
        if (prev_of.count(kv.first)) continue;   // not a chain head
        std::vector<std::string> chain;
        std::set<std::string> seen;
        for (std::string n = kv.first; !n.empty() && !seen.count(n); n = next_of(n)) {
            chain.push_back(n);
            seen.insert(n);
        }
        // greedy: longest supported prefix first
        size_t i = 0;
        while (i < chain.size()) {
            size_t best = 1;
            for (size_t len = std::min(chain.size() - i, (size_t)kMaxFusedOps); len >= 2; --len) {
                std::vector<Op> ops = ops_of_members(plan, std::vector<std::string>(chain.begin() + i, chain.begin() + i + len), {}, nullptr);
                if (!stream_supported(ops.data(), (int)len, allow_jit)) continue;
                if (!side_effects_stay_inside(std::vector<std::string>(chain.begin() + i, chain.begin() + i + len))) continue;
                best = len;
                break;
            }
            if (best >= 2) groups.emplace_back(chain.begin() + i, chain.begin() + i + best);
            i += best;
        }
    }
    for (const auto& g : groups) {
        PipelineInfo f;
        for (size_t k = 0; k < g.size(); ++k) {
            f.name += (k ? "+" : "") + g[k];
            f.members.push_back(g[k]);
        }
        // private binding numbers: a fused chain is never an in-place alias
        f.input_images = {{infos.at(g.front()).input_images[0].first, 1000}};
        f.output_images = {{infos.at(g.back()).output_images[0].first, 1001}};
        for (const auto& n : g) infos.erase(n);
        infos[f.name] = f;
    }
}

bool build_plan(const Config& cfg, uint32_t flags, Plan& plan, std::string& err)
{
    plan = Plan();
    // synthesize_config, vkutils.rs:140-196
    for (const auto& kv : cfg.graph_pipelines) {
        const std::string& name = kv.first;
        const std::string& tname = cfg.type_of(name);
        std::string why;
        const NodeType* type = find_type(tname, &why, (flags & kPlanGlslNodes) != 0);
        if (!type) {   // Shader::from_path -> None (utils.rs:23)
            err = "Error reading node type '" + tname + "' for node '" + name + "': " + (why.empty() ? std::string("no such filter") : why);
            return false;
        }
        PipelineInfo info;
        info.name = name;
        info.members = {name};
        for (int side = 0; side < 2; ++side) {
            const auto& descs = side == 0 ? kv.second.inputs : kv.second.outputs;
            auto& dst = side == 0 ? info.input_images : info.output_images;
            auto& bdst = side == 0 ? info.input_ssbos : info.output_ssbos;
            for (const auto& d : descs) {
                int b = type->binding(d.descriptor_name);
                if (b < 0) {
                    // not an image variable: a storage buffer, by its block type name (vkutils.rs:165-170)
                    if (const NodeType::BufferDef* bd = type->buffer(d.descriptor_name)) {
                        const std::pair<std::string, int> e{d.resource_name, bd->binding};
                        if (std::find(bdst.begin(), bdst.end(), e) == bdst.end()) bdst.push_back(e);
                        size_t& bytes = plan.buffer_bytes[d.resource_name];
                        bytes = std::max(bytes, bd->bytes);                  // max over users, pipeline_graph.rs:158-175
                        continue;
                    }
                }
                if (b < 0) {   // vkutils.rs:179
                    err = "Shader " + tname + " has no binding named: " + d.descriptor_name;
                    return false;
                }
                // A node named in several graph expressions ("aa -> bb" and "aa -> cc") gets the same
                // (resource, binding) pushed once per occurrence (config.rs:149-190).  The planner
                // treats the lists as SETS: with a duplicate, reusable_image_remapping (:398-424)
                // first remaps the output onto a free image and then, free list now empty, also
                // records it as an allocation of its own -- an image that is both an alias and a
                // recyclable allocation, which later hands a stencil its own input as output.
                const std::pair<std::string, int> e{d.resource_name, b};
                if (std::find(dst.begin(), dst.end(), e) == dst.end()) dst.push_back(e);
            }
        }
        plan.infos[name] = info;

        NodeParams np;
        np.type = type;
        const auto& given = cfg.params_of(name);
        for (const auto& pd : type->params) {
            auto it = given.find(pd.name);
            np.values[pd.name] = parse_param(it == given.end() ? nullptr : &it->second, pd.type);
        }
        plan.nodes[name] = np;
    }

    if (!(flags & kPlanNoFusion)) {
        size_t before = plan.infos.size();
        fuse_chains(plan, plan.infos, !(flags & kPlanNoJit));
        plan.fused = plan.infos.size() != before;
    }

    if (!order_by_execution(plan.infos, plan.layers, err)) return false;
    plan.reuse = reusable_image_remapping(plan.layers, plan.infos);

    // images created per frame, pipeline_graph.rs:205-224
    std::set<std::string> images;
    for (const auto& layer : plan.layers) {
        for (const auto& n : layer) {
            const PipelineInfo& info = plan.infos.at(n);
            for (const auto& in : info.input_images)
                if (in.first == kFileInput) images.insert(in.first);
            for (const auto& out : info.output_images) images.insert(plan.resolve(out.first));
        }
    }
    plan.images.assign(images.begin(), images.end());

    // storage buffers, pipeline_graph.rs:240-260: an output on the binding of an input is that input's buffer (point op);
    // every other output allocates; sizes of aliased names merge (the reference sizes each NAME separately and creates the
    // buffer under the resolved one, :252-253)
    for (const auto& layer : plan.layers)
        for (const auto& n : layer) {
            const PipelineInfo& info = plan.infos.at(n);
            for (const auto& out : info.output_ssbos)
                for (const auto& in : info.input_ssbos)
                    if (out.second == in.second) plan.buffer_reuse[out.first] = in.first;
        }
    std::set<std::string> buffers;
    for (const auto& layer : plan.layers)
        for (const auto& n : layer)
            for (const auto& out : plan.infos.at(n).output_ssbos) {
                const std::string& nm = plan.resolve_buffer(out.first);
                buffers.insert(nm);
                plan.buffer_bytes[nm] = std::max(plan.buffer_bytes[nm], plan.buffer_bytes[out.first]);
            }
    plan.buffers.assign(buffers.begin(), buffers.end());
    return true;
}

std::vector<Op> ops_of_members(const Plan& plan, const std::vector<std::string>& members, const std::vector<int>& member_slot,
                               const std::map<std::string, float*>* dev_weights)
{
    std::vector<Op> ops;
    for (size_t k = 0; k < members.size(); ++k) {
        const float* w = nullptr;
        if (dev_weights) {
            auto it = dev_weights->find(members[k]);
            if (it != dev_weights->end()) w = it->second;
        }
        Op op = plan.nodes.at(members[k]).to_op(w);
        op.slot = k < member_slot.size() ? member_slot[k] : 0;
        ops.push_back(op);
    }
    return ops;
}

static bool point_kind(int kind) { return kind == OP_PASSTHROUGH || kind == OP_GRADE || kind == OP_SPLIT || kind == OP_USERN; }    // (device kinds: conv2d_weights is a passthrough, pulse a grade)

bool build_launches(const Plan& plan, std::vector<LaunchDesc>& out, std::string& err)
{
    out.clear();
    for (size_t layer = 0; layer < plan.layers.size(); ++layer) {
        for (const auto& unit : plan.layers[layer]) {
            // a node named in several graph expressions lists the same (resource, binding)
            // more than once (config.rs:149-190 pushes per occurrence); the reference binds
            // the same image to the same slot again, which is harmless
            PipelineInfo info = plan.infos.at(unit);
            auto dedupe = [](std::vector<std::pair<std::string, int>>& v) {
                std::vector<std::pair<std::string, int>> u;
                for (const auto& e : v)
                    if (std::find(u.begin(), u.end(), e) == u.end()) u.push_back(e);
                v.swap(u);
            };
            dedupe(info.input_images);
            dedupe(info.output_images);
            if (info.output_images.empty()) continue;   // nothing observable is written
            LaunchDesc L;
            L.label = unit;
            L.members = info.members;
            L.member_slot = info.member_slot;
            L.layer = (int)layer;
            const int kind0 = plan.nodes.at(info.members[0]).type->kind;
            if (info.input_images.empty()) {
                err = "node '" + unit + "' has no input image (a graph must start at 'input')";
                return false;
            }
            const UserStage* unode = kind0 == OP_USERN ? user_stage_of(plan.nodes.at(info.members[0]).type) : nullptr;
            if (kind0 == OP_USERN && !unode) { err = "node '" + unit + "': its stage file is no longer registered"; return false; }
            if (unode) {
                // every declared input image, in declaration order (= binding order)
                for (size_t i = 0; i < unode->inputs.size(); ++i) {
                    std::string res;
                    for (const auto& in : info.input_images)
                        if (in.second == unode->in_binding[i]) {
                            if (!res.empty() && res != in.first) { err = "node '" + unit + "' wires two images to " + unode->inputs[i]; return false; }
                            res = in.first;
                        }
                    if (res.empty()) { err = "node '" + unit + "' needs an image wired to " + unode->inputs[i]; return false; }
                    L.src.push_back(plan.resolve(res));
                }
                auto name_of = [&](int binding) {
                    for (const auto& im : unode->node_type.images)
                        if (im.second == binding) return std::string(im.first);
                    return std::string("?");
                };
                for (const auto& in : info.input_images)
                    if (std::find(unode->in_binding.begin(), unode->in_binding.end(), in.second) == unode->in_binding.end()) {
                        err = "node '" + unit + "': " + name_of(in.second) + " is an output image of " + unode->type_name + ", the graph wires it as an input";
                        return false;
                    }
                for (const auto& o : info.output_images)
                    if (std::find(unode->out_binding.begin(), unode->out_binding.end(), o.second) == unode->out_binding.end()) {
                        err = "node '" + unit + "': " + name_of(o.second) + " is an input image of " + unode->type_name + ", the graph wires it as an output";
                        return false;
                    }
            } else if (kind0 == OP_MIX) {
                std::string a, b;
                for (const auto& in : info.input_images) {
                    if (in.second == 0) a = in.first;
                    if (in.second == 1) b = in.first;
                }
                if (a.empty() || b.empty() || info.input_images.size() != 2) {
                    err = "node '" + unit + "' needs exactly input_image0 and input_image1";
                    return false;
                }
                L.src = {plan.resolve(a), plan.resolve(b)};
            } else {
                if (info.input_images.size() != 1) {
                    err = "node '" + unit + "' takes one input image, the graph wires " + std::to_string(info.input_images.size());
                    return false;
                }
                L.src = {plan.resolve(info.input_images[0].first)};
            }
            if (kind0 == OP_SPLIT || kind0 == OP_USERN) {
                // several output bindings: an allocated image each (pipeline_graph.rs:205-224), in binding order
                std::vector<std::pair<int, std::string>> outs;
                for (const auto& o : info.output_images) outs.push_back({o.second, plan.resolve(o.first)});
                std::sort(outs.begin(), outs.end());
                for (size_t k = 0; k + 1 < outs.size(); ++k) {
                    if (outs[k].first == outs[k + 1].first) { err = "node '" + unit + "' wires one output binding to two images"; return false; }
                    if (outs[k].second == outs[k + 1].second) { err = "node '" + unit + "' would write two outputs into one image"; return false; }
                }
                for (const auto& o : outs) { L.dsts.push_back(o.second); L.dst_bindings.push_back(o.first); }
            } else {
                if (info.output_images.size() != 1) {
                    err = "node '" + unit + "' writes one output image, the graph wires " + std::to_string(info.output_images.size());
                    return false;
                }
                L.dsts = {plan.resolve(info.output_images[0].first)};
                L.dst_bindings = {info.output_images[0].second};
            }
            L.dst = L.dsts[0];
            for (const auto& b : info.input_ssbos) {
                const std::string& nm = plan.resolve_buffer(b.first);
                if (std::find(plan.buffers.begin(), plan.buffers.end(), nm) == plan.buffers.end()) {
                    err = "No buffer found for input " + b.first;   // pipeline_graph.rs:269
                    return false;
                }
                L.in_buffers.push_back(nm);
                L.in_buffer_bindings.push_back(b.second);
            }
            for (const auto& b : info.output_ssbos) { L.out_buffers.push_back(plan.resolve_buffer(b.first)); L.out_buffer_bindings.push_back(b.second); }
            if (unode) {
                auto has = [](const std::vector<UserStage::Buffer>& v, int binding) {
                    for (const auto& b : v)
                        if (b.binding == binding) return true;
                    return false;
                };
                auto block_of = [&](int binding) {
                    for (const auto* list : {&unode->buf_in, &unode->buf_out})
                        for (const auto& b : *list)
                            if (b.binding == binding) return b.name;
                    return std::string("?");
                };
                // a block type name wired on the wrong side (the planner looked it up without regard to direction)
                for (int bb : L.in_buffer_bindings)
                    if (!has(unode->buf_in, bb)) { err = "node '" + unit + "': " + block_of(bb) + " is the buffer " + unode->type_name + " writes, the graph wires it as an input"; return false; }
                for (int bb : L.out_buffer_bindings)
                    if (!has(unode->buf_out, bb)) { err = "node '" + unit + "': " + block_of(bb) + " is the buffer " + unode->type_name + " reads, the graph wires it as an output"; return false; }
                // every block the node only READS must be wired (a block it fills may stay unwired: a stage file's is then not filled, a .comp file's is private to the node)
                for (const auto& b : unode->buf_in) {
                    if (has(unode->buf_out, b.binding)) continue;      // updated in place: wired on either side, or private to the node (rf_graph.cpp allocates it)
                    if (std::find(L.in_buffer_bindings.begin(), L.in_buffer_bindings.end(), b.binding) == L.in_buffer_bindings.end()) {
                        err = "node '" + unit + "' needs a storage buffer wired to " + b.name;
                        return false;
                    }
                }
            }
            for (const auto& s : L.src) {
                if (std::find(plan.images.begin(), plan.images.end(), s) == plan.images.end()) {
                    err = "No image found for input " + s;   // pipeline_graph.rs:236
                    return false;
                }
            }
            bool all_point = true;
            std::vector<Op> ops = ops_of_members(plan, L.members, L.member_slot, nullptr);
            for (const auto& o : ops) all_point = all_point && (point_kind(o.kind) || (o.kind == OP_USER && o.radius == 0)) && !(o.kind == OP_USERN && o.radius > 0);      // (OP_USER of radius 0: a point stage -- a .comp point shader may be declared in place)
            L.radius = ops_radius(ops.data(), (int)ops.size());
            bool reads_what_it_writes = false;
            for (const auto& sname : L.src)
                for (const auto& d : L.dsts) reads_what_it_writes = reads_what_it_writes || sname == d;
            if (kind0 == OP_USERN ? (reads_what_it_writes && !all_point) : (kind0 != OP_MIX && L.src[0] == L.dst && !all_point)) {
                err = "node '" + unit + "' would run a stencil in place";
                return false;
            }
            out.push_back(L);
        }
    }
    // Intra-layer hazards.  The reference runs the nodes of a layer concurrently behind one
    // barrier (command.rs:226-240) and aliases a point op written in place (X:image) onto its
    // input even when another node of the same layer reads that image: a data race there.  Here
    // such a layer runs in plan order -- name order, as the oracle executes it -- so the result
    // is defined: the launches that sort before the in-place writer see the original image.
    for (size_t a = 0; a < out.size(); ++a) {
        for (size_t b = 0; b < out.size(); ++b) {
            if (a == b || out[a].layer != out[b].layer) continue;
            bool touches = false;
            for (const auto& da : out[a].dsts) {
                for (const auto& db : out[b].dsts) touches = touches || da == db;
                for (const auto& sname : out[b].src) touches = touches || sname == da;
            }
            if (touches) out[a].serial = out[b].serial = true;
        }
    }
    for (auto& L : out) {
        L.result_only = true;
        for (const auto& other : out)
            for (const auto& sname : other.src)
                for (const auto& d : L.dsts) L.result_only = L.result_only && sname != d;
    }
    for (size_t a = 0; a < out.size(); ++a)      // the whole layer, not only the pair
        for (size_t b = 0; b < out.size(); ++b)
            if (out[a].layer == out[b].layer && out[b].serial) out[a].serial = true;
    return true;
}

void halo_schedule(std::vector<LaunchDesc>& launches, bool multi_rank, bool exchange, int& need_input, int& ghost)
{
    need_input = 0;
    ghost = 0;
    if (!multi_rank || exchange) {
        // single rank: the clamp bounds are clipped to the frame, nothing is allocated
        for (auto& L : launches) {
            L.need_src = L.radius;
            L.need_dst = 0;
            if (multi_rank) ghost = std::max(ghost, L.radius);
        }
        return;
    }
    // over-fetch: walk the frame backwards; need[image] = ghost rows its pending readers want
    std::map<std::string, int> need;
    for (size_t k = launches.size(); k-- > 0;) {
        LaunchDesc& L = launches[k];
        int nd = 0;
        for (const auto& d : L.dsts) {
            auto it = need.find(d);
            if (it != need.end()) { nd = std::max(nd, it->second); need.erase(it); }
        }
        L.need_dst = nd;
        L.need_src = nd + L.radius;
        for (const auto& s : L.src) need[s] = std::max(need.count(s) ? need[s] : 0, L.need_src);
        ghost = std::max(ghost, L.need_src);
    }
    auto it = need.find(kFileInput);
    if (it != need.end()) need_input = it->second;
}

void strip_rows(int height, int world, int rank, int& y0, int& y1)
{
    // contiguous strips; the first (height % world) ranks hold one extra row
    const int base = height / world, extra = height % world;
    y0 = rank * base + std::min(rank, extra);
    y1 = y0 + base + (rank < extra ? 1 : 0);
}

}  // namespace rf
