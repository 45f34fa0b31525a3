/*
 * rfhip.h -- C ABI of librfhip.so: the MI355X (gfx950) HIP runtime that replaces
 * reforge's `vulkan::*` backend and its GLSL compute shaders for the
 * render-graph execution path.
 *
 * The reference has no FFI in code; the seam this ABI replaces is the set of calls
 * `src/render.rs` / `src/main.rs` make into the `src/vulkan` modules.
 * Every entry point below cites the reference call it stands in for.  The
 * reference-side binding a maintainer would add (Rust `extern "C"` block) is in
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; opaque handles; the library owns handles, the caller owns
 *     host buffers.
 *   - every function returns an rf_status (0 = ok).  Nothing aborts or throws
 *     across the boundary.  rf_last_error() returns the message of the calling
 *     thread's most recent failure (the reference prints it with warnln!,
 *     src/utils.rs:13-18, or panics for device errors).
 *   - one host thread per rf_ctx (the reference is single-threaded, Rc<RefCell>).
 *   - functions marked [host] touch no GPU and work on a machine without one.
 *   - there is NO CPU fallback: any device entry point fails with
 *     RF_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef RFHIP_H
#define RFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_ABI_VERSION 4

typedef enum rf_status {
    RF_OK              = 0,
    RF_ERR_INVALID     = 1,   /* bad argument / null handle */
    RF_ERR_CONFIG      = 2,   /* config DSL rejected (reference: parse -> None + warnln!) */
    RF_ERR_GRAPH       = 3,   /* graph cannot be built (unknown type/binding, cycle)     */
    RF_ERR_NO_DEVICE   = 4,   /* no usable gfx950 device                                  */
    RF_ERR_DEVICE      = 5,   /* HIP / RCCL runtime error (reference: unwrap()/panic!)    */
    RF_ERR_UNSUPPORTED = 6,
    RF_WARN_UNKNOWN_PARAM = 16 /* set_param on a name the node type lacks: ignored,
                                  like an unmatched UBO member (render.rs:197-203)       */
} rf_status;

/* --shader-format (src/main.rs:27-41) */
typedef enum rf_format {
    RF_FORMAT_RGBA8   = 0,    /* VK_FORMAT_R8G8B8A8_UNORM      */
    RF_FORMAT_RGBA32F = 1     /* VK_FORMAT_R32G32B32A32_SFLOAT */
} rf_format;

typedef enum rf_param_type { RF_PARAM_F32 = 0, RF_PARAM_I32 = 1, RF_PARAM_BOOL = 2 } rf_param_type;

/* rf_graph_options.flags */
#define RF_GRAPH_TIMERS      0x1u  /* per-node hipEvent timers (GpuTimer, vkutils.rs:47-135)   */
#define RF_GRAPH_NO_FUSION   0x2u  /* one kernel launch per node, exactly as command.rs:220-240 */
#define RF_GRAPH_HIPGRAPH    0x4u  /* replay the recorded frame as one hipGraph                 */
#define RF_GRAPH_NO_HALO_XCHG 0x8u /* multi-rank: over-fetch the cumulative halo at upload
                                      instead of a per-node RCCL exchange                      */
#define RF_GRAPH_NO_JIT      0x10u /* fuse only chains whose kernel is in the ahead-of-time catalogue: nothing is
                                      compiled at graph creation (the reference compiles every node's shader
                                      there, shader.rs:29-93; here only chains the catalogue lacks need it)      */
#define RF_GRAPH_GLSL_NODES  0x20u /* every {type}.comp file is a node with a kernel of its own: none becomes a row stage of the
                                      stream kernel (what rf_graph_create checks such a stage against; A/B measurements) */

typedef struct rf_ctx    rf_ctx;     /* VkCore            src/vulkan/core.rs:47-64   */
typedef struct rf_config rf_config;  /* config::Config    src/config/config.rs:35-38 */
typedef struct rf_plan   rf_plan;    /* layers + aliasing src/vulkan/pipeline_graph.rs:358-497 */
typedef struct rf_graph  rf_graph;   /* PipelineGraph + its frames  pipeline_graph.rs:43-57    */

/* rf_graph_options.exec_flags: executor variants kept for measurement and for the tests.
 * Each also has an environment override read at rf_graph_create (RF_SYNC_LAUNCHES=1, ...):
 * the environment wins when set. */
#define RF_EXEC_SYNC_LAUNCHES     0x1u  /* host-synchronise after every launch (debugging aid)       */
#define RF_EXEC_CONCURRENT_LAYERS 0x2u  /* launches of a hazard-free layer on side streams, as the
                                           reference lets a layer overlap (command.rs:194-240)      */
#define RF_EXEC_FORCE_SPLIT       0x4u  /* interior/boundary three-part stencil launch without an
                                           exchange (how the multi-rank split is tested on one GPU) */
#define RF_EXEC_NO_ALTERNATE      0x8u  /* every chunk walks top-down (default: chosen per launch)      */
#define RF_EXEC_ALTERNATE         0x20u /* odd chunks walk bottom-up (halo rows shared through L2)      */
#define RF_EXEC_GLSL_NO_WINDOW    0x40u /* a {type}.comp stencil runs on its generic kernel only (the
                                          * LDS-tiled window kernel is what rf_graph_create checks it against) */

typedef struct rf_graph_options {
    int       width;        /* RenderInfo.width   src/render.rs:40 */
    int       height;       /* RenderInfo.height  (the FULL frame height on every rank) */
    rf_format format;       /* RenderInfo.format  */
    int       num_frames;   /* frames in flight   src/main.rs:69-70; >= 1 */
    uint32_t  flags;        /* RF_GRAPH_* */
    /* tuning, 0 = the library's own choice (RF_ABI_VERSION >= 2; no reference counterpart) */
    int       rows_per_chunk; /* rows a wave walks per chunk (env RF_ROWS_PER_CHUNK)              */
    int       conv_path;      /* conv2d kernel: 0 auto, 1 LDS tile, 2 MFMA band, 3 VALU (env RF_CONV_PATH)      */
    uint32_t  exec_flags;     /* RF_EXEC_* */
    int       texels_per_lane;/* stream kernels, rgba32f: 0 auto, 1 = 64-wide strips, 2 = 128-wide
                                 (env RF_TEXELS_PER_LANE)                                         */
} rf_graph_options;

/* ------------------------------------------------------------------------- */
/* Errors                                                                      */
/* ------------------------------------------------------------------------- */
/* [host] message of this thread's last failing call ("" if none) */
const char* rf_last_error(void);
/* [host] RF_ABI_VERSION the library was built with */
int rf_abi_version(void);

/* ------------------------------------------------------------------------- */
/* Config DSL  (src/config/config.rs, config_grammar.lalrpop)                  */
/* ------------------------------------------------------------------------- */
/* [host] config::parse            src/config/config.rs:98-205
 *        expects_input = an input image exists (has_input_image, render.rs:46) */
rf_status rf_config_parse(const char* text, int expects_input, rf_config** out);
/* [host] the generated parser alone: `config_grammar::ExprListParser::new().parse(contents)`, src/config/config.rs:105
 *        (grammar: src/config/config_grammar.lalrpop:7-81) -- the syntax tree of `text` as JSON,
 *          {"exprs": [["pipeline", name, type, [[key, value], ...]] | ["graph", [[name, descriptor | null], ...]] | ["comment", text]]}
 *        (parameters in source order, duplicates kept), written NUL-terminated to buf[0..cap); *len = its length without the
 *        NUL (also when cap is too small: RF_ERR_INVALID, call again).  RF_ERR_CONFIG + rf_last_error for a text the
 *        grammar rejects.  None of config::parse's own checks (empty text, 'output' never used ...) is applied. */
rf_status rf_config_syntax(const char* text, char* buf, size_t cap, size_t* len);
/* [host] config::single_shader_parse  src/config/config.rs:77-90
 *        `type_name` = the shader's file stem */
rf_status rf_config_single(const char* type_name, int expects_input, rf_config** out);
void      rf_config_destroy(rf_config* cfg);
/* [host] enumeration of Config.graph_pipelines (name-sorted) */
int         rf_config_num_nodes(const rf_config* cfg);
const char* rf_config_node_name(const rf_config* cfg, int node);
/* node type: the instance's pipeline_type, else the node name (config.rs:59-75) */
const char* rf_config_node_type(const rf_config* cfg, int node);
int         rf_config_node_num_inputs(const rf_config* cfg, int node);
int         rf_config_node_num_outputs(const rf_config* cfg, int node);
/* ConfigDescriptor {resource_name, descriptor_name}  config.rs:17-21 */
const char* rf_config_node_input_resource(const rf_config* cfg, int node, int i);
const char* rf_config_node_input_descriptor(const rf_config* cfg, int node, int i);
const char* rf_config_node_output_resource(const rf_config* cfg, int node, int i);
const char* rf_config_node_output_descriptor(const rf_config* cfg, int node, int i);
/* PipelineInstance.parameters (config.rs:30-33), key-sorted; value in string form */
int         rf_config_node_num_params(const rf_config* cfg, int node);
const char* rf_config_node_param_key(const rf_config* cfg, int node, int i);
const char* rf_config_node_param_value(const rf_config* cfg, int node, int i);

/* ------------------------------------------------------------------------- */
/* Plan: layering + image aliasing (+ this build's fusion groups)             */
/* ------------------------------------------------------------------------- */
/* [host] vkutils::synthesize_config (vkutils.rs:140-196) +
 *        PipelineGraph::order_by_execution (pipeline_graph.rs:429-497) +
 *        PipelineGraph::reusable_image_remapping (pipeline_graph.rs:358-427) */
rf_status rf_plan_create(const rf_config* cfg, uint32_t flags, rf_plan** out);
void      rf_plan_destroy(rf_plan* plan);
int         rf_plan_num_layers(const rf_plan* plan);
int         rf_plan_layer_size(const rf_plan* plan, int layer);
const char* rf_plan_layer_node(const rf_plan* plan, int layer, int i);   /* name-sorted */
/* image_reuse_remapping entries, key-sorted (pipeline_graph.rs:31,:358) */
int         rf_plan_num_aliases(const rf_plan* plan);
const char* rf_plan_alias_from(const rf_plan* plan, int i);
const char* rf_plan_alias_to(const rf_plan* plan, int i);
/* images actually allocated per frame, name-sorted (pipeline_graph.rs:205-224) */
int         rf_plan_num_images(const rf_plan* plan);
const char* rf_plan_image_name(const rf_plan* plan, int i);
/* remap_resource_name (pipeline_graph.rs:75-79) */
const char* rf_plan_resolve(const rf_plan* plan, const char* resource);
/* storage buffers (SSBO edges: `kw:ConvWeights -> conv:ConvWeights`; pipeline_graph.rs:142-175, :240-265): the names
 * allocated per graph (name-sorted), their size = max over users of the block's bytes, and the point-op aliasing
 * (an output on the binding of an input is that input's buffer) */
int         rf_plan_num_buffers(const rf_plan* plan);
const char* rf_plan_buffer_name(const rf_plan* plan, int i);
size_t      rf_plan_buffer_bytes(const rf_plan* plan, int i);
const char* rf_plan_resolve_buffer(const rf_plan* plan, const char* resource);
/* kernel launches per frame after fusion, and the nodes each one covers
 * ("a+b+c"); with RF_GRAPH_NO_FUSION one launch per node */
int         rf_plan_num_launches(const rf_plan* plan);
const char* rf_plan_launch_label(const rf_plan* plan, int i);
int         rf_plan_launch_layer(const rf_plan* plan, int i);
int         rf_plan_launch_num_members(const rf_plan* plan, int i);
const char* rf_plan_launch_member(const rf_plan* plan, int i, int k);
/* a fused fork/join launch (two branches from one image + the `combination` that joins them) lists the nodes of the branch
 * feeding input_image0 (slot 1), then those of the branch feeding input_image1 (slot 2), then the join (slot 0); every member
 * of any other launch has slot 0 */
int         rf_plan_launch_member_slot(const rf_plan* plan, int i, int k);
/* allocated images a launch reads (1; 2 for an unfused `combination`) and writes */
int         rf_plan_launch_num_inputs(const rf_plan* plan, int i);
const char* rf_plan_launch_input(const rf_plan* plan, int i, int k);
const char* rf_plan_launch_output(const rf_plan* plan, int i);
/* a node with several output bindings writes several allocated images (one per binding, pipeline_graph.rs:205-224; the
 * built-in `split_luma` has luma_image and chroma_image): all of them, in binding order */
int         rf_plan_launch_num_outputs(const rf_plan* plan, int i);
const char* rf_plan_launch_output_at(const rf_plan* plan, int i, int k);
/* rows a launch reads above/below the rows it writes: the sum of its members' stencil radii; for a fused fork/join launch
 * pre + max(a, b) + post -- its two branches run side by side inside the kernel (the shorter one ends in a delay line), so this
 * is also what a row strip exchanges or over-fetches for it */
int         rf_plan_launch_radius(const rf_plan* plan, int i);
/* 1 if the launch's layer runs in plan order on one stream: one of its launches writes an image
 * another launch of the layer reads or writes (an in-place point op beside a second consumer).
 * The reference runs such a layer concurrently (command.rs:194-240), a data race. */
int         rf_plan_launch_serial(const rf_plan* plan, int i);
/* [host] a 64-bit digest of the launch list (labels, radii, layers, outputs; never 0 for a valid plan).  Row-strip ranks
 * in exchange mode must run the SAME list -- it fixes how many rows every neighbour send/recv carries -- so a caller
 * compares rf_plan_signature(rf_graph_plan(g)) across ranks before the first frame (bench.py does, with an all-reduce).
 * rf_graph_create never lets a rank change its list on its own there: if a fused launch cannot be compiled in exchange
 * mode it fails with RF_ERR_UNSUPPORTED instead of falling back to catalogue-only fusion. */
uint64_t    rf_plan_signature(const rf_plan* plan);
/* Filter types that are FILES.  In the reference a type T is {shader_path}/T.comp (src/config/config.rs:59-75), compiled
 * and reflected when the graph is built (src/vulkan/shader.rs:29-59,:106-160).  Here a type the built-in registry lacks is
 * looked for as {shader_path}/T.stage.hip: `struct Params { float|int|bool members = the config's parameter names };
 * static constexpr int RADIUS = 0|1; RF_STAGE f4 apply(const Params&, f4)` (point op) or `apply(const Params&, const f4
 * (&n)[3][3])` (3x3 neighbourhood, clamp-to-edge), image bindings input_image / output_image as passthrough.comp:4-5.
 * It becomes a row stage of the stream kernel -- user nodes fuse with the built-in ones -- compiled by hiprtc in
 * rf_graph_create; a file that does not parse or compile fails rf_plan_create / rf_graph_create with RF_ERR_GRAPH and the
 * compiler's message (the caller keeps the graph it has: render.rs:121-136).  shaders/edge_detect.stage.hip and
 * shaders/invert.stage.hip are the first two.
 * A file may DECLARE what a .comp file declares (found by name, shader.rs:144-153): `RF_INPUTS(a, b); RF_OUTPUTS(c, d);` -- up to 4
 * input and 4 output image variables, a name on both sides = one binding = written in place (pipeline_graph.rs:402-406) --
 * `RF_BUFFER_IN(BlockType, floats);` / `RF_BUFFER_OUT(BlockType, floats);` -- one storage block read (a `const float*` argument),
 * one filled (`RF_STAGE float fill(const Params&, int i)`).  Such a type is a node with a run-time compiled kernel of its own:
 * `apply(const Params&, const f4 (&in)[NI], f4 (&out)[NO])`, or with RADIUS R in 1..15 `const Window (&in)[NI]` and
 * `in[k].at(dx, dy)` (clamp-to-edge) -- every file with RADIUS >= 2 takes that form (shaders/unsharp_mask, tone_curve,
 * apply_curve, local_contrast .stage.hip).
 * [host] process-wide, like Render's shader path (render.rs:537-588). */
rf_status   rf_set_shader_path(const char* dir);
const char* rf_shader_path(void);
/* [host] modification time (ns) of {shader_path}/{type}.stage.hip (or {type}.comp) as last loaded, -1 if there is no such file: what a
 * live-reload loop polls (reload_changed_pipelines, render.rs:225-249) */
long long   rf_user_stage_mtime(const char* type_name);
/* Filter types in the reference's OWN file form: {shader_path}/T.comp, a GLSL 450 compute shader -- what a reforge user already
 * has (src/config/config.rs:59-75; shaders/passthrough.comp is the one the reference ships).  A type that is neither built in nor
 * a T.stage.hip is looked for as T.comp.  No GLSL compiler is needed: rf_glsl.cpp TRANSLATES the subset a compute filter uses to
 * HIP device source (vectors with swizzles, matrices, the built-in functions, image2D storage images and sampler2D combined image
 * samplers by variable name -- texture() filters as the reference's one sampler does, src/vulkan/vkutils.rs:358-365 --, uniform
 * blocks -- their scalar members are the node's parameters, std140 --, storage blocks by block TYPE name -- std430 --, structs,
 * #define, shared variables and barrier()), hiprtc compiles it at rf_graph_create, and rfglsl::glsl_node_kernel runs one invocation
 * per thread over the reference's dispatch, ceil(W/16) x ceil(H/16) workgroups of the file's local_size (src/vulkan/command.rs:167-168).
 * What shaderc + spirv-reflect give the reference (src/vulkan/shader.rs:73-160) falls out of the same parse.  Any number of images
 * (<= 32) and storage blocks (<= 32) per node; an image or block without readonly / writeonly is read and written in place
 * (pipeline_graph.rs:228,:240-247).  `#pragma rf radius N` (a GLSL compiler ignores it) states how many rows above / below its own
 * an invocation reads: required of a node that is to be split into row strips.  A file outside the subset is refused with
 * "T.comp:LINE: why", like a file that does not compile (Option::None + warning, shader.rs:92; the caller keeps its graph).
 * A file that is provably a POINT operation on one image (every load and store at the invocation's own texel, position used only in
 * the frame guard) also becomes a row stage of the stream kernel and fuses with its neighbours (rf_plan_launch_label shows it).
 * A stencil file (radius >= 2) that is provably translation-invariant also runs on the LDS-tiled window kernel of the stage files;
 * rf_graph_create compares that kernel with the file's generic kernel on a random frame and keeps it only if they agree (rf_graph_note).
 * [host] the translation of `text` (HIP device source; its first line is a comment naming the namespace it lives in) */
rf_status   rf_glsl_translate(const char* type_name, const char* text, char* buf, size_t cap, size_t* len);
/* [host] the reflection of `text` as JSON: {"point", "stencil", "box" (how the file can run besides its generic kernel: a fused row stage /
 * the LDS-tiled window kernel / a fused 3 x 3 row stage), "stencil_why_not" (the first thing that kept it from being a recognised
 * stencil: line and name), "local_size": [x, y, z], "grouped", "radius" (-1: not stated), "uniform_bytes",
 * "images": [{"name", "binding", "readonly", "writeonly"}], "uniform_blocks" / "storage_blocks": [{"type_name", "instance",
 * "binding", "bytes", "base", "members": [{"name", "base": "f|i|u|b", "comps", "cols", "dims", "offset", "stride", "bytes"}]}]} */
rf_status   rf_glsl_reflect(const char* type_name, const char* text, char* buf, size_t cap, size_t* len);
/* [host] which wins when a type name is both built in and a file in {shader_path}: 0 (default) the built-in, hand-written
 * kernel; 1 the FILE -- the reference's rule, where the file is the type: a reforge user's shader directory then runs as it is */
rf_status   rf_set_type_lookup(int files_first);
int         rf_type_lookup(void);
/* Kernels compiled at graph creation.  A fused launch whose stage list the ahead-of-time kernel catalogue lacks is
 * compiled by rf_graph_create with hiprtc from the library's own device source -- the counterpart of
 * Shader::from_path + Pipeline::new_compute (src/vulkan/shader.rs:29-93, pipeline.rs:73-88), which run at the same
 * point of the reference.  RF_GRAPH_NO_JIT (or env RF_NO_JIT=1) plans with the catalogue alone.
 * Optional disk cache of code objects: env RF_JIT_CACHE_DIR. */
/* [host] 1 if libhiprtc can be loaded and RF_NO_JIT is unset */
int         rf_jit_available(void);
/* [host] path of the libhiprtc in use ("" if none).  The library binds libhiprtc by SONAME on first use: a process that
 * imported PyTorch first compiles with the copy PyTorch bundles, another compiler build than /opt/rocm's (both are checked
 * by tests/test_jit_isa.py; the disk cache keeps their objects apart) */
const char* rf_jit_library(void);
/* [host] kernels this process has compiled so far (cache hits not counted) */
int         rf_jit_compile_count(void);
/* [host] 1 if launch i needs a kernel the catalogue does not hold */
int         rf_plan_launch_needs_jit(const rf_plan* plan, int i);
/* [host] compiles every such kernel of the plan for gfx950 WITHOUT a device (nothing is loaded): proves on a
 * GPU-less machine that the generated instantiations build; *code_bytes = total code object size */
rf_status   rf_plan_jit_compile(const rf_plan* plan, int format, size_t* code_bytes);
/* [host] the same for the variant with `texels_per_lane` (1 or 2) texels per lane; launches that have no two-texel variant are skipped */
rf_status   rf_plan_jit_compile_texels(const rf_plan* plan, int format, int texels_per_lane, size_t* code_bytes);
/* [host] ghost-row schedule of a row-strip partition (new: SURVEY.md 8e).
 *   exchange != 0: before launch i its input's need_src[i] = radius edge rows are
 *                  exchanged with the neighbour ranks; need_dst[i] = 0.
 *   exchange == 0: over-fetch (RF_GRAPH_NO_HALO_XCHG): the input carries *need_input
 *                  ghost rows; launch i reads need_src[i] and also writes need_dst[i]
 *                  ghost rows, so no per-launch communication is needed.
 * n = capacity of the arrays (>= rf_plan_num_launches); *ghost = rows to allocate. */
rf_status   rf_plan_halo_schedule(const rf_plan* plan, int exchange, int* need_src, int* need_dst, int n,
                                  int* need_input, int* ghost);

/* ------------------------------------------------------------------------- */
/* Node-type registry: what SPIR-V reflection gives the reference             */
/* (src/vulkan/shader.rs:106-160)                                              */
/* ------------------------------------------------------------------------- */
/* [host] */
int         rf_registry_num_types(void);
const char* rf_registry_type_name(int t);
/* binding index of an image variable name, -1 if the type has none */
int         rf_registry_binding(const char* type_name, const char* descriptor);
/* binding index of a storage buffer by its block TYPE name (shader.rs:144-147), -1 if the type has none */
int         rf_registry_buffer_binding(const char* type_name, const char* block_type_name);
/* vertical / horizontal stencil radius a node of this type reads (0 = point op);
 * radius-parameterised types report their maximum */
int         rf_registry_radius(const char* type_name);

/* ------------------------------------------------------------------------- */
/* Row-strip partition (new: the reference is single-device)                   */
/* ------------------------------------------------------------------------- */
/* [host] rows [*y0, *y1) of a frame of `height` rows owned by `rank` of `world` */
rf_status rf_strip_rows(int height, int world, int rank, int* y0, int* y1);

/* ------------------------------------------------------------------------- */
/* Context  (VkCore::new src/vulkan/core.rs:66-146; Drop :266-286)             */
/* ------------------------------------------------------------------------- */
/* single device, single process */
rf_status rf_ctx_create(int device, rf_ctx** out);
/* [host] fills a 128-byte RCCL unique id on the calling rank (rank 0) for
 * rf_ctx_create_dist; the caller broadcasts the bytes to the other ranks */
rf_status rf_comm_unique_id(void* id128);
/* [host] path of the RCCL library the halo exchange calls into (loaded on first use; "" if it
 * cannot be loaded).  dlopen by SONAME: a copy the process has already mapped wins. */
const char* rf_comm_library(void);
/* one process per GPU: rank `rank` of `world` ranks on one node, neighbour halo
 * exchange over RCCL.  `id128` = the bytes produced by rf_comm_unique_id on rank 0 */
rf_status rf_ctx_create_dist(int device, int rank, int world, const void* id128, rf_ctx** out);
void      rf_ctx_destroy(rf_ctx* ctx);
/* vkDeviceWaitIdle (render.rs:123, pipeline_graph.rs:610) */
rf_status rf_ctx_synchronize(rf_ctx* ctx);
int       rf_ctx_rank(const rf_ctx* ctx);
int       rf_ctx_world(const rf_ctx* ctx);
/* device name, e.g. "gfx950" */
const char* rf_ctx_device_arch(const rf_ctx* ctx);

/* ------------------------------------------------------------------------- */
/* Graph  (Render::create_graph src/render.rs:80-98)                           */
/* ------------------------------------------------------------------------- */
/* synthesize_config + PipelineGraph::new (pipeline_graph.rs:499-592) + per-frame
 * image allocation (pipeline_graph.rs:133-323) + initialize_ubos from the
 * config's instance parameters (render.rs:167-210) */
rf_status rf_graph_create(rf_ctx* ctx, const rf_config* cfg, const rf_graph_options* opt, rf_graph** out);
/* Drop for PipelineGraph (pipeline_graph.rs:605-620); hot reload = destroy + create
 * (recreate_graph render.rs:121-136) */
void      rf_graph_destroy(rf_graph* g);
/* [host view] the plan the graph was built from (owned by the graph) */
const rf_plan* rf_graph_plan(const rf_graph* g);
/* [host] what rf_graph_create has to say about a graph it ACCEPTED ("" = nothing): "catalogue-only fusion: <why>" when a fused
 *        chain could not be compiled and the graph was planned again in pieces, "... spills N bytes per lane" for a user stage
 *        whose apply() needs scratch memory (it runs -- the reference runs every shader that compiles, src/vulkan/shader.rs:29-93 --
 *        only slower).  The reference's counterpart is a warnln! beside a graph that keeps running (src/render.rs:121-136).
 *        Owned by the graph. */
const char* rf_graph_note(const rf_graph* g);
/* rows [y0,y1) of the frame this rank holds (whole frame when world == 1) */
rf_status rf_graph_strip(const rf_graph* g, int* y0, int* y1);

/* initialize_ubos / update_ubos (render.rs:167-223): writes one uniform member of
 * node `node`.  Unknown name -> RF_WARN_UNKNOWN_PARAM (nothing written).  A conv2d node's
 * `sigma` regenerates its default weights (replacing any set with rf_graph_set_weights). */
rf_status rf_graph_set_param(rf_graph* g, const char* node, const char* name,
                             rf_param_type type, const void* value);
/* K x K weights, row-major f32 with K = the node's resolved ksize: of a conv2d node that has its own, or of a
 * conv2d_weights node (the contents of the ConvWeights storage buffer it writes).  A conv2d fed through a buffer edge
 * has none of its own: RF_ERR_INVALID.  (No reference counterpart: SSBO contents are written by other nodes there.) */
rf_status rf_graph_set_weights(rf_graph* g, const char* node, const float* weights, int count);
/* update_ubos (render.rs:212-223): every member whose name ends in `_rf_time` */
rf_status rf_graph_set_time(rf_graph* g, float seconds);

/* record_initial_image_load (render.rs:264-313): RGBA8 sRGB rows of the rank's strip
 * (strip_rows x width) -> sRGB decode -> the graph's linear input image */
rf_status rf_graph_upload_srgb8(rf_graph* g, const uint8_t* rgba, size_t row_stride);
/* raw texels of the graph format (u8x4 or f32x4), no colour conversion */
rf_status rf_graph_upload_raw(rf_graph* g, const void* texels, size_t row_stride);
/* on-device synthetic input: hash32(seed, y*W+x, c) */
rf_status rf_graph_fill_synthetic(rf_graph* g, uint32_t seed);
/* ramps + impulse at (W/2, H/2) */
rf_status rf_graph_fill_structured(rf_graph* g);

/* Render::record + submit (render.rs:359-404, :441-495) ->
 * command::execute_pipeline_graph (command.rs:166-242).  Asynchronous. */
rf_status rf_graph_execute(rf_graph* g, int frame_slot);
/* wait_for_frame_fence (render.rs:328-337) */
rf_status rf_graph_wait(rf_graph* g, int frame_slot);

/* write_output_to_buffer (render.rs:406-433): linear -> sRGB8 of the rank's strip */
rf_status rf_graph_download_srgb8(rf_graph* g, int frame_slot, uint8_t* rgba, size_t row_stride);
rf_status rf_graph_download_raw(rf_graph* g, int frame_slot, void* texels, size_t row_stride);
/* rows [y0, y1) (local to the rank's strip) of the output image, raw texels: what a caller that
 * only needs part of a very large frame copies back (a 16384^2 rgba32f frame is 4 GiB) */
rf_status rf_graph_download_rows(rf_graph* g, int frame_slot, int y0, int y1, void* texels, size_t row_stride);
/* raw texels of any allocated image (debug / tests), by resource name */
rf_status rf_graph_download_image(rf_graph* g, int frame_slot, const char* resource,
                                  void* texels, size_t row_stride);

/* last_frame_gpu_times (render.rs:521-523, vkutils.rs:104-134): per-launch GPU
 * milliseconds of the slot's last completed frame, name-sorted.  Needs
 * RF_GRAPH_TIMERS.  *n in = capacity, out = count. */
rf_status rf_graph_node_times(rf_graph* g, int frame_slot, const char** names, float* ms, int* n);
/* the same as the reference's status string "name: 0.123ms, name2: ..." */
rf_status rf_graph_times_string(rf_graph* g, int frame_slot, char* buf, size_t cap);

/* ------------------------------------------------------------------------- */
/* Measurement helpers (bench.py): HIP-event timing on the graph's own stream  */
/* ------------------------------------------------------------------------- */
/* runs `iters` frames back to back on slot 0 and returns the total elapsed GPU
 * milliseconds between a hipEvent recorded before the first and after the last */
rf_status rf_graph_time_frames(rf_graph* g, int iters, float* total_ms);
/* the same with frame i on frame slot i % num_frames (its own input/output images), all slots submitted to slot 0's
 * queue -- the reference's frames in flight: images per frame, ONE queue (src/main.rs:164-170, src/vulkan/core.rs:123).
 * With num_frames x (images of a slot) beyond the 256 MiB Infinity Cache this is the cache-cold rate. */
rf_status rf_graph_time_frames_rotating(rf_graph* g, int iters, float* total_ms);
/* same, but one launch only (index into rf_plan_launch_label): average ms */
rf_status rf_graph_time_launch(rf_graph* g, int launch, int iters, float* avg_ms);
/* `iters` whole frames on slot 0 with a hipEvent pair around every launch, recorded on
 * the stream that launch runs on: avg_ms[k] = average duration of launch k (execution
 * order); n = capacity of avg_ms (>= rf_plan_num_launches) */
rf_status rf_graph_time_launches(rf_graph* g, int iters, float* avg_ms, int n);
/* `iters` whole frames on slot 0, a hipEvent pair around EACH frame: ms_each[i] = GPU
 * milliseconds of frame i (SURVEY.md 8d asks for median and min next to the mean; the
 * marker packets cost ~2 us per frame, so the mean of these is above rf_graph_time_frames) */
rf_status rf_graph_time_each_frame(rf_graph* g, int iters, float* ms_each);
/* one-rank RCCL round trip (communicator of world 1, grouped send+recv to self of
 * `bytes` bytes on `device`): librccl loads and is called with the right ABI */
rf_status rf_comm_selftest(int device, size_t bytes);
/* copy of `bytes` device bytes (rounded down to whole 7680-texel rgba32f rows), `iters` times, by the library's own
 * passthrough launch -- the stream kernel with no arithmetic and no halo: achieved GB/s (read + written), what a launch of
 * this design can reach on this box beyond the Infinity Cache.  (Until round 4 this was a float4 grid-stride loop that read
 * 5.1-5.3 TB/s where the passthrough launch itself reaches 6.0: a yardstick below what it measures.) */
rf_status rf_ctx_copy_bandwidth(rf_ctx* ctx, size_t bytes, int iters, float* gbps);

#ifdef __cplusplus
}
#endif
#endif
