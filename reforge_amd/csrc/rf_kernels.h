// rf_kernels.h -- host-side launch interface of the gfx950 kernels (rf_stream.hip, rf_conv.hip, rf_misc.hip).
//
// These replace the reference's GLSL compute shaders (shaders/*.comp, of which only
// passthrough.comp exists) and the per-node vkCmdDispatch of
// src/vulkan/command.rs:166-242.  Node arithmetic is specified in DESIGN.md
// "Node specifications"; the CPU restatement the parity tests check against is
// oracle/rf_oracle.c.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace rf {

constexpr int kFmtRGBA8   = 0;
constexpr int kFmtRGBA32F = 1;
constexpr int kPxF32Stream = 2;     // kernel selection only: rgba32f whose row stores carry the non-temporal hint (PxF32NT, rf_device.h)
constexpr int kMaxRadius  = 15;     // conv2d up to 31x31, gaussian radius up to 15
constexpr int kMaxFusedOps = 8;     // nodes one streaming launch may cover
constexpr int kMaxUserImages = 4;   // input images, and output images, of a user node (OP_USERN)

inline size_t bytes_per_pixel(int fmt) { return fmt == kFmtRGBA8 ? 4 : 16; }

// One node's device work.  `kind` selects the stage(s) the node contributes to a
// streaming pipeline.
enum OpKind : int {
    OP_PASSTHROUGH = 0,   // shaders/passthrough.comp:7-13
    OP_GAUSSIAN    = 1,   // separable, radius r: H taps then V taps on f32
    OP_GRADE       = 2,   // colour grade point op
    OP_SHARPEN     = 3,   // 3x3 cross
    OP_CONV2D      = 4,   // dense KxK (never fused; own tile kernel)
    OP_MIX         = 5,   // two-input blend a + mix*(b-a) (the "combination" node of
                          // pipeline_graph.rs:462-468's example; own kernel)
    // registry-level kinds (NodeParams::to_op turns them into one of the device ops above)
    OP_WEIGHTS     = 6,   // conv2d_weights: writes a ConvWeights storage buffer, passes its image through
    OP_PULSE       = 7,   // pulse: a colour grade whose slope follows `phase_rf_time`
    OP_SPLIT       = 9,   // split_luma: one input image, TWO output images (luma_image, chroma_image); own kernel, never fused
    OP_USER        = 8,   // a type that is a file, {shader_path}/{type}.stage.hip (rf_user.h): point op or 3x3 neighbourhood
    OP_USERN       = 10   // such a file that declares its images (RF_INPUTS / RF_OUTPUTS): up to 4 input and 4 output images, a point
                          // op with a kernel of its own (rf_user_dev.h), never fused
};
// node kinds that run a kernel of their own and never join a fused stream launch
inline bool own_kernel_kind(int kind) { return kind == OP_MIX || kind == OP_CONV2D || kind == OP_SPLIT || kind == OP_USERN; }

struct Op {
    int   kind = OP_PASSTHROUGH;
    int   radius = 0;                // gaussian radius / conv K/2
    float w[kMaxRadius + 1] = {};    // gaussian half kernel
    float slope = 0, offset = 0, saturation = 0;   // grade (slope doubles as the mix factor of OP_MIX)
    float wc = 1, ws = 0;            // sharpen centre / side weight
    const float* dev_weights = nullptr;   // conv2d: device pointer, [K][K]
    int   user_id = -1;              // OP_USER / OP_USERN: the stage (rf_user.h) ...
    unsigned char user_params[256] = {};  // ... and its Params block, laid out as the device compiler does (stage files: up to 56 bytes;
                                     // a .comp file: its uniform blocks, std140, up to 256)
    int   slot = 0;                  // inside a fused fork/join launch: 1 = node of the branch feeding input_image0, 2 = of the branch
                                     // feeding input_image1; 0 = before the fork, the join itself, after the join, or a plain chain
};

// A 2-D image (or a row strip of one with ghost rows): `base` addresses local row 0,
// rows outside [0, rows) may exist as ghost rows; `pitch` in bytes.
struct Image {
    void*  base = nullptr;
    size_t pitch = 0;
};

// Geometry of one launch.  Rows are in the strip's local coordinates.
struct Geom {
    int W = 0;          // frame width in pixels
    int row_lo = 0;     // lowest readable row (clamp-to-edge bound), may be negative (ghost rows)
    int row_hi = 0;     // highest readable row, inclusive
    int y0 = 0, y1 = 0; // output rows [y0, y1)
    bool nt_store = false;  // the image written is read by no launch of the frame (the graph's result): rgba32f stream kernels store it
                            // with the non-temporal hint, so that it does not displace input rows other workgroups still want from L2
    int yb0 = 0, yb1 = 0;   // a SECOND range of output rows in the same launch (stream launches only; empty by default): the two
                            // boundary slivers of a row strip in exchange mode run as ONE launch (run_launch, rf_graph.cpp)
};

// Tuning knobs (0 = heuristic).  Read once from the environment by rf_graph.
struct StreamTuning {
    int rows_per_chunk = 0;   // RF_ROWS_PER_CHUNK
    int walk = 0;             // chunk walk direction: 0 auto, 1 odd chunks bottom-up (halo rows shared through L2), 2 all top-down
    int texels_per_lane = 0;  // RF_TEXELS_PER_LANE: 0 auto, 1 or 2 (rgba32f stream kernels)
    int conv_path = 0;        // RF_CONV_PATH: 0 = register-blocked VALU kernel, 1 = 16x16 LDS tile, 2 = MFMA (K >= 9), 3 = VALU
};

// Row stages of a streaming launch (rf_stream_dev.h): the run-time description that selects -- or, for a list the
// ahead-of-time catalogue lacks, GENERATES -- the kernel, and lays out its parameter block.
enum StageKind : int { ST_NODE_END = 0, ST_HTAP = 1, ST_VTAP = 2, ST_GRADE = 3, ST_CROSS3 = 4, ST_DUP = 5, ST_MIX = 6, ST_USER = 7,
                       ST_DELAY = 8 };     // r rows of delay at the end of the fork/join branch with the smaller vertical radius (StDelay)
enum StageSlot : int { SLOT_PLAIN = 0, SLOT_SOLO = 1, SLOT_ON0 = 2, SLOT_ON1 = 3 };   // wrapper of a stage in a fork/join (pair) pipeline
struct StageList {
    static constexpr int kMax = 3 * kMaxFusedOps + 4;
    int n = 0;
    struct { int kind, r, slot, op, user; } st[kMax];   // op: index of the node (in the launch's op list) whose parameters the stage takes, -1 none; user: ST_USER's stage id
    bool pair() const { return n > 0 && st[0].slot != SLOT_PLAIN; }
    bool has_user() const { for (int i = 0; i < n; ++i) if (st[i].kind == ST_USER) return true; return false; }
    std::string key() const;          // "H2 V2 E G E C ": catalogue / cache key
    std::string type_list() const;    // "rf::StHTap<2>, rf::StVTap<2>, ...": the template arguments of stream_kernel
    int sum_rh() const, sum_rv() const, max_rv() const, taps() const;
    int vgpr_estimate(int texels) const;
};
constexpr size_t kMaxParamBytes = 2048;
bool ops_to_stages(const Op* ops, int n, StageList& out);
bool stream_in_catalogue(const StageList& sl);
// which stream kernel a launch of this stage list takes: the format, or kPxF32Stream (rgba32f, non-temporal row stores) when
// the launch's result is read by no launch (`nt_store`) and the pipeline is bound by the memory path, not by issue (rf_stream.hip)
int stream_kernel_code(int fmt, const StageList& sl, bool nt_store);
bool stream_jit_admissible(const StageList& sl);

// true if `ops[0..n)` can run as ONE streaming launch (a fused pipeline): the catalogue holds the kernel, or
// (allow_jit) it can be compiled when the graph is created
bool stream_supported(const Op* ops, int n, bool allow_jit);
// rf_graph_create: make the kernel of a fused launch available (compiles it if the catalogue lacks it); false + err if it cannot
// (*note: something worth telling the caller about a kernel that WAS accepted -- a single user stage that spills)
bool stream_prepare(int fmt, const Op* ops, int n, int W, int rows, const StreamTuning& tune, bool nt_store, std::string& err, std::string* note = nullptr);
// horizontal / vertical halo a fused pipeline reads beyond its output
int  ops_radius(const Op* ops, int n);

// Launch ops[0..n) as one kernel: dst = op[n-1](...op[0](src)).  src == dst is
// allowed only when every op is a point op.  Returns hipSuccess or the launch error.
hipError_t launch_ops(int fmt, const Op* ops, int n, Image src, Image dst, const Geom& g,
                      const StreamTuning& tune, hipStream_t stream);

// dst = a + mix*(b-a) per channel (fmaf(mix, b-a, a)); a == dst or b == dst allowed
hipError_t launch_mix(int fmt, Image a, Image b, Image dst, const Geom& g, float mix, hipStream_t stream);

// split_luma: src -> luma and / or chroma (an Image with a null base is not written); any of them may alias src (point op)
hipError_t launch_split_luma(int fmt, Image src, Image luma, Image chroma, const Geom& g, hipStream_t stream);

// synthetic / structured fills of rows [y_begin, y_end) (local), whose global row is y + y_global0
hipError_t launch_fill_synthetic(int fmt, Image dst, int W, int y_begin, int y_end, int y_global0,
                                 uint32_t seed, hipStream_t stream);
hipError_t launch_fill_structured(int fmt, Image dst, int W, int y_begin, int y_end, int y_global0,
                                  int Hfull, hipStream_t stream);

// sRGB boundary (src/render.rs:264-313, :406-433).  `tables` = device float[256+255]
// (EOTF table then encode thresholds).  `rgba` = device RGBA8 rows.
hipError_t launch_upload_srgb8(int fmt, const uint8_t* rgba, size_t stride, Image dst, int W, int rows,
                               const float* tables, hipStream_t stream);
hipError_t launch_download_srgb8(int fmt, Image src, uint8_t* rgba, size_t stride, int W, int rows,
                                 const float* tables, hipStream_t stream);

// host-side parameter derivation shared with nothing else (the oracle has its own)
void gaussian_weights(float sigma, int radius, float* w);          // w[0..radius]
void sharpen_weights(float amount, float* centre, float* side);
void default_conv_weights(int K, float sigma, float* w);           // [K][K]
void srgb_tables(float* eotf256, float* thr255);

}  // namespace rf
