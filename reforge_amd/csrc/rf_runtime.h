// rf_runtime.h -- device runtime behind the C ABI: context, per-frame images, the
// stream/event executor and the row-strip halo exchange.
//
// Counterparts in the reference: VkCore (src/vulkan/core.rs), Frame
// (src/vulkan/frame.rs), PipelineGraph / PipelineGraphFrame
// (src/vulkan/pipeline_graph.rs:24-57,:133-323), command::execute_pipeline_graph
// (src/vulkan/command.rs:166-242), GpuTimer (src/vulkan/vkutils.rs:47-135) and the
// upload/download halves of Render (src/render.rs:264-313,:406-433).
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/rfhip.h"
#include "rf_config.h"
#include "rf_kernels.h"
#include "rf_plan.h"

namespace rf {

// thread-local error text behind rf_last_error()
void set_error(const std::string& msg);
const char* last_error();

struct RcclApi;   // lazily dlopen'ed librccl (only a multi-rank context needs it)

}  // namespace rf

struct rf_config {
    rf::Config cfg;
    std::vector<std::string> node_names;   // name-sorted
};

struct rf_plan {
    rf::Plan plan;
    std::vector<rf::LaunchDesc> launches;                       // execution order; empty if launch_error is set
    std::string launch_error;
    // fills launches / aliases from `plan`
    void index();
    std::vector<std::pair<std::string, std::string>> aliases;   // key-sorted
};

struct rf_ctx {
    int device = 0;
    int rank = 0, world = 1;
    std::string arch;
    hipStream_t util_stream = nullptr;     // uploads, fills, probes
    float* d_tables = nullptr;             // sRGB eotf[256] ++ thr[255]
    void* comm = nullptr;                  // ncclComm_t
    const rf::RcclApi* rccl = nullptr;
};

namespace rf {

struct DeviceImage {
    void* alloc = nullptr;   // hipMalloc'd block: (ghost + rows + ghost) * pitch
    char* base = nullptr;    // address of local row 0
    size_t pitch = 0;
    Image view() const { return Image{base, pitch}; }
};

// one kernel launch of the frame (a node, or a fused chain of nodes): the planner's
// description plus the resolved device ops
struct Launch : LaunchDesc {
    std::vector<Op> ops;
};

struct FrameSlot {
    std::map<std::string, DeviceImage> images;
    hipStream_t stream = nullptr;            // the frame's queue (Frame.cmd_buffer, frame.rs:10-18)
    std::vector<hipStream_t> aux;            // side streams for multi-node layers
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> join;
    hipStream_t comm = nullptr;              // halo exchange runs here while interior rows compute
    hipEvent_t src_ready = nullptr, halo_ready = nullptr;

    std::vector<hipEvent_t> t0, t1;          // GpuTimer query pairs, one per launch
    bool timed_once = false;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
};

}  // namespace rf

struct rf_graph {
    rf_ctx* ctx = nullptr;
    rf_graph_options opt{};
    rf_plan plan;                      // owned view handed out by rf_graph_plan
    int strip_y0 = 0, strip_y1 = 0;    // global rows this rank owns
    int ghost = 0;                     // ghost rows allocated above and below every image
    std::vector<rf::Launch> launches;  // execution order
    std::vector<rf::FrameSlot> frames;
    std::map<std::string, float*> dev_weights;   // owned: the K x K weights of conv2d nodes that take none through a buffer edge
    std::map<std::string, float*> dev_buffers;   // owned: storage buffers by allocated name (PipelineGraphFrame buffers, pipeline_graph.rs:249-260)
    std::map<std::string, float*> weights_of;    // conv2d node -> the weights it reads (its own, or the storage buffer wired to it)
    std::map<std::string, std::vector<float*>> written_by;   // conv2d_weights node -> the storage buffers it writes
    std::string input_image;           // allocated name of rf:file-input ("" if the graph has none)
    std::string output_image;          // allocated name rf:final-output resolves to
    int need_input = 0;                // ghost rows of the input the frame reads
    rf::StreamTuning tune;
    bool nt_stores = true;             // RF_NT_STORE=0: never pick the non-temporal-store kernels (A/B measurements)
    bool force_split = false;          // RF_FORCE_SPLIT=1: interior/boundary split without an exchange (tests)
    bool sync_launches = false;        // RF_SYNC_LAUNCHES=1: host-synchronise after every launch (debugging aid)
    bool concurrent_layers = false;    // RF_CONCURRENT_LAYERS=1: the launches of a hazard-free layer run on side streams (slower, measured)
    uint8_t* d_staging = nullptr;      // RGBA8 staging rows (render.rs:552-564)
    size_t staging_bytes = 0;
    std::vector<std::string> time_names;   // scratch for rf_graph_node_times
    std::string jit_note;                  // why the graph fell back to catalogue-only fusion ("" if it did not)
    bool exchanged_once = false;           // the first halo exchange of THIS graph is waited for with a deadline
    // A storage buffer a user node FILLS on the device (RF_BUFFER_OUT) is one per graph, where the reference has one per frame
    // slot: frames of such a graph on different slots are ordered one behind the other (submit_frame) instead of overlapping
    bool glsl_no_window = false;           // RF_EXEC_GLSL_NO_WINDOW: .comp stencils run on their generic kernel only
    std::set<std::string> glsl_window_ok;  // launches (labels) of .comp stencils whose window kernel agreed with the generic one at graph creation
    bool fills_buffers = false;
    hipEvent_t buffers_idle = nullptr;
    bool buffers_idle_set = false;
};
