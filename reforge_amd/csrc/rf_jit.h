// rf_jit.h -- stream kernels compiled at graph creation (hiprtc): the counterpart of the reference compiling a
// node's GLSL when the graph is built (src/vulkan/shader.rs:29-93, pipeline.rs:45-88).  The ahead-of-time catalogue
// (rf_stream.hip) holds the kernels of single nodes and of the BASELINE chains; any other chain of fusable nodes
// gets its stream_kernel<> instantiation generated from the SAME device source (rf_stream_dev.h, embedded in the
// library) the first time a graph needs it.  libhiprtc is loaded on first use; without it nothing is compiled and
// the planner cuts such a chain into catalogue pieces.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "rf_kernels.h"

namespace rf {

struct JitKernel {
    hipFunction_t fn = nullptr;
    int texels = 1;
    int resident_workgroups = 512;    // workgroups of this kernel the chip holds at once (occupancy query)
    int vgprs = 0, scratch_bytes = 0;
    bool disabled = false;            // jit_forget: an optional variant that turned out to spill -- kept loaded, never looked up again
};

// libhiprtc can be loaded and RF_NO_JIT is not set (host only: no device needed to ask, nor to compile)
bool jit_available();
// compile (or fetch from the process / disk cache) and load stream_kernel<Px, pf, texels, stages...> on the current device
bool jit_compile(int fmt, int pf, int texels, const StageList& sl, int waves_per_block, std::string& err);
// the loaded kernel, nullptr if jit_compile has not produced it on the current device
const JitKernel* jit_lookup(int fmt, int pf, int texels, const StageList& sl);
// unregister a loaded kernel (an optional variant that turned out to spill): jit_lookup no longer returns it
void jit_forget(int fmt, int pf, int texels, const StageList& sl);
hipError_t jit_launch(const JitKernel& k, unsigned grid, unsigned block, void* args, size_t size, hipStream_t stream);
// path of the libhiprtc this process compiles with ("" if none): a process that imported PyTorch first gets PyTorch's copy
std::string jit_library();
// number of kernels compiled by this process (not served from a cache): measurement / tests
int jit_compile_count();
// [host] compile only: code object size, 0 + err on failure (CPU tests: no device needed)
size_t jit_compile_only(int fmt, int pf, int texels, const StageList& sl, int waves_per_block, std::string& err);
// user NODES (a stage file that declares its images, rf_user.h): user_node_kernel<Px, Stage> of rf_user_dev.h, launched with
// 256 threads per workgroup and a UserNodeArgs block
// wide: the node's images are 4 GiB or larger (a .comp node then addresses its texels with 64-bit pointers instead of buffer loads with a 32-bit offset)
bool jit_compile_user_node(int fmt, int user_id, std::string& err, bool wide = false);
const JitKernel* jit_lookup_user_node(int fmt, int user_id, bool wide = false);
const JitKernel* jit_lookup_user_fill(int user_id);
// a .comp stencil (UserStage::glsl_window): its window kernel, rf::user_node_kernel<Px, the generated stage> (compiled by jit_compile_user_node with the generic one)
const JitKernel* jit_lookup_glsl_window(int fmt, int user_id);      // user_fill_kernel<Stage> of a node that declares RF_BUFFER_OUT
size_t jit_compile_only_user_node(int fmt, int user_id, std::string& err);

}  // namespace rf
