// rf_user_dev.h -- device code of a user NODE: a filter type that is a file ({shader_path}/{type}.stage.hip, rf_user.h) and
// declares its images (RF_INPUTS / RF_OUTPUTS).  Where the reference binds every image variable of {type}.comp to an allocated
// image (src/vulkan/pipeline_graph.rs:205-236) and dispatches the shader over the frame (src/vulkan/command.rs:166-194), this
// kernel loads one texel of every input image, calls the file's apply() and stores one texel to every wired output image.
// RADIUS 0, a point op: HBM-bound, (NI + NO) x W x H x bytes-per-pixel per launch, every byte moved once; one texel per lane, a wave
// reads 64 adjacent texels of a row (1 KiB rgba32f / 256 B rgba8 per image).  Compiled at rf_graph_create by hiprtc (rf_jit.cpp)
// as part of the same translation unit as rf_device.h; never built ahead of time (there is no user type ahead of time).
//
// The host (rf_graph.cpp) includes this file for UserNodeArgs only; the kernel exists under the run-time compiler alone.
#pragma once

#ifndef __HIPCC_RTC__
#include "rf_kernels.h"
#endif

namespace rf {

constexpr int kUserNodeImages = 4;
#ifndef __HIPCC_RTC__
static_assert(kUserNodeImages == kMaxUserImages, "rf_kernels.h and rf_user_dev.h disagree");
#endif

struct UserNodeArgs {
    const char* src[kUserNodeImages];
    char* dst[kUserNodeImages];                   // nullptr: the graph does not wire this output
    unsigned long long src_pitch[kUserNodeImages], dst_pitch[kUserNodeImages];
    const float* buf_in;                          // RF_BUFFER_IN: the storage buffer the graph wires to it (never null when declared)
    float* buf_out;                               // RF_BUFFER_OUT: nullptr if the graph does not wire it (then nothing is filled)
    int W, y0, y1;                                // output rows [y0, y1) of the strip
    int grid_x;                                   // workgroups along x (the launch is 1-D: blockIdx.x = by * grid_x + bx)
    int row_lo, row_hi;                           // RADIUS > 0: lowest / highest readable row of the inputs (clamp-to-edge bounds; ghost rows of a strip)
    unsigned char params[56];                     // the file's `struct Params`, laid out as the device compiler does
};
static_assert(sizeof(UserNodeArgs) == 4 * 8 * kUserNodeImages + 16 + 16 + 8 + 56, "UserNodeArgs is passed as a byte block");

#ifdef __HIPCC_RTC__
// RADIUS > 0: what apply() sees of an input image -- the texel's neighbourhood, read where it lies (L1 / L2 serve the re-use:
// a texel is asked for by up to (2R+1)^2 lanes of neighbouring rows and columns); clamp-to-edge like every stencil of the library.
// `bpp` is a constant of the instantiation: the format branch folds away.
struct Window {
    const char* base;
    unsigned long long pitch;
    int x, y, W, row_lo, row_hi, bpp;
    RF_DEV f4 at(int dx, int dy) const
    {
        int xx = x + dx, yy = y + dy;
        xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        yy = yy < row_lo ? row_lo : (yy > row_hi ? row_hi : yy);
        const char* row = base + (long long)yy * (long long)pitch;
        return bpp == 4 ? PxU8::decode(PxU8::load(row, (unsigned)xx * 4u)) : PxF32::decode(PxF32::load(row, (unsigned)xx * 16u));
    }
};

template <class Px, class U>
__global__ __launch_bounds__(256) void user_node_kernel(UserNodeArgs A)
{
    const unsigned bx = blockIdx.x % (unsigned)A.grid_x, by = blockIdx.x / (unsigned)A.grid_x, gy = gridDim.x / (unsigned)A.grid_x;
    const int x = (int)(bx * 256u + threadIdx.x);
    if (x >= A.W) return;
    typename U::P p;
    __builtin_memcpy(&p, A.params, sizeof(p));
    const unsigned xoff = (unsigned)x * (unsigned)Px::BPP;
    for (int y = A.y0 + (int)by; y < A.y1; y += (int)gy) {
        f4 out[U::NO];
#pragma unroll
        for (int o = 0; o < U::NO; ++o) out[o] = f4_zero();
        if constexpr (U::R > 0) {
            Window in[U::NI];
#pragma unroll
            for (int i = 0; i < U::NI; ++i) in[i] = Window{A.src[i], A.src_pitch[i], x, y, A.W, A.row_lo, A.row_hi, Px::BPP};
            U::node(p, in, out, A.buf_in);
        } else {
            f4 in[U::NI];
#pragma unroll
            for (int i = 0; i < U::NI; ++i) in[i] = Px::decode(Px::load(A.src[i] + (long long)y * (long long)A.src_pitch[i], xoff));
            U::node(p, in, out, A.buf_in);
        }
        // every load of this texel is done before its first store: an output written in place (same binding as an input) is safe
#pragma unroll
        for (int o = 0; o < U::NO; ++o)
            if (A.dst[o]) Px::store(A.dst[o] + (long long)y * (long long)A.dst_pitch[o], xoff, out[o]);
    }
}

// RF_BUFFER_OUT(Name, N): element i of the buffer = fill(params, i).  N <= 65536 floats: a handful of workgroups in front of
// the node's own kernel on the same stream, every frame (the reference's node would rewrite its block every dispatch too).
template <class U>
__global__ __launch_bounds__(256) void user_fill_kernel(UserNodeArgs A)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= U::FILL || !A.buf_out) return;
    typename U::P p;
    __builtin_memcpy(&p, A.params, sizeof(p));
    A.buf_out[i] = U::fill_at(p, i);
}
#endif

}  // namespace rf
