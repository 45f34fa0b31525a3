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

// RADIUS > 0: the workgroup's TILE.  64 x TH output texels per workgroup (256 threads = 64 columns x 4 thread rows, TY = TH / 4
// outputs each, stacked vertically); the (64 + 2R) x (TH + 2R) texels they read are staged in LDS once -- raw, as the image holds
// them; clamp-to-edge is resolved while the tile is filled -- and Window::at is an LDS read at a constant offset.  Sized so that
// the tiles of all inputs fit the 64 KiB a workgroup may declare; a node too large for that (radius 15 with three rgba32f
// inputs) keeps reading its windows where they lie.  Shared with the host, which sizes the grid.
struct UserTile {
    int th, ty;          // output rows per workgroup, per thread
    bool lds;            // the tiles fit: the LDS kernel
};
constexpr int kUserRegWindowRadius = 2;      // up to this radius a thread ALSO keeps its window in registers and slides it down its outputs
inline __host__ __device__ constexpr UserTile user_tile(int bpp, int radius, int n_inputs)
{
    for (int ty = radius <= kUserRegWindowRadius ? 8 : (radius <= 8 ? 4 : 2); ty >= 1; --ty) {
        const long bytes = (long)n_inputs * (64 + 2 * radius) * (4 * ty + 2 * radius) * bpp;
        if (bytes <= 64 * 1024) return UserTile{4 * ty, ty, true};
    }
    return UserTile{4, 1, false};
}

#ifdef __HIPCC_RTC__
// RADIUS > 0: what apply() sees of an input image -- the texel's neighbourhood, clamp-to-edge like every stencil of the library.
// Backed by the workgroup's LDS tile (`lds` = the LDS byte address of the texel itself, `lpitch` the tile's row pitch: dx and dy
// are constants of the caller's unrolled loops, so a tap is one ds_read at an immediate offset; the outputs a thread stacks
// vertically share most of their taps, which the compiler reads once), or, for a node whose tiles do not fit, by the image where
// it lies (L1 / L2 serve the re-use).  `bpp` and the backing are constants of the instantiation: the branches fold away.
struct Window {
    const char* base;
    unsigned long long pitch;
    int x, y, W, row_lo, row_hi, bpp;
    unsigned lds, lpitch;
    int rr;              // > 0: `w` holds a (2 rr + 1)^2 copy of the neighbourhood in registers (small radii)
    f4 w[(2 * kUserRegWindowRadius + 1) * (2 * kUserRegWindowRadius + 1)];
    RF_DEV f4 at(int dx, int dy) const
    {
        // small radii: the register copy the thread slides down its stack of outputs (2r+1 LDS reads per output instead of
        // (2r+1)^2).  The offsets of a caller's unrolled loops are constants and pick registers; offsets only the run knows make
        // the compiler index the copy in memory -- correct, slow: such a stage is better written with RADIUS >= 3 taps it needs.
        // (indexed about the centre of the 5 x 5 storage whatever the radius, with constants only: the copy must become registers
        // under every compiler build a process may get -- PyTorch bundles an older one -- and an index through `rr` did not)
        if (rr > 0) return w[(dy + kUserRegWindowRadius) * (2 * kUserRegWindowRadius + 1) + dx + kUserRegWindowRadius];
        if (lds != 0u) {
            const unsigned a = lds + (unsigned)(dy * (int)lpitch + dx * bpp);
            if (bpp == 4) return PxU8::decode(*reinterpret_cast<const __attribute__((address_space(3))) unsigned*>(a));
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = *reinterpret_cast<const __attribute__((address_space(3))) v4f*>(a);
            return make_float4(t.x, t.y, t.z, t.w);
        }
        int xx = x + dx, yy = y + dy;
        xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        yy = yy < row_lo ? row_lo : (yy > row_hi ? row_hi : yy);
        const char* row = base + (long long)yy * (long long)pitch;
        return bpp == 4 ? PxU8::decode(PxU8::load(row, (unsigned)xx * 4u)) : PxF32::decode(PxF32::load(row, (unsigned)xx * 16u));
    }
};

template <class Px, class U>
__global__ __launch_bounds__(256, 2) void user_node_kernel(UserNodeArgs A)
{
    typename U::P p;
    __builtin_memcpy(&p, A.params, sizeof(p));
    if constexpr (U::R > 0 && user_tile(Px::BPP, U::R, U::NI).lds) {
        constexpr UserTile kT = user_tile(Px::BPP, U::R, U::NI);
        constexpr int R = U::R, RW = 64 + 2 * R, RH = kT.th + 2 * R;
        __shared__ __attribute__((aligned(16))) typename Px::Raw tile[U::NI][RH][RW];
        // XCD-aware order (as the stream kernels'): workgroups are dealt round-robin over the 8 XCDs, so every XCD takes a
        // contiguous range of tiles in raster order -- tiles that share halo columns and rows then share an L2
        const int per_xcd = (int)gridDim.x >> 3;
        const int q = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
        const int tiles_y = (A.y1 - A.y0 + kT.th - 1) / kT.th;
        if (q >= A.grid_x * tiles_y) return;                                  // workgroup-uniform: in front of the barrier
        const int tx0 = (q % A.grid_x) * 64, ty0 = A.y0 + (q / A.grid_x) * kT.th;
        for (int i = (int)threadIdx.x; i < RW * RH; i += 256) {
            const int r = i / RW, c = i - r * RW;
            int gx = tx0 - R + c, gy = ty0 - R + r;
            gx = gx < 0 ? 0 : (gx > A.W - 1 ? A.W - 1 : gx);
            gy = gy < A.row_lo ? A.row_lo : (gy > A.row_hi ? A.row_hi : gy);
#pragma unroll
            for (int k = 0; k < U::NI; ++k)
                tile[k][r][c] = Px::load(A.src[k] + (long long)gy * (long long)A.src_pitch[k], (unsigned)gx * (unsigned)Px::BPP);
        }
        __syncthreads();
        const int lx = (int)(threadIdx.x & 63), ly = (int)(threadIdx.x >> 6) * kT.ty;
        const int x = tx0 + lx;
        const unsigned tile0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(&tile[0][0][0]);
        const unsigned lpitch = (unsigned)(RW * (int)sizeof(typename Px::Raw));
        const bool col_ok = x < A.W;
        const unsigned xoff = (unsigned)(col_ok ? x : 0) * (unsigned)Px::BPP;
        auto lds_of = [&](int k, int row) { return tile0 + (unsigned)(((k * RH + row + R) * RW + lx + R) * (int)sizeof(typename Px::Raw)); };
        auto emit = [&](int j, const f4 (&out)[U::NO]) {
            const int y = ty0 + ly + j;
            if (col_ok && y < A.y1) {
#pragma unroll
                for (int o = 0; o < U::NO; ++o)
                    if (A.dst[o]) Px::store(A.dst[o] + (long long)y * (long long)A.dst_pitch[o], xoff, out[o]);
            }
        };
        // every output of the thread is computed -- texels beyond the frame's edges are clamped copies in the tile -- and only
        // the stores are masked
        if constexpr (R <= kUserRegWindowRadius) {
            // the neighbourhood in registers, slid down the thread's stack of outputs: a new bottom row per output
            constexpr int C = kUserRegWindowRadius, S = 2 * C + 1;      // storage: S x S about its centre (C, C); rows / columns -R .. R are in use
            Window in[U::NI];
#pragma unroll
            for (int k = 0; k < U::NI; ++k) {
                in[k].base = nullptr; in[k].pitch = 0ull; in[k].x = x; in[k].W = A.W; in[k].row_lo = A.row_lo; in[k].row_hi = A.row_hi;
                in[k].bpp = Px::BPP; in[k].lpitch = lpitch; in[k].rr = R;
#pragma unroll
                for (int r = -R + 1; r <= R; ++r)          // the window of output -1: its rows -R+1 .. R are the rows -R .. R-1 of output 0
#pragma unroll
                    for (int c = -R; c <= R; ++c)
                        in[k].w[(r + C) * S + c + C] = Px::decode(*reinterpret_cast<const __attribute__((address_space(3))) typename Px::Raw*>(
                            lds_of(k, ly + r - 1) + (unsigned)(c * (int)sizeof(typename Px::Raw))));
            }
#pragma unroll
            for (int j = 0; j < kT.ty; ++j) {
#pragma unroll
                for (int k = 0; k < U::NI; ++k) {
#pragma unroll
                    for (int r = -R; r < R; ++r)
#pragma unroll
                        for (int c = -R; c <= R; ++c) in[k].w[(r + C) * S + c + C] = in[k].w[(r + 1 + C) * S + c + C];      // (renaming: the loops are unrolled)
#pragma unroll
                    for (int c = -R; c <= R; ++c)
                        in[k].w[(R + C) * S + c + C] = Px::decode(*reinterpret_cast<const __attribute__((address_space(3))) typename Px::Raw*>(
                            lds_of(k, ly + j + R) + (unsigned)(c * (int)sizeof(typename Px::Raw))));
                    in[k].y = ty0 + ly + j;
                    in[k].lds = lds_of(k, ly + j);
                }
                f4 out[U::NO];
#pragma unroll
                for (int o = 0; o < U::NO; ++o) out[o] = f4_zero();
                U::node(p, in, out, A.buf_in);
                emit(j, out);
                __builtin_amdgcn_sched_barrier(0);      // one output at a time: the rows of the next are not fetched before this one is done
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < kT.ty; ++j) {
                Window in[U::NI];
#pragma unroll
                for (int k = 0; k < U::NI; ++k)
                    in[k] = Window{nullptr, 0ull, x, ty0 + ly + j, A.W, A.row_lo, A.row_hi, Px::BPP, lds_of(k, ly + j), lpitch, 0, {}};
                f4 out[U::NO];
#pragma unroll
                for (int o = 0; o < U::NO; ++o) out[o] = f4_zero();
                U::node(p, in, out, A.buf_in);
                emit(j, out);
            }
        }
        return;
    } else {
    const unsigned bx = blockIdx.x % (unsigned)A.grid_x, by = blockIdx.x / (unsigned)A.grid_x, gy = gridDim.x / (unsigned)A.grid_x;
    const int x = (int)(bx * 256u + threadIdx.x);
    if (x >= A.W) return;
    const unsigned xoff = (unsigned)x * (unsigned)Px::BPP;
    for (int y = A.y0 + (int)by; y < A.y1; y += (int)gy) {
        f4 out[U::NO];
#pragma unroll
        for (int o = 0; o < U::NO; ++o) out[o] = f4_zero();
        if constexpr (U::R > 0) {
            Window in[U::NI];
#pragma unroll
            for (int i = 0; i < U::NI; ++i) in[i] = Window{A.src[i], A.src_pitch[i], x, y, A.W, A.row_lo, A.row_hi, Px::BPP, 0u, 0u, 0, {}};
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
