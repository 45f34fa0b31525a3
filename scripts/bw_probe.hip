// bw_probe.hip -- exploration only (not part of the product): what streaming rate does this
// MI355X box actually sustain, and with which access shape?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float4 f4;

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

__global__ __launch_bounds__(256) void copy_gs(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

__global__ __launch_bounds__(256) void copy_gs_nt(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 v;
        v.x = __builtin_nontemporal_load(&s[i].x);
        v.y = __builtin_nontemporal_load(&s[i].y);
        v.z = __builtin_nontemporal_load(&s[i].z);
        v.w = __builtin_nontemporal_load(&s[i].w);
        __builtin_nontemporal_store(v.x, &d[i].x);
        __builtin_nontemporal_store(v.y, &d[i].y);
        __builtin_nontemporal_store(v.z, &d[i].z);
        __builtin_nontemporal_store(v.w, &d[i].w);
    }
}

// each thread copies U consecutive-in-stride elements with all loads issued first
template <int U>
__global__ __launch_bounds__(256) void copy_unroll(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) d[i + k * stride] = v[k];
    }
    for (; i < n; i += stride) d[i] = s[i];
}

// one block copies one contiguous chunk (block-contiguous instead of grid-stride)
template <int U>
__global__ __launch_bounds__(256) void copy_chunk(const f4* __restrict__ s, f4* __restrict__ d, size_t n, size_t per_block)
{
    size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block;
    if (b1 > n) b1 = n;
    size_t i = b0 + threadIdx.x;
    for (; i + (U - 1) * 256 < b1; i += U * 256) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; ++k) d[i + k * 256] = v[k];
    }
    for (; i < b1; i += 256) d[i] = s[i];
}

__global__ __launch_bounds__(256) void read_only(const f4* __restrict__ s, float* out, size_t n)
{
    f4 a = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 v = s[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

__global__ __launch_bounds__(256) void write_only(f4* __restrict__ d, size_t n)
{
    f4 v = make_float4(1, 2, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = v;
}

// the product's access shape: a wave copies a 64-texel column strip down `rows` rows of a W-texel image
template <int PF>
__global__ __launch_bounds__(256) void copy_strips(const f4* __restrict__ s, f4* __restrict__ d, int W, int H, int rows_per_chunk)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + wave;
    const int x = strip * 64 + lane;
    if (strip * 64 >= W) return;
    const int y0 = blockIdx.y * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > H) y1 = H;
    const int xc = x < W ? x : W - 1;
    f4 ring[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (y0 + j < y1) ring[j] = s[(size_t)(y0 + j) * W + xc];
    for (int base = y0; base < y1; base += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            int y = base + j;
            if (y < y1) {
                f4 v = ring[j];
                if (y + PF < y1) ring[j] = s[(size_t)(y + PF) * W + xc];
                if (x < W) d[(size_t)y * W + x] = v;
            }
        }
    }
}

int main()
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)3840 * 2160 * 16, (size_t)512 << 20, (size_t)2048 << 20};
    for (size_t bytes : sizes) {
        f4 *a, *b;
        float* o;
        CHECK(hipMalloc(&a, bytes));
        CHECK(hipMalloc(&b, bytes));
        CHECK(hipMalloc(&o, 4));
        CHECK(hipMemset(a, 1, bytes));
        CHECK(hipMemset(b, 2, bytes));
        const size_t n = bytes / 16;
        printf("---- buffer %.1f MiB (copy moves 2x) ----\n", bytes / 1048576.0);
        auto timeit = [&](const char* name, auto launch, double moved) {
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0, 0);
            const int it = 20;
            for (int i = 0; i < it; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s %8.1f GB/s  (%.1f us)\n", name, moved * it / (ms * 1e-3) / 1e9, ms * 1e3 / it);
        };
        for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
            char nm[64];
            snprintf(nm, sizeof nm, "copy grid-stride blocks=%d", blocks);
            timeit(nm, [&] { hipLaunchKernelGGL(copy_gs, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        }
        timeit("copy grid-stride nt 4096", [&] { hipLaunchKernelGGL(copy_gs_nt, dim3(4096), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        timeit("copy unroll4 blocks=2048", [&] { hipLaunchKernelGGL(copy_unroll<4>, dim3(2048), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        timeit("copy unroll8 blocks=1024", [&] { hipLaunchKernelGGL(copy_unroll<8>, dim3(1024), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        timeit("copy unroll4 blocks=4096", [&] { hipLaunchKernelGGL(copy_unroll<4>, dim3(4096), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        for (int blocks : {2048, 8192, 32768}) {
            char nm[64];
            snprintf(nm, sizeof nm, "copy chunk u4 blocks=%d", blocks);
            size_t per = (n + blocks - 1) / blocks;
            timeit(nm, [&] { hipLaunchKernelGGL(copy_chunk<4>, dim3(blocks), dim3(256), 0, 0, a, b, n, per); }, 2.0 * bytes);
        }
        timeit("read only  blocks=4096", [&] { hipLaunchKernelGGL(read_only, dim3(4096), dim3(256), 0, 0, a, o, n); }, 1.0 * bytes);
        timeit("write only blocks=4096", [&] { hipLaunchKernelGGL(write_only, dim3(4096), dim3(256), 0, 0, b, n); }, 1.0 * bytes);
        if (bytes == sizes[0]) {
            const int W = 3840, H = 2160;
            for (int rpc : {34, 68, 135, 270}) {
                char nm[64];
                dim3 grid((W / 64 + 3) / 4, (H + rpc - 1) / rpc);
                snprintf(nm, sizeof nm, "strips PF4 rpc=%d", rpc);
                timeit(nm, [&] { hipLaunchKernelGGL(copy_strips<4>, grid, dim3(256), 0, 0, a, b, W, H, rpc); }, 2.0 * bytes);
                snprintf(nm, sizeof nm, "strips PF8 rpc=%d", rpc);
                timeit(nm, [&] { hipLaunchKernelGGL(copy_strips<8>, grid, dim3(256), 0, 0, a, b, W, H, rpc); }, 2.0 * bytes);
                snprintf(nm, sizeof nm, "strips PF16 rpc=%d", rpc);
                timeit(nm, [&] { hipLaunchKernelGGL(copy_strips<16>, grid, dim3(256), 0, 0, a, b, W, H, rpc); }, 2.0 * bytes);
            }
        }
        hipFree(a);
        hipFree(b);
        hipFree(o);
    }
    return 0;
}
