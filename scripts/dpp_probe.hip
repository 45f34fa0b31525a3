// Exploration (GPU box): what the wave-wide DPP shifts of gfx9 (wave_shr:1 / wave_shl:1) do on gfx950 and what they cost:
//  1. semantics: which neighbour a lane receives, what lanes 0 / 63 get;
//  2. v_fmac_f32_dpp == fmaf(neighbour, w, acc) bit for bit (one rounding);
//  3. issue rate of v_fmac_f32_dpp / v_mov_b32_dpp against plain v_fmac_f32 and v_pk_fma_f32, 1..4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dpp_probe scripts/dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void sem_kernel(float* out, const float* in)
{
    const int l = threadIdx.x;
    const float v = in[l];
    float shr = -1.0f, shl = -1.0f, shr_b = -1.0f, shl_b = -1.0f;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(shr) : "v"(v));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(shl) : "v"(v));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(shr_b) : "v"(v));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(shl_b) : "v"(v));
    out[l] = shr; out[64 + l] = shl; out[128 + l] = shr_b; out[192 + l] = shl_b;
}

__global__ void fmac_kernel(float* out, const float* in, const float* acc0, float w)
{
    const int l = threadIdx.x;
    const float v = in[l];
    float a = acc0[l], b = acc0[l];
    const float wv = w;
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(v), "v"(wv));
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(v), "v"(wv));
    out[l] = a; out[64 + l] = b;
}

// MODE 0: 16 v_fmac_f32 (no dpp)   1: 16 v_fmac_f32_dpp wave_shr:1   2: 8 v_pk_fma_f32 (same flops as 16 fmac)
// MODE 3: 16 x (v_mov_b32_dpp + v_fmac_f32)   4: 16 v_fmac_f32_dpp row_shr:1
template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float* out, const float* in, int iters)
{
    const int l = threadIdx.x;
    float v = in[l & 63], w = in[64 + (l & 63)];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = in[128 + i];
    typedef float v2f __attribute__((ext_vector_type(2)));
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(v), "v"(w));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(v), "v"(w));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f acc = {a[i], a[i + 1]}, vv = {v, v}, ww = {w, w};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(vv), "v"(ww));
                a[i] = acc.x; a[i + 1] = acc.y;
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float s;
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(s) : "v"(v));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(w));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(v), "v"(w));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + l] = s;
}

template <int MODE> static int rate(const char* name, float* d_out, const float* d_in)
{
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;      // 256-thread blocks: one wave per SIMD each
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_in, 1000);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_in, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)blocks * 4 * iters * (MODE == 2 ? 8 : 16) * (MODE == 3 ? 2 : 1);
        const double per_simd_ns = ms * 1e6 / (wave_instr / 1024.0);      // ns per wave-instruction per SIMD
        std::printf("%-28s waves/SIMD %d: %.3f ms  %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)  %.1f Tflop/s-equivalent\n", name, wps, ms,
                    per_simd_ns, per_simd_ns * 2.4, (double)blocks * 4 * iters * 16 * 64 * 2 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}

int main()
{
    float *d_in, *d_out, *d_acc;
    CK(hipMalloc(&d_in, 4096)); CK(hipMalloc(&d_acc, 4096)); CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    std::vector<float> in(1024), acc(64), out(256);
    for (int i = 0; i < 1024; ++i) in[i] = 1.0f + (float)i;
    CK(hipMemcpy(d_in, in.data(), 4096, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, d_out, d_in);
    CK(hipMemcpy(out.data(), d_out, 1024, hipMemcpyDeviceToHost));
    const char* names[4] = {"wave_shr:1", "wave_shl:1", "wave_shr:1 bound_ctrl:0", "wave_shl:1 bound_ctrl:0"};
    for (int k = 0; k < 4; ++k) {
        std::printf("%s (in[l] = l+1, old = -1):", names[k]);
        for (int l : {0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 62, 63}) std::printf(" l%d=%g", l, out[64 * k + l]);
        std::printf("\n");
    }
    // fmac_dpp against fmaf, on awkward values
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xffffff) / 16777216.0f * 3.0f - 1.5f; };
    int bad = 0;
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 64; ++i) { in[i] = rnd(); acc[i] = rnd(); }
        const float w = rnd();
        CK(hipMemcpy(d_in, in.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_acc, acc.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(fmac_kernel, dim3(1), dim3(64), 0, 0, d_out, d_in, d_acc, w);
        CK(hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost));
        for (int l = 1; l < 63; ++l) {
            const float a = fmaf(in[l - 1], w, acc[l]), b = fmaf(in[l + 1], w, acc[l]);
            if (std::memcmp(&a, &out[l], 4) != 0 || std::memcmp(&b, &out[64 + l], 4) != 0) ++bad;
        }
    }
    std::printf("v_fmac_f32_dpp vs fmaf(neighbour, w, acc): %d mismatches over 200 x 62 lanes x 2 directions\n", bad);
    for (int i = 0; i < 1024; ++i) in[i] = 1e-3f * (float)(i % 7);
    CK(hipMemcpy(d_in, in.data(), 4096, hipMemcpyHostToDevice));
    if (rate<0>("v_fmac_f32", d_out, d_in)) return 1;
    if (rate<1>("v_fmac_f32_dpp wave_shr:1", d_out, d_in)) return 1;
    if (rate<4>("v_fmac_f32_dpp row_shr:1", d_out, d_in)) return 1;
    if (rate<2>("v_pk_fma_f32 (x8 = same flops)", d_out, d_in)) return 1;
    if (rate<3>("v_mov_b32_dpp + v_fmac_f32", d_out, d_in)) return 1;
    return 0;
}
