// valu_probe.hip -- exploration only (not part of the product): what f32 multiply-add rate does
// the VALU of this MI355X sustain, per instruction form and waves per SIMD?  Build + run on the
// GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

constexpr int NACC = 16;
constexpr int ITERS = 4096;

// MODE 0: v_pk_fma_f32 acc, w (low half broadcast), x, acc      -- the conv kernel's form
// MODE 1: v_pk_fma_f32 acc, w, x, acc with plain pairs
// MODE 2: v_fma_f32
template <int MODE>
__global__ void fma_kernel(float* out, float seed)
{
    v2f acc[NACC], x[4];
    v2f w = {seed, seed * 0.5f};
    for (int i = 0; i < NACC; ++i) acc[i] = v2f{(float)i, (float)threadIdx.x};
    for (int i = 0; i < 4; ++i) x[i] = v2f{seed + i, seed - i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(w), "v"(x[i & 3]));
            } else if (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(x[i & 3]));
            } else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(w.x), "v"(x[i & 3].x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(w.y), "v"(x[i & 3].y));
            }
        }
    }
    v2f s = {0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE>
static int run(const char* name, int threads, int blocks_per_cu)
{
    float* out;
    const int blocks = 256 * blocks_per_cu;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fma_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fma_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double flops = (double)blocks * threads * ITERS * NACC * 4.0;   // 2 lanes x (mul + add)
    printf("%-28s threads/block %4d blocks/CU %d : %8.3f ms  %7.1f TFLOP/s\n", name, threads, blocks_per_cu, ms, flops / ms / 1e9);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    for (int threads : {256, 512, 1024}) {
        for (int bpc : {1, 2}) {
            if (run<0>("pk_fma w-broadcast (op_sel)", threads, bpc)) return 1;
            if (run<1>("pk_fma plain", threads, bpc)) return 1;
            if (run<2>("2 x v_fma_f32", threads, bpc)) return 1;
        }
    }
    return 0;
}
