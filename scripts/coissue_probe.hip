// coissue_probe.hip -- exploration only (not part of the product): can a wave that issues f32 MFMAs ALSO issue packed f32
// FMAs in their shadow, and what does the chip sustain when it does?  (The guide: a 16x16x4 f32 MFMA occupies the matrix
// pipe for 32 cycles but holds the vector issue port for 8.)  Register operands only -- no LDS, no memory -- so the number is
// the ceiling any MFMA + VALU hybrid of the 31x31 convolution could reach.  For each mix of P packed FMAs per MFMA it prints
// the combined multiply-add rate, split by pipe, and the in-kernel clock (s_memtime / s_memrealtime, guide: DVFS give-back 6).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/coissue_probe.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

constexpr int ITERS = 2048;
constexpr int NM = 4;      // independent MFMA accumulators (40-cycle dependent latency vs 32-cycle issue)
constexpr int NV = 12;     // independent packed-FMA accumulators

// P packed FMAs after every MFMA; M = 0: VALU only; P = 0: MFMA only
template <int M, int P>
__global__ __launch_bounds__(256) void mix_kernel(float* out, unsigned long long* clk, float seed)
{
    f32x4 macc[NM];
    v2f vacc[NV], x[4];
    v2f w = {seed, seed * 0.5f};
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
    for (int i = 0; i < NM; ++i) macc[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    for (int i = 0; i < NV; ++i) vacc[i] = v2f{(float)i, (float)threadIdx.x};
    for (int i = 0; i < 4; ++i) x[i] = v2f{seed + i, seed - i};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < (M ? NM : 1); ++m) {
            if (M) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(macc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int i = (m * P + p) % NV;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vacc[i]) : "v"(w), "v"(x[i & 3]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NM; ++i) s += macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3];
    for (int i = 0; i < NV; ++i) s += vacc[i].x + vacc[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int M, int P>
static int run(int waves_per_simd)
{
    const int threads = 256, blocks = 256 * waves_per_simd;
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * sizeof(float)));
    CHECK(hipMalloc(&clk, (size_t)blocks * 2 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL((mix_kernel<M, P>), dim3(blocks), dim3(threads), 0, 0, out, clk, 1.0f);   // settle the clock
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix_kernel<M, P>), dim3(blocks), dim3(threads), 0, 0, out, clk, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    unsigned long long h[2 * 64];
    CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int i = 0; i < 64; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;     // memrealtime ticks at 100 MHz
    ghz /= 64;
    const double waves = (double)blocks * threads / 64;
    const int per_iter_m = M ? NM : 0, per_iter_v = (M ? NM : 1) * P;
    const double mfma_tf = waves * ITERS * per_iter_m * (16.0 * 16 * 4 * 2) / ms / 1e9;
    const double valu_tf = waves * ITERS * per_iter_v * (64.0 * 2 * 2) / ms / 1e9;
    printf("waves/SIMD %d  MFMA %s + %d pk_fma each : %7.3f ms  MFMA %6.1f TF + VALU %6.1f TF = %6.1f TF   clock %.2f GHz\n", waves_per_simd, M ? "yes" : "no ", P, ms, mfma_tf,
           valu_tf, mfma_tf + valu_tf, ghz);
    CHECK(hipFree(out));
    CHECK(hipFree(clk));
    return 0;
}

int main()
{
    for (int w : {1, 2}) {
        if (run<0, 8>(w)) return 1;     // VALU only
        if (run<1, 0>(w)) return 1;     // MFMA only
        if (run<1, 2>(w)) return 1;
        if (run<1, 4>(w)) return 1;
        if (run<1, 6>(w)) return 1;
        if (run<1, 8>(w)) return 1;
    }
    return 0;
}
