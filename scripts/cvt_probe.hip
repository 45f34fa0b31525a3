// Exploration (GPU box): what v_cvt_pk_u8_f32 does with values outside [0, 255], with ties and with NaN -- can the
// clamp and the v_rndne in front of it (PxU8::pack) go?   hipcc --offload-arch=gfx950 -O2 scripts/cvt_probe.hip -o /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, unsigned* raw, unsigned* ref, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    raw[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
    float c = fminf(fmaxf(in[i], 0.0f), 255.0f);
    ref[i] = (unsigned)rintf(c);
}
int main()
{
    std::vector<float> v;
    for (int i = -8; i <= 1040; ++i) v.push_back(i * 0.25f);            // every quarter from -2 to 260: ties included
    const float extra[] = {-1e30f, 1e30f, INFINITY, -INFINITY, NAN, -0.0f, 254.5f, 255.49999f, 255.5f, 0.49999997f, 0.5f, 0.50000006f, 1.5f, 2.5f};
    for (float e : extra) v.push_back(e);
    for (int i = 0; i < 255; ++i) { float m = i + 0.5f; v.push_back(nextafterf(m, 0.f)); v.push_back(nextafterf(m, 1e9f)); }
    int n = (int)v.size();
    float* din; unsigned *draw, *dref;
    hipMalloc(&din, n * 4); hipMalloc(&draw, n * 4); hipMalloc(&dref, n * 4);
    hipMemcpy(din, v.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(din, draw, dref, n);
    std::vector<unsigned> raw(n), ref(n);
    hipMemcpy(raw.data(), draw, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ref.data(), dref, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (raw[i] != ref[i] && !(std::isnan(v[i]))) { if (bad++ < 12) printf("x=%.9g cvt=%u clamp+rint=%u\n", v[i], raw[i], ref[i]); }
    printf("%d values, %d differ from clamp+round-to-nearest-even; NaN -> %u\n", n, bad, raw[1049 + 4]);
    return 0;
}
