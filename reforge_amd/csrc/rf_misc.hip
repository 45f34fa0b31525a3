// rf_misc.hip -- the small kernels off the timed path (synthetic/structured fills, the sRGB
// upload/download blits of src/render.rs:264-313,:406-433, the two-input blend, the
// bandwidth-probe copy) and the host-side parameter derivation.
#include "rf_device.h"

#include <string.h>

namespace rf {

// ---------------------------------------------------------------------------------
// Synthetic inputs, identical to rfo_fill_* in the oracle
// ---------------------------------------------------------------------------------
RF_DEV uint32_t hash32(uint32_t seed, uint32_t idx, uint32_t c)
{
    uint32_t h = seed ^ ((idx * 4u + c) * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

template <class Px>
__global__ __launch_bounds__(256) void fill_synthetic_kernel(char* dst, size_t pitch, int W, int y_begin, int y_end,
                                                             int y_global0, uint32_t seed)
{
    const int x = (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= W) return;
    for (int y = y_begin + (int)blockIdx.y; y < y_end; y += (int)gridDim.y) {
        uint32_t idx = (uint32_t)(y + y_global0) * (uint32_t)W + (uint32_t)x;
        uint32_t u0 = hash32(seed, idx, 0), u1 = hash32(seed, idx, 1), u2 = hash32(seed, idx, 2), u3 = hash32(seed, idx, 3);
        char* row = dst + (ptrdiff_t)y * (ptrdiff_t)pitch;
        if constexpr (Px::QUANT) {
            *reinterpret_cast<unsigned*>(row + (size_t)x * 4) =
                (u0 >> 24) | ((u1 >> 24) << 8) | ((u2 >> 24) << 16) | ((u3 >> 24) << 24);
        } else {
            *reinterpret_cast<f4*>(row + (size_t)x * 16) =
                make_float4((float)(u0 >> 8) * 0x1p-24f, (float)(u1 >> 8) * 0x1p-24f,
                            (float)(u2 >> 8) * 0x1p-24f, (float)(u3 >> 8) * 0x1p-24f);
        }
    }
}

template <class Px>
__global__ __launch_bounds__(256) void fill_structured_kernel(char* dst, size_t pitch, int W, int y_begin, int y_end,
                                                              int y_global0, int Hfull)
{
    const int x = (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= W) return;
    for (int y = y_begin + (int)blockIdx.y; y < y_end; y += (int)gridDim.y) {
        int gy = y + y_global0;
        unsigned c0 = (unsigned)(x & 255), c1 = (unsigned)(gy & 255), c2 = (unsigned)((x + gy) & 255), c3 = 255u;
        if (x == W / 2 && gy == Hfull / 2) c0 = c1 = c2 = 255u;
        char* row = dst + (ptrdiff_t)y * (ptrdiff_t)pitch;
        if constexpr (Px::QUANT) {
            *reinterpret_cast<unsigned*>(row + (size_t)x * 4) = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
        } else {
            *reinterpret_cast<f4*>(row + (size_t)x * 16) =
                make_float4(unorm8_to_f32(c0), unorm8_to_f32(c1), unorm8_to_f32(c2), unorm8_to_f32(c3));
        }
    }
}

// ---------------------------------------------------------------------------------
// sRGB boundary (src/render.rs:264-313, :406-433).  tables = eotf[256] ++ thr[255]
// ---------------------------------------------------------------------------------
template <class Px>
__global__ __launch_bounds__(256) void upload_srgb8_kernel(const uint8_t* rgba, size_t stride, char* dst, size_t pitch,
                                                           int W, int rows, const float* __restrict__ tables)
{
    __shared__ float eotf[256];
    eotf[threadIdx.x] = tables[threadIdx.x];
    __syncthreads();
    const int x = (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= W) return;
    for (int y = (int)blockIdx.y; y < rows; y += (int)gridDim.y) {
        unsigned c = *reinterpret_cast<const unsigned*>(rgba + (size_t)y * stride + (size_t)x * 4);
        f4 v = make_float4(eotf[c & 255u], eotf[(c >> 8) & 255u], eotf[(c >> 16) & 255u], unorm8_to_f32(c >> 24));
        Px::store(dst + (ptrdiff_t)y * (ptrdiff_t)pitch, (unsigned)x * (unsigned)Px::BPP, v);
    }
}

RF_DEV unsigned srgb_encode(float v, const float* thr)
{
    // number of thresholds <= v (NaN -> 0): 8-step binary search over thr[0..254]
    int lo = 0, hi = 255;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        int mid = (lo + hi) >> 1;
        bool up = (lo < hi) && (thr[mid] <= v);
        bool dn = (lo < hi) && !up;
        lo = up ? mid + 1 : lo;
        hi = dn ? mid : hi;
    }
    return (unsigned)lo;
}

template <class Px>
__global__ __launch_bounds__(256) void download_srgb8_kernel(const char* src, size_t pitch, uint8_t* rgba, size_t stride,
                                                             int W, int rows, const float* __restrict__ tables)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = threadIdx.x < 255 ? tables[256 + threadIdx.x] : INFINITY;
    __syncthreads();
    const int x = (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= W) return;
    for (int y = (int)blockIdx.y; y < rows; y += (int)gridDim.y) {
        f4 v = Px::decode(Px::load(src + (ptrdiff_t)y * (ptrdiff_t)pitch, (unsigned)x * (unsigned)Px::BPP));
        unsigned c = srgb_encode(v.x, thr) | (srgb_encode(v.y, thr) << 8) | (srgb_encode(v.z, thr) << 16) |
                     (f32_to_unorm8(v.w) << 24);
        *reinterpret_cast<unsigned*>(rgba + (size_t)y * stride + (size_t)x * 4) = c;
    }
}

template <class Px>
__global__ __launch_bounds__(256) void mix_kernel(const char* a, size_t a_pitch, const char* b, size_t b_pitch, char* dst,
                                                  size_t dst_pitch, int W, int y0, int y1, float mix)
{
    const int x = (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= W) return;
    const unsigned xoff = (unsigned)x * (unsigned)Px::BPP;
    for (int y = y0 + (int)blockIdx.y; y < y1; y += (int)gridDim.y) {
        f4 va = Px::decode(Px::load(a + (ptrdiff_t)y * (ptrdiff_t)a_pitch, xoff));
        f4 vb = Px::decode(Px::load(b + (ptrdiff_t)y * (ptrdiff_t)b_pitch, xoff));
        f4 o = make_float4(fmaf(mix, vb.x - va.x, va.x), fmaf(mix, vb.y - va.y, va.y), fmaf(mix, vb.z - va.z, va.z),
                           fmaf(mix, vb.w - va.w, va.w));
        Px::store(dst + (ptrdiff_t)y * (ptrdiff_t)dst_pitch, xoff, o);
    }
}

// split_luma: ONE input image, TWO output images (a node with several output bindings: the reference allocates one image per
// output binding of every node, pipeline_graph.rs:205-224).  luma = fma(0.0722, b, fma(0.7152, g, 0.2126 r)) -- Rec.709, as
// colour_grade; luma_image = (luma, luma, luma, a); chroma_image = (fma(0.5, c - luma, 0.5) for c in r, g, b; a).  Either
// output may be absent (not wired by the graph).
template <class Px>
__global__ __launch_bounds__(256) void split_luma_kernel(const char* src, size_t src_pitch, char* luma, size_t luma_pitch, char* chroma,
                                                         size_t chroma_pitch, int W, int y0, int y1)
{
    const int x = (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= W) return;
    const unsigned xoff = (unsigned)x * (unsigned)Px::BPP;
    for (int y = y0 + (int)blockIdx.y; y < y1; y += (int)gridDim.y) {
        const f4 v = Px::decode(Px::load(src + (ptrdiff_t)y * (ptrdiff_t)src_pitch, xoff));
        const float l = fmaf(0.0722f, v.z, fmaf(0.7152f, v.y, 0.2126f * v.x));
        if (luma) Px::store(luma + (ptrdiff_t)y * (ptrdiff_t)luma_pitch, xoff, make_float4(l, l, l, v.w));
        if (chroma)
            Px::store(chroma + (ptrdiff_t)y * (ptrdiff_t)chroma_pitch, xoff, make_float4(fmaf(0.5f, v.x - l, 0.5f), fmaf(0.5f, v.y - l, 0.5f), fmaf(0.5f, v.z - l, 0.5f), v.w));
    }
}

static dim3 fill_grid(int W, int rows)
{
    unsigned gy = (unsigned)(rows < 1 ? 1 : (rows > 1024 ? 1024 : rows));
    return dim3((unsigned)((W + 255) / 256), gy);
}

hipError_t launch_fill_synthetic(int fmt, Image dst, int W, int y_begin, int y_end, int y_global0, uint32_t seed,
                                 hipStream_t stream)
{
    if (y_end <= y_begin || W <= 0) return hipSuccess;
    dim3 grid = fill_grid(W, y_end - y_begin);
    char* d = static_cast<char*>(dst.base);
    if (fmt == kFmtRGBA8)
        hipLaunchKernelGGL((fill_synthetic_kernel<PxU8>), grid, dim3(256), 0, stream, d, dst.pitch, W, y_begin, y_end, y_global0, seed);
    else
        hipLaunchKernelGGL((fill_synthetic_kernel<PxF32>), grid, dim3(256), 0, stream, d, dst.pitch, W, y_begin, y_end, y_global0, seed);
    return hipGetLastError();
}

hipError_t launch_fill_structured(int fmt, Image dst, int W, int y_begin, int y_end, int y_global0, int Hfull,
                                  hipStream_t stream)
{
    if (y_end <= y_begin || W <= 0) return hipSuccess;
    dim3 grid = fill_grid(W, y_end - y_begin);
    char* d = static_cast<char*>(dst.base);
    if (fmt == kFmtRGBA8)
        hipLaunchKernelGGL((fill_structured_kernel<PxU8>), grid, dim3(256), 0, stream, d, dst.pitch, W, y_begin, y_end, y_global0, Hfull);
    else
        hipLaunchKernelGGL((fill_structured_kernel<PxF32>), grid, dim3(256), 0, stream, d, dst.pitch, W, y_begin, y_end, y_global0, Hfull);
    return hipGetLastError();
}

hipError_t launch_upload_srgb8(int fmt, const uint8_t* rgba, size_t stride, Image dst, int W, int rows,
                               const float* tables, hipStream_t stream)
{
    if (rows <= 0 || W <= 0) return hipSuccess;
    dim3 grid = fill_grid(W, rows);
    char* d = static_cast<char*>(dst.base);
    if (fmt == kFmtRGBA8)
        hipLaunchKernelGGL((upload_srgb8_kernel<PxU8>), grid, dim3(256), 0, stream, rgba, stride, d, dst.pitch, W, rows, tables);
    else
        hipLaunchKernelGGL((upload_srgb8_kernel<PxF32>), grid, dim3(256), 0, stream, rgba, stride, d, dst.pitch, W, rows, tables);
    return hipGetLastError();
}

hipError_t launch_download_srgb8(int fmt, Image src, uint8_t* rgba, size_t stride, int W, int rows,
                                 const float* tables, hipStream_t stream)
{
    if (rows <= 0 || W <= 0) return hipSuccess;
    dim3 grid = fill_grid(W, rows);
    const char* s = static_cast<const char*>(src.base);
    if (fmt == kFmtRGBA8)
        hipLaunchKernelGGL((download_srgb8_kernel<PxU8>), grid, dim3(256), 0, stream, s, src.pitch, rgba, stride, W, rows, tables);
    else
        hipLaunchKernelGGL((download_srgb8_kernel<PxF32>), grid, dim3(256), 0, stream, s, src.pitch, rgba, stride, W, rows, tables);
    return hipGetLastError();
}

hipError_t launch_mix(int fmt, Image a, Image b, Image dst, const Geom& g, float mix, hipStream_t stream)
{
    if (g.y1 <= g.y0 || g.W <= 0) return hipSuccess;
    dim3 grid = fill_grid(g.W, g.y1 - g.y0);
    const char* pa = static_cast<const char*>(a.base);
    const char* pb = static_cast<const char*>(b.base);
    char* pd = static_cast<char*>(dst.base);
    if (fmt == kFmtRGBA8)
        hipLaunchKernelGGL((mix_kernel<PxU8>), grid, dim3(256), 0, stream, pa, a.pitch, pb, b.pitch, pd, dst.pitch, g.W, g.y0, g.y1, mix);
    else
        hipLaunchKernelGGL((mix_kernel<PxF32>), grid, dim3(256), 0, stream, pa, a.pitch, pb, b.pitch, pd, dst.pitch, g.W, g.y0, g.y1, mix);
    return hipGetLastError();
}

hipError_t launch_split_luma(int fmt, Image src, Image luma, Image chroma, const Geom& g, hipStream_t stream)
{
    if (g.y1 <= g.y0 || g.W <= 0 || (!luma.base && !chroma.base)) return hipSuccess;
    dim3 grid = fill_grid(g.W, g.y1 - g.y0);
    const char* ps = static_cast<const char*>(src.base);
    char *pl = static_cast<char*>(luma.base), *pc = static_cast<char*>(chroma.base);
    if (fmt == kFmtRGBA8)
        hipLaunchKernelGGL((split_luma_kernel<PxU8>), grid, dim3(256), 0, stream, ps, src.pitch, pl, luma.pitch, pc, chroma.pitch, g.W, g.y0, g.y1);
    else
        hipLaunchKernelGGL((split_luma_kernel<PxF32>), grid, dim3(256), 0, stream, ps, src.pitch, pl, luma.pitch, pc, chroma.pitch, g.W, g.y0, g.y1);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Host-side parameter derivation (the product's own; the oracle restates it)
// ---------------------------------------------------------------------------------
void gaussian_weights(float sigma, int radius, float* w)
{
    if (!(sigma > 0.0f)) {
        w[0] = 1.0f;
        for (int i = 1; i <= radius; ++i) w[i] = 0.0f;
        return;
    }
    double e[kMaxRadius + 1];
    const double s2 = 2.0 * (double)sigma * (double)sigma;
    for (int i = 0; i <= radius; ++i) e[i] = exp(-(double)(i * i) / s2);
    double sum = e[0];
    for (int i = 1; i <= radius; ++i) sum += 2.0 * e[i];
    for (int i = 0; i <= radius; ++i) w[i] = (float)(e[i] / sum);
}

void sharpen_weights(float amount, float* centre, float* side)
{
    *centre = fmaf(4.0f, amount, 1.0f);
    *side = -amount;
}

void default_conv_weights(int K, float sigma, float* w)
{
    const int r = K / 2;
    double g[kMaxRadius + 1];
    if (!(sigma > 0.0f)) {
        g[0] = 1.0;
        for (int i = 1; i <= r; ++i) g[i] = 0.0;
    } else {
        const double s2 = 2.0 * (double)sigma * (double)sigma;
        double sum = 0.0;
        for (int i = 0; i <= r; ++i) g[i] = exp(-(double)(i * i) / s2);
        sum = g[0];
        for (int i = 1; i <= r; ++i) sum += 2.0 * g[i];
        for (int i = 0; i <= r; ++i) g[i] /= sum;
    }
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) w[(dy + r) * K + (dx + r)] = (float)(g[dy < 0 ? -dy : dy] * g[dx < 0 ? -dx : dx]);
}

static double srgb_eotf(double cs) { return cs <= 0.04045 ? cs / 12.92 : pow((cs + 0.055) / 1.055, 2.4); }

void srgb_tables(float* eotf256, float* thr255)
{
    for (int c = 0; c < 256; ++c) eotf256[c] = (float)srgb_eotf((double)c / 255.0);
    for (int q = 0; q < 255; ++q) thr255[q] = (float)srgb_eotf(((double)q + 0.5) / 255.0);
}

}  // namespace rf
